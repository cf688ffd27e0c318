// planner_goal_arc.h -- connect_to_goal_curve_alt (gym_rrt/envs/rrt_dubins.py:374-423) evaluated by ONE wavefront and written to
// the episode's record: the goal connection of the multi-wavefront planner kernel's main stage (planner_pipe_kernel.h: the
// first step of a launch, which has no verdict from the goal-arc stage yet).  Rounds 3-4 kept a two / three-wavefront planner
// kernel (prrt_duo_kernel) in this place; the four-wavefront pipeline replaced it and it was removed in round 5 (it was
// reachable only through an environment knob).
#ifndef AUVP_PLANNER_GOAL_ARC_H
#define AUVP_PLANNER_GOAL_ARC_H
#include "planner_rrt_kernel.h"
#include "rrt_duo_kernel.h"

namespace auvp {

// connect_to_goal_curve_alt(mps_list[-1]) (:374-423) from the node (lx, ly, th0) = node `last`: true when the arc to the goal is
// free -- then the planning is over and the result record gets the arc and the length of the path (the walk to the root).
// n_arc_out: number of arc samples (-1: no arc: bearing error above pi / 2 or degenerate).
template <int J>
__device__ __forceinline__ bool prrt_goal_arc(const PrrtParamsDev& P, const double (&ox)[J], const double (&oy)[J], const double (&ot)[J],
                                              const double (&orr)[J], double gx, double gy, double lx, double ly, double th0,
                                              const PrrtNode* nodes, int last, PrrtSummary& sum, int& n_arc_out) {
  const int lane = lane_id();
  int n_arc = -1;
  bool is_free = false;
  {
      const double theta = auvp_atan2(gy - ly, gx - lx);
      const double diff = prrt_angle_wrap(theta - th0);
      if (!(auvp_fabs(diff) > AUVP_PI / 2)) {
        const double r_G = auvp_hypot(gx - lx, gy - ly);
        const double phi_G = theta;
        if (phi_G - th0 != 0) {
          double phi = 2 * prrt_angle_wrap(phi_G - th0);
          const double sn0 = auvp_sin(phi_G - th0);
          if (sn0 != 0) {
            const double radius = r_G / (2 * sn0);
            double length = radius * phi;
            if (phi > AUVP_PI) { phi -= 2 * AUVP_PI; length = -radius * phi; }
            else if (phi < -AUVP_PI) { phi += 2 * AUVP_PI; length = -radius * phi; }
            const double ang_vel = phi / (length / P.exp_rate);
            double s0, c0;
            auvp_sincos(th0, &s0, &c0);
            const double x_C = lx - radius * s0;
            const double y_C = ly + radius * c0;
            const double ne = auvp_floor(length / P.exp_rate);
            n_arc = (ne >= 0 && ne < 1e8) ? (int)ne + 1 : 0;
            n_arc = uni(n_arc);
            bool free_ = true;
            for (int i0 = 0; i0 < n_arc && free_; i0 += 64) {
              const int nv = (n_arc - i0) < 64 ? (n_arc - i0) : 64;
              const int i = i0 + lane;
              double ax = 0.0, ay = 0.0;
              bool outside = false;
              if (lane < nv) {
                double sa, ca;
                auvp_sincos(ang_vel * i + th0, &sa, &ca);
                ax = x_C + radius * sa;
                ay = y_C - radius * ca;
                const bool wx = (ax >= P.rect[0]) && (ax <= P.rect[2]);
                const bool wy = (ay >= P.rect[1]) && (ay <= P.rect[3]);
                outside = !(wx && wy);
              }
              if (wave_any(outside)) { free_ = false; break; }
              const double rad = auvp_fabs(radius), dth = auvp_fabs(ang_vel) * (double)(nv - 1);
              double bx0, by0, bx1, by1;
              if (dth < AUVP_PI) {
                const double x0 = readlane_f64(ax, 0), y0 = readlane_f64(ay, 0);
                const double x1 = readlane_f64(ax, nv - 1), y1 = readlane_f64(ay, nv - 1);
                double sag = rad * dth * dth * 0.125;
                sag = sag < 2.0 * rad ? sag : 2.0 * rad;
                bx0 = (x0 < x1 ? x0 : x1) - sag; bx1 = (x0 < x1 ? x1 : x0) + sag;
                by0 = (y0 < y1 ? y0 : y1) - sag; by1 = (y0 < y1 ? y1 : y0) + sag;
              } else {
                bx0 = x_C - rad; bx1 = x_C + rad; by0 = y_C - rad; by1 = y_C + rad;
              }
              const double cxm = (bx0 + bx1) * 0.5, cym = (by0 + by1) * 0.5;
              const double slack = 0x1p-30 * (auvp_fabs(bx0) + auvp_fabs(bx1) + auvp_fabs(by0) + auvp_fabs(by1) + rad + 1.0);
              const double hx = (bx1 - bx0) * 0.5 + slack, hy = (by1 - by0) * 0.5 + slack;
              bool hitl = false;
#pragma unroll
              for (int j = 0; j < J; j++) {
                const bool cand = !(auvp_fabs(ox[j] - cxm) > hx + orr[j] || auvp_fabs(oy[j] - cym) > hy + orr[j]);
                unsigned long long cm = wave_ballot(cand);
                while (cm) {
                  const int l = __ffsll((long long)cm) - 1;
                  cm &= cm - 1ull;
                  const double oxl = readlane_f64(ox[j], l), oyl = readlane_f64(oy[j], l), otl = readlane_f64(ot[j], l);
                  const double ex = ax - oxl, ey = ay - oyl;
                  hitl |= (lane < nv) && (ex * ex + ey * ey <= otl);
                }
              }
              if (wave_any(hitl)) free_ = false;
            }
            if (free_) {
              is_free = true;
              int L = 1 + n_arc;
              for (int m = last;;) {
                const int4 r = *reinterpret_cast<const int4*>(&nodes[m].step);
                const int gp = uni(r.y);
                if (gp < 0) break;
                L += uni(r.w) + 1;
                m = gp;
              }
              if (lane == 0) {
                sum.path_len = L; sum.last_node = last; sum.n_arc = n_arc;
                sum.arc[0] = x_C; sum.arc[1] = y_C; sum.arc[2] = radius; sum.arc[3] = ang_vel; sum.arc[4] = th0;
                sum.arc[5] = length;
              }
            }
          }
        }
      }
  }
  n_arc_out = n_arc;
  return is_free;
}

}  // namespace auvp
#endif
