/* orc_planner.c -- CPU restatement of the goal-directed RRT of the RL environment.
 *
 * TEST INFRASTRUCTURE (oracle/): checker + reported CPU baseline only (see orc_api.h).
 *
 * Follows (paths relative to /root/reference):
 *   Planner_RRT.__init__ / discretize_env / add_node_to_grid  gym_rrt/envs/rrt_dubins.py:34-159
 *   Planner_RRT.planning                                      gym_rrt/envs/rrt_dubins.py:162-202
 *   Planner_RRT.generate_one_node                             gym_rrt/envs/rrt_dubins.py:205-248
 *   Planner_RRT.steer                                         gym_rrt/envs/rrt_dubins.py:251-289
 *   Planner_RRT.connect_to_goal_curve_alt                     gym_rrt/envs/rrt_dubins.py:374-423
 *   Planner_RRT.angle_wrap / check_collision_free / check_within_boundary  :425-484
 *   Planner_RRT.generate_final_course                         gym_rrt/envs/rrt_dubins.py:317-327
 *   Grid_cell_RRT (delta_theta)                               gym_rrt/envs/grid_cell_rrt.py:34-56
 * random.choice -> _randbelow (one 32-bit output per try) is in cpyrandom.h.
 * The blocking input() calls (:151,:219) are not reproduced: an empty bucket returns (False, None)
 * and the "invalid subsection" branch just applies its `-= 1`.
 */
#include <stdlib.h>
#include <string.h>
#include "cpyrandom.h"
#include "orc_api.h"
#include "orc_math.h"

static double angle_wrap(double a) {
  /* recursion of :425-433 as a loop */
  for (;;) {
    if (-M_PI <= a && a <= M_PI) return a;
    if (a > M_PI) a += (-2 * M_PI);
    else if (a < -M_PI) a += (2 * M_PI);
    else return a; /* NaN */
  }
}

/* check_collision_free (:435-458): same shared-dList quirk as RRT.check_collision, then the closed
 * rectangle test of check_within_boundary (:469-484) */
static int collision_free(const orc_world* w, const double* rect, int npts, const double* pts_xy) {
  double run_min = INFINITY;
  for (int k = 0; k < w->n_obstacles; k++) {
    double ox = w->obstacles[3 * k], oy = w->obstacles[3 * k + 1], r = w->obstacles[3 * k + 2];
    for (int i = 0; i < npts; i++) {
      double dx = pts_xy[2 * i] - ox, dy = pts_xy[2 * i + 1] - oy;
      double d = ORC_SQRT(ORC_POW2(dx) + ORC_POW2(dy));
      if (d < run_min) run_min = d;
    }
    if (run_min <= r) return 0;
  }
  for (int i = 0; i < npts; i++) {
    double x = pts_xy[2 * i], y = pts_xy[2 * i + 1];
    int wx = (x >= rect[0]) && (x <= rect[2]);
    int wy = (y >= rect[1]) && (y <= rect[3]);
    if (!(wx && wy)) return 0;
  }
  return 1;
}

/* Python int(y / cs) list index with negative wrap; returns -1 for IndexError */
static int py_index(double v, double cs, int len) {
  int i = (int)(v / cs);
  if (i < 0) { i += len; if (i < 0) return -2; }
  return i;
}

typedef struct {
  const orc_prrt_params* p;
  orc_prrt_out* o;
  int rows, cols, S;
  double delta_theta;
  int n_nodes, n_points, n_occ;
} prrt;

/* add_node_to_grid (:108-159); returns <0 on an IndexError the reference would raise */
static int add_node_to_grid(prrt* t, int node) {
  const double* nd = t->o->nodes + 5 * (size_t)node;
  int row = py_index(nd[1], t->p->cell_side_length, t->rows);
  int col = py_index(nd[0], t->p->cell_side_length, t->cols);
  t->o->node_bucket[node] = -1;
  /* the two `>=` tests (:118-124) come before the first list access (:127), and a negative index never passes them: a row past
   * the top returns quietly even when the column would have raised */
  if (row >= t->rows) return 0; /* "out of the habitat environment bound": not bucketed */
  if (col >= t->cols) return 0;
  if (row == -2 || col == -2) return -1;
  double raw = nd[2] / t->delta_theta;
  double fl = floor(raw);
  int sub = (int)fl;
  if (sub < 0) sub = (int)(t->S + sub);
  if (sub == t->S) sub -= 1;
  if (sub < 0) { sub += t->S; if (sub < 0) return -1; } /* python negative index */
  if (sub >= t->S) return -1;
  int b = (row * t->cols + col) * t->S + sub;
  t->o->node_bucket[node] = b;
  if (++t->o->bucket_counts[b] == 1) t->o->occupied[t->n_occ++] = b;
  return 0;
}

int orc_prrt_planning(const orc_world* w, const orc_prrt_params* p, uint64_t seed, orc_prrt_out* o) {
  cpy_rng rng;
  cpy_seed_u64(&rng, seed);
  prrt T;
  memset(&T, 0, sizeof T);
  T.p = p; T.o = o;
  T.S = p->subsections;
  /* discretize_env (:77-93) */
  double env_w = p->rect[2] - p->rect[0], env_h = p->rect[3] - p->rect[1];
  int ics = (int)p->cell_side_length;
  if (ics <= 0 || T.S <= 0) return ORC_ERR_ARG;
  T.rows = (int)env_h / ics;
  T.cols = (int)env_w / ics;
  T.delta_theta = (double)(2.0 * M_PI) / (double)T.S;
  o->grid_rows = T.rows; o->grid_cols = T.cols;
  const int n_buckets = T.rows * T.cols * T.S;
  if (n_buckets > o->cap_buckets || o->cap_nodes < 1) return ORC_ERR_CAPACITY;
  memset(o->bucket_counts, 0, sizeof(int32_t) * (size_t)n_buckets);
  /* mps_list = [start] */
  double* n0 = o->nodes;
  n0[0] = p->start[0]; n0[1] = p->start[1]; n0[2] = p->start[2]; n0[3] = p->start[3]; n0[4] = 0.0;
  o->parent[0] = -1; o->pt_off[0] = 0; o->pt_cnt[0] = 0;
  T.n_nodes = 1;
  int status = ORC_OK;
  if (add_node_to_grid(&T, 0) < 0) status = ORC_ERR_ARG;
  const int max_pts = (int)p->freq + 2;
  double* path_xy = (double*)malloc(sizeof(double) * 2 * (size_t)(max_pts + 1));
  double* tmp = (double*)malloc(sizeof(double) * 4 * (size_t)(max_pts + 1));
  double* arc = NULL; /* x,y,theta per arc point */
  double* arc_xy = NULL;
  int arc_cap = 0;
  int step = 0, done = 0;
  o->path_len = 0;
  for (; status == ORC_OK && step < p->max_step && !done;) {
    /* planning(): random.choice(occupied_grid_cells_array) (:186) */
    if (T.n_occ == 0) { status = ORC_ERR_ARG; break; } /* IndexError in the reference */
    int b = o->occupied[cpy_randbelow(&rng, (uint32_t)T.n_occ)];
    /* generate_one_node: random.choice(grid_cell.node_array) (:223) */
    int cnt = o->bucket_counts[b];
    int r = (int)cpy_randbelow(&rng, (uint32_t)cnt);
    int par = -1;
    for (int m = 0, seen = 0; m < T.n_nodes; m++) {
      if (o->node_bucket[m] == b) { if (seen == r) { par = m; break; } seen++; }
    }
    /* steer (:251-289): velocity = 1, every taken sub-arc is appended, theta is wrapped */
    const double* pn = o->nodes + 5 * (size_t)par;
    double x = pn[0], y = pn[1], th = pn[2], tt = pn[3];
    int n_expand = (int)floor(cpy_uniform(&rng, 0.0, p->freq) / 1);
    int c = 0;
    path_xy[0] = x; path_xy[1] = y;
    for (int s = 0; s < n_expand; s++) {
      double dist = cpy_uniform(&rng, 0.0, p->dist_to_end);
      double diff = cpy_uniform(&rng, -p->diff_max, p->diff_max);
      if (fabs(dist) > fabs(diff)) {
        double s1 = dist + diff, s2 = dist - diff;
        double radius = (s1 + s2) / (-s1 + s2);
        double phi = (s1 + s2) / (2 * radius);
        double ori = th;
        th = angle_wrap(th + phi);
        double dx = radius * (ORC_SIN(th) - ORC_SIN(ori));
        double dy = radius * (-ORC_COS(th) + ORC_COS(ori));
        x += dx;
        y += dy;
        tt += (ORC_SQRT(ORC_POW2(dx) + ORC_POW2(dy))) / 1;
        double* q = tmp + 4 * (size_t)c;
        q[0] = x; q[1] = y; q[2] = th; q[3] = tt;
        c++;
        path_xy[2 * c] = x; path_xy[2 * c + 1] = y;
      }
    }
    int ok = collision_free(w, p->rect, c + 1, path_xy);
    o->st_bucket[step] = b; o->st_picked[step] = par; o->st_accepted[step] = (int8_t)ok;
    o->st_npath[step] = c + 1; o->st_arc_n[step] = -1; o->st_arc_free[step] = 0; o->st_done[step] = 0;
    if (ok) {
      if (T.n_nodes >= o->cap_nodes || T.n_points + c > o->cap_points) { status = ORC_ERR_CAPACITY; break; }
      int me = T.n_nodes++;
      double* nn = o->nodes + 5 * (size_t)me;
      nn[0] = x; nn[1] = y; nn[2] = th; nn[3] = tt; nn[4] = 0.0; /* length += parent.length == 0 (:232) */
      o->parent[me] = par; o->pt_off[me] = T.n_points; o->pt_cnt[me] = c;
      memcpy(o->points + 4 * (size_t)T.n_points, tmp, sizeof(double) * 4 * (size_t)c);
      T.n_points += c;
      if (add_node_to_grid(&T, me) < 0) { status = ORC_ERR_ARG; break; }
    }
    /* connect_to_goal_curve_alt(self.mps_list[-1]) (:237,:374-423) */
    const int last = T.n_nodes - 1;
    const double* ln = o->nodes + 5 * (size_t)last;
    const double lx = ln[0], ly = ln[1], theta_0 = ln[2];
    int have_arc = 0, n_arc = 0;
    double arc_len = 0.0;
    {
      double theta = ORC_ATAN2(p->goal[1] - ly, p->goal[0] - lx);
      double diff = angle_wrap(theta - theta_0);
      if (!(fabs(diff) > M_PI / 2)) {
        double r_G = ORC_HYPOT(p->goal[0] - lx, p->goal[1] - ly);
        double phi_G = ORC_ATAN2(p->goal[1] - ly, p->goal[0] - lx);
        if (phi_G - theta_0 != 0) {
          double phi = 2 * angle_wrap(phi_G - theta_0);
          double sn = ORC_SIN(phi_G - theta_0);
          if (sn != 0) {
            double radius = r_G / (2 * sn);
            double length = radius * phi;
            if (phi > M_PI) { phi -= 2 * M_PI; length = -radius * phi; }
            else if (phi < -M_PI) { phi += 2 * M_PI; length = -radius * phi; }
            double ang_vel = phi / (length / p->exp_rate);
            double x_C = lx - radius * ORC_SIN(theta_0);
            double y_C = ly + radius * ORC_COS(theta_0);
            double ne = floor(length / p->exp_rate);
            n_arc = (ne >= 0 && ne < 1e8) ? (int)ne + 1 : 0;
            if (n_arc > arc_cap) {
              arc_cap = 2 * n_arc;
              arc = (double*)realloc(arc, sizeof(double) * 3 * (size_t)arc_cap);
              arc_xy = (double*)realloc(arc_xy, sizeof(double) * 2 * (size_t)arc_cap);
            }
            for (int i = 0; i < n_arc; i++) {
              double a = ang_vel * i + theta_0;
              arc[3 * i] = x_C + radius * ORC_SIN(a);
              arc[3 * i + 1] = y_C - radius * ORC_COS(a);
              arc[3 * i + 2] = a;
              arc_xy[2 * i] = arc[3 * i]; arc_xy[2 * i + 1] = arc[3 * i + 1];
            }
            have_arc = 1;
            arc_len = length;
          }
        }
      }
    }
    if (have_arc) {
      o->st_arc_n[step] = n_arc;
      /* an empty dList makes min() raise in the reference when there are obstacles; n_arc >= 1 here */
      int fr = collision_free(w, p->rect, n_arc, arc_xy);
      o->st_arc_free[step] = (int8_t)fr;
      if (fr) {
        done = 1;
        o->st_done[step] = 1;
        /* generate_final_course(final_node) (:317-327): final node, its arc points last to first,
         * then every ancestor segment; elements = x,y,theta,traj_t,length */
        int L = 1 + n_arc;
        for (int m = last; o->parent[m] >= 0; m = o->parent[m]) L += o->pt_cnt[m] + 1;
        o->path_len = L;
        if (L > o->cap_path) { status = ORC_ERR_CAPACITY; break; }
        double* e = o->path;
        /* final node: position/angle of the last arc point, traj_t of the node it grew from */
        e[0] = n_arc ? arc[3 * (n_arc - 1)] : lx; e[1] = n_arc ? arc[3 * (n_arc - 1) + 1] : ly;
        e[2] = n_arc ? arc[3 * (n_arc - 1) + 2] : theta_0; e[3] = ln[3]; e[4] = arc_len;
        e += 5;
        for (int i = n_arc - 1; i >= 0; i--, e += 5) { e[0] = arc[3 * i]; e[1] = arc[3 * i + 1]; e[2] = arc[3 * i + 2]; e[3] = 0.0; e[4] = 0.0; }
        for (int m = last; o->parent[m] >= 0; m = o->parent[m]) {
          for (int k = o->pt_cnt[m] - 1; k >= 0; k--, e += 5) {
            const double* q = o->points + 4 * (size_t)(o->pt_off[m] + k);
            e[0] = q[0]; e[1] = q[1]; e[2] = q[2]; e[3] = q[3]; e[4] = 0.0;
          }
          const double* q = o->nodes + 5 * (size_t)o->parent[m];
          e[0] = q[0]; e[1] = q[1]; e[2] = q[2]; e[3] = q[3]; e[4] = q[4];
          e += 5;
        }
      }
    }
    step++;
  }
  o->steps = step; o->done = done; o->n_nodes = T.n_nodes; o->n_points = T.n_points; o->n_occ = T.n_occ;
  o->n_buckets = n_buckets;
  o->n_draw32 = rng.n_draw32; /* outputs consumed by the planner itself */
  o->rng_after = cpy_random(&rng);
  o->status = status;
  free(path_xy); free(tmp); free(arc); free(arc_xy);
  return status;
}
