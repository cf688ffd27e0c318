// planner_pipe_kernel.h -- Planner_RRT.planning (gym_rrt/envs/rrt_dubins.py:162-289,374-423) for LATENCY runs (config 4: 512
// episodes, two per CU) as a FEED-FORWARD pipeline of four wavefronts per episode.
//
// A step of the planner is ~11 500 shader clocks of dependent work on one wavefront (tools/prrt_duo_probe.py): bucket choice,
// node pick and the sub-arc draws ~5 000, the steer's chains and the collision test ~3 500, the insert ~2 000, the goal arc
// ~4 000 (6 800 when the step yields a node).  Only the insert needs the tree as it is NOW; everything else is a function of
// the random stream, of node records that never change once written, and of the static obstacles:
//   H  owns the generator: `_randbelow` over the occupied buckets (:186), `_randbelow` over the bucket's members (:223), the walk
//      to the picked node along the bucket's member list, the parent's record, the number of sub-arcs and every sub-arc's
//      dist / diff draws with radius and angle (:262-271)                                                         -> part A
//   S  the theta / x / y / t chains of the steer over the taken sub-arcs (:271-289), check_collision_free (:435-458), the
//      sin / cos of the candidate's heading for G                                                                 -> part B
//   G  connect_to_goal_curve_alt of the CANDIDATE node (:374-423) -- the reference evaluates it on mps_list[-1] right after the
//      insert, and the arc is a function of that node and the obstacles alone -- and, when it is free, the length of the path
//      (the walk from the parent to the root)                                                                     -> part C
//   M  checks that what H looked at is still true, computes the new node's bucket (:291-320), inserts, ends the planning on a
//      free arc.
//   D  (round 6, NW = 5: at most three episodes per workgroup) the sub-arc draws of the step as a wavefront of its own between
//      H and S: H was the slowest stage (5 700 clocks of work per step against 4 800-5 000 for S and G) and ~850 of them were
//      the tempering of the 2 n sub-arc draws, their dist / diff and the two quotients -- none of which H's own chain needs
//      (the stream advances by 4 n words whatever is taken).  H now hands over the ring slot of the first sub-arc word
//      (`sa_cslot`; the ring keeps every word since the step M is at) and moves on; D turns it into radius / phi / tmask.
// Step k + 3 is drawn while k + 2 is steered, k + 1 has its arc tested and k is inserted: the stages hand a ring of eight slots
// along, H at most five steps ahead of M (tags {redo epoch, step + 1}, written last; a reader copies its part out and re-checks
// the tag, a writer clears the tag before it rewrites a part: a stage still working for an epoch that ended reads consistent --
// if outdated -- values).  Measured: one episode 11.0 -> 5.5 ms per planning(2000), config 4 12.6 -> 7.3 ms.
//
// Why H may run three inserts ahead: an insert changes what H looked at only if it went into the bucket the packet chose AND
// `_randbelow(len(bucket))` now comes out differently (members are kept in creation order, a new one goes to the end: the same
// index is the same node unless the size's bit length changed or a thrown-away try is now below the size), or it occupied a new
// bucket AND `_randbelow(n_occ)` now comes out differently (n_occ's bit length changed, or a thrown-away try is now below n_occ); M keeps the buckets of its last eight inserts
// and checks every insert the packet's snapshot did not know.  On a conflict M starts a new epoch: H rewinds the
// generator to the first word of that step (the ring keeps 1 248 words: every word since the oldest unfinished step's start
// stays available) and the stages start over from there.
//
// Bit-identical to prrt_kernel: trees, bucket lists, counters, paths, generator state and position
// (tests/test_gpu_planner_duo.py, tests/experiments/soak_planner_duo.py).
// Limits (the host falls back to prrt_kernel): plan mode (not the one-step mode of the environment), no step log, freq <= 30,
// <= 256 obstacles, at most four episodes per CU.  An episode whose bounded wait runs out ends with AUVP_ST_PIPELINE and is
// redone by prrt_kernel (planner_rrt_host.h: pipeline fallback).
#ifndef AUVP_PLANNER_PIPE_KERNEL_H
#define AUVP_PLANNER_PIPE_KERNEL_H
#include "planner_goal_arc.h"
#include "rrt_trio_kernel.h"

namespace auvp {

constexpr int PPIPE_EP = 4;    // episodes per workgroup at most (sixteen wavefronts)
#ifndef AUVP_PPIPE_LEAD
#define AUVP_PPIPE_LEAD 5
#endif
#ifndef AUVP_PPIPE_LEAD5
#define AUVP_PPIPE_LEAD5 7   // with the draw wavefront the pipeline is a stage deeper (measured, config 4: 5 -> 6.66 ms, 6 -> 6.57, 7 -> 6.49, 8 -> 6.49-6.53)
#endif
constexpr int PPIPE_RING = 8;  // slots
constexpr int PPIPE_LEAD = AUVP_PPIPE_LEAD;  // steps H may be ahead of M (<= PPIPE_RING): slack between stages whose times vary, against snapshots that age
constexpr int PPIPE_HIST = 8;  // inserts M remembers the bucket of

// EXPERIMENT ONLY (tools/prrt_duo_probe.py; -DAUVP_DUO_DIAG=1: shader clocks a stage spends AT WORK per step, =2: the clocks it
// spends BETWEEN two pieces of work, i.e. waiting for its input)
#ifdef AUVP_DUO_DIAG
#if AUVP_DUO_DIAG == 2
#define PPIPE_DIAG_BEGIN(acc, t0, tprev) const unsigned long long t0 = __builtin_amdgcn_s_memtime(); if (tprev) acc += t0 - tprev;
#define PPIPE_DIAG_END(acc, t0, tprev) tprev = __builtin_amdgcn_s_memtime(); (void)t0;
#else
#define PPIPE_DIAG_BEGIN(acc, t0, tprev) const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define PPIPE_DIAG_END(acc, t0, tprev) acc += __builtin_amdgcn_s_memtime() - t0; (void)tprev;
#endif
#endif

struct PpipeSlot {
  // ---- A (H)
  unsigned long long tagA;
  int ver, n_occ, rmin, status;  // the snapshot it was built from (nodes, occupied buckets); smallest rejected bucket try; 0 / -1 / -4
  int kind, b, par, n_total;     // kind 0: a steer follows; 1: the chosen bucket was empty (the step is used up)
  unsigned long long tmask;
  int cnt_b, rmin2;              // the chosen bucket's size as H read it; smallest rejected member try (0x7fffffff: none)
  int head_b, sa_cslot;          // ... and its newest member as H read it (>= ver: H saw an insert its snapshot does not cover); NW = 5: ring slot of the first sub-arc word
  double px, py, pth, ptt;       // the parent's record
  // ---- A2: the sub-arcs (H itself with four wavefronts per episode; D with five: tagD, and tmask above is D's then)
  unsigned long long tagD;
  double radius[DUO_CS], phi[DUO_CS];
  // ---- B (S)
  unsigned long long tagB;
  int ok, cnt, have_sc, _p1;     // collision free?  path points; sin_c / cos_c are there
  double cx, cy, cth, ctt;       // the candidate node
  double sin_c, cos_c;           // sin / cos of its heading as the steer evaluated them
  double pts[DUO_CS][4];         // its path points (x, y, theta, t)
  // ---- C (G)
  unsigned long long tagC;
  int n_arc, free_, path_len, _p0;
  double arc[6];
};

struct PpipeCtl {  // (first eight words: what the waits look at)
  int ver;         // nodes in the tree (M)
  int n_occ;       // occupied buckets (M)
  int m_done;      // M has finished with the slots of steps < m_done
  int epoch;       // redo epoch (M)
  int restart_k;   // the step a new epoch starts over from (written before the epoch)
  int stop, abort, final_step;  // final_step: the last step M executed (the stream ends after its draws)
  double final_after;
  unsigned long long final_drawn;
  int h_done, s_done, g_done, d_done;
  // H's own: stream position at the first word of the steps in flight (slot k & 7)
  unsigned long long sp_drawn[PPIPE_RING];
  uint32_t sp_cslot[PPIPE_RING];
  double diag[4];
};

// next_nodes > 0: the episode also keeps the member lists' `next` links (4 bytes per node) in LDS -- H walks count - 1 - r of
// them per step, one dependent L2 read each otherwise
// bucket_words > 0 (round 6): ... and a mirror of the bucket table (8 bytes per bucket: count | epoch, newest member) and of the
// list of occupied buckets (4 bytes each, at most min(buckets, nodes) of them): H's `_randbelow` over the occupied buckets is
// followed by two dependent reads -- the list entry, then that bucket's word -- which were L2 round trips
__host__ __device__ inline int ppipe_occ_entries(int n_buckets, int cap_nodes) { return n_buckets < cap_nodes ? n_buckets : cap_nodes; }
__host__ __device__ inline int ppipe_per_episode_bytes(int max_pts, int next_nodes = 0, int bucket_words = 0, int occ_entries = 0) {
  int b = TRIO_GEN * 4;
  b += (next_nodes * 4 + 15) & ~15;
  b += (bucket_words * 8 + 15) & ~15;
  b += (occ_entries * 4 + 15) & ~15;
  b += ((max_pts * 16) + 15) & ~15;
  b += (int)((sizeof(PpipeCtl) + 15) & ~(size_t)15);
  b += PPIPE_RING * (int)((sizeof(PpipeSlot) + 15) & ~(size_t)15);
  return b;
}

struct PpipeView { int ver, n_occ, m_done, epoch, restart_k, stop, abort; };
__device__ __forceinline__ PpipeView ppipe_look(const PpipeCtl* c) {
  __asm__ volatile("" ::: "memory");
  const int4 a = *reinterpret_cast<const int4*>(&c->ver);
  const int4 b = *reinterpret_cast<const int4*>(&c->restart_k);
  __asm__ volatile("" ::: "memory");
  PpipeView v;
  v.ver = uni(a.x); v.n_occ = uni(a.y); v.m_done = uni(a.z); v.epoch = uni(a.w);
  v.restart_k = uni(b.x); v.stop = uni(b.y); v.abort = uni(b.z);
  return v;
}

// random._randbelow(n) on the ring generator; RMIN: also the smallest try that was thrown away (>= n; 0x7fffffff: none).
// `more()` generates another block of words if the ring has room for it (false: it has not).
template <bool RMIN, typename More>
__device__ __forceinline__ uint32_t ppipe_randbelow(RingRng& r, uint32_t n, int& rmin, bool& ok, More&& more) {
  const int lane = lane_id();
  const int k = 32 - __clz((int)n);
  ok = true;
  for (;;) {
    if (r.avail < 8u && !more()) { ok = false; return 0u; }
    uint32_t v = 0xffffffffu;
    if (lane < 8) v = ring_word(r, (uint32_t)lane) >> (32 - k);
    const unsigned long long okm = wave_ballot(lane < 8 && v < n);
    const int f = okm ? (__ffsll((long long)okm) - 1) : 8;
    if (RMIN) {
      // tries in front of the success (all eight when there is none) were >= n: their minimum over lanes 0..7 on the DPP path
      int mine = (lane < f) ? (int)(v & 0x7fffffffu) : 0x7fffffff;
      int t = __builtin_amdgcn_update_dpp(mine, mine, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
      mine = t < mine ? t : mine;
      t = __builtin_amdgcn_update_dpp(mine, mine, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
      mine = t < mine ? t : mine;
      t = __builtin_amdgcn_update_dpp(mine, mine, 0x141, 0xf, 0xf, false);     // row_half_mirror: lane i <-> 7 - i
      mine = t < mine ? t : mine;
      const int m0 = __builtin_amdgcn_readfirstlane(mine);
      rmin = m0 < rmin ? m0 : rmin;
    }
    if (okm) {
      const uint32_t res = (uint32_t)__builtin_amdgcn_readlane((int)v, f);
      ring_advance(r, (uint32_t)(f + 1));
      return res;
    }
    ring_advance(r, 8u);
  }
}

// prrt_angle_wrap with its first pass free of branches (the same operations in the same order: the same double)
__device__ __forceinline__ double ppipe_angle_wrap(double a) {
  const double b = a > AUVP_PI ? a + (-2 * AUVP_PI) : (a < -AUVP_PI ? a + (2 * AUVP_PI) : a);
  if (-AUVP_PI <= b && b <= AUVP_PI) return b;
  return prrt_angle_wrap(b);
}

// connect_to_goal_curve_alt (:374-423) from (lx, ly, th0): the arc and whether it is free; nothing is written.
// out[0..5] = x_C, y_C, radius, ang_vel, th0, length (as the result record holds them); n_arc -1: no arc.
template <int J>
__device__ __forceinline__ bool prrt_goal_arc_eval(const PrrtParamsDev& P, const double (&ox)[J], const double (&oy)[J], const double (&ot)[J],
                                                   const double (&orr)[J], double gx, double gy, double lx, double ly, double th0,
                                                   bool have_sc, double sin_th0, double cos_th0, int& n_arc_out, double (&out)[6]) {
  const int lane = lane_id();
  int n_arc = -1;
  bool is_free = false;
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
        double s0 = sin_th0, c0 = cos_th0;  // (of th0, by auvp_sincos, when the caller has them already)
        if (!have_sc) auvp_sincos(th0, &s0, &c0);
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
          // prrt_hits: a conservative box of this piece of the arc picks the candidate obstacles
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
          out[0] = x_C; out[1] = y_C; out[2] = radius; out[3] = ang_vel; out[4] = th0; out[5] = length;
        }
      }
    }
  }
  n_arc_out = n_arc;
  return is_free;
}

constexpr int PPIPE_EP5 = 3;   // ... with five wavefronts per episode (fifteen wavefronts)
template <int J, int NW>
__global__ __launch_bounds__((NW == 5 ? PPIPE_EP5 : PPIPE_EP) * 64 * NW, 1) void prrt_pipe_kernel(WorldDev W, PrrtParamsDev P, PrrtBuffers B, int n_episodes, int next_lds) {
  static_assert(NW == 4 || NW == 5, "four wavefronts per episode, or five with the draw wavefront");
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = uni((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  const int n_ep_wg = (int)(blockDim.x / (64 * NW));
  const int eidx = wave / NW, role = wave % NW;  // role 0: M, 1: H, 2: S, 3: G, 4: D
  // (`next_lds`: bit 0 = the member links in LDS, bit 1 = the bucket table and the occupied list in LDS)
  const int bk_lds = uni((next_lds >> 1) & 1);
  next_lds = uni(next_lds & 1);
  const int n_occ_cap = ppipe_occ_entries(P.n_buckets, B.cap_nodes);
  const int per_ep = ppipe_per_episode_bytes(B.max_pts, next_lds ? B.cap_nodes : 0, bk_lds ? P.n_buckets : 0, bk_lds ? n_occ_cap : 0);
  unsigned char* eb = smem + (size_t)eidx * per_ep;
  uint32_t* gen = reinterpret_cast<uint32_t*>(eb);
  eb += TRIO_GEN * 4;
  int32_t* mnx = reinterpret_cast<int32_t*>(eb);  // [cap_nodes] when next_lds: node -> the bucket member created before it
  if (next_lds) eb += (B.cap_nodes * 4 + 15) & ~15;
  int2* bkl = reinterpret_cast<int2*>(eb);        // [n_buckets] when bk_lds: the bucket words (M writes both copies)
  if (bk_lds) eb += (P.n_buckets * 8 + 15) & ~15;
  int32_t* occl = reinterpret_cast<int32_t*>(eb);  // [n_occ_cap] when bk_lds: the occupied buckets in the order they were first used
  if (bk_lds) eb += (n_occ_cap * 4 + 15) & ~15;
  double(*spts)[2] = reinterpret_cast<double(*)[2]>(eb);  // S: the candidate's points for the collision test
  eb += ((B.max_pts * 16) + 15) & ~15;
  PpipeCtl* ctl = reinterpret_cast<PpipeCtl*>(eb);
  eb += (sizeof(PpipeCtl) + 15) & ~(size_t)15;
  constexpr int SLOT_STRIDE = (int)((sizeof(PpipeSlot) + 15) & ~(size_t)15);
  unsigned char* slot_base = eb;
  auto slot_of = [&](int k) -> PpipeSlot* { return reinterpret_cast<PpipeSlot*>(slot_base + (size_t)(k & (PPIPE_RING - 1)) * SLOT_STRIDE); };

  const int ep = (int)blockIdx.x * n_ep_wg + eidx;
  const bool valid_ep = ep < n_episodes;
  const size_t eps = (size_t)(valid_ep ? ep : 0);
  const int capn = B.cap_nodes;
  const size_t capp = (size_t)B.cap_points;
  PrrtNode* nodes = B.nodes + eps * capn;
  int32_t* nbucket = B.node_bucket + eps * capn;
  double* ptF = B.points + eps * capp * 4;
  int32_t* occupied = B.occupied + eps * capn;
  int2* buckets = B.buckets + eps * P.n_buckets;
  const int epoch_b = B.bucket_epoch;
  PrrtSummary& sum = B.summary[eps];
  const int step0 = uni(sum.steps);
  if (role == 0) {
    if (lane == 0) {
      ctl->ver = sum.n_nodes; ctl->n_occ = sum.n_occ; ctl->m_done = step0; ctl->epoch = 0; ctl->restart_k = step0; ctl->stop = 0; ctl->abort = 0;
      ctl->final_step = step0 - 1; ctl->final_after = 0.0; ctl->final_drawn = 0ull;
      ctl->h_done = 0; ctl->s_done = 0; ctl->g_done = 0; ctl->d_done = 0;
      for (int q = 0; q < PPIPE_RING; q++) { slot_of(q)->tagA = 0ull; slot_of(q)->tagB = 0ull; slot_of(q)->tagC = 0ull; slot_of(q)->tagD = 0ull; }
    }
  } else if (role == 1) {
    // the stored generator (624 words in place: logical word D + j in slot (pslot + j) mod 624, avail of them not yet consumed)
    // unrolled into the ring: slots 0..623 = words D + avail - 624 .. D + avail - 1
    const int p = uni(B.rng_state[4 * eps]), a = uni(B.rng_state[4 * eps + 1]);
    const int pc = p < 0 ? 0 : (p > 623 ? 623 : p), ac = a < 0 ? 0 : (a > 624 ? 624 : a);
    for (int r = lane; r < 624; r += 64) gen[r] = B.mt[eps * 624 + (size_t)((pc + r + ac) % 624)];
  }
  if (next_lds && valid_ep) {
    const int n0 = uni(sum.n_nodes);
    for (int i = (int)threadIdx.x - eidx * 64 * NW; i < n0 && i < capn; i += 64 * NW) mnx[i] = nodes[i].next;
  }
  if (bk_lds && valid_ep) {
    const int o0 = uni(sum.n_occ);
    for (int i = (int)threadIdx.x - eidx * 64 * NW; i < P.n_buckets; i += 64 * NW) bkl[i] = buckets[i];
    for (int i = (int)threadIdx.x - eidx * 64 * NW; i < o0 && i < n_occ_cap; i += 64 * NW) occl[i] = occupied[i];
  }
  __threadfence_block();
  __syncthreads();
  if (!valid_ep) return;  // (all four wavefronts of the episode: no barrier after this point)
  auto give_up = [&]() { if (lane == 0) duo_poke(&ctl->abort, 1); };

  if (role == 1) {
    // ================================================================================================================ H
    RingRng rng;
    rng.s = gen;
    {
      const int a = uni(B.rng_state[4 * eps + 1]);
      const int ac = a < 0 ? 0 : (a > 624 ? 624 : a);
      rng.gslot = 624u; rng.cslot = (uint32_t)(624 - ac); rng.avail = (uint32_t)ac;
      rng.drawn = ((unsigned long long)(uint32_t)uni(B.rng_state[4 * eps + 2])) | ((unsigned long long)(uint32_t)uni(B.rng_state[4 * eps + 3]) << 32);
    }
    int epoch = 0;
    int cur = step0 - 1;  // the latest packet started
    // stream position at the first word of the packets in flight (slot k & 7): where a new epoch, or the end of the planning, rewinds to
    if (lane == 0)
      for (int q = 0; q < PPIPE_RING; q++) { ctl->sp_cslot[q] = rng.cslot; ctl->sp_drawn[q] = rng.drawn; }
    wave_sync();
    auto sp_get = [&](int kk, uint32_t& cs, unsigned long long& dr) {
      const int q = kk & (PPIPE_RING - 1);
      cs = (uint32_t)uni((int)ctl->sp_cslot[q]);
      const unsigned long long d = ctl->sp_drawn[q];
      dr = ((unsigned long long)(uint32_t)uni((int)(d >> 32)) << 32) | (uint32_t)uni((int)(d & 0xffffffffull));
    };
    auto sp_set = [&](int kk, uint32_t cs, unsigned long long dr) {
      const int q = kk & (PPIPE_RING - 1);
      if (lane == 0) { ctl->sp_cslot[q] = cs; ctl->sp_drawn[q] = dr; }
      wave_sync();
    };
    auto rewind_to = [&](int kk) {
      uint32_t cs; unsigned long long dr;
      sp_get(kk, cs, dr);
      rng.avail = (uint32_t)uni((int)(rng.avail + (uint32_t)(rng.drawn - dr)));
      rng.cslot = cs; rng.drawn = dr;
    };
    // oldest position a rewind may ask for (every word since then stays in the ring): the first word of the step M is at.  Kept
    // lazily -- an older value only keeps more -- and looked up again when the ring seems full.
    unsigned long long floor_drawn = rng.drawn;
    int k = step0;
    auto refresh_floor = [&]() {
      const PpipeView v2 = ppipe_look(ctl);
      const int md = v2.m_done < k ? v2.m_done : k;
      uint32_t cs; unsigned long long dr;
      sp_get(md, cs, dr);
      floor_drawn = dr;
    };
    auto more = [&]() -> bool {
      if ((rng.drawn - floor_drawn) + rng.avail + 64ull > (unsigned long long)TRIO_GEN) {
        refresh_floor();
        if ((rng.drawn - floor_drawn) + rng.avail + 64ull > (unsigned long long)TRIO_GEN) return false;
      }
      ring_generate64(rng);
      return true;
    };
    auto ensure = [&](uint32_t need) -> bool {
      while (rng.avail < need)
        if (!more()) return false;
      return true;
    };
#ifdef AUVP_DUO_DIAG
    unsigned long long diag_h = 0ull, tprev_h = 0ull;
#endif
    for (;;) {
      PpipeView cv;
      {
        int spins = 0;
        for (;;) {
          cv = ppipe_look(ctl);
          if (cv.stop || cv.abort) goto h_end;
          if (cv.epoch != epoch) {  // start over from step restart_k (M waits there: its slot is free)
            epoch = cv.epoch;
            k = cv.restart_k;
            rewind_to(k);
            cur = k - 1;
            continue;
          }
          if (k < P.max_step && k < cv.m_done + (NW == 5 ? AUVP_PPIPE_LEAD5 : PPIPE_LEAD)) break;
          if (++spins > pipe_spin_limit()) { give_up(); goto h_end; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      // ---------------------------------------------------------------- build part A of step k (from the snapshot `cv` just taken)
#ifdef AUVP_DUO_DIAG
      PPIPE_DIAG_BEGIN(diag_h, t_h0, tprev_h)
#endif
      cur = k;
      sp_set(k, rng.cslot, rng.drawn);
      if (rng.avail < 160u) (void)ensure(160u);  // (a step's words, generated in one go while nothing waits on them; as far as the ring has room)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int ver = cv.ver, n_occ = cv.n_occ;
      PpipeSlot* q = slot_of(k);
      if (lane == 0) duo_poke64(&q->tagA, 0ull);  // (a stage of an epoch that ended may still be reading this part)
      int status = 0, kind = 0, b = 0, par = 0, n_total = 0, rmin = 0x7fffffff, rmin2 = 0x7fffffff, cnt_snap = 0, head_snap = 0, sa_cslot = 0;
      unsigned long long tmask = 0ull;
      bool fits = true;
      if (n_occ <= 0) status = -1;
      if (status == 0) {
        bool ok1 = true;
        const uint32_t oi = ppipe_randbelow<true>(rng, (uint32_t)n_occ, rmin, ok1, more);
        fits = ok1;
        if (fits) {
          int bb = bk_lds ? lds_peek(&occl[oi < (uint32_t)n_occ_cap ? oi : 0u]) : duo_ld_i32(occupied + (oi < (uint32_t)capn ? oi : 0u));
          b = uni(bb < 0 ? 0 : (bb >= P.n_buckets ? P.n_buckets - 1 : bb));
          const long long bwl = bk_lds ? (long long)lds_peek64(reinterpret_cast<const unsigned long long*>(bkl + b))
                                       : __builtin_nontemporal_load(reinterpret_cast<const long long*>(buckets + b));
          const int2 bw = make_int2((int)(bwl & 0xffffffffll), (int)(bwl >> 32));
          const int cnt_b = uni(prrt_bucket_count(bw, epoch_b));
          if (cnt_b == 0) kind = 1;
          else {
            bool ok2 = true;
            cnt_snap = cnt_b; head_snap = uni(bw.y);
            const int rsel = (int)ppipe_randbelow<true>(rng, (uint32_t)cnt_b, rmin2, ok2, more);
            fits = ok2;
            if (fits) {
              // the rsel-th member (creation order) of bucket b: count - 1 - rsel steps from the head of its list, or -- further
              // away -- a scan of the bucket ids of the nodes the snapshot knows
              const int hops = cnt_b - 1 - rsel;
              int pv = -1;
              if (hops <= 6) {
                pv = uni(bw.y);
                pv = pv < 0 ? 0 : (pv >= capn ? capn - 1 : pv);
                for (int h = 0; h < hops; h++) {
                  int nx = uni(next_lds ? lds_peek(&mnx[pv]) : duo_ld_i32(&nodes[pv].next));
                  pv = nx < 0 ? 0 : (nx >= capn ? capn - 1 : nx);
                }
              } else {
                const int n_known = ver < capn ? ver : capn;
                for (int base = 0, seen = 0; base < n_known && pv < 0; base += 256) {
                  int v[4];
#pragma unroll
                  for (int c = 0; c < 4; c++) {
                    const int m = base + 64 * c + lane;
                    v[c] = m < n_known ? duo_ld_i32(nbucket + m) : -1;
                  }
#pragma unroll
                  for (int c = 0; c < 4; c++) {
                    if (pv < 0) {
                      const bool is = v[c] == b;
                      const unsigned long long bal = wave_ballot(is);
                      const int cc = __popcll(bal);
                      if (seen + cc > rsel) {
                        const int want = rsel - seen;
                        const unsigned long long sel = wave_ballot(is && (int)__popcll(bal & ((1ull << lane) - 1ull)) == want);
                        pv = base + 64 * c + (__ffsll((long long)sel) - 1);
                      }
                      seen += cc;
                    }
                  }
                }
                pv = uni(pv);
                if (pv < 0) { status = -4; pv = 0; }  // (or a snapshot that raced with an insert: M has the step rebuilt)
              }
              par = pv;
              const double* pr = &nodes[par].x;
              const double p0 = duo_ld_f64(pr), p1 = duo_ld_f64(pr + 1), p2 = duo_ld_f64(pr + 2), p3 = duo_ld_f64(pr + 3);
              fits = ensure(2u);
              if (fits) {
                const double u = ring_random_at(rng, 0u);
                ring_advance(rng, 2u);
                n_total = uni((int)auvp_floor(py_uniform(0.0, P.freq, u) / 1));
                n_total = n_total < 0 ? 0 : (n_total > DUO_MAX_FREQ ? DUO_MAX_FREQ : n_total);
                const int n = n_total;
                fits = ensure((uint32_t)(4 * n));
                if (fits && NW == 5) {
                  // the draw wavefront takes it from here: where the step's sub-arc words start in the ring
                  sa_cslot = (int)rng.cslot;
                  ring_advance(rng, (uint32_t)(4 * n));
                  if (lane == 0) { q->px = p0; q->py = p1; q->pth = p2; q->ptt = p3; }
                } else if (fits) {
                  const bool active = lane < n;
                  double radius = 0.0, phi = 0.0;
                  bool taken = false;
                  if (active) {
                    const double dist = py_uniform(0.0, P.dist_to_end, ring_random_at(rng, (uint32_t)(2 * lane)));
                    const double diff = py_uniform(-P.diff_max, P.diff_max, ring_random_at(rng, (uint32_t)(2 * lane + 1)));
                    taken = auvp_fabs(dist) > auvp_fabs(diff);
                    if (taken) {
                      const double s1 = dist + diff, s2 = dist - diff;
                      radius = auvp_div_plain(s1 + s2, -s1 + s2);
                      phi = auvp_div_plain(s1 + s2, 2 * radius);
                    }
                  }
                  tmask = wave_ballot(taken);
                  ring_advance(rng, (uint32_t)(4 * n));
                  if (lane < DUO_CS) { q->radius[lane] = radius; q->phi[lane] = phi; }
                  if (lane == 0) { q->px = p0; q->py = p1; q->pth = p2; q->ptt = p3; }
                }
              }
            }
          }
        }
      }
      if (!fits) {
        // (a step is a few dozen words; the generator can only run out of room behind a corrupt state)
        give_up();
        goto h_end;
      }
      if (lane == 0) {
        q->ver = ver; q->n_occ = n_occ; q->rmin = rmin; q->status = status; q->kind = kind; q->b = b; q->par = par; q->n_total = n_total;
        q->cnt_b = cnt_snap; q->rmin2 = rmin2; q->head_b = head_snap;
        if (NW == 5) q->sa_cslot = sa_cslot;  // (tmask, radius, phi: the draw wavefront's)
        else q->tmask = tmask;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke64(&q->tagA, duo_tag(epoch, k));
#ifdef AUVP_DUO_DIAG
      PPIPE_DIAG_END(diag_h, t_h0, tprev_h)
#endif
      k++;
    }
  h_end:
    {
      // the generator goes back to HBM where the planning ended -- after the draws of the last step M executed; steps drawn
      // or begun beyond it are undone -- in the in-place form the other kernels continue from: pslot 0, the next
      // min(avail, 624) words in slots 0.., the words before them behind
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int fs = uni(duo_peek(&ctl->final_step));
      if (cur > fs) rewind_to(fs + 1);
      if (rng.avail < 2u) ring_generate64(rng);  // (nothing before this position is asked for again: the oldest block may go)
      const unsigned long long drawn = rng.drawn;
      const uint32_t a_out = rng.avail < 624u ? rng.avail : 624u;
      for (int o = lane; o < 624; o += 64) {
        const uint32_t j = (uint32_t)o < a_out ? (uint32_t)o : (uint32_t)(o + TRIO_GEN - 624);  // (o - 624 mod ring)
        B.mt[eps * 624 + o] = gen[ring_wrap(ring_wrap(rng.cslot + j))];
      }
      if (lane == 0) {
        B.rng_state[4 * eps] = 0;
        B.rng_state[4 * eps + 1] = (int32_t)a_out;
        B.rng_state[4 * eps + 2] = (int32_t)(uint32_t)(drawn & 0xffffffffull);
        B.rng_state[4 * eps + 3] = (int32_t)(uint32_t)(drawn >> 32);
      }
      const double after = ring_random_at(rng, 0u);
      if (lane == 0) { ctl->final_after = after; ctl->final_drawn = drawn; }
#ifdef AUVP_DUO_DIAG
      if (lane == 0) ctl->diag[0] = (double)diag_h;
#endif
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke(&ctl->h_done, 1);
    }
    return;
  }

  if (NW == 5 && role == 4) {
    // ================================================================================================================ D
    // part A2 of step k from part A: the 2 n sub-arc draws (:262-271) out of the generator ring -- H generated them and keeps
    // every word since the step M is at -- their dist / diff, which sub-arcs are taken, radius and angle.  A pure function of
    // (sa_cslot, n_total): no state of its own besides the epoch and the step it is at.
    int epoch = 0, k = step0;
    RingRng rr;
    rr.s = gen; rr.gslot = 0u; rr.cslot = 0u; rr.avail = 0u; rr.drawn = 0ull;
#ifdef AUVP_DUO_DIAG
    unsigned long long diag_d = 0ull, tprev_d = 0ull;
#endif
    for (;;) {
      PpipeSlot* q = nullptr;
      {
        int spins = 0;
        for (;;) {
          q = slot_of(k);
          const unsigned long long tg = duo_peek64(&q->tagA);
          const PpipeView cv = ppipe_look(ctl);
          if (cv.stop || cv.abort) goto d_end;
          if (cv.epoch != epoch) { epoch = cv.epoch; k = cv.restart_k; continue; }
          if (tg == duo_tag(epoch, k)) break;
          if (++spins > pipe_spin_limit()) { give_up(); goto d_end; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef AUVP_DUO_DIAG
      PPIPE_DIAG_BEGIN(diag_d, t_d0, tprev_d)
#endif
      const int a_status = uni(q->status), a_kind = uni(q->kind);
      int n = uni(q->n_total);
      n = n < 0 ? 0 : (n > DUO_MAX_FREQ ? DUO_MAX_FREQ : n);
      int sa = uni(q->sa_cslot);
      sa = sa < 0 ? 0 : (sa >= TRIO_GEN ? TRIO_GEN - 1 : sa);
      __asm__ volatile("" ::: "memory");
      if (duo_peek64(&q->tagA) != duo_tag(epoch, k)) continue;  // rewritten under the copy (a new epoch): look again
      if (lane == 0) duo_poke64(&q->tagD, 0ull);
      double radius = 0.0, phi = 0.0;
      bool taken = false;
      if (a_status == 0 && a_kind == 0 && lane < n) {
        rr.cslot = (uint32_t)sa;
        const double dist = py_uniform(0.0, P.dist_to_end, ring_random_at(rr, (uint32_t)(2 * lane)));
        const double diff = py_uniform(-P.diff_max, P.diff_max, ring_random_at(rr, (uint32_t)(2 * lane + 1)));
        taken = auvp_fabs(dist) > auvp_fabs(diff);
        if (taken) {
          const double s1 = dist + diff, s2 = dist - diff;
          radius = auvp_div_plain(s1 + s2, -s1 + s2);
          phi = auvp_div_plain(s1 + s2, 2 * radius);
        }
      }
      const unsigned long long tmask = wave_ballot(taken);
      if (lane < DUO_CS) { q->radius[lane] = radius; q->phi[lane] = phi; }
      if (lane == 0) q->tmask = tmask;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke64(&q->tagD, duo_tag(epoch, k));
#ifdef AUVP_DUO_DIAG
      PPIPE_DIAG_END(diag_d, t_d0, tprev_d)
#endif
      k++;
    }
  d_end:
#ifdef AUVP_DUO_DIAG
    if (lane == 0) ctl->diag[3] = (double)diag_d;
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) duo_poke(&ctl->d_done, 1);
    return;
  }

  // ================================================================================================== S / G / M: the obstacles
  double ox[J], oy[J], ot[J], orr[J];
#pragma unroll
  for (int j = 0; j < J; j++) {
    const int i = j * 64 + lane;
    const bool ok = i < W.n_obstacles;
    ox[j] = ok ? W.ox[i] : 0.0;
    oy[j] = ok ? W.oy[i] : 0.0;
    ot[j] = ok ? W.ot[i] : -1.0;
    orr[j] = ot[j] >= 0.0 ? auvp_sqrt(ot[j]) * (1.0 + 0x1p-30) + 0x1p-40 : -__builtin_inf();
  }
  const double gx = readfirst_f64(B.goal[2 * eps]), gy = readfirst_f64(B.goal[2 * eps + 1]);
  auto lane_f64 = [](double v, int src) {
    const long long bits = __double_as_longlong(v);
    const int lo = __shfl((int)(bits & 0xffffffffll), src, 64), hi = __shfl((int)(bits >> 32), src, 64);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  };

  if (role == 2) {
    // ================================================================================================================ S
    int epoch = 0, k = step0;
#ifdef AUVP_DUO_DIAG
    unsigned long long diag_s = 0ull, tprev_s = 0ull;
#endif
    for (;;) {
      PpipeSlot* q = nullptr;
      {
        int spins = 0;
        for (;;) {
          q = slot_of(k);
          // (five wavefronts: the draw wavefront's tag -- it is only ever set behind a matching tagA)
          const unsigned long long tg = duo_peek64(NW == 5 ? &q->tagD : &q->tagA);
          const PpipeView cv = ppipe_look(ctl);
          if (cv.stop || cv.abort) goto s_end;
          if (cv.epoch != epoch) { epoch = cv.epoch; k = cv.restart_k; continue; }
          if (tg == duo_tag(epoch, k)) break;
          if (++spins > pipe_spin_limit()) { give_up(); goto s_end; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef AUVP_DUO_DIAG
      PPIPE_DIAG_BEGIN(diag_s, t_s0, tprev_s)
#endif
      // ---- part A, copied out
      const int a_status = uni(q->status), a_kind = uni(q->kind);
      int n_total = uni(q->n_total);
      n_total = n_total < 0 ? 0 : (n_total > DUO_MAX_FREQ ? DUO_MAX_FREQ : n_total);
      unsigned long long tmask;
      {
        // (a scalar: the loops over its bits and the lane reads they index stay on the scalar unit)
        const unsigned long long tv = q->tmask & ((1ull << DUO_MAX_FREQ) - 1ull);
        tmask = ((unsigned long long)(uint32_t)uni((int)(tv >> 32)) << 32) | (uint32_t)uni((int)(tv & 0xffffffffull));
      }
      double cx = readfirst_f64(q->px), cy = readfirst_f64(q->py), cth = readfirst_f64(q->pth), ctt = readfirst_f64(q->ptt);
      double radius = 0.0, phi = 0.0;
      if (lane < DUO_CS) { radius = q->radius[lane]; phi = q->phi[lane]; }
      __asm__ volatile("" ::: "memory");
      if (duo_peek64(&q->tagA) != duo_tag(epoch, k)) continue;  // rewritten under the copy (a new epoch): look again
      if (NW == 5 && duo_peek64(&q->tagD) != duo_tag(epoch, k)) continue;
      if (lane == 0) duo_poke64(&q->tagB, 0ull);
      int ok = 0, cnt = 0, have_sc = 0;
      double sin_c = 0.0, cos_c = 0.0;
      if (a_status == 0 && a_kind == 0) {
        // ---------------------------------------------------------------- steer, the half that needs the parent (:271-289)
        if (lane == 0) { spts[0][0] = cx; spts[0][1] = cy; }
        double bbx0 = cx, bbx1 = cx, bby0 = cy, bby1 = cy;
        if (n_total > 0) {
          const int n = n_total;
          const bool taken = (tmask >> lane) & 1ull;
          // (a sub-arc that is not taken changes nothing, :262-271: the chains run over the taken ones)
          double th = cth, myth = cth;
          for (unsigned long long tm = tmask; tm; tm &= tm - 1ull) {
            const int s = __ffsll((long long)tm) - 1;
            th = ppipe_angle_wrap(th + readlane_f64(phi, s));
            if (lane == s) myth = th;
          }
          (void)n;
          double sn, cs;
          auvp_sincos(myth, &sn, &cs);
          {
            // sin / cos of the candidate's heading for the goal arc (:395-396 takes them of the same angle): the last taken
            // sub-arc's lane, or the idle lane with the entry angle
            const int src = tmask ? (63 - __clzll((long long)tmask)) : (DUO_CS - 1);
            sin_c = readlane_f64(sn, src); cos_c = readlane_f64(cs, src);
            have_sc = 1;
          }
          double dx = 0.0, dy = 0.0, dt = 0.0;
          {
            const unsigned long long below = tmask & ((1ull << lane) - 1ull);
            const int prev = below ? (63 - __clzll((long long)below)) : (DUO_CS - 1);  // lane 31 is idle: the entry angle
            const double so = lane_f64(sn, prev), co = lane_f64(cs, prev);
            if (taken) {
              dx = radius * (sn - so);
              dy = radius * (-cs + co);
              dt = auvp_sqrt_plain(dx * dx + dy * dy) / 1;
            }
          }
          double mx = 0.0, my = 0.0, mt_ = 0.0;
          for (unsigned long long tm = tmask; tm; tm &= tm - 1ull) {
            const int s = __ffsll((long long)tm) - 1;
            cx = cx + readlane_f64(dx, s);
            cy = cy + readlane_f64(dy, s);
            ctt = ctt + readlane_f64(dt, s);
            bbx0 = __builtin_fmin(cx, bbx0); bbx1 = __builtin_fmax(cx, bbx1);
            bby0 = __builtin_fmin(cy, bby0); bby1 = __builtin_fmax(cy, bby1);
            if (lane == s) { mx = cx; my = cy; mt_ = ctt; }
          }
          cth = th;
          cnt = __popcll(tmask);
          if (taken) {
            const int rank = __popcll(tmask & ((1ull << lane) - 1ull));
            double2* pr = reinterpret_cast<double2*>(&q->pts[rank][0]);
            pr[0] = make_double2(mx, my); pr[1] = make_double2(myth, mt_);
            if (rank + 1 < B.max_pts) { spts[rank + 1][0] = mx; spts[rank + 1][1] = my; }
          }
        }
        wave_sync();
        int P_n = cnt + 1;
        P_n = P_n > B.max_pts ? B.max_pts : P_n;  // (more points than a tree may hold: M reports the step, the test's result is not used)
        // ---------------------------------------------------------------- check_collision_free (:435-458)
        {
          bool outside = false;
          for (int p = lane; p < P_n; p += 64) {
            const double x = spts[p][0], y = spts[p][1];
            const bool wx = (x >= P.rect[0]) && (x <= P.rect[2]);
            const bool wy = (y >= P.rect[1]) && (y <= P.rect[3]);
            outside = outside | !(wx && wy);
          }
          ok = (!prrt_hits<J>(ox, oy, ot, orr, spts, P_n, bbx0, bby0, bbx1, bby1) && !wave_any(outside)) ? 1 : 0;
        }
      }
      if (lane == 0) {
        q->ok = ok; q->cnt = cnt; q->have_sc = have_sc;
        q->cx = cx; q->cy = cy; q->cth = cth; q->ctt = ctt; q->sin_c = sin_c; q->cos_c = cos_c;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke64(&q->tagB, duo_tag(epoch, k));
#ifdef AUVP_DUO_DIAG
      PPIPE_DIAG_END(diag_s, t_s0, tprev_s)
#endif
      k++;
    }
  s_end:
#ifdef AUVP_DUO_DIAG
    if (lane == 0) ctl->diag[1] = (double)diag_s;
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) duo_poke(&ctl->s_done, 1);
    return;
  }

  if (role == 3) {
    // ================================================================================================================ G
    int epoch = 0, k = step0;
#ifdef AUVP_DUO_DIAG
    unsigned long long diag_g = 0ull, tprev_g = 0ull;
#endif
    for (;;) {
      PpipeSlot* q = nullptr;
      {
        int spins = 0;
        for (;;) {
          q = slot_of(k);
          const unsigned long long tg = duo_peek64(&q->tagB);
          const PpipeView cv = ppipe_look(ctl);
          if (cv.stop || cv.abort) goto g_end;
          if (cv.epoch != epoch) { epoch = cv.epoch; k = cv.restart_k; continue; }
          if (tg == duo_tag(epoch, k)) break;
          if (++spins > pipe_spin_limit()) { give_up(); goto g_end; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef AUVP_DUO_DIAG
      PPIPE_DIAG_BEGIN(diag_g, t_g0, tprev_g)
#endif
      const int ok = uni(q->ok), cnt = uni(q->cnt), have_sc = uni(q->have_sc);
      int par = uni(q->par);
      const double lx = readfirst_f64(q->cx), ly = readfirst_f64(q->cy), th0 = readfirst_f64(q->cth);
      const double sin_c = readfirst_f64(q->sin_c), cos_c = readfirst_f64(q->cos_c);
      __asm__ volatile("" ::: "memory");
      if (duo_peek64(&q->tagB) != duo_tag(epoch, k)) continue;
      int n_arc = -1, is_free = 0, L = 0;
      double arc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      if (ok) {
        if (prrt_goal_arc_eval<J>(P, ox, oy, ot, orr, gx, gy, lx, ly, th0, have_sc != 0, sin_c, cos_c, n_arc, arc)) {
          is_free = 1;
          // the path: the arc, the candidate node with its points, and the walk from its parent to the root
          L = 1 + n_arc + cnt + 1;
          par = par < 0 ? 0 : (par >= capn ? capn - 1 : par);
          for (int m = par, guard = 0; guard < capn; guard++) {
            const int gp = uni(duo_ld_i32(&nodes[m].parent));
            if (gp < 0) break;
            L += uni(duo_ld_i32(&nodes[m].pt_cnt)) + 1;
            m = gp >= capn ? capn - 1 : gp;
          }
        }
      }
      if (lane == 0) {
        q->n_arc = n_arc; q->free_ = is_free; q->path_len = L;
#pragma unroll
        for (int i = 0; i < 6; i++) q->arc[i] = arc[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke64(&q->tagC, duo_tag(epoch, k));
#ifdef AUVP_DUO_DIAG
      PPIPE_DIAG_END(diag_g, t_g0, tprev_g)
#endif
      k++;
    }
  g_end:
#ifdef AUVP_DUO_DIAG
    if (lane == 0) ctl->diag[2] = (double)diag_g;
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) duo_poke(&ctl->g_done, 1);
    return;
  }

  // ================================================================================================================== M
  int n_nodes = uni(sum.n_nodes), n_points = uni(sum.n_points), n_occ = uni(sum.n_occ), step = step0;
  int done = uni(sum.done), status = uni(sum.status);
  int last_accepted = 0, last_new = -1, my_epoch = 0;
  bool have_prev_arc = false;
  int hist = -2;  // lane l < 8: the bucket of the latest insert whose node index is l mod 8
#ifdef AUVP_DUO_DIAG
  unsigned long long diag_m = 0ull, tprev_m = 0ull;
#endif
  while (status == 0 && !done && step < P.max_step) {
    status = uni(status); done = uni(done); step = uni(step);
    n_nodes = uni(n_nodes); n_points = uni(n_points); n_occ = uni(n_occ);
    PpipeSlot* q = slot_of(step);
    {
      int spins = 0;
      for (;;) {
        if (duo_peek64(&q->tagC) == duo_tag(my_epoch, step)) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          // drawn some inserts ago?  Then only if none of them changed what the choices looked at
          const int pv = uni(q->ver);
          bool conflict = false;
          if (pv != n_nodes) {
            const int pn = uni(q->n_occ), qb = uni(q->b);
            const int newest = n_nodes - 1 - ((n_nodes - 1 - lane) & (PPIPE_HIST - 1));  // the node index lane l's entry belongs to
            // inserts into the chosen bucket since the snapshot: its members are kept in creation order and a new one goes to the
            // end, so `_randbelow(len)` (:223) picks the same node unless the size's bit length changed or a try the packet threw
            // away (>= the old size) is below the new size -- the rule of the bucket choice, one level down
            const int added = __popcll(wave_ballot(lane < PPIPE_HIST && newest >= pv && newest >= 0 && hist == qb));
            const int c0 = uni(q->cnt_b), c2 = c0 + added;
            // (a bucket word read AFTER one of those inserts may be ahead of the records it points to: nothing is published between a
            // record and its bucket word -- such a packet is redone)
            const bool touched = added > 0 && (uni(q->head_b) >= pv || (32 - __clz(c0)) != (32 - __clz(c2)) || uni(q->rmin2) < c2);
            conflict = pv > n_nodes || n_nodes - pv > PPIPE_HIST || uni(q->status) != 0 || touched ||
                       (pn != n_occ && ((32 - __clz(pn)) != (32 - __clz(n_occ)) || uni(q->rmin) < n_occ));
          }
          if (!conflict) break;
          my_epoch++;
          if (lane == 0) { ctl->restart_k = step; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) duo_poke(&ctl->epoch, my_epoch);
        }
        if (uni(duo_peek(&ctl->abort))) { status = AUVP_ST_PIPELINE; break; }
        if (++spins > pipe_spin_limit()) { give_up(); status = AUVP_ST_PIPELINE; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (uni(status)) break;
#ifdef AUVP_DUO_DIAG
    PPIPE_DIAG_BEGIN(diag_m, t_m0, tprev_m)
#endif
    // ---- the slot, copied out; then H may have it back
    const int a_status = uni(q->status), a_kind = uni(q->kind), par = uni(q->par), n_total = uni(q->n_total);
    const int ok = uni(q->ok), cnt = uni(q->cnt);
    const double cx = readfirst_f64(q->cx), cy = readfirst_f64(q->cy), cth = readfirst_f64(q->cth), ctt = readfirst_f64(q->ctt);
    const int is_free = uni(q->free_), n_arc = uni(q->n_arc), path_len = uni(q->path_len);
    double2 pa = make_double2(0.0, 0.0), pb = make_double2(0.0, 0.0);
    if (lane < cnt && lane < DUO_CS) {
      const double2* pr = reinterpret_cast<const double2*>(&q->pts[lane][0]);
      pa = pr[0]; pb = pr[1];
    }
    double arc_l = 0.0;
    if (lane < 6) arc_l = q->arc[lane];
    __asm__ volatile("" ::: "memory");
    if (lane == 0) duo_poke(&ctl->m_done, step + 1);
    if (a_status != 0) { status = a_status; break; }
    last_accepted = 0; last_new = -1;
    if (lane == 0) ctl->final_step = step;  // (this step's draws count from here on)
    if (a_kind == 1) {  // generate_one_node on an empty bucket: (False, None) (:214-220)
      step++;
      continue;
    }
    if (n_total > 0 && (n_points + cnt > (int)capp || cnt + 2 > B.max_pts)) { status = -2; break; }
    int me = -1;
    if (ok) {
      if (n_nodes >= capn) { status = -2; break; }
      // the bucket of the new node (:291-320)
      // (an index below -len is the reference's IndexError: mps_list holds the node by then (:229-230), so it is stored -- in no
      // bucket -- and the episode ends with AUVP_ERR_ARG before the goal connection's verdict)
      int bk = -1;
      bool bad_idx = false;
      {
        bool ie = false;
        bk = prrt_bucket_of(P, cx, cy, cth, ie);
        bad_idx = wave_any(ie);
        bk = uni(bk);
      }
      me = n_nodes;
      int2 bwn = make_int2(0, 0);
      if (bk >= 0) bwn = bk_lds ? bkl[bk] : buckets[bk];
      const int c_before = bk >= 0 ? uni(prrt_bucket_count(bwn, epoch_b)) : -1;
      const int h_before = bk >= 0 ? uni(bwn.y) : -1;
      if (lane < cnt && lane < DUO_CS) {
        double2* pr = reinterpret_cast<double2*>(ptF + (size_t)(n_points + lane) * 4);
        pr[0] = pa; pr[1] = pb;
      }
      if (lane < 4) {
        const int nx = (bk >= 0 && c_before > 0) ? h_before : -1;
        int4 r4;
        if (lane == 0) r4 = make_int4(__double2loint(cx), __double2hiint(cx), __double2loint(cy), __double2hiint(cy));
        else if (lane == 1) r4 = make_int4(__double2loint(cth), __double2hiint(cth), __double2loint(ctt), __double2hiint(ctt));
        else if (lane == 2) r4 = make_int4(step, par, n_points, cnt);
        else r4 = make_int4(bk, nx, 0, 0);
        reinterpret_cast<int4*>(&nodes[me])[lane] = r4;
        if (next_lds && lane == 3) mnx[me] = nx;
      }
      if (lane == 0) {
        nbucket[me] = bk;
        if (bk >= 0) {
          buckets[bk] = prrt_bucket_word(c_before + 1, me, epoch_b);
          if (c_before == 0) occupied[n_occ] = bk;
          if (bk_lds) {  // (the mirrors H reads: published with the insert, by the fence below)
            bkl[bk] = prrt_bucket_word(c_before + 1, me, epoch_b);
            if (c_before == 0 && n_occ < n_occ_cap) occl[n_occ] = bk;
          }
        }
      }
      if (lane == (me & (PPIPE_HIST - 1))) hist = bk;
      if (c_before == 0) n_occ++;
      n_nodes++;
      n_points += cnt;
      last_accepted = 1; last_new = me;
      // the insert is published: record, bucket word and occupied list first, then the counters H's snapshots start from
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) { ctl->n_occ = n_occ; duo_poke(&ctl->ver, n_nodes); }
      if (bad_idx) { status = -1; break; }
      // ---- connect_to_goal_curve_alt(mps_list[-1]) (:374-423): G's verdict for this node
      if (is_free) {
        done = 1;
        if (lane == 0) { sum.path_len = path_len; sum.last_node = me; sum.n_arc = n_arc; }
        if (lane < 6) sum.arc[lane] = arc_l;
      }
      have_prev_arc = true;
    } else if (!have_prev_arc) {
      // no new node and no verdict yet for the newest one (the first step of a launch): the arc from that node, here
      const int last = n_nodes - 1;
      const double2 a = *reinterpret_cast<const double2*>(&nodes[last].x);
      const double lx = readfirst_f64(a.x), ly = readfirst_f64(a.y), th0 = readfirst_f64(nodes[last].theta);
      int na = -1;
      if (prrt_goal_arc<J>(P, ox, oy, ot, orr, gx, gy, lx, ly, th0, nodes, last, sum, na)) done = 1;
      have_prev_arc = true;
    }
    step++;
#ifdef AUVP_DUO_DIAG
    PPIPE_DIAG_END(diag_m, t_m0, tprev_m)
#endif
  }
  // ---- the planning is over: H stores the generator where it ended, M the record ----
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) duo_poke(&ctl->stop, 1);
  {
    int spins = 0;
    while (!uni(duo_peek(&ctl->h_done)) || !uni(duo_peek(&ctl->s_done)) || !uni(duo_peek(&ctl->g_done)) ||
           (NW == 5 && !uni(duo_peek(&ctl->d_done)))) {
      if (++spins > pipe_spin_limit()) { status = AUVP_ST_PIPELINE; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // a stage that gave up (abort) may have posted its final words before this wavefront had finished: the episode's stream
    // position is then not to be trusted even though every step went through -- it is redone like any other failed episode
    if (status == 0 && uni(duo_peek(&ctl->abort))) status = AUVP_ST_PIPELINE;
  }
  if (lane == 0) {
    pipe_report(B.pipe_fail, status);
    sum.status = status; sum.n_nodes = n_nodes; sum.n_points = n_points; sum.n_occ = n_occ; sum.steps = step;
    sum.done = done; sum.last_accepted = last_accepted; sum.last_new_node = last_new;
    sum.rng_after = ctl->final_after; sum.n_draw32 = ctl->final_drawn;
    if (!done) sum.path_len = 0;
#ifdef AUVP_DUO_DIAG
    if (!done) { sum.arc[0] = (double)diag_m; sum.arc[1] = ctl->diag[0]; sum.arc[2] = ctl->diag[1]; sum.arc[3] = ctl->diag[2]; sum.arc[4] = (double)my_epoch; sum.arc[5] = NW == 5 ? ctl->diag[3] : 0.0; }
#endif
  }
}

}  // namespace auvp
#endif
