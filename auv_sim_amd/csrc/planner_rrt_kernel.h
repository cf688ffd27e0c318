// planner_rrt_kernel.h -- Planner_RRT (gym_rrt/envs/rrt_dubins.py) on gfx950: the goal-directed RRT
// the RL environment drives one node at a time (RRTEnv.step -> generate_one_node, rrt_env.py:224) and
// that planning() (:162-202) loops over.
//
// One wavefront = one episode; the tree, the (cell, theta-subsection) bucket table and the RNG
// state stay in HBM between launches, so the same kernel serves
//   * planning(max_step): device-side loop, bucket drawn on the device (random.choice, :186)
//   * generate_one_node(bucket): one step per launch with the bucket chosen by the caller (:205)
// Per step: pick a node of the bucket (random.choice :223), steer (:251-289; lane = sub-arc),
// collision + closed-rectangle test (:435-484), grid insert (:108-159), then the arc from the LAST
// list node to the goal (:374-423), sampled 64 points per pass with an early exit at the first
// obstacle / boundary hit.  Obstacle tests use the same suffix-max / squared-threshold form as the
// exploring kernel behind a conservative bounding-box cull: for the steer the lanes are the obstacles
// and loop over the few path points; for the goal arc the lanes are the 64 points of the pass and the
// candidate obstacles are broadcast one by one.
#ifndef AUVP_PLANNER_RRT_KERNEL_H
#define AUVP_PLANNER_RRT_KERNEL_H
#include "auvp_math.h"
#include "auvp_types.h"
#include "auvp_wave.h"
#include "auvp_seed.h"

namespace auvp {

struct PrrtParamsDev {
  double rect[4];
  double exp_rate, dist_to_end, diff_max, freq, cell;
  int32_t S, rows, cols, n_buckets, max_step, flags, step_mode, _pad;
  double delta_theta;
};

// add_node_to_grid's index arithmetic (gym_rrt/envs/rrt_dubins.py:108-159) -- the ONE place every planner kernel takes it from
// (prrt_kernel, prrt_init_kernel, prrt_rows_kernel, prrt_pipe_kernel).  The reference divides the ABSOLUTE coordinates by the
// cell side (:115-116: the boundary's origin is ignored), truncates with int() (toward zero) and indexes Python
// lists with the result, so an index in [-len, -1] wraps to the far end; `>= len` returns without inserting (:118-124: the
// node is in mps_list but in no bucket: -1 here) and `< -len` raises IndexError (`err`).  The subsection: floor(theta /
// delta_theta), S + sub when negative, S -> S - 1 (:127-157), then the same list indexing.
static __device__ __forceinline__ int prrt_bucket_of(const PrrtParamsDev& P, double x, double y, double th, bool& err) {
  int row = (int)(y / P.cell), col = (int)(x / P.cell);
  if (row >= P.rows || col >= P.cols) return -1;  // both `>=` tests precede the first list access; a negative index never passes them
  if (row < 0) row += P.rows;
  if (col < 0) col += P.cols;
  if (row < 0 || col < 0) { err = true; return -1; }
  int sub = (int)auvp_floor(th / P.delta_theta);
  if (sub < 0) sub = (int)(P.S + sub);
  if (sub == P.S) sub -= 1;
  if (sub < 0) { sub += P.S; err |= sub < 0; }
  err |= sub >= P.S;
  return err ? -1 : (row * P.cols + col) * P.S + sub;
}

struct PrrtSummary {  // must match auvp_prrt_summary in include/auvplan.h
  int32_t status, n_nodes, n_points, n_occ, steps, done, path_len, last_node;
  int32_t last_accepted, last_new_node, n_arc, _pad;
  double arc[6];  // x_C, y_C, radius, ang_vel, theta_0, length  (of the successful goal connection)
  double rng_after;
  unsigned long long n_draw32;
};

// One tree node = one 64-byte line (round 4; rounds 1-3 kept node_f / node_i / node_bucket / node_next as four arrays, and
// the L2 hands every store instruction's sectors on to memory, so an accepted node cost four partial-line writes: see
// profiles/r4_config5_traffic.md).  `length` stays 0 (:232) and is not stored.
struct alignas(64) PrrtNode {
  double x, y, theta, t;                  // the state
  int32_t step, parent, pt_off, pt_cnt;   // step it was created in, parent node, its run of path points
  int32_t bucket, next, _p0, _p1;         // its bucket; the member added to that bucket before it (-1: it was the first)
};
static_assert(sizeof(PrrtNode) == 64, "PrrtNode is one line");

struct PrrtBuffers {
  int32_t cap_nodes, cap_points, max_pts, bucket_epoch;
  PrrtNode* nodes;       // [E][cap_nodes]
  int32_t* node_bucket;  // [E][cap_nodes] compact copy of PrrtNode::bucket: what prrt_kernel's bucket-id scan reads (written by prrt_kernel only)
  double* points;        // [E][cap_points][4] x, y, theta, traj_t of the stored path points (a node's run is contiguous)
  int32_t* occupied;     // [E][cap_nodes]
  // [E][n_buckets] {epoch << 24 | len(node_array), head}: the bucket's size -- valid when its top byte equals bucket_epoch,
  // else the bucket is empty: a new batch bumps the epoch instead of clearing the table (a clear every 255 batches) -- and
  // the last node added to it.  The buckets' member lists run newest first: head, then PrrtNode::next.  Member k of the
  // reference's node_array (creation order) is count - 1 - k steps from the head.
  int2* buckets;
  uint32_t* mt;          // [E][624] generator words (lazy in-place format between launches)
  int32_t* rng_state;    // [E][4] pslot, avail, drawn_lo, drawn_hi
  const double* start;   // [E][4] x, y, theta, traj_t
  const double* goal;    // [E][2]
  const int32_t* step_bucket;  // step mode: [E] bucket id chosen by the caller (<0: skip episode)
  PrrtSummary* summary;  // [E]
  int32_t* st_log;       // optional [E][max_step][8]: bucket, picked, accepted, done, npath, arc_n, arc_free, -
  // ---- device-resident RRTEnv loop (gym_rrt/envs/rrt_env.py:182-247), step mode of prrt_kernel only.  env_flags == 0: a
  // plain planner step.  PRRT_ENV_OUTCOME: the launch also writes the step's reward / done flag (what RRTEnv.step returns);
  // PRRT_ENV_AGENT: a stand-in agent inside the launch picks the bucket (a uniformly random occupied one) instead of
  // step_bucket -- two launches per environment step (this one + the observation arrays) instead of four.
  int32_t env_flags, _pad_env;
  uint8_t* env_done;          // [E] the loop's own "finished" flag (also set when an episode fails on the device)
  long long* env_reward;      // [E]
  uint8_t* env_done_out;      // [E] optional copy of env_done for the caller
  int32_t* env_bucket_out;    // [E] PRRT_ENV_AGENT: the bucket the agent picked (-1: finished environment)
  unsigned long long env_agent_seed;
  int32_t* env_err;           // [2] {status, environment} of the first episode that failed on the device (0: none)
  int32_t* pipe_fail;         // host-mapped word, set to 1 by an episode that ends with AUVP_ST_PIPELINE (null: not reported)
  const uint8_t* redo_mask;   // [E] or null: only episodes whose byte is non-zero are touched (the pipeline fallback's re-run)
  // PRRT_ENV_DELTA: the caller's observation arrays (rrt_env.py:250-295) hold the previous step's observation and this
  // launch updates the entries of the ONE bucket per environment whose node_array grew (a step adds at most one node):
  // rrt_grid[e][b][3] = has_node... -- one launch per environment step instead of a rewrite of all E x n_buckets entries
  double* env_obs_grid;       // [E][n_buckets][4]
  long long* env_obs_has;     // [E][n_buckets] (may be null)
  long long* env_obs_num;     // [E][n_buckets] (may be null)
};
__device__ __forceinline__ int prrt_bucket_count(int2 w, int epoch) {
  return (int)((uint32_t)w.x >> 24) == epoch ? (w.x & 0xffffff) : 0;
}
__device__ __forceinline__ int2 prrt_bucket_word(int count, int head, int epoch) { return make_int2((epoch << 24) | count, head); }
#define PRRT_ENV_OUTCOME 1
#define PRRT_ENV_AGENT 2
#define PRRT_ENV_DELTA 4

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
// The stand-in agent of the device-resident loop (tests and measurements; a real agent writes step_bucket itself): every
// live environment picks one of its occupied buckets uniformly -- entry `want` of the planner's own list of occupied
// buckets, the same set the observation's has_node array marks (rrt_env.py:250-265) -- with a counter-based generator of its
// own (splitmix64 of seed / environment / the environment's step count: the agent's randomness, not the planner's stream;
// the count lives in the episode's record in HBM, so a captured graph of one step draws anew at every replay).  -1 for a
// finished environment (the step skips it), 0 when nothing is occupied.
__device__ __forceinline__ int prrt_agent_pick(const PrrtBuffers& B, int e, bool finished, unsigned long long seed) {
  if (finished) return -1;
  const int n_occ = B.summary[e].n_occ;
  const unsigned long long call = (unsigned long long)(uint32_t)B.summary[e].steps;
  if (n_occ <= 0) return 0;
  const int want = (int)(splitmix64(seed ^ splitmix64(((unsigned long long)e << 32) | (call & 0xffffffffull))) % (unsigned long long)n_occ);
  return B.occupied[(size_t)e * B.cap_nodes + want];
}
// RRTEnv.step's outcome of one environment (rrt_env.py:224-247): R_FOUND_PATH (300) when the goal arc was free,
// R_CREATE_NODE (0) when a node was added, R_INVALID_NODE (-1) otherwise; an environment that had finished before the step,
// or that the agent skipped (bucket < 0), gets reward 0 and keeps its flag; an episode that failed on the device (status < 0:
// capacity, bad bucket, generator phase) is flagged finished, rewarded 0 and reported through env_err.
__device__ __forceinline__ void prrt_env_outcome(const PrrtBuffers& B, int e, bool was, bool skipped, int status, int done, int last_accepted) {
  const bool err = status < 0;
  const bool now = done != 0;
  long long r = 0;
  if (!was && !skipped && !err) r = now ? 300 : (last_accepted ? 0 : -1);
  const uint8_t flag = (was || (!skipped && now) || err) ? 1 : 0;
  B.env_reward[e] = r;
  B.env_done[e] = flag;
  if (B.env_done_out) B.env_done_out[e] = flag;
  if (err && B.env_err && atomicCAS(&B.env_err[0], 0, status) == 0) B.env_err[1] = e;
}

// LDS per wave: the generator state and the path points; the throughput instantiation adds its steer scratch
__host__ __device__ inline int prrt_lds_per_wave(int max_pts, int nfreq, bool lat) {
  const int base = 624 * 4 + ((max_pts * 16 + 15) & ~15);
  if (lat) return base;
  const int C = nfreq < 1 ? 1 : (nfreq > 63 ? 63 : nfreq);
  const int u = (2 * C + 2) * 8;
  const int sb = ((C + 1) * 6 + 4) * 8;  // inc[(C+1)*3], sc[(C+1)*2], phi[C+1], path bounding box [4]
  return base + (((u > sb ? u : sb) + 15) & ~15);
}

// angle_wrap (:425-433)
__device__ __forceinline__ double prrt_angle_wrap(double a) {
  for (int guard = 0; guard < 64; guard++) {
    if (-AUVP_PI <= a && a <= AUVP_PI) return a;
    if (a > AUVP_PI) a += (-2 * AUVP_PI);
    else if (a < -AUVP_PI) a += (2 * AUVP_PI);
    else return a;
  }
  return a;
}

// random._randbelow(n): getrandbits(bit_length(n)) until < n; one 32-bit output per try.
// Eight tries are tempered at once (lanes 0..7); the first success is taken.
__device__ __forceinline__ uint32_t rng_randbelow(WaveRng& r, uint32_t n) {
  const int lane = lane_id();
  const int k = 32 - __clz((int)n);  // bit_length
  for (;;) {
    rng_ensure(r, 8u);
    uint32_t v = 0xffffffffu;
    if (lane < 8) v = rng_word(r, (uint32_t)lane) >> (32 - k);
    unsigned long long okm = wave_ballot(lane < 8 && v < n);
    if (okm) {
      int f = __ffsll((long long)okm) - 1;
      uint32_t res = (uint32_t)__builtin_amdgcn_readlane((int)v, f);
      rng_advance_words(r, (uint32_t)(f + 1));
      return res;
    }
    rng_advance_words(r, 8u);
  }
}

// any of n points (x,y in LDS) inside an obstacle's effective disc?  lanes = obstacles (J each);
// slot j is skipped when no obstacle of the slot can reach the points' bounding box.
template <int J>
__device__ __forceinline__ bool prrt_hits(const double (&ox)[J], const double (&oy)[J], const double (&ot)[J],
                                          const double (&orr)[J], const double (*pts)[2], int n, double bx0, double by0,
                                          double bx1, double by1) {
  // conservative cull (a candidate slot runs the exact test): bounding square of half-width orr >= sqrt(T) against
  // the box around its centre, both inflated by 2^-30 relative (see rrt_explore_kernel.h)
  const double cxm = (bx0 + bx1) * 0.5, cym = (by0 + by1) * 0.5;
  const double slack = 0x1p-30 * (auvp_fabs(bx0) + auvp_fabs(bx1) + auvp_fabs(by0) + auvp_fabs(by1) + 1.0);
  const double hx = (bx1 - bx0) * 0.5 + slack, hy = (by1 - by0) * 0.5 + slack;
  int hit = 0;
#pragma unroll
  for (int j = 0; j < J; j++) {
    const bool cand = !(auvp_fabs(ox[j] - cxm) > hx + orr[j] || auvp_fabs(oy[j] - cym) > hy + orr[j]);
    // lanes = path points: each candidate is broadcast and tested against every point at once
    unsigned long long cm = wave_ballot(cand);
    while (cm) {
      const int c = __ffsll((long long)cm) - 1;
      cm &= cm - 1ull;
      const double cox = readlane_f64(ox[j], c), coy = readlane_f64(oy[j], c), cot = readlane_f64(ot[j], c);
      for (int p = lane_id(); p < n; p += 64) {
        const double2 q = *reinterpret_cast<const double2*>(&pts[p][0]);
        const double ex = q.x - cox, ey = q.y - coy;
        hit |= (ex * ex + ey * ey <= cot) ? 1 : 0;
      }
    }
  }
  return wave_any(hit != 0);
}

// Two register budgets per J: batches that fill the chip run five waves per SIMD (96 VGPRs, a few spilled; six or eight
// spill 35+ and measure slower); batches that leave most SIMDs with one wave (config 4: 512 episodes) are latency runs
// and take the whole register file (LAT: no spills, no scalar spills into vector lanes).
template <int J, bool LAT>
__global__ __launch_bounds__(RRT_WAVES * 64, (LAT ? 1 : (J <= 4 ? 5 : (J <= 8 ? 2 : 1)))) void prrt_kernel(WorldDev W, PrrtParamsDev P, PrrtBuffers B, int n_episodes) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = uni((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  const int ep = (int)blockIdx.x * (int)(blockDim.x >> 6) + wave;  // (RRT_WAVES episodes per workgroup; fewer for small batches)
  if (ep >= n_episodes) return;
  if (B.redo_mask && !uni((int)B.redo_mask[ep])) return;  // the pipeline fallback's re-run: the other episodes are not touched
  const int nfreq = (int)P.freq;
  const int C = nfreq < 1 ? 1 : (nfreq > 63 ? 63 : nfreq);
  const int per_wave = prrt_lds_per_wave(B.max_pts, nfreq, LAT);
  unsigned char* wbase = smem + (size_t)wave * per_wave;
  uint32_t* mt = reinterpret_cast<uint32_t*>(wbase);
  double(*pts)[2] = reinterpret_cast<double(*)[2]>(wbase + 624 * 4);
  // steer scratch of the throughput instantiation, behind the path points
  double* scratch = reinterpret_cast<double*>(wbase + 624 * 4 + ((B.max_pts * 16 + 15) & ~15));
  double* u_win = scratch;                        // [2C+2]
  double* inc = scratch;                          // [(C+1)*3]  aliases u_win
  double* sc = scratch + (size_t)(C + 1) * 3;     // [(C+1)*2]
  double* phi_l = scratch + (size_t)(C + 1) * 5;  // [C+1]
  double* bbox_l = scratch + (size_t)(C + 1) * 6;  // [4] xmin, ymin, xmax, ymax of the steer
  (void)u_win; (void)inc; (void)sc; (void)phi_l; (void)bbox_l;

  int step_bucket = 0;
  const bool env = P.step_mode && B.env_flags != 0;
  bool env_was = false;
  if (P.step_mode) {
    if (env) env_was = uni((int)B.env_done[ep]) != 0;
    if (env && (B.env_flags & PRRT_ENV_AGENT)) {
      step_bucket = uni(prrt_agent_pick(B, ep, env_was, B.env_agent_seed));
      if (lane == 0) B.env_bucket_out[ep] = step_bucket;
    } else {
      step_bucket = uni(B.step_bucket[ep]);
    }
    if (step_bucket < 0 || env_was) {  // skipped by the caller / finished: the episode is not touched at all
      if (env && lane == 0) prrt_env_outcome(B, ep, env_was, true, uni(B.summary[ep].status), 0, 0);
      return;
    }
  }

  double ox[J], oy[J], ot[J], orr[J];
#pragma unroll
  for (int j = 0; j < J; j++) {
    int i = j * 64 + lane;
    bool ok = i < W.n_obstacles;
    ox[j] = ok ? W.ox[i] : 0.0;
    oy[j] = ok ? W.oy[i] : 0.0;
    ot[j] = ok ? W.ot[i] : -1.0;
    orr[j] = ot[j] >= 0.0 ? auvp_sqrt(ot[j]) * (1.0 + 0x1p-30) + 0x1p-40 : -__builtin_inf();
  }

  const int capn = B.cap_nodes;
  const size_t capp = (size_t)B.cap_points;
  PrrtNode* nodes = B.nodes + (size_t)ep * capn;
  int32_t* nbucket = B.node_bucket + (size_t)ep * capn;
  double* ptF = B.points + (size_t)ep * capp * 4;
  int32_t* occupied = B.occupied + (size_t)ep * capn;
  int2* buckets = B.buckets + (size_t)ep * P.n_buckets;
  const int epoch = B.bucket_epoch;
  PrrtSummary& sum = B.summary[ep];
  const double gx = readfirst_f64(B.goal[2 * (size_t)ep]), gy = readfirst_f64(B.goal[2 * (size_t)ep + 1]);
  const bool logst = (P.flags & 1) != 0 && B.st_log != nullptr;

  WaveRng rng;
  rng.s = mt;
  for (int i = lane; i < 624; i += 64) mt[i] = B.mt[(size_t)ep * 624 + i];
  rng.pslot = (uint32_t)uni(B.rng_state[4 * (size_t)ep]);
  rng.avail = (uint32_t)uni(B.rng_state[4 * (size_t)ep + 1]);
  rng.drawn = ((unsigned long long)(uint32_t)uni(B.rng_state[4 * (size_t)ep + 2])) |
              ((unsigned long long)(uint32_t)uni(B.rng_state[4 * (size_t)ep + 3]) << 32);
  wave_sync();

  int n_nodes = uni(sum.n_nodes), n_points = uni(sum.n_points), n_occ = uni(sum.n_occ), step = uni(sum.steps);
  int done = uni(sum.done), status = uni(sum.status);
  int last_accepted = 0, last_new = -1;
  int last_bk = -1, last_cnt = 0;  // bucket of the node the last step added and that bucket's new size (PRRT_ENV_DELTA)
  int prev_n_arc = -1;
  bool have_prev_arc = false;
  const int step_end = P.step_mode ? step + 1 : P.max_step;

  while (status == 0 && !done && step < step_end) {
    // the episode's counters are wave-uniform; saying so keeps them (and everything derived from them) in scalar
    // registers: 156 -> 112 vector registers, four waves per SIMD instead of three
    status = uni(status); done = uni(done); step = uni(step);
    n_nodes = uni(n_nodes); n_points = uni(n_points); n_occ = uni(n_occ);
    // ---------------------------------------------------------------- bucket + node choice
    int b;
    if (P.step_mode) {
      b = step_bucket;
      if (b >= P.n_buckets) { status = -1; break; }
    } else {
      if (n_occ == 0) { status = -1; break; }
      b = uni(occupied[rng_randbelow(rng, (uint32_t)n_occ)]);
    }
    const int2 bw = buckets[b];  // size and head in one read
    const int cnt_b = uni(prrt_bucket_count(bw, epoch));
    const int head_b = bw.y;  // (used only when the bucket is not empty)
    last_accepted = 0; last_new = -1;
    if (cnt_b == 0) {  // generate_one_node on an empty bucket: (False, None) (:214-220, input() not reproduced)
      if (logst && lane == 0) {
        int32_t* l = B.st_log + ((size_t)ep * P.max_step + step) * 8;
        l[0] = b; l[1] = -1; l[2] = 0; l[3] = 0; l[4] = 0; l[5] = -1; l[6] = 0; l[7] = 0;
      }
      step++;
      continue;
    }
    const int rsel = (int)rng_randbelow(rng, (uint32_t)cnt_b);
    int par = -1;
    // the rsel-th node (list order) whose bucket is b.  A few steps along the bucket's member list when the node is
    // near its head ...
    const int hops = cnt_b - 1 - rsel;
    if (hops <= 6) {
      par = uni(head_b);
      for (int q = 0; q < hops; q++) par = uni(nodes[par].next);
    }
    // ... else a scan of the bucket ids: 256 are requested at a time so their loads overlap (one dependent round trip per
    // 256 nodes instead of per 64)
    for (int base = 0, seen = 0; base < n_nodes && par < 0; base += 256) {
      int v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int m = base + 64 * k + lane;
        v[k] = m < n_nodes ? nbucket[m] : -1;  // b >= 0: the filler never matches
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (par < 0) {
          const bool is = v[k] == b;
          const unsigned long long bal = wave_ballot(is);
          const int c = __popcll(bal);
          if (seen + c > rsel) {
            const int want = rsel - seen;
            const unsigned long long sel = wave_ballot(is && (int)__popcll(bal & ((1ull << lane) - 1ull)) == want);
            par = base + 64 * k + (__ffsll((long long)sel) - 1);
          }
          seen += c;
        }
      }
    }
    par = uni(par);
    if (par < 0) { status = -4; break; }

    // ---------------------------------------------------------------- steer (:251-289)
    double cx, cy, cth, ctt;
    {
      const double2 a = *reinterpret_cast<const double2*>(&nodes[par].x);
      const double2 c = *reinterpret_cast<const double2*>(&nodes[par].theta);
      cx = readfirst_f64(a.x); cy = readfirst_f64(a.y); cth = readfirst_f64(c.x); ctt = readfirst_f64(c.y);
    }
    int n_total;
    {
      double u = rng_next_random(rng);
      n_total = uni((int)auvp_floor(py_uniform(0.0, P.freq, u) / 1));
    }
    int cnt = 0;
    if (lane == 0) { pts[0][0] = cx; pts[0][1] = cy; }
    bool cap_err = false;
    // extent of x / y over every prefix position of the steer (= the path points and their parent): the collision cull's box
    double bbx0 = cx, bbx1 = cx, bby0 = cy, bby1 = cy;
    auto lane_f64 = [](double v, int src) {  // v of another lane (per-lane source)
      const long long bits = __double_as_longlong(v);
      const int lo = __shfl((int)(bits & 0xffffffffll), src, 64), hi = __shfl((int)(bits >> 32), src, 64);
      return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    };
    if constexpr (LAT) {
      // latency runs: nothing staged in LDS -- every lane tempers its own two draws, the theta and x/y/t chains are
      // uniform loops over cross-lane reads
      for (int c0 = 0; c0 < n_total; c0 += C) {
        const int n = (n_total - c0) < C ? (n_total - c0) : C;
        rng_ensure(rng, (uint32_t)(4 * n));
        // sub-arc s draws random() numbers 2s (dist) and 2s + 1 (diff) of the chunk: every lane tempers its own two
        const bool active = lane < n;
        double radius = 0.0, phi = 0.0;
        bool taken = false;
        if (active) {
          double dist = py_uniform(0.0, P.dist_to_end, rng_random_at(rng, (uint32_t)(2 * lane)));
          double diff = py_uniform(-P.diff_max, P.diff_max, rng_random_at(rng, (uint32_t)(2 * lane + 1)));
          taken = auvp_fabs(dist) > auvp_fabs(diff);
          if (taken) {
            double s1 = dist + diff, s2 = dist - diff;
            radius = auvp_div_plain(s1 + s2, -s1 + s2);
            phi = auvp_div_plain(s1 + s2, 2 * radius);
          }
        }
        const unsigned long long tmask = wave_ballot(taken);
        // theta = angle_wrap(theta + phi) for the taken sub-arcs only, left to right: a uniform loop over the sub-arcs (phi of
        // lane s read across the wave); lane s keeps the angle after its step, idle lanes the chunk-entry angle
        double th = cth, myth = cth;
        for (int s = 0; s < n; s++) {
          if ((tmask >> s) & 1ull) th = prrt_angle_wrap(th + readlane_f64(phi, s));
          if (lane == s) myth = th;
        }
        double sn, cs;
        auvp_sincos(myth, &sn, &cs);
        double dx = 0.0, dy = 0.0, dt = 0.0;
        {
          // sin / cos of the previous TAKEN sub-arc's angle (lane C is idle: it holds the chunk-entry angle)
          const unsigned long long below = tmask & ((1ull << lane) - 1ull);
          const int prev = below ? (63 - __clzll((long long)below)) : C;
          const double so = lane_f64(sn, prev), co = lane_f64(cs, prev);
          if (taken) {
            dx = radius * (sn - so);
            dy = radius * (-cs + co);
            dt = auvp_sqrt_plain(dx * dx + dy * dy) / 1;
          }
        }
        // x += dx; y += dy; t += dt, left to right (untaken sub-arcs add an exact 0.0): the same uniform loop
        double mx = 0.0, my = 0.0, mt_ = 0.0;
        for (int s = 0; s < n; s++) {
          cx = cx + readlane_f64(dx, s);
          cy = cy + readlane_f64(dy, s);
          ctt = ctt + readlane_f64(dt, s);
          bbx0 = __builtin_fmin(cx, bbx0); bbx1 = __builtin_fmax(cx, bbx1);  // (nothing here is NaN)
          bby0 = __builtin_fmin(cy, bby0); bby1 = __builtin_fmax(cy, bby1);
          if (lane == s) { mx = cx; my = cy; mt_ = ctt; }
        }
        cth = th;
        const int napp = __popcll(tmask);
        if (n_points + cnt + napp > (int)capp || cnt + napp + 2 > B.max_pts) { cap_err = true; break; }
        if (taken) {
          int rank = __popcll(tmask & ((1ull << lane) - 1ull));
          size_t gi = (size_t)(n_points + cnt + rank);
          double2* pr = reinterpret_cast<double2*>(ptF + gi * 4);
          pr[0] = make_double2(mx, my); pr[1] = make_double2(myth, mt_);
          pts[cnt + rank + 1][0] = mx;
          pts[cnt + rank + 1][1] = my;
        }
        cnt += napp;
        rng_advance_words(rng, (uint32_t)(4 * n));
        wave_sync();
      }
    } else {
      // throughput runs: the round-1 steer (draw window, theta chain on one lane, x/y/t chains on three lanes, all through
      // the wave's LDS scratch): fewer instructions per step, more latency
      for (int c0 = 0; c0 < n_total; c0 += C) {
        const int n = (n_total - c0) < C ? (n_total - c0) : C;
        rng_ensure(rng, (uint32_t)(4 * n));
        for (int jj = lane; jj < 2 * n; jj += 64) u_win[jj] = rng_random_at(rng, (uint32_t)jj);
        wave_sync();
        const bool active = lane < n;
        double radius = 0.0, phi = 0.0;
        bool taken = false;
        if (active) {
          double dist = py_uniform(0.0, P.dist_to_end, u_win[2 * lane]);
          double diff = py_uniform(-P.diff_max, P.diff_max, u_win[2 * lane + 1]);
          taken = auvp_fabs(dist) > auvp_fabs(diff);
          if (taken) {
            double s1 = dist + diff, s2 = dist - diff;
            radius = auvp_div_plain(s1 + s2, -s1 + s2);
            phi = auvp_div_plain(s1 + s2, 2 * radius);
          }
        }
        const unsigned long long tmask = wave_ballot(taken);
        wave_sync();
        if (lane <= C) phi_l[lane] = phi;
        wave_sync();
        // theta = angle_wrap(theta + phi) for the taken sub-arcs only, left to right, one lane
        if (lane == 0) {
          double th = cth;
          for (int s = 0; s < n; s++) {
            if ((tmask >> s) & 1ull) th = prrt_angle_wrap(th + phi_l[s]);
            phi_l[s] = th;
          }
        }
        wave_sync();
        const double myth = active ? phi_l[lane] : cth;
        double sn, cs;
        auvp_sincos(myth, &sn, &cs);
        if (lane <= C) { sc[2 * lane] = sn; sc[2 * lane + 1] = cs; }
        wave_sync();
        double dx = 0.0, dy = 0.0, dt = 0.0;
        if (taken) {
          unsigned long long below = tmask & ((1ull << lane) - 1ull);
          int prev = below ? (63 - __clzll((long long)below)) : C;
          double so = sc[2 * prev], co = sc[2 * prev + 1];
          dx = radius * (sn - so);
          dy = radius * (-cs + co);
          dt = auvp_sqrt_plain(dx * dx + dy * dy) / 1;
        }
        if (active) { inc[3 * lane] = dx; inc[3 * lane + 1] = dy; inc[3 * lane + 2] = dt; }
        wave_sync();
        if (lane < 3) {
          double acc = lane == 0 ? cx : (lane == 1 ? cy : ctt);
          // lanes 0/1 also track the extent of x / y over every prefix position (= the path points and their parent)
          double bmin = acc, bmax = acc;
          if (c0 != 0 && lane < 2) { bmin = bbox_l[lane]; bmax = bbox_l[2 + lane]; }
  #pragma unroll 4
          for (int s = 0; s < n; s++) {
            acc = acc + inc[3 * s + lane];
            inc[3 * s + lane] = acc;
            bmin = __builtin_fmin(acc, bmin);  // v_min_f64 / v_max_f64; nothing here is NaN
            bmax = __builtin_fmax(acc, bmax);
          }
          if (lane < 2) { bbox_l[lane] = bmin; bbox_l[2 + lane] = bmax; }
        }
        wave_sync();
        double mx = 0.0, my = 0.0, mt_ = 0.0;
        if (active) { mx = inc[3 * lane]; my = inc[3 * lane + 1]; mt_ = inc[3 * lane + 2]; }
        const int napp = __popcll(tmask);
        if (n_points + cnt + napp > (int)capp || cnt + napp + 2 > B.max_pts) { cap_err = true; break; }
        if (taken) {
          int rank = __popcll(tmask & ((1ull << lane) - 1ull));
          size_t gi = (size_t)(n_points + cnt + rank);
          double2* pr = reinterpret_cast<double2*>(ptF + gi * 4);
          pr[0] = make_double2(mx, my); pr[1] = make_double2(myth, mt_);
          pts[cnt + rank + 1][0] = mx;
          pts[cnt + rank + 1][1] = my;
        }
        cnt += napp;
        if (n > 0) {
          cx = readlane_f64(mx, n - 1); cy = readlane_f64(my, n - 1); ctt = readlane_f64(mt_, n - 1);
          cth = readlane_f64(myth, n - 1);
        }
        rng_advance_words(rng, (uint32_t)(4 * n));
        wave_sync();
      }
      wave_sync();
      if (n_total > 0) { bbx0 = bbox_l[0]; bby0 = bbox_l[1]; bbx1 = bbox_l[2]; bby1 = bbox_l[3]; }
    }
    if (cap_err) { status = -2; break; }
    wave_sync();
    const int P_n = cnt + 1;

    // ---------------------------------------------------------------- check_collision_free (:435-458)
    bool ok;
    {
      // bounding box of the path points (lanes = points)
      bool outside = false;
      for (int p = lane; p < P_n; p += 64) {
        double x = pts[p][0], y = pts[p][1];
        bool wx = (x >= P.rect[0]) && (x <= P.rect[2]);
        bool wy = (y >= P.rect[1]) && (y <= P.rect[3]);
        outside = outside | !(wx && wy);
      }
      const double bx0 = bbx0, by0 = bby0, bx1 = bbx1, by1 = bby1;
      ok = !prrt_hits<J>(ox, oy, ot, orr, pts, P_n, bx0, by0, bx1, by1) && !wave_any(outside);
    }
    int me = -1;
    if (ok) {
      if (n_nodes >= capn) { status = -2; break; }
      me = n_nodes;
      // add_node_to_grid (:108-159).  An index below -len is the reference's IndexError: mps_list holds the node by then
      // (:229-230), so it is stored -- in no bucket -- and the episode ends with AUVP_ERR_ARG before the goal connection
      bool idx_err = false;
      int bk = prrt_bucket_of(P, cx, cy, cth, idx_err);
      const bool bad_idx = wave_any(idx_err);
      bk = uni(bk);
      int2 bwn = make_int2(0, 0);
      if (bk >= 0) bwn = buckets[bk];
      const int c_before = bk >= 0 ? uni(prrt_bucket_count(bwn, epoch)) : -1;
      const int h_before = bk >= 0 ? uni(bwn.y) : -1;
      if (lane < 4) {
        // the node's 64-byte record: lanes 0..3 store one 16-byte quarter each, ONE store instruction for the line
        const int nx = (bk >= 0 && c_before > 0) ? h_before : -1;
        int4 q;
        if (lane == 0) q = make_int4(__double2loint(cx), __double2hiint(cx), __double2loint(cy), __double2hiint(cy));
        else if (lane == 1) q = make_int4(__double2loint(cth), __double2hiint(cth), __double2loint(ctt), __double2hiint(ctt));
        else if (lane == 2) q = make_int4(step, par, n_points, cnt);
        else q = make_int4(bk, nx, 0, 0);
        reinterpret_cast<int4*>(&nodes[me])[lane] = q;
      }
      if (lane == 0) {
        nbucket[me] = bk;
        if (bk >= 0) {
          buckets[bk] = prrt_bucket_word(c_before + 1, me, epoch);
          if (c_before == 0) occupied[n_occ] = bk;  // first node of the bucket (:157-159)
        }
      }
      if (c_before == 0) n_occ++;
      n_nodes++;
      n_points += cnt;
      last_accepted = 1; last_new = me;
      last_bk = bk; last_cnt = c_before + 1;
      if (bad_idx) { status = -1; break; }
    }
    // ---------------------------------------------------------------- connect_to_goal_curve_alt(mps_list[-1]) (:374-423)
    const int last = n_nodes - 1;
    double lx, ly, th0, ltt;
    if (ok) { lx = cx; ly = cy; th0 = cth; ltt = ctt; }
    else {
      const double2 a = *reinterpret_cast<const double2*>(&nodes[last].x);
      const double2 c = *reinterpret_cast<const double2*>(&nodes[last].theta);
      lx = readfirst_f64(a.x); ly = readfirst_f64(a.y); th0 = readfirst_f64(c.x); ltt = readfirst_f64(c.y);
    }
    int n_arc = -1, arc_free = 0;
    // connect_to_goal_curve_alt is a pure function of the newest node, the goal and the obstacles: a step that added
    // no node repeats the previous step's evaluation, which was "not free" (or planning would have ended), so its
    // result is reused instead of sampling the same arc again
    if (!ok && have_prev_arc) n_arc = prev_n_arc;
    else {
      const double theta = auvp_atan2(gy - ly, gx - lx);
      const double diff = prrt_angle_wrap(theta - th0);
      if (!(auvp_fabs(diff) > AUVP_PI / 2)) {
        const double r_G = auvp_hypot(gx - lx, gy - ly);
        const double phi_G = theta;  // same atan2 arguments (:387)
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
            // sample the arc 64 points per pass; stop at the first pass that is not free
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
                bool wx = (ax >= P.rect[0]) && (ax <= P.rect[2]);
                bool wy = (ay >= P.rect[1]) && (ay <= P.rect[3]);
                outside = !(wx && wy);
              }
              if (wave_any(outside)) { free_ = false; break; }
              // Here the lanes hold the POINTS (up to 64 of them), so the roles are swapped with respect to
              // prrt_hits: a conservative box of this piece of the arc picks the candidate obstacles (lanes =
              // obstacles, one ballot per slot), and each candidate is broadcast and tested against all points
              // at once -- a few dozen instructions per candidate instead of a 64-point loop per slot.
              // Box: the samples lie on the circle (centre C, radius |radius|) between the first and the last
              // sample of the pass; an arc of angle d < pi stays within |radius| (1 - cos(d/2)) <= |radius| d^2 / 8
              // of its chord, a longer one within the circle's own box.  Inflated by 2^-30 relative.
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
            arc_free = free_ ? 1 : 0;
            if (free_) {
              done = 1;
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
    (void)ltt;
    prev_n_arc = n_arc; have_prev_arc = true;
    if (logst && lane == 0) {
      int32_t* l = B.st_log + ((size_t)ep * P.max_step + step) * 8;
      l[0] = b; l[1] = par; l[2] = ok ? 1 : 0; l[3] = done; l[4] = P_n; l[5] = n_arc; l[6] = arc_free; l[7] = me;
    }
    step++;
  }

  // ---- persist the episode state for the next launch ----
  const unsigned long long drawn = rng.drawn;
  for (int i = lane; i < 624; i += 64) B.mt[(size_t)ep * 624 + i] = mt[i];
  if (lane == 0) {
    B.rng_state[4 * (size_t)ep] = (int32_t)rng.pslot;
    B.rng_state[4 * (size_t)ep + 1] = (int32_t)rng.avail;
    B.rng_state[4 * (size_t)ep + 2] = (int32_t)(uint32_t)(drawn & 0xffffffffull);
    B.rng_state[4 * (size_t)ep + 3] = (int32_t)(uint32_t)(drawn >> 32);
  }
  // peek the next random() without consuming it (parity probe)
  WaveRng peek = rng;
  wave_sync();
  rng_ensure(peek, 2u);  // may generate ahead in LDS only; the stored words above are untouched
  const double after = rng_random_at(peek, 0u);
  if (lane == 0) {
    sum.status = status; sum.n_nodes = n_nodes; sum.n_points = n_points; sum.n_occ = n_occ; sum.steps = step;
    sum.done = done; sum.last_accepted = last_accepted; sum.last_new_node = last_new;
    sum.rng_after = after; sum.n_draw32 = drawn;
    if (!done) { sum.path_len = 0; }
    if (env) {
      prrt_env_outcome(B, ep, false, false, status, done, last_accepted);
      if ((B.env_flags & PRRT_ENV_DELTA) && last_accepted && last_bk >= 0) {
        // the observation of this environment changes in one bucket: len(node_array), has_node, node count
        const size_t at = (size_t)ep * P.n_buckets + (size_t)last_bk;
        B.env_obs_grid[at * 4 + 3] = (double)last_cnt;
        if (B.env_obs_has) B.env_obs_has[at] = 1;
        if (B.env_obs_num) B.env_obs_num[at] = last_cnt;
      }
    }
  }
}

// generate_final_course(final_node) (:317-327) in the order planning() returns it: final node, arc
// points last to first, then every ancestor segment down to the start.  Element = x,y,theta,traj_t,length.
static __global__ __launch_bounds__(64) void prrt_final_course_kernel(PrrtBuffers B, const int64_t* __restrict__ offsets,
                                                               double* __restrict__ out, int n_episodes) {
  const int ep = blockIdx.x;
  if (ep >= n_episodes) return;
  const int lane = lane_id();
  const PrrtSummary s = B.summary[ep];
  if (!s.done || s.path_len <= 0) return;
  const int capn = B.cap_nodes;
  const size_t capp = (size_t)B.cap_points;
  const PrrtNode* nodes = B.nodes + (size_t)ep * capn;
  const double* ptF = B.points + (size_t)ep * capp * 4;
  double* o = out + 5 * (size_t)offsets[ep];
  const double x_C = s.arc[0], y_C = s.arc[1], radius = s.arc[2], ang_vel = s.arc[3], th0 = s.arc[4];
  const int n_arc = s.n_arc;
  // arc point i sits at element 1 + (n_arc - 1 - i); the final node repeats the last arc point
  for (int i = lane; i < n_arc; i += 64) {
    double sa, ca;
    const double a = ang_vel * i + th0;
    auvp_sincos(a, &sa, &ca);
    double* e = o + 5 * (size_t)(1 + (n_arc - 1 - i));
    e[0] = x_C + radius * sa; e[1] = y_C - radius * ca; e[2] = a; e[3] = 0.0; e[4] = 0.0;
    if (i == n_arc - 1) {
      o[0] = e[0]; o[1] = e[1]; o[2] = a; o[3] = nodes[s.last_node].t; o[4] = s.arc[5];
    }
  }
  if (n_arc == 0 && lane == 0) {
    const PrrtNode& nf = nodes[s.last_node];
    o[0] = nf.x; o[1] = nf.y; o[2] = nf.theta; o[3] = nf.t; o[4] = s.arc[5];
  }
  int pos = 1 + n_arc;
  for (int m = s.last_node;;) {
    const int4 r = *reinterpret_cast<const int4*>(&nodes[m].step);
    if (r.y < 0) break;
    const int cnt = r.w, off = r.z;
    for (int k = lane; k < cnt; k += 64) {
      double* e = o + 5 * (size_t)(pos + (cnt - 1 - k));
      size_t gi = (size_t)off + k;
      const double2 p0 = *reinterpret_cast<const double2*>(ptF + gi * 4), p1 = *reinterpret_cast<const double2*>(ptF + gi * 4 + 2);
      e[0] = p0.x; e[1] = p0.y; e[2] = p1.x; e[3] = p1.y; e[4] = 0.0;
    }
    pos += cnt;
    if (lane == 0) {
      const PrrtNode& nf = nodes[r.y];
      double* e = o + 5 * (size_t)pos;
      e[0] = nf.x; e[1] = nf.y; e[2] = nf.theta; e[3] = nf.t; e[4] = 0.0;
    }
    pos++;
    m = r.y;
  }
}

// ---- pipeline fallback (planner_rrt_host.h): prrt_pipe_kernel is speculative; an episode it gives up on (AUVP_ST_PIPELINE)
// is taken back to where the launch found it and planned again by prrt_kernel.
// Before the launch: the episode's record, generator words and generator position (one workgroup per episode).
static __global__ __launch_bounds__(256) void prrt_snapshot_kernel(PrrtBuffers B, PrrtSummary* __restrict__ snap_sum, int32_t* __restrict__ snap_rng,
                                                             uint32_t* __restrict__ snap_mt) {
  const size_t e = blockIdx.x;
  const int t = threadIdx.x;
  const int32_t* ss = reinterpret_cast<const int32_t*>(B.summary + e);
  int32_t* sd = reinterpret_cast<int32_t*>(snap_sum + e);
  if (t < (int)(sizeof(PrrtSummary) / 4)) sd[t] = ss[t];
  if (t < 4) snap_rng[4 * e + t] = B.rng_state[4 * e + t];
  for (int i = t; i < 624; i += 256) snap_mt[e * 624 + i] = B.mt[e * 624 + i];
}
// After it: one thread per episode.  The tree is append-only (nodes and points past the old counts are ignored); the bucket
// table is not: every insert of the failed launch is taken out again, newest first (a node's `next` is the member that headed
// its bucket before it), then record, generator and position return to the snapshot.  redo_mask[e] = 1 for these episodes.
static __global__ __launch_bounds__(64) void prrt_undo_kernel(PrrtParamsDev P, PrrtBuffers B, int n_episodes, const PrrtSummary* __restrict__ snap_sum,
                                                        const int32_t* __restrict__ snap_rng, const uint32_t* __restrict__ snap_mt,
                                                        uint8_t* __restrict__ redo_mask, int32_t* __restrict__ redo_count) {
  const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (e >= n_episodes) return;
  PrrtSummary& cur = B.summary[e];
  if (cur.status != AUVP_ST_PIPELINE) { redo_mask[e] = 0; return; }
  redo_mask[e] = 1;
  atomicAdd(redo_count, 1);
  const PrrtSummary old = snap_sum[e];
  const PrrtNode* nodes = B.nodes + (size_t)e * B.cap_nodes;
  int2* buckets = B.buckets + (size_t)e * P.n_buckets;
  for (int me = cur.n_nodes - 1; me >= old.n_nodes; me--) {
    const int bk = nodes[me].bucket;
    if (bk < 0) continue;
    const int c = prrt_bucket_count(buckets[bk], B.bucket_epoch);
    buckets[bk] = prrt_bucket_word(c > 0 ? c - 1 : 0, nodes[me].next, B.bucket_epoch);
  }
  cur = old;
  for (int k = 0; k < 4; k++) B.rng_state[4 * (size_t)e + k] = snap_rng[4 * (size_t)e + k];
  for (int i = 0; i < 624; i++) B.mt[(size_t)e * 624 + i] = snap_mt[(size_t)e * 624 + i];
}

// Planner_RRT.__init__ (:34-75): mps_list = [start]; add_node_to_grid(start).  One thread per episode.
static __global__ __launch_bounds__(256) void prrt_init_kernel(PrrtParamsDev P, PrrtBuffers B, int n_episodes, uint8_t* __restrict__ env_done,
                                                        int32_t* __restrict__ env_err) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0 && env_err) { env_err[0] = 0; env_err[1] = 0; }
  if (e >= n_episodes) return;
  if (env_done) env_done[e] = 0;
  const double* st = B.start + 4 * (size_t)e;
  const double sx = st[0], sy = st[1], sth = st[2], stt = st[3];
  // same index arithmetic as the step kernels
  bool err = false;
  int bk = prrt_bucket_of(P, sx, sy, sth, err);
  PrrtNode n0;
  n0.x = sx; n0.y = sy; n0.theta = sth; n0.t = stt;
  n0.step = 0; n0.parent = -1; n0.pt_off = 0; n0.pt_cnt = 0;
  n0.bucket = bk; n0.next = -1; n0._p0 = 0; n0._p1 = 0;
  B.nodes[(size_t)e * B.cap_nodes] = n0;
  B.node_bucket[(size_t)e * B.cap_nodes] = bk;
  if (bk >= 0) {
    B.buckets[(size_t)e * P.n_buckets + bk] = prrt_bucket_word(1, 0, B.bucket_epoch);
    B.occupied[(size_t)e * B.cap_nodes] = bk;
  }
  PrrtSummary s;
  s.status = err ? -1 : 0; s.n_nodes = 1; s.n_points = 0; s.n_occ = bk >= 0 ? 1 : 0; s.steps = 0; s.done = 0;
  s.path_len = 0; s.last_node = 0; s.last_accepted = 0; s.last_new_node = -1; s.n_arc = 0; s._pad = 0;
  for (int i = 0; i < 6; i++) s.arc[i] = 0.0;
  s.rng_after = 0.0; s.n_draw32 = 0ull;
  B.summary[e] = s;
}

// Config 5 glue: particle (f, p) of a filter batch -> Planner_RRT episode f * N + p.
//   goal  = clamp(particle.xy * scale_f + offset_f) into a rectangle (filter f's shark frame mapped into the planner's
//           workspace; the clamp keeps a stray hypothesis inside the boundary)
//   start = one shared AUV state
//   generator = CPython random.seed(seed_base + e): init_by_array over the 32-bit limbs of the seed (the same
//           arithmetic as seed_mt() on the host), one thread per episode
struct PrrtGoalMap {
  double start[4];
  const double* xform;  // [F][4] per filter: gx = x * xform[0] + xform[1];  gy = y * xform[2] + xform[3]
  double clamp[4];  // x0, y0, x1, y1
  unsigned long long seed_base;
  int32_t n_filters, n_particles;
};

__device__ inline void mt_seed_by_array(unsigned long long seed, uint32_t* mt) {
  const uint32_t key[2] = {(uint32_t)(seed & 0xffffffffull), (uint32_t)(seed >> 32)};
  const int klen = key[1] ? 2 : 1;
  mt[0] = 19650218u;
  for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  int i = 1, j = 0;
  for (int k = 624; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    i++; j++;
    if (i >= 624) { mt[0] = mt[623]; i = 1; }
    if (j >= klen) j = 0;
  }
  for (int k = 623; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    i++;
    if (i >= 624) { mt[0] = mt[623]; i = 1; }
  }
  mt[0] = 0x80000000u;
}

// One wavefront seeds 64 generators at once.  init_by_array is a serial recurrence per generator (1 871 dependent steps), so
// the work is one thread per episode -- but with the state column in LDS (a read-modify-write of global memory per step
// made the round-2 kernel 0.3 ms for 12 500 episodes).  Layout mtl[word * 65 + thread]: conflict-free both for the
// per-thread recurrence (bank = word + thread) and for the coalesced write-out (one episode's 624 words by 64 lanes).
constexpr int PRRT_SEED_LDS = MT_SEED_LDS;
static __global__ __launch_bounds__(64) void prrt_from_particles_kernel(PrrtBuffers B, PrrtGoalMap M, const double* __restrict__ pf_state,
                                                                 int n_episodes) {
  extern __shared__ __align__(16) unsigned char seed_smem[];
  uint32_t* mtl = reinterpret_cast<uint32_t*>(seed_smem);
  const int t = (int)threadIdx.x;
  const int e0 = (int)blockIdx.x * 64;
  const int e = e0 + t;
  const bool valid = e < n_episodes;
  if (valid) {
    const int f = e / M.n_particles, p = e - f * M.n_particles;
    const double* st = pf_state + (size_t)f * 5 * M.n_particles;  // SoA [F][5][N]: x, y, ...
    const double* xf = M.xform + 4 * (size_t)f;
    double gx = st[p] * xf[0] + xf[1];
    double gy = st[(size_t)M.n_particles + p] * xf[2] + xf[3];
    gx = gx < M.clamp[0] ? M.clamp[0] : (gx > M.clamp[2] ? M.clamp[2] : gx);
    gy = gy < M.clamp[1] ? M.clamp[1] : (gy > M.clamp[3] ? M.clamp[3] : gy);
    double* g = const_cast<double*>(B.goal) + 2 * (size_t)e;
    g[0] = gx; g[1] = gy;
    double* s = const_cast<double*>(B.start) + 4 * (size_t)e;
    s[0] = M.start[0]; s[1] = M.start[1]; s[2] = M.start[2]; s[3] = M.start[3];
    int32_t* rs = B.rng_state + 4 * (size_t)e;
    rs[0] = 0; rs[1] = 0; rs[2] = 0; rs[3] = 0;  // a freshly seeded generator: nothing generated yet
  }
  // random.seed(seed_base + e) = init_by_array over the 32-bit limbs of the seed (the arithmetic of seed_mt() on the host)
  mt_seed_by_array_column(M.seed_base + (unsigned long long)(valid ? e : 0), mtl + t);
  __syncthreads();
  const int n_here = (n_episodes - e0) < 64 ? (n_episodes - e0) : 64;
  for (int q = 0; q < n_here; q++) {
    uint32_t* dst = B.mt + (size_t)(e0 + q) * 624;
    for (int w = t; w < 624; w += 64) dst[w] = mtl[w * 65 + q];
  }
}

// RRTEnv's per-step observation arrays (gym_rrt/envs/rrt_env.py:250-295), elementwise over
// (episode, bucket): [cell.x, cell.y, subsection.theta, len(node_array)], has_node, node counts.
// The reference rebuilds these three O(#buckets) Python lists after every node.
static __global__ __launch_bounds__(256) void prrt_observation_kernel(PrrtParamsDev P, PrrtBuffers B, const double* __restrict__ thetas,
                                                               int n_episodes, double* __restrict__ rrt_grid,
                                                               long long* __restrict__ has_node, long long* __restrict__ num_nodes) {
  const long long total = (long long)n_episodes * P.n_buckets;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i % P.n_buckets);
    const int k = b % P.S, cell = b / P.S, col = cell % P.cols, row = cell / P.cols;
    const int c = prrt_bucket_count(B.buckets[i], B.bucket_epoch);
    double4 v;
    v.x = P.rect[0] + col * P.cell;  // env_btm_left_corner.x + col * cell_side_length (rrt_dubins.py:91)
    v.y = P.rect[1] + row * P.cell;
    v.z = thetas[k];
    v.w = (double)c;
    reinterpret_cast<double4*>(rrt_grid)[i] = v;
    if (has_node) has_node[i] = c != 0 ? 1 : 0;
    if (num_nodes) num_nodes[i] = c;
  }
}

// ---- device-resident RRTEnv loop (gym_rrt/envs/rrt_env.py:182-247): nothing of a step crosses PCIe -----------------
// The step's outcome and the stand-in agent live inside prrt_kernel's step mode (PrrtBuffers::env_*).  The kernels below
// serve the launches that cannot carry them: the four-episodes-per-wavefront kernel (batches of more than eight
// environments per CU) and callers that want the agent as a launch of its own.
static __global__ __launch_bounds__(256) void prrt_env_outcome_kernel(PrrtBuffers B, int n_episodes, const int32_t* __restrict__ bucket_ids) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_episodes) return;
  const bool was = B.env_done[e] != 0;
  const PrrtSummary& s = B.summary[e];
  const bool skipped = was || (bucket_ids && bucket_ids[e] < 0);
  prrt_env_outcome(B, e, was, skipped, s.status, s.done, s.last_accepted);
}

static __global__ __launch_bounds__(256) void prrt_policy_random_kernel(PrrtBuffers B, int n_episodes, unsigned long long seed,
                                                                 int32_t* __restrict__ bucket_out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_episodes) return;
  bucket_out[e] = prrt_agent_pick(B, e, B.env_done[e] != 0, seed);
}

}  // namespace auvp
#endif
