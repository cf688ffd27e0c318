// astar_kernel.h -- the four runnable A* variants of path_planning/astar*.py on gfx950, one wavefront
// per search instance (start / goal / length limit), E instances per launch over one shared world.
//
//   variant 0  astar.py           box bounds, squared-distance g/h, exact-goal stop        (:193-271)
//   variant 1  astar_real.py      polygon-fan bounds, stop within 10 m of the goal        (:144-222)
//   variant 2  astar_fixLen.py    fixed path length, habitat-coverage cost, visited bitmap (:286-418)
//   variant 3  astar_fixLenSOG.py fixLen + shark-occupancy reward / top-n heuristic        (:551-657)
//
// The search itself is a serial chain (pop the first minimum f, expand 8 neighbours); the lanes
// split the work inside one expansion:
//   open-set pop        min over (f, index) of a compact open list kept in LDS (768 entries per instance); an
//                       instance that outgrows it scans the open flags of every node in HBM instead
//   bounds + collision  lane = (neighbour k, slice s): 8 x 8 lanes cover the 8 neighbours, each slice
//                       takes every 8th fan triangle / obstacle
//   SOG cell lookup     one sweep over the cells per expansion, 64 cells per pass in dict order, each cell
//                       tested against all eight neighbour positions, first match per neighbour by ballot
//   SOG children        lane k = neighbour k: time bin, probability, top-n prefix, visited flag and the node
//                       record of all children at once (they touch eight distinct lattice points)
// Quirks kept: no dedup apart from the visited bitmap, fixLen's h with pathLen == 0,
// update_habitat_coverage popping while it enumerates, get_cell_prob's rounded corners.
#ifndef AUVP_ASTAR_KERNEL_H
#define AUVP_ASTAR_KERNEL_H
#include "auvp_math.h"
#include "auvp_types.h"
#include "auvp_wave.h"

namespace auvp {

struct AstarWorldDev {
  int32_t n_obstacles, n_habitats, n_poly, n_bins, n_cells, _pad;
  const double* ox;      // [O]
  const double* oy;
  const double* ot;      // [O] T(size_i): `d <= size` on the squared distance (plain, per obstacle)
  const double* hab;     // [H,4] x, y, size, T(size)
  const double* poly;    // [V,2]
  const double* bins;    // [T,2]
  const double* rcells;  // [C,4] cell corners rounded to 2 decimals (get_cell_prob, :500-501)
  const double* prob;    // [T,C]
  const double* topn;    // [T,C+1] prefix sums of the descending-sorted probabilities (get_top_n_prob)
  double cx, cy;         // polygon centroid (fan apex)
  // Product grid (detected on the host): cell r * g_ncol + c = [gx0[c], gx1[c]] x [gy0[r], gy1[r]] in list order, both
  // edge tables ascending, intervals touching at most.  get_cell_prob's first match (:485-514) is then the first
  // matching row and the first matching column among the three around the lower bounds of the point -- no sweep over
  // the cells.  g_ncol = 0: any other cell list (the sweep).
  int32_t g_ncol, g_nrow;
  const double* gx0;
  const double* gx1;
  const double* gy0;
  const double* gy1;
  double g_inv_dx, g_inv_dy;  // 1 / mean spacing of gx1 / gy1 (first guess of the lower bound only)
  double g_x1_0, g_y1_0;      // gx1[0], gy1[0]
};

// first index i in [0, n] with a[i] >= v (n if none), searched from the guess g; `a` ascending
template <typename PtrT>
__device__ __forceinline__ int astar_lower_bound(PtrT a, int n, double v, int g) {
  g = g < 0 ? 0 : (g > n ? n : g);
  while (g > 0 && a[g - 1] >= v) g--;
  while (g < n && a[g] < v) g++;
  return g;
}
// two of them at once (columns, rows): the entries on either side of both guesses are read together -- one round trip to the
// table when the guesses are right (on a regular grid they are), the walks above otherwise
template <typename PtrT>
__device__ __forceinline__ void astar_lower_bound2(PtrT ax, int nx, double vx, int gx_, PtrT ay, int ny, double vy, int gy_, int& rx, int& ry) {
  gx_ = gx_ < 0 ? 0 : (gx_ > nx ? nx : gx_);
  gy_ = gy_ < 0 ? 0 : (gy_ > ny ? ny : gy_);
  const double xl = gx_ > 0 ? ax[gx_ - 1] : 0.0, xh = gx_ < nx ? ax[gx_] : 0.0;
  const double yl = gy_ > 0 ? ay[gy_ - 1] : 0.0, yh = gy_ < ny ? ay[gy_] : 0.0;
  const bool okx = (gx_ == 0 || xl < vx) && (gx_ == nx || !(xh < vx));
  const bool oky = (gy_ == 0 || yl < vy) && (gy_ == ny || !(yh < vy));
  rx = okx ? gx_ : astar_lower_bound(ax, nx, vx, gx_);
  ry = oky ? gy_ : astar_lower_bound(ay, ny, vy, gy_);
}

struct AstarParamsDev {
  int32_t variant, cap_nodes, cap_exp, flags;
  double box[4];
  double velocity;
  double w[4];
  int32_t vx, vy;  // visited bitmap dims
  uint32_t epoch;  // 1..255: tag of this batch in the cell-info words (see AstarBuffers::cellinfo)
  int32_t _pad;
};

struct AstarSummary {  // must match auvp_astar_summary
  int32_t status, found, n_nodes, n_expansions, n_children, path_len, smooth_len, n_hab_left, visited_count, leaf;
  uint32_t open_scanned_lo, open_scanned_hi;  // sum over pops of len(open_list): entries the reference's min-f scan reads
};

struct AstarBuffers {
  const double* start;   // [E,2]
  const double* goal;    // [E,2]   (variants 0,1)
  const double* limit;   // [E]     (variants 2,3)
  // [E][cap_nodes][8] node records {x, y, g, h | f, cost, pathLen, parent i32 : time_stamp i32} (64 B: one line per node --
  // a pop reads one line, a child is four 16-byte stores), then [E][cap_nodes] f once more, contiguous, for the variants
  // whose pop scans every open node
  double* nodes;
  int32_t* node_i;       // [E][cap_nodes] (rounds 1-3: the open flag; round 4 folds it into the contiguous f copy -- a closed node's f there is a nan)
  // [E][vx*vy] one word per entry of the reference's visited_nodes array (variants 2,3):
  //   bits 31..24  epoch of the batch that wrote the word; a word of another epoch reads as "never touched", so a new
  //                batch needs no 360 KB-per-instance clear (the host bumps the epoch; a full clear every 255 batches)
  //   bit 16       visited_nodes[x][y] == 1
  //   bits 15..0   variant 3: 1 + cell key of this lattice point once get_cell_prob found it (0 = not yet); within one
  //                instance the visited index identifies the point (points are 10 apart)
  uint32_t* cellinfo;
  int32_t* hab_left;     // [E][H]
  double* exp_log;       // optional [E][cap_exp][8]
  AstarSummary* summary; // [E]
  int32_t* pipe_fail;    // host-mapped word, set to 1 by an instance that ends with AUVP_ST_PIPELINE (null: not reported)
};

__device__ __forceinline__ double astar_sqdist(double ax, double ay, double bx, double by) {
  double dx = auvp_fabs(ax - bx), dy = auvp_fabs(ay - by);
  return dx * dx + dy * dy;
}

// same_side (astar_real.py:62-68): np.cross of 2-vectors = a0*b1 - a1*b0
__device__ __forceinline__ bool astar_same_side(double p1x, double p1y, double p2x, double p2y, double ax, double ay,
                                                double bx, double by) {
  double ex = bx - ax, ey = by - ay;
  double cp1 = ex * (p1y - ay) - ey * (p1x - ax);
  double cp2 = ex * (p2y - ay) - ey * (p2x - ax);
  return cp1 * cp2 >= 0;
}

__device__ __forceinline__ bool astar_in_triangle(double px, double py, double ax, double ay, double bx, double by,
                                                  double cx, double cy) {
  return astar_same_side(px, py, ax, ay, bx, by, cx, cy) & astar_same_side(px, py, bx, by, ax, ay, cx, cy) &
         astar_same_side(px, py, cx, cy, ax, ay, bx, by);
}

// wave-cooperative collision_free(position) (:120-135): true when no obstacle covers the point
__device__ __forceinline__ bool astar_point_free(const AstarWorldDev& W, double px, double py) {
  bool hit = false;
  for (int i = lane_id(); i < W.n_obstacles; i += 64) {
    double dx = px - W.ox[i], dy = py - W.oy[i];
    hit = hit | (dx * dx + dy * dy <= W.ot[i]);
  }
  return !wave_any(hit);
}

constexpr int ASTAR_WAVES = 4;
constexpr int ASTAR_MAX_HAB = 64;
constexpr int ASTAR_OPEN_CAP = 768;
constexpr int ASTAR_MAX_BINS = 64;
constexpr int ASTAR_LDS_OBST = 256, ASTAR_LDS_POLY = 64, ASTAR_LDS_GRID = 256;

// PAIR (variant 3 on a product grid, latency batches): a second wavefront per instance evaluates what depends on the popped
// node's position alone -- the cell of every neighbour, path length, time stamp and time bin and the prob / topn table
// values -- while the first runs the bounds and collision tests (see the kernel).
struct AstarPairBox {
  int seq_m;      // the searching wavefront: expansion number + 1 whose node is posted
  int seq_x;      // the second wavefront: expansion number + 1 whose results are posted
  int stop, _p0;
  double cx, cy, clen;
  double len_[8], pr[8], tn[8];
  int ts_[8], tb[8], key[8];
};

// one instantiation per variant (VARIANT = AstarParamsDev::variant): the other variants' state and branches are gone
template <int VARIANT, bool PAIR = false>
__global__ __launch_bounds__(ASTAR_WAVES * 64 * (PAIR ? 2 : 1)) void astar_kernel(AstarWorldDev W, AstarParamsDev P, AstarBuffers B, int n_inst) {
  static_assert(!PAIR || VARIANT == 3, "the paired form is variant 3's");
  __shared__ AstarPairBox s_box[PAIR ? ASTAR_WAVES : 1];
  __shared__ int32_t s_hopen[ASTAR_WAVES][ASTAR_MAX_HAB];
  __shared__ int32_t s_hclosed[ASTAR_WAVES][ASTAR_MAX_HAB];
  __shared__ int32_t s_keys[ASTAR_WAVES][8];
  // compact list of the open nodes (f, index): the pop is a min over this list while it fits; the open flags in
  // HBM stay authoritative, so an instance that outgrows the list falls back to scanning them
  __shared__ double s_of[ASTAR_WAVES][ASTAR_OPEN_CAP];
  __shared__ int32_t s_oi[ASTAR_WAVES][ASTAR_OPEN_CAP];
  // the world is the same for the workgroup's four instances: one copy of its small tables
  __shared__ double s_bins[ASTAR_MAX_BINS][2];
  __shared__ double s_hab[ASTAR_MAX_HAB][3];  // x, y, T(size) of every habitat
  __shared__ double s_obs[3][ASTAR_LDS_OBST];  // x, y, T(size) of the first ASTAR_LDS_OBST obstacles
  __shared__ double s_poly[ASTAR_LDS_POLY][2];
  __shared__ double s_grid[ASTAR_LDS_GRID];    // product-grid edge tables gx0 | gx1 | gy0 | gy1 (when they fit)
  const int wave_id = uni((int)(threadIdx.x >> 6));
  // PAIR: wavefronts 0..3 search, wavefront 4 + ((i + 2) & 3) is the second wavefront of instance i (another SIMD than its partner's)
  const bool second = PAIR && wave_id >= ASTAR_WAVES;
  const int wave = second ? ((wave_id - ASTAR_WAVES + 2) & (ASTAR_WAVES - 1)) : wave_id;
  const int lane = lane_id();
  const int ep = (int)blockIdx.x * ASTAR_WAVES + wave;
  const int H = W.n_habitats, C = W.n_cells, T = W.n_bins;
  const bool obs_lds = W.n_obstacles <= ASTAR_LDS_OBST, poly_lds = W.n_poly <= ASTAR_LDS_POLY;
  const bool grid_lds = W.g_ncol > 0 && 2 * (W.g_ncol + W.g_nrow) <= ASTAR_LDS_GRID;
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    s_hab[i][0] = W.hab[4 * (size_t)i]; s_hab[i][1] = W.hab[4 * (size_t)i + 1]; s_hab[i][2] = W.hab[4 * (size_t)i + 3];
  }
  for (int i = threadIdx.x; i < 2 * T && i < 2 * ASTAR_MAX_BINS; i += blockDim.x) (&s_bins[0][0])[i] = W.bins[i];
  if (obs_lds)
    for (int i = threadIdx.x; i < W.n_obstacles; i += blockDim.x) { s_obs[0][i] = W.ox[i]; s_obs[1][i] = W.oy[i]; s_obs[2][i] = W.ot[i]; }
  if (poly_lds)
    for (int i = threadIdx.x; i < 2 * W.n_poly; i += blockDim.x) (&s_poly[0][0])[i] = W.poly[i];
  if (grid_lds)
    for (int i = threadIdx.x; i < 2 * (W.g_ncol + W.g_nrow); i += blockDim.x) s_grid[i] = W.gx0[i];  // one contiguous upload
  if (PAIR && threadIdx.x < ASTAR_WAVES) { s_box[threadIdx.x].seq_m = 0; s_box[threadIdx.x].seq_x = 0; s_box[threadIdx.x].stop = 0; }
  __syncthreads();
  if (ep >= n_inst) return;  // no workgroup barrier after this point
  constexpr int V = VARIANT;
  AstarPairBox* box = &s_box[PAIR ? wave : 0];
  (void)box;
  const int cap = P.cap_nodes;
  double4* rec = reinterpret_cast<double4*>(B.nodes + (size_t)ep * 8 * cap);  // two per node; 64-byte aligned
  // f once more, contiguous, behind the records of all instances (the scan of the variants without an open list in LDS)
  double* nf = B.nodes + (size_t)n_inst * 8 * cap + (size_t)ep * cap;
  auto put_node = [&](int c, double x, double y, double g, double hh, double f, double cost, double len, int par, int ts, int open) {
    rec[2 * (size_t)c] = make_double4(x, y, g, hh);
    rec[2 * (size_t)c + 1] = make_double4(f, cost, len, __longlong_as_double(((long long)ts << 32) | (long long)(uint32_t)par));
    nf[c] = open ? f : __builtin_nan("");  // (the scan's copy of f doubles as the open flag: a closed node's entry is a nan)
  };
  uint32_t* cellinfo = B.cellinfo ? B.cellinfo + (size_t)ep * P.vx * P.vy : nullptr;
  const uint32_t ep_tag = P.epoch << 24;
  int32_t* hopen = s_hopen[wave];
  int32_t* hclosed = s_hclosed[wave];
  const double sx = readfirst_f64(B.start[2 * (size_t)ep]), sy = readfirst_f64(B.start[2 * (size_t)ep + 1]);
  double gx = 0.0, gy = 0.0, limit = 0.0;
  if (V <= 1) { gx = readfirst_f64(B.goal[2 * (size_t)ep]); gy = readfirst_f64(B.goal[2 * (size_t)ep + 1]); }
  else limit = readfirst_f64(B.limit[ep]);
  const double w2 = P.w[1], w3 = P.w[2], w4 = P.w[3];
  const bool logx = (P.flags & 1) != 0 && B.exp_log != nullptr;

  if (!second) for (int i = lane; i < H; i += 64) hopen[i] = i;
  int n_hopen = H, n_hclosed = 0;
  unsigned long long closedmask = 0ull;  // bit h: habitat h is in the closed list
  if (lane == 0 && !second) {
    put_node(0, sx, sy, 0.0, 0.0, 0.0, 0.0, 0.0, -1, 0, 1);
  }
  wave_sync();
  int n_nodes = 1, n_open = 1, n_exp = 0, n_children = 0, status = 0, found = -1, visited_count = 0;
  unsigned long long open_scanned = 0ull;
  int first_open = 0;  // every node below this index is closed
  double* of_l = s_of[wave];
  int32_t* oi_l = s_oi[wave];
  int n_list = 1;
  bool list_ok = V >= 2 && !(P.flags & AUVP_KFLAG_ASTAR_NO_LIST);  // astar.py / astar_real.py never close a cell: their open sets outgrow the list at once
  if (lane == 0 && !second) { of_l[0] = 0.0; oi_l[0] = 0; }
  wave_sync();

  // neighbour offsets: lane k = lane >> 3 handles neighbour k, slice s = lane & 7
  const int k8 = lane >> 3, s8 = lane & 7;
  int offx, offy, offx7, offy7;  // (offx7 / offy7: of neighbour lane & 7 -- the lane that owns child lane & 7 in the children's phase)
  {
    // astar.py:88 order vs the other variants' (astar_fixLen.py:146)
    const int ax[8] = {0, 0, -10, 10, 10, 10, -10, -10}, ay[8] = {-10, 10, 0, 0, 10, -10, 10, -10};
    const int bx[8] = {0, 0, -10, 10, -10, -10, 10, 10}, by[8] = {-10, 10, 0, 0, -10, 10, -10, 10};
    offx = V == 0 ? ax[k8] : bx[k8];
    offy = V == 0 ? ay[k8] : by[k8];
    offx7 = V == 0 ? ax[lane & 7] : bx[lane & 7];
    offy7 = V == 0 ? ay[lane & 7] : by[lane & 7];
  }

  if (PAIR && second) {
    // ======================================================================================== the second wavefront (PAIR)
    // per expansion, from the popped node's position and path length alone: for each of the eight neighbours (lane kk < 8
    // = neighbour kk; the cell lookup on all 64 lanes as below) cell key, path length, time stamp, time bin, cell-info word,
    // prob / topn values -- whether or not the neighbour becomes a child (the searching wavefront knows after its bounds and
    // collision tests, which run meanwhile)
    for (int e = 0;; e++) {
      {
        int spins = 0;
        bool stop = false;
        for (;;) {
          const int sm = lds_peek(&box->seq_m);
          if (lds_peek(&box->stop)) { stop = true; break; }
          if (uni(sm) == e + 1) break;
          if (++spins > pipe_spin_limit()) { stop = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (stop) break;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");  // (it reads nothing the searching wavefront writes outside LDS)
      const double cxp = readfirst_f64(box->cx), cyp = readfirst_f64(box->cy), clen = readfirst_f64(box->clen);
      // every lane of group k8 works on neighbour k8 and lane (k8, 0) owns its row of results: no value crosses lanes except
      // through ballots, and the table reads below are issued together -- three dependent round trips (edge tables + time
      // bins; the rows / columns around the bounds; prob / topn) instead of a dozen
      const double qx = cxp + (double)offx, qy = cyp + (double)offy;
      const double sq_ = astar_sqdist(cxp, cyp, qx, qy);
      const bool lattice = sq_ == 100.0 || sq_ == 200.0;
      double root_ = sq_ == 100.0 ? 10.0 : 0x1.c48c6001f0acp+3;
      if (!wave_all(lattice)) root_ = lattice ? root_ : auvp_sqrt(sq_);
      const double len_ = clen + root_;
      const double dist_left = auvp_fabs(limit - len_);
      const int ts_ = (int)(P.velocity == 1.0 ? len_ : len_ / P.velocity);
      const double tsd = (double)ts_;
      const int ntop = (int)dist_left;
      int key = -1, tb = -1;
      auto tables = [&](const auto* gx0, const auto* gx1, const auto* gy0, const auto* gy1) {
        // ---- round trip 1: the entries either side of both lower-bound guesses, time bin s8, AND -- on the guess, which a
        // regular grid makes right -- column gc-1+s8 (s8 < 3) or row gr-4+s8 (3 <= s8 < 6) for the reference's float predicate
        int gc = (int)((qx - W.g_x1_0) * W.g_inv_dx), gr = (int)((qy - W.g_y1_0) * W.g_inv_dy);
        gc = gc < 0 ? 0 : (gc > W.g_ncol ? W.g_ncol : gc);
        gr = gr < 0 ? 0 : (gr > W.g_nrow ? W.g_nrow : gr);
        const bool colj = s8 < 3;
        const int n = colj ? W.g_ncol : W.g_nrow;
        const double v = colj ? qx : qy;
        int idx = colj ? gc - 1 + s8 : gr - 4 + s8;
        const double xl = gc > 0 ? gx1[gc - 1] : 0.0, xh = gc < W.g_ncol ? gx1[gc] : 0.0;
        const double yl = gr > 0 ? gy1[gr - 1] : 0.0, yh = gr < W.g_nrow ? gy1[gr] : 0.0;
        double2 bb = make_double2(0.0, 0.0);
        if (s8 < T) bb = *reinterpret_cast<const double2*>(&s_bins[s8][0]);
        double a = 0.0, b = 0.0;
        bool have = s8 < 6 && idx >= 0 && idx < n;
        if (have) { a = colj ? gx0[idx] : gy0[idx]; b = colj ? gx1[idx] : gy1[idx]; }
        const bool okx = (gc == 0 || xl < qx) && (gc == W.g_ncol || !(xh < qx));
        const bool oky = (gr == 0 || yl < qy) && (gr == W.g_nrow || !(yh < qy));
        if (wave_any(!okx || !oky)) {
          // ---- a guess was off: the walk, and the rows / columns around the true bounds
          if (!okx) gc = astar_lower_bound(gx1, W.g_ncol, qx, gc);
          if (!oky) gr = astar_lower_bound(gy1, W.g_nrow, qy, gr);
          idx = colj ? gc - 1 + s8 : gr - 4 + s8;
          have = s8 < 6 && idx >= 0 && idx < n;
          if (have) { a = colj ? gx0[idx] : gy0[idx]; b = colj ? gx1[idx] : gy1[idx]; }
        }
        bool m = false;
        if (have) {
          const double dd = auvp_fabs(a - b);
          m = auvp_fabs(v - a) <= dd && auvp_fabs(v - b) <= dd;
        }
        const unsigned long long bm = wave_ballot(m);
        const unsigned g8 = (unsigned)((bm >> (k8 * 8)) & 0xffull);
        const unsigned cm3 = g8 & 7u, rm3 = (g8 >> 3) & 7u;
        if (cm3 && rm3) key = (gr - 1 + (__ffs((int)rm3) - 1)) * W.g_ncol + (gc - 1 + (__ffs((int)cm3) - 1));
        // the first time bin that holds the time stamp (:520-527): bins 0..7 from the read above, further ones eight at a time
        {
          const unsigned long long tm = wave_ballot(s8 < T && tsd <= bb.y && tsd >= bb.x);
          const unsigned t8 = (unsigned)((tm >> (k8 * 8)) & 0xffull);
          if (t8) tb = __ffs((int)t8) - 1;
        }
        for (int t0 = 8; t0 < T && !wave_all(tb >= 0); t0 += 8) {
          const int t = t0 + s8;
          bool mt = false;
          if (t < T) { const double2 b2 = *reinterpret_cast<const double2*>(&s_bins[t][0]); mt = tsd <= b2.y && tsd >= b2.x; }
          const unsigned long long tm = wave_ballot(mt);
          const unsigned t8 = (unsigned)((tm >> (k8 * 8)) & 0xffull);
          if (tb < 0 && t8) tb = t0 + (__ffs((int)t8) - 1);
        }
      };
      if (grid_lds) tables(s_grid, s_grid + W.g_ncol, s_grid + 2 * W.g_ncol, s_grid + 2 * W.g_ncol + W.g_nrow);
      else tables(W.gx0, W.gx1, W.gy0, W.gy1);
      // ---- round trip 3: the table values
      double pr = 0.0, tn = 0.0;
      if (s8 == 0 && tb >= 0 && key >= 0 && key < C && ntop >= 0 && ntop <= C) { pr = W.prob[(size_t)tb * C + key]; tn = W.topn[(size_t)tb * (C + 1) + ntop]; }
      if (s8 == 0) {
        box->len_[k8] = len_; box->pr[k8] = pr; box->tn[k8] = tn;
        box->ts_[k8] = ts_; box->tb[k8] = tb; box->key[k8] = key;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");  // (the results are in LDS; its table reads have returned)
      if (lane == 0) lds_poke(&box->seq_x, e + 1);
    }
    return;
  }
  while (n_open > 0) {
    // ------------------------------------------------------------ pop the first minimum f
    double bf = __builtin_inf();
    int bi = 0x7fffffff, bpos = -1;
    if (list_ok) {
      for (int q = lane; q < n_list; q += 64) {
        const double f = of_l[q];
        const int i = oi_l[q];
        if (bi == 0x7fffffff || f < bf || (f == bf && i < bi)) { bf = f; bi = i; bpos = q; }
      }
    } else {
      // (one read per node -- a closed node's f is a nan -- and four blocks of 64 in flight per lane; rounds 1-3 read an open flag
      // and then, dependent on it, the f)
      for (int i0 = first_open; i0 < n_nodes; i0 += 256) {
        double fv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = i0 + 64 * u + lane;
          fv[u] = i < n_nodes ? nf[i] : __builtin_nan("");
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const double f = fv[u];
          if (f == f && (bi == 0x7fffffff || f < bf)) { bf = f; bi = i0 + 64 * u + lane; }
        }
      }
    }
    // reduce: smaller f wins; on equal f the smaller index wins (list order).  The minimum f first (six min steps),
    // then the lanes that hold it: almost always one
    {
      const bool have = bi != 0x7fffffff;
      const double fmin = wave_min_f64_dpp(have ? bf : __builtin_inf());
      const unsigned long long eq = wave_ballot(have && bf == fmin);
      if (__popcll(eq) == 1) {
        const int l = __ffsll((long long)eq) - 1;
        bi = __builtin_amdgcn_readlane(bi, l);
        bpos = __builtin_amdgcn_readlane(bpos, l);
      } else {
        if (eq != 0ull && !((eq >> lane) & 1ull)) bi = 0x7fffffff;  // (eq == 0: a nan among the f values -- the full comparison decides)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
          double of = __shfl_xor(bf, o, 64);
          int oi = __shfl_xor(bi, o, 64);
          int op = __shfl_xor(bpos, o, 64);
          bool take = (oi != 0x7fffffff) && (bi == 0x7fffffff || of < bf || (of == bf && oi < bi));
          bf = take ? of : bf;
          bi = take ? oi : bi;
          bpos = take ? op : bpos;
        }
      }
    }
    const int cur = uni(bi);
    // open nodes are left but none could be chosen: every remaining f is a nan (non-finite weights or grid values -- outside
    // the reference's domain; in the scan a nan also marks a closed node).  An error status, never an index out of bounds.
    if (cur == 0x7fffffff) { status = -1; break; }
    if (list_ok) {  // the last entry takes the place of the popped one (the order of the list does not matter)
      const int pp = uni(bpos);
      if (lane == 0 && pp != n_list - 1) { of_l[pp] = of_l[n_list - 1]; oi_l[pp] = oi_l[n_list - 1]; }
      n_list--;
      wave_sync();
    }
    if (lane == 0) nf[cur] = __builtin_nan("");
    open_scanned += (unsigned long long)n_open;
    n_open--;
    if (cur == first_open) first_open++;
    const double4 cur_a = rec[2 * (size_t)cur], cur_b = rec[2 * (size_t)cur + 1];  // one line
    const double cxp = readfirst_f64(cur_a.x), cyp = readfirst_f64(cur_a.y);
    const double cg = readfirst_f64(cur_a.z), ccost = readfirst_f64(cur_b.y), clen = readfirst_f64(cur_b.z);
    bool stop;
    if (V == 0) stop = (cxp == gx && cyp == gy);
    else if (V == 1) stop = astar_sqdist(cxp, cyp, gx, gy) <= 100;
    else stop = auvp_fabs(clen - limit) <= 10;
    if (stop) { found = cur; break; }
    if (logx && n_exp < P.cap_exp && lane == 0) {
      double* e = B.exp_log + ((size_t)ep * P.cap_exp + n_exp) * 8;
      e[0] = cxp; e[1] = cyp; e[2] = cg; e[3] = cur_a.w; e[4] = cur_b.x; e[5] = ccost; e[6] = clen;
      e[7] = (double)(int)(__double_as_longlong(cur_b.w) >> 32);
    }
    n_exp++;
    if (PAIR) {
      if (lane == 0) { box->cx = cxp; box->cy = cyp; box->clen = clen; }
      // (an LDS-only release: the partner reads the box and read-only tables, nothing this wavefront stores to memory -- a full
      // workgroup release would wait here for the previous expansion's node stores to be acknowledged)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      if (lane == 0) lds_poke(&box->seq_m, n_exp);
    }
    // ------------------------------------------------------------ neighbours: bounds, then collision
    const double qx = cxp + (double)offx, qy = cyp + (double)offy;
    // SOG: the visited/cell-info word of every neighbour is requested now (lane k < 8 = neighbour k), so the read is
    // under way while the bounds and collision tests run; only the children's words are used
    uint32_t ciw_early = 0u;
    double ex = 0.0, ey = 0.0;
    // (neighbour lane & 7's position: the same sum its own lanes form -- no cross-lane read on the chain)
    if (V >= 2) { ex = cxp + (double)offx7; ey = cyp + (double)offy7; }
    if (V >= 2 && lane < 8) {
      int xi = (int)(ex + 500), yi = (int)(ey + 200);
      if (xi < 0) xi += P.vx;
      if (yi < 0) yi += P.vy;
      if (xi >= 0 && xi < P.vx && yi >= 0 && yi < P.vy) ciw_early = cellinfo[(size_t)xi * P.vy + yi];
    }
    bool inb;
    if (V == 0) {
      inb = (qx >= P.box[0] && qx <= P.box[2]) && (qy >= P.box[1] && qy <= P.box[3]);
    } else {
      bool any_tri = false;
      auto fan = [&](const auto* pl) {
        for (int i = s8; i < W.n_poly; i += 8) {
          int j = (i != W.n_poly - 1) ? i + 1 : 0;
          any_tri = any_tri | astar_in_triangle(qx, qy, pl[2 * i], pl[2 * i + 1], pl[2 * j], pl[2 * j + 1], W.cx, W.cy);
        }
      };
      if (poly_lds) fan(&s_poly[0][0]);
      else fan(W.poly);
      unsigned long long bm = wave_ballot(any_tri);
      inb = ((bm >> (k8 * 8)) & 0xffull) != 0;
    }
    bool hit = false;
    if (inb) {
      auto circles = [&](const auto* ox, const auto* oy, const auto* ot) {
        // four obstacles per trip: their twelve reads are in flight together (the tests are independent)
        for (int i = s8; i < W.n_obstacles; i += 32) {
          double x[4], y[4], t[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int iu = i + 8 * u;
            const int ic = iu < W.n_obstacles ? iu : i;
            x[u] = ox[ic]; y[u] = oy[ic]; t[u] = ot[ic];
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const double dx = qx - x[u], dy = qy - y[u];
            hit = hit | (dx * dx + dy * dy <= t[u]);  // (a clamped index repeats obstacle i: same answer)
          }
        }
      };
      if (obs_lds) circles(s_obs[0], s_obs[1], s_obs[2]);
      else circles(W.ox, W.oy, W.ot);
    }
    const unsigned long long hm = wave_ballot(hit);
    const unsigned long long im = wave_ballot(inb && s8 == 0);
    int childmask = 0;  // bit k: neighbour k becomes a child
#pragma unroll
    for (int k = 0; k < 8; k++) {
      bool in_k = (im >> (k * 8)) & 1ull;
      bool hit_k = ((hm >> (k * 8)) & 0xffull) != 0;
      n_children += in_k ? 1 : 0;
      childmask |= (in_k && !hit_k) ? (1 << k) : 0;
    }
    const int nch = __popc(childmask);
    if (n_nodes + nch > cap) { status = -2; break; }
    // ------------------------------------------------------------ children, in neighbour order
    unsigned long long covm_v = 0ull;  // fixLen, lane k: bit set of the habitats that cover child k
    if (V == 2) {
      // first loop of :350-363: habitat coverage update per new node (mutates the lists).  Which habitats cover a child does
      // not depend on the lists: all eight children at once, lane (k, s) = child k against habitats s, s + 8, .. (H <= 64) -- two
      // rounds of table reads for ten habitats instead of eight serial ones with the child's position shuffled across lanes
      {
        unsigned long long cmine = 0ull;  // lane k < 8: the habitats that cover neighbour k
        for (int h0 = 0; h0 < H; h0 += 8) {
          const int hb = h0 + s8;
          bool cov = false;
          if (hb < H) {
            const double ddx = qx - s_hab[hb][0], ddy = qy - s_hab[hb][1];
            cov = ddx * ddx + ddy * ddy <= s_hab[hb][2];
          }
          const unsigned long long bm = wave_ballot(cov);
          cmine |= ((bm >> (8 * (lane & 7))) & 0xffull) << h0;
        }
        if (lane < 8 && ((childmask >> lane) & 1)) covm_v = cmine;
      }
      for (int k = 0; k < 8; k++) {
        if (!((childmask >> k) & 1)) continue;
        const unsigned long long cm = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(covm_v >> 32), k) << 32) |
                                      (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(covm_v & 0xffffffffull), k);
        if ((cm & ~closedmask) == 0ull) continue;  // no open habitat covers it: the lists stay as they are (most children)
        unsigned long long removed = 0ull;
        if (lane == 0) {
          // update_habitat_coverage (:182-199): pop(index) while enumerating skips the next element
          for (int idx = 0; idx < n_hopen; idx++) {
            const int hb = hopen[idx];
            if ((cm >> hb) & 1ull) {
              hclosed[n_hclosed++] = hb;
              removed |= 1ull << hb;
              for (int m = idx; m < n_hopen - 1; m++) hopen[m] = hopen[m + 1];
              n_hopen--;
            }
          }
        }
        n_hopen = __shfl(n_hopen, 0, 64);
        n_hclosed = __shfl(n_hclosed, 0, 64);
        closedmask |= ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(removed >> 32), 0, 64) << 32) |
                      (unsigned long long)(uint32_t)__shfl((int)(uint32_t)(removed & 0xffffffffull), 0, 64);
        wave_sync();
      }
    }
    if (V == 2) {
      // second loop (:364-416), lane k < 8 = child k: with the coverage bit sets nothing a child does is seen by another
      // (eight distinct lattice points), so costs, visited words, node stores and the open-list append go out together.
      // cost_of_edge with the lists as they stand after all children were created (:395-400, cost.py:66-101): every
      // habitat is in exactly one of the two lists and the coverage test is the one of the first loop, so d2 = some
      // habitat covers the child, d3 = some habitat of the closed list does
      const int kk = lane & 7;
      const bool mine = lane < 8 && ((childmask >> kk) & 1);
      const double px = ex, py = ey;
      const double d2 = covm_v ? 1.0 : 0.0, d3 = (covm_v & closedmask) ? 1.0 : 0.0;
      const double g_ = ccost - w2 * d2 - w3 * d3;
      const double h_ = -w2 * auvp_fabs(limit - 0.0) - w3 * (double)n_hopen;  // child.pathLen is still 0 (:401-403)
      const double f_ = g_ + h_;
      // (lattice steps: the squared distance is exactly 100 or 200 unless a coordinate carries fraction bits the step rounded)
      const double sq_ = astar_sqdist(cxp, cyp, px, py);
      const bool lattice = sq_ == 100.0 || sq_ == 200.0;
      double root_ = sq_ == 100.0 ? 10.0 : 0x1.c48c6001f0acp+3;
      if (!wave_all(lattice)) root_ = lattice ? root_ : auvp_sqrt(sq_);
      const double len_ = clen + root_;
      // visited bitmap (:414-416), numpy index semantics (negative wraps)
      int xi = (int)(px + 500), yi = (int)(py + 200);
      if (xi < 0) xi += P.vx;
      if (yi < 0) yi += P.vy;
      const bool oob = mine && (xi < 0 || xi >= P.vx || yi < 0 || yi >= P.vy);
      if (wave_any(oob)) { status = -1; break; }
      const size_t vi = (size_t)xi * P.vy + yi;
      const uint32_t ciw = mine ? ciw_early : 0u;  // (requested before the bounds test; same index)
      const bool was = (ciw & 0xff000000u) == ep_tag && (ciw & 0x10000u);
      const int open_ = was ? 0 : 1;
      if (mine) {
        const int c = n_nodes + __popc(childmask & ((1 << kk) - 1));
        put_node(c, px, py, g_, h_, f_, g_, len_, cur, 0, open_);
        if (!was) cellinfo[vi] = ep_tag | 0x10000u;
      }
      const unsigned long long om = wave_ballot(mine && open_);
      const int opened = __popcll(om);
      if (list_ok) {
        if (n_list + opened > ASTAR_OPEN_CAP) list_ok = false;
        else {
          if (mine && open_) {
            const int q = n_list + __popcll(om & ((1ull << lane) - 1ull));
            of_l[q] = f_; oi_l[q] = n_nodes + __popc(childmask & ((1 << kk) - 1));
          }
          n_list += opened;
        }
      }
      n_open += opened;
      visited_count += opened;
      wave_sync();
    }
    // the children's nodes once their inputs are known (variant 3): costs, stores, visited words, open-list append
    auto finish_children = [&](const int kk, const bool mine, const double px, const double py, const double len_, const double dist_left,
                               const int ts_, const size_t vi, const int key, const bool need_key, const double pr, const double tn,
                               const int was) {
      const double g_ = ccost - w4 * pr;
      const double h_ = -w2 * dist_left - w3 * (double)H - w4 * tn;
      const double f_ = g_ + h_;
      const int open_ = was ? 0 : 1;
      if (mine) {
        const int c = n_nodes + __popc(childmask & ((1 << kk) - 1));
        put_node(c, px, py, g_, h_, f_, g_, len_, cur, ts_, open_);
        if (!was || need_key) cellinfo[vi] = ep_tag | 0x10000u | (uint32_t)(key + 1);  // visited from now on, key kept
      }
      const unsigned long long om = wave_ballot(mine && open_);
      const int opened = __popcll(om);
      if (list_ok) {
        if (n_list + opened > ASTAR_OPEN_CAP) list_ok = false;
        else {
          if (mine && open_) {
            const int q = n_list + __popcll(om & ((1ull << lane) - 1ull));
            of_l[q] = f_; oi_l[q] = n_nodes + __popc(childmask & ((1 << kk) - 1));
          }
          n_list += opened;
        }
      }
      n_open += opened;
      visited_count += opened;
      wave_sync();
    };
    if (V == 3) {
      if (PAIR) {
        // the second wavefront's results of this expansion (it started when the popped node was posted)
        {
          int spins = 0;
          while (uni(lds_peek(&box->seq_x)) != n_exp) {
            if (++spins > pipe_spin_limit()) { status = AUVP_ST_PIPELINE; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          if (status) break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        const int kk = lane & 7;
        const bool mine = lane < 8 && ((childmask >> kk) & 1);
        const double px = ex, py = ey;
        const double len_ = box->len_[kk], pr_x = box->pr[kk], tn_x = box->tn[kk];
        const int ts_ = box->ts_[kk], tb = box->tb[kk], key_x = box->key[kk];
        const uint32_t ciw = mine ? ciw_early : 0u;  // (requested before the bounds test; same index)
        const double dist_left = auvp_fabs(limit - len_);
        const int ntop = (int)dist_left;
        int xi = (int)(px + 500), yi = (int)(py + 200);
        if (xi < 0) xi += P.vx;
        if (yi < 0) yi += P.vy;
        if (wave_any(mine && (xi < 0 || xi >= P.vx || yi < 0 || yi >= P.vy))) { status = -1; break; }
        const size_t vi = (size_t)xi * P.vy + yi;
        const bool ci_live = (ciw & 0xff000000u) == ep_tag;
        const int key = mine ? key_x : 0;
        const bool bad = mine && (tb < 0 || key < 0 || ntop > C);
        if (wave_any(bad)) { status = -1; break; }
        const int was = (mine && ci_live && (ciw & 0x10000u)) ? 1 : 0;
        finish_children(kk, mine, px, py, len_, dist_left, ts_, vi, key, false, mine ? pr_x : 0.0, mine ? tn_x : 0.0, was);
      } else {
      // get_cell_prob's cell of every neighbour on a product grid: lane (k8, s8) tests column gc-1+s8 (s8 < 3) or row
      // gr-4+s8 (3 <= s8 < 6) of neighbour k8 with the reference's own float predicate; the first matching row and the
      // first matching column give the first matching cell of the list (every (row, column) pair of matches is a match)
      int key_grid = -1;
      const bool grid = W.g_ncol > 0;
      if (grid) {
        int gc = 0, gr = 0;
        bool m = false;
        auto lookup = [&](const auto* gx0, const auto* gx1, const auto* gy0, const auto* gy1) {
          astar_lower_bound2(gx1, W.g_ncol, qx, (int)((qx - W.g_x1_0) * W.g_inv_dx), gy1, W.g_nrow, qy, (int)((qy - W.g_y1_0) * W.g_inv_dy), gc, gr);
          const bool colj = s8 < 3;
          const int idx = colj ? gc - 1 + s8 : gr - 4 + s8;
          const int n = colj ? W.g_ncol : W.g_nrow;
          if (s8 < 6 && idx >= 0 && idx < n) {
            const double a = colj ? gx0[idx] : gy0[idx], b = colj ? gx1[idx] : gy1[idx], v = colj ? qx : qy;
            const double dd = auvp_fabs(a - b);
            m = auvp_fabs(v - a) <= dd && auvp_fabs(v - b) <= dd;
          }
        };
        if (grid_lds) lookup(s_grid, s_grid + W.g_ncol, s_grid + 2 * W.g_ncol, s_grid + 2 * W.g_ncol + W.g_nrow);
        else lookup(W.gx0, W.gx1, W.gy0, W.gy1);
        const unsigned long long bm = wave_ballot(m);
        const unsigned g8 = (unsigned)((bm >> (k8 * 8)) & 0xffull);
        const unsigned cm3 = g8 & 7u, rm3 = (g8 >> 3) & 7u;
        if (cm3 && rm3) key_grid = (gr - 1 + (__ffs((int)rm3) - 1)) * W.g_ncol + (gc - 1 + (__ffs((int)cm3) - 1));
      }
      // children of the SOG variant, lane k < 8 = neighbour k: nothing one child does is seen by another (eight
      // distinct lattice points, so eight distinct visited cells), so their table reads go out together
      const int kk = lane & 7;
      const int key_g = __shfl(key_grid, kk * 8, 64);
      const bool mine = lane < 8 && ((childmask >> kk) & 1);
      const double px = ex, py = ey;
      // (lattice steps: the squared distance is exactly 100 or 200 unless the coordinates carry fraction bits that the step
      // rounded; the correctly rounded roots of those two are constants)
      const double sq_ = astar_sqdist(cxp, cyp, px, py);
      const bool lattice = sq_ == 100.0 || sq_ == 200.0;
      double root_ = sq_ == 100.0 ? 10.0 : 0x1.c48c6001f0acp+3;
      if (!wave_all(lattice)) root_ = lattice ? root_ : auvp_sqrt(sq_);
      const double len_ = clen + root_;
      const double dist_left = auvp_fabs(limit - len_);
      const int ts_ = (int)(P.velocity == 1.0 ? len_ : len_ / P.velocity);  // x / 1.0 is x: no division on the chain
      // the first time bin that holds ts_ (:520-527): lane (k8, s8) tests bins s8, s8 + 8, .. for neighbour k8's time stamp
      int tb = -1;
      {
        const double tsn = (double)__shfl(ts_, k8, 64);  // (lane kk < 8 holds child kk's)
        for (int t0 = 0; t0 < T; t0 += 8) {
          const int t = t0 + s8;
          bool m = false;
          if (t < T) { const double2 bb = *reinterpret_cast<const double2*>(&s_bins[t][0]); m = tsn <= bb.y && tsn >= bb.x; }
          const unsigned long long bm = wave_ballot(m);
          const unsigned mine8 = (unsigned)((bm >> (8 * (lane & 7))) & 0xffull);  // lane kk < 8: its own child's slices
          if (tb < 0 && mine8) tb = t0 + (__ffs((int)mine8) - 1);
          if (wave_all(tb >= 0 || lane >= 8)) break;
        }
      }
      const int ntop = (int)dist_left;
      int xi = (int)(px + 500), yi = (int)(py + 200);
      if (xi < 0) xi += P.vx;
      if (yi < 0) yi += P.vy;
      const bool oob = mine && (xi < 0 || xi >= P.vx || yi < 0 || yi >= P.vy);
      if (wave_any(oob)) { status = -1; break; }
      const size_t vi = (size_t)xi * P.vy + yi;
      // the cell of a lattice point is looked up once per search: within one instance the visited-bitmap index
      // identifies the point (points are 10 apart), so the key is kept next to it
      const uint32_t ciw = mine ? ciw_early : 0u;  // (same index: lane kk's neighbour is px, py)
      const bool ci_live = (ciw & 0xff000000u) == ep_tag;  // written by this batch (or uploaded for it)
      int key = mine ? (grid ? key_g : (ci_live ? (int)(ciw & 0xffffu) - 1 : -1)) : 0;
      const bool need_key = !grid && mine && key < 0;
    // get_cell_prob (:485-514) for ALL children of this expansion in one sweep over the cells: a lane loads one
    // cell per pass and tests it against the (uniform) positions of the eight neighbours, so the sweep costs
    // ceil(C / 64) independent loads instead of that many dependent round trips per child
      if (wave_any(need_key)) {
        int keys[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
      double cpx[8], cpy[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { cpx[k] = readfirst_f64(__shfl(qx, k * 8, 64)); cpy[k] = readfirst_f64(__shfl(qy, k * 8, 64)); }
      for (int c0 = 0; c0 < C; c0 += 64) {
        const int ci = c0 + lane;
        const bool in = ci < C;
        const double4 r = in ? reinterpret_cast<const double4*>(W.rcells)[ci] : make_double4(0.0, 0.0, 0.0, 0.0);
        const double ddx = auvp_fabs(r.x - r.z), ddy = auvp_fabs(r.y - r.w);
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const bool m = in && (auvp_fabs(cpx[k] - r.x) <= ddx && auvp_fabs(cpx[k] - r.z) <= ddx) &&
                         (auvp_fabs(cpy[k] - r.y) <= ddy && auvp_fabs(cpy[k] - r.w) <= ddy);
          const unsigned long long mm = wave_ballot(m);
          if (keys[k] < 0 && mm) keys[k] = c0 + (__ffsll((long long)mm) - 1);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; k++) if (lane == k) s_keys[wave][k] = keys[k];
      wave_sync();
    }
      if (need_key) key = s_keys[wave][kk];
      const bool bad = mine && (tb < 0 || key < 0 || ntop > C);
      if (wave_any(bad)) { status = -1; break; }
      double pr = 0.0, tn = 0.0;
      int was = 0;
      if (mine) { pr = W.prob[(size_t)tb * C + key]; tn = W.topn[(size_t)tb * (C + 1) + ntop]; was = (ci_live && (ciw & 0x10000u)) ? 1 : 0; }
      finish_children(kk, mine, px, py, len_, dist_left, ts_, vi, key, need_key, pr, tn, was);
      }
    }
    if (V <= 1 && status == 0) {
      // children of astar.py / astar_real.py (:244-266 / :195-217), lane k < 8 = child of neighbour k: eight distinct lattice
      // points, created in neighbour order (their indices by rank in the child mask); nothing is ever closed there, so a child is
      // one record and one open flag.  (Their open sets outgrow the LDS list at once: list_ok is false from the start.)
      const int kk = lane & 7;
      const bool mine = lane < 8 && ((childmask >> kk) & 1);
      const double px = cxp + (double)offx7, py = cyp + (double)offy7;
      const double g_ = cg + astar_sqdist(px, py, cxp, cyp);
      const double h_ = astar_sqdist(px, py, gx, gy);
      const double f_ = g_ + h_;
      if (mine) put_node(n_nodes + __popc(childmask & ((1 << kk) - 1)), px, py, g_, h_, f_, 0.0, 0.0, cur, 0, 1);
      n_open += nch;
      wave_sync();
    }
    if (status) break;
    n_nodes += nch;
  }

  if (PAIR && lane == 0) lds_poke(&box->stop, 1);
  for (int i = lane; i < n_hopen; i += 64) B.hab_left[(size_t)ep * (H > 0 ? H : 1) + i] = hopen[i];
  if (lane == 0) {
    AstarSummary& s = B.summary[ep];
    pipe_report(B.pipe_fail, status);
    s.status = status; s.found = found >= 0 ? 1 : 0; s.n_nodes = n_nodes; s.n_expansions = n_exp; s.n_children = n_children;
    int L = 0;
    for (int m = found; m >= 0; m = (int)(__double_as_longlong(rec[2 * (size_t)m + 1].w) & 0xffffffffll)) L++;
    s.path_len = L; s.smooth_len = 0; s.n_hab_left = n_hopen; s.visited_count = visited_count; s.leaf = found;
    s.open_scanned_lo = (uint32_t)(open_scanned & 0xffffffffull); s.open_scanned_hi = (uint32_t)(open_scanned >> 32);
  }
}

// Backtrack of the popped goal node (:231-243 / :319-333 / :593-618) and, for variant 3, smoothPath
// (:404-440).  out per instance: path [L,3] root->leaf x,y,round(time_stamp,2); cost_list [L] leaf->root;
// node_path [L,8] root->leaf; smooth [<=L,3].  One wave per instance.
__global__ __launch_bounds__(64) void astar_path_kernel(AstarWorldDev W, AstarParamsDev P, AstarBuffers B,
                                                        const int64_t* __restrict__ offsets, double* __restrict__ path,
                                                        double* __restrict__ cost_list, double* __restrict__ node_path,
                                                        double* __restrict__ smooth, int n_inst) {
  const int ep = blockIdx.x;
  if (ep >= n_inst) return;
  const int lane = lane_id();
  AstarSummary& s = B.summary[ep];
  if (!s.found || s.path_len <= 0) return;
  const int cap = P.cap_nodes, L = s.path_len;
  const double4* rec = reinterpret_cast<const double4*>(B.nodes + (size_t)ep * 8 * cap);
  const size_t o = (size_t)offsets[ep];
  double* pth = path + 3 * o;
  if (lane == 0) {
    int k = 0;
    for (int m = s.leaf; m >= 0; k++) {
      const double4 a = rec[2 * (size_t)m], b = rec[2 * (size_t)m + 1];
      const long long pk = __double_as_longlong(b.w);
      const int par = (int)(pk & 0xffffffffll), ts = (int)(pk >> 32);
      cost_list[o + k] = b.y;
      double* e = pth + 3 * (size_t)(L - 1 - k);
      e[0] = a.x; e[1] = a.y; e[2] = (double)ts;  // round(int, 2) == int
      double* q = node_path + 8 * (o + (size_t)(L - 1 - k));
      q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = (double)ts;
      m = par;
    }
  }
  __threadfence_block();
  __syncthreads();
  if (P.variant != 3) return;
  // smoothPath: keep[] lives in the smooth buffer's tail until compaction (L doubles are enough)
  int status = 0;
  unsigned long long keepm[8];  // L <= 512 path nodes as bit masks
  for (int i = 0; i < 8; i++) keepm[i] = ~0ull;
  if (L > 512) { if (lane == 0) s.status = -2; return; }
  if (L >= 2) {
    int index = 1, check = 0, curp = 1;
    while (index < L - 1) {
      const double ax = pth[3 * (size_t)check], ay = pth[3 * (size_t)check + 1];
      const double bx = pth[3 * (size_t)curp], by = pth[3 * (size_t)curp + 1];
      // Walkable (:218-242)
      double wx = ax, wy = ay;
      const int step_x = (int)(auvp_fabs(bx - ax) / 5), step_y = (int)(auvp_fabs(by - ay) / 5);
      bool walk = true;
      int guard = 0;
      while (wx <= bx && wy <= by) {
        const double ix = wx, iy = wy;
        wx += step_x; wy += step_y;
        if (!astar_point_free(W, ix, iy)) { walk = false; break; }
        if (++guard > 1000000) { status = -1; break; }  // both steps 0: the reference never returns
      }
      if (status) break;
      if (walk) {
        // inside_habitats (:304-320): module-level euclidean_dist = sqrt(|dx|^2 + |dy|^2)
        bool inside = false;
        for (int h = lane; h < W.n_habitats; h += 64) {
          double d = auvp_sqrt(astar_sqdist(W.hab[4 * h], W.hab[4 * h + 1], bx, by));
          inside = inside | (d <= W.hab[4 * h + 2]);
        }
        if (!wave_any(inside)) keepm[curp >> 6] &= ~(1ull << (curp & 63));
        index += 1; curp = index;
      } else {
        check = curp; index += 1; curp = index;
      }
    }
  }
  if (lane == 0) {
    int n = 0;
    for (int i = 0; i < L; i++) {
      if ((keepm[i >> 6] >> (i & 63)) & 1ull) {
        smooth[3 * (o + n)] = pth[3 * (size_t)i]; smooth[3 * (o + n) + 1] = pth[3 * (size_t)i + 1];
        smooth[3 * (o + n) + 2] = pth[3 * (size_t)i + 2];
        n++;
      }
    }
    s.smooth_len = n;
    if (status) s.status = status;
  }
}

}  // namespace auvp
#endif
