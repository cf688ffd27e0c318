// rrt_explore_kernel.h -- RRT.exploring (path_planning/rrt_dubins.py:92-176) on gfx950.
//
// One wavefront = one episode, persistent over the whole iteration budget.  A 512-thread workgroup
// carries 8 independent episodes that share the LDS copy of the small world tables (habitats,
// polygon, time bins, obstacle tile); the tree lives in HBM.
//
// Inside one expansion the 64 lanes split the work:
//   steer        lane s = sub-arc s: RNG window tempering, arc geometry and sin/cos in parallel;
//                only the running sums (theta; then x, y, t, length on four lanes) are serial, and
//                they must be, to keep the reference's left-to-right fp64 addition order
//   collision    cull: lane = obstacle (J slots of 64) against the path's bounding box; the few candidates that
//                survive are tested one at a time with lane = path point
//   polygon      lane = path point
//   cost         not here: the qualifying-leaf cost of exploring (:158-171) only decides which leaf is returned, never
//                how the tree grows, so rrt_leaf_kernel evaluates it on the finished tree (lane = node, all lanes busy)
// The steer draw count is data dependent (a sub-arc consumes 2 or 3 random() values): the lanes
// temper a window of the stream, ballot the "taken" predicate for every possible start offset, and
// a lane-parallel fixed-point iteration resolves where each sub-arc starts.
//
// Everything that is the same for all lanes of an episode is forced into SGPRs (readfirstlane), so
// episode-level control flow is scalar branches and the VGPR budget stays with the per-lane work.
#ifndef AUVP_RRT_EXPLORE_KERNEL_H
#define AUVP_RRT_EXPLORE_KERNEL_H
#include "auvp_math.h"
#include "auvp_types.h"
#include "auvp_wave.h"

namespace auvp {

constexpr int RRT_WAVES = 4;       // episodes per workgroup (Planner_RRT)
// RRT.exploring: 8 episodes share one copy of the world tables and the obstacle tile, so three workgroups (24 waves,
// six per SIMD) fit the CU's 160 KB of LDS; the register budget for six waves is 80 VGPRs, and what does not fit
// (loop-invariant lane constants) is spilled once and reloaded about twice per expansion
constexpr int RRT_X_WAVES = 8;
constexpr int RRT_MAX_CHUNK = 63;  // sub-arcs per steer pass (one lane stays idle: chunk-entry angle)
constexpr int RRT_MAX_HAB = 64;    // visited-habitat mask is one 64-bit word
constexpr int RRT_MAX_POLY = 64;
constexpr int RRT_MAX_BINS = 64;
constexpr int RRT_ELIST = 192;     // path elements collected per cost pass group

// The world tables every episode of a workgroup shares, at the start of its LDS and sized by the world at hand
// (a fixed 64-row layout costs 4.4 KB where the Catalina-like world needs 0.9 KB; LDS is what limits the number of
// resident episodes per CU):
//   [WorldDev]  copy of the kernel argument for the code that runs rarely (evaluating a path element for the
//               cost): reading the table pointers from here when they are needed keeps ~40 scalar registers out
//               of the expansion loop, which otherwise spills them to vector-register lanes
//   [RrtParamsDev]  copy of the planning parameters (rrt_explore_kernel)
//   [hab]       n_habitats x {x, y, size, T(size)}      [poly] n_poly x {x, y}      [bins] n_bins x {t0, t1}
struct RrtTables {
  WorldDev* world;
  RrtParamsDev* params;  // RRT.exploring only (staged by the kernel itself)
  double (*hab)[4];
  double (*poly)[2];
  double (*bins)[2];
};
constexpr int RRT_WORLD_ONLY_BYTES = (int)((sizeof(WorldDev) + 15) & ~(size_t)15);
constexpr int RRT_WORLD_BYTES = RRT_WORLD_ONLY_BYTES + (int)((sizeof(RrtParamsDev) + 15) & ~(size_t)15);
__host__ __device__ inline int rrt_tables_bytes(int n_habitats, int n_poly, int n_bins) {
  return RRT_WORLD_BYTES + n_habitats * 32 + n_poly * 16 + n_bins * 16;
}
__device__ __forceinline__ RrtTables rrt_tables_view(unsigned char* base, int n_habitats, int n_poly) {
  RrtTables t;
  t.world = reinterpret_cast<WorldDev*>(base);
  t.params = reinterpret_cast<RrtParamsDev*>(base + RRT_WORLD_ONLY_BYTES);
  t.hab = reinterpret_cast<double(*)[4]>(base + RRT_WORLD_BYTES);
  t.poly = reinterpret_cast<double(*)[2]>(base + RRT_WORLD_BYTES + n_habitats * 32);
  t.bins = reinterpret_cast<double(*)[2]>(base + RRT_WORLD_BYTES + n_habitats * 32 + n_poly * 16);
  return t;
}
// fill the tables (whole workgroup; the caller synchronises)
__device__ __forceinline__ void rrt_tables_stage(const RrtTables& S, const WorldDev& W) {
  for (int i = threadIdx.x; i < W.n_habitats; i += blockDim.x) {
    S.hab[i][0] = W.hab[3 * i]; S.hab[i][1] = W.hab[3 * i + 1]; S.hab[i][2] = W.hab[3 * i + 2];
    S.hab[i][3] = W.hab_t[i];
  }
  if (threadIdx.x == 0) *S.world = W;
  for (int i = threadIdx.x; i < W.n_poly * 2; i += blockDim.x) (&S.poly[0][0])[i] = W.poly[i];
  for (int i = threadIdx.x; i < W.n_bins * 2; i += blockDim.x) (&S.bins[0][0])[i] = W.bins[i];
}

// Per-wave LDS layout (bytes), all sizes multiples of 16:
//   [scratch]  steer: u[3C+3] | {inc[(C+1)*4], sc[(C+1)*2], phi[C+1]}
//   [mt]       624 u32
//   [pts]      max_pts * 2 f64
//   [bins]     (K+2) i32
// before them the world tables (RrtTables); after the RRT_X_WAVES per-wave blocks the obstacle tile shared by the
// workgroup: x, y, T as f64 [J*64] each and the cull radius as f32 [J*64]
struct RrtLdsPlan {
  int chunk;  // C
  int tables, scratch, mt, pts, bins, per_wave, total;
};

__host__ __device__ inline RrtLdsPlan rrt_lds_plan(int K, int max_pts, int nfreq, int obst_slots, int tables_bytes,
                                                   int waves = RRT_X_WAVES) {
  RrtLdsPlan p;
  p.chunk = nfreq < 1 ? 1 : (nfreq > RRT_MAX_CHUNK ? RRT_MAX_CHUNK : nfreq);
  const int C = p.chunk;
  int steer_u = (64 + 3 * C + 3) * 8;  // 64 leading random() values + the steer window
  // inc rows, sc and phi all padded to the even row length CS = C + 1 rounded up
  int steer_s = (7 * ((C + 2) & ~1) + 4) * 8;
  int s = steer_u > steer_s ? steer_u : steer_s;
  p.scratch = (s + 15) & ~15;
  p.mt = 624 * 4;
  p.pts = ((max_pts * 16) + 15) & ~15;
  p.bins = (((K + 2) * 4) + 15) & ~15;
  p.per_wave = p.scratch + p.mt + p.pts + p.bins;
  p.tables = (tables_bytes + 15) & ~15;
  p.total = p.tables + waves * p.per_wave + obst_slots * (3 * 8 + 4);  // + obstacle tile x,y,T (f64), r (f32)
  return p;
}

// Point(x,y).within(polygon): even-odd crossing number with strict comparisons -- the definition
// pinned in tests/golden/_refstubs/install.py (shapely itself is absent).
// Wave-parallel over (point, edge) pairs: lane = p_local * nv + e, so one pass evaluates the edge
// test (with its fp64 division) for floor(64/nv) points at once; a point is inside iff the number
// of crossing edges in its nv-bit group of the ballot is odd.  Returns true when ANY of the
// n_pts points is not strictly inside.
__device__ __forceinline__ bool any_point_outside(const double (*poly)[2], int nv, const double (*pts)[2], int n_pts) {
  const int lane = lane_id();
  // no boundary polygon: no point is within it (the crossing loop of the definition never runs), so a path
  // with any point at all is outside; also keeps 64 / nv below well defined
  if (nv <= 0) return n_pts > 0;
  const int per = 64 / nv;  // points per pass (nv <= 64)
  const int e = lane % nv, pl = lane / nv;
  const int ej = e == 0 ? nv - 1 : e - 1;  // j trails i by one vertex
  const double xi = poly[e][0], yi = poly[e][1], xj = poly[ej][0], yj = poly[ej][1];
  bool outside = false;
  for (int p0 = 0; p0 < n_pts; p0 += per) {
    const int p = p0 + pl;
    const bool live = pl < per && p < n_pts;
    bool cross = false;
    if (live) {
      const double x = pts[p][0], y = pts[p][1];
      if ((yi > y) != (yj > y)) cross = x < (xj - xi) * (y - yi) / (yj - yi) + xi;
    }
    const unsigned long long cm = wave_ballot(cross);
    // lanes 0..per-1 each judge one point of this pass
    const int q = p0 + lane;
    if (lane < per && q < n_pts) {
      const unsigned long long grp = (cm >> (lane * nv)) & (nv == 64 ? ~0ull : ((1ull << nv) - 1ull));
      outside = outside | ((__popcll(grp) & 1) == 0);
    }
  }
  return wave_any(outside);
}

// cost.py:181-184 first-match cell scan through the x-bucket index; returns cell id or -1
// first index i in [0, n) with a[i] >= v (n if none): lower bound on a non-decreasing table, found from a guess by
// comparing against the table's own values, so the result is exact whatever the guess
__device__ __forceinline__ int lower_bound_from(const double* __restrict__ a, int n, double v, int i) {
  i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
  while (i > 0 && a[i - 1] >= v) i--;
  while (i < n && a[i] < v) i++;
  return i;
}

// `grid_lds`: optional copy of the separable-grid tables in LDS, laid out X1[ncol] X0[ncol] Y1[nrow] Y0[nrow]
__device__ __forceinline__ int cell_lookup(const WorldDev& W, double x, double y, const double* grid_lds = nullptr) {
  if (W.n_cells == 0) return -1;
  if (W.sg_enabled && grid_lds) {
    const double *x1 = grid_lds, *x0 = grid_lds + W.sg_ncol, *y1 = x0 + W.sg_ncol, *y0 = y1 + W.sg_nrow;
    int c = (int)auvp_floor((x - W.sg_x1_0) * W.sg_inv_dx) + 1, r = (int)auvp_floor((x - W.sg_y1_0) * W.sg_inv_dy) + 1;
    c = c < 0 ? 0 : (c > W.sg_ncol - 1 ? W.sg_ncol - 1 : c);
    r = r < 0 ? 0 : (r > W.sg_nrow - 1 ? W.sg_nrow - 1 : r);
    // the guess is the lower bound iff a1[g-1] < v <= a1[g]: checked without a loop; the stepping search only runs for
    // the rare lane whose guess is off (rounding next to an edge, a point outside the grid)
    const double xc1 = x1[c], xcm = c > 0 ? x1[c - 1] : -__builtin_inf();
    const double yr1 = y1[r], yrm = r > 0 ? y1[r - 1] : -__builtin_inf();
    if (!(xcm < x && xc1 >= x)) c = lower_bound_from(x1, W.sg_ncol, x, c);
    if (!(yrm < x && yr1 >= x)) r = lower_bound_from(y1, W.sg_nrow, x, r);
    if (c >= W.sg_ncol || r >= W.sg_nrow) return -1;
    return (x0[c] <= x && y0[r] <= y) ? r * W.sg_ncol + c : -1;
  }
  if (W.sg_enabled) {
    // row-major grid: first row with Y0[r] <= y and x <= Y1[r] (sic, cost.py:182), first column with X0[c] <= x <= X1[c].
    // Both guesses are checked with one independent 32-byte read each (the usual case: one round trip); a guess that
    // is off -- rounding next to a cell edge, or a point outside the grid -- falls back to the stepping search.
    int gc = (int)auvp_floor((x - W.sg_x1_0) * W.sg_inv_dx) + 1, gr = (int)auvp_floor((x - W.sg_y1_0) * W.sg_inv_dy) + 1;
    gc = gc < 0 ? 0 : (gc > W.sg_ncol - 1 ? W.sg_ncol - 1 : gc);
    gr = gr < 0 ? 0 : (gr > W.sg_nrow - 1 ? W.sg_nrow - 1 : gr);
    const double4 cc = reinterpret_cast<const double4*>(W.sg_col)[gc];
    const double4 rr = reinterpret_cast<const double4*>(W.sg_row)[gr];
    if (cc.x < x && cc.y >= x && rr.x < x && rr.y >= x) return (cc.z <= x && rr.z <= y) ? gr * W.sg_ncol + gc : -1;
    const int c = lower_bound_from(W.sg_x1, W.sg_ncol, x, gc);
    const int r = lower_bound_from(W.sg_y1, W.sg_nrow, x, gr);
    if (c >= W.sg_ncol || r >= W.sg_nrow) return -1;
    if (!(W.sg_x0[c] <= x) || !(W.sg_y0[r] <= y)) return -1;
    return r * W.sg_ncol + c;
  }
  double fb = auvp_floor((x - W.xb_x0) * W.xb_inv_w);
  int b = fb < 0.0 ? 0 : (fb >= (double)W.n_xbuckets ? W.n_xbuckets - 1 : (int)fb);
  if (W.rg_enabled) {
    // region of x: breakpoints in earlier buckets are < x, in later buckets > x (the bucket map is monotone)
    int lo = W.rg_first[b], hi = W.rg_first[b + 1];
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (W.rg_bp[mid] < x) lo = mid + 1; else hi = mid; }
    const int r = 2 * lo + ((lo < W.n_rg_bp && W.rg_bp[lo] == x) ? 1 : 0);
    const int s = W.rg_off[r], e = W.rg_off[r + 1];
    if (s == e) return -1;
    if (W.rg_pm[s] <= y) return W.rg_id[s];      // the usual case on a grid
    if (!(W.rg_pm[e - 1] <= y)) return -1;       // no candidate reaches down to y
    lo = s; hi = e - 1;                          // pm[lo] > y >= pm[hi]: first index with pm <= y
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (W.rg_pm[mid] <= y) hi = mid; else lo = mid; }
    return W.rg_id[hi];
  }
  int e = W.xb_off[b + 1];
  for (int k = W.xb_off[b]; k < e; k++) {
    const double4 d = reinterpret_cast<const double4*>(W.xb_data)[k];
    if (y < d.w) return -1;  // every remaining candidate has miny > y
    // sic: x is compared with maxy as well as maxx (path_planning/cost.py:182): d.y = min(maxx, maxy)
    if (x >= d.x && x <= d.y && y >= d.z) return W.xb_items[k];
  }
  return -1;
}

// ---- time-bin member lists (RrtBuffers::bin_items): direct-mapped head + on-demand 64-entry chunks --------------------
// (every index below fits 32 bits: the host refuses a bin_stride of 2^31 words or more)
struct BinLists {
  int32_t* direct;  // [K+1][AUVP_BIN_HEAD]
  int32_t* over;    // [n_over][64]
  int32_t* dir;     // [n_slots][K+1]
  int n_over, k1;
};

__device__ __forceinline__ BinLists bin_lists(const RrtBuffers& B, size_t episode, int K) {
  BinLists L;
  L.n_over = B.bin_over; L.k1 = K + 1;
  L.direct = B.bin_items + episode * (size_t)B.bin_stride;
  L.over = L.direct + (size_t)L.k1 * AUVP_BIN_HEAD;
  L.dir = L.over + (size_t)L.n_over * 64;
  return L;
}

// member k of bin b
__device__ __forceinline__ int bin_member(const BinLists& L, int b, int k) {
  if (k < AUVP_BIN_HEAD) return L.direct[b * AUVP_BIN_HEAD + k];
  const int kk = k - AUVP_BIN_HEAD;
  const int chunk = L.dir[(kk >> 6) * L.k1 + b];
  return L.over[chunk * 64 + (kk & 63)];
}

// where member c of bin b goes (nullptr: out of chunks).  The values are uniform over the episode's lanes, every lane
// keeps its copy of next_chunk; only `writer` touches memory.
__device__ __forceinline__ int32_t* bin_slot_for_append(const BinLists& L, int b, int c, int& next_chunk, bool writer) {
  if (c < AUVP_BIN_HEAD) return L.direct + (b * AUVP_BIN_HEAD + c);
  const int kk = c - AUVP_BIN_HEAD;
  int32_t* d = L.dir + ((kk >> 6) * L.k1 + b);
  int chunk;
  if ((kk & 63) == 0) {
    if (next_chunk >= L.n_over) return nullptr;
    chunk = next_chunk++;
    if (writer) *d = chunk;
  } else {
    chunk = *d;
  }
  return L.over + (chunk * 64 + (kk & 63));
}

struct CostAcc {
  double c2;                   // running shark term, summed in path order
  unsigned long long visited;  // habitat bit set
  int hits;                    // number of path points inside some habitat
};

// One path element of habitat_shark_cost_func (path_planning/cost.py:171-193): its shark term w3*prob
// (0.0 when it has none: x + 0.0 == x, an exact no-op in the ordered sum) and the first habitat that
// contains it (-1: none).  An element whose time stamp lies in no bin of [bin_lo, bin_hi) is skipped
// entirely by the reference: (0.0, -1).
// `hab_near` (wave-uniform): false when the caller knows that no habitat can contain the element (its bounding-box
// cull), which skips the habitat scan.
__device__ __forceinline__ void cost_element(const WorldDev& W, const RrtTables& S, int bin_lo, int bin_hi, double w3,
                                             double x, double y, double t, double& tv, int& hab, bool hab_near = true,
                                             const double* grid_lds = nullptr) {
  int tb = -1;
  if (W.bins_sorted) {
    // bins with non-decreasing ends: the first match is the first bin whose upper end reaches t, if it starts at or
    // before t -- a lower bound, checked against the table's own values (LDS)
    if (bin_hi > bin_lo) {
      int i = (int)auvp_floor((t - W.bins_t1_0) * W.bins_inv_len) + 1;
      i = i < bin_lo ? bin_lo : (i > bin_hi - 1 ? bin_hi - 1 : i);
      const double2 bi = *reinterpret_cast<const double2*>(&S.bins[i][0]);
      const double pm = i > bin_lo ? S.bins[i - 1][1] : -__builtin_inf();
      if (pm < t && bi.y >= t) {  // the guess is the lower bound (the usual case: no loop)
        if (bi.x <= t) tb = i;
      } else {
        while (i > bin_lo && S.bins[i - 1][1] >= t) i--;
        while (i < bin_hi && S.bins[i][1] < t) i++;
        if (i < bin_hi && S.bins[i][0] <= t) tb = i;
      }
    }
  } else {
    // first matching bin in table order, scanned backwards without early exits (the last overwrite is the first
    // match): the table reads (one LDS address for all lanes) are then independent of the compares
    for (int b = bin_hi - 1; b >= bin_lo; b--) {
      const double2 r = *reinterpret_cast<const double2*>(&S.bins[b][0]);
      tb = (t >= r.x && t <= r.y) ? b : tb;
    }
  }
  tv = 0.0;
  hab = -1;
  if (tb >= 0) {
    int c = cell_lookup(W, x, y, grid_lds);
    if (c >= 0) tv = w3 * W.prob[(size_t)tb * W.n_cells + c];
    if (hab_near) {
      if (W.hg_n > 0) {
        // habitat mask grid: the few habitats whose bounding square touches the point's cell, in list order
        const double fx = auvp_floor((x - W.hg_x0) * W.hg_inv_w), fy = auvp_floor((y - W.hg_y0) * W.hg_inv_h);
        unsigned long long m = 0ull;
        if (fx >= 0.0 && fy >= 0.0 && fx < (double)W.hg_n && fy < (double)W.hg_n) m = W.hg_mask[(int)fy * W.hg_n + (int)fx];
        while (m) {
          const int h = __ffsll((long long)m) - 1;
          m &= m - 1ull;
          const double2 hxy = *reinterpret_cast<const double2*>(&S.hab[h][0]);
          const double ddx = hxy.x - x, ddy = hxy.y - y;
          if (ddx * ddx + ddy * ddy <= S.hab[h][3]) { hab = h; break; }
        }
      } else {
        for (int h = W.n_habitats - 1; h >= 0; h--) {
          // dist <= size  <=>  d2 <= T(size): same decision as RN(sqrt(d2)) <= size, no sqrt
          const double2 hxy = *reinterpret_cast<const double2*>(&S.hab[h][0]);
          const double ddx = hxy.x - x, ddy = hxy.y - y;
          hab = (ddx * ddx + ddy * ddy <= S.hab[h][3]) ? h : hab;
        }
      }
    }
  }
}

// One pass of up to 64 evaluated elements (lane order = path order) into the running cost.
// `term` is 64 doubles of wave LDS.
__device__ __forceinline__ void cost_accumulate(int n_valid, double tv, int hab, double* term, CostAcc& acc) {
  const int lane = lane_id();
  term[lane] = tv;
  unsigned long long hm = wave_ballot(hab >= 0);
  acc.hits += __popcll(hm);
  unsigned long long vis = acc.visited;
  while (hm) {  // at most H distinct habitats; usually one or two per pass
    int l = __ffsll((long long)hm) - 1;
    int h = __builtin_amdgcn_readlane(hab, l);
    vis |= (1ull << h);
    hm &= ~wave_ballot(hab == h);
  }
  acc.visited = vis;
  wave_sync();
  // cost[2] += w3*prob in path order: one dependent add per element, LDS reads run ahead
  // Entries past n_valid hold +0.0 (every lane stored its tv) and c2 is never -0.0, so adding them is an exact
  // no-op: the sum runs in batches of 8 -- 8 LDS reads in flight, then 8 dependent adds.
  double c2 = acc.c2;
  for (int i = 0; i < n_valid; i += 8) {
    double2 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const double2*>(term + i + 2 * k);
#pragma unroll
    for (int k = 0; k < 4; k++) { c2 = c2 + v[k].x; c2 = c2 + v[k].y; }
  }
  acc.c2 = c2;
  wave_sync();
}

// cost_element in two halves, for callers that evaluate several elements per lane (the leaf pass): cost_pre does the index
// arithmetic (time bin, cell, habitat-mask cell; LDS tables only when grid_lds is given), the CALLER then issues the two
// global reads of every element back to back -- prob[tb][c] and the habitat mask word -- and cost_post finishes.  Same
// decisions and the same floats as cost_element (which stays the one-element form).
struct CostPre {
  int tb, c, midx;  // time bin (-1: none: the element is skipped), cell (-1: none), habitat-mask cell (-1: outside / no mask grid)
};
__device__ __forceinline__ CostPre cost_pre(const WorldDev& W, const RrtTables& S, int bin_lo, int bin_hi, double x, double y, double t,
                                            const double* grid_lds) {
  CostPre q;
  q.tb = -1; q.c = -1; q.midx = -1;
  if (W.bins_sorted) {
    if (bin_hi > bin_lo) {
      int i = (int)auvp_floor((t - W.bins_t1_0) * W.bins_inv_len) + 1;
      i = i < bin_lo ? bin_lo : (i > bin_hi - 1 ? bin_hi - 1 : i);
      const double2 bi = *reinterpret_cast<const double2*>(&S.bins[i][0]);
      const double pm = i > bin_lo ? S.bins[i - 1][1] : -__builtin_inf();
      if (pm < t && bi.y >= t) {
        if (bi.x <= t) q.tb = i;
      } else {
        while (i > bin_lo && S.bins[i - 1][1] >= t) i--;
        while (i < bin_hi && S.bins[i][1] < t) i++;
        if (i < bin_hi && S.bins[i][0] <= t) q.tb = i;
      }
    }
  } else {
    for (int b = bin_hi - 1; b >= bin_lo; b--) {
      const double2 r = *reinterpret_cast<const double2*>(&S.bins[b][0]);
      q.tb = (t >= r.x && t <= r.y) ? b : q.tb;
    }
  }
  if (q.tb >= 0) {
    q.c = cell_lookup(W, x, y, grid_lds);
    if (W.hg_n > 0) {
      const double fx = auvp_floor((x - W.hg_x0) * W.hg_inv_w), fy = auvp_floor((y - W.hg_y0) * W.hg_inv_h);
      if (fx >= 0.0 && fy >= 0.0 && fx < (double)W.hg_n && fy < (double)W.hg_n) q.midx = (int)fy * W.hg_n + (int)fx;
    }
  }
  return q;
}
__device__ __forceinline__ void cost_post(const WorldDev& W, const RrtTables& S, double w3, double x, double y, const CostPre& q,
                                          double prob, unsigned long long m, double& tv, int& hab) {
  tv = 0.0;
  hab = -1;
  if (q.tb >= 0) {
    if (q.c >= 0) tv = w3 * prob;
    if (W.hg_n > 0) {
      while (m) {
        const int h = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        const double2 hxy = *reinterpret_cast<const double2*>(&S.hab[h][0]);
        const double ddx = hxy.x - x, ddy = hxy.y - y;
        if (ddx * ddx + ddy * ddy <= S.hab[h][3]) { hab = h; break; }
      }
    } else {
      for (int h = W.n_habitats - 1; h >= 0; h--) {
        const double2 hxy = *reinterpret_cast<const double2*>(&S.hab[h][0]);
        const double ddx = hxy.x - x, ddy = hxy.y - y;
        hab = (ddx * ddx + ddy * ddy <= S.hab[h][3]) ? h : hab;
      }
    }
  }
}

// evaluate + accumulate up to 64 elements given by value (the standalone cost probe)
__device__ __forceinline__ void cost_pass(const WorldDev& W, const RrtTables& S, int bin_lo, int bin_hi, double w3,
                                          int n_valid, double x, double y, double t, double* term, CostAcc& acc) {
  double tv = 0.0;
  int hab = -1;
  if (lane_id() < n_valid) cost_element(W, S, bin_lo, bin_hi, w3, x, y, t, tv, hab);
  cost_accumulate(n_valid, tv, hab, term, acc);
}

// ---- get_closest_mps (path_planning/rrt_dubins.py:505-513) as a streaming scan ------------------------------------------
// The reference walks mps_list and keeps the FIRST node with the smallest dist = RN(sqrt(dx**2 + dy**2)) (strict <).  The
// scan reads the episode's contiguous x,y mirror (16 B per node: one global_load_dwordx4 per lane, RRT_NN_UNROLL of them in
// flight per lane, 1 KB per wave-instruction) and ranks by the SQUARED distance -- sqrt is monotone, so the minimum of
// RN(sqrt(d2)) is taken where d2 is smallest -- with two running minima per lane:
//   bd / bt  smallest d2 of the lane's nodes and the block it first appeared in (strict <: the first of equal values)
//   sd       second smallest DISTINCT d2 of the lane's nodes (a value equal to the running minimum is skipped: duplicated
//            positions are common -- a steer with zero sub-arcs copies its parent -- and a lane that holds the minimum
//            twice must still see a third, slightly larger value)
// Afterwards gmin = wave minimum of bd.  Two different d2 can still round to the same sqrt; such a value lies within a few
// ulps above gmin.  A lane's values other than its smallest are all >= its sd (sd is the smallest value that differs from
// the final bd: a value skipped as "equal to the running minimum" either equals the final bd or was that minimum when a
// smaller one replaced it, which put it into sd), so if no lane's bd or sd falls in (gmin, gmin (1 + 2^-49)] no node at all
// has its d2 there: every node with RN(sqrt(d2)) == RN(sqrt(gmin)) has d2 == gmin exactly, and the answer is the smallest
// index among them.
// Otherwise (never observed: it needs two nodes equidistant from the sample to 1e-15) the scan is repeated the reference's
// way, sqrt per node.  Entries past the tree's end hold +inf: a block is always read whole.
constexpr int RRT_NN_UNROLL = 8;
constexpr int RRT_NN_BLOCK = 64 * RRT_NN_UNROLL;
__host__ __device__ inline long long rrt_nn_stride(int cap_nodes) {
  return ((long long)cap_nodes + RRT_NN_BLOCK - 1) / RRT_NN_BLOCK * RRT_NN_BLOCK;
}

// `force_exact` (wave-uniform): take the sqrt-per-node path regardless (tests); `*slow` reports which path ran.
__device__ __forceinline__ int nn_closest(const double2* __restrict__ xy, int n_nodes, double rx, double ry, bool force_exact = false,
                                          int* slow = nullptr) {
  const int lane = lane_id();
  const double inf = __builtin_inf();
  double bd = inf, sd = inf;
  int bt = 0;
  for (int base = 0; base < n_nodes; base += RRT_NN_BLOCK) {
    double2 q[RRT_NN_UNROLL];
#pragma unroll
    for (int u = 0; u < RRT_NN_UNROLL; u++) q[u] = xy[base + u * 64 + lane];
#pragma unroll
    for (int u = 0; u < RRT_NN_UNROLL; u++) {
      const double ddx = rx - q[u].x, ddy = ry - q[u].y;
      const double d2 = ddx * ddx + ddy * ddy;
      const bool lt = d2 < bd;
      // the larger of {old minimum, newcomer} competes for second place -- unless they are equal (second DISTINCT value)
      sd = (d2 != bd) ? __builtin_fmin(sd, __builtin_fmax(bd, d2)) : sd;
      bd = lt ? d2 : bd;
      bt = lt ? base + u * 64 : bt;
    }
  }
  const double gmin = wave_min_f64(bd);
  const double band = gmin + gmin * 0x1p-49;
  const bool suspect = (bd > gmin && bd <= band) || (sd > gmin && sd <= band);
  int cand = 0x7fffffff;
  const bool exact = wave_any(suspect) || force_exact;
  if (slow) *slow = exact ? 1 : 0;
  if (!exact) {
    const unsigned long long em = wave_ballot(bd == gmin);
    if (__popcll(em) == 1) return uni(__builtin_amdgcn_readlane(bt, __ffsll((long long)em) - 1) + (__ffsll((long long)em) - 1));
    cand = (bd == gmin) ? bt + lane : 0x7fffffff;
  } else {
    // the reference's own ranking
    double bs = inf;
    for (int m = lane; m < n_nodes; m += 64) {
      const double2 q = xy[m];
      const double ddx = rx - q.x, ddy = ry - q.y;
      const double d = auvp_sqrt(ddx * ddx + ddy * ddy);
      if (d < bs) { bs = d; cand = m; }
    }
    const double smin = wave_min_f64(bs);
    cand = (bs == smin) ? cand : 0x7fffffff;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const int t = __shfl_xor(cand, o, 64);
    cand = t < cand ? t : cand;
  }
  return uni(cand);
}

template <int J, int MODE, bool DIAG>
__global__ __launch_bounds__(RRT_X_WAVES * 64, (MODE == 2 ? (J <= 4 ? 5 : 2) : (J <= 4 ? 6 : 2))) void rrt_explore_kernel(WorldDev W, RrtParamsDev P, RrtBuffers B,
                                                                     int n_episodes, int max_pts) {
  extern __shared__ __align__(16) unsigned char smem[];
  const RrtTables S = rrt_tables_view(smem, W.n_habitats, W.n_poly);
  const int wave = uni((int)(threadIdx.x >> 6));  // wave-uniform: keeps every per-episode address scalar
  const int lane = lane_id();
  const int nfreq = (int)P.freq;
  const int xw = (int)(blockDim.x >> 6);  // episodes per workgroup (RRT_X_WAVES; fewer for batches that would leave CUs idle)
  const RrtLdsPlan plan = rrt_lds_plan(P.K, max_pts, nfreq, J * 64, rrt_tables_bytes(W.n_habitats, W.n_poly, W.n_bins), xw);
  const int C = plan.chunk;
  unsigned char* wbase = smem + plan.tables + (size_t)wave * plan.per_wave;
  double* scratch = reinterpret_cast<double*>(wbase);
  double* u_win = scratch;                         // [3C+3]      (steer, phase 1)
  double* inc = scratch;                           // [(C+1)*4]   (steer, phase 2: aliases u_win)
  double* sc = scratch + (size_t)4 * ((C + 2) & ~1);  // [(C+1)*2]
  double* phi_l = sc + (size_t)2 * ((C + 2) & ~1);    // [CS]
  uint32_t* mt = reinterpret_cast<uint32_t*>(wbase + plan.scratch);
  double(*pts)[2] = reinterpret_cast<double(*)[2]>(wbase + plan.scratch + plan.mt);
  int32_t* bin_count = reinterpret_cast<int32_t*>(wbase + plan.scratch + plan.mt + plan.pts);

  // ---- stage the shared world tables (whole workgroup) ----
  rrt_tables_stage(S, W);
  // The fp64 parameters are read from an LDS copy inside the expansion loop: as kernel arguments they sit in ~24
  // scalar registers for the whole loop, get spilled to vector-register lanes and cost a VALU instruction per
  // reload, whereas an LDS read feeds a vector operand directly (109 -> 92 spilled SGPRs, -4 % VALU instructions).
  if (threadIdx.x == 0) *S.params = P;
  const RrtParamsDev& Q = *S.params;
  // obstacles: SoA tile shared by the episodes of the workgroup, padded to J*64 (slot j, lane l =
  // obstacle j*64 + l); with the bounding-box cull most slots are only touched by 3 reads per expansion
  double* olx = reinterpret_cast<double*>(smem + plan.tables + (size_t)xw * plan.per_wave);
  double* oly = olx + J * 64;
  double* olt = oly + J * 64;
  // cull radius: >= sqrt(T) with margin, rounded up to a float (it only has to be conservative; 1 KB less LDS
  // at 256 obstacles); -inf where nothing can collide
  float* olr = reinterpret_cast<float*>(olt + J * 64);
  for (int i = threadIdx.x; i < J * 64; i += blockDim.x) {
    const bool ok = i < W.n_obstacles;
    const double t = ok ? W.ot[i] : -1.0;  // d2 >= 0 > -1: padding never collides
    olx[i] = ok ? W.ox[i] : 0.0;
    oly[i] = ok ? W.oy[i] : 0.0;
    olt[i] = t;
    const double rd = t >= 0.0 ? auvp_sqrt(t) * (1.0 + 0x1p-30) + 0x1p-40 : -__builtin_inf();
    float rf = (float)rd;
    if ((double)rf < rd) rf = __uint_as_float(__float_as_uint(rf) + 1u);  // rd > 0 here: the next float up
    olr[i] = rf;
  }
  __syncthreads();

  const int ep = (int)blockIdx.x * xw + wave;
  if (ep >= n_episodes) return;  // no workgroup barrier after this point

  // ---- per-episode views (scalar bases) ----
  const int capn = B.cap_nodes, capp = B.cap_points, bcap = B.bin_cap;
  double* nodeF = B.node_f + (size_t)ep * capn * 8;                       // [capn][8] x,y,theta,t,length
  int4* nodeI = reinterpret_cast<int4*>(B.node_i) + (size_t)ep * capn;    // plan_iter,parent,pt_off,pt_cnt
  uint8_t* nodeQ = B.node_q + (size_t)ep * capn;
  double2* nodeXY = MODE == 2 ? reinterpret_cast<double2*>(B.node_xy) + (size_t)ep * (size_t)B.xy_stride : nullptr;
  unsigned long long nn_scanned = 0ull;
  double* ptF = B.points + (size_t)ep * capp * 6;                         // [capp][3] x,y,t then [capp][3] theta,v,length
  const BinLists bins = bin_lists(B, (size_t)ep, P.K);
  int next_chunk = 0;
  const double* init = B.init + (size_t)ep * 6;
  const int K = P.K;
  const bool log_it = DIAG && (P.flags & 1) != 0;
  const size_t logb = (size_t)ep * P.max_iter;

  WaveRng rng;
  rng.s = mt;
  for (int i = lane; i < 624; i += 64) mt[i] = B.mt[(size_t)ep * 624 + i];
  {
    // words [idx, 624) of the incoming state are generated and unconsumed (CPython's index)
    int idx = B.mt_index ? uni(B.mt_index[ep]) : 624;
    idx = idx < 0 ? 0 : (idx > 624 ? 624 : idx);
    rng.pslot = idx == 624 ? 0u : (uint32_t)idx;
    rng.avail = (uint32_t)(624 - idx);
    rng.drawn = 0ull;
  }
  for (int i = lane; i < K + 2; i += 64) bin_count[i] = 0;
  wave_sync();

  if (MODE == 2) {  // the x,y mirror starts out as +inf everywhere (a scan block is always read whole)
    const double inf = __builtin_inf();
    for (long long i = lane; i < B.xy_stride; i += 64) nodeXY[i] = make_double2(inf, inf);
  }
  // mps_list = [initial]; time_bin[bin_interval].append(initial)  (:105,:114)
  if (lane == 0) {
    if (MODE == 2) nodeXY[0] = make_double2(init[0], init[1]);
    nodeF[0] = init[0]; nodeF[1] = init[1]; nodeF[2] = init[2]; nodeF[3] = init[3]; nodeF[4] = init[5];
    nodeI[0] = make_int4(0, -1, 0, 0);
    nodeQ[0] = 0;  // the start state is never a leaf candidate
    if (MODE == 0) { bins.direct[(K >= 1 ? 1 : 0) * AUVP_BIN_HEAD] = 0; bin_count[K >= 1 ? 1 : 0] = 1; }
  }
  wave_sync();
  int n_nodes = 1, n_points = 0, status = 0;
  int it = 0, n_cand = 0;  // n_cand: obstacles that survived the cull (exact tests run), whole episode
  // optional per-phase shader-clock accounting (AUVP_FLAG_PHASE_CLOCKS): select, steer, collision, accept
  const bool clk = DIAG && (P.flags & 4) != 0;
  unsigned long long tph[5] = {0, 0, 0, 0, 0}, t_prev = 0;
#define AUVP_PHASE(i) do { if (clk) { unsigned long long t_now = __builtin_amdgcn_s_memtime(); tph[i] += t_now - t_prev; t_prev = t_now; } } while (0)

  for (; it < P.max_iter; it++) {
    if (log_it && lane == 0) {
      B.it_parent[logb + it] = -1;
      B.it_accepted[logb + it] = 0;
      B.it_npath[logb + it] = 0;
    }
    // ------------------------------------------------------------ parent selection (:121-139)
    if (clk) t_prev = __builtin_amdgcn_s_memtime();
    int par = 0, par_v = 0;
    // The next 64 random() values of the stream are tempered in one pass (lane j holds number j);
    // the iteration's scalar draws are read out of that window and `base` counts how many of them
    // the selection + n_expand draws consumed.  The steer window continues in the same buffer.
    int base = 0;
    double u_me;
    if (MODE == 0) {
      int rb = 0, cnt = 0, f = -1;
      for (;;) {
        rng_ensure(rng, 128u);
        u_me = rng_random_at(rng, (uint32_t)lane);
        // ran_bin = int(uniform(1, K+1)) until that bin is non-empty (:123-125): every lane tries its
        // own draw, the first success in stream order wins; a key beyond K before it is a KeyError
        const int rbj = (int)py_uniform(1.0, (double)(K + 1), u_me);
        const bool cand = lane < 60;  // leave room for the two draws that follow the successful one
        const bool badkey = cand && rbj > K;
        const int cj = (cand && !badkey) ? bin_count[rbj] : 0;
        const unsigned long long okm = wave_ballot(cj != 0), badm = wave_ballot(badkey);
        const int fo = okm ? (__ffsll((long long)okm) - 1) : 64, fb = badm ? (__ffsll((long long)badm) - 1) : 64;
        if (fb < fo) { status = -5; break; }
        if (fo < 64) {
          f = fo;
          rb = __builtin_amdgcn_readlane(rbj, fo);
          cnt = __builtin_amdgcn_readlane(cj, fo);
          break;
        }
        rng_advance_words(rng, 120u);  // 60 unsuccessful draws (only while most bins are still empty)
      }
      if (uni(status)) break;  // (uni: the compiler cannot see that status is wave-uniform)
      const int ri = uni((int)py_uniform(0.0, (double)cnt, readlane_f64(u_me, f + 1)));
      par_v = bin_member(bins, rb, ri);  // every lane reads the same word; made uniform when the record is fetched
      base = uni(f + 2);
    } else if (MODE == 1) {
      double u = rng_next_random(rng);
      double ran_time = py_uniform(0.0, Q.max_plan_time * Q.freq, u);
      int lo = 0, hi = n_nodes;  // list slicing of get_closest_mps_time (:515-528)
      while (hi - lo > 3) {
        int n = hi - lo;
        double ld = auvp_fabs((double)nodeI[lo + n / 2 - 1].x - ran_time);
        double rd = auvp_fabs((double)nodeI[lo + n / 2 + 1].x - ran_time);
        if (ld >= rd) lo += n / 2; else hi = lo + n / 2;
        lo = uni(lo); hi = uni(hi);
      }
      par = lo;
      par_v = par;
      if (nodeF[(size_t)par * 8 + 3] > Q.max_traj_time) continue;
    } else {
      // get_random_mps (:333-343): x, y, theta, size draws; only x,y are used
      rng_ensure(rng, 8);
      double rx = py_uniform(W.bb[0], W.bb[2], rng_random_at(rng, 0));
      double ry = py_uniform(W.bb[1], W.bb[3], rng_random_at(rng, 1));
      rng_advance_words(rng, 8);
      // get_closest_mps (:505-513): first index with the smallest RN(sqrt(d2)), a streaming scan of the x,y mirror
      nn_scanned += (unsigned long long)n_nodes;
      par = nn_closest(nodeXY, n_nodes, readfirst_f64(rx), readfirst_f64(ry), (P.flags & AUVP_KFLAG_NN_EXACT) != 0);
      par_v = par;
      if (nodeF[(size_t)par * 8 + 3] > Q.max_traj_time) continue;
    }

    AUVP_PHASE(0);
    // ------------------------------------------------------------ steer (:252-295)
    // The parent's record is fetched as late as possible -- right before the first chunk's theta chain: the random
    // window, the "taken" predicate, the draw-offset fixed point and the arc radii need nothing of it, and run while the
    // parent's id and then its record are on their way from memory.
    double cx = 0.0, cy = 0.0, cth = 0.0, ctt = 0.0, clen = 0.0;
    double px0 = 0.0, py0 = 0.0, clen0 = 0.0;  // the parent's end: centre of the collision cull's box
    auto fetch_parent = [&]() {
      par = uni(par_v);
      const double2 a = *reinterpret_cast<const double2*>(nodeF + (size_t)par * 8);
      const double2 b = *reinterpret_cast<const double2*>(nodeF + (size_t)par * 8 + 2);
      cx = readfirst_f64(a.x); cy = readfirst_f64(a.y); cth = readfirst_f64(b.x); ctt = readfirst_f64(b.y);
      clen = readfirst_f64(nodeF[(size_t)par * 8 + 4]);
      px0 = cx; py0 = cy; clen0 = clen;
      if (lane == 0) { pts[0][0] = cx; pts[0][1] = cy; }
    };
    if (MODE != 0) {  // modes 1/2 consumed their selection draws one by one; open the window here
      rng_ensure(rng, 128u);
      u_me = rng_random_at(rng, (uint32_t)lane);
      base = 0;
    }
    const int n_total = uni((int)auvp_floor(py_uniform(0.0, Q.freq, readlane_f64(u_me, base)) / 1));
    base += 1;
    int cnt = 0;  // appended path points
    if (n_total == 0) fetch_parent();
    bool cap_err = false;
    for (int c0 = 0; c0 < n_total; c0 += C) {
      const int n = (n_total - c0) < C ? (n_total - c0) : C;
      const int nwin = 3 * n;
      // window entry j of this chunk = random() number base + j of the stream
      if (c0 == 0) {
        u_win[lane] = u_me;
        if (base + nwin > 64) {
          rng_ensure(rng, (uint32_t)(2 * (base + nwin)));
          for (int jj = 64 + lane; jj < base + nwin; jj += 64) u_win[jj] = rng_random_at(rng, (uint32_t)jj);
        }
      } else {
        base = 0;
        rng_ensure(rng, (uint32_t)(2 * nwin));
        for (int jj = lane; jj < nwin; jj += 64) u_win[jj] = rng_random_at(rng, (uint32_t)jj);
      }
      wave_sync();
      const double* uw = u_win + base;
      // "taken" predicate for every possible start offset
      unsigned long long msk[3] = {0ull, 0ull, 0ull};
#pragma unroll
      for (int t = 0; t < 3; t++) {
        if (64 * t + 1 < nwin) {  // wave-uniform: short steers need one pass only
          int jj = lane + 64 * t;
          bool f = false;
          if (jj + 1 < nwin) {
            double dist = py_uniform(0.0, Q.dist_to_end, uw[jj]);
            double diff = py_uniform(-Q.diff_max, Q.diff_max, uw[jj + 1]);
            f = auvp_fabs(dist) > auvp_fabs(diff);
          }
          msk[t] = wave_ballot(f);
        }
      }
      // Where does sub-arc s start?  pos_s = 2s + (#taken among sub-arcs < s).  Fixed point of
      //   taken_s = T[2s + c_s],  c_s = popcount(taken below s)
      // started from "everything taken"; sub-arcs 0..k are exact after k+1 rounds, and in practice
      // the loop ends after (number of untaken sub-arcs + 1) rounds.
      const bool active = lane < n;
      int cbelow = lane;
      unsigned long long tmask;
      // this lane only ever looks at bits 2*lane .. 3*lane of the 192-bit predicate: cut that window out once
      unsigned long long win;
      {
        const int sh = 2 * lane;
        const unsigned long long lo = sh < 64 ? msk[0] : msk[1], hi = sh < 64 ? msk[1] : msk[2];
        const int s6 = sh & 63;
        win = (lo >> s6) | ((hi << 1) << (63 - s6));
      }
      // (votes as ballots of ONE compare each, the lanes that take no part made neutral through their data: a vote on
      // `active && x` costs two more vector instructions on this chain -- a 0 / 1 and its compare with zero)
      {
        const unsigned long long win_a = active ? win : 0ull;  // bits 0 .. lane are looked at (a chunk has up to 63 sub-arcs)
        const unsigned long long below_me = active ? ((1ull << lane) - 1ull) : 0ull;
        cbelow = active ? lane : 0;
        for (;;) {
          tmask = __builtin_amdgcn_uicmp((uint32_t)(win_a >> cbelow) & 1u, 0u, 33 /* != */);
          const int cnew = __popcll(tmask & below_me);
          const unsigned long long chg = __builtin_amdgcn_uicmp((unsigned)cnew, (unsigned)cbelow, 33 /* != */);
          cbelow = cnew;
          if (chg == 0ull) break;
        }
      }
      const int mypos = 2 * lane + cbelow;
      const int used = 2 * n + __popcll(tmask);
      const bool taken = (tmask >> lane) & 1ull;
      double radius = 0.0, phi = 0.0, vt = 1.0;
      if (taken) {
        double dist = py_uniform(0.0, Q.dist_to_end, uw[mypos]);
        double diff = py_uniform(-Q.diff_max, Q.diff_max, uw[mypos + 1]);
        double s1 = dist + diff, s2 = dist - diff;
        radius = auvp_div_plain(s1 + s2, -s1 + s2);
        phi = auvp_div_plain(s1 + s2, 2 * radius);
        vt = py_uniform(0.0, 2 * Q.v, uw[mypos + 2]);
      }
      wave_sync();  // the window is dead: its LDS becomes the steer scratch
      if (c0 == 0) fetch_parent();
      const int CS = (C + 2) & ~1;  // chain-major: chain c owns inc[c*CS .. c*CS+C], 16-byte aligned rows
      if (lane < CS) phi_l[lane] = phi;  // untaken / idle lanes add an exact 0.0
      wave_sync();
      // theta += phi, left to right, by one lane; prefix angles written back in place
      if (lane == 0) {
        double th = cth;
#pragma unroll 2
        for (int s = 0; s < n; s += 2) {  // two steps per 16-byte access; entries past n hold exact zeros
          double2 v = *reinterpret_cast<double2*>(phi_l + s);
          th = th + v.x; v.x = th;
          th = th + v.y; v.y = th;
          *reinterpret_cast<double2*>(phi_l + s) = v;
        }
      }
      wave_sync();
      const double myth = active ? phi_l[lane] : cth;  // idle lanes evaluate the chunk-entry angle
      double sn, cs;
      auvp_sincos_sk(myth, &sn, &cs);
      if (lane <= C) { sc[2 * lane] = sn; sc[2 * lane + 1] = cs; }
      wave_sync();
      double dx = 0.0, dy = 0.0, mv = 0.0, dt = 0.0;
      if (taken) {
        unsigned long long below = tmask & ((1ull << lane) - 1ull);
        int prev = below ? (63 - __clzll((long long)below)) : C;  // lane C is idle: entry angle
        double so = sc[2 * prev], co = sc[2 * prev + 1];
        dx = radius * (sn - so);
        dy = radius * (-cs + co);
        mv = auvp_sqrt_plain(dx * dx + dy * dy);
        dt = auvp_div_plain(mv, vt);
      }
      if (active) { inc[lane] = dx; inc[CS + lane] = dy; inc[2 * CS + lane] = dt; inc[3 * CS + lane] = mv; }
      else if (lane < CS) { inc[lane] = 0.0; inc[CS + lane] = 0.0; inc[2 * CS + lane] = 0.0; inc[3 * CS + lane] = 0.0; }
      wave_sync();
      // x += dx; y += dy; t += dt; length += movement: four serial chains, one lane each
      if (lane < 4) {
        double acc = lane == 0 ? cx : (lane == 1 ? cy : (lane == 2 ? ctt : clen));
        double* row = inc + lane * CS;
#pragma unroll 2
        for (int s = 0; s < n; s += 2) {  // two steps per 16-byte access (the row is zero past n: adding 0.0 changes nothing)
          double2 v = *reinterpret_cast<double2*>(row + s);
          acc = acc + v.x; v.x = acc;
          acc = acc + v.y; v.y = acc;
          *reinterpret_cast<double2*>(row + s) = v;
        }
      }
      wave_sync();
      double mx = 0.0, my = 0.0, mt_ = 0.0, ml = 0.0;
      if (active) { mx = inc[lane]; my = inc[CS + lane]; mt_ = inc[2 * CS + lane]; ml = inc[3 * CS + lane]; }
      const bool app = taken && (mv >= Q.min_dist);
      const unsigned long long amask = wave_ballot(app);
      const int napp = __popcll(amask);
      if (n_points + cnt + napp > capp || cnt + napp + 1 > max_pts) { cap_err = true; break; }
      if (app) {
        int rank = __popcll(amask & ((1ull << lane) - 1ull));
        size_t gi = (size_t)(n_points + cnt + rank);  // speculative: committed only if the node is accepted
        // two 24-byte records per path point (auvp_types.h): what the leaf pass reads, and the rest
        double* ra = ptF + gi * 3;
        double* rb = ptF + (size_t)capp * 3 + gi * 3;
        *reinterpret_cast<double2*>(ra) = make_double2(mx, my); ra[2] = mt_;
        *reinterpret_cast<double2*>(rb) = make_double2(myth, vt); rb[2] = ml;
        pts[cnt + rank + 1][0] = mx;
        pts[cnt + rank + 1][1] = my;
      }
      cnt += napp;
      if (n > 0) {
        cx = readlane_f64(mx, n - 1); cy = readlane_f64(my, n - 1);
        ctt = readlane_f64(mt_, n - 1); clen = readlane_f64(ml, n - 1);
        cth = readlane_f64(myth, n - 1);
      }
      rng_advance_words(rng, (uint32_t)(2 * (base + used)));
      wave_sync();
    }
    if (n_total == 0) {
      rng_advance_words(rng, (uint32_t)(2 * base));  // selection + n_expand draws only
    }
    if (cap_err) { status = -2; break; }
    wave_sync();
    const int P_n = cnt + 1;

    AUVP_PHASE(1);
    // ------------------------------------------------------------ check_collision (:530-549)
    // Exact cull first: an obstacle whose effective disc does not reach the bounding box of the
    // path cannot be within T_i of any path point (a point in the box is at least as far from the
    // centre as the box is), so a slot of 64 obstacles with no candidate is skipped as a whole.
    // The cull only has to be conservative (a candidate slot runs the exact test below): the obstacle's
    // bounding square of half-width olr >= sqrt(T_i) against the box around its centre, both inflated by
    // 2^-30 relative -- eight orders of magnitude above any rounding in these few operations.
    // The box: every prefix position of the steer lies within the steer's total movement (the growth of the
    // length chain, which adds every sub-arc's chord) of the parent's end -- a square around the parent.  Looser than
    // the exact extent, but at these obstacle densities it still leaves well under one candidate per expansion, and
    // it costs nothing to maintain (tracking min/max in the serial chains was 30 instructions per expansion).
    const double reach = clen - clen0;
    const double bx0 = px0 - reach, by0 = py0 - reach, bx1 = px0 + reach, by1 = py0 + reach;
    double cxm = px0, cym = py0;
    const double slack = 0x1p-30 * (auvp_fabs(bx0) + auvp_fabs(bx1) + auvp_fabs(by0) + auvp_fabs(by1) + 1.0);
    double hx = reach + slack, hy = reach + slack;
    // lane = path point (the usual steer has < 64 of them): the few obstacles that survive the cull are tested one
    // at a time against every point at once, read back from the tile with a wave-uniform address
    int hit = 0;
    const bool pv0 = lane < P_n;
    double2 q0 = make_double2(0.0, 0.0);
    if (pv0) q0 = *reinterpret_cast<const double2*>(&pts[lane][0]);
    // Where the host expects dense obstacles (AUVP_KFLAG_TIGHT_CULL, set at launch) the cull box is the tight box of the
    // path points themselves (the parent's end is pts[0]): four wave reductions, far fewer exact tests
    if ((P.flags & AUVP_KFLAG_TIGHT_CULL) && P_n <= 64) {
      const double inf = __builtin_inf();
      const double mnx = wave_min_f64(pv0 ? q0.x : inf), mxx = wave_max_f64(pv0 ? q0.x : -inf);
      const double mny = wave_min_f64(pv0 ? q0.y : inf), mxy = wave_max_f64(pv0 ? q0.y : -inf);
      const double ts = 0x1p-30 * (auvp_fabs(mnx) + auvp_fabs(mxx) + auvp_fabs(mny) + auvp_fabs(mxy) + 1.0);
      cxm = (mnx + mxx) * 0.5; cym = (mny + mxy) * 0.5;
      hx = (mxx - mnx) * 0.5 + ts; hy = (mxy - mny) * 0.5 + ts;
    }
#pragma unroll
    for (int j = 0; j < J; j++) {
      const double oxj = olx[j * 64 + lane], oyj = oly[j * 64 + lane], orj = (double)olr[j * 64 + lane];
      const bool cand = !(auvp_fabs(oxj - cxm) > hx + orj || auvp_fabs(oyj - cym) > hy + orj);
      unsigned long long cm = wave_ballot(cand);
      n_cand += __popcll(cm);
      while (cm) {
        const int idx = uni(j * 64 + (__ffsll((long long)cm) - 1));
        cm &= cm - 1ull;
        const double ox = olx[idx], oy = oly[idx], ot = olt[idx];
        {
          const double ddx = q0.x - ox, ddy = q0.y - oy;
          const double d2 = ddx * ddx + ddy * ddy;
          hit |= (pv0 && d2 <= ot) ? 1 : 0;
        }
        for (int p = 64 + lane; p < P_n; p += 64) {  // only steers with freq > 63
          const double2 q = *reinterpret_cast<const double2*>(&pts[p][0]);
          const double ddx = q.x - ox, ddy = q.y - oy;
          const double d2 = ddx * ddx + ddy * ddy;
          hit |= (d2 <= ot) ? 1 : 0;
        }
      }
    }
    // polygon: when the path's bounding box lies strictly inside an axis-aligned rectangular boundary
    // every point is strictly inside it and the crossing test would say so too; skip it then
    const double* sb = S.world->safe_box;  // the LDS copy: four doubles less held in scalar registers
    const bool box_inside = W.has_safe_box && bx0 > sb[0] && by0 > sb[1] && bx1 < sb[2] && by1 < sb[3];
    const bool ok = !wave_any(hit != 0) && (box_inside || !any_point_outside(S.poly, W.n_poly, pts, P_n));
    if (log_it && lane == 0) {
      B.it_parent[logb + it] = par;
      B.it_accepted[logb + it] = ok ? 1 : 0;
      B.it_npath[logb + it] = P_n;
    }
    AUVP_PHASE(2);
    if (!ok) continue;
    if (n_nodes >= capn) { status = -2; break; }

    // ------------------------------------------------------------ accept (:144-151)
    const int me = n_nodes;
    if (lane == 0) nodeI[me] = make_int4(it, par, n_points, cnt);
    if (MODE == 0) {
      // curr_bin = (t // bin_interval + 1) * bin_interval, exact floor of the true quotient
      double q = auvp_floor(ctt * Q.inv_bin_interval);  // within one of the true floor; the remainder below settles it
      double r = auvp_fma(-q, Q.bin_interval, ctt);
      if (r < 0.0) q -= 1.0;
      else if (r >= Q.bin_interval) q += 1.0;
      double fi = q + 1.0;
      double curr_bin = fi * Q.bin_interval;
      bool over = curr_bin > Q.max_traj_time;
      if (!over || fi <= (double)K) {
        int bi = uni((int)fi);
        int c = over ? 0 : uni(bin_count[bi]);  // an overflowing regular key is reset first (:149-151)
        if (c >= bcap) { status = -2; break; }
        int32_t* slot = bin_slot_for_append(bins, bi, c, next_chunk, lane == 0);
        if (!slot) { status = -2; break; }
        if (lane == 0) { *slot = me; bin_count[bi] = c + 1; }
      }
      wave_sync();
    }

    if (lane == 0) {
      double* nf = nodeF + (size_t)me * 8;
      *reinterpret_cast<double2*>(nf) = make_double2(cx, cy);
      *reinterpret_cast<double2*>(nf + 2) = make_double2(cth, ctt);
      nf[4] = clen;
      if (MODE == 2) nodeXY[me] = make_double2(cx, cy);
      nodeQ[me] = ctt >= Q.max_traj_time - 30 ? 1 : 0;  // a qualifying leaf (:158); ranked by rrt_leaf_kernel
    }
    n_nodes++;
    n_points += cnt;
    AUVP_PHASE(3);
  }

  if (clk && lane == 0 && B.phase_clocks) {
    for (int i = 0; i < 5; i++) B.phase_clocks[(size_t)ep * 5 + i] = tph[i];
  }
  const unsigned long long drawn = rng.drawn;
  double after = rng_next_random(rng);
  for (int i = lane; i < K + 1; i += 64) B.bin_count[(size_t)ep * (K + 1) + i] = bin_count[i];
  if (lane == 0) {
    // the tree is complete; rrt_leaf_kernel ranks its qualifying leaves and fills in the rest of the record
    RrtSummary& s = B.summary[ep];
    s.status = status; s.n_nodes = n_nodes; s.n_points = n_points; s.n_leaves = 0;
    s.best_leaf = -1; s.best_path_len = 0; s.iters_run = it; s.n_candidates = n_cand;
    s.best_cost[0] = __builtin_inf(); s.best_cost[1] = 0.0; s.best_cost[2] = 0.0; s.best_cost[3] = 0.0;
    s.best_length = 0.0;
    s.rng_after = after; s.leaf_elems = 0; s.n_draw32 = drawn; s.nn_scanned = nn_scanned;
  }
}

// The qualifying-leaf bookkeeping of exploring (:158-171) for a finished tree, one wavefront per episode.
//
// The reference evaluates habitat_shark_cost_func on the path of every accepted node whose traj_time_stamp is within
// 30 s of max_traj_time, in creation order, and keeps the first strict minimum.  None of that feeds back into the
// tree, so it runs here, after the expansion kernel, in ONE sweep over the nodes in creation order, 64 per pass:
//
// 1. terms.  A path element's contribution to a leaf's cost -- w3*prob of its cell in its time bin, and the habitat it
//    lies in -- does not depend on the leaf (its bin is always part of the leaf's sub-dict), so it is evaluated once.
//    The path points of the 64 nodes of a pass are one contiguous run of records: lane = point (two per lane in flight,
//    coalesced reads); each term is added to its owner's slot in LDS (the owner = the last node of the pass whose
//    pt_off is <= the point index: a 6-step search over the pass's offsets; LDS atomics).  Then lane = node: its own
//    state's term.
// 2. running sums down the tree (a node's parent always comes earlier; parents inside the pass are resolved in rounds):
//      node_c = {elements inside some habitat, elements, visited-habitat bit set} of the root..node path (exact)
//      node_f[5] = S = sum of the shark terms of that path, in tree order
// 3. ranking.  cost[0] and cost[1] of a leaf follow from the exact integers as the reference computes them.  S differs
//    from the reference's leaf->root ordered sum only by rounding: both add the same L terms, so each is within
//    gamma_L * sum|term| of the exact sum (gamma_L = L u / (1 - L u), u = 2^-53), and |term| <= |w3| * max|prob|;
//    when the probabilities all have one sign (they do: they are probabilities) sum|term| is also |exact sum|, i.e.
//    about |S| -- a path without any shark term has S = 0 exactly, like the reference's sum, and ties stay ties.
//    That gives every leaf an interval [lo, hi] containing the reference's total.  A leaf whose lo is not below the
//    smallest hi of the leaves before it cannot be a strict minimum; the others (a handful per episode: the record
//    setters and exact ties) are re-summed in the reference's order -- leaf, its points last to first, its parent,
//    ... root, one rounded add per element, the terms evaluated again from the records -- and compared exactly like
//    the reference does.  With the leaf log requested every qualifying leaf is re-summed (the log holds the
//    reference's per-leaf costs).
// The order in which the LDS atomics add a node's terms is not defined; S only steers which leaves are re-summed, every
// reported number comes from the exact integers and the ordered re-summation.
constexpr int RRT_LEAF_WAVES = 4;  // episodes per workgroup of the leaf pass (they share the world tables in LDS)
#ifndef AUVP_LEAF_NPL
#define AUVP_LEAF_NPL 2
#endif
#ifndef AUVP_LEAF_WPE
#define AUVP_LEAF_WPE 4
#endif
__host__ __device__ inline int rrt_leaf_grid_lds_bytes(int sg_enabled, int ncol, int nrow) {
  const long long b = 16LL * ((long long)ncol + nrow);
  return (sg_enabled && b <= 48 * 1024) ? (int)b : 0;  // bigger grids are looked up in the global copy
}
// 32-bit words of the per-episode "ancestor of a qualifying leaf" bit set in LDS (0: the tree is too large, sweep it whole)
__host__ __device__ inline int rrt_leaf_mark_words(int cap_nodes) {
  const int w = (cap_nodes + 31) / 32;
  return w <= 4096 ? w : 0;
}

static __global__ __launch_bounds__(RRT_LEAF_WAVES * 64) __attribute__((amdgpu_waves_per_eu(AUVP_LEAF_WPE, AUVP_LEAF_WPE))) void rrt_leaf_kernel(WorldDev W, RrtParamsDev P, RrtBuffers B, int n_episodes,
                                                                      int mark_words) {
  __shared__ __align__(16) unsigned char tables[RRT_WORLD_BYTES + RRT_MAX_HAB * 32 + RRT_MAX_POLY * 16 + RRT_MAX_BINS * 16];
  __shared__ double w_term[RRT_LEAF_WAVES][64];
  __shared__ double w_S[RRT_LEAF_WAVES][64];
  __shared__ int32_t w_hits[RRT_LEAF_WAVES][64], w_elems[RRT_LEAF_WAVES][64], w_par[RRT_LEAF_WAVES][64], w_off[RRT_LEAF_WAVES][64];
  __shared__ int32_t w_cpos[RRT_LEAF_WAVES][64], w_ids[RRT_LEAF_WAVES][128];
  __shared__ unsigned long long w_vis[RRT_LEAF_WAVES][64];
  __shared__ __align__(16) uint8_t w_owner[RRT_LEAF_WAVES][2048];  // owner lane of every point of the pass in flight (re-summation: 256 doubles)
  extern __shared__ __align__(16) unsigned char leaf_dyn[];
  const RrtTables St = rrt_tables_view(tables, W.n_habitats, W.n_poly);
  const int wave = uni((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  rrt_tables_stage(St, W);
  const double* grid_lds = nullptr;
  if (rrt_leaf_grid_lds_bytes(W.sg_enabled, W.sg_ncol, W.sg_nrow)) {
    double* g = reinterpret_cast<double*>(leaf_dyn);
    for (int i = threadIdx.x; i < W.sg_ncol; i += blockDim.x) { g[i] = W.sg_x1[i]; g[W.sg_ncol + i] = W.sg_x0[i]; }
    for (int i = threadIdx.x; i < W.sg_nrow; i += blockDim.x) { g[2 * W.sg_ncol + i] = W.sg_y1[i]; g[2 * W.sg_ncol + W.sg_nrow + i] = W.sg_y0[i]; }
    grid_lds = g;
  }
  __syncthreads();
  const int ep = (int)blockIdx.x * RRT_LEAF_WAVES + wave;
  if (ep >= n_episodes) return;  // no workgroup barrier after this point
  uint32_t* mark = mark_words > 0 ? reinterpret_cast<uint32_t*>(leaf_dyn + rrt_leaf_grid_lds_bytes(W.sg_enabled, W.sg_ncol, W.sg_nrow)) +
                                        (size_t)wave * mark_words
                                  : nullptr;
  double* term = w_term[wave];
  double* c_S = w_S[wave];
  int32_t *c_hits = w_hits[wave], *c_elems = w_elems[wave], *c_par = w_par[wave], *c_off = w_off[wave];
  int32_t *c_cpos = w_cpos[wave], *c_ids = w_ids[wave];
  unsigned long long* c_vis = w_vis[wave];
  uint8_t* c_owner = w_owner[wave];
  const double (*s_bins)[2] = St.bins;
  RrtSummary& sum = B.summary[ep];
  const int status_in = sum.status;
  if (status_in < 0) return;  // the expansion failed: nothing to rank
  const int capn = B.cap_nodes;
  const size_t capp = (size_t)B.cap_points;
  double* nodeF = B.node_f + (size_t)ep * capn * 8;
  const int4* nodeI = reinterpret_cast<const int4*>(B.node_i) + (size_t)ep * capn;
  // the running sums of a node as one 32-byte record {S, hits | elements, visited mask, -}: a child fetches its parent's
  // sums with one read
  double4* nodeC = reinterpret_cast<double4*>(B.node_c) + (size_t)ep * capn;
  const double* ptF = B.points + (size_t)ep * capp * 6;
  const int n_nodes = sum.n_nodes;
  const bool log_leaf = (P.flags & 2) != 0 && B.leaf_cost != nullptr;
  const double init_t = B.init[(size_t)ep * 6 + 3];
  const double w1 = P.w[0], w2 = P.w[1], w3 = P.w[2];
  const double thresh = P.max_traj_time - 30;
  const double term_max = auvp_fabs(w3) * W.prob_absmax;  // |shark term of one element| <= this
  const bool w2_int = (w2 == auvp_rint(w2) && auvp_fabs(w2) < 1048576.0);
  const int H = W.n_habitats;
  wave_sync();

  // cost[0], cost[1] and the scaled shark term of a leaf, given the ordered or unordered sum `c2num`
  auto total_of = [&](int hits, unsigned long long vis, double ctt, double c2num, double& c0, double& c1, double& c2) {
    c0 = 0.0; c1 = 0.0; c2 = c2num;
    if (w2_int) c1 = w2 * (double)hits;  // exact: equals `hits` successive rounded additions of an integer weight
    else for (int h = 0; h < hits; h++) c1 = c1 + w2;
    if (ctt > 0) { c1 = c1 / ctt; c2 = c2 / ctt; }
    if (H != 0) c0 = w1 * (double)__popcll(vis) / (double)H;
    return ((0.0 + c0) + c1) + c2;
  };

  int n_leaves = 0, best_leaf = -1, best_L = 0;
  long long leaf_elems = 0;
  int st_nodes = 0, st_points = 0, st_resummed = 0, st_releaves = 0;  // RrtBuffers::leaf_stats (wave-uniform: scalar registers)
  double best_tot = __builtin_inf(), best_c0 = 0.0, best_c1 = 0.0, best_c2 = 0.0, best_len = 0.0;
  double min_hi = __builtin_inf();  // smallest upper bound among the qualifying leaves seen so far
  // ---------------------------------------------------------------- 0. which nodes matter
  // Only the qualifying leaves and their ancestors enter any cost (about a quarter of the bench's trees).  Backwards over
  // the nodes, 64 at a time: a node is marked if it qualifies (node_q, from the expansion) or a child marked it; it
  // marks its parent.  Children come after their parents, so one backward sweep settles every mark; parents inside the
  // block in flight are reached by repeating until no lane changes.
  const uint8_t* nodeQ = B.node_q + (size_t)ep * capn;
  if (mark) {
    for (int i = lane; i < mark_words; i += 64) mark[i] = 0u;
    wave_sync();
    // (the parent link and the leaf flag of the NEXT block are requested before this block's rounds: every block was a full
    // memory round trip on its own -- ~150 of them per episode, one after the other)
    const int n_top = ((n_nodes - 1) >> 6) << 6;
    int par_nx = (n_top + lane < n_nodes) ? nodeI[n_top + lane].y : -1;
    uint8_t q_nx = (n_top + lane < n_nodes) ? nodeQ[n_top + lane] : (uint8_t)0;
    for (int n0 = n_top; n0 >= 0; n0 -= 64) {
      const int m = n0 + lane;
      const bool live = m < n_nodes;
      const int par = par_nx;
      const uint8_t q_me = q_nx;
      if (n0 >= 64) { par_nx = nodeI[m - 64].y; q_nx = nodeQ[m - 64]; }  // (blocks below the top one are full)
      bool need = live && m >= 1 && q_me != 0;
      bool pushed = false;
      for (;;) {
        need = need || (live && ((mark[m >> 5] >> (m & 31)) & 1u));
        const bool push = need && !pushed && par >= 0;
        if (push) { atomicOr(&mark[par >> 5], 1u << (par & 31)); pushed = true; }
        // another round only if some lane just marked a parent inside this block
        if (!wave_any(push && par >= n0)) break;
        wave_sync();
      }
      if (need) atomicOr(&mark[m >> 5], 1u << (m & 31));
      wave_sync();
    }
  }

  // ---------------------------------------------------------------- the sweep: marked nodes in creation order, 64 per pass
  int qn = 0, scan = 0;  // ids waiting in c_ids[0..qn); next block of nodes to look at
  for (;;) {
    while (qn < 64 && scan < n_nodes) {
      const int mm = scan + lane;
      const bool f = mm < n_nodes && (!mark || ((mark[mm >> 5] >> (mm & 31)) & 1u));
      const unsigned long long fm = wave_ballot(f);
      if (f) c_ids[qn + __popcll(fm & ((1ull << lane) - 1ull))] = mm;
      qn += __popcll(fm);
      scan += 64;
    }
    wave_sync();
    if (qn == 0) break;
    const int nlive = qn < 64 ? qn : 64;
    const bool live = lane < nlive;
    const int m = live ? c_ids[lane] : 0x7fffffff;
    const int first_id = __builtin_amdgcn_readfirstlane(m);
    int4 r = make_int4(0, -1, 0, 0);
    if (live) r = nodeI[m];
    // ---------------------------------------------------------------- 1. terms of the pass's path elements
    // the runs of the pass's nodes, packed: point slot j of the pass = point c_off[o] + (j - c_cpos[o]) of its owner o
    int incl = live ? r.w : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    const int cpos = incl - (live ? r.w : 0);
    const int n_slots = __builtin_amdgcn_readlane(incl, 63);
    st_nodes = uni(st_nodes + nlive); st_points = uni(st_points + n_slots);
    c_cpos[lane] = live ? cpos : 0x7fffffff;
    c_off[lane] = r.z;
    c_S[lane] = 0.0; c_hits[lane] = 0; c_vis[lane] = 0ull;
    // every node marks its own slots: owner[slot] = its lane (2048 slots here; the search below serves longer passes)
    const bool own_tab = n_slots <= 2048;
    if (own_tab && live)
      for (int k = 0; k < r.w; k++) c_owner[cpos + k] = (uint8_t)lane;
    wave_sync();
    // the node's own record and its parent's running sums are requested now; they are used after the point rounds
    double2 n_xy = make_double2(0.0, 0.0), n_tl = make_double2(0.0, 0.0);
    if (live) {
      n_xy = *reinterpret_cast<const double2*>(nodeF + (size_t)m * 8);
      n_tl = *reinterpret_cast<const double2*>(nodeF + (size_t)m * 8 + 3);  // traj_t, length (unaligned pair)
    }
    const bool par_before = live && r.y >= 0 && r.y < first_id;
    double4 par_rec = make_double4(0.0, 0.0, 0.0, 0.0);
    if (par_before) par_rec = nodeC[r.y];
    // Point rounds, two points per lane and round, software-pipelined: the records of round k + 1 are requested before
    // the terms of round k are evaluated.  owner = the node whose run holds the slot: from the table, or the last node
    // whose first slot is <= the slot (nodes without points share their successor's first slot and are skipped)
    struct Slot { bool v; int o; double2 xy; double t; };
    auto fetch = [&](int j) {
      Slot q;
      q.v = j < n_slots;
      q.o = 0;
      if (own_tab) q.o = q.v ? (int)c_owner[j] : 0;
      else {
#pragma unroll
        for (int st = 32; st >= 1; st >>= 1)
          if (q.o + st < 64 && c_cpos[q.o + st] <= j) q.o += st;
      }
      const int pidx = q.v ? c_off[q.o] + (j - c_cpos[q.o]) : 0;
      const double* rec = ptF + (size_t)pidx * 3;
      // (read once: non-temporal, so that the stream of point records does not push the probability table out of L2 -- 1-3 %)
      typedef double nt_f64x2 __attribute__((ext_vector_type(2)));
      const nt_f64x2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f64x2*>(rec));
      q.xy = make_double2(v.x, v.y);
      q.t = __builtin_nontemporal_load(rec + 2);
      return q;
    };
    constexpr int NPL = AUVP_LEAF_NPL;  // points per lane and round
    Slot sn[NPL];
#pragma unroll
    for (int u = 0; u < NPL; u++) sn[u] = fetch(u * 64 + lane);
    for (int j0 = 0; j0 < n_slots; j0 += 64 * NPL) {
      Slot c[NPL];
#pragma unroll
      for (int u = 0; u < NPL; u++) c[u] = sn[u];
      if (j0 + 64 * NPL < n_slots) {
#pragma unroll
        for (int u = 0; u < NPL; u++) sn[u] = fetch(j0 + 64 * NPL + u * 64 + lane);
      }
      double tv[NPL];
      int hb[NPL];
      {
        // every element's index arithmetic first, then their dependent global reads (prob, habitat mask) back to back
        CostPre q[NPL];
        double pr[NPL];
        unsigned long long mk[NPL];
#pragma unroll
        for (int u = 0; u < NPL; u++) {
          q[u].tb = -1; q[u].c = -1; q[u].midx = -1;
          if (c[u].v) q[u] = cost_pre(W, St, 0, W.n_bins, c[u].xy.x, c[u].xy.y, c[u].t, grid_lds);
        }
#pragma unroll
        for (int u = 0; u < NPL; u++) {
          pr[u] = 0.0; mk[u] = 0ull;
          if (q[u].c >= 0) pr[u] = W.prob[(size_t)q[u].tb * W.n_cells + q[u].c];
          if (q[u].midx >= 0) mk[u] = W.hg_mask[q[u].midx];
        }
#pragma unroll
        for (int u = 0; u < NPL; u++) cost_post(W, St, P.w[2], c[u].xy.x, c[u].xy.y, q[u], pr[u], mk[u], tv[u], hb[u]);
      }
#pragma unroll
      for (int u = 0; u < NPL; u++) {
        if (c[u].v) {
          if (tv[u] != 0.0) atomicAdd(&c_S[c[u].o], tv[u]);
          if (hb[u] >= 0) { atomicAdd(&c_hits[c[u].o], 1); atomicOr(&c_vis[c[u].o], 1ull << hb[u]); }
        }
      }
    }
    wave_sync();
    double own = c_S[lane], ntv = 0.0, ctt = 0.0, nlen = 0.0;
    int own_hits = c_hits[lane], nhab = -1;
    unsigned long long own_vis = c_vis[lane];
    if (live) {
      ctt = n_tl.x; nlen = n_tl.y;
      cost_element(W, St, 0, W.n_bins, P.w[2], n_xy.x, n_xy.y, ctt, ntv, nhab, true, grid_lds);
      own = own + ntv;
      if (nhab >= 0) { own_hits++; own_vis |= (1ull << nhab); }
    }
    // ---------------------------------------------------------------- 2. running sums down the tree
    // the parent's sums: from memory when it belongs to an earlier pass, else from the lanes of this one
    double pS = 0.0;
    int4 pc = make_int4(0, 0, 0, 0);
    unsigned long long pvis = 0ull;
    if (par_before) {
      const double4 pr = par_rec;
      pS = pr.x;
      const long long he = __double_as_longlong(pr.y);
      pc.x = (int)(he & 0xffffffffll); pc.y = (int)(he >> 32);
      pvis = (unsigned long long)__double_as_longlong(pr.z);
    }
    wave_sync();
    // a parent inside this pass: its lane = its position among the pass's ids (ascending; a marked node's parent is marked)
    int plane = 0;
    if (live && r.y >= first_id) {
#pragma unroll
      for (int st = 32; st >= 1; st >>= 1)
        if (plane + st < nlive && c_ids[plane + st] <= r.y) plane += st;
    }
    c_par[lane] = plane;
    c_S[lane] = pS + own; c_hits[lane] = pc.x + own_hits; c_elems[lane] = pc.y + r.w + 1; c_vis[lane] = pvis | own_vis;
    wave_sync();
    // parents inside this pass: a lane is ready once its parent's entry is final (a parent always has the smaller
    // index, so the lowest pending lane is ready in every round); all ready lanes add their parent's sums at once
    unsigned long long pending = wave_ballot(live && r.y >= first_id);
    while (pending) {
      const int p = c_par[lane];
      const bool mine = (pending >> lane) & 1ull;
      const bool ready = mine && !((pending >> (p & 63)) & 1ull);
      double aS = 0.0;
      int aH = 0, aE = 0;
      unsigned long long aV = 0ull;
      if (ready) { aS = c_S[p]; aH = c_hits[p]; aE = c_elems[p]; aV = c_vis[p]; }
      wave_sync();
      if (ready) { c_S[lane] = aS + c_S[lane]; c_hits[lane] += aH; c_elems[lane] += aE; c_vis[lane] |= aV; }
      wave_sync();
      pending &= ~wave_ballot(ready);
    }
    const double S = c_S[lane];
    const int hits = c_hits[lane], elems = c_elems[lane];
    const unsigned long long vis = c_vis[lane];
    if (live) {
      *reinterpret_cast<double2*>(nodeF + (size_t)m * 8 + 6) = make_double2(ntv, (double)nhab);
      nodeC[m] = make_double4(S, __longlong_as_double(((long long)elems << 32) | (long long)(uint32_t)hits),
                              __longlong_as_double((long long)vis), 0.0);
    }
    // later passes (parents) and the re-summation below read these back: make the stores visible to the wave first
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    wave_sync();
    // ---------------------------------------------------------------- 3. ranking of the pass's qualifying leaves
    bool q = false;
    double lo = __builtin_inf(), hi = __builtin_inf();
    if (live && m >= 1) {  // node 0 is the start state: never a leaf candidate (:144-171)
      q = ctt >= thresh;
      if (q) {
        double c0, c1, c2;
        const double tot = total_of(hits, vis, ctt, S, c0, c1, c2);
        // |S - ordered sum| <= 2 gamma_L L term_max; one more rounding each for the division and the two additions
        const double L = (double)elems;
        const double gam = 2.0 * (L + 2.0) * 0x1p-53;
        // sum|term| <= L term_max; with probabilities of one sign also sum|term| = |exact sum| <= |S| / (1 - gamma_L).
        // The second bound is what separates exact ties: a path without any shark term has S = 0 = the reference's sum.
        double mag = L * term_max;
        if (W.prob_one_sign) { const double ms = auvp_fabs(S) * (1.0 + 0x1p-20); mag = ms < mag ? ms : mag; }
        double e2 = 2.0 * gam * mag;
        if (ctt > 0) e2 = e2 / ctt;
        // e2 == 0: the sums are the same number, and so is everything computed from them
        const double err = e2 == 0.0 ? 0.0 : 1.25 * e2 + 0x1p-50 * (auvp_fabs(c0) + auvp_fabs(c1) + auvp_fabs(c2) + e2);
        lo = tot - err; hi = tot + err;
        if (!(err == err) || !(tot == tot)) { lo = -__builtin_inf(); hi = __builtin_inf(); }  // nan: decide exactly
      }
    }
    const unsigned long long qm = wave_ballot(q);
    if (qm != 0ull) {
    n_leaves += __popcll(qm);
    // exclusive prefix minimum of hi over the lanes (creation order), seeded with the earlier passes
    double pm = hi;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double t = __shfl_up(pm, o, 64);
      if (lane >= o) pm = t < pm ? t : pm;
    }
    double before = __shfl_up(pm, 1, 64);
    if (lane == 0) before = __builtin_inf();
    before = before < min_hi ? before : min_hi;
    leaf_elems += q ? (long long)elems : 0ll;  // (per lane; summed over the wavefront once, where the record is written)
    const bool cand = q && (log_leaf || lo < before);
    unsigned long long cm = wave_ballot(cand);
    min_hi = readlane_f64(pm, 63) < min_hi ? readlane_f64(pm, 63) : min_hi;
    while (cm) {
      const int l = __ffsll((long long)cm) - 1;
      cm &= cm - 1ull;
      const int leaf = __builtin_amdgcn_readlane(m, l);
      const double lo_l = readlane_f64(lo, l);
      if (!log_leaf && !(lo_l < best_tot)) continue;  // an exact total found meanwhile already rules it out
      // ---- the reference's ordered sum: [leaf] + reversed(leaf.path[1:]) + [parent] + reversed(parent.path[1:]) ... root
      // In chunks of up to 64 chain nodes: (A) the parent links are walked on their own -- a chain of dependent reads,
      // nothing else waits on it -- and leave one descriptor per node in LDS (arrays of the pass that are dead by now);
      // (B) the chunk's elements -- a node's own term, then its points last to first -- are evaluated lane = element, every
      // lane busy, into an LDS buffer in the reference's order, RS_CAP at a time; (C) the buffer is summed left to right, one
      // rounded add per element.  The same terms in the same order as a walk that evaluates node after node.  (Built and
      // dropped: skip links -- every node record carrying its depth and its nearest ancestor at a depth that is a multiple of
      // 8, so that a chunk costs 8 + 8 dependent reads instead of 64 -- bit-identical and no faster: profiles/r5_leaf_pass.md.)
      constexpr int RS_CAP = 256;  // elements per window: the owner table's 2 048 bytes as doubles
      double* rs_buf = reinterpret_cast<double*>(c_owner);
      int32_t *d_off = c_off, *d_w = c_cpos, *d_pos = c_par;
      double* d_tv = term;
      double c2num = 0.0;
      int mm = leaf;
      while (mm >= 0) {
        // (A) descriptors
        int nh = 0, cnt = 0;
        while (nh < 64 && mm >= 0) {
          const int4 rr = nodeI[mm];
          const double tvn = nodeF[(size_t)mm * 8 + 6];
          const int par_m = uni(rr.y);
          const int w = par_m >= 0 ? uni(rr.w) : 0;  // the root has no path of its own
          if (lane == 0) { d_off[nh] = rr.z; d_w[nh] = w; d_pos[nh] = cnt; d_tv[nh] = tvn; }
          cnt += 1 + w; nh++;
          mm = par_m;
        }
        wave_sync();
        for (int e0 = 0; e0 < cnt; e0 += RS_CAP) {
          const int ne = (cnt - e0) < RS_CAP ? (cnt - e0) : RS_CAP;
          // (B) the window's elements, 64 per round
          for (int s0 = 0; s0 < ne; s0 += 64) {
            const int sl = e0 + s0 + lane;
            if (s0 + lane < ne) {
              int h = 0;
#pragma unroll
              for (int st = 32; st >= 1; st >>= 1)
                if (h + st < nh && d_pos[h + st] <= sl) h += st;
              const int k = sl - d_pos[h];
              double tv_e = 0.0;
              if (k == 0) tv_e = d_tv[h];  // the node's own state comes before the points that led to it
              else {                        // point w - k: last point first; the term is evaluated again from the record
                const double* rec = ptF + ((size_t)d_off[h] + (size_t)(d_w[h] - k)) * 3;
                const double2 xy = *reinterpret_cast<const double2*>(rec);
                int habp = -1;
                cost_element(W, St, 0, W.n_bins, P.w[2], xy.x, xy.y, rec[2], tv_e, habp, true, grid_lds);
              }
              rs_buf[s0 + lane] = tv_e;
            }
          }
          wave_sync();
          // (C) one rounded add per element, in order
          int i = 0;
          for (; i + 4 <= ne; i += 4) {
            const double2 a = *reinterpret_cast<const double2*>(rs_buf + i), b = *reinterpret_cast<const double2*>(rs_buf + i + 2);
            c2num = c2num + a.x; c2num = c2num + a.y; c2num = c2num + b.x; c2num = c2num + b.y;
          }
          for (; i < ne; i++) c2num = c2num + rs_buf[i];
          wave_sync();
        }
      }
      const int lhits = __builtin_amdgcn_readlane(hits, l), lelems = __builtin_amdgcn_readlane(elems, l);
      st_resummed = uni(st_resummed + lelems); st_releaves = uni(st_releaves + 1);
      const unsigned long long lvis = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(vis >> 32), l) << 32) |
                                      (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(vis & 0xffffffffull), l);
      const double lctt = readlane_f64(ctt, l), llen = readlane_f64(nlen, l);
      double c0, c1, c2;
      double tot = total_of(lhits, lvis, lctt, c2num, c0, c1, c2);
      tot = readfirst_f64(tot);
      if (log_leaf) {
        // position of this leaf among the qualifying ones = leaves before this pass + qualifying lanes below l
        const int pos = n_leaves - __popcll(qm) + __popcll(qm & ((1ull << l) - 1ull));
        if (pos < B.cap_leaves && lane == 0) {
          // number of shark-grid bins in the leaf's sub-dict (:160-165)
          int nsel = 0;
          for (int b = 0; b < W.n_bins; b++) {
            const double b0 = s_bins[b][0], b1 = s_bins[b][1];
            nsel += ((init_t >= b0 && init_t <= b1) || (b0 >= init_t && b1 <= lctt) || (lctt >= b0 && lctt <= b1)) ? 1 : 0;
          }
          double* lc = B.leaf_cost + ((size_t)ep * B.cap_leaves + pos) * 6;
          lc[0] = tot; lc[1] = c0; lc[2] = c1; lc[3] = c2; lc[4] = (double)lelems; lc[5] = (double)nsel;
          B.leaf_iter[(size_t)ep * B.cap_leaves + pos] = nodeI[leaf].x;
        }
      }
      if (tot < best_tot) {
        best_tot = tot; best_leaf = leaf; best_L = lelems;
        best_c0 = c0; best_c1 = c1; best_c2 = c2; best_len = llen;
      }
    }
    }  // qm
    // the ids of the pass are done: the rest of the queue moves to its front
    wave_sync();
    const int carry = (lane + 64 < qn) ? c_ids[lane + 64] : 0;
    wave_sync();
    if (lane + 64 < qn) c_ids[lane] = carry;
    qn = qn > 64 ? qn - 64 : 0;
    wave_sync();
  }
  {
    // (leaf_elems was kept per lane)
    long long el = leaf_elems;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) el += __shfl_xor(el, o, 64);
    leaf_elems = el;
  }
  if (lane == 0) {
    if (B.leaf_stats) {
      atomicAdd(&B.leaf_stats[0], (unsigned long long)st_nodes); atomicAdd(&B.leaf_stats[1], (unsigned long long)st_points);
      atomicAdd(&B.leaf_stats[2], (unsigned long long)st_resummed); atomicAdd(&B.leaf_stats[3], (unsigned long long)st_releaves);
      // [4]: the most 32-bit outputs one episode of the batch drew -- what the host sizes the next batch's pre-generated
      // random stream from (auvplan.hip: option ROWS_STREAM)
      atomicMax(&B.leaf_stats[4], (unsigned long long)sum.n_draw32);
    }
    sum.n_leaves = n_leaves;
    sum.leaf_elems = leaf_elems;
    sum.best_leaf = best_leaf;
    sum.best_path_len = best_L;
    if (best_leaf >= 0) {
      sum.best_cost[0] = best_tot; sum.best_cost[1] = best_c0; sum.best_cost[2] = best_c1; sum.best_cost[3] = best_c2;
      sum.best_length = best_len;
    } else if (status_in == 0) {
      sum.status = 1;  // no qualifying leaf: opt_path stays None (:174)
    }
  }
}

// generate_final_course (:321-331) of the best leaf, written root -> leaf (exploring reverses it, :174)
static __global__ __launch_bounds__(64) void rrt_final_course_kernel(RrtBuffers B, const int64_t* __restrict__ offsets,
                                                              double* __restrict__ out, int n_episodes) {
  const int ep = blockIdx.x;
  if (ep >= n_episodes) return;
  const int lane = lane_id();
  const RrtSummary s = B.summary[ep];
  if (s.best_leaf < 0) return;
  const int capn = B.cap_nodes;
  const size_t capp = (size_t)B.cap_points;
  const double* nodeF = B.node_f + (size_t)ep * capn * 8;
  const int4* nodeI = reinterpret_cast<const int4*>(B.node_i) + (size_t)ep * capn;
  const double* ptF = B.points + (size_t)ep * capp * 6;
  const double* init = B.init + (size_t)ep * 6;
  double* o = out + 7 * (size_t)offsets[ep];
  int pos = s.best_path_len - 1;  // element index of the leaf
  auto node_elem = [&](int m, int at) {
    double* e = o + 7 * (size_t)at;
    const int4 r = nodeI[m];
    if (r.y < 0) {
      e[0] = init[0]; e[1] = init[1]; e[2] = init[2]; e[3] = 0.0; e[4] = init[3]; e[5] = init[4]; e[6] = init[5];
    } else {
      const double* nf = nodeF + (size_t)m * 8;
      e[0] = nf[0]; e[1] = nf[1]; e[2] = nf[2]; e[3] = 0.0; e[4] = nf[3]; e[5] = (double)r.x; e[6] = nf[4];
    }
  };
  if (lane == 0) node_elem(s.best_leaf, pos);
  pos--;
  for (int m = s.best_leaf;;) {
    const int4 r = nodeI[m];
    if (r.y < 0) break;
    const int cnt = r.w, off = r.z;
    for (int k = lane; k < cnt; k += 64) {
      // point k of the node sits k places after the node it grew from
      double* e = o + 7 * (size_t)(pos - cnt + 1 + k);
      size_t gi = (size_t)off + k;
      const double* ra = ptF + gi * 3;
      const double* rb = ptF + capp * 3 + gi * 3;
      e[0] = ra[0]; e[1] = ra[1]; e[2] = rb[0]; e[3] = rb[1];
      e[4] = ra[2]; e[5] = (double)r.x; e[6] = rb[2];
    }
    pos -= cnt;
    if (lane == 0) node_elem(r.y, pos);
    pos--;
    m = r.y;
  }
}

}  // namespace auvp
#endif
