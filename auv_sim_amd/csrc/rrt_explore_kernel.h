// rrt_explore_kernel.h -- RRT.exploring (path_planning/rrt_dubins.py:92-176) on gfx950.
//
// One wavefront = one episode, persistent over the whole iteration budget.  A 256-thread workgroup
// carries 4 independent episodes that share the LDS copy of the small world tables (habitats,
// polygon, time bins); obstacles live in registers (J per lane), the tree lives in HBM as SoA.
//
// Inside one expansion the 64 lanes split the work:
//   steer        lane s = sub-arc s: RNG window tempering, arc geometry and sin/cos in parallel;
//                only the running sums (theta, x, y, t, length) are serial, and they must be, to
//                keep the reference's left-to-right fp64 addition order
//   collision    lane = obstacle (J each), loop over the path points broadcast from LDS
//   polygon      lane = path point
//   cost         lane = path element of one ancestor segment; ordered accumulation of the shark term
// The steer draw count is data dependent (a sub-arc consumes 2 or 3 random() values), so the lanes
// temper a window of the stream, ballot the "taken" predicate for every possible start offset, and
// a short scalar loop resolves where each sub-arc starts.
#ifndef AUVP_RRT_EXPLORE_KERNEL_H
#define AUVP_RRT_EXPLORE_KERNEL_H
#include "auvp_math.h"
#include "auvp_types.h"
#include "auvp_wave.h"

namespace auvp {

constexpr int RRT_WAVES = 4;         // episodes per workgroup
constexpr int RRT_MAX_CHUNK = 63;    // sub-arcs per steer pass; lane 63 stays idle (chunk-entry theta)
constexpr int RRT_MAX_HAB = 64;      // visited-habitat mask is one 64-bit word
constexpr int RRT_MAX_POLY = 64;
constexpr int RRT_MAX_BINS = 64;

// Per-wave LDS block (doubles first for alignment)
struct RrtWaveLds {
  double u[3 * RRT_MAX_CHUNK + 3];  // random() window
  double inc[64][4];                // dx,dy,dt,movement -> running x,y,t,length
  double sc[64][2];                 // sin,cos per sub-arc end angle
  double phi[64];
  uint32_t mt[624];
};

struct RrtSharedLds {
  double hab[RRT_MAX_HAB][3];
  double poly[RRT_MAX_POLY][2];
  double bins[RRT_MAX_BINS][2];
};

__host__ __device__ inline size_t rrt_lds_bytes(int K, int max_pts) {
  size_t b = sizeof(RrtSharedLds) + RRT_WAVES * sizeof(RrtWaveLds);
  b += (size_t)RRT_WAVES * (size_t)(max_pts) * 2 * sizeof(double);  // path points x,y
  b += (size_t)RRT_WAVES * (size_t)(K + 2) * sizeof(int32_t);        // bin counts
  return (b + 15) & ~(size_t)15;
}

// Point(x,y).within(polygon): even-odd crossing number with strict comparisons -- the definition
// pinned in tests/golden/_refstubs/install.py (shapely itself is absent).
__device__ __forceinline__ bool point_within(const double (*poly)[2], int nv, double x, double y) {
  bool inside = false;
  int j = nv - 1;
  for (int i = 0; i < nv; i++) {
    double xi = poly[i][0], yi = poly[i][1], xj = poly[j][0], yj = poly[j][1];
    if ((yi > y) != (yj > y)) {
      if (x < (xj - xi) * (y - yi) / (yj - yi) + xi) inside = !inside;
    }
    j = i;
  }
  return inside;
}

// cost.py:181-184 first-match cell scan through the x-bucket index; returns cell id or -1
__device__ __forceinline__ int cell_lookup(const WorldDev& W, double x, double y) {
  if (W.n_cells == 0) return -1;
  double fb = auvp_floor((x - W.xb_x0) * W.xb_inv_w);
  int b = fb < 0.0 ? 0 : (fb >= (double)W.n_xbuckets ? W.n_xbuckets - 1 : (int)fb);
  int e = W.xb_off[b + 1];
  for (int k = W.xb_off[b]; k < e; k++) {
    const double4 d = reinterpret_cast<const double4*>(W.xb_data)[k];
    if (y < d.w) return -1;  // every remaining candidate has miny > y
    // sic: x is compared with maxy as well as maxx (path_planning/cost.py:182): d.y = min(maxx, maxy)
    if (x >= d.x && x <= d.y && y >= d.z) return W.xb_items[k];
  }
  return -1;
}

struct CostAcc {
  double c2;
  unsigned long long visited;
  int hits;
};

// One segment of path elements (one per lane, `valid` lanes) in path order = lane order.
// habitat_shark_cost_func body, path_planning/cost.py:171-193.
__device__ __forceinline__ void cost_segment(const WorldDev& W, const RrtSharedLds& S, int bin_lo, int bin_hi,
                                             double w3, bool valid, double x, double y, double t, CostAcc& acc) {
  int tb = -1;
  if (valid) {
    for (int b = bin_lo; b < bin_hi; b++) {
      if (t >= S.bins[b][0] && t <= S.bins[b][1]) { tb = b; break; }
    }
  }
  bool inbin = valid && tb >= 0;
  double term = 0.0;
  bool has_term = false;
  int hab = -1;
  if (inbin) {
    int c = cell_lookup(W, x, y);
    if (c >= 0) { term = w3 * W.prob[(size_t)tb * W.n_cells + c]; has_term = true; }
    for (int h = 0; h < W.n_habitats; h++) {
      double ddx = S.hab[h][0] - x, ddy = S.hab[h][1] - y;
      double d = auvp_sqrt(ddx * ddx + ddy * ddy);
      if (d <= S.hab[h][2]) { hab = h; break; }
    }
  }
  // ordered accumulation: cost[2] += w3*prob in path order
  unsigned long long m = __ballot(has_term);
  double c2 = acc.c2;
  while (m) {
    int l = __ffsll((long long)m) - 1;
    m &= m - 1;
    c2 = c2 + readlane_f64(term, l);
  }
  acc.c2 = c2;
  unsigned long long hm = __ballot(hab >= 0);
  acc.hits += __popcll(hm);
  unsigned long long vis = acc.visited;
  for (int h = 0; h < W.n_habitats; h++) {
    if (__ballot(hab == h)) vis |= (1ull << h);
  }
  acc.visited = vis;
}

template <int J>
__global__ __launch_bounds__(RRT_WAVES * 64) void rrt_explore_kernel(WorldDev W, RrtParamsDev P, RrtBuffers B,
                                                                     int n_episodes, int max_pts) {
  extern __shared__ __align__(16) unsigned char smem[];
  RrtSharedLds& S = *reinterpret_cast<RrtSharedLds*>(smem);
  const int wave = (int)(threadIdx.x >> 6);
  const int lane = lane_id();
  RrtWaveLds& L = reinterpret_cast<RrtWaveLds*>(smem + sizeof(RrtSharedLds))[wave];
  unsigned char* tail = smem + sizeof(RrtSharedLds) + RRT_WAVES * sizeof(RrtWaveLds);
  double(*pts)[2] = reinterpret_cast<double(*)[2]>(tail) + (size_t)wave * max_pts;
  int32_t* bin_count =
      reinterpret_cast<int32_t*>(tail + (size_t)RRT_WAVES * max_pts * 2 * sizeof(double)) + (size_t)wave * (P.K + 2);

  // ---- stage the shared world tables (whole workgroup) ----
  for (int i = threadIdx.x; i < W.n_habitats * 3; i += blockDim.x) (&S.hab[0][0])[i] = W.hab[i];
  for (int i = threadIdx.x; i < W.n_poly * 2; i += blockDim.x) (&S.poly[0][0])[i] = W.poly[i];
  for (int i = threadIdx.x; i < W.n_bins * 2; i += blockDim.x) (&S.bins[0][0])[i] = W.bins[i];
  __syncthreads();

  const int ep = (int)blockIdx.x * RRT_WAVES + wave;
  if (ep >= n_episodes) return;  // no workgroup barrier after this point

  // ---- obstacles into registers: obstacle i = j*64 + lane ----
  double ox[J], oy[J], ot[J];
#pragma unroll
  for (int j = 0; j < J; j++) {
    int i = j * 64 + lane;
    bool ok = i < W.n_obstacles;
    ox[j] = ok ? W.ox[i] : 0.0;
    oy[j] = ok ? W.oy[i] : 0.0;
    ot[j] = ok ? W.ot[i] : -1.0;  // d2 >= 0 > -1: padding never collides
  }

  // ---- per-episode views ----
  const size_t nb = (size_t)ep * B.cap_nodes, pb = (size_t)ep * B.cap_points;
  double *nx = B.nx + nb, *ny = B.ny + nb, *nth = B.nth + nb, *ntt = B.ntt + nb, *nlen = B.nlen + nb;
  int32_t *nplan = B.nplan + nb, *parent = B.parent + nb, *pt_off = B.pt_off + nb, *pt_cnt = B.pt_cnt + nb;
  double *px = B.px + pb, *py = B.py + pb, *pth = B.pth + pb, *pv = B.pv + pb, *ptt = B.ptt + pb, *plen = B.plen + pb;
  int32_t* bin_items = B.bin_items + (size_t)ep * (P.K + 1) * B.bin_cap;
  const double* init = B.init + (size_t)ep * 6;
  const int K = P.K;
  const bool log_it = (P.flags & 1) != 0, log_leaf = (P.flags & 2) != 0;

  WaveRng rng;
  rng.s = L.mt;
  for (int i = lane; i < 624; i += 64) L.mt[i] = B.mt[(size_t)ep * 624 + i];
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

  // mps_list = [initial]; time_bin[bin_interval].append(initial)  (:105,:114)
  if (lane == 0) {
    nx[0] = init[0]; ny[0] = init[1]; nth[0] = init[2]; ntt[0] = init[3]; nlen[0] = init[5];
    nplan[0] = 0; parent[0] = -1; pt_off[0] = 0; pt_cnt[0] = 0;
    if (P.mode == 0) { bin_items[(size_t)(K >= 1 ? 1 : 0) * B.bin_cap] = 0; bin_count[K >= 1 ? 1 : 0] = 1; }
  }
  wave_sync();
  const double init_t = init[3];
  int n_nodes = 1, n_points = 0, n_leaves = 0, status = 0, best_leaf = -1, best_L = 0;
  double best[4] = {__builtin_inf(), 0.0, 0.0, 0.0};
  double best_len = 0.0;
  long long leaf_elems = 0;
  int it = 0;
  // optional per-phase shader-clock accounting (AUVP_FLAG_PHASE_CLOCKS): select, steer, collision,
  // accept, cost walk
  const bool clk = (P.flags & 4) != 0;
  unsigned long long tph[5] = {0, 0, 0, 0, 0}, t_prev = 0;
#define AUVP_PHASE(i) do { if (clk) { unsigned long long t_now = __builtin_amdgcn_s_memtime(); tph[i] += t_now - t_prev; t_prev = t_now; } } while (0)

  for (; it < P.max_iter; it++) {
    if (log_it && lane == 0) {
      B.it_parent[(size_t)ep * P.max_iter + it] = -1;
      B.it_accepted[(size_t)ep * P.max_iter + it] = 0;
      B.it_npath[(size_t)ep * P.max_iter + it] = 0;
    }
    // ------------------------------------------------------------ parent selection (:121-139)
    if (clk) t_prev = __builtin_amdgcn_s_memtime();
    int par;
    if (P.mode == 0) {
      int rb, cnt;
      for (;;) {
        double u = rng_next_random(rng);
        rb = uni((int)py_uniform(1.0, (double)(K + 1), u));
        if (rb > K) { status = -5; break; }
        cnt = uni(bin_count[rb]);
        if (cnt != 0) break;
      }
      if (status) break;
      double u = rng_next_random(rng);
      int ri = uni((int)py_uniform(0.0, (double)cnt, u));
      par = uni(bin_items[(size_t)rb * B.bin_cap + ri]);
    } else if (P.mode == 1) {
      double u = rng_next_random(rng);
      double ran_time = py_uniform(0.0, P.max_plan_time * P.freq, u);
      int lo = 0, hi = n_nodes;  // list slicing of get_closest_mps_time (:515-528)
      while (hi - lo > 3) {
        int n = hi - lo;
        double ld = auvp_fabs((double)nplan[lo + n / 2 - 1] - ran_time);
        double rd = auvp_fabs((double)nplan[lo + n / 2 + 1] - ran_time);
        if (ld >= rd) lo += n / 2; else hi = lo + n / 2;
        lo = uni(lo); hi = uni(hi);
      }
      par = lo;
      if (ntt[par] > P.max_traj_time) continue;
    } else {
      // get_random_mps (:333-343): x, y, theta, size draws; only x,y are used
      rng_ensure(rng, 8);
      double rx = py_uniform(W.bb[0], W.bb[2], rng_random_at(rng, 0));
      double ry = py_uniform(W.bb[1], W.bb[3], rng_random_at(rng, 1));
      rng_advance_words(rng, 8);
      // get_closest_mps (:505-513): first index with the smallest RN(sqrt(d2))
      double bd = __builtin_inf();
      int bi = 0x7fffffff;
      for (int m = lane; m < n_nodes; m += 64) {
        double ddx = rx - nx[m], ddy = ry - ny[m];
        double d = auvp_sqrt(ddx * ddx + ddy * ddy);
        if (d < bd) { bd = d; bi = m; }
      }
      double gmin = wave_min_f64(bd);
      int cand = (bd == gmin) ? bi : 0x7fffffff;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        int t = __shfl_xor(cand, o, 64);
        cand = t < cand ? t : cand;
      }
      par = uni(cand);
      if (ntt[par] > P.max_traj_time) continue;
    }

    AUVP_PHASE(0);
    // ------------------------------------------------------------ steer (:252-295)
    const double par_x = nx[par], par_y = ny[par];
    double cx = par_x, cy = par_y, cth = nth[par], ctt = ntt[par], clen = nlen[par];
    int n_total;
    {
      double u = rng_next_random(rng);
      n_total = uni((int)auvp_floor(py_uniform(0.0, P.freq, u) / 1));
    }
    int cnt = 0;  // appended path points
    if (lane == 0) { pts[0][0] = par_x; pts[0][1] = par_y; }
    bool cap_err = false;
    for (int c0 = 0; c0 < n_total; c0 += RRT_MAX_CHUNK) {
      const int n = (n_total - c0) < RRT_MAX_CHUNK ? (n_total - c0) : RRT_MAX_CHUNK;
      const int nwin = 3 * n;
      rng_ensure(rng, (uint32_t)(2 * nwin));
      for (int jj = lane; jj < nwin; jj += 64) L.u[jj] = rng_random_at(rng, (uint32_t)jj);
      wave_sync();
      // "taken" predicate for every possible start offset
      unsigned long long msk[3];
#pragma unroll
      for (int t = 0; t < 3; t++) {
        int jj = lane + 64 * t;
        bool f = false;
        if (jj + 1 < nwin) {
          double dist = py_uniform(0.0, P.dist_to_end, L.u[jj]);
          double diff = py_uniform(-P.diff_max, P.diff_max, L.u[jj + 1]);
          f = auvp_fabs(dist) > auvp_fabs(diff);
        }
        msk[t] = __ballot(f);
      }
      // where does sub-arc s start?  pos += 2 + taken(pos)
      int pos = 0, mypos = 0;
      for (int s = 0; s < n; s++) {
        if (lane == s) mypos = pos;
        unsigned long long mm = pos < 64 ? msk[0] : (pos < 128 ? msk[1] : msk[2]);
        int bit = (int)((mm >> (pos & 63)) & 1ull);
        pos = uni(pos + 2 + bit);
      }
      const int used = pos;
      const bool active = lane < n;
      double radius = 0.0, phi = 0.0, vt = 1.0;
      bool taken = false;
      if (active) {
        double dist = py_uniform(0.0, P.dist_to_end, L.u[mypos]);
        double diff = py_uniform(-P.diff_max, P.diff_max, L.u[mypos + 1]);
        taken = auvp_fabs(dist) > auvp_fabs(diff);
        if (taken) {
          double s1 = dist + diff, s2 = dist - diff;
          radius = (s1 + s2) / (-s1 + s2);
          phi = (s1 + s2) / (2 * radius);
          vt = py_uniform(0.0, 2 * P.v, L.u[mypos + 2]);
        }
      }
      const unsigned long long tmask = __ballot(taken);
      L.phi[lane] = phi;
      wave_sync();
      // theta += phi, left to right
      double th = cth, myth = cth;
      for (int s = 0; s < n; s++) {
        if ((tmask >> s) & 1ull) th = th + L.phi[s];
        if (lane == s) myth = th;
      }
      double sn, cs;
      auvp_sincos(myth, &sn, &cs);  // idle lanes evaluate the chunk-entry angle
      L.sc[lane][0] = sn;
      L.sc[lane][1] = cs;
      wave_sync();
      double dx = 0.0, dy = 0.0, mv = 0.0, dt = 0.0;
      if (taken) {
        unsigned long long below = tmask & ((1ull << lane) - 1ull);
        int prev = below ? (63 - __clzll((long long)below)) : 63;  // lane 63 holds the entry angle
        double so = L.sc[prev][0], co = L.sc[prev][1];
        dx = radius * (sn - so);
        dy = radius * (-cs + co);
        mv = auvp_sqrt(dx * dx + dy * dy);
        dt = mv / vt;
      }
      L.inc[lane][0] = dx; L.inc[lane][1] = dy; L.inc[lane][2] = dt; L.inc[lane][3] = mv;
      wave_sync();
      // x += dx; y += dy; t += dt; length += movement: four serial chains, one lane each
      if (lane < 4) {
        double acc = lane == 0 ? cx : (lane == 1 ? cy : (lane == 2 ? ctt : clen));
        for (int s = 0; s < n; s++) {
          if ((tmask >> s) & 1ull) acc = acc + L.inc[s][lane];
          L.inc[s][lane] = acc;
        }
      }
      wave_sync();
      const double mx = L.inc[lane][0], my = L.inc[lane][1], mt_ = L.inc[lane][2], ml = L.inc[lane][3];
      const bool app = taken && (mv >= P.min_dist);
      const unsigned long long amask = __ballot(app);
      const int napp = __popcll(amask);
      if (n_points + cnt + napp > B.cap_points || cnt + napp + 1 > max_pts) { cap_err = true; break; }
      if (app) {
        int rank = __popcll(amask & ((1ull << lane) - 1ull));
        int gi = n_points + cnt + rank;  // speculative: committed only if the node is accepted
        px[gi] = mx; py[gi] = my; pth[gi] = myth; pv[gi] = vt; ptt[gi] = mt_; plen[gi] = ml;
        pts[cnt + rank + 1][0] = mx;
        pts[cnt + rank + 1][1] = my;
      }
      cnt += napp;
      if (n > 0) {
        cx = readlane_f64(mx, n - 1); cy = readlane_f64(my, n - 1);
        ctt = readlane_f64(mt_, n - 1); clen = readlane_f64(ml, n - 1);
        cth = th;
      }
      rng_advance_words(rng, (uint32_t)(2 * used));
      wave_sync();
    }
    if (cap_err) { status = -2; break; }
    wave_sync();
    const int P_n = cnt + 1;

    AUVP_PHASE(1);
    // ------------------------------------------------------------ check_collision (:530-549)
    bool hit = false;
    for (int p = 0; p < P_n; p++) {
      const double qx = pts[p][0], qy = pts[p][1];
#pragma unroll
      for (int j = 0; j < J; j++) {
        double ddx = qx - ox[j], ddy = qy - oy[j];
        double d2 = ddx * ddx + ddy * ddy;
        hit = hit || (d2 <= ot[j]);
      }
    }
    bool outside = false;
    for (int p = lane; p < P_n; p += 64) outside = outside || !point_within(S.poly, W.n_poly, pts[p][0], pts[p][1]);
    const bool ok = !__any(hit) && !__any(outside);
    if (log_it && lane == 0) {
      B.it_parent[(size_t)ep * P.max_iter + it] = par;
      B.it_accepted[(size_t)ep * P.max_iter + it] = ok ? 1 : 0;
      B.it_npath[(size_t)ep * P.max_iter + it] = P_n;
    }
    AUVP_PHASE(2);
    if (!ok) continue;
    if (n_nodes >= B.cap_nodes) { status = -2; break; }

    // ------------------------------------------------------------ accept (:144-151)
    const int me = n_nodes;
    if (lane == 0) {
      nx[me] = cx; ny[me] = cy; nth[me] = cth; ntt[me] = ctt; nlen[me] = clen;
      nplan[me] = it; parent[me] = par; pt_off[me] = n_points; pt_cnt[me] = cnt;
    }
    n_nodes++;
    n_points += cnt;
    if (P.mode == 0) {
      // curr_bin = (t // bin_interval + 1) * bin_interval, exact floor of the true quotient
      double q = auvp_floor(ctt / P.bin_interval);
      double r = auvp_fma(-q, P.bin_interval, ctt);
      if (r < 0.0) q -= 1.0;
      else if (r >= P.bin_interval) q += 1.0;
      double fi = q + 1.0;
      double curr_bin = fi * P.bin_interval;
      bool over = curr_bin > P.max_traj_time;
      if (!over || fi <= (double)K) {
        int bi = (int)fi;
        int c = over ? 0 : bin_count[bi];  // an overflowing regular key is reset first (:149-151)
        if (c >= B.bin_cap) { status = -2; break; }
        if (lane == 0) { bin_items[(size_t)bi * B.bin_cap + c] = me; bin_count[bi] = c + 1; }
      }
      wave_sync();
    }

    AUVP_PHASE(3);
    // ------------------------------------------------------------ qualifying leaf (:158-171)
    if (ctt >= P.max_traj_time - 30) {
      __threadfence_block();  // this wave's own stores above are re-read below through L1
      int blo = -1, bhi = -1, nsel = 0;
      bool contiguous = true;
      for (int b = 0; b < W.n_bins; b++) {
        double b0 = S.bins[b][0], b1 = S.bins[b][1];
        if ((init_t >= b0 && init_t <= b1) || (b0 >= init_t && b1 <= ctt) || (ctt >= b0 && ctt <= b1)) {
          if (blo < 0) blo = b;
          else if (b != bhi) contiguous = false;
          bhi = b + 1;
          nsel++;
        }
      }
      if (nsel == 0) blo = bhi = 0;
      if (!contiguous) { status = -1; break; }
      CostAcc acc;
      acc.c2 = 0.0; acc.visited = 0ull; acc.hits = 0;
      int Lp = 1;
      // path = [leaf] + reversed(leaf.path) + reversed(parent.path) + ...   (:321-331)
      cost_segment(W, S, blo, bhi, P.w[2], lane == 0, cx, cy, ctt, acc);
      int m = me, mcnt = cnt, moff = n_points - cnt, mpar = par;
      for (;;) {
        // segment of node m: its appended points last-to-first, then the node it grew from
        const int seg = mcnt + 1;
        for (int s0 = 0; s0 < seg; s0 += 64) {
          int i = s0 + lane;
          bool valid = i < seg;
          double ex = 0.0, ey = 0.0, et = 0.0;
          if (valid) {
            if (i < mcnt) { int gi = moff + (mcnt - 1 - i); ex = px[gi]; ey = py[gi]; et = ptt[gi]; }
            else { ex = nx[mpar]; ey = ny[mpar]; et = ntt[mpar]; }
          }
          cost_segment(W, S, blo, bhi, P.w[2], valid, ex, ey, et, acc);
        }
        Lp += seg;
        m = mpar;
        int gp = uni(parent[m]);
        if (gp < 0) break;
        mcnt = uni(pt_cnt[m]); moff = uni(pt_off[m]); mpar = gp;
      }
      double c0 = 0.0, c1 = 0.0, c2 = acc.c2;
      const double w2 = P.w[1];
      if (w2 == auvp_rint(w2) && auvp_fabs(w2) < 1048576.0) c1 = w2 * (double)acc.hits;  // exact
      else for (int h = 0; h < acc.hits; h++) c1 = c1 + w2;
      if (ctt > 0) { c1 = c1 / ctt; c2 = c2 / ctt; }
      if (W.n_habitats != 0) c0 = P.w[0] * (double)__popcll(acc.visited) / (double)W.n_habitats;
      double tot = ((0.0 + c0) + c1) + c2;
      if (log_leaf && n_leaves < B.cap_leaves && lane == 0) {
        double* lc = B.leaf_cost + ((size_t)ep * B.cap_leaves + n_leaves) * 6;
        lc[0] = tot; lc[1] = c0; lc[2] = c1; lc[3] = c2; lc[4] = (double)Lp; lc[5] = (double)nsel;
        B.leaf_iter[(size_t)ep * B.cap_leaves + n_leaves] = it;
      }
      n_leaves++;
      leaf_elems += Lp;
      AUVP_PHASE(4);
      if (tot < best[0]) {
        best[0] = tot; best[1] = c0; best[2] = c1; best[3] = c2;
        best_leaf = me; best_L = Lp; best_len = clen;
      }
    }
  }

  if (clk && lane == 0 && B.phase_clocks) {
    for (int i = 0; i < 5; i++) B.phase_clocks[(size_t)ep * 5 + i] = tph[i];
  }
  const unsigned long long drawn = rng.drawn;
  double after = rng_next_random(rng);
  for (int i = lane; i < K + 1; i += 64) B.bin_count[(size_t)ep * (K + 1) + i] = bin_count[i];
  if (lane == 0) {
    RrtSummary& s = B.summary[ep];
    if (status == 0 && best_leaf < 0) status = 1;
    s.status = status; s.n_nodes = n_nodes; s.n_points = n_points; s.n_leaves = n_leaves;
    s.best_leaf = best_leaf; s.best_path_len = best_L; s.iters_run = it; s._pad = 0;
    s.best_cost[0] = best[0]; s.best_cost[1] = best[1]; s.best_cost[2] = best[2]; s.best_cost[3] = best[3];
    s.best_length = best_len; s.rng_after = after; s.leaf_elems = leaf_elems; s.n_draw32 = drawn;
  }
}

// generate_final_course (:321-331) of the best leaf, written root -> leaf (exploring reverses it, :174)
__global__ __launch_bounds__(64) void rrt_final_course_kernel(RrtBuffers B, const int64_t* __restrict__ offsets,
                                                              double* __restrict__ out, int n_episodes) {
  const int ep = blockIdx.x;
  if (ep >= n_episodes) return;
  const int lane = lane_id();
  const RrtSummary s = B.summary[ep];
  if (s.best_leaf < 0) return;
  const size_t nb = (size_t)ep * B.cap_nodes, pb = (size_t)ep * B.cap_points;
  const double* init = B.init + (size_t)ep * 6;
  double* o = out + 7 * (size_t)offsets[ep];
  int pos = s.best_path_len - 1;  // element index of the leaf
  auto node_elem = [&](int m, int at) {
    double* e = o + 7 * (size_t)at;
    if (B.parent[nb + m] < 0) {
      e[0] = init[0]; e[1] = init[1]; e[2] = init[2]; e[3] = 0.0; e[4] = init[3]; e[5] = init[4]; e[6] = init[5];
    } else {
      e[0] = B.nx[nb + m]; e[1] = B.ny[nb + m]; e[2] = B.nth[nb + m]; e[3] = 0.0; e[4] = B.ntt[nb + m];
      e[5] = (double)B.nplan[nb + m]; e[6] = B.nlen[nb + m];
    }
  };
  if (lane == 0) node_elem(s.best_leaf, pos);
  pos--;
  for (int m = s.best_leaf; B.parent[nb + m] >= 0; m = B.parent[nb + m]) {
    const int cnt = B.pt_cnt[nb + m], off = B.pt_off[nb + m];
    for (int k = lane; k < cnt; k += 64) {
      // point k of the node sits k places after the node it grew from
      double* e = o + 7 * (size_t)(pos - cnt + 1 + k);
      size_t gi = pb + off + k;
      e[0] = B.px[gi]; e[1] = B.py[gi]; e[2] = B.pth[gi]; e[3] = B.pv[gi]; e[4] = B.ptt[gi];
      e[5] = (double)B.nplan[nb + m]; e[6] = B.plen[gi];
    }
    pos -= cnt;
    if (lane == 0) node_elem(B.parent[nb + m], pos);
    pos--;
  }
}

}  // namespace auvp
#endif
