// rrt_rows_kernel.h -- RRT.exploring (path_planning/rrt_dubins.py:92-176), time-bin sampling, FOUR episodes per
// wavefront: one 16-lane row (the DPP row) per episode.
//
// rrt_explore_kernel gives a whole wavefront to one episode and is bound by vector-instruction issue with ~20 % of
// the lanes doing useful work (a steer has 14.5 sub-arcs on average, the bookkeeping is one lane's worth).  Here every
// vector instruction serves four episodes: what was wave-uniform (and lived in scalar registers) becomes row-uniform
// (the same value in the 16 lanes of a row), ballots are cut into 16-bit row masks, cross-lane reads stay inside a
// row.  A steer of up to 30 sub-arcs takes two passes of 15 (lane 15 of a row carries the pass-entry angle), the
// running sums keep the reference's left-to-right order (theta by a DPP row shift chain, x/y/t/length by four lanes
// per row through LDS).  The results are bit-identical to rrt_explore_kernel's; rrt_leaf_kernel finishes both.
//
// LDS per episode: the MT19937 state (2 496 B), one scratch block (the tempered random() window of a pass, then the
// four running-sum rows: 576 B) and the time-bin counters as u16.  A workgroup is 12 waves = 48 episodes sharing the
// world tables and the obstacle tile: one workgroup per CU, three waves per SIMD.
//
// Limits (the host falls back to rrt_explore_kernel beyond them): time-bin mode, no diagnostics, freq <= 30,
// <= 256 obstacles, <= 65 534 iterations (u16 bin counters).
#ifndef AUVP_RRT_ROWS_KERNEL_H
#define AUVP_RRT_ROWS_KERNEL_H
#include <type_traits>

#include "rrt_explore_kernel.h"

namespace auvp {

constexpr int RW_WAVES = 12;  // most waves per workgroup (48 episodes: one workgroup fills a CU's LDS); small batches use fewer
constexpr int RW_ROWS = 4;    // episodes per wave
constexpr int RW_C = 15;      // sub-arcs per steer pass (lane 15 of the row: pass-entry angle)
constexpr int RW_MAX_FREQ = 2 * RW_C;
constexpr int RW_MAX_OBST = 256;
constexpr int RW_WIN = 3 * RW_C + 3;  // random() values a pass may look at

struct RowsLdsPlan {
  int tables, mt, scratch, bins, per_ep, obst, total;
};

__host__ __device__ inline RowsLdsPlan rrt_rows_lds_plan(int K, int n_obst_slots, int tables_bytes, int waves = RW_WAVES) {
  RowsLdsPlan p;
  p.tables = (tables_bytes + 15) & ~15;
  p.mt = 624 * 4;
  const int win = RW_WIN * 8, inc = 4 * 18 * 8;
  p.scratch = ((win > inc ? win : inc) + 15) & ~15;
  p.bins = (((K + 2) * 2) + 15) & ~15;
  p.per_ep = p.mt + p.scratch + p.bins;
  p.obst = n_obst_slots * (8 + 8 + 4);  // x, y f64; cull radius f32
  p.total = p.tables + waves * RW_ROWS * p.per_ep + p.obst;
  return p;
}

// ---- row helpers (16 lanes = one episode) ----
__device__ __forceinline__ uint32_t row_ballot(bool pred, int rowbase) {
  return (uint32_t)((wave_ballot(pred) >> rowbase) & 0xffffull);
}
__device__ __forceinline__ int row_read(int v, int src_lane) { return __shfl(v, src_lane, 64); }
__device__ __forceinline__ double row_read_f64(double v, int src_lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __shfl((int)(b & 0xffffffffll), src_lane, 64), hi = __shfl((int)(b >> 32), src_lane, 64);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// minimum / maximum over the 16 lanes of a row, in every lane: four rotate steps on the DPP path (row_ror 8, 4, 2, 1) instead of
// four xor shuffles -- `__shfl_xor` is a ds_bpermute per 32-bit half, an LDS round trip per step for a lone chain
template <int N>
__device__ __forceinline__ double row_ror_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x120 + N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x120 + N, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double row_min_f64(double v) {
  v = __builtin_fmin(v, row_ror_f64<8>(v)); v = __builtin_fmin(v, row_ror_f64<4>(v));
  v = __builtin_fmin(v, row_ror_f64<2>(v)); v = __builtin_fmin(v, row_ror_f64<1>(v));
  return v;
}
__device__ __forceinline__ double row_max_f64(double v) {
  v = __builtin_fmax(v, row_ror_f64<8>(v)); v = __builtin_fmax(v, row_ror_f64<4>(v));
  v = __builtin_fmax(v, row_ror_f64<2>(v)); v = __builtin_fmax(v, row_ror_f64<1>(v));
  return v;
}
// value of the previous lane of the row (DPP row_shr:1); lane 0 of every row gets `first`
__device__ __forceinline__ double row_prev_f64(double v, double first) {
  const long long b = __double_as_longlong(v), f = __double_as_longlong(first);
  const int lo = __builtin_amdgcn_update_dpp((int)(f & 0xffffffffll), (int)(b & 0xffffffffll), 0x111, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(f >> 32), (int)(b >> 32), 0x111, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// theta += phi, left to right, for the four rows of a wavefront at once: every lane ends with
// (((theta0 + phi_0) + phi_1) + ... + phi_rl), phi_j = the value lane j of its row wrote to `row_phi[j]`.  Lane s adds
// phi_0 .. phi_s itself, the values broadcast out of LDS, under an exec mask that drops lanes rl < j at step j: ONE vector
// instruction per step (the DPP row-shift chain this replaces cost five: two v_mov_dpp + the add + the merge of the first
// lane).  The masks are constants -- lanes with (lane & 15) >= j -- and only shrink, so the LDS reads issued under them still
// reach every lane that will use them; four reads stay in flight.  Expects all 64 lanes enabled at the call (the kernels
// call it from wave-uniform control flow only: the step masks are absolute) and restores the mask it found.
__device__ __forceinline__ double rows_theta_chain(double theta0, const double* row_phi) {
  double th = theta0, t0, t1, t2, t3;
  const uint32_t a = (uint32_t)(size_t)((const __attribute__((address_space(3))) double*)row_phi);
  // (the entry mask is saved and put back: should the compiler ever lower an enclosing condition as divergent, lanes that were
  // masked off stay off -- ADVICE r5; one scalar move)
  unsigned long long exec_in;
  __asm__ volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "ds_read_b64 %[t0], %[a] offset:0\n\t"
      "ds_read_b64 %[t1], %[a] offset:8\n\t"
      "ds_read_b64 %[t2], %[a] offset:16\n\t"
      "ds_read_b64 %[t3], %[a] offset:24\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "v_add_f64 %[th], %[th], %[t0]\n\t"
      "ds_read_b64 %[t0], %[a] offset:32\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "s_mov_b32 exec_lo, 0xfffefffe\n\ts_mov_b32 exec_hi, 0xfffefffe\n\t"
      "v_add_f64 %[th], %[th], %[t1]\n\t"
      "ds_read_b64 %[t1], %[a] offset:40\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "s_mov_b32 exec_lo, 0xfffcfffc\n\ts_mov_b32 exec_hi, 0xfffcfffc\n\t"
      "v_add_f64 %[th], %[th], %[t2]\n\t"
      "ds_read_b64 %[t2], %[a] offset:48\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "s_mov_b32 exec_lo, 0xfff8fff8\n\ts_mov_b32 exec_hi, 0xfff8fff8\n\t"
      "v_add_f64 %[th], %[th], %[t3]\n\t"
      "ds_read_b64 %[t3], %[a] offset:56\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "s_mov_b32 exec_lo, 0xfff0fff0\n\ts_mov_b32 exec_hi, 0xfff0fff0\n\t"
      "v_add_f64 %[th], %[th], %[t0]\n\t"
      "ds_read_b64 %[t0], %[a] offset:64\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "s_mov_b32 exec_lo, 0xffe0ffe0\n\ts_mov_b32 exec_hi, 0xffe0ffe0\n\t"
      "v_add_f64 %[th], %[th], %[t1]\n\t"
      "ds_read_b64 %[t1], %[a] offset:72\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "s_mov_b32 exec_lo, 0xffc0ffc0\n\ts_mov_b32 exec_hi, 0xffc0ffc0\n\t"
      "v_add_f64 %[th], %[th], %[t2]\n\t"
      "ds_read_b64 %[t2], %[a] offset:80\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "s_mov_b32 exec_lo, 0xff80ff80\n\ts_mov_b32 exec_hi, 0xff80ff80\n\t"
      "v_add_f64 %[th], %[th], %[t3]\n\t"
      "ds_read_b64 %[t3], %[a] offset:88\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "s_mov_b32 exec_lo, 0xff00ff00\n\ts_mov_b32 exec_hi, 0xff00ff00\n\t"
      "v_add_f64 %[th], %[th], %[t0]\n\t"
      "ds_read_b64 %[t0], %[a] offset:96\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "s_mov_b32 exec_lo, 0xfe00fe00\n\ts_mov_b32 exec_hi, 0xfe00fe00\n\t"
      "v_add_f64 %[th], %[th], %[t1]\n\t"
      "ds_read_b64 %[t1], %[a] offset:104\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "s_mov_b32 exec_lo, 0xfc00fc00\n\ts_mov_b32 exec_hi, 0xfc00fc00\n\t"
      "v_add_f64 %[th], %[th], %[t2]\n\t"
      "ds_read_b64 %[t2], %[a] offset:112\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "s_mov_b32 exec_lo, 0xf800f800\n\ts_mov_b32 exec_hi, 0xf800f800\n\t"
      "v_add_f64 %[th], %[th], %[t3]\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "s_mov_b32 exec_lo, 0xf000f000\n\ts_mov_b32 exec_hi, 0xf000f000\n\t"
      "v_add_f64 %[th], %[th], %[t0]\n\t"
      "s_waitcnt lgkmcnt(1)\n\t"
      "s_mov_b32 exec_lo, 0xe000e000\n\ts_mov_b32 exec_hi, 0xe000e000\n\t"
      "v_add_f64 %[th], %[th], %[t1]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_mov_b32 exec_lo, 0xc000c000\n\ts_mov_b32 exec_hi, 0xc000c000\n\t"
      "v_add_f64 %[th], %[th], %[t2]\n\t"
      "s_mov_b64 exec, %[sv]"
      : [th] "+v"(th), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [sv] "=&s"(exec_in)
      : [a] "v"(a)
      : "memory");
  return th;
}

// Per-row view of the MT19937 stream (state in the episode's LDS block, refilled in place like WaveRng, 16 words per
// round and row).  pslot / avail / drawn are row-uniform.
struct RowRng {
  uint32_t* s;
  uint32_t pslot, avail;
  unsigned long long drawn;
};

// make sure every row that asks (`want`) has `need` words generated ahead of its consumer.  The generation frontier
// pslot + avail is kept a multiple of 16 (624 = 39 x 16; it starts at 624 = 0): whole 16-word blocks are regenerated, so
// a block never wraps and lane rl's word is block + rl.
__device__ __forceinline__ void rows_ensure(RowRng& r, bool want, uint32_t need, int rl) {
  // (a row that does not ask needs 0 words: the loop's vote is then the ballot of ONE compare, `avail < need` -- a vote on
  // `want && ...` costs two more vector instructions, a 0 / 1 and its compare with zero)
  const uint32_t need_w = want ? need : 0u;
  for (;;) {
    const bool go = r.avail < need_w;
    if (__builtin_amdgcn_uicmp(r.avail, need_w, 36 /* unsigned < */) == 0ull) break;
    // up to 32 words per row and round, two per lane (words 227 apart are independent, so any 32 consecutive are)
    uint32_t n = (624u - r.avail) & ~15u;
    n = n < 32u ? n : 32u;
    n = go ? n : 0u;
    uint32_t f0 = r.pslot + r.avail;
    f0 = f0 >= 624u ? f0 - 624u : f0;
    uint32_t v2[2], kk[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      uint32_t fh = f0 + 16u * (uint32_t)h;
      fh = fh >= 624u ? fh - 624u : fh;
      const uint32_t k = fh + (uint32_t)rl;
      const uint32_t k1 = (k == 623u) ? 0u : k + 1u;
      uint32_t km = k + 397u;
      km = km >= 624u ? km - 624u : km;
      const uint32_t a = r.s[k], b = r.s[k1], c = r.s[km];
      const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
      v2[h] = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      kk[h] = k;
    }
    wave_sync();  // every lane's reads are issued before any lane's write
    if (n >= 16u) r.s[kk[0]] = v2[0];
    if (n >= 32u) r.s[kk[1]] = v2[1];
    wave_sync();
    r.avail += n;
  }
}
__device__ __forceinline__ double rows_random_at(const RowRng& r, uint32_t j) {
  uint32_t k = r.pslot + 2u * j;
  k = k >= 624u ? k - 624u : k;
  const uint32_t k1 = (k + 1u == 624u) ? 0u : k + 1u;
  const uint32_t a = mt_temper(r.s[k]) >> 5, b = mt_temper(r.s[k1]) >> 6;
  return py_random_from(a, b);
}
__device__ __forceinline__ void rows_advance(RowRng& r, bool on, uint32_t nwords) {
  if (on) {
    uint32_t p = r.pslot + nwords;  // nwords < 624: one wrap at most
    p = p >= 624u ? p - 624u : p;
    r.pslot = p;
    r.avail -= nwords;
    r.drawn += nwords;
  }
}

// (internal linkage: the kernel is compiled -- and launched -- by rows_kernels.hip, a translation unit with its own scheduler
// strategy; other units that include this header for the row helpers drop their unused copy)
static __global__ __launch_bounds__(RW_WAVES * 64, 1) void rrt_rows_kernel(WorldDev W, RrtParamsDev P, RrtBuffers B, int n_episodes) {
  extern __shared__ __align__(16) unsigned char smem[];
  const RrtTables S = rrt_tables_view(smem, W.n_habitats, W.n_poly);
  const int wave = (int)(threadIdx.x >> 6);
  const int lane = lane_id();
  const int row = lane >> 4, rl = lane & 15, rowbase = lane & 48;
  const int K = P.K;
  const int wg_waves = (int)(blockDim.x >> 6);
  const RowsLdsPlan plan = rrt_rows_lds_plan(K, RW_MAX_OBST, rrt_tables_bytes(W.n_habitats, W.n_poly, W.n_bins), wg_waves);
  unsigned char* ebase = smem + plan.tables + (size_t)(wave * RW_ROWS + row) * plan.per_ep;
  uint32_t* mt = reinterpret_cast<uint32_t*>(ebase);
  double* win = reinterpret_cast<double*>(ebase + plan.mt);  // [RW_WIN] tempered random() values of a pass
  double* inc = win;                                         // [4][18] running sums (aliases the window once it is dead)
  uint16_t* bin_count = reinterpret_cast<uint16_t*>(ebase + plan.mt + plan.scratch);

  rrt_tables_stage(S, W);
  if (threadIdx.x == 0) *S.params = P;
  const RrtParamsDev& Q = *S.params;
  double* olx = reinterpret_cast<double*>(smem + plan.tables + (size_t)wg_waves * RW_ROWS * plan.per_ep);
  double* oly = olx + RW_MAX_OBST;
  float* olr = reinterpret_cast<float*>(oly + RW_MAX_OBST);
  // the spatially sorted tile (WorldDev::os_*): slot s = obstacles 16 s .. 16 s + 15
  for (int i = threadIdx.x; i < RW_MAX_OBST; i += blockDim.x) { olx[i] = W.os_x[i]; oly[i] = W.os_y[i]; olr[i] = W.os_r[i]; }
  __syncthreads();

  const int ep = ((int)blockIdx.x * wg_waves + wave) * RW_ROWS + row;
  bool live = ep < n_episodes;  // row-uniform; a row that fails keeps running as a no-op until the wave is done
  const int eps = live ? ep : 0;
  if (!wave_any(live)) return;

  // ---- per-episode views ----
  const int capn = B.cap_nodes, capp = B.cap_points, bcap = B.bin_cap;
  double* nodeF = B.node_f + (size_t)eps * capn * 8;
  int4* nodeI = reinterpret_cast<int4*>(B.node_i) + (size_t)eps * capn;
  uint8_t* nodeQ = B.node_q + (size_t)eps * capn;
  double* ptF = B.points + (size_t)eps * capp * 6;
  const BinLists bins = bin_lists(B, (size_t)eps, K);
  int next_chunk = 0;
  const double* init = B.init + (size_t)eps * 6;

  RowRng rng;
  rng.s = mt;
  for (int i = rl; i < 624; i += 16) mt[i] = B.mt[(size_t)eps * 624 + i];
  {
    int idx = B.mt_index ? B.mt_index[eps] : 624;
    idx = idx < 0 ? 0 : (idx > 624 ? 624 : idx);
    rng.pslot = idx == 624 ? 0u : (uint32_t)idx;
    rng.avail = (uint32_t)(624 - idx);
    rng.drawn = 0ull;
  }
  for (int i = rl; i < K + 2; i += 16) bin_count[i] = 0;
  wave_sync();
  if (live && rl == 0) {
    nodeF[0] = init[0]; nodeF[1] = init[1]; nodeF[2] = init[2]; nodeF[3] = init[3]; nodeF[4] = init[5];
    nodeI[0] = make_int4(0, -1, 0, 0);
    nodeQ[0] = 0;  // the start state is never a leaf candidate
    bins.direct[(K >= 1 ? 1 : 0) * AUVP_BIN_HEAD] = 0;
    bin_count[K >= 1 ? 1 : 0] = 1;
  }
  wave_sync();
  int n_nodes = 1, n_points = 0, status = 0, n_cand = 0, iters_run = 0;
  // lane rl keeps the bounding box of obstacle slot rl: one compare round tells which slots a steer can touch
  const double4 sbox = reinterpret_cast<const double4*>(W.os_box)[rl];
  const int nv_poly = W.n_poly;

  for (int it = 0; it < P.max_iter; it++) {
    if (!wave_any(live)) break;
    // ------------------------------------------------------------ parent selection (:121-127)
    // lane rl of a row tries draw rl: ran_bin = int(uniform(1, K+1)) until that bin is non-empty; the first success in
    // stream order wins, a key beyond K before it is a KeyError.  14 tries per round leave room for the two draws
    // that follow the successful one.
    int par = 0, n_total = 0, base = 0;
    {
      bool search = live;
      int fo = 0, rb = 0, cnt = 0;
      double u = 0.0;
      for (;;) {
        rows_ensure(rng, search, 32u, rl);
        // (the round's sixteen draws also go to the window area, free at this point: they ARE window entries 0 .. 15 of the
        // first pass -- stream entry e at win[e] -- and the two draws after the winner's are read from there)
        if (search) { u = rows_random_at(rng, (uint32_t)rl); win[rl] = u; }
        const int rbj = (int)py_uniform(1.0, (double)(K + 1), u);
        const bool cand = search && rl < 14;
        const bool badkey = cand && rbj > K;
        const int cj = (cand && !badkey) ? (int)bin_count[rbj] : 0;
        const uint32_t okm = row_ballot(cj != 0, rowbase), badm = row_ballot(badkey, rowbase);
        const int f_ok = okm ? (__ffs((int)okm) - 1) : 16, f_bad = badm ? (__ffs((int)badm) - 1) : 16;
        if (search) {
          if (f_bad < f_ok) { status = -5; live = false; iters_run = it; search = false; }
          else if (f_ok < 16) { fo = f_ok; search = false; }
        }
        const bool again = search;
        rows_advance(rng, again, 28u);  // 14 unsuccessful draws
        const int src = rowbase + (again ? 0 : fo);
        // rows that just finished pick up the winner's values (rows still searching read garbage they overwrite later)
        const int rb_n = row_read(rbj, src), cnt_n = row_read(cj, src);
        if (!again && live && cnt == 0) { rb = rb_n; cnt = cnt_n; }
        if (!wave_any(again)) break;
      }
      wave_sync();
      const double u1 = win[fo + 1], u2 = win[fo + 2];
      if (live) {
        const int ri = (int)py_uniform(0.0, (double)cnt, u1);
        par = bin_member(bins, rb, ri);
        n_total = (int)auvp_floor(py_uniform(0.0, Q.freq, u2) / 1);
        base = fo + 3;
      }
    }
    // ------------------------------------------------------------ steer (:252-295), passes of RW_C sub-arcs
    // window entry j of a pass = random() number b0 + j of the row's stream
    auto make_window = [&](bool on, int nwin, int b0) {
      rows_ensure(rng, on, (uint32_t)(2 * (b0 + nwin)), rl);
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const int j = rl + 16 * t;
        if (on && j < nwin) win[j] = rows_random_at(rng, (uint32_t)(b0 + j));
      }
      wave_sync();
    };
    // the first pass's window needs nothing of the parent: it is generated while the parent's id (and then its record)
    // is still on its way from memory
    const bool on0 = live && 0 < n_total;
    const int n0_ = on0 ? (n_total < RW_C ? n_total : RW_C) : 0;
    {
      // first pass: window entry j = stream entry base + j = win[base + j]; entries below 16 are there already
      const int e_end = base + 3 * n0_;
      rows_ensure(rng, on0, (uint32_t)(2 * e_end), rl);
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const int e = 16 + rl + 16 * t;
        if (on0 && e < e_end) win[e] = rows_random_at(rng, (uint32_t)e);
      }
      wave_sync();
    }
    double cx = 0.0, cy = 0.0, cth = 0.0, ctt = 0.0, clen = 0.0;
    if (live) {
      const double2 a = *reinterpret_cast<const double2*>(nodeF + (size_t)par * 8);
      const double2 b = *reinterpret_cast<const double2*>(nodeF + (size_t)par * 8 + 2);
      cx = a.x; cy = a.y; cth = b.x; ctt = b.y;
      clen = nodeF[(size_t)par * 8 + 4];
    }
    const double px0 = cx, py0 = cy, clen0 = clen;
    int cnt = 0;  // appended path points of this row's steer
    // the lane's own path point of pass 0 / pass 1 (kept for the exact collision and boundary tests)
    double ptx[2] = {0.0, 0.0}, pty[2] = {0.0, 0.0};
    bool ptv[2] = {false, false};
    bool first_pass = true;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const int c0 = pass * RW_C;
      const bool on = live && c0 < n_total;  // rows with sub-arcs left
      if (!wave_any(on)) break;
      const int n = on ? ((n_total - c0) < RW_C ? (n_total - c0) : RW_C) : 0;
      const int nwin = 3 * n;
#ifdef AUVP_ROWS_PAD
      // EXPERIMENT ONLY (tools/rows_pass_probe.py, profiles/r4_rows_packing.md; never defined in the product build): AUVP_ROWS_PAD
      // extra vector instructions per steer pass -- independent fp64 adds on four registers, results unused -- to measure what
      // a pass can afford to carry before packing the rows' sub-arcs stops paying
      {
        double pd0 = cth, pd1 = cx, pd2 = cy, pd3 = ctt;
#pragma unroll
        for (int q = 0; q < AUVP_ROWS_PAD / 4; q++)
          asm volatile("v_add_f64 %0, %0, 1.0\n\tv_add_f64 %1, %1, 1.0\n\tv_add_f64 %2, %2, 1.0\n\tv_add_f64 %3, %3, 1.0"
                       : "+v"(pd0), "+v"(pd1), "+v"(pd2), "+v"(pd3));
      }
#endif
      const int b0 = first_pass ? base : 0;  // later passes start at the (advanced) head of the stream
      if (pass != 0) make_window(on, nwin, b0);
      const double* wp = win + (pass == 0 ? base : 0);  // window entry j of this pass
      // "taken" predicate for every possible start offset, 48 bits per row
      unsigned long long tpred = 0ull;
#pragma unroll
      for (int t = 0; t < 3; t++) {
        // every lane compares (entries past the window are whatever the scratch holds: never a trap, masked below); the
        // predicate is the AND of two single-compare ballots, taken in scalar registers (nwin = 0 for rows that are not `on`)
        const int j = rl + 16 * t;
        const double dist = py_uniform(0.0, Q.dist_to_end, wp[j]);
        const double diff = py_uniform(-Q.diff_max, Q.diff_max, wp[j + 1]);
        const unsigned long long fb = __builtin_amdgcn_fcmp(auvp_fabs(dist), auvp_fabs(diff), 2 /* ordered > */) &
                                      __builtin_amdgcn_uicmp((unsigned)(j + 1), (unsigned)nwin, 36 /* unsigned < */);
        tpred |= ((fb >> rowbase) & 0xffffull) << (16 * t);
      }
      // where does sub-arc s start?  pos_s = 2s + (#taken among sub-arcs < s): fixed point of
      //   taken_s = T[2s + c_s],  c_s = popcount(taken below s),  started from "everything taken"
      // (the votes are ballots of ONE compare each -- `__builtin_amdgcn_uicmp` -- with the lanes that do not take part made
      // neutral through their data: a vote on `active && x` costs two more vector instructions, a 0 / 1 and its compare)
      const bool active = rl < n;
      const uint32_t mywin = active ? (uint32_t)(tpred >> (2 * rl)) : 0u;  // bits 0 .. rl are looked at (cbelow <= rl <= 14)
      const uint32_t below_me = active ? ((1u << rl) - 1u) : 0u;
      int cbelow = active ? rl : 0;
      uint32_t tmask;
      for (;;) {
        const unsigned long long tb = __builtin_amdgcn_uicmp((mywin >> cbelow) & 1u, 0u, 33 /* != */);
        tmask = (uint32_t)((tb >> rowbase) & 0xffffull);
        const int cnew = __popc(tmask & below_me);
        const unsigned long long chg = __builtin_amdgcn_uicmp((unsigned)cnew, (unsigned)cbelow, 33 /* != */);
        cbelow = cnew;
        if (chg == 0ull) break;
      }
      const int mypos = 2 * rl + cbelow;
      const int used = 2 * n + __popc(tmask);
      const bool taken = (tmask >> rl) & 1u;
      double radius = 0.0, phi = 0.0, vt = 1.0;
      if (taken) {
        const double dist = py_uniform(0.0, Q.dist_to_end, wp[mypos]);
        const double diff = py_uniform(-Q.diff_max, Q.diff_max, wp[mypos + 1]);
        const double s1 = dist + diff, s2 = dist - diff;
        radius = auvp_div_plain(s1 + s2, -s1 + s2);
        phi = auvp_div_plain(s1 + s2, 2 * radius);
        vt = py_uniform(0.0, 2 * Q.v, wp[mypos + 2]);
      }
      wave_sync();  // the window is dead: its LDS becomes the running-sum scratch
      // theta += phi, left to right: lane s ends with (((theta0 + phi_0) + phi_1) + ... + phi_s); untaken and idle lanes add
      // an exact 0.0 (rows_theta_chain: the phis go through the row's LDS scratch)
      inc[rl] = phi;
      wave_sync();
      const double th = rows_theta_chain(cth, inc);
      const double myth = (rl == 15) ? cth : th;  // lane 15: the pass-entry angle
      double sn, cs;
      auvp_sincos_sk(myth, &sn, &cs);
      double dx = 0.0, dy = 0.0, mv = 0.0, dt = 0.0;
      {
        const uint32_t below = tmask & ((1u << rl) - 1u);
        const int prev = below ? (31 - __clz((int)below)) : 15;
        const double so = row_read_f64(sn, rowbase + prev), co = row_read_f64(cs, rowbase + prev);
        if (taken) {
          dx = radius * (sn - so);
          dy = radius * (-cs + co);
          mv = auvp_sqrt_plain(dx * dx + dy * dy);
          dt = auvp_div_plain(mv, vt);
        }
      }
      // x += dx; y += dy; t += dt; length += movement: four serial chains per row, lanes 0..3, 18-double rows in LDS
      if (rl < 15) { inc[rl] = dx; inc[18 + rl] = dy; inc[36 + rl] = dt; inc[54 + rl] = mv; }
      else { inc[15] = 0.0; inc[33] = 0.0; inc[51] = 0.0; inc[69] = 0.0; }  // entry 15 pads the last 16-byte pair
      wave_sync();
      if (rl < 4 && on) {
        // entries past n hold exact zeros, so all 16 steps can run: eight independent 16-byte reads up front, sixteen
        // chained additions, eight writes -- no loop, no round trip per step
        double acc = rl == 0 ? cx : (rl == 1 ? cy : (rl == 2 ? ctt : clen));
        double2* rowp = reinterpret_cast<double2*>(inc + rl * 18);
        double2 v[8];
#pragma unroll
        for (int s2 = 0; s2 < 8; s2++) v[s2] = rowp[s2];
#pragma unroll
        for (int s2 = 0; s2 < 8; s2++) { acc = acc + v[s2].x; v[s2].x = acc; acc = acc + v[s2].y; v[s2].y = acc; }
#pragma unroll
        for (int s2 = 0; s2 < 8; s2++) rowp[s2] = v[s2];
      }
      wave_sync();
      double mx = 0.0, my = 0.0, mt_ = 0.0, ml = 0.0;
      if (active) { mx = inc[rl]; my = inc[18 + rl]; mt_ = inc[36 + rl]; ml = inc[54 + rl]; }
      const bool app = taken && (mv >= Q.min_dist);
      const uint32_t amask = row_ballot(app, rowbase);
      const int napp = __popc(amask);
      if (on && (n_points + cnt + napp > capp)) { status = -2; live = false; iters_run = it; }
      const bool wr = app && live;
      if (wr) {
        const int rank = __popc(amask & ((1u << rl) - 1u));
        const size_t gi = (size_t)(n_points + cnt + rank);  // speculative: committed only if the node is accepted
        double* ra = ptF + gi * 3;                       // x, y, traj_t: what the leaf pass reads
        double* rb = ptF + (size_t)capp * 3 + gi * 3;    // theta, v, length
        // (92 GB per launch that this kernel never reads back: non-temporal stores, -0.3 %)
        typedef double nt_f64x2 __attribute__((ext_vector_type(2)));
        nt_f64x2 va, vb; va.x = mx; va.y = my; vb.x = myth; vb.y = vt;
        __builtin_nontemporal_store(va, reinterpret_cast<nt_f64x2*>(ra)); __builtin_nontemporal_store(mt_, ra + 2);
        __builtin_nontemporal_store(vb, reinterpret_cast<nt_f64x2*>(rb)); __builtin_nontemporal_store(ml, rb + 2);
      }
      ptx[pass] = mx; pty[pass] = my; ptv[pass] = wr;
      {
        // the row's state after this pass = the prefix values of its last sub-arc
        const int last = n > 0 ? n - 1 : 0;
        const double th_last = row_read_f64(myth, rowbase + last);
        if (on) {
          cnt += napp;
          cx = inc[last]; cy = inc[18 + last]; ctt = inc[36 + last]; clen = inc[54 + last];
          cth = th_last;
        }
      }
      rows_advance(rng, on && live, (uint32_t)(2 * (b0 + used)));  // (a row that just failed keeps its stream position, like the one-episode kernel)
      first_pass = false;
      wave_sync();
    }
    // a steer without sub-arcs consumed only the selection and n_expand draws
    rows_advance(rng, live && n_total == 0, (uint32_t)(2 * base));

    // ------------------------------------------------------------ check_collision (:530-549)
    // conservative cull: the square around the parent's end that holds every prefix position (half-width = the steer's
    // total movement) against each obstacle's bounding square; survivors get the exact test d2 <= T_i on every point
    const double reach = clen - clen0;
    const double bx0 = px0 - reach, by0 = py0 - reach, bx1 = px0 + reach, by1 = py0 + reach;
    const double slack = 0x1p-30 * (auvp_fabs(bx0) + auvp_fabs(bx1) + auvp_fabs(by0) + auvp_fabs(by1) + 1.0);
    const double hx = reach + slack;
    bool hit = false;
    // Two instances of the same loop, picked by a launch-uniform flag: where the host expects dense obstacles the cull goes
    // on with the tight box of the row's path points (and the parent's end) instead of the reach square -- the path
    // wanders inside a fraction of it, and an obstacle can only be hit if its bounding square meets that box.  The
    // sparse instance is the plain loop (the ~100 instructions of the tight box would cost more than they save there).
    // (the extent used by the boundary test below: the reach square, or the tight box once the dense instance has it)
    double ex0 = bx0, ey0 = by0, ex1 = bx1, ey1 = by1;
    auto cull_and_test = [&](auto tight_tag) {
      constexpr bool TIGHT = decltype(tight_tag)::value;
      const double hs = hx + slack;
      const bool slot_hit = live && !(sbox.z < px0 - hs || sbox.x > px0 + hs || sbox.w < py0 - hs || sbox.y > py0 + hs);
      uint32_t sm = row_ballot(slot_hit, rowbase);
      double tcx = px0, tcy = py0, thx = hx, thy = hx;
      if (TIGHT) {  // (dense worlds: nearly every steer has a slot within reach, and the boundary test profits as well)
        double mnx = px0, mxx = px0, mny = py0, mxy = py0;
#pragma unroll
        for (int q = 0; q < 2; q++) {
          if (ptv[q]) {
            mnx = __builtin_fmin(mnx, ptx[q]); mxx = __builtin_fmax(mxx, ptx[q]);
            mny = __builtin_fmin(mny, pty[q]); mxy = __builtin_fmax(mxy, pty[q]);
          }
        }
        mnx = row_min_f64(mnx); mxx = row_max_f64(mxx); mny = row_min_f64(mny); mxy = row_max_f64(mxy);
        const double ts = 0x1p-30 * (auvp_fabs(mnx) + auvp_fabs(mxx) + auvp_fabs(mny) + auvp_fabs(mxy) + 1.0);
        tcx = (mnx + mxx) * 0.5; tcy = (mny + mxy) * 0.5;
        thx = (mxx - mnx) * 0.5 + ts; thy = (mxy - mny) * 0.5 + ts;
        const bool tight = slot_hit && !(sbox.z < mnx - ts || sbox.x > mxx + ts || sbox.w < mny - ts || sbox.y > mxy + ts);
        sm = row_ballot(tight, rowbase);
        ex0 = mnx; ey0 = mny; ex1 = mxx; ey1 = mxy;
      }
      while (wave_any(sm != 0u)) {  // slots some row has to look into (none at all for most steers of a sparse world)
        const bool hs_ = sm != 0u;
        const int j0 = hs_ ? 16 * (__ffs((int)sm) - 1) : 0;
        sm &= sm - 1u;
        const int oi = j0 + rl;
        const double oxj = olx[oi], oyj = oly[oi], orj = (double)olr[oi];
        const double otj = W.os_t[oi];  // this lane's obstacle of the slot: one coalesced read, handed out below
        const bool cand = hs_ && (TIGHT ? !(auvp_fabs(oxj - tcx) > thx + orj || auvp_fabs(oyj - tcy) > thy + orj)
                                        : !(auvp_fabs(oxj - px0) > hx + orj || auvp_fabs(oyj - py0) > hx + orj));
        uint32_t cm = row_ballot(cand, rowbase);
        n_cand += __popc(cm);
        while (wave_any(cm != 0u)) {
          const bool has = cm != 0u;
          const int cl = has ? (__ffs((int)cm) - 1) : 0;
          cm &= cm - 1u;
          const double ox = row_read_f64(oxj, rowbase + cl), oy = row_read_f64(oyj, rowbase + cl), ot = row_read_f64(otj, rowbase + cl);
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const double ddx = ptx[q] - ox, ddy = pty[q] - oy;
            hit |= has && ptv[q] && (ddx * ddx + ddy * ddy <= ot);
          }
          // the parent's end, path[0], is a path point too
          if (rl == 15) { const double ddx = px0 - ox, ddy = py0 - oy; hit |= has && (ddx * ddx + ddy * ddy <= ot); }
          // a row that has its collision is done with this slot's candidates (they are counted above already)
          if (row_ballot(hit, rowbase) != 0u) cm = 0u;
        }
      }
    };
    if (P.flags & AUVP_KFLAG_TIGHT_CULL) cull_and_test(std::true_type{});
    else cull_and_test(std::false_type{});
    // boundary: strictly inside an axis-aligned rectangle implies Point.within; otherwise the crossing test per point
    const double* sb = S.world->safe_box;
    const bool box_inside = W.has_safe_box && ex0 > sb[0] && ey0 > sb[1] && ex1 < sb[2] && ey1 < sb[3];
    bool outside = false;
    if (wave_any(live && !box_inside)) {
      // lane = path point (two passes' points + the parent's end on lane 15); edges of the polygon one at a time
      auto crossing_outside = [&](double x, double y) {
        if (nv_poly <= 0) return true;
        int par_ = 0;
        for (int e = 0; e < nv_poly; e++) {
          const int ej = e == 0 ? nv_poly - 1 : e - 1;
          const double xi = S.poly[e][0], yi = S.poly[e][1], xj = S.poly[ej][0], yj = S.poly[ej][1];
          if ((yi > y) != (yj > y)) par_ ^= (x < (xj - xi) * (y - yi) / (yj - yi) + xi) ? 1 : 0;
        }
        return (par_ & 1) == 0;
      };
      const bool need = live && !box_inside;
#pragma unroll
      for (int q = 0; q < 2; q++)
        if (need && ptv[q]) outside |= crossing_outside(ptx[q], pty[q]);
      if (need && rl == 15) outside |= crossing_outside(px0, py0);
    }
    const bool bad = row_ballot(hit || outside, rowbase) != 0u;
    const bool ok = live && !bad;
    // ------------------------------------------------------------ accept (:144-151)
    if (ok && n_nodes >= capn) { status = -2; live = false; iters_run = it; }
    const bool acc_ = ok && live;
    const int me = n_nodes;
    if (acc_) {
      // curr_bin = (t // bin_interval + 1) * bin_interval, exact floor of the true quotient
      double q = auvp_floor(ctt * Q.inv_bin_interval);
      const double r = auvp_fma(-q, Q.bin_interval, ctt);
      if (r < 0.0) q -= 1.0;
      else if (r >= Q.bin_interval) q += 1.0;
      const double fi = q + 1.0;
      const double curr_bin = fi * Q.bin_interval;
      const bool over = curr_bin > Q.max_traj_time;
      bool stored = true;
      if (!over || fi <= (double)K) {
        const int bi = (int)fi;
        const int c = over ? 0 : (int)bin_count[bi];  // an overflowing regular key is reset first (:149-151)
        int32_t* slot = (c >= bcap || c >= 65535) ? nullptr : bin_slot_for_append(bins, bi, c, next_chunk, rl == 0);
        if (!slot) { status = -2; live = false; iters_run = it; stored = false; }
        else if (rl == 0) { *slot = me; bin_count[bi] = (uint16_t)(c + 1); }
      }
      if (stored) {
        if (rl == 0) {
          double* nf = nodeF + (size_t)me * 8;
          *reinterpret_cast<double2*>(nf) = make_double2(cx, cy);
          *reinterpret_cast<double2*>(nf + 2) = make_double2(cth, ctt);
          nf[4] = clen;
          nodeI[me] = make_int4(it, par, n_points, cnt);
          nodeQ[me] = ctt >= Q.max_traj_time - 30 ? 1 : 0;  // a qualifying leaf (:158); ranked by rrt_leaf_kernel
        }
        n_nodes++;
        n_points += cnt;
      }
    }
    wave_sync();
  }

  // ---- epilogue ----
  const bool valid = ep < n_episodes;
  if (live) iters_run = P.max_iter;
  const unsigned long long drawn = rng.drawn;
  rows_ensure(rng, valid, 2u, rl);
  const double after = rows_random_at(rng, 0u);
  if (valid) {
    for (int i = rl; i < K + 1; i += 16) B.bin_count[(size_t)ep * (K + 1) + i] = (int32_t)bin_count[i];
    if (rl == 0) {
      RrtSummary& s = B.summary[ep];
      s.status = status; s.n_nodes = n_nodes; s.n_points = n_points; s.n_leaves = 0;
      s.best_leaf = -1; s.best_path_len = 0; s.iters_run = iters_run; s.n_candidates = n_cand;
      s.best_cost[0] = __builtin_inf(); s.best_cost[1] = 0.0; s.best_cost[2] = 0.0; s.best_cost[3] = 0.0;
      s.best_length = 0.0;
      s.rng_after = after; s.leaf_elems = 0; s.n_draw32 = drawn; s.nn_scanned = 0ull;
    }
  }
}

}  // namespace auvp
#endif
