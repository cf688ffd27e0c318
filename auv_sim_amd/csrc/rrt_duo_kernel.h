// rrt_duo_kernel.h -- RRT.exploring (path_planning/rrt_dubins.py:92-176), time-bin sampling, for the LATENCY runs (one
// episode, config 2's 1 024 replicas): TWO wavefronts per episode.
//
// rrt_explore_kernel gives an episode one wavefront, and a lone wavefront issues in order: its 4.1 us per expansion are the
// sum of its instructions' latencies (profiles/r2_latency/README.md), and every in-wave attempt to start the next iteration
// early lost (selection ahead, round 3).  About half of an iteration depends on nothing but the random stream and the bins'
// sizes: the selection's candidate draws, the number of sub-arcs, the window of random() values, the draw-offset fixed
// point, the radii / angles / speeds of the sub-arcs (two divisions each).  Here a HELPER wavefront owns the episode's
// generator and produces that half one iteration ahead, as a packet in LDS; the MAIN wavefront does what needs the tree:
// theta chain, sin / cos, chords, running sums, path points, collision, append.  The two sit on different SIMDs of the CU.
//
// Why the helper may run ahead.  Packet i+1 is built while main runs iteration i, so exactly one append (iteration i's) can
// fall between the helper's look at the bins and the packet's use.  Selectable bins only grow.  The packet records the
// number of appends it was built after (`ver`); main, holding a packet built one append earlier, redoes it only if that
// append touched what the selection looked at: the chosen bin itself (its size feeds `ri`), or -- when the append made a
// bin non-empty -- any bin the selection had found empty.  A redo rewinds the generator to the packet's first word (the
// state is regenerated in place and the helper never lets generation overwrite the words since that point) and rebuilds.
// Results are bit-identical to rrt_explore_kernel's (tests/test_gpu_duo_kernel.py); rrt_leaf_kernel finishes both.
//
// Limits (the host falls back to rrt_explore_kernel beyond them): time-bin mode, no diagnostics, freq <= 30 (one steer
// chunk), <= 256 obstacles, batches of at most four episodes per CU.
#ifndef AUVP_RRT_DUO_KERNEL_H
#define AUVP_RRT_DUO_KERNEL_H
#include "rrt_explore_kernel.h"

namespace auvp {

constexpr int DUO_MAX_FREQ = 30;
constexpr int DUO_CS = 32;        // sub-arc slots of a steer (freq <= 30: lanes 0..29, lane 30 = the entry angle)
constexpr int DUO_EP = 4;         // episodes per workgroup at most (eight wavefronts)
constexpr int DUO_WIN = 160;      // window entries: up to 62 selection draws + 1 + 3 x 30, rounded up

struct DuoPacket {  // LDS: what one iteration needs of the random stream
  // {redo epoch << 32 | iteration + 1}: written LAST, as one 64-bit word; main accepts a packet only under the tag it expects,
  // so a packet that is being rebuilt (same iteration, next epoch) or that belongs to iteration - 2 never matches
  unsigned long long tag;
  int ver, status, rb, rejects, n_total, par;
  unsigned long long tmask;
  double cx, cy, cth, ctt, clen;  // the parent's record
  double radius[DUO_CS], phi[DUO_CS], vt[DUO_CS];
};

struct DuoCtl {  // LDS, per episode
  int ver;            // appends so far (main)
  int last_bi;        // bin of the latest append, and its size before it (main)
  int last_c_before;
  int valid_seq;      // main has accepted packets < valid_seq (the helper may build packet valid_seq)
  int done_seq;       // main has finished iterations < done_seq
  int redo_epoch;     // bumped by main to have the latest packet rebuilt
  int stop;           // main is done (budget used up or an error): the helper posts its final stream position
  int abort;          // a wait gave up
  int helper_done;
  int _pad;
  double final_after;
  unsigned long long final_drawn;
};

__host__ __device__ inline int duo_per_episode_bytes(int K, int max_pts) {
  int b = 624 * 4;                                   // generator (helper)
  b += DUO_WIN * 8;                                  // helper's window of random() values
  b += (((K + 2) * 4) + 15) & ~15;                   // bin sizes (main writes, helper reads)
  b += (int)((sizeof(DuoCtl) + 15) & ~(size_t)15);
  b += 2 * (int)((sizeof(DuoPacket) + 15) & ~(size_t)15);
  b += 7 * DUO_CS * 8;                               // main: inc[4][CS], sc[2][CS], phi_l[CS]
  b += ((max_pts * 16) + 15) & ~15;                  // main: path points x, y
  return b;
}
__host__ __device__ inline int duo_lds_bytes(int K, int max_pts, int n_obst_slots, int tables_bytes, int episodes) {
  return ((tables_bytes + 15) & ~15) + episodes * duo_per_episode_bytes(K, max_pts) + n_obst_slots * (8 + 8 + 8 + 4);
}

// hand-over words in LDS: auvp_wave.h's lds_peek / lds_poke
__device__ __forceinline__ int duo_peek(const int* p) { return lds_peek(p); }
__device__ __forceinline__ void duo_poke(int* p, int v) { lds_poke(p, v); }
__device__ __forceinline__ unsigned long long duo_peek64(const unsigned long long* p) { return lds_peek64(p); }
__device__ __forceinline__ void duo_poke64(unsigned long long* p, unsigned long long v) { lds_poke64(p, v); }
__device__ __forceinline__ unsigned long long duo_tag(int epoch, int iteration) {
  return ((unsigned long long)(uint32_t)epoch << 32) | (unsigned long long)(uint32_t)(iteration + 1);
}
// What the HELPER reads of global memory that MAIN writes (member lists, node records): loads that are served by L2, and
// indices clamped into the episode's own storage -- a packet built on a half-published append is thrown away by main's
// check, but its reads must not leave the allocation.
__device__ __forceinline__ int duo_ld_i32(const int32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ double duo_ld_f64(const double* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ int duo_bin_member(const BinLists& L, int b, int k, int cap_nodes) {
  int m;
  if (k < AUVP_BIN_HEAD) m = duo_ld_i32(L.direct + (b * AUVP_BIN_HEAD + k));
  else {
    const int kk = k - AUVP_BIN_HEAD;
    int chunk = duo_ld_i32(L.dir + ((kk >> 6) * L.k1 + b));
    chunk = chunk < 0 ? 0 : (chunk >= L.n_over ? L.n_over - 1 : chunk);
    m = duo_ld_i32(L.over + (chunk * 64 + (kk & 63)));
  }
  return m < 0 ? 0 : (m >= cap_nodes ? cap_nodes - 1 : m);
}

template <int J>
__global__ __launch_bounds__(DUO_EP * 128, 2) void rrt_duo_kernel(WorldDev W, RrtParamsDev P, RrtBuffers B, int n_episodes, int max_pts) {
  extern __shared__ __align__(16) unsigned char smem[];
  const RrtTables S = rrt_tables_view(smem, W.n_habitats, W.n_poly);
  const int wave = uni((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  const int n_ep_wg = (int)(blockDim.x >> 7);  // episodes of this workgroup: waves 2e (main) and 2e + 1 (helper)
  const int K = P.K;
  const int tables_b = (rrt_tables_bytes(W.n_habitats, W.n_poly, W.n_bins) + 15) & ~15;
  const int per_ep = duo_per_episode_bytes(K, max_pts);
  unsigned char* eb = smem + tables_b + (size_t)(wave >> 1) * per_ep;
  uint32_t* mt = reinterpret_cast<uint32_t*>(eb);
  eb += 624 * 4;
  double* u_win = reinterpret_cast<double*>(eb);
  eb += DUO_WIN * 8;
  int32_t* bin_count = reinterpret_cast<int32_t*>(eb);
  eb += (((K + 2) * 4) + 15) & ~15;
  DuoCtl* ctl = reinterpret_cast<DuoCtl*>(eb);
  eb += (sizeof(DuoCtl) + 15) & ~(size_t)15;
  DuoPacket* pk = reinterpret_cast<DuoPacket*>(eb);  // [2], 16-byte aligned stride
  constexpr int PK_STRIDE = (int)((sizeof(DuoPacket) + 15) & ~(size_t)15);
  eb += 2 * PK_STRIDE;
  double* inc = reinterpret_cast<double*>(eb);       // [4][CS]
  double* sc = inc + 4 * DUO_CS;                     // [CS][2]
  double* phi_l = sc + 2 * DUO_CS;                   // [CS]
  eb += 7 * DUO_CS * 8;
  double(*pts)[2] = reinterpret_cast<double(*)[2]>(eb);
  auto packet = [&](int i) -> DuoPacket* { return reinterpret_cast<DuoPacket*>(reinterpret_cast<unsigned char*>(pk) + (size_t)(i & 1) * PK_STRIDE); };

  rrt_tables_stage(S, W);
  if (threadIdx.x == 0) *S.params = P;
  const RrtParamsDev& Q = *S.params;
  double* olx = reinterpret_cast<double*>(smem + tables_b + (size_t)n_ep_wg * per_ep);
  double* oly = olx + J * 64;
  double* olt = oly + J * 64;
  float* olr = reinterpret_cast<float*>(olt + J * 64);
  for (int i = threadIdx.x; i < J * 64; i += blockDim.x) {
    const bool ok = i < W.n_obstacles;
    const double t = ok ? W.ot[i] : -1.0;
    olx[i] = ok ? W.ox[i] : 0.0;
    oly[i] = ok ? W.oy[i] : 0.0;
    olt[i] = t;
    const double rd = t >= 0.0 ? auvp_sqrt(t) * (1.0 + 0x1p-30) + 0x1p-40 : -__builtin_inf();
    float rf = (float)rd;
    if ((double)rf < rd) rf = __uint_as_float(__float_as_uint(rf) + 1u);
    olr[i] = rf;
  }
  const int ep = (int)blockIdx.x * n_ep_wg + (wave >> 1);
  const bool helper = (wave & 1) != 0;
  const bool valid_ep = ep < n_episodes;
  const size_t eps = (size_t)(valid_ep ? ep : 0);
  // ---- the episode's shared state, set up by its main wavefront before the workgroup barrier ----
  const int capn = B.cap_nodes, capp = B.cap_points, bcap = B.bin_cap;
  double* nodeF = B.node_f + eps * capn * 8;
  int4* nodeI = reinterpret_cast<int4*>(B.node_i) + eps * capn;
  uint8_t* nodeQ = B.node_q + eps * capn;
  double* ptF = B.points + eps * capp * 6;
  const BinLists bins = bin_lists(B, eps, K);
  const double* init = B.init + eps * 6;
  if (!helper) {
    for (int i = lane; i < K + 2; i += 64) bin_count[i] = 0;
    if (lane == 0) {
      ctl->ver = 0; ctl->last_bi = -1; ctl->last_c_before = 0; ctl->valid_seq = 0; ctl->done_seq = 0; ctl->redo_epoch = 0;
      ctl->stop = 0; ctl->abort = 0; ctl->helper_done = 0; ctl->final_after = 0.0; ctl->final_drawn = 0ull;
      packet(0)->tag = 0ull; packet(1)->tag = 0ull;
    }
    wave_sync();
    if (valid_ep && lane == 0) {
      nodeF[0] = init[0]; nodeF[1] = init[1]; nodeF[2] = init[2]; nodeF[3] = init[3]; nodeF[4] = init[5];
      nodeI[0] = make_int4(0, -1, 0, 0);
      nodeQ[0] = 0;
      bins.direct[(K >= 1 ? 1 : 0) * AUVP_BIN_HEAD] = 0;
      bin_count[K >= 1 ? 1 : 0] = 1;
    }
  } else {
    for (int i = lane; i < 624; i += 64) mt[i] = B.mt[eps * 624 + i];
  }
  __threadfence_block();
  __syncthreads();
  if (!valid_ep) return;  // (both wavefronts of the episode: no barrier after this point)

  // a bounded wait on an LDS word; false: gave up (the episode is abandoned, both wavefronts leave)
  auto give_up = [&]() { if (lane == 0) duo_poke(&ctl->abort, 1); };

  if (helper) {
    // ======================================================================================================= HELPER
    WaveRng rng;
    rng.s = mt;
    {
      int idx = B.mt_index ? uni(B.mt_index[ep]) : 624;
      idx = idx < 0 ? 0 : (idx > 624 ? 624 : idx);
      rng.pslot = idx == 624 ? 0u : (uint32_t)idx;
      rng.avail = (uint32_t)(624 - idx);
      rng.drawn = 0ull;
    }
    int epoch = 0;
    uint32_t sp_pslot = rng.pslot;             // stream position at the start of the latest packet
    unsigned long long sp_drawn = rng.drawn;
    // generate ahead, but never over the words since the latest packet's start: a redo must find them
    auto ensure = [&](uint32_t need) -> bool {
      while (rng.avail < need) {
        const unsigned long long held = rng.drawn - sp_drawn;
        if (held + rng.avail + 64ull > 624ull) return false;
        const uint32_t a0 = rng.avail;
        rng_ensure(rng, a0 + 1u);  // one block of up to 64 words
      }
      return true;
    };
    int i = 0;
#ifdef AUVP_DUO_DIAG
    unsigned long long diag_build = 0ull;  // EXPERIMENT ONLY: shader clocks spent building packets (waits excluded)
#endif
    for (;;) {
      // ---- wait: main accepts packet i - 1 (then packet i may be built), or wants it redone, or is done
      {
        int spins = 0;
        for (;;) {
          const int st = uni(duo_peek(&ctl->stop)), ab = uni(duo_peek(&ctl->abort));
          if (st || ab) goto helper_end;
          const int re = uni(duo_peek(&ctl->redo_epoch));
          if (re != epoch) {  // packet i - 1 is to be rebuilt from its first word
            epoch = re;
            rng.avail = (uint32_t)uni((int)(rng.avail + (uint32_t)(rng.drawn - sp_drawn)));
            rng.pslot = sp_pslot; rng.drawn = sp_drawn;
            i -= 1;
            break;
          }
          if (i < P.max_iter && uni(duo_peek(&ctl->valid_seq)) >= i) break;
          if (++spins > pipe_spin_limit()) { give_up(); goto helper_end; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      // ---------------------------------------------------------------- build packet i
#ifdef AUVP_DUO_DIAG
      const unsigned long long t_build0 = __builtin_amdgcn_s_memtime();
#endif
      sp_pslot = rng.pslot; sp_drawn = rng.drawn;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      int ver = uni(duo_peek(&ctl->ver));
      bool synced = false;  // main is known to wait for this packet: nothing can invalidate it any more
      int status = 0, rb = 0, cnt = 0, f = -1, rejects = 0;
      double u_me = 0.0;
      // ensure() refuses to generate over the packet's own first words (a selection that has consumed hundreds of words while
      // most bins are still empty).  Then: wait until main has finished the previous iteration; if it appended meanwhile the
      // build starts over (1); from here on main waits for this packet and nothing can invalidate it, so no rewind will be
      // asked for and the words behind are free (0).  -1: the episode is over.
      auto sync_with_main = [&]() -> int {
        if (!synced) {
          int spins = 0;
          while (uni(duo_peek(&ctl->done_seq)) < i) {
            if (uni(duo_peek(&ctl->abort)) || uni(duo_peek(&ctl->stop))) return -1;
            if (++spins > pipe_spin_limit()) { give_up(); return -1; }
            __builtin_amdgcn_s_sleep(1);
          }
          synced = true;
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          const int v2 = uni(duo_peek(&ctl->ver));
          if (v2 != ver) {
            rng.avail = (uint32_t)uni((int)(rng.avail + (uint32_t)(rng.drawn - sp_drawn)));
            rng.pslot = sp_pslot; rng.drawn = sp_drawn;
            ver = v2;
            return 1;
          }
        }
        sp_pslot = rng.pslot; sp_drawn = rng.drawn;
        return 0;
      };
      for (;;) {
        if (!ensure(128u)) {
          const int sr = sync_with_main();
          if (sr < 0) goto helper_end;
          if (sr > 0) { rejects = 0; }  // bins changed under the selection: it starts over from the packet's first word
          continue;
        }
        u_me = rng_random_at(rng, (uint32_t)lane);
        const int rbj = (int)py_uniform(1.0, (double)(K + 1), u_me);
        const bool cand = lane < 60;
        const bool badkey = cand && rbj > K;
        const int cj = (cand && !badkey) ? bin_count[rbj] : 0;
        const unsigned long long okm = wave_ballot(cj != 0), badm = wave_ballot(badkey);
        const int fo = okm ? (__ffsll((long long)okm) - 1) : 64, fb = badm ? (__ffsll((long long)badm) - 1) : 64;
        // candidates in front of the decision that were found empty: what an append could have changed
        const int first = fo < fb ? fo : fb;
        const unsigned long long before = first >= 64 ? ~0ull : ((1ull << first) - 1ull);
        if ((wave_ballot(cand && !badkey && cj == 0) & before) != 0ull) rejects = 1;
        if (fb < fo) { status = -5; break; }
        if (fo < 64) {
          f = fo;
          rb = __builtin_amdgcn_readlane(rbj, fo);
          cnt = __builtin_amdgcn_readlane(cj, fo);
          break;
        }
        rng_advance_words(rng, 120u);
      }
      DuoPacket* q = packet(i);
      int n_total = 0, par = 0;
      unsigned long long tmask = 0ull;
      if (status == 0) {
        const int ri = uni((int)py_uniform(0.0, (double)cnt, readlane_f64(u_me, f + 1)));
        const int par_v = duo_bin_member(bins, rb, ri, capn);
        int base = uni(f + 2);
        n_total = uni((int)auvp_floor(py_uniform(0.0, Q.freq, readlane_f64(u_me, base)) / 1));
        base += 1;
        const int n = n_total, nwin = 3 * n;
        // window entry j = random() number base + j of the stream (the first 64 numbers are tempered already)
        u_win[lane] = u_me;
        if (base + nwin > 64) {
          bool restart = false;
          while (!ensure((uint32_t)(2 * (base + nwin)))) {
            const int sr = sync_with_main();
            if (sr < 0) goto helper_end;
            if (sr > 0) { restart = true; break; }
          }
          if (restart) continue;  // (the whole packet: the outer loop comes back here with i unchanged)
          for (int jj = 64 + lane; jj < base + nwin; jj += 64) u_win[jj] = rng_random_at(rng, (uint32_t)jj);
        }
        wave_sync();
        const double* uw = u_win + base;
        unsigned long long msk[2] = {0ull, 0ull};
#pragma unroll
        for (int t = 0; t < 2; t++) {
          if (64 * t + 1 < nwin) {
            const int jj = lane + 64 * t;
            bool fl = false;
            if (jj + 1 < nwin) {
              const double dist = py_uniform(0.0, Q.dist_to_end, uw[jj]);
              const double diff = py_uniform(-Q.diff_max, Q.diff_max, uw[jj + 1]);
              fl = auvp_fabs(dist) > auvp_fabs(diff);
            }
            msk[t] = wave_ballot(fl);
          }
        }
        const bool active = lane < n;
        int cbelow = lane;
        unsigned long long win;
        {
          const int sh = 2 * lane;  // lane < 30: sh < 64
          const int s6 = sh & 63;
          const unsigned long long lo = sh < 64 ? msk[0] : msk[1], hi = sh < 64 ? msk[1] : 0ull;
          win = (lo >> s6) | ((hi << 1) << (63 - s6));
        }
        // (votes as ballots of ONE compare each, the lanes that take no part made neutral through their data: a vote on
        // `active && x` costs two more vector instructions on this chain -- a 0 / 1 and its compare with zero)
        {
          const uint32_t win32 = active ? (uint32_t)win : 0u;  // bits 0 .. lane are looked at (cbelow <= lane <= 29)
          const uint32_t below_me = active ? ((1u << lane) - 1u) : 0u;
          cbelow = active ? lane : 0;
          for (;;) {
            tmask = __builtin_amdgcn_uicmp((win32 >> cbelow) & 1u, 0u, 33 /* != */);
            const int cnew = __popc((uint32_t)tmask & below_me);
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
          const double dist = py_uniform(0.0, Q.dist_to_end, uw[mypos]);
          const double diff = py_uniform(-Q.diff_max, Q.diff_max, uw[mypos + 1]);
          const double s1 = dist + diff, s2 = dist - diff;
          radius = auvp_div_plain(s1 + s2, -s1 + s2);
          phi = auvp_div_plain(s1 + s2, 2 * radius);
          vt = py_uniform(0.0, 2 * Q.v, uw[mypos + 2]);
        }
        // the parent's record (main wrote it at least one finished iteration ago -- or the packet is redone)
        par = uni(par_v);
        const double* pr = nodeF + (size_t)par * 8;
        const double p0 = duo_ld_f64(pr), p1 = duo_ld_f64(pr + 1), p2 = duo_ld_f64(pr + 2), p3 = duo_ld_f64(pr + 3), p4 = duo_ld_f64(pr + 4);
        wave_sync();  // (u_win is read by every lane above)
        if (lane < DUO_CS) { q->radius[lane] = radius; q->phi[lane] = phi; q->vt[lane] = vt; }
        if (lane == 0) { q->cx = p0; q->cy = p1; q->cth = p2; q->ctt = p3; q->clen = p4; }
        rng_advance_words(rng, (uint32_t)(2 * (base + used)));
      }
      if (lane == 0) {
        q->ver = ver; q->status = status; q->rb = rb; q->rejects = rejects;
        q->n_total = n_total; q->par = par; q->tmask = tmask;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke64(&q->tag, duo_tag(epoch, i));
#ifdef AUVP_DUO_DIAG
      diag_build += __builtin_amdgcn_s_memtime() - t_build0;
#endif
      i++;  // (a KeyError packet: main stops at it, or asks for a redo first)
    }
  helper_end:
    {
      // the stream position the episode ends at: after the last packet main consumed (all of them when the budget ran out; an
      // episode that failed reports where the helper stood -- not part of the contract, tests/test_gpu_rows_kernel.py)
      const unsigned long long drawn = rng.drawn;
      rng_ensure(rng, 2u);
      const double after = rng_random_at(rng, 0u);
      if (lane == 0) { ctl->final_after = after; ctl->final_drawn = drawn; }
#ifdef AUVP_DUO_DIAG
      if (lane == 0) ctl->_pad = (int)(diag_build >> 8);
#endif
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke(&ctl->helper_done, 1);
    }
    return;
  }

  // =========================================================================================================== MAIN
  int next_chunk = 0;
  int n_nodes = 1, n_points = 0, status = 0, n_cand = 0, it = 0, my_epoch = 0;
#ifdef AUVP_DUO_DIAG
  unsigned long long diag_work = 0ull;
#endif
  for (; it < P.max_iter; it++) {
    // ---------------------------------------------------------------- the iteration's packet
    DuoPacket* q = packet(it);
    {
      int spins = 0;
      for (;;) {
        if (duo_peek64(&q->tag) == duo_tag(my_epoch, it)) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          // built one append ago?  Then only if that append did not touch what its selection looked at
          const int pv = uni(q->ver), vnow = n_nodes - 1;
          bool conflict = false;
          if (pv != vnow) {
            const int lb = uni(ctl->last_bi), lcb = uni(ctl->last_c_before);
            conflict = pv != vnow - 1 || (uni(q->status) == 0 && lb == uni(q->rb)) || (lcb == 0 && (uni(q->rejects) != 0 || uni(q->status) != 0));
          }
          if (!conflict) break;
          my_epoch++;
          if (lane == 0) duo_poke(&ctl->redo_epoch, my_epoch);
        }
        if (uni(duo_peek(&ctl->abort))) { status = AUVP_ST_PIPELINE; break; }
        if (++spins > pipe_spin_limit()) { give_up(); status = AUVP_ST_PIPELINE; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (uni(status)) break;
#ifdef AUVP_DUO_DIAG
    const unsigned long long t_work0 = __builtin_amdgcn_s_memtime();  // EXPERIMENT ONLY (tools/duo_probe.py)
#endif
    if (lane == 0) duo_poke(&ctl->valid_seq, it + 1);  // the helper may build the next packet now
    if (uni(q->status) != 0) { status = uni(q->status); break; }  // KeyError (:124)
    const int par = uni(q->par), n_total = uni(q->n_total);
    const unsigned long long tmask = q->tmask;
    double cx = readfirst_f64(q->cx), cy = readfirst_f64(q->cy), cth = readfirst_f64(q->cth), ctt = readfirst_f64(q->ctt), clen = readfirst_f64(q->clen);
    const double px0 = cx, py0 = cy, clen0 = clen;
    if (lane == 0) { pts[0][0] = cx; pts[0][1] = cy; }
    int cnt = 0;
    bool cap_err = false;
    if (n_total > 0) {
      // ---------------------------------------------------------------- steer, the half that needs the tree (:259-295)
      const int n = n_total;
      const bool active = lane < n;
      const bool taken = (tmask >> lane) & 1ull;
      double radius = 0.0, phi = 0.0, vt = 1.0;
      if (lane < DUO_CS) { radius = q->radius[lane]; phi = q->phi[lane]; vt = q->vt[lane]; }
      if (lane < DUO_CS) phi_l[lane] = phi;  // untaken / idle lanes add an exact 0.0
      wave_sync();
      if (lane == 0) {
        // eight entries per LDS round trip: four 16-byte reads up front, eight chained additions, four writes (entries past n
        // hold exact zeros; this wavefront has registers to spare and nothing else to do meanwhile)
        double th = cth;
        for (int s = 0; s < n; s += 8) {
          double2 v[4];
#pragma unroll
          for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<double2*>(phi_l + s + 2 * k);
#pragma unroll
          for (int k = 0; k < 4; k++) { th = th + v[k].x; v[k].x = th; th = th + v[k].y; v[k].y = th; }
#pragma unroll
          for (int k = 0; k < 4; k++) *reinterpret_cast<double2*>(phi_l + s + 2 * k) = v[k];
        }
      }
      wave_sync();
      const double myth = active ? phi_l[lane] : cth;
      double sn, cs;
      auvp_sincos_sk(myth, &sn, &cs);
      if (lane < DUO_CS) { sc[2 * lane] = sn; sc[2 * lane + 1] = cs; }
      wave_sync();
      double dx = 0.0, dy = 0.0, mv = 0.0, dt = 0.0;
      if (taken) {
        const unsigned long long below = tmask & ((1ull << lane) - 1ull);
        const int prev = below ? (63 - __clzll((long long)below)) : (DUO_CS - 1);  // lane 31 is idle: entry angle
        const double so = sc[2 * prev], co = sc[2 * prev + 1];
        dx = radius * (sn - so);
        dy = radius * (-cs + co);
        mv = auvp_sqrt_plain(dx * dx + dy * dy);
        dt = auvp_div_plain(mv, vt);
      }
      if (lane < DUO_CS) { inc[lane] = dx; inc[DUO_CS + lane] = dy; inc[2 * DUO_CS + lane] = dt; inc[3 * DUO_CS + lane] = mv; }
      wave_sync();
      if (lane < 4) {
        double acc = lane == 0 ? cx : (lane == 1 ? cy : (lane == 2 ? ctt : clen));
        double* row = inc + lane * DUO_CS;
        for (int s = 0; s < n; s += 8) {
          double2 v[4];
#pragma unroll
          for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<double2*>(row + s + 2 * k);
#pragma unroll
          for (int k = 0; k < 4; k++) { acc = acc + v[k].x; v[k].x = acc; acc = acc + v[k].y; v[k].y = acc; }
#pragma unroll
          for (int k = 0; k < 4; k++) *reinterpret_cast<double2*>(row + s + 2 * k) = v[k];
        }
      }
      wave_sync();
      double mx = 0.0, my = 0.0, mt_ = 0.0, ml = 0.0;
      if (active) { mx = inc[lane]; my = inc[DUO_CS + lane]; mt_ = inc[2 * DUO_CS + lane]; ml = inc[3 * DUO_CS + lane]; }
      const bool app = taken && (mv >= Q.min_dist);
      const unsigned long long amask = wave_ballot(app);
      const int napp = __popcll(amask);
      if (n_points + napp > capp || napp + 1 > max_pts) cap_err = true;
      if (!cap_err) {
        if (app) {
          const int rank = __popcll(amask & ((1ull << lane) - 1ull));
          const size_t gi = (size_t)(n_points + rank);  // speculative: committed only if the node is accepted
          double* ra = ptF + gi * 3;
          double* rbp = ptF + (size_t)capp * 3 + gi * 3;
          *reinterpret_cast<double2*>(ra) = make_double2(mx, my); ra[2] = mt_;
          *reinterpret_cast<double2*>(rbp) = make_double2(myth, vt); rbp[2] = ml;
          pts[rank + 1][0] = mx;
          pts[rank + 1][1] = my;
        }
        cnt = napp;
        cx = readlane_f64(mx, n - 1); cy = readlane_f64(my, n - 1);
        ctt = readlane_f64(mt_, n - 1); clen = readlane_f64(ml, n - 1);
        cth = readlane_f64(myth, n - 1);
      }
    }
    if (cap_err) { status = -2; break; }
    wave_sync();
    const int P_n = cnt + 1;
    // ---------------------------------------------------------------- check_collision (:530-549), as rrt_explore_kernel
    const double reach = clen - clen0;
    const double bx0 = px0 - reach, by0 = py0 - reach, bx1 = px0 + reach, by1 = py0 + reach;
    double cxm = px0, cym = py0;
    const double slack = 0x1p-30 * (auvp_fabs(bx0) + auvp_fabs(bx1) + auvp_fabs(by0) + auvp_fabs(by1) + 1.0);
    double hx = reach + slack, hy = reach + slack;
    int hit = 0;
    const bool pv0 = lane < P_n;
    double2 q0 = make_double2(0.0, 0.0);
    if (pv0) q0 = *reinterpret_cast<const double2*>(&pts[lane][0]);
    if (P.flags & AUVP_KFLAG_TIGHT_CULL) {
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
        const double ddx = q0.x - ox, ddy = q0.y - oy;
        const double d2 = ddx * ddx + ddy * ddy;
        hit |= (pv0 && d2 <= ot) ? 1 : 0;
      }
    }
    const double* sb = S.world->safe_box;
    const bool box_inside = W.has_safe_box && bx0 > sb[0] && by0 > sb[1] && bx1 < sb[2] && by1 < sb[3];
    const bool ok = !wave_any(hit != 0) && (box_inside || !any_point_outside(S.poly, W.n_poly, pts, P_n));
    if (ok) {
      if (n_nodes >= capn) { status = -2; break; }
      // ---------------------------------------------------------------- accept (:144-151)
      const int me = n_nodes;
      if (lane == 0) nodeI[me] = make_int4(it, par, n_points, cnt);
      double qf = auvp_floor(ctt * Q.inv_bin_interval);
      const double r = auvp_fma(-qf, Q.bin_interval, ctt);
      if (r < 0.0) qf -= 1.0;
      else if (r >= Q.bin_interval) qf += 1.0;
      const double fi = qf + 1.0;
      const double curr_bin = fi * Q.bin_interval;
      const bool over = curr_bin > Q.max_traj_time;
      int app_bi = -1, app_c = 0;
      if (!over || fi <= (double)K) {
        const int bi = uni((int)fi);
        const int c = over ? 0 : uni(bin_count[bi]);
        if (c >= bcap) { status = -2; break; }
        int32_t* slot = bin_slot_for_append(bins, bi, c, next_chunk, lane == 0);
        if (!slot) { status = -2; break; }
        if (lane == 0) *slot = me;
        app_bi = bi; app_c = over ? uni(bin_count[bi]) : c;  // (an overflowing key's list is reset: its old size is what the helper saw)
        if (lane == 0) bin_count[bi] = c + 1;
      }
      if (lane == 0) {
        double* nf = nodeF + (size_t)me * 8;
        *reinterpret_cast<double2*>(nf) = make_double2(cx, cy);
        *reinterpret_cast<double2*>(nf + 2) = make_double2(cth, ctt);
        nf[4] = clen;
        nodeQ[me] = ctt >= Q.max_traj_time - 30 ? 1 : 0;
      }
      n_nodes++;
      n_points += cnt;
      // the append is published: record, member list and bin size first, then what the helper's packets are checked against
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) { ctl->last_bi = app_bi; ctl->last_c_before = app_c; duo_poke(&ctl->ver, n_nodes - 1); }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) duo_poke(&ctl->done_seq, it + 1);
#ifdef AUVP_DUO_DIAG
    diag_work += __builtin_amdgcn_s_memtime() - t_work0;
#endif
  }
  // ---- the episode is over: the helper posts the stream position, main writes the record ----
  if (lane == 0) duo_poke(&ctl->stop, 1);
  {
    int spins = 0;
    while (!uni(duo_peek(&ctl->helper_done))) {
      if (++spins > pipe_spin_limit()) { status = AUVP_ST_PIPELINE; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // a stage that gave up (abort) may have posted its final words before this wavefront had finished: the episode's stream
    // position is then not to be trusted even though every step went through -- it is redone like any other failed episode
    if (status == 0 && uni(duo_peek(&ctl->abort))) status = AUVP_ST_PIPELINE;
  }
  for (int i = lane; i < K + 1; i += 64) B.bin_count[(size_t)ep * (K + 1) + i] = bin_count[i];
  if (lane == 0) {
    RrtSummary& s = B.summary[ep];
    pipe_report(B.pipe_fail, status);
    s.status = status; s.n_nodes = n_nodes; s.n_points = n_points; s.n_leaves = 0;
    s.best_leaf = -1; s.best_path_len = 0; s.iters_run = it; s.n_candidates = n_cand;
    s.best_cost[0] = __builtin_inf(); s.best_cost[1] = 0.0; s.best_cost[2] = 0.0; s.best_cost[3] = 0.0;
    s.best_length = 0.0;
    s.rng_after = ctl->final_after; s.leaf_elems = 0; s.n_draw32 = ctl->final_drawn; s.nn_scanned = 0ull;
#ifdef AUVP_DUO_DIAG
    s.n_candidates = (int)(diag_work >> 8);                       // main: clocks / 256 at work (packet waits excluded)
    s.nn_scanned = (unsigned long long)(uint32_t)ctl->_pad;       // helper: clocks / 256 building packets
#endif
  }
}

}  // namespace auvp
#endif
