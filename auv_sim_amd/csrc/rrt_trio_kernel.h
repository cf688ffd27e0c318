// rrt_trio_kernel.h -- RRT.exploring (path_planning/rrt_dubins.py:92-176), time-bin sampling, for the LATENCY runs: THREE
// wavefronts per episode, a three-stage pipeline over the iterations.
//
// rrt_duo_kernel split an iteration in two: a helper wavefront produces what depends only on the random stream one iteration
// ahead, the main wavefront does the rest -- and was the longer half (6 350 against 4 850 shader clocks per iteration, of which
// collision + append are 4 250).  Here the iteration has three owners, each on its own SIMD, each one iteration apart:
//   H  (stream)    owns the generator: selection draws, number of sub-arcs, window, draw-offset fixed point (which random()
//                  numbers belong to which sub-arc), the parent's record          -> packet k     (rrt_dubins.py:121-127,252-262)
//   M  (geometry)  radii / angles / speeds, theta chain, sin / cos, chords, running sums, path points
//                                                                                -> geo k        (:262-295)
//   T  (tree)      owns the tree: validates packet k against the appends made since it was built, collision and boundary
//                  tests, appends node k (time-bin insert, record, path points), keeps the counters   (:530-549, :144-151)
// H builds packet k+2 while M computes geo k+1 while T appends node k.  M and H therefore run SPECULATIVELY: when they start,
// the appends of up to three earlier iterations are still pending.  Selectable bins only grow, and a packet depends on the
// tree only through the bins its selection looked at and the record of the parent it picked, so T -- the one place where the
// appends happen, in order -- checks packet k against every append made after the snapshot it was built from (`ver`): the
// chosen bin itself (its size feeds `ri`; the parent could be the node just appended), or -- when an append made a bin
// non-empty -- a selection that had found a bin empty.  On a conflict T starts a new epoch at iteration k: H rewinds the
// generator to the first word of packet k, M drops what it has, both redo from k.  Conflicts cost about one iteration and
// happen in 2-4 % of the iterations once the bins are filled.
//
// The generator lives in a ring of 1 248 words (two states): a word, once generated, stays valid, so a rewind only moves the
// read position back; H keeps the distance between the oldest position it may have to return to (the first word of the oldest
// unfinished iteration) and the generation frontier below the ring's size -- a selection that cannot (hundreds of empty-bin
// draws at the start of an episode) waits until T has caught up, after which nothing can invalidate it.
//
// Every hand-over is one 64-bit LDS word {epoch, iteration} written last; every wait is bounded.  Results are bit-identical to
// rrt_explore_kernel's and the checker's, stream position included (tests/test_gpu_duo_kernel.py).
//
// Limits as rrt_duo_kernel: time-bin mode, no diagnostics, freq <= 30, <= 256 obstacles, small batches.
#ifndef AUVP_RRT_TRIO_KERNEL_H
#define AUVP_RRT_TRIO_KERNEL_H
#include "rrt_duo_kernel.h"

namespace auvp {

constexpr int TRIO_EP = 4;          // episodes per workgroup at most (twelve wavefronts)
constexpr int TRIO_RING = 4;        // packet / geo slots per episode (iterations in flight)
constexpr int TRIO_GEN = 1248;      // generator ring, words
constexpr int TRIO_HIST = 8;        // appends T remembers (bin, size before): more than can be pending

struct TrioPacket {  // H -> M (and T: ver, rb, rejects, status)
  unsigned long long tag;    // the packet is complete (what M waits for)
  unsigned long long tag_s;  // four-wavefront form: the stream part is complete, the parent lookup (L) is still to come
  int ver, status, rb, rejects, n_total, par;
  int ri, _p0;
  unsigned long long tmask;
  double cx, cy, cth, ctt, clen;  // the parent's record
  double ud[DUO_CS], uf[DUO_CS], uv[DUO_CS];  // the taken sub-arcs' random() numbers: dist, diff, v (M turns them into radius, phi, v)
};

struct TrioGeo {  // M -> T
  unsigned long long tag;
  int status, cnt, par, _p0;
  double cx, cy, cth, ctt, clen;      // the new node's state
  double px0, py0, clen0;             // the parent's end: first point of the path, centre of the collision cull
  double px[DUO_CS], py[DUO_CS], pt[DUO_CS], pth[DUO_CS], pv[DUO_CS], pl[DUO_CS];  // its path points, in order
};

struct TrioCtl {  // (the first eight words are what the waits look at: read as two 16-byte pieces, one LDS round trip)
  int ver;            // appends so far (T)
  int done_seq;       // T has finished iterations < done_seq
  int epoch;          // bumped by T: everything from iteration restart_k on is redone
  int restart_k;
  int stop, abort, h_done, m_done;
  int l_done, _pc0, _pc1, _pc2;
  double final_after;
  unsigned long long final_drawn;
  int hist_bin[TRIO_HIST], hist_cb[TRIO_HIST];  // append number a (1-based) -> bin, size before; slot a & 7
  // H's own: stream position at the first word of the packets in flight (slot k & 3).  In LDS because the compiler turns a
  // register array under a computed index into a scratch array: 32 B per lane and a scratch read on H's chain per packet
  unsigned long long sp_drawn[TRIO_RING];
  uint32_t sp_cslot[TRIO_RING];
};

__host__ __device__ inline int trio_per_episode_bytes(int K) {
  int b = TRIO_GEN * 4;
  b += DUO_WIN * 8;
  b += (((K + 2) * 4) + 15) & ~15;
  b += (int)((sizeof(TrioCtl) + 15) & ~(size_t)15);
  b += TRIO_RING * (int)((sizeof(TrioPacket) + 15) & ~(size_t)15);
  b += TRIO_RING * (int)((sizeof(TrioGeo) + 15) & ~(size_t)15);
  b += 7 * DUO_CS * 8;      // M: inc[4][CS], sc[2][CS], phi_l[CS]
  b += (DUO_CS + 2) * 16;   // T: path points x, y (boundary test)
  return b;
}
__host__ __device__ inline int trio_lds_bytes(int K, int n_obst_slots, int tables_bytes, int episodes) {
  return ((tables_bytes + 15) & ~15) + episodes * trio_per_episode_bytes(K) + n_obst_slots * (8 + 8 + 8 + 4);
}

// CPython's MT19937 stream in a ring of TRIO_GEN words.  gslot / cslot: ring slots of the next word to generate / to consume;
// avail = generated, not consumed; drawn = words consumed since the start (the stream position).
struct RingRng {
  uint32_t* s;
  uint32_t gslot, cslot, avail;
  unsigned long long drawn;
};
__device__ __forceinline__ uint32_t ring_wrap(uint32_t k) { return k >= (uint32_t)TRIO_GEN ? k - (uint32_t)TRIO_GEN : k; }
// one block of 64 words: x[q] = x[q-227] ^ twist(x[q-624], x[q-623]); reads and writes never meet (different slots)
__device__ __forceinline__ void ring_generate64(RingRng& r) {
  const uint32_t l = (uint32_t)lane_id();
  const uint32_t q = ring_wrap(r.gslot + l);
  const uint32_t a = r.s[ring_wrap(q + (uint32_t)(TRIO_GEN - 624))], b = r.s[ring_wrap(q + (uint32_t)(TRIO_GEN - 623))];
  const uint32_t c = r.s[ring_wrap(q + (uint32_t)(TRIO_GEN - 227))];
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  r.s[q] = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  wave_sync();
  r.gslot = (uint32_t)uni((int)ring_wrap(r.gslot + 64u));
  r.avail = (uint32_t)uni((int)(r.avail + 64u));
}
__device__ __forceinline__ uint32_t ring_word(const RingRng& r, uint32_t j) { return mt_temper(r.s[ring_wrap(r.cslot + j)]); }
__device__ __forceinline__ double ring_random_at(const RingRng& r, uint32_t j) {
  const uint32_t a = ring_word(r, 2u * j) >> 5, b = ring_word(r, 2u * j + 1u) >> 6;
  return py_random_from(a, b);
}
__device__ __forceinline__ void ring_advance(RingRng& r, uint32_t n) {
  r.cslot = (uint32_t)uni((int)ring_wrap(r.cslot + n));
  r.avail = (uint32_t)uni((int)(r.avail - n));
  r.drawn += n;
}

// the control words of an episode as the waits see them: one look = one LDS round trip (a lone wavefront pays ~100 clocks per
// dependent LDS access, and a wait that looked at stop, abort, epoch and done_seq one after the other cost ~400 per iteration)
struct TrioView { int ver, done_seq, epoch, restart_k, stop, abort; };
__device__ __forceinline__ TrioView trio_look(const TrioCtl* c) {
  __asm__ volatile("" ::: "memory");
  const int4 a = *reinterpret_cast<const int4*>(&c->ver);
  const int2 b = *reinterpret_cast<const int2*>(&c->stop);
  __asm__ volatile("" ::: "memory");
  TrioView v;
  v.ver = uni(a.x); v.done_seq = uni(a.y); v.epoch = uni(a.z); v.restart_k = uni(a.w); v.stop = uni(b.x); v.abort = uni(b.y);
  return v;
}

// NW = 3: H, M, T.  NW = 4: the parent lookup -- the member's id out of the bin's list, then its record: two dependent global
// reads, ~1 200 clocks of H's ~5 000 that nothing in H covers -- is a wavefront of its own (L) between H and M.
template <int J, int NW>
__global__ __launch_bounds__(TRIO_EP * 64 * NW, 1) void rrt_trio_kernel(WorldDev W, RrtParamsDev P, RrtBuffers B, int n_episodes) {
  extern __shared__ __align__(16) unsigned char smem[];
  const RrtTables S = rrt_tables_view(smem, W.n_habitats, W.n_poly);
  const int wave = uni((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  const int n_ep_wg = (int)(blockDim.x / (64 * NW));
  const int eidx = wave / NW, role = wave - NW * eidx;  // role 0: M (geometry), 1: H (stream), 2: T (tree), 3: L (parent lookup)
  const int K = P.K;
  const int tables_b = (rrt_tables_bytes(W.n_habitats, W.n_poly, W.n_bins) + 15) & ~15;
  const int per_ep = trio_per_episode_bytes(K);
  unsigned char* eb = smem + tables_b + (size_t)eidx * per_ep;
  uint32_t* gen = reinterpret_cast<uint32_t*>(eb);
  eb += TRIO_GEN * 4;
  double* u_win = reinterpret_cast<double*>(eb);
  eb += DUO_WIN * 8;
  int32_t* bin_count = reinterpret_cast<int32_t*>(eb);
  eb += (((K + 2) * 4) + 15) & ~15;
  TrioCtl* ctl = reinterpret_cast<TrioCtl*>(eb);
  eb += (sizeof(TrioCtl) + 15) & ~(size_t)15;
  constexpr int PK_STRIDE = (int)((sizeof(TrioPacket) + 15) & ~(size_t)15);
  constexpr int GEO_STRIDE = (int)((sizeof(TrioGeo) + 15) & ~(size_t)15);
  unsigned char* pk_base = eb;
  eb += TRIO_RING * PK_STRIDE;
  unsigned char* geo_base = eb;
  eb += TRIO_RING * GEO_STRIDE;
  double* inc = reinterpret_cast<double*>(eb);
  double* sc = inc + 4 * DUO_CS;
  double* phi_l = sc + 2 * DUO_CS;
  eb += 7 * DUO_CS * 8;
  double(*pts)[2] = reinterpret_cast<double(*)[2]>(eb);
  auto packet = [&](int k) -> TrioPacket* { return reinterpret_cast<TrioPacket*>(pk_base + (size_t)(k & (TRIO_RING - 1)) * PK_STRIDE); };
  auto geo = [&](int k) -> TrioGeo* { return reinterpret_cast<TrioGeo*>(geo_base + (size_t)(k & (TRIO_RING - 1)) * GEO_STRIDE); };

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
  const int ep = (int)blockIdx.x * n_ep_wg + eidx;
  const bool valid_ep = ep < n_episodes;
  const size_t eps = (size_t)(valid_ep ? ep : 0);
  const int capn = B.cap_nodes, capp = B.cap_points, bcap = B.bin_cap;
  double* nodeF = B.node_f + eps * capn * 8;
  int4* nodeI = reinterpret_cast<int4*>(B.node_i) + eps * capn;
  uint8_t* nodeQ = B.node_q + eps * capn;
  double* ptF = B.points + eps * capp * 6;
  const BinLists bins = bin_lists(B, eps, K);
  const double* init = B.init + eps * 6;
  if (role == 2) {
    for (int i = lane; i < K + 2; i += 64) bin_count[i] = 0;
    if (lane == 0) {
      ctl->ver = 0; ctl->done_seq = 0; ctl->epoch = 0; ctl->restart_k = 0; ctl->stop = 0; ctl->abort = 0; ctl->h_done = 0; ctl->m_done = 0;
      ctl->l_done = NW == 4 ? 0 : 1;
      ctl->final_after = 0.0; ctl->final_drawn = 0ull;
      for (int k = 0; k < TRIO_RING; k++) { packet(k)->tag = 0ull; packet(k)->tag_s = 0ull; geo(k)->tag = 0ull; }
      for (int k = 0; k < TRIO_HIST; k++) { ctl->hist_bin[k] = -1; ctl->hist_cb[k] = 1; }
    }
    wave_sync();
    if (valid_ep && lane == 0) {
      nodeF[0] = init[0]; nodeF[1] = init[1]; nodeF[2] = init[2]; nodeF[3] = init[3]; nodeF[4] = init[5];
      nodeI[0] = make_int4(0, -1, 0, 0);
      nodeQ[0] = 0;
      bins.direct[(K >= 1 ? 1 : 0) * AUVP_BIN_HEAD] = 0;
      bin_count[K >= 1 ? 1 : 0] = 1;
    }
  } else if (role == 1) {
    for (int i = lane; i < 624; i += 64) gen[i] = B.mt[eps * 624 + i];
  }
  __threadfence_block();
  __syncthreads();
  if (!valid_ep) return;  // (all three wavefronts of the episode: no barrier after this point)
  auto give_up = [&]() { if (lane == 0) duo_poke(&ctl->abort, 1); };

  if (role == 1) {
    // ===================================================================================================== H: the stream
    RingRng rng;
    rng.s = gen;
    {
      int idx = B.mt_index ? uni(B.mt_index[ep]) : 624;
      idx = idx < 0 ? 0 : (idx > 624 ? 624 : idx);
      rng.gslot = 624u; rng.cslot = (uint32_t)idx; rng.avail = (uint32_t)(624 - idx); rng.drawn = 0ull;
    }
    int epoch = 0, k = 0;
#ifdef AUVP_DUO_DIAG
    unsigned long long diag_h = 0ull;
#endif
    // stream position at the first word of the packets in flight (slot k & 3): where a new epoch rewinds to
    if (lane == 0)
      for (int q = 0; q < TRIO_RING; q++) { ctl->sp_cslot[q] = rng.cslot; ctl->sp_drawn[q] = 0ull; }
    wave_sync();
    unsigned long long floor_drawn = 0ull;  // stream position H may still have to return to (first word of the oldest unfinished iteration)
    auto sp_get = [&](int kk, uint32_t& cs, unsigned long long& dr) {
      const int q = kk & (TRIO_RING - 1);
      cs = (uint32_t)uni((int)ctl->sp_cslot[q]);
      const unsigned long long d = ctl->sp_drawn[q];
      dr = ((unsigned long long)(uint32_t)uni((int)(d >> 32)) << 32) | (uint32_t)uni((int)(d & 0xffffffffull));
    };
    auto sp_set = [&](int kk, uint32_t cs, unsigned long long dr) {
      const int q = kk & (TRIO_RING - 1);
      if (lane == 0) { ctl->sp_cslot[q] = cs; ctl->sp_drawn[q] = dr; }
      wave_sync();
    };
    auto rewind_to = [&](uint32_t cs, unsigned long long dr) {
      rng.avail = (uint32_t)uni((int)(rng.avail + (uint32_t)(rng.drawn - dr)));
      rng.cslot = cs; rng.drawn = dr;
    };
    // generate ahead, but keep everything since floor_drawn in the ring
    auto ensure = [&](uint32_t need) -> bool {
      while (rng.avail < need) {
        if ((rng.drawn - floor_drawn) + rng.avail + 64ull > (unsigned long long)TRIO_GEN) return false;
        ring_generate64(rng);
      }
      return true;
    };
    for (;;) {
      // ---- wait for a free slot (T has finished iteration k - TRIO_RING), a new epoch, or the end
      int done = 0, snap_ver = 0;
      {
        int spins = 0;
        for (;;) {
          TrioView cv = trio_look(ctl);
          if (cv.stop || cv.abort) goto h_end;
          if (cv.epoch != epoch) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            cv = trio_look(ctl);  // (restart_k is written before the epoch)
            epoch = cv.epoch;
            k = cv.restart_k;
            uint32_t cs; unsigned long long dr;
            sp_get(k, cs, dr);
            rewind_to(cs, dr);
          }
          done = cv.done_seq;
          snap_ver = cv.ver;
          if (k < P.max_iter && done >= k - (TRIO_RING - 1)) break;
          if (++spins > pipe_spin_limit()) { give_up(); goto h_end; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      // the oldest iteration that can still be redone is `done` (T validates in order): its first word is the floor
      {
        uint32_t cs; unsigned long long dr;
        const int oldest = done < k ? done : k;
        sp_get(oldest, cs, dr);
        floor_drawn = oldest == k ? rng.drawn : dr;
      }
      // ---------------------------------------------------------------- build packet k
#ifdef AUVP_DUO_DIAG
      const unsigned long long t_b0 = __builtin_amdgcn_s_memtime();  // EXPERIMENT ONLY (tools/duo_probe.py)
#endif
      sp_set(k, rng.cslot, rng.drawn);
      // (the snapshot of `ver` is the one the wait just took -- BEFORE any look at the bins: an append after it is checked by T)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      int ver = snap_ver;
      bool synced = false;
      int status = 0, rb = 0, cnt = 0, f = -1, rejects = 0;
      double u_me = 0.0;
      bool again = false;
      // the ring cannot hold everything since the floor: wait until T has caught up with k (nothing pending, so nothing can
      // invalidate this packet any more); 1: the bins changed meanwhile, the build starts over; -1: the episode is over
      auto sync_with_tree = [&]() -> int {
        if (!synced) {
          int spins = 0;
          for (;;) {
            if (uni(duo_peek(&ctl->abort)) || uni(duo_peek(&ctl->stop))) return -1;
            if (uni(duo_peek(&ctl->epoch)) != epoch) return 2;  // a new epoch: back to the top
            if (uni(duo_peek(&ctl->done_seq)) >= k) break;
            if (++spins > pipe_spin_limit()) { give_up(); return -1; }
            __builtin_amdgcn_s_sleep(1);
          }
          synced = true;
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          const int v2 = uni(duo_peek(&ctl->ver));
          if (v2 != ver) {
            uint32_t cs; unsigned long long dr;
            sp_get(k, cs, dr);
            rewind_to(cs, dr);
            ver = v2;
            floor_drawn = rng.drawn;
            return 1;
          }
        }
        floor_drawn = rng.drawn;  // (synced: no rewind will be asked for)
        return 0;
      };
      for (;;) {
        if (!ensure(128u)) {
          const int sr = sync_with_tree();
          if (sr < 0) goto h_end;
          if (sr == 2) { again = true; break; }
          if (sr == 1) rejects = 0;
          continue;
        }
        u_me = ring_random_at(rng, (uint32_t)lane);
        const int rbj = (int)py_uniform(1.0, (double)(K + 1), u_me);
        const bool cand = lane < 60;
        const bool badkey = cand && rbj > K;
        const int cj = (cand && !badkey) ? bin_count[rbj] : 0;
        const unsigned long long okm = wave_ballot(cj != 0), badm = wave_ballot(badkey);
        const int fo = okm ? (__ffsll((long long)okm) - 1) : 64, fb = badm ? (__ffsll((long long)badm) - 1) : 64;
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
        ring_advance(rng, 120u);
      }
      if (again) continue;
      TrioPacket* q = packet(k);
      int n_total = 0, par = 0, ri_out = 0;
      unsigned long long tmask = 0ull;
      if (status == 0) {
        const int ri = uni((int)py_uniform(0.0, (double)cnt, readlane_f64(u_me, f + 1)));
        ri_out = ri;
        int par_v = 0;
        if (NW == 3) par_v = duo_bin_member(bins, rb, ri, capn);
        int base = uni(f + 2);
        n_total = uni((int)auvp_floor(py_uniform(0.0, Q.freq, readlane_f64(u_me, base)) / 1));
        base += 1;
        const int n = n_total, nwin = 3 * n;
        u_win[lane] = u_me;
        if (base + nwin > 64) {
          int sr = 0;
          while (!ensure((uint32_t)(2 * (base + nwin)))) {
            sr = sync_with_tree();
            if (sr != 0) break;
          }
          if (sr < 0) goto h_end;
          if (sr != 0) continue;  // the whole packet again (a new epoch: from the top of the loop)
          for (int jj = 64 + lane; jj < base + nwin; jj += 64) u_win[jj] = ring_random_at(rng, (uint32_t)jj);
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
        // the parent's record is requested here, between the predicate and the fixed point: its id (requested before the window)
        // has had the predicate's time to arrive, and the record has the fixed point's
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0;
        if (NW == 3) {
          par = uni(par_v);
          const double* pr = nodeF + (size_t)par * 8;
          p0 = duo_ld_f64(pr); p1 = duo_ld_f64(pr + 1); p2 = duo_ld_f64(pr + 2); p3 = duo_ld_f64(pr + 3); p4 = duo_ld_f64(pr + 4);
        }
        const bool active = lane < n;
        int cbelow = lane;
        unsigned long long win;
        {
          const int sh = 2 * lane, s6 = sh & 63;
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
        double ud = 0.0, uf = 0.0, uv = 0.0;
        if (taken) { ud = uw[mypos]; uf = uw[mypos + 1]; uv = uw[mypos + 2]; }
        wave_sync();  // (u_win is read by every lane above)
        if (lane < DUO_CS) { q->ud[lane] = ud; q->uf[lane] = uf; q->uv[lane] = uv; }
        if (NW == 3 && lane == 0) { q->cx = p0; q->cy = p1; q->cth = p2; q->ctt = p3; q->clen = p4; }
        ring_advance(rng, (uint32_t)(2 * (base + used)));
      }
      if (lane == 0) {
        q->ver = ver; q->status = status; q->rb = rb; q->rejects = rejects;
        q->n_total = n_total; q->tmask = tmask; q->ri = ri_out;
        if (NW == 3) q->par = par;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke64(NW == 3 ? &q->tag : &q->tag_s, duo_tag(epoch, k));
#ifdef AUVP_DUO_DIAG
      diag_h += __builtin_amdgcn_s_memtime() - t_b0;
#endif
      k++;
    }
  h_end:
    {
      // the stream position the episode ends at: after packet max_iter - 1 when the budget ran out (H never builds beyond it;
      // a failed episode reports where H stood: not part of the contract)
      const unsigned long long drawn = rng.drawn;
      floor_drawn = rng.drawn;
      (void)ensure(2u);
      const double after = ring_random_at(rng, 0u);
      if (lane == 0) { ctl->final_after = after; ctl->final_drawn = drawn; }
#ifdef AUVP_DUO_DIAG
      if (lane == 0) ctl->hist_bin[0] = (int)(diag_h >> 8);
#endif
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke(&ctl->h_done, 1);
    }
    return;
  }

  if (NW == 4 && role == 3) {
    // ================================================================================================ L: the parent lookup
    int epoch = 0, k = 0;
    for (;;) {
      TrioPacket* q = nullptr;
      {
        int spins = 0;
        for (;;) {
          q = packet(k);
          const unsigned long long tg = duo_peek64(&q->tag_s);
          TrioView cv = trio_look(ctl);
          if (cv.stop || cv.abort) goto l_end;
          if (cv.epoch != epoch) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            cv = trio_look(ctl);
            epoch = cv.epoch;
            k = cv.restart_k;
            continue;
          }
          if (k < P.max_iter && tg == duo_tag(epoch, k)) break;
          if (++spins > pipe_spin_limit()) { give_up(); goto l_end; }
          __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      if (uni(q->status) == 0) {
        // the ri-th member of bin rb and its record (T wrote both before the `ver` the packet was built from -- or T redoes the
        // packet: a read that raced with a later append is clamped into the episode's storage and thrown away)
        const int par = uni(duo_bin_member(bins, uni(q->rb), uni(q->ri), capn));
        const double* pr = nodeF + (size_t)par * 8;
        const double p0 = duo_ld_f64(pr), p1 = duo_ld_f64(pr + 1), p2 = duo_ld_f64(pr + 2), p3 = duo_ld_f64(pr + 3), p4 = duo_ld_f64(pr + 4);
        if (lane == 0) { q->par = par; q->cx = p0; q->cy = p1; q->cth = p2; q->ctt = p3; q->clen = p4; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke64(&q->tag, duo_tag(epoch, k));
      k++;
    }
  l_end:
    if (lane == 0) duo_poke(&ctl->l_done, 1);
    return;
  }

  if (role == 0) {
    // =================================================================================================== M: the geometry
    int epoch = 0, k = 0;
#ifdef AUVP_DUO_DIAG
    unsigned long long diag_m = 0ull;
#endif
    for (;;) {
      TrioPacket* q = nullptr;
      {
        int spins = 0;
        for (;;) {
          q = packet(k);
          const unsigned long long tg = duo_peek64(&q->tag);  // (issued with the control words: one round trip)
          TrioView cv = trio_look(ctl);
          if (cv.stop || cv.abort) goto m_end;
          if (cv.epoch != epoch) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            cv = trio_look(ctl);
            epoch = cv.epoch;
            k = cv.restart_k;
            continue;
          }
          if (k < P.max_iter && tg == duo_tag(epoch, k)) break;
          if (++spins > pipe_spin_limit()) { give_up(); goto m_end; }
          __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
#ifdef AUVP_DUO_DIAG
      const unsigned long long t_m0 = __builtin_amdgcn_s_memtime();
#endif
      TrioGeo* g = geo(k);  // (free: H built packet k only after T finished iteration k - TRIO_RING)
      const int pstatus = uni(q->status);
      const int par = uni(q->par), n_total = pstatus == 0 ? uni(q->n_total) : 0;
      const unsigned long long tmask = q->tmask;
      double cx = readfirst_f64(q->cx), cy = readfirst_f64(q->cy), cth = readfirst_f64(q->cth), ctt = readfirst_f64(q->ctt), clen = readfirst_f64(q->clen);
      const double px0 = cx, py0 = cy, clen0 = clen;
      int cnt = 0;
      if (n_total > 0) {
        // ---------------------------------------------------------------- steer, the half that needs the parent (:259-295)
        const int n = n_total;
        const bool active = lane < n;
        const bool taken = (tmask >> lane) & 1ull;
        double radius = 0.0, phi = 0.0, vt = 1.0;
        if (taken) {
          const double dist = py_uniform(0.0, Q.dist_to_end, q->ud[lane]);
          const double diff = py_uniform(-Q.diff_max, Q.diff_max, q->uf[lane]);
          const double s1 = dist + diff, s2 = dist - diff;
          radius = auvp_div_plain(s1 + s2, -s1 + s2);
          phi = auvp_div_plain(s1 + s2, 2 * radius);
          vt = py_uniform(0.0, 2 * Q.v, q->uv[lane]);
        }
        if (lane < DUO_CS) phi_l[lane] = phi;
        wave_sync();
        if (lane == 0) {
          double th = cth;
          for (int s = 0; s < n; s += 8) {
            double2 v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] = *reinterpret_cast<double2*>(phi_l + s + 2 * c);
#pragma unroll
            for (int c = 0; c < 4; c++) { th = th + v[c].x; v[c].x = th; th = th + v[c].y; v[c].y = th; }
#pragma unroll
            for (int c = 0; c < 4; c++) *reinterpret_cast<double2*>(phi_l + s + 2 * c) = v[c];
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
          const int prev = below ? (63 - __clzll((long long)below)) : (DUO_CS - 1);
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
            for (int c = 0; c < 4; c++) v[c] = *reinterpret_cast<double2*>(row + s + 2 * c);
#pragma unroll
            for (int c = 0; c < 4; c++) { acc = acc + v[c].x; v[c].x = acc; acc = acc + v[c].y; v[c].y = acc; }
#pragma unroll
            for (int c = 0; c < 4; c++) *reinterpret_cast<double2*>(row + s + 2 * c) = v[c];
          }
        }
        wave_sync();
        double mx = 0.0, my = 0.0, mt_ = 0.0, ml = 0.0;
        if (active) { mx = inc[lane]; my = inc[DUO_CS + lane]; mt_ = inc[2 * DUO_CS + lane]; ml = inc[3 * DUO_CS + lane]; }
        const bool app = taken && (mv >= Q.min_dist);
        const unsigned long long amask = wave_ballot(app);
        cnt = __popcll(amask);
        if (app) {
          const int rank = __popcll(amask & ((1ull << lane) - 1ull));
          g->px[rank] = mx; g->py[rank] = my; g->pt[rank] = mt_; g->pth[rank] = myth; g->pv[rank] = vt; g->pl[rank] = ml;
        }
        cx = readlane_f64(mx, n - 1); cy = readlane_f64(my, n - 1);
        ctt = readlane_f64(mt_, n - 1); clen = readlane_f64(ml, n - 1);
        cth = readlane_f64(myth, n - 1);
      }
      if (lane == 0) {
        g->status = pstatus; g->cnt = cnt; g->par = par;
        g->cx = cx; g->cy = cy; g->cth = cth; g->ctt = ctt; g->clen = clen;
        g->px0 = px0; g->py0 = py0; g->clen0 = clen0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke64(&g->tag, duo_tag(epoch, k));
#ifdef AUVP_DUO_DIAG
      diag_m += __builtin_amdgcn_s_memtime() - t_m0;
#endif
      k++;
    }
  m_end:
#ifdef AUVP_DUO_DIAG
    if (lane == 0) ctl->hist_bin[1] = (int)(diag_m >> 8);
#endif
    if (lane == 0) duo_poke(&ctl->m_done, 1);
    return;
  }

  // ======================================================================================================= T: the tree
  int next_chunk = 0;
  int n_nodes = 1, n_points = 0, status = 0, n_cand = 0, it = 0, epoch = 0;
#ifdef AUVP_DUO_DIAG
  unsigned long long diag_t = 0ull;
#endif
  for (; it < P.max_iter; it++) {
    TrioGeo* g = geo(it);
    TrioPacket* q = packet(it);
    bool got = false;
    for (;;) {
      {
        int spins = 0;
        for (;;) {
          if (duo_peek64(&g->tag) == duo_tag(epoch, it)) break;
          if (uni(duo_peek(&ctl->abort))) { status = AUVP_ST_PIPELINE; break; }
          if (++spins > pipe_spin_limit()) { give_up(); status = AUVP_ST_PIPELINE; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (uni(status)) break;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      // ---- the packet this geometry came from: still what the tree would give?  Every append made after its snapshot
      const int pv = uni(q->ver), vnow = n_nodes - 1;
      bool conflict = false;
      if (pv != vnow) {
        const int prb = uni(q->rb), pst = uni(q->status), prj = uni(q->rejects);
        if (vnow - pv > TRIO_HIST || pv > vnow) conflict = true;
        for (int a = pv + 1; a <= vnow && !conflict; a++) {
          const int hb = uni(ctl->hist_bin[a & (TRIO_HIST - 1)]), hc = uni(ctl->hist_cb[a & (TRIO_HIST - 1)]);
          conflict = (pst == 0 && hb == prb) || (hc == 0 && (prj != 0 || pst != 0));
        }
      }
      if (!conflict) { got = true; break; }
      // a new epoch from this iteration on: H rewinds to the packet's first word, M drops what it has
      epoch++;
      if (lane == 0) ctl->restart_k = it;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke(&ctl->epoch, epoch);
    }
    if (!got || uni(status)) break;
#ifdef AUVP_DUO_DIAG
    const unsigned long long t_t0 = __builtin_amdgcn_s_memtime();
#endif
    if (uni(g->status) != 0) { status = uni(g->status); break; }  // KeyError (:124)
    const int cnt = uni(g->cnt);
    if (n_points + cnt > capp) { status = -2; break; }
    const double cx = readfirst_f64(g->cx), cy = readfirst_f64(g->cy), cth = readfirst_f64(g->cth), ctt = readfirst_f64(g->ctt), clen = readfirst_f64(g->clen);
    // ---------------------------------------------------------------- check_collision (:530-549), as rrt_explore_kernel
    bool rejected;
    {
      const double px0 = readfirst_f64(g->px0), py0 = readfirst_f64(g->py0), clen0 = readfirst_f64(g->clen0);
      const int P_n = cnt + 1;
      const double reach = clen - clen0;
      const double bx0 = px0 - reach, by0 = py0 - reach, bx1 = px0 + reach, by1 = py0 + reach;
      double cxm = px0, cym = py0;
      const double slack = 0x1p-30 * (auvp_fabs(bx0) + auvp_fabs(bx1) + auvp_fabs(by0) + auvp_fabs(by1) + 1.0);
      double hx = reach + slack, hy = reach + slack;
      int hit = 0;
      const bool pv0 = lane < P_n;
      double2 q0 = make_double2(px0, py0);  // lane 0: the parent's end; lane p: path point p - 1
      if (pv0 && lane > 0) q0 = make_double2(g->px[lane - 1], g->py[lane - 1]);
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
          hit |= (pv0 && ddx * ddx + ddy * ddy <= ot) ? 1 : 0;
        }
      }
      const double* sb = S.world->safe_box;
      const bool box_inside = W.has_safe_box && bx0 > sb[0] && by0 > sb[1] && bx1 < sb[2] && by1 < sb[3];
      rejected = wave_any(hit != 0);
      if (!rejected && !box_inside) {
        if (pv0) { pts[lane][0] = q0.x; pts[lane][1] = q0.y; }
        wave_sync();
        rejected = any_point_outside(S.poly, W.n_poly, pts, P_n);
      }
    }
    if (!rejected) {
      if (n_nodes >= capn) { status = -2; break; }
      // ---------------------------------------------------------------- accept (:144-151)
      const int me = n_nodes, par = uni(g->par);
      if (lane < cnt) {
        const size_t gi = (size_t)(n_points + lane);
        double* ra = ptF + gi * 3;
        double* rbp = ptF + (size_t)capp * 3 + gi * 3;
        *reinterpret_cast<double2*>(ra) = make_double2(g->px[lane], g->py[lane]); ra[2] = g->pt[lane];
        *reinterpret_cast<double2*>(rbp) = make_double2(g->pth[lane], g->pv[lane]); rbp[2] = g->pl[lane];
      }
      if (lane == 0) nodeI[me] = make_int4(it, par, n_points, cnt);
      double qf = auvp_floor(ctt * Q.inv_bin_interval);
      const double r = auvp_fma(-qf, Q.bin_interval, ctt);
      if (r < 0.0) qf -= 1.0;
      else if (r >= Q.bin_interval) qf += 1.0;
      const double fi = qf + 1.0;
      const double curr_bin = fi * Q.bin_interval;
      const bool over = curr_bin > Q.max_traj_time;
      int app_bi = -1, app_c = 1;
      if (!over || fi <= (double)K) {
        const int bi = uni((int)fi);
        const int c = over ? 0 : uni(bin_count[bi]);
        if (c >= bcap) { status = -2; break; }
        int32_t* slot = bin_slot_for_append(bins, bi, c, next_chunk, lane == 0);
        if (!slot) { status = -2; break; }
        if (lane == 0) *slot = me;
        app_bi = bi; app_c = over ? uni(bin_count[bi]) : c;
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
      if (lane == 0) { ctl->hist_bin[(n_nodes - 1) & (TRIO_HIST - 1)] = app_bi; ctl->hist_cb[(n_nodes - 1) & (TRIO_HIST - 1)] = app_c; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke(&ctl->ver, n_nodes - 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) duo_poke(&ctl->done_seq, it + 1);
#ifdef AUVP_DUO_DIAG
    diag_t += __builtin_amdgcn_s_memtime() - t_t0;
#endif
  }
  // ---- the episode is over: H posts the stream position, T writes the record ----
  if (lane == 0) duo_poke(&ctl->stop, 1);
  {
    int spins = 0;
    while (!uni(duo_peek(&ctl->h_done)) || !uni(duo_peek(&ctl->m_done)) || !uni(duo_peek(&ctl->l_done))) {
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
    // EXPERIMENT ONLY: flushes in n_candidates; shader clocks / 256 at work: H | M << 20 | T << 40 in nn_scanned
    s.n_candidates = epoch;
    s.nn_scanned = (unsigned long long)(uint32_t)ctl->hist_bin[0] | ((unsigned long long)(uint32_t)ctl->hist_bin[1] << 20) | ((diag_t >> 8) << 40);
#endif
  }
}

}  // namespace auvp
#endif
