// planner_duo_kernel.h -- Planner_RRT.planning (gym_rrt/envs/rrt_dubins.py:162-289,374-423) for LATENCY runs (config 4: 512
// episodes, two per CU): TWO wavefronts per episode, the split rrt_duo_kernel makes for RRT.exploring.
//
// A step of the planner on one wavefront is ~7 us of dependent work (profiles/r2_latency/README.md): bucket choice 13 %, node
// pick 7.5 %, steer 30 %, collision 8 %, insert 11 %, goal arc 30 %.  What depends only on the random stream and on a few words
// of the tree -- the two `_randbelow` draws (:186, :223), the walk to the picked node along the bucket's member list, the
// parent's record, the number of sub-arcs and every sub-arc's dist / diff draws with their radius and angle (:262-271) --
// is produced ONE STEP AHEAD by a HELPER wavefront that owns the generator; the MAIN wavefront runs the theta / x / y / t
// chains, the collision test, the insert and the goal arc.
//
// Why the helper may run ahead.  Packet k+1 is built while main runs step k, so one insert can fall between the helper's look
// at the tree and the packet's use.  An insert changes what the helper looked at only if
//   * it went into the bucket the packet chose (its size feeds the node pick, its head moved), or
//   * it occupied a new bucket AND `_randbelow(n_occ)` would now come out differently: n_occ's bit length changed, or one of the
//     tries the helper threw away (>= old n_occ) is below the new n_occ -- the packet keeps the smallest rejected try.
// Main checks exactly that (the insert's bucket and the new n_occ are its own registers) and otherwise has the packet rebuilt
// from its first word: the generator is regenerated in place and the helper never lets generation overwrite the words since
// the packet's start.  Hand-over words, bounded waits and L2-served / clamped reads of what main writes: as rrt_duo_kernel.h.
// Bit-identical to prrt_kernel: trees, bucket lists, counters, paths, generator state and position
// (tests/test_gpu_planner_duo.py).
//
// Limits (the host falls back to prrt_kernel): plan mode (not the one-step mode of the environment), no step log,
// freq <= 30, <= 256 obstacles, at most four episodes per CU.  Since planner_pipe_kernel.h (four wavefronts, feed-forward: the
// default for these batches) this kernel is the selectable fallback (AUVP_PRRT_PIPE=0; AUVP_PRRT_TRIO=0 for two wavefronts).
#ifndef AUVP_PLANNER_DUO_KERNEL_H
#define AUVP_PLANNER_DUO_KERNEL_H
#include "planner_rrt_kernel.h"
#include "rrt_duo_kernel.h"

namespace auvp {

constexpr int PDUO_EP = 4;  // episodes per workgroup at most

struct PduoPacket {
  unsigned long long tag;      // {redo epoch, step + 1}, written last
  int ver, n_occ, rmin, status;  // the snapshot it was built from (nodes, occupied buckets); smallest rejected bucket try; 0 / -1 / -4
  int kind, b, par, n_total;     // kind 0: a steer follows; 1: the chosen bucket was empty (the step is used up)
  unsigned long long tmask;
  double cx, cy, cth, ctt;     // the parent's record
  double radius[DUO_CS], phi[DUO_CS];
};

struct PduoCtl {  // (first eight words: what the waits look at)
  int ver;            // nodes in the tree (main)
  int n_occ;          // occupied buckets (main)
  int valid_seq;      // main has accepted packets < valid_seq
  int redo_epoch;
  int stop, abort, helper_done, final_step;  // final_step: the last step main executed (the stream ends after its draws)
  double final_after;
  unsigned long long final_drawn;
  // three-wavefront form: the goal arc (connect_to_goal_curve_alt, :374-423) of step k is evaluated by a wavefront of its own
  // (G) while main runs step k + 1; main takes the verdict after that step's insert and takes the insert back when the arc
  // was free (the planning ended at step k)
  unsigned long long arc_tag;  // request: {0, step + 1}, written last
  double arc_lx, arc_ly, arc_th0;
  int arc_last, arc_done_seq, arc_free, g_done;  // arc_done_seq: requests of steps < arc_done_seq are decided; arc_free: the latest verdict
};

__host__ __device__ inline int pduo_per_episode_bytes(int max_pts) {
  int b = 624 * 4;
  b += ((max_pts * 16) + 15) & ~15;
  b += (int)((sizeof(PduoCtl) + 15) & ~(size_t)15);
  b += 2 * (int)((sizeof(PduoPacket) + 15) & ~(size_t)15);
  return b;
}

struct PduoView { int ver, n_occ, valid_seq, redo_epoch, stop, abort; };
__device__ __forceinline__ PduoView pduo_look(const PduoCtl* c) {
  __asm__ volatile("" ::: "memory");
  const int4 a = *reinterpret_cast<const int4*>(&c->ver);
  const int2 b = *reinterpret_cast<const int2*>(&c->stop);
  __asm__ volatile("" ::: "memory");
  PduoView v;
  v.ver = uni(a.x); v.n_occ = uni(a.y); v.valid_seq = uni(a.z); v.redo_epoch = uni(a.w); v.stop = uni(b.x); v.abort = uni(b.y);
  return v;
}

// random._randbelow(n) as rng_randbelow, also reporting the smallest try that was thrown away (>= n; 0x7fffffff: none)
__device__ __forceinline__ uint32_t pduo_randbelow(WaveRng& r, uint32_t n, int& rmin, bool& ok, uint64_t held_limit_drawn) {
  const int lane = lane_id();
  const int k = 32 - __clz((int)n);
  ok = true;
  for (;;) {
    // (eight words: never more than 64 ahead of the consumer, far inside what the packet may hold -- see `ensure` below)
    if (r.avail < 8u) {
      if ((r.drawn - held_limit_drawn) + r.avail + 64ull > 624ull) { ok = false; return 0u; }
      rng_ensure(r, r.avail + 1u);
    }
    uint32_t v = 0xffffffffu;
    if (lane < 8) v = rng_word(r, (uint32_t)lane) >> (32 - k);
    const unsigned long long okm = __ballot(lane < 8 && v < n);
    const int f = okm ? (__ffsll((long long)okm) - 1) : 8;
    // tries in front of the success (all eight when there is none) were >= n
    int mine = (lane < f) ? (int)(v & 0x7fffffffu) : 0x7fffffff;
#pragma unroll
    for (int o = 4; o >= 1; o >>= 1) { const int t = __shfl_xor(mine, o, 64); mine = t < mine ? t : mine; }
    const int m0 = __builtin_amdgcn_readfirstlane(mine);
    rmin = m0 < rmin ? m0 : rmin;
    if (okm) {
      const uint32_t res = (uint32_t)__builtin_amdgcn_readlane((int)v, f);
      rng_advance_words(r, (uint32_t)(f + 1));
      return res;
    }
    rng_advance_words(r, 8u);
  }
}

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
              if (__any(outside)) { free_ = false; break; }
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
                unsigned long long cm = __ballot(cand);
                while (cm) {
                  const int l = __ffsll((long long)cm) - 1;
                  cm &= cm - 1ull;
                  const double oxl = readlane_f64(ox[j], l), oyl = readlane_f64(oy[j], l), otl = readlane_f64(ot[j], l);
                  const double ex = ax - oxl, ey = ay - oyl;
                  hitl |= (lane < nv) && (ex * ex + ey * ey <= otl);
                }
              }
              if (__any(hitl)) free_ = false;
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

// NW = 2: helper + main.  NW = 3: helper + main + goal-arc wavefront.
template <int J, int NW>
__global__ __launch_bounds__(PDUO_EP * 64 * NW, 1) void prrt_duo_kernel(WorldDev W, PrrtParamsDev P, PrrtBuffers B, int n_episodes) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = uni((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  const int n_ep_wg = (int)(blockDim.x / (64 * NW));
  const int eidx = wave / NW, role = wave - NW * eidx;  // role 0: main, 1: helper, 2: goal arc
  const int per_ep = pduo_per_episode_bytes(B.max_pts);
  unsigned char* eb = smem + (size_t)eidx * per_ep;
  uint32_t* mt = reinterpret_cast<uint32_t*>(eb);
  eb += 624 * 4;
  double(*pts)[2] = reinterpret_cast<double(*)[2]>(eb);
  eb += ((B.max_pts * 16) + 15) & ~15;
  PduoCtl* ctl = reinterpret_cast<PduoCtl*>(eb);
  eb += (sizeof(PduoCtl) + 15) & ~(size_t)15;
  constexpr int PK_STRIDE = (int)((sizeof(PduoPacket) + 15) & ~(size_t)15);
  unsigned char* pk_base = eb;
  auto packet = [&](int k) -> PduoPacket* { return reinterpret_cast<PduoPacket*>(pk_base + (size_t)(k & 1) * PK_STRIDE); };
  (void)n_ep_wg;

  const int ep = (int)blockIdx.x * n_ep_wg + eidx;
  const bool helper = role == 1;
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
      ctl->arc_tag = 0ull; ctl->arc_done_seq = step0; ctl->arc_free = 0; ctl->g_done = NW == 3 ? 0 : 1;
      ctl->ver = sum.n_nodes; ctl->n_occ = sum.n_occ; ctl->valid_seq = step0; ctl->redo_epoch = 0; ctl->stop = 0; ctl->abort = 0;
      ctl->helper_done = 0; ctl->final_step = step0 - 1; ctl->final_after = 0.0; ctl->final_drawn = 0ull;
      packet(0)->tag = 0ull; packet(1)->tag = 0ull;
    }
  } else if (role == 1) {
    for (int i = lane; i < 624; i += 64) mt[i] = B.mt[eps * 624 + i];
  }
  __threadfence_block();
  __syncthreads();
  if (!valid_ep) return;
  auto give_up = [&]() { if (lane == 0) duo_poke(&ctl->abort, 1); };

  if (helper) {
    // ======================================================================================================= HELPER
    WaveRng rng;
    rng.s = mt;
    rng.pslot = (uint32_t)uni(B.rng_state[4 * eps]);
    rng.avail = (uint32_t)uni(B.rng_state[4 * eps + 1]);
    rng.drawn = ((unsigned long long)(uint32_t)uni(B.rng_state[4 * eps + 2])) | ((unsigned long long)(uint32_t)uni(B.rng_state[4 * eps + 3]) << 32);
    int epoch = 0;
    // stream positions at the first word of the latest packet (slot cur & 1) and of the one before (the other slot): main can
    // ask for the latest to be rebuilt, and -- three-wavefront form -- the planning can end one step BEFORE the latest packet
    // main had accepted (a free goal arc drops the step in progress)
    uint32_t spp0 = rng.pslot, spp1 = rng.pslot;
    unsigned long long spd0 = rng.drawn, spd1 = rng.drawn;
    int cur = step0 - 1;                        // the latest packet started
    unsigned long long hold_drawn = rng.drawn;  // oldest position a rewind may ask for: generation stays within 624 words of it
    auto rewind_to = [&](int kk) {
      const uint32_t ps = (kk & 1) ? spp1 : spp0;
      const unsigned long long dr = (kk & 1) ? spd1 : spd0;
      rng.avail = (uint32_t)uni((int)(rng.avail + (uint32_t)(rng.drawn - dr)));
      rng.pslot = ps; rng.drawn = dr;
    };
    auto ensure = [&](uint32_t need) -> bool {
      while (rng.avail < need) {
        if ((rng.drawn - hold_drawn) + rng.avail + 64ull > 624ull) return false;
        rng_ensure(rng, rng.avail + 1u);
      }
      return true;
    };
    int k = step0;
#ifdef AUVP_DUO_DIAG
    unsigned long long diag_h = 0ull;
#endif
    for (;;) {
      PduoView cv;
      {
        int spins = 0;
        for (;;) {
          cv = pduo_look(ctl);
          if (cv.stop || cv.abort) goto helper_end;
          if (cv.redo_epoch != epoch) {  // packet k - 1 is rebuilt from its first word
            epoch = cv.redo_epoch;
            k -= 1;
            rewind_to(k);
            break;
          }
          if (k < P.max_step && cv.valid_seq >= k) break;
          if (++spins > DUO_SPIN_LIMIT) { give_up(); goto helper_end; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      // ---------------------------------------------------------------- build packet k (from the snapshot `cv` just taken)
#ifdef AUVP_DUO_DIAG
      const unsigned long long t_h0 = __builtin_amdgcn_s_memtime();  // EXPERIMENT ONLY (tools/prrt_duo_probe.py)
#endif
      cur = k;
      if (k & 1) { spp1 = rng.pslot; spd1 = rng.drawn; } else { spp0 = rng.pslot; spd0 = rng.drawn; }
      hold_drawn = k > step0 ? ((k & 1) ? spd0 : spd1) : rng.drawn;  // (the packet before this one, while there is one)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int ver = cv.ver, n_occ = cv.n_occ;
      PduoPacket* q = packet(k);
      int status = 0, kind = 0, b = 0, par = 0, n_total = 0, rmin = 0x7fffffff;
      unsigned long long tmask = 0ull;
      bool fits = true;
      if (n_occ <= 0) status = -1;
      if (status == 0) {
        bool ok1 = true;
        const uint32_t oi = pduo_randbelow(rng, (uint32_t)n_occ, rmin, ok1, hold_drawn);
        fits = ok1;
        if (fits) {
          int bb = duo_ld_i32(occupied + (oi < (uint32_t)capn ? oi : 0u));
          b = uni(bb < 0 ? 0 : (bb >= P.n_buckets ? P.n_buckets - 1 : bb));
          const long long bwl = __builtin_nontemporal_load(reinterpret_cast<const long long*>(buckets + b));
          const int2 bw = make_int2((int)(bwl & 0xffffffffll), (int)(bwl >> 32));
          const int cnt_b = uni(prrt_bucket_count(bw, epoch_b));
          if (cnt_b == 0) kind = 1;
          else {
            int dummy = 0x7fffffff;
            bool ok2 = true;
            const int rsel = (int)pduo_randbelow(rng, (uint32_t)cnt_b, dummy, ok2, hold_drawn);
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
                  int nx = uni(duo_ld_i32(&nodes[pv].next));
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
                      const unsigned long long bal = __ballot(is);
                      const int cc = __popcll(bal);
                      if (seen + cc > rsel) {
                        const int want = rsel - seen;
                        const unsigned long long sel = __ballot(is && (int)__popcll(bal & ((1ull << lane) - 1ull)) == want);
                        pv = base + 64 * c + (__ffsll((long long)sel) - 1);
                      }
                      seen += cc;
                    }
                  }
                }
                pv = uni(pv);
                if (pv < 0) { status = -4; pv = 0; }  // (or a snapshot that raced with an insert: main has the packet rebuilt)
              }
              par = pv;
              const double* pr = &nodes[par].x;
              const double p0 = duo_ld_f64(pr), p1 = duo_ld_f64(pr + 1), p2 = duo_ld_f64(pr + 2), p3 = duo_ld_f64(pr + 3);
              fits = ensure(2u);
              if (fits) {
                const double u = rng_random_at(rng, 0u);
                rng_advance_words(rng, 2u);
                n_total = uni((int)auvp_floor(py_uniform(0.0, P.freq, u) / 1));
                const int n = n_total;
                fits = ensure((uint32_t)(4 * n));
                if (fits) {
                  const bool active = lane < n;
                  double radius = 0.0, phi = 0.0;
                  bool taken = false;
                  if (active) {
                    const double dist = py_uniform(0.0, P.dist_to_end, rng_random_at(rng, (uint32_t)(2 * lane)));
                    const double diff = py_uniform(-P.diff_max, P.diff_max, rng_random_at(rng, (uint32_t)(2 * lane + 1)));
                    taken = auvp_fabs(dist) > auvp_fabs(diff);
                    if (taken) {
                      const double s1 = dist + diff, s2 = dist - diff;
                      radius = (s1 + s2) / (-s1 + s2);
                      phi = (s1 + s2) / (2 * radius);
                    }
                  }
                  tmask = __ballot(taken);
                  rng_advance_words(rng, (uint32_t)(4 * n));
                  if (lane < DUO_CS) { q->radius[lane] = radius; q->phi[lane] = phi; }
                  if (lane == 0) { q->cx = p0; q->cy = p1; q->cth = p2; q->ctt = p3; }
                }
              }
            }
          }
        }
      }
      if (!fits) {
        // (a packet is a few dozen words; the generator can only run out of room behind a corrupt state)
        give_up();
        goto helper_end;
      }
      if (lane == 0) {
        q->ver = ver; q->n_occ = n_occ; q->rmin = rmin; q->status = status; q->kind = kind; q->b = b; q->par = par; q->n_total = n_total;
        q->tmask = tmask;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke64(&q->tag, duo_tag(epoch, k));
#ifdef AUVP_DUO_DIAG
      diag_h += __builtin_amdgcn_s_memtime() - t_h0;
#endif
      k++;
    }
  helper_end:
    {
      // the generator goes back to HBM where the planning ended: after the draws of the last step main executed (a packet built
      // or begun beyond it is undone)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int fs = uni(duo_peek(&ctl->final_step));
      if (cur > fs) rewind_to(fs + 1);  // (fs + 1 is the latest packet or the one before it)
      const unsigned long long drawn = rng.drawn;
      for (int i = lane; i < 624; i += 64) B.mt[eps * 624 + i] = mt[i];
      if (lane == 0) {
        B.rng_state[4 * eps] = (int32_t)rng.pslot;
        B.rng_state[4 * eps + 1] = (int32_t)rng.avail;
        B.rng_state[4 * eps + 2] = (int32_t)(uint32_t)(drawn & 0xffffffffull);
        B.rng_state[4 * eps + 3] = (int32_t)(uint32_t)(drawn >> 32);
      }
      WaveRng peek = rng;
      wave_sync();
      rng_ensure(peek, 2u);  // may generate ahead in LDS only; the stored words above are untouched
      const double after = rng_random_at(peek, 0u);
      if (lane == 0) { ctl->final_after = after; ctl->final_drawn = drawn; }
#ifdef AUVP_DUO_DIAG
      if (lane == 0) ctl->arc_lx = (double)diag_h;
#endif
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke(&ctl->helper_done, 1);
    }
    return;
  }

  // ================================================================================================ MAIN / GOAL ARC
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
  if (NW == 3 && role == 2) {
    // ---- G: the goal arcs main asks for, one at a time (main takes a verdict before it posts the next request)
#ifdef AUVP_DUO_DIAG
    unsigned long long diag_g = 0ull;
#endif
    for (int n = 0;; n++) {
      int spins = 0;
      bool stop = false;
      for (;;) {
        const unsigned long long tg = duo_peek64(&ctl->arc_tag);
        const PduoView cv = pduo_look(ctl);
        if (cv.stop || cv.abort) { stop = true; break; }
        if (tg == duo_tag(0, n)) break;
        if (++spins > DUO_SPIN_LIMIT) { give_up(); stop = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (stop) break;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const double lx = readfirst_f64(ctl->arc_lx), ly = readfirst_f64(ctl->arc_ly), th0 = readfirst_f64(ctl->arc_th0);
      const int last = uni(ctl->arc_last);
      int n_arc = -1;
#ifdef AUVP_DUO_DIAG
      const unsigned long long t_g0 = __builtin_amdgcn_s_memtime();
#endif
      const bool is_free = prrt_goal_arc<J>(P, ox, oy, ot, orr, gx, gy, lx, ly, th0, nodes, last, sum, n_arc);
#ifdef AUVP_DUO_DIAG
      diag_g += __builtin_amdgcn_s_memtime() - t_g0;
#endif
      if (lane == 0) ctl->arc_free = is_free ? 1 : 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke(&ctl->arc_done_seq, n + 1);
    }
#ifdef AUVP_DUO_DIAG
    if (lane == 0) ctl->arc_ly = (double)diag_g;
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) duo_poke(&ctl->g_done, 1);
    return;
  }
  int n_nodes = uni(sum.n_nodes), n_points = uni(sum.n_points), n_occ = uni(sum.n_occ), step = step0;
  int done = uni(sum.done), status = uni(sum.status);
  int last_accepted = 0, last_new = -1, last_bk = -2, my_epoch = 0;
  int prev_n_arc = -1;
  bool have_prev_arc = false;
  // three-wavefront form: the arc request in flight (at most one), the step in progress, what the previous step reported
#ifdef AUVP_DUO_DIAG
  unsigned long long diag_m = 0ull, diag_w = 0ull;
#endif
  bool arc_pending = false, in_step = false;
  int arc_reqs = 0, prev_acc = 0, prev_new = -1;
  // what the insert of the step in progress replaced (three-wavefront form: that insert is taken back when the arc of the step
  // before turns out free -- the planning ended there)
  bool un_ins = false, un_occ = false;
  int un_bk = -1, un_cnt = 0;
  int2 un_word = make_int2(0, 0);
  // the verdict of the request in flight.  A free arc ends the planning at the step that posted it: the step in progress is
  // dropped -- its insert, if it got that far, is taken back (bucket word and the three counters; what it appended lies past
  // them) -- and the stream ends before its draws
  auto take_verdict = [&]() {
    if (!arc_pending) return;
    int spins = 0;
    while (uni(duo_peek(&ctl->arc_done_seq)) < arc_reqs) {
      if (uni(duo_peek(&ctl->abort)) || ++spins > DUO_SPIN_LIMIT) { give_up(); status = -9; arc_pending = false; return; }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    arc_pending = false;
    if (uni(duo_peek(&ctl->arc_free)) != 0) {
      done = 1;
      if (in_step) {
        status = 0;
        last_accepted = prev_acc; last_new = prev_new;
        if (lane == 0) ctl->final_step = step - 1;
        if (un_ins) {
          if (lane == 0 && un_bk >= 0) buckets[un_bk] = un_word;
          if (un_occ) n_occ--;
          n_nodes--;
          n_points -= un_cnt;
          un_ins = false;
        }
        in_step = false;
      }
    }
  };
  auto lane_f64 = [](double v, int src) {
    const long long bits = __double_as_longlong(v);
    const int lo = __shfl((int)(bits & 0xffffffffll), src, 64), hi = __shfl((int)(bits >> 32), src, 64);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  };
  while (status == 0 && !done && step < P.max_step) {
    status = uni(status); done = uni(done); step = uni(step);
    n_nodes = uni(n_nodes); n_points = uni(n_points); n_occ = uni(n_occ);
    // ---------------------------------------------------------------- the step's packet
    PduoPacket* q = packet(step);
    {
      int spins = 0;
      for (;;) {
        if (duo_peek64(&q->tag) == duo_tag(my_epoch, step)) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          // built one insert ago?  Then only if that insert did not change what its choices looked at
          const int pv = uni(q->ver);
          bool conflict = false;
          if (pv != n_nodes) {
            const int pn = uni(q->n_occ);
            conflict = pv != n_nodes - 1 || uni(q->status) != 0 || last_bk == uni(q->b) ||
                       (pn != n_occ && ((32 - __clz(pn)) != (32 - __clz(n_occ)) || uni(q->rmin) < n_occ));
          }
          if (!conflict) break;
          my_epoch++;
          if (lane == 0) duo_poke(&ctl->redo_epoch, my_epoch);
        }
        if (uni(duo_peek(&ctl->abort))) { status = -9; break; }
        if (++spins > DUO_SPIN_LIMIT) { give_up(); status = -9; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (uni(status)) break;
#ifdef AUVP_DUO_DIAG
    const unsigned long long t_m0 = __builtin_amdgcn_s_memtime();
#endif
    if (lane == 0) duo_poke(&ctl->valid_seq, step + 1);  // the helper may build the next packet now
    in_step = true;
    un_ins = false;
    prev_acc = last_accepted; prev_new = last_new;
    if (uni(q->status) != 0) { status = uni(q->status); break; }
    last_accepted = 0; last_new = -1;
    if (lane == 0) ctl->final_step = step;  // (this step's draws count from here on)
    if (uni(q->kind) == 1) {  // generate_one_node on an empty bucket: (False, None) (:214-220)
      take_verdict();
      if (done || uni(status)) break;
      step++;
      in_step = false;
      continue;
    }
    const int par = uni(q->par), n_total = uni(q->n_total);
    const unsigned long long tmask = q->tmask;
    double cx = readfirst_f64(q->cx), cy = readfirst_f64(q->cy), cth = readfirst_f64(q->cth), ctt = readfirst_f64(q->ctt);
    // ---------------------------------------------------------------- steer, the half that needs the parent (:271-289)
    int cnt = 0;
    if (lane == 0) { pts[0][0] = cx; pts[0][1] = cy; }
    double bbx0 = cx, bbx1 = cx, bby0 = cy, bby1 = cy;
    bool cap_err = false;
    if (n_total > 0) {
      const int n = n_total;
      const bool taken = (tmask >> lane) & 1ull;
      double radius = 0.0, phi = 0.0;
      if (lane < DUO_CS) { radius = q->radius[lane]; phi = q->phi[lane]; }
      double th = cth, myth = cth;
      for (int s = 0; s < n; s++) {
        if ((tmask >> s) & 1ull) th = prrt_angle_wrap(th + readlane_f64(phi, s));
        if (lane == s) myth = th;
      }
      double sn, cs;
      auvp_sincos(myth, &sn, &cs);
      double dx = 0.0, dy = 0.0, dt = 0.0;
      {
        const unsigned long long below = tmask & ((1ull << lane) - 1ull);
        const int prev = below ? (63 - __clzll((long long)below)) : (DUO_CS - 1);  // lane 31 is idle: the entry angle
        const double so = lane_f64(sn, prev), co = lane_f64(cs, prev);
        if (taken) {
          dx = radius * (sn - so);
          dy = radius * (-cs + co);
          dt = auvp_sqrt(dx * dx + dy * dy) / 1;
        }
      }
      double mx = 0.0, my = 0.0, mt_ = 0.0;
      for (int s = 0; s < n; s++) {
        cx = cx + readlane_f64(dx, s);
        cy = cy + readlane_f64(dy, s);
        ctt = ctt + readlane_f64(dt, s);
        bbx0 = __builtin_fmin(cx, bbx0); bbx1 = __builtin_fmax(cx, bbx1);
        bby0 = __builtin_fmin(cy, bby0); bby1 = __builtin_fmax(cy, bby1);
        if (lane == s) { mx = cx; my = cy; mt_ = ctt; }
      }
      cth = th;
      const int napp = __popcll(tmask);
      if (n_points + napp > (int)capp || napp + 2 > B.max_pts) cap_err = true;
      if (!cap_err) {
        if (taken) {
          const int rank = __popcll(tmask & ((1ull << lane) - 1ull));
          const size_t gi = (size_t)(n_points + rank);
          double2* pr = reinterpret_cast<double2*>(ptF + gi * 4);
          pr[0] = make_double2(mx, my); pr[1] = make_double2(myth, mt_);
          pts[rank + 1][0] = mx;
          pts[rank + 1][1] = my;
        }
        cnt = napp;
      }
    }
    if (cap_err) { status = -2; break; }
    wave_sync();
    const int P_n = cnt + 1;
    // ---------------------------------------------------------------- check_collision_free (:435-458)
    bool ok;
    {
      bool outside = false;
      for (int p = lane; p < P_n; p += 64) {
        const double x = pts[p][0], y = pts[p][1];
        const bool wx = (x >= P.rect[0]) && (x <= P.rect[2]);
        const bool wy = (y >= P.rect[1]) && (y <= P.rect[3]);
        outside = outside | !(wx && wy);
      }
      ok = !prrt_hits<J>(ox, oy, ot, orr, pts, P_n, bbx0, bby0, bbx1, bby1) && !__any(outside);
    }
    int me = -1;
    last_bk = -2;
    if (ok) {
      if (n_nodes >= capn) { status = -2; break; }
      me = n_nodes;
      int row = (int)(cy / P.cell), col = (int)(cx / P.cell);
      bool idx_err = false;
      if (row < 0) { row += P.rows; idx_err |= row < 0; }
      if (col < 0) { col += P.cols; idx_err |= col < 0; }
      int bk = -1;
      if (!idx_err && row < P.rows && col < P.cols) {
        const double raw = cth / P.delta_theta;
        int sub = (int)auvp_floor(raw);
        if (sub < 0) sub = (int)(P.S + sub);
        if (sub == P.S) sub -= 1;
        if (sub < 0) { sub += P.S; idx_err |= sub < 0; }
        idx_err |= sub >= P.S;
        bk = (row * P.cols + col) * P.S + sub;
      }
      if (__any(idx_err)) { status = -1; break; }
      bk = uni(bk);
      int2 bwn = make_int2(0, 0);
      if (bk >= 0) bwn = buckets[bk];
      const int c_before = bk >= 0 ? uni(prrt_bucket_count(bwn, epoch_b)) : -1;
      const int h_before = bk >= 0 ? uni(bwn.y) : -1;
      if (lane < 4) {
        const int nx = (bk >= 0 && c_before > 0) ? h_before : -1;
        int4 r4;
        if (lane == 0) r4 = make_int4(__double2loint(cx), __double2hiint(cx), __double2loint(cy), __double2hiint(cy));
        else if (lane == 1) r4 = make_int4(__double2loint(cth), __double2hiint(cth), __double2loint(ctt), __double2hiint(ctt));
        else if (lane == 2) r4 = make_int4(step, par, n_points, cnt);
        else r4 = make_int4(bk, nx, 0, 0);
        reinterpret_cast<int4*>(&nodes[me])[lane] = r4;
      }
      if (lane == 0) {
        nbucket[me] = bk;
        if (bk >= 0) {
          buckets[bk] = prrt_bucket_word(c_before + 1, me, epoch_b);
          if (c_before == 0) occupied[n_occ] = bk;
        }
      }
      un_ins = true; un_occ = c_before == 0; un_bk = bk; un_cnt = cnt; un_word = bwn;
      if (c_before == 0) n_occ++;
      n_nodes++;
      n_points += cnt;
      last_accepted = 1; last_new = me;
      last_bk = bk;
      // the insert is published: record, bucket word and occupied list first, then the counters the helper's snapshots start from
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) { ctl->n_occ = n_occ; duo_poke(&ctl->ver, n_nodes); }
    }
#ifdef AUVP_DUO_DIAG
    const unsigned long long t_m1 = __builtin_amdgcn_s_memtime();
    diag_m += t_m1 - t_m0;
#endif
    take_verdict();  // the arc of the step before: G had this whole step for it
#ifdef AUVP_DUO_DIAG
    const unsigned long long t_m2 = __builtin_amdgcn_s_memtime();
    diag_w += t_m2 - t_m1;
#endif
    if (done || uni(status)) break;
    // ---------------------------------------------------------------- connect_to_goal_curve_alt(mps_list[-1]) (:374-423)
    const int last = n_nodes - 1;
    double lx, ly, th0;
    if (ok) { lx = cx; ly = cy; th0 = cth; }
    else {
      const double2 a = *reinterpret_cast<const double2*>(&nodes[last].x);
      lx = readfirst_f64(a.x); ly = readfirst_f64(a.y); th0 = readfirst_f64(nodes[last].theta);
    }
    int n_arc = -1;
    if (!ok && have_prev_arc) n_arc = prev_n_arc;
    else if (NW == 2) {
      if (prrt_goal_arc<J>(P, ox, oy, ot, orr, gx, gy, lx, ly, th0, nodes, last, sum, n_arc)) done = 1;
    } else {
      // the arc of this step goes to G; main goes on with the next step and picks the verdict up before its insert
      if (lane == 0) { ctl->arc_lx = lx; ctl->arc_ly = ly; ctl->arc_th0 = th0; ctl->arc_last = last; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) duo_poke64(&ctl->arc_tag, duo_tag(0, arc_reqs));
      arc_reqs++;
      arc_pending = true;
    }
    prev_n_arc = n_arc; have_prev_arc = true;
    step++;
    in_step = false;
#ifdef AUVP_DUO_DIAG
    diag_m += __builtin_amdgcn_s_memtime() - t_m2;
#endif
  }
  take_verdict();  // (the last step's arc; or the one in flight when a later step failed: a free arc ends the planning before it)
  // ---- the planning is over: the helper stores the generator where it ended, main the record ----
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) duo_poke(&ctl->stop, 1);
  {
    int spins = 0;
    while (!uni(duo_peek(&ctl->helper_done)) || !uni(duo_peek(&ctl->g_done))) {
      if (++spins > DUO_SPIN_LIMIT) { status = -9; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  if (lane == 0) {
    sum.status = status; sum.n_nodes = n_nodes; sum.n_points = n_points; sum.n_occ = n_occ; sum.steps = step;
    sum.done = done; sum.last_accepted = last_accepted; sum.last_new_node = last_new;
    sum.rng_after = ctl->final_after; sum.n_draw32 = ctl->final_drawn;
    if (!done) sum.path_len = 0;
#ifdef AUVP_DUO_DIAG
    if (!done) { sum.arc[0] = (double)diag_m; sum.arc[1] = (double)diag_w; sum.arc[2] = ctl->arc_lx; sum.arc[3] = ctl->arc_ly; sum.arc[4] = (double)my_epoch; }
#endif
  }
}

}  // namespace auvp
#endif
