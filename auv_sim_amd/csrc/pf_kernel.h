// pf_kernel.h -- the shark particle filter (particleFilter.py) on gfx950: one T-thread workgroup = one filter
// (N particles, PPT consecutive list positions per thread; T = 512 measured fastest, pf_host.h), F filters per
// launch, S steps per launch.
//
//   pf_create_kernel   Particle.__init__ x N                       particleFilter.py:44-53, 311-317
//   pf_step_kernel     create_and_update                           :277-282  (Particle.update_particle :55-75)
//                      update_weights (weight, normalize, correct) :285-310, 92-116, 127-151, 179-252
//                      particleMean, meanError                     :153-177
//
// The reference draws from numpy's global legacy RandomState (its `random` is numpy.random, :8):
// MT19937 state lives in LDS, two buffers: a refill reads the old state and writes the new one into the other
// buffer, thread t forming words t, 227 + t and 454 + t (each needs old words and the thread's own previous
// result: the recurrence reads 397 words ahead), so a regeneration has no barrier inside (round 4; six before).  Draws
// whose count is data dependent (randint's masked rejection in random.choice) are taken a state block
// at a time: temper, flag `word & mask <= rng`, block scan, the r-th accepted word is the r-th choice.
//
// Object identity: `correct` deep-copies and then picks list indices with replacement, so one Particle
// object can sit at several list positions and create_and_update moves it once per position, each time
// with that position's two uniforms.  Positions that share an object are applied in list order by
// rounds of atomicMin over the pending positions (#rounds = largest multiplicity, ~5 at N = 1000).
//
// Ordered sums (particleMean, the per-AUV weight sum) keep the reference's left-to-right order; max()
// is order free.  Math through auvp_math.h / auvp_exp.h: bit-identical to the portable checker build.
#ifndef AUVP_PF_KERNEL_H
#define AUVP_PF_KERNEL_H
#include "auvp_exp.h"
#include "auvp_math_late.h"
#include "pf_types.h"

namespace auvp {

#define PF_T 256
#define PF_UPPER 0x80000000u
#define PF_LOWER 0x7fffffffu
#define PF_MAG 0x9908b0dfu
#define PF_PI 3.141592653589793

struct PfRng {
  uint32_t* mt;   // LDS [624]: the current state
  uint32_t* alt;  // LDS [624]: the buffer the next regeneration writes
  int pos;        // uniform
  unsigned long long drawn;
};

__device__ __forceinline__ uint32_t pf_twist(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t y = (a & PF_UPPER) | (b & PF_LOWER);
  return c ^ (y >> 1) ^ ((y & 1u) ? PF_MAG : 0u);
}

// one MT19937 state regeneration: thread t < 227 forms the new words t and 227 + t, t < 170 also 454 + t, from the old
// state and its own results (word 623 needs the new word 0: thread 169 forms that one again) and writes them to the other
// buffer -- no barrier inside; the CALLER's barrier publishes the new state (and separates this regeneration's reads
// from the next one's writes to the same buffer).  With dst: the tempered words [0, take) go there as well.
__device__ __forceinline__ void pf_refill(PfRng& r, int tid, uint32_t* dst = nullptr, int take = 0) {
  const uint32_t* o = r.mt;
  uint32_t* n = r.alt;
  if (tid < 227) {
    const uint32_t a = pf_twist(o[tid], o[tid + 1], o[tid + 397]);
    const uint32_t b = pf_twist(o[227 + tid], o[228 + tid], a);
    n[tid] = a;
    n[227 + tid] = b;
    if (tid < take) dst[tid] = mt_temper(a);
    if (227 + tid < take) dst[227 + tid] = mt_temper(b);
    if (tid < 170) {
      const uint32_t nx = tid == 169 ? pf_twist(o[0], o[1], o[397]) : o[455 + tid];
      const uint32_t c = pf_twist(o[454 + tid], nx, b);
      n[454 + tid] = c;
      if (454 + tid < take) dst[454 + tid] = mt_temper(c);
    }
  }
  r.alt = r.mt;
  r.mt = n;
  r.pos = 0;
}

// the next `count` tempered outputs -> dst[0..count) (LDS); ends with a barrier
template <int T>
__device__ __forceinline__ void pf_gen_words(PfRng& r, uint32_t* dst, int count, int tid) {
  static_assert(T >= 227, "a regeneration is one pass of 227 threads");
  int done = 0;
  bool open = true;  // words written since the last barrier
  while (done < count) {
    if (r.pos >= 624) {
      const int take = min(624, count - done);
      pf_refill(r, tid, dst + done, take);
      r.pos = take;
      done += take;
      __syncthreads();
      open = false;
    } else {  // what is left of the current state: the regeneration that follows reads this buffer too, no barrier
      const int take = min(624 - r.pos, count - done);
      for (int i = tid; i < take; i += T) dst[done + i] = mt_temper(r.mt[r.pos + i]);
      r.pos += take;
      done += take;
      open = true;
    }
  }
  if (open) __syncthreads();
  r.drawn += (unsigned long long)count;
}

__device__ __forceinline__ double pf_double(uint32_t w0, uint32_t w1) {
  return ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) / 9007199254740992.0;
}

// angle_wrap (:18-33): one rounded add per recursion level; false = nan / deeper than CPython recurses
// The direction never flips: 2*PF_PI is exactly twice PF_PI and rounding is monotone, so a value above pi lands
// in [-pi, pi] or stays above pi (and likewise below -pi).  One compare and one add per level.
__device__ __forceinline__ bool pf_angle_wrap(double& a) {
  double ang = a;
  int depth = 0;
  while (ang > PF_PI && depth < 899) { ang += (-2 * PF_PI); depth++; }   // 899 adds = the deepest chain the
  while (ang < -PF_PI && depth < 899) { ang += (2 * PF_PI); depth++; }   // checker's 900-level loop accepts
  a = ang;
  return -PF_PI <= ang && ang <= PF_PI;  // false: nan, or a chain deeper than CPython recurses
}

template <int T>
__device__ __forceinline__ double pf_block_max(double v, double* red, int tid) {
  v = auvp::wave_max_f64(v);  // (DPP path: auvp_wave.h)
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double m = red[0];
  for (int w = 1; w < T / 64; w++) m = red[w] > m ? red[w] : m;
  return m;
}

// two maxima at once (the weights of two AUV measurements): same barriers as one
template <int T>
__device__ __forceinline__ void pf_block_max2(double& v0, double& v1, double* red, int tid) {
  v0 = auvp::wave_max_f64(v0);
  v1 = auvp::wave_max_f64(v1);
  __syncthreads();
  if ((tid & 63) == 0) { red[tid >> 6] = v0; red[20 + (tid >> 6)] = v1; }
  __syncthreads();
  double m0 = red[0], m1 = red[20];
  for (int w = 1; w < T / 64; w++) { m0 = red[w] > m0 ? red[w] : m0; m1 = red[20 + w] > m1 ? red[20 + w] : m1; }
  v0 = m0; v1 = m1;
}

// exclusive scan of one int per thread over the workgroup; *total = sum
template <int T>
__device__ __forceinline__ int pf_block_scan(int v, int* red, int tid, int* total) {
  int inc = v;
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if ((tid & 63) >= o) inc += t; }
  __syncthreads();
  if ((tid & 63) == 63) red[tid >> 6] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < T / 64; w++) { if (w < (tid >> 6)) base += red[w]; tot += red[w]; }
  *total = tot;
  return base + inc - v;
}

// weight of one particle for one AUV measurement (:92-116, 285-300): calc_particle_alpha, calc_particle_range, the two
// Gaussians
__device__ __forceinline__ double pf_weight(double px, double py, double mx, double my, double mth, double auv_alpha, double auv_range,
                                            const double* etab, int& status) {
  double pa = auvp_atan2_late((-my + py), (px + -mx)) - mth;     // calc_particle_alpha
  if (!pf_angle_wrap(pa)) status = PF_ERR_ANGLE;
  const double dy = my - py, dx = mx - px;
  const double pr = auvp_sqrt_plain(dy * dy + dx * dx);          // calc_particle_range (0 or >= the squared spacing of the coordinates)
  double d = pa - auv_alpha;
  if (!pf_angle_wrap(d)) status = PF_ERR_ANGLE;
  const double constant = 1.2533141375;
  const double fa = .001 + (1 / (constant) * (auvp_pow_e_t((-(d * d)) / (0.5), etab)));
  const double dr = pr - auv_range;
  const double fw = .001 + (1 / (100 * constant) * (auvp_pow_e_t(-auvp_div_plain(dr * dr, 20000), etab)));  // (-(x)) / c == -(x / c)
  return fw * fa;
}

struct PfLds {
  uint32_t *mt, *mt2; int* red_i; double* red_d; double* etab; double *sx, *sy, *sv, *sth, *sw; int* off; uint32_t* wbuf; int* slot;
};
__host__ __device__ inline size_t pf_lds_bytes(int N) {
  const size_t n2 = (size_t)(N + 2) & ~(size_t)1;
  return 2 * 624 * 4 + 32 * 4 + 40 * 8 + AUVP_EXP_TBL_DOUBLES * 8 + 5 * n2 * 8 + n2 * 4 + (size_t)5 * n2 * 4;
}
__device__ __forceinline__ PfLds pf_carve(unsigned char* smem, int N) {
  const size_t n2 = (size_t)(N + 2) & ~(size_t)1;
  PfLds L;
  L.mt = (uint32_t*)smem;
  L.mt2 = L.mt + 624;
  L.red_i = (int*)(L.mt2 + 624);
  L.red_d = (double*)(L.red_i + 32);  // red_i [0,16): reductions, 16: choice cursor
  L.etab = L.red_d + 40;              // red_d [0,16): reductions, 16..17: means; etab: auvp_exp_table() (pf_step_kernel)
  L.sx = L.etab + AUVP_EXP_TBL_DOUBLES;
  L.sy = L.sx + n2; L.sv = L.sy + n2; L.sth = L.sv + n2; L.sw = L.sth + n2;
  L.off = (int*)(L.sw + n2);
  L.wbuf = (uint32_t*)(L.off + n2);  // 5*n2 words: the RNG window, then the alias slots, then copy -> source particle (off: the drawn indices)
  L.slot = (int*)L.wbuf;
  return L;
}

__global__ __launch_bounds__(PF_T) void pf_create_kernel(PfDev D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int f = blockIdx.x, tid = threadIdx.x, N = D.N;
  PfLds L = pf_carve(smem, N);
  PfRng r{L.mt, L.mt2, D.mtpos[f], 0ull};
  for (int i = tid; i < 624; i += PF_T) L.mt[i] = D.mt[(size_t)f * 624 + i];
  __syncthreads();
  const double x0 = D.shark0[2 * f], y0 = D.shark0[2 * f + 1];
  double* st = D.st + (size_t)f * 5 * N;
  const int chunk = (5 * ((N + 2) & ~1)) / 8;  // particles per RNG window (8 words each)
  for (int base = 0; base < N; base += chunk) {
    const int cnt = min(chunk, N - base);
    pf_gen_words<PF_T>(r, L.wbuf, 8 * cnt, tid);
    for (int i = tid; i < cnt; i += PF_T) {
      const uint32_t* w = L.wbuf + 8 * i;
      const int p = base + i;
      st[p] = x0 + (-150.0 + 300.0 * pf_double(w[0], w[1]));
      st[N + p] = y0 + (-150.0 + 300.0 * pf_double(w[2], w[3]));
      st[2 * N + p] = 0.0 + 5.0 * pf_double(w[4], w[5]);
      st[3 * N + p] = -PF_PI + (PF_PI - -PF_PI) * pf_double(w[6], w[7]);
      st[4 * N + p] = 1.0 / 1000;  // NUMBER_OF_PARTICLES is the literal 1000 in Particle.__init__ (:47,53)
      D.ent[(size_t)f * N + p] = p;
    }
    __syncthreads();
  }
  for (int i = tid; i < 624; i += PF_T) D.mt[(size_t)f * 624 + i] = r.mt[i];
  if (tid == 0) { D.mtpos[f] = r.pos; D.llen[f] = N; D.ndraw[f] += r.drawn; D.status[f] = PF_OK; }
}

template <int T, int PPT>
// (the register budget that lets the workgroups LDS admits be resident: two per CU up to 1024 particles = T / 128 wavefronts
// per SIMD, one beyond)
__global__ __launch_bounds__(T, (T * PPT <= 1280 ? 2 : 1) * T / 256) void pf_step_kernel(PfDev D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int f = blockIdx.x, tid0 = threadIdx.x, N = D.N, A = D.A;
  int tid = tid0;
  PfLds L = pf_carve(smem, N);
  PfRng r{L.mt, L.mt2, D.mtpos[f], 0ull};
  double* st = D.st + (size_t)f * 5 * N;
  _Pragma("unroll 1") for (int i = tid; i < 624; i += T) L.mt[i] = D.mt[(size_t)f * 624 + i];
  if (tid < AUVP_EXP_TBL_DOUBLES) L.etab[tid] = auvp_exp_table()[tid];  // 1 KB: the exp table's per-lane lookups from LDS
  _Pragma("unroll 1") for (int i = tid; i < N; i += T) {
    L.sx[i] = st[i]; L.sy[i] = st[N + i]; L.sv[i] = st[2 * N + i]; L.sth[i] = st[3 * N + i]; L.sw[i] = st[4 * N + i];
  }
  // the object id of a list position = the index the last `correct` drew for it: it lives in L.off (the choice array, only
  // rewritten by the next correct), not in registers across the step loop
  int p0 = tid * PPT;
#pragma unroll
  for (int j = 0; j < PPT; j++) if (p0 + j < N) L.off[p0 + j] = D.ent[(size_t)f * N + p0 + j];
  int llen = D.llen[f];
  int status = D.status[f];
  __syncthreads();
#ifdef AUVP_PF_DIAG  // clocks per section of the step, summed over the launch's steps -> err[k][f] (tools/pf_phases.py --clocks)
  unsigned long long pf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pf_t0 = __builtin_amdgcn_s_memtime();
#define PF_STAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pf_acc[k] += t_ - pf_t0; pf_t0 = t_; }
#else
#define PF_STAMP(k)
#endif

  for (int s = 0; s < D.S; s++) {
    // the thread id through an opaque move once per step: every per-thread LDS address is formed from it INSIDE the step --
    // hoisted out of the step loop (dozens of loop-invariant address registers) they were what the allocator spilled
    __asm__ volatile("" : "+v"(tid));
    p0 = tid * PPT;
    if (D.phases & PF_PHASE_UPDATE) {
      // ---- create_and_update: list position p consumes uniforms 2p, 2p+1 of this step
      pf_gen_words<T>(r, L.wbuf, 4 * N, tid);
      PF_STAMP(0)
      double u0[PPT], u1[PPT];
#pragma unroll
      for (int j = 0; j < PPT; j++) {
        const int p = p0 + j;
        if (p < N) { u0[j] = pf_double(L.wbuf[4 * p], L.wbuf[4 * p + 1]); u1[j] = pf_double(L.wbuf[4 * p + 2], L.wbuf[4 * p + 3]); }
      }
      __syncthreads();
      _Pragma("unroll 1") for (int i = tid; i < llen; i += T) L.slot[i] = 0x7fffffff;
      __syncthreads();
      // rounds: every pending position offers ((4096 - round) << 12 | position) to its object's slot with atomicMin -- a later
      // round's offers are below every earlier one, so the table is not reset in between; the smallest pending position of an
      // object is applied
      static_assert(T * PPT <= 4096, "positions take 12 bits of a slot");
      int lead[PPT], e[PPT];
      unsigned pending = 0;
#pragma unroll
      for (int j = 0; j < PPT; j++) { e[j] = 0; if (p0 + j < N) { pending |= 1u << j; e[j] = L.off[p0 + j]; } }
      for (int rd = 0;; rd++) {
        const int kb = (4096 - rd) << 12;
#pragma unroll
        for (int j = 0; j < PPT; j++) if (pending >> j & 1u) atomicMin(&L.slot[e[j]], kb | (p0 + j));
        __syncthreads();
        unsigned applied = 0;
#pragma unroll
        for (int j = 0; j < PPT; j++) {
          if (pending >> j & 1u) {
            const int top = L.slot[e[j]];
            if (rd == 0) lead[j] = top & 4095;  // the object's state lives at its first position
            if (top == (kb | (p0 + j))) {
              const int q = lead[j];
              double v = L.sv[q], th = L.sth[q];
              v += 0.0 + 5.0 * u0[j];                                   // uniform(0, RANDOM_VELOCITY)
              for (int d = 0; d < 900 && v > 5; d++) v += -5;           // velocity_wrap
              th += -(PF_PI / 2) + (PF_PI / 2 - -(PF_PI / 2)) * u1[j];  // uniform(-RANDOM_THETA, RANDOM_THETA)
              if (!pf_angle_wrap(th)) status = PF_ERR_ANGLE;
              double sn, cs;
              auvp_sincos(th, &sn, &cs);
              L.sv[q] = v; L.sth[q] = th;
              L.sx[q] += v * cs * .1;
              L.sy[q] += v * sn * .1;
              applied |= 1u << j;
            }
          }
        }
        pending &= ~applied;
        if (!__syncthreads_or(pending != 0)) break;
      }
      // every position reads its object's state (leaders' slots are only read here)
      double cx[PPT], cy[PPT], cv[PPT], ct[PPT];
#pragma unroll
      for (int j = 0; j < PPT; j++) if (p0 + j < N) { const int q = lead[j]; cx[j] = L.sx[q]; cy[j] = L.sy[q]; cv[j] = L.sv[q]; ct[j] = L.sth[q]; }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < PPT; j++) if (p0 + j < N) { const int p = p0 + j; L.sx[p] = cx[j]; L.sy[p] = cy[j]; L.sv[p] = cv[j]; L.sth[p] = ct[j]; }
      __syncthreads();
      if (D.updated) {
        double* u = D.updated + ((size_t)s * D.F + f) * N * 5;
        _Pragma("unroll 1") for (int i = tid; i < N; i += T) { u[5 * i] = L.sx[i]; u[5 * i + 1] = L.sy[i]; u[5 * i + 2] = L.sv[i]; u[5 * i + 3] = L.sth[i]; u[5 * i + 4] = L.sw[i]; }
      }
      // positions are independent copies again after correct; until then the shared object id stays
      PF_STAMP(1)
    }

    if (D.phases & PF_PHASE_WEIGHTS) {
      // ---- update_weights: per AUV measurement, weight of every particle (:285-300)
      // (the per-AUV lists stay in registers: a particle's weight is the sum of its normalised entries in AUV order, :302-305)
      // One particle at a time through the transcendental chain (atan2, two pow, sqrt): the loop over a thread's positions is
      // NOT unrolled and the per-AUV weight goes through LDS (wq: the RNG window's space, idle between create_and_update and
      // correct), so one instance of the chain is live at a time -- unrolled over PPT with the weights in registers the
      // 128-VGPR budget spilled 132 B per lane (2 GB of scratch write-back per launch, profiles/r4_pf_sog.md).
      double nw[PPT];
#pragma unroll
      for (int j = 0; j < PPT; j++) nw[j] = 0;
      double* wq = reinterpret_cast<double*>(L.wbuf);
      const double* mbase = D.meas + (((size_t)s * D.F + f) * A) * 5;
      int a = 0;
#ifdef AUVP_PF_AUV_PAIRS
      // two AUV measurements per trip: two independent chains per particle (the chain's latency, not issue, bounds the step)
      // and one pair of barriers for both maxima; the sums below still add in AUV order
      double* wq1 = wq + (((size_t)N + 1) & ~(size_t)1);
      for (; a + 2 <= A; a += 2) {
        const double* m = mbase + (size_t)a * 5;
        const double mx0 = m[0], my0 = m[1], mth0 = m[2], al0 = m[3], rg0 = m[4];
        const double mx1 = m[5], my1 = m[6], mth1 = m[7], al1 = m[8], rg1 = m[9];
        double lmax0 = -__builtin_inf(), lmax1 = -__builtin_inf();
#pragma unroll 1
        for (int j = 0; j < PPT; j++) {
          const int p = p0 + j;
          if (p < N) {
            const double px = L.sx[p], py = L.sy[p];
            const double w0 = pf_weight(px, py, mx0, my0, mth0, al0, rg0, L.etab, status);
            const double w1 = pf_weight(px, py, mx1, my1, mth1, al1, rg1, L.etab, status);
            wq[p] = w0; wq1[p] = w1;
            lmax0 = w0 > lmax0 ? w0 : lmax0;
            lmax1 = w1 > lmax1 ? w1 : lmax1;
          }
        }
        pf_block_max2<T>(lmax0, lmax1, L.red_d, tid);
#pragma unroll
        for (int j = 0; j < PPT; j++) if (p0 + j < N) { nw[j] += (1 / lmax0) * wq[p0 + j]; nw[j] += (1 / lmax1) * wq1[p0 + j]; }
      }
#endif
      for (; a < A; a++) {
        const double* m = mbase + (size_t)a * 5;
        const double mx = m[0], my = m[1], mth = m[2], auv_alpha = m[3], auv_range = m[4];
        double lmax = -__builtin_inf();
#pragma unroll 1
        for (int j = 0; j < PPT; j++) {
          const int p = p0 + j;
          if (p < N) {
            const double w = pf_weight(L.sx[p], L.sy[p], mx, my, mth, auv_alpha, auv_range, L.etab, status);
            wq[p] = w;   // (read back below by the thread that wrote it: no barrier needed for wq itself)
            lmax = w > lmax ? w : lmax;
          }
        }
        const double den = pf_block_max<T>(lmax, L.red_d, tid);
        // normalize (:133-139) in place
#pragma unroll
        for (int j = 0; j < PPT; j++) if (p0 + j < N) nw[j] += (1 / den) * wq[p0 + j];
      }
      PF_STAMP(2)
      double lmax = -__builtin_inf();
#pragma unroll
      for (int j = 0; j < PPT; j++) if (p0 + j < N) lmax = nw[j] > lmax ? nw[j] : lmax;
      const double fden = pf_block_max<T>(lmax, L.red_d, tid);
      // ---- correct (:179-252): 1..5 deep copies by weight class, then N index draws
      int k[PPT], ksum = 0;
#pragma unroll
      for (int j = 0; j < PPT; j++) {
        const int p = p0 + j;
        k[j] = 0;
        if (p < N) {
          const double w = (1 / fden) * nw[j];
          L.sw[p] = w;
          k[j] = w < 0.2 ? 1 : (w < 0.4 ? 2 : (w < 0.6 ? 3 : (w < .8 ? 4 : (w <= 1.0 ? 5 : 0))));
          ksum += k[j];
        }
      }
      int total = 0;
      int run = pf_block_scan<T>(ksum, L.red_i, tid, &total);
      // copy c of list_of_new_particles -> the particle it was copied from (at most five entries per particle)
      int* inv = L.slot;
#pragma unroll
      for (int j = 0; j < PPT; j++) if (p0 + j < N) { for (int c = 0; c < k[j]; c++) inv[run + c] = p0 + j; run += k[j]; }
      const int len = total;
      __syncthreads();
      if (len == 0) { status = PF_ERR_EMPTY; break; }  // numpy raises ValueError
      PF_STAMP(3)
      int* cho = L.off;
      if (len == 1) {
        _Pragma("unroll 1") for (int i = tid; i < N; i += T) cho[i] = 0;  // randint(0, 1): no draw
        __syncthreads();
      } else {
        const uint32_t rng = (uint32_t)len - 1u;
        uint32_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        int got = 0;
        while (got < N) {
          if (r.pos >= 624) { pf_refill(r, tid); __syncthreads(); }
          const int avail = 624 - r.pos;
          // thread t looks at words [WPT t, WPT t + WPT) of the block (WPT T >= 624)
          constexpr int WPT = (624 + T - 1) / T;
          uint32_t val[WPT];
          int cnt = 0;
#pragma unroll
          for (int c = 0; c < WPT; c++) {
            const int i = WPT * tid + c;
            val[c] = 0xffffffffu;
            if (i < avail) { const uint32_t v = mt_temper(r.mt[r.pos + i]) & mask; if (v <= rng) { val[c] = v; cnt++; } }
          }
          int tot = 0;
          int rank = got + pf_block_scan<T>(cnt, L.red_i, tid, &tot);
          if (tid == 0) L.red_i[16] = avail - 1;
          __syncthreads();
#pragma unroll
          for (int c = 0; c < WPT; c++) {
            if (val[c] != 0xffffffffu) {
              if (rank < N) cho[rank] = (int)val[c];
              if (rank == N - 1) L.red_i[16] = WPT * tid + c;  // the word that completes the N draws
              rank++;
            }
          }
          __syncthreads();
          const int used = L.red_i[16] + 1;
          r.pos += used;
          r.drawn += (unsigned long long)used;
          got = min(N, got + tot);
          __syncthreads();
        }
      }
      PF_STAMP(4)
      if (D.choice) for (int i = tid; i < N; i += T) D.choice[((size_t)s * D.F + f) * N + i] = cho[i];
      // ---- the new list: position n = copy of the source of list_of_new_particles[cho[n]]
      // in two halves (x, y / v, theta, weight): fewer doubles live across each barrier (all five spilled four of them)
      int lo[PPT];
#pragma unroll
      for (int j = 0; j < PPT; j++) lo[j] = (p0 + j < N) ? inv[cho[p0 + j]] : 0;
      {
        double gx[PPT], gy[PPT];
#pragma unroll
        for (int j = 0; j < PPT; j++) if (p0 + j < N) { gx[j] = L.sx[lo[j]]; gy[j] = L.sy[lo[j]]; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PPT; j++) if (p0 + j < N) { L.sx[p0 + j] = gx[j]; L.sy[p0 + j] = gy[j]; }
      }
      {
        double gv[PPT], gt[PPT], gw[PPT];
#pragma unroll
        for (int j = 0; j < PPT; j++) if (p0 + j < N) { gv[j] = L.sv[lo[j]]; gt[j] = L.sth[lo[j]]; gw[j] = L.sw[lo[j]]; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PPT; j++) if (p0 + j < N) { L.sv[p0 + j] = gv[j]; L.sth[p0 + j] = gt[j]; L.sw[p0 + j] = gw[j]; }
      }
      llen = len;
      if (tid == 0) D.out_len[(size_t)s * D.F + f] = len;
      __syncthreads();
      PF_STAMP(5)
    }

    if (D.phases & PF_PHASE_MEAN) {
      // ---- particleMean (:153-167): left-to-right sums, x on wave 0 and y on wave 1
      if (tid == 0 || tid == 64) {
        const double* src = tid == 0 ? L.sx : L.sy;
        double sum = 0;
        int i = 0;
#ifndef AUVP_PF_MEAN_PIPE
        for (; i + 16 <= N; i += 16) {  // 16 LDS reads in flight, then the 16 dependent adds in list order
          double v[16];
#pragma unroll
          for (int k = 0; k < 16; k++) v[k] = src[i + k];
#pragma unroll
          for (int k = 0; k < 16; k++) sum += v[k];
        }
#else
        // the MB dependent adds of one block run while the next block's MB LDS reads are in flight (two register blocks)
#ifndef AUVP_PF_MEAN_BLOCK
#define AUVP_PF_MEAN_BLOCK 8
#endif
        constexpr int MB = AUVP_PF_MEAN_BLOCK;
        double va[MB], vb[MB];
        if (N >= MB) {
#pragma unroll
          for (int k = 0; k < MB; k++) va[k] = src[k];
        }
#pragma unroll 1
        for (; i + 2 * MB <= N; i += 2 * MB) {
#pragma unroll
          for (int k = 0; k < MB; k++) vb[k] = src[i + MB + k];
#pragma unroll
          for (int k = 0; k < MB; k++) sum += va[k];
          if (i + 3 * MB <= N) {
#pragma unroll
            for (int k = 0; k < MB; k++) va[k] = src[i + 2 * MB + k];
          }
#pragma unroll
          for (int k = 0; k < MB; k++) sum += vb[k];
        }
        if (i + MB <= N) {  // va holds [i, i + MB): loaded up front (i = 0) or by the last trip
#pragma unroll
          for (int k = 0; k < MB; k++) sum += va[k];
          i += MB;
        }
#endif
        for (; i < N; i++) sum += src[i];
        L.red_d[16 + (tid >> 6)] = sum / N;
      }
      __syncthreads();
      if (tid == 0) {
        const double xm = L.red_d[16], ym = L.red_d[17];
        const double* sh = D.shark + ((size_t)s * D.F + f) * 2;
        const double xd = xm - sh[0], yd = ym - sh[1];
        D.mean[((size_t)s * D.F + f) * 2] = xm;
        D.mean[((size_t)s * D.F + f) * 2 + 1] = ym;
        D.err[(size_t)s * D.F + f] = auvp_sqrt((xd * xd) + (yd * yd));  // meanError (:169-177)
      }
      __syncthreads();
      PF_STAMP(6)
    }
  }

  // ---- persist
  _Pragma("unroll 1") for (int i = tid; i < 624; i += T) D.mt[(size_t)f * 624 + i] = r.mt[i];
  _Pragma("unroll 1") for (int i = tid; i < N; i += T) {
    st[i] = L.sx[i]; st[N + i] = L.sy[i]; st[2 * N + i] = L.sv[i]; st[3 * N + i] = L.sth[i]; st[4 * N + i] = L.sw[i];
  }
#pragma unroll
  for (int j = 0; j < PPT; j++) if (p0 + j < N) D.ent[(size_t)f * N + p0 + j] = L.off[p0 + j];
  if (__syncthreads_or(status != PF_OK) && tid == 0 && D.status[f] == PF_OK) D.status[f] = status != PF_OK ? status : PF_ERR_ANGLE;
  if (tid == 0) { D.mtpos[f] = r.pos; D.llen[f] = llen; D.ndraw[f] += r.drawn; }
#ifdef AUVP_PF_DIAG
  if (tid == 0) for (int k = 0; k < 7 && k < D.S; k++) D.err[(size_t)k * D.F + f] = (double)pf_acc[k];
#endif
#undef PF_STAMP
}

}  // namespace auvp
#endif
