// auvp_wave.h -- wave64 building blocks for the gfx950 planner kernels.
//
// One wavefront (64 lanes) runs one planning episode.  Everything an episode does in order
// (RNG stream, tree growth) is wave-uniform control flow; the lanes split the work inside one
// expansion.  The helpers here are the cross-lane pieces: an in-LDS MT19937 that the whole wave
// refills cooperatively, ordered reductions, readlane for doubles.
#ifndef AUVP_WAVE_H
#define AUVP_WAVE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace auvp {

constexpr int WAVE = 64;

// LDS traffic between lanes of ONE wave: the hardware keeps a wave's DS operations in order; this
// only stops the compiler from moving LDS accesses across the hand-off point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// Words in LDS that wavefronts of one workgroup hand to each other (flags, sequence numbers, tags).  A volatile access through a
// generic pointer stays a FLAT instruction (address-space inference leaves volatile accesses alone): `flat_load ... sc0 sc1`
// + `s_waitcnt vmcnt(0) lgkmcnt(0)` per poll -- every look at a flag also waits for the wavefront's outstanding global loads
// and stores.  With the explicit LDS pointer they are ds_read / ds_write, which wait on the LDS counter alone.
#define AUVP_LDS_PTR(T, p) ((volatile __attribute__((address_space(3))) T*)(p))
// Wavefront votes on a BOOLEAN: HIP's __any / __all / __ballot take an int, so a predicate is first turned into 0 / 1 in a
// vector register and then compared with zero again -- two vector instructions per vote on kernels that are bound by
// vector issue and vote dozens of times per iteration.  The builtin takes the i1 as it is: the vote is the compare itself
// (and an s_and with exec).
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }

// a speculative pipeline gave up (a bounded wait ran out: status AUVP_ST_PIPELINE): tell the host through its mapped flag, so
// that it re-runs the batch on the one-wavefront kernel (auvplan.hip: pipeline fallback) -- the status alone would need a
// read-back of every summary after every launch
#define AUVP_ST_GENERATOR (-7)
#define AUVP_ST_PIPELINE (-9)
#define AUVP_ST_STREAM (-10)  // an episode ran past its pre-generated random stream (rrt_rows_stream_kernel.h): the batch is redone by rrt_rows_kernel
__device__ __forceinline__ void pipe_report(int32_t* flag, int status) {
  if (status == AUVP_ST_PIPELINE && flag) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// polls before a bounded wait of a speculative pipeline gives up (a protocol bug or a descheduled wavefront must not hang
// the GPU): ~seconds.  -DAUVP_PIPE_DIAG builds (libauvplan_diag.so, tests only) take the limit and a delay injection from a
// device table the host fills at auvp_create from AUVP_DIAG_SPIN / AUVP_DIAG_JITTER:
//   [0] spin limit (0: the default)   [1] jitter mode: 0 off, 1 wavefronts with wave % [2] == [3], 2 wavefronts with wave / [2] == [3]
//   [4] longest delay in units of s_sleep 1 (~64 clocks)   [5] seed
// pipe_jitter() runs before every hand-over word (lds_poke / lds_poke64): a pseudo-random delay of 0 .. [4] units on the chosen
// stage's wavefronts -- it moves the relative speed of the stages around, which is what the redo paths depend on.
#define AUVP_PIPE_SPIN_DEFAULT (1 << 24)
#ifdef AUVP_PIPE_DIAG
__device__ int auvp_diag_cfg[8];
__device__ __forceinline__ int pipe_spin_limit() { const int v = auvp_diag_cfg[0]; return v > 0 ? v : AUVP_PIPE_SPIN_DEFAULT; }
__device__ __forceinline__ void pipe_jitter() {
  const int mode = auvp_diag_cfg[1];
  if (!mode) return;
  const int wave = (int)(threadIdx.x >> 6), mod = auvp_diag_cfg[2] > 0 ? auvp_diag_cfg[2] : 1;
  if ((mode == 1 ? wave % mod : wave / mod) != auvp_diag_cfg[3]) return;
  uint32_t x = (uint32_t)__builtin_amdgcn_s_memtime() * 2654435761u ^ ((uint32_t)auvp_diag_cfg[5] + blockIdx.x * 0x9e3779b9u + (uint32_t)wave * 0x85ebca6bu);
  x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
  const int n = (int)(x % (uint32_t)(auvp_diag_cfg[4] + 1));
  for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(1);
}
#else
__device__ __forceinline__ constexpr int pipe_spin_limit() { return AUVP_PIPE_SPIN_DEFAULT; }
__device__ __forceinline__ void pipe_jitter() {}
#endif

__device__ __forceinline__ int lds_peek(const int* p) { return *AUVP_LDS_PTR(const int, p); }
__device__ __forceinline__ void lds_poke(int* p, int v) { pipe_jitter(); *AUVP_LDS_PTR(int, p) = v; }
__device__ __forceinline__ unsigned long long lds_peek64(const unsigned long long* p) { return *AUVP_LDS_PTR(const unsigned long long, p); }
__device__ __forceinline__ void lds_poke64(unsigned long long* p, unsigned long long v) { pipe_jitter(); *AUVP_LDS_PTR(unsigned long long, p) = v; }

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  long long b = __double_as_longlong(v);
  int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src_lane);
  int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double readfirst_f64(double v) {
  long long b = __double_as_longlong(v);
  int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
  int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// wave-wide min / max over the lanes (values identical on return) on the DPP path: four rotate steps inside the 16-lane rows,
// then the four row results.  (Rounds 1-3 used six xor shuffles: `__shfl_xor` of a double is two ds_bpermute, an LDS round trip
// per step -- ~100 clocks each on a chain nothing else covers.)  Same comparison as before, (t < v) ? t : v; the order in which
// lanes meet differs, which matters only with a nan among the values.
#define AUVP_ROW_ROR_F64(v, N, out)                                                                           \
  do {                                                                                                        \
    const long long b__ = __double_as_longlong(v);                                                            \
    const int lo__ = __builtin_amdgcn_update_dpp(0, (int)(b__ & 0xffffffffll), 0x120 + (N), 0xf, 0xf, false); \
    const int hi__ = __builtin_amdgcn_update_dpp(0, (int)(b__ >> 32), 0x120 + (N), 0xf, 0xf, false);          \
    out = __longlong_as_double(((long long)hi__ << 32) | (unsigned int)lo__);                                 \
  } while (0)
__device__ __forceinline__ double wave_min_f64(double v) {
  double t;
  AUVP_ROW_ROR_F64(v, 8, t); v = (t < v) ? t : v;
  AUVP_ROW_ROR_F64(v, 4, t); v = (t < v) ? t : v;
  AUVP_ROW_ROR_F64(v, 2, t); v = (t < v) ? t : v;
  AUVP_ROW_ROR_F64(v, 1, t); v = (t < v) ? t : v;
  const double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
  const double ab = (b < a) ? b : a, cd = (d < c) ? d : c;
  return (cd < ab) ? cd : ab;
}
__device__ __forceinline__ double wave_min_f64_dpp(double v) { return wave_min_f64(v); }

__device__ __forceinline__ double wave_max_f64(double v) {
  double t;
  AUVP_ROW_ROR_F64(v, 8, t); v = (t > v) ? t : v;
  AUVP_ROW_ROR_F64(v, 4, t); v = (t > v) ? t : v;
  AUVP_ROW_ROR_F64(v, 2, t); v = (t > v) ? t : v;
  AUVP_ROW_ROR_F64(v, 1, t); v = (t > v) ? t : v;
  const double a = readlane_f64(v, 0), b = readlane_f64(v, 16), c = readlane_f64(v, 32), d = readlane_f64(v, 48);
  const double ab = (b > a) ? b : a, cd = (d > c) ? d : c;
  return (cd > ab) ? cd : ab;
}
#undef AUVP_ROW_ROR_F64

// sum over the 64 lanes in a fixed but unspecified order (NOT for sums whose rounding the reference defines): four
// rotate-and-add steps inside the 16-lane rows on the DPP path (no LDS crossbar), then the four row totals
__device__ __forceinline__ double wave_sum_any_order(double v) {
#define AUVP_ROW_ROR_ADD(N)                                                                                 \
  do {                                                                                                      \
    const long long b__ = __double_as_longlong(v);                                                          \
    const int lo__ = __builtin_amdgcn_update_dpp(0, (int)(b__ & 0xffffffffll), 0x120 + (N), 0xf, 0xf, false); \
    const int hi__ = __builtin_amdgcn_update_dpp(0, (int)(b__ >> 32), 0x120 + (N), 0xf, 0xf, false);         \
    v = v + __longlong_as_double(((long long)hi__ << 32) | (unsigned int)lo__);                             \
  } while (0)
  AUVP_ROW_ROR_ADD(8);
  AUVP_ROW_ROR_ADD(4);
  AUVP_ROW_ROR_ADD(2);
  AUVP_ROW_ROR_ADD(1);
#undef AUVP_ROW_ROR_ADD
  return ((readlane_f64(v, 0) + readlane_f64(v, 16)) + readlane_f64(v, 32)) + readlane_f64(v, 48);
}

// ---------------------------------------------------------------------------------------------
// CPython-compatible MT19937 stream, state in LDS (624 words per wave), refilled lazily IN PLACE:
// logical word q overwrites the slot of word q-624 as soon as that one has been consumed, so up to
// 624 words ahead of the consumer are available without a second buffer and a refill step is one
// read-3/write-1 pass of the whole wave (64 words per step; dependencies reach back 227 words, so
// 64 consecutive words are independent of each other).
//   new[q] = s[(q+397)%624] ^ twist(upper(s[q%624]) | lower(s[(q+1)%624]))
// Replaces the `random` module calls of the reference (random.uniform / random.choice:
// path_planning/rrt_dubins.py:123-129,259-279,336-339; gym_rrt/envs/rrt_dubins.py:186,223,262-267).
// ---------------------------------------------------------------------------------------------
struct WaveRng {
  uint32_t* s;     // LDS, 624 words
  uint32_t pslot;  // slot of the next unconsumed word
  uint32_t avail;  // generated, not yet consumed
  unsigned long long drawn;  // 32-bit outputs consumed so far
};

// random.random() from two tempered outputs: (a * 2^26 + b) / 2^53 with a < 2^27, b < 2^26.  Every step of that expression is
// exact (a 53-bit integer, then a power-of-two scale), so any exact evaluation gives the same double: a * 2^-27 and b * 2^-53
// are exact, their sum is a multiple of 2^-53 below 1 -- one multiply and one fused multiply-add instead of multiply, add, multiply.
__device__ __forceinline__ double py_random_from(uint32_t a27, uint32_t b26) {
  return __builtin_fma((double)a27, 0x1p-27, (double)b26 * 0x1p-53);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

__device__ __forceinline__ void rng_ensure(WaveRng& r, uint32_t need_words) {
  const uint32_t lane = (uint32_t)lane_id();
  while (r.avail < need_words) {
    uint32_t n = 624u - r.avail;
    n = n < 64u ? n : 64u;
    uint32_t k = r.pslot + r.avail + lane;
    k = k >= 624u ? k - 624u : k;
    k = k >= 624u ? k - 624u : k;
    uint32_t k1 = (k + 1u == 624u) ? 0u : k + 1u;
    uint32_t km = k + 397u;
    km = km >= 624u ? km - 624u : km;
    uint32_t a = r.s[k], b = r.s[k1], c = r.s[km];
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    uint32_t v = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    wave_sync();  // every lane's reads are issued before any lane's write
    if (lane < n) r.s[k] = v;
    wave_sync();
    r.avail = (uint32_t)uni((int)(r.avail + n));
  }
}

// tempered 32-bit output number `j` ahead of the consumer (j < avail)
__device__ __forceinline__ uint32_t rng_word(const WaveRng& r, uint32_t j) {
  uint32_t k = r.pslot + j;      // pslot < 624 and j < avail <= 624: one wrap is enough
  k = k >= 624u ? k - 624u : k;
  return mt_temper(r.s[k]);
}

// random.random() number `j` ahead (consumes words 2j, 2j+1)
__device__ __forceinline__ double rng_random_at(const WaveRng& r, uint32_t j) {
  uint32_t a = rng_word(r, 2u * j) >> 5, b = rng_word(r, 2u * j + 1u) >> 6;
  return py_random_from(a, b);
}

__device__ __forceinline__ void rng_advance_words(WaveRng& r, uint32_t nwords) {
  uint32_t p = (uint32_t)uni((int)(r.pslot + nwords));  // scalar: the wrap below stays off the vector unit
  while (p >= 624u) p -= 624u;
  r.pslot = (uint32_t)uni((int)p);
  r.avail = (uint32_t)uni((int)(r.avail - nwords));
  r.drawn += nwords;
}

// one uniform random() for the whole wave (every lane returns the same value)
__device__ __forceinline__ double rng_next_random(WaveRng& r) {
  rng_ensure(r, 2u);
  double v = rng_random_at(r, 0u);
  rng_advance_words(r, 2u);
  return v;
}

// random.uniform(a, b) = a + (b-a) * random()
// A literal 0.0 lower end (most calls): 0.0 + (b - 0.0) * u = RN(b * u) + 0.0; b - 0.0 is b and adding +0.0 to a rounded product
// is exact, so ONE fused multiply-add with a +0.0 addend returns the same double, sign of zero included ((-0.0) + (+0.0) = +0.0
// both ways) -- one vector instruction instead of two.
__device__ __forceinline__ double py_uniform(double a, double b, double u) {
  if (__builtin_constant_p(a) && a == 0.0) return __builtin_fma(b, u, 0.0);
  return a + (b - a) * u;
}

}  // namespace auvp
#endif
