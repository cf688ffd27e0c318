// auvp_math.h -- portable fp64 elementary functions shared by the HIP kernels (device) and by the
// "portable" build of the CPU checker (oracle/, host).  Every operation is an explicit IEEE-754
// binary64 add/mul/div/sqrt/fma, so a host build with -ffp-contract=off and a gfx950 build with
// -ffp-contract=off produce bit-identical results.  That is what makes the kernel <-> checker
// comparison bit-exact (DESIGN.md "Numerics").
//
// Why not libm / ocml: the reference calls CPython math.sin/cos/atan2/hypot (glibc, < 1 ulp, not
// correctly rounded) at path_planning/rrt_dubins.py:275-276,280 and gym_rrt/envs/rrt_dubins.py:
// 276-277,387-402.  ocml's fp64 sin/cos differ from glibc's in the last bit on some inputs, and a
// last-bit difference can flip an accept/reject comparison.  These functions are < 1 ulp too
// (measured in tests/test_portable_math.py against mpmath), so they agree with glibc except for
// rare last-bit cases; tree decisions (parents, accept flags, grid indices) are unaffected and
// floats agree to ~1e-15 relative.
#ifndef AUVP_MATH_H
#define AUVP_MATH_H

#if defined(__HIPCC__)
#define AUVP_HD __host__ __device__ __forceinline__
#else
#define AUVP_HD static inline
#endif

// AUVP_K(c): how a reduction constant reaches its use; AUVP_FMA_K(a, b, c) = fma(a, b, c) with a CONSTANT addend -- a Horner
// step.  Plain: the literal / __builtin_fma.  The compiler selects the two-address v_fmac_f64 for such a step and therefore
// first copies the constant into the destination (v_mov_b64, or two v_mov_b32 out of scalar registers), and a kernel whose
// scalar registers are full keeps the ~16 constants of a sin / cos in VECTOR registers for the whole loop.  The `_sk`
// instantiation of the sin / cos family (auvp_sincos_sk: device code only, below) forms a constant in a scalar register pair
// right where it is used -- two s_mov_b32 inside an asm, which nothing hoists -- and spells a Horner step as those two moves into
// a FIXED scalar pair + ONE v_fma_f64 that reads it.  Same doubles either way.  Measured (profiles/r5_valu_issue.md, DESIGN.md
// section 0): worth 1-2.5 % in the RRT.exploring kernels (vector-issue bound), -2..-5 % in the planner and particle-filter
// kernels (short of SCALAR issue: a scalar move costs a wavefront as much as an fp64 instruction there) -- hence two names.
#if defined(__HIP_DEVICE_COMPILE__)
template <unsigned LO, unsigned HI>
__device__ __forceinline__ double auvp_sgpr_f64() {
  unsigned a, b;
  __asm__ volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(a), "=s"(b) : "i"(LO), "i"(HI));
  return __builtin_bit_cast(double, ((unsigned long long)b << 32) | (unsigned long long)a);
}
template <unsigned LO, unsigned HI>
__device__ __forceinline__ double auvp_fma_sgpr_k(double a, double b) {
  double d;
  __asm__("s_mov_b32 s96, %3\n\ts_mov_b32 s97, %4\n\tv_fma_f64 %0, %1, %2, s[96:97]" : "=v"(d) : "v"(a), "v"(b), "i"(LO), "i"(HI) : "s96", "s97");
  return d;
}
#define AUVP_K_BITS(c) __builtin_bit_cast(unsigned long long, (double)(c))
#endif
#define AUVP_K(c) (c)
#define AUVP_FMA_K(a, b, c) __builtin_fma(a, b, c)

AUVP_HD double auvp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
AUVP_HD double auvp_sqrt(double a) { return __builtin_sqrt(a); }
AUVP_HD double auvp_fabs(double a) { return __builtin_fabs(a); }
AUVP_HD double auvp_rint(double a) { return __builtin_rint(a); }
AUVP_HD double auvp_floor(double a) { return __builtin_floor(a); }

// a / b and sqrt(x) for operands of UNEXCEPTIONAL magnitude -- the steer's quotients and chord lengths (rrt_dubins.py:268-281):
// |a|, |b|, |a / b| within 2^+-500 (a may be +0), b != 0; x = 0 or within 2^+-500.  Host: the IEEE operation.  Device: the
// same correctly rounded result from the core of the compiler's own expansion -- reciprocal (square root) estimate, two
// Newton steps, one residual correction -- without the steps that only serve exponents near the ends of the range
// (v_div_scale x 2 + v_div_fixup: 8 instead of 11 vector instructions; the input scaling of sqrt: 13 instead of 19).  With a
// zero divisor the reference raises ZeroDivisionError (a 2^-53 draw); host and device both return a non-finite value then.
AUVP_HD double auvp_div_plain(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
  const double q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
#else
  // (a zero divisor: the device form above gives NaN -- rcp(0) = inf through the Newton steps -- so the host form does too;
  // IEEE's +-inf would make the checker and the device differ in WHICH non-finite value the ZeroDivisionError path carries)
  return b == 0.0 ? __builtin_nan("") : a / b;
#endif
}
AUVP_HD double auvp_sqrt_plain(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  return x == 0.0 ? x : g;
#else
  return __builtin_sqrt(x);
#endif
}

// pi/2 = PIO2_HI + PIO2_LO + PIO2_LO2 (each the nearest double to the running remainder)
#define AUVP_PIO2_HI 0x1.921fb54442d18p+0
#define AUVP_PIO2_LO 0x1.1a62633145c07p-54
#define AUVP_PIO2_LO2 (-0x1.f1976b7ed8fbcp-110)
#define AUVP_INV_PIO2 0x1.45f306dc9c883p-1
#define AUVP_PI 0x1.921fb54442d18p+1

// the sin / cos family: auvp_rem_pio2, auvp_ksin, auvp_kcos, auvp_sincos, auvp_sin, auvp_cos (auvp_sincos_body.h)
#define AUVP_SC(n) n
#include "auvp_sincos_body.h"
#undef AUVP_SC
#if defined(__HIPCC__)
// ... and again as auvp_sincos_sk etc. for device code whose constants are formed in scalar registers at their use (above);
// the host pass of a HIP unit compiles the plain form under these names
#define AUVP_SC(n) n##_sk
#if defined(__HIP_DEVICE_COMPILE__)
#undef AUVP_K
#undef AUVP_FMA_K
#define AUVP_K(c) auvp_sgpr_f64<(unsigned)(AUVP_K_BITS(c) & 0xffffffffull), (unsigned)(AUVP_K_BITS(c) >> 32)>()
#define AUVP_FMA_K(a, b, c) auvp_fma_sgpr_k<(unsigned)(AUVP_K_BITS(c) & 0xffffffffull), (unsigned)(AUVP_K_BITS(c) >> 32)>(a, b)
#endif
#include "auvp_sincos_body.h"
#undef AUVP_SC
#undef AUVP_K
#undef AUVP_FMA_K
#define AUVP_K(c) (c)
#define AUVP_FMA_K(a, b, c) __builtin_fma(a, b, c)
#endif

// atan(x), fdlibm-class: argument reduction to [0, 7/16] around 0.5, 1, 1.5, inf; odd polynomial
// split into even/odd halves.  < 1 ulp (tests/test_portable_math.py).
#define AUVP_ATAN_FN auvp_atan
#define AUVP_ATAN2_FN auvp_atan2
#define AUVP_ATAN_K(c) (c)
#include "auvp_atan_body.h"
#undef AUVP_ATAN_FN
#undef AUVP_ATAN2_FN
#undef AUVP_ATAN_K

// hypot(x, y) for finite, unexceptional magnitudes (|x|,|y| in ~[1e-140, 1e140]): Borges' fused
// formulation -- sqrt of the fma'd sum of squares plus one exact-residual correction.
AUVP_HD double auvp_hypot(double x, double y) {
  double ax = auvp_fabs(x), ay = auvp_fabs(y);
  if (ax < ay) { double t = ax; ax = ay; ay = t; }
  if (ax == 0.0) return ax;  // (both zero: nothing to divide by; a zero smaller operand falls out of the formula: h = ax, r = 0)
  double h = auvp_sqrt_plain(auvp_fma(ax, ax, ay * ay));
  if (h == 0.0) return ax;   // (ax^2 underflowed, |ax| < ~2^-511: the correction below would be 0 / 0)
  double h2 = h * h;
  double ax2 = ax * ax;
  double r = auvp_fma(-ay, ay, h2 - ax2) + auvp_fma(h, h, -h2) - auvp_fma(ax, ax, -ax2);
  return h - auvp_div_plain(r, 2.0 * h);
}

#endif  // AUVP_MATH_H
