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

AUVP_HD double auvp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
AUVP_HD double auvp_sqrt(double a) { return __builtin_sqrt(a); }
AUVP_HD double auvp_fabs(double a) { return __builtin_fabs(a); }
AUVP_HD double auvp_rint(double a) { return __builtin_rint(a); }
AUVP_HD double auvp_floor(double a) { return __builtin_floor(a); }

// pi/2 = PIO2_HI + PIO2_LO + PIO2_LO2 (each the nearest double to the running remainder)
#define AUVP_PIO2_HI 0x1.921fb54442d18p+0
#define AUVP_PIO2_LO 0x1.1a62633145c07p-54
#define AUVP_PIO2_LO2 (-0x1.f1976b7ed8fbcp-110)
#define AUVP_INV_PIO2 0x1.45f306dc9c883p-1
#define AUVP_PI 0x1.921fb54442d18p+1

// x = n*(pi/2) + (r + t), |r+t| <= ~pi/4; returns n mod 4 in [0,3].
// hi = x - n*PIO2_HI is exact (single fma rounding of a value that fits 53 bits); the PIO2_LO
// product is split exactly with a second fma, so r+t carries ~100 bits of the reduced argument
// for |x| up to ~1e9.  Beyond that accuracy degrades gracefully (still deterministic).
AUVP_HD int auvp_rem_pio2(double x, double* r, double* t) {
  double n = auvp_rint(x * AUVP_INV_PIO2);
  double hi = auvp_fma(-n, AUVP_PIO2_HI, x);
  double p = n * AUVP_PIO2_LO;
  double pe = auvp_fma(n, AUVP_PIO2_LO, -p);  // exact error of p
  double rr = hi - p;
  double tt = (hi - rr) - p;                  // rounding error of the subtraction
  tt = tt - pe;
  tt = auvp_fma(-n, AUVP_PIO2_LO2, tt);
  *r = rr;
  *t = tt;
  // n is integral and |n| < 2^53: take it modulo 4 without leaving fp64
  double q = n - 4.0 * auvp_floor(n * 0.25);
  return (int)q;
}

// sin(x + y) for |x| <= ~pi/4 with tail y
AUVP_HD double auvp_ksin(double x, double y) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

// cos(x + y) for |x| <= ~pi/4 with tail y
AUVP_HD double auvp_kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}

AUVP_HD void auvp_sincos(double x, double* s, double* c) {
  double r, t;
  int q = auvp_rem_pio2(x, &r, &t);
  double ks = auvp_ksin(r, t);
  double kc = auvp_kcos(r, t);
  double ss = (q & 1) ? kc : ks;
  double cc = (q & 1) ? ks : kc;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}

AUVP_HD double auvp_sin(double x) {
  double s, c;
  auvp_sincos(x, &s, &c);
  return s;
}

AUVP_HD double auvp_cos(double x) {
  double s, c;
  auvp_sincos(x, &s, &c);
  return c;
}

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
  if (ay == 0.0) return ax;
  double h = auvp_sqrt(auvp_fma(ax, ax, ay * ay));
  double h2 = h * h;
  double ax2 = ax * ax;
  double r = auvp_fma(-ay, ay, h2 - ax2) + auvp_fma(h, h, -h2) - auvp_fma(ax, ax, -ax2);
  return h - r / (2.0 * h);
}

#endif  // AUVP_MATH_H
