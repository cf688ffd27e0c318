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
AUVP_HD double auvp_atan(double x) {
  const double hi0 = 4.63647609000806093515e-01, hi1 = 7.85398163397448278999e-01,
               hi2 = 9.82793723247329054082e-01, hi3 = 1.57079632679489655800e+00;
  const double lo0 = 2.26987774529616870924e-17, lo1 = 3.06161699786838301793e-17,
               lo2 = 1.39033110312309984516e-17, lo3 = 6.12323399573676603587e-17;
  const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01,
               a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01,
               a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
               a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02,
               a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02,
               a10 = 1.62858201153657823623e-02;
  const int neg = x < 0.0;
  double ax = auvp_fabs(x);
  if (x != x) return x;
  if (ax >= 0x1p66) {  // |x| >= 2^66: pi/2
    double r = hi3 + lo3;
    return neg ? -r : r;
  }
  // one division for whichever reduction applies: the lanes of a wavefront usually need different ones, and four divergent
  // branches with a division each would run one after the other (same operations per lane either way)
  int id;
  double hi = 0.0, lo = 0.0, num = 0.0, den = 1.0;
  if (ax < 0.4375) {
    if (ax < 0x1p-27) return x;
    id = -1;
  } else if (ax < 0.6875) { id = 0; num = 2.0 * ax - 1.0; den = 2.0 + ax; hi = hi0; lo = lo0; }
  else if (ax < 1.1875) { id = 1; num = ax - 1.0; den = ax + 1.0; hi = hi1; lo = lo1; }
  else if (ax < 2.4375) { id = 2; num = ax - 1.5; den = 1.0 + 1.5 * ax; hi = hi2; lo = lo2; }
  else { id = 3; num = -1.0; den = ax; hi = hi3; lo = lo3; }
  if (id >= 0) ax = num / den;
  double z = ax * ax;
  double w = z * z;
  double s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
  double s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
  if (id < 0) {
    double r = ax - ax * (s1 + s2);
    return neg ? -r : r;
  }
  double r = hi - ((ax * (s1 + s2) - lo) - ax);
  return neg ? -r : r;
}

// atan2(y, x) with the usual quadrant logic (finite inputs; infinities map through atan's limits).
// Measured <= 1.2 ulp vs mpmath (the y/x rounding adds to atan's own error); glibc's atan2 is
// nearly correctly rounded, so the two agree bit-for-bit on ~80 % of inputs and differ by one ulp
// otherwise -- used only by Planner_RRT's goal connection, where it feeds floats, one
// `abs(diff) > pi/2` test and one floor(length/exp_rate).
AUVP_HD double auvp_atan2(double y, double x) {
  const double pi = AUVP_PI, pi_lo = 1.2246467991473531772E-16, pio2 = AUVP_PIO2_HI;
  if (x != x || y != y) return x + y;
  if (y == 0.0) {
    // atan2(+-0, +x) = +-0 ; atan2(+-0, -x) = +-pi
    int xneg = (x < 0.0) || (x == 0.0 && __builtin_signbit(x));
    if (!xneg) return y;
    return __builtin_signbit(y) ? -pi : pi;
  }
  if (x == 0.0) return y < 0.0 ? -pio2 : pio2;
  double ax = auvp_fabs(x), ay = auvp_fabs(y);
  double z;
  if (ax == __builtin_inf() || ay == __builtin_inf()) {
    if (ax == ay) z = 0.5 * pio2;            // pi/4
    else if (ay == __builtin_inf()) z = pio2;
    else z = 0.0;
  } else {
    double q = ay / ax;
    if (q >= 0x1p64) z = pio2 + 0.5 * pi_lo;
    else if (x < 0.0 && q < 0x1p-64) z = 0.0;
    else z = auvp_atan(q);
  }
  if (x > 0.0) return y < 0.0 ? -z : z;
  double r = pi - (z - pi_lo);
  return y < 0.0 ? -r : r;
}

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
