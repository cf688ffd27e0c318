// auvp_sincos_body.h -- the bodies of the sin / cos family of auvp_math.h, included once per instantiation:
//   AUVP_SC(name)         the function names (plain: name; scalar-constant instantiation: name_sk)
//   AUVP_K(c)             how a reduction constant reaches its use
//   AUVP_FMA_K(a, b, c)   fma(a, b, c) with a CONSTANT addend -- a Horner step
// Same operations in the same order either way: the same doubles (auvp_math.h has the why).
// x = n*(pi/2) + (r + t), |r+t| <= ~pi/4; returns n mod 4 in [0,3].
// hi = x - n*PIO2_HI is exact (single fma rounding of a value that fits 53 bits); the PIO2_LO
// product is split exactly with a second fma, so r+t carries ~100 bits of the reduced argument
// for |x| up to ~1e9.  Beyond that accuracy degrades gracefully (still deterministic).
AUVP_HD int AUVP_SC(auvp_rem_pio2)(double x, double* r, double* t) {
  const double pio2_lo = AUVP_K(AUVP_PIO2_LO);
  double n = auvp_rint(x * AUVP_K(AUVP_INV_PIO2));
  double hi = auvp_fma(-n, AUVP_K(AUVP_PIO2_HI), x);
  double p = n * pio2_lo;
  double pe = auvp_fma(n, pio2_lo, -p);  // exact error of p
  double rr = hi - p;
  double tt = (hi - rr) - p;                  // rounding error of the subtraction
  tt = tt - pe;
  tt = auvp_fma(-n, AUVP_K(AUVP_PIO2_LO2), tt);
  *r = rr;
  *t = tt;
  // n is integral: n mod 4 = the two low mantissa bits of n + 1.5 * 2^52 (exact for |n| < 2^51, i.e. |x| < 3.5e15; beyond
  // that still deterministic, the same bits on host and device)
  union { double d; unsigned long long u; } m;
  m.d = n + AUVP_K(0x1.8p52);
  return (int)(m.u & 3ull);
}

// sin(x + y) for |x| <= ~pi/4 with tail y
AUVP_HD double AUVP_SC(auvp_ksin)(double x, double y) {
#define AUVP_S1 -1.66666666666666324348e-01
#define AUVP_S2 8.33333333332248946124e-03
#define AUVP_S3 -1.98412698298579493134e-04
#define AUVP_S4 2.75573137070700676789e-06
#define AUVP_S5 -2.50507602534068634195e-08
#define AUVP_S6 1.58969099521155010221e-10
  double z = x * x;
  double v = z * x;
  double r = AUVP_FMA_K(z, AUVP_FMA_K(z, AUVP_FMA_K(z, AUVP_FMA_K(z, AUVP_K(AUVP_S6), AUVP_S5), AUVP_S4), AUVP_S3), AUVP_S2);
  return x - auvp_fma(-v, AUVP_K(AUVP_S1), auvp_fma(z, auvp_fma(-v, r, 0.5 * y), -y));
}

// cos(x + y) for |x| <= ~pi/4 with tail y
AUVP_HD double AUVP_SC(auvp_kcos)(double x, double y) {
#define AUVP_C1 4.16666666666666019037e-02
#define AUVP_C2 -1.38888888888741095749e-03
#define AUVP_C3 2.48015872894767294178e-05
#define AUVP_C4 -2.75573143513906633035e-07
#define AUVP_C5 2.08757232129817482790e-09
#define AUVP_C6 -1.13596475577881948265e-11
  double z = x * x;
  double r = z * AUVP_FMA_K(z, AUVP_FMA_K(z, AUVP_FMA_K(z, AUVP_FMA_K(z, AUVP_FMA_K(z, AUVP_K(AUVP_C6), AUVP_C5), AUVP_C4), AUVP_C3), AUVP_C2),
                            AUVP_C1);
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + auvp_fma(z, r, -(x * y)));
}

AUVP_HD void AUVP_SC(auvp_sincos)(double x, double* s, double* c) {
  double r, t;
  int q = AUVP_SC(auvp_rem_pio2)(x, &r, &t);
  double ks = AUVP_SC(auvp_ksin)(r, t);
  double kc = AUVP_SC(auvp_kcos)(r, t);
  double ss = (q & 1) ? kc : ks;
  double cc = (q & 1) ? ks : kc;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}

AUVP_HD double AUVP_SC(auvp_sin)(double x) {
  double s, c;
  AUVP_SC(auvp_sincos)(x, &s, &c);
  return s;
}

AUVP_HD double AUVP_SC(auvp_cos)(double x) {
  double s, c;
  AUVP_SC(auvp_sincos)(x, &s, &c);
  return c;
}

