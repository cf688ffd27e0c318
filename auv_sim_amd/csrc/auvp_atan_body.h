// auvp_atan_body.h -- the bodies of atan / atan2 of auvp_math.h, included once per instantiation:
//   AUVP_ATAN_FN, AUVP_ATAN2_FN   the function names
//   AUVP_ATAN_K(c)                how a reduction constant reaches its use: (c) as is, or through an opaque move so that the
//                                 compiler materialises it there instead of keeping it in registers across a caller's loop
//                                 (planner_rows_kernel.h: the hoisted constants were the kernel's only spill)
// Same operations in the same order either way: the same doubles.
AUVP_HD double AUVP_ATAN_FN(double x) {
  // (the reduction constants are spelled as literals at their use -- AUVP_ATAN_K may need them as compile-time constants)
#define AUVP_ATAN_HI0 4.63647609000806093515e-01
#define AUVP_ATAN_HI1 7.85398163397448278999e-01
#define AUVP_ATAN_HI2 9.82793723247329054082e-01
#define AUVP_ATAN_HI3 1.57079632679489655800e+00
#define AUVP_ATAN_LO0 2.26987774529616870924e-17
#define AUVP_ATAN_LO1 3.06161699786838301793e-17
#define AUVP_ATAN_LO2 1.39033110312309984516e-17
#define AUVP_ATAN_LO3 6.12323399573676603587e-17
  const double hi3 = AUVP_ATAN_HI3, lo3 = AUVP_ATAN_LO3;
#define AUVP_AT0 3.33333333333329318027e-01
#define AUVP_AT1 -1.99999999998764832476e-01
#define AUVP_AT2 1.42857142725034663711e-01
#define AUVP_AT3 -1.11111104054623557880e-01
#define AUVP_AT4 9.09088713343650656196e-02
#define AUVP_AT5 -7.69187620504482999495e-02
#define AUVP_AT6 6.66107313738753120669e-02
#define AUVP_AT7 -5.83357013379057348645e-02
#define AUVP_AT8 4.97687799461593236017e-02
#define AUVP_AT9 -3.65315727442169155270e-02
#define AUVP_AT10 1.62858201153657823623e-02
  const int neg = x < 0.0;
  double ax = auvp_fabs(x);
  if (!(ax < 0x1p66)) {  // one test for the two rare cases: nan, |x| >= 2^66 (pi/2)
    if (x != x) return x;
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
  } else if (ax < 0.6875) { id = 0; num = 2.0 * ax - 1.0; den = 2.0 + ax; hi = AUVP_ATAN_K(AUVP_ATAN_HI0); lo = AUVP_ATAN_K(AUVP_ATAN_LO0); }
  else if (ax < 1.1875) { id = 1; num = ax - 1.0; den = ax + 1.0; hi = AUVP_ATAN_K(AUVP_ATAN_HI1); lo = AUVP_ATAN_K(AUVP_ATAN_LO1); }
  else if (ax < 2.4375) { id = 2; num = ax - 1.5; den = 1.0 + 1.5 * ax; hi = AUVP_ATAN_K(AUVP_ATAN_HI2); lo = AUVP_ATAN_K(AUVP_ATAN_LO2); }
  else { id = 3; num = -1.0; den = ax; hi = AUVP_ATAN_K(AUVP_ATAN_HI3); lo = AUVP_ATAN_K(AUVP_ATAN_LO3); }
  if (id >= 0) ax = auvp_div_plain(num, den);  // den in [1, 2^66), |num| <= den: never near the ends of the exponent range
  double z = ax * ax;
  double w = z * z;
  double s1 = z * AUVP_FMA_K(w, AUVP_FMA_K(w, AUVP_FMA_K(w, AUVP_FMA_K(w, AUVP_FMA_K(w, AUVP_K(AUVP_AT10), AUVP_AT8), AUVP_AT6), AUVP_AT4), AUVP_AT2), AUVP_AT0);
  double s2 = w * AUVP_FMA_K(w, AUVP_FMA_K(w, AUVP_FMA_K(w, AUVP_FMA_K(w, AUVP_K(AUVP_AT9), AUVP_AT7), AUVP_AT5), AUVP_AT3), AUVP_AT1);
  if (id < 0) {
    double r = auvp_fma(-ax, s1 + s2, ax);
    return neg ? -r : r;
  }
  double r = hi - (auvp_fma(ax, s1 + s2, -lo) - ax);
  return neg ? -r : r;
}

#undef AUVP_ATAN_HI0
#undef AUVP_ATAN_HI1
#undef AUVP_ATAN_HI2
#undef AUVP_ATAN_HI3
#undef AUVP_ATAN_LO0
#undef AUVP_ATAN_LO1
#undef AUVP_ATAN_LO2
#undef AUVP_ATAN_LO3

// atan2(y, x) with the usual quadrant logic (finite inputs; infinities map through atan's limits).
// Measured <= 1.2 ulp vs mpmath (the y/x rounding adds to atan's own error); glibc's atan2 is
// nearly correctly rounded, so the two agree bit-for-bit on ~80 % of inputs and differ by one ulp
// otherwise -- used only by Planner_RRT's goal connection, where it feeds floats, one
// `abs(diff) > pi/2` test and one floor(length/exp_rate).
AUVP_HD double AUVP_ATAN2_FN(double y, double x) {
  const double pi = AUVP_PI, pi_lo = 1.2246467991473531772E-16, pio2 = AUVP_PIO2_HI;
  const double ax = auvp_fabs(x), ay = auvp_fabs(y);
  double z;
  if (ax > 0.0 && ay > 0.0 && ax < __builtin_inf() && ay < __builtin_inf()) {
    // the regular case first (finite, non-zero operands: one test instead of a chain of five)
    double q = ay / ax;
    if (q >= 0x1p64) z = pio2 + 0.5 * pi_lo;
    else if (x < 0.0 && q < 0x1p-64) z = 0.0;
    else z = AUVP_ATAN_FN(q);
  } else {
    if (x != x || y != y) return x + y;
    if (y == 0.0) {
      // atan2(+-0, +x) = +-0 ; atan2(+-0, -x) = +-pi
      int xneg = (x < 0.0) || (x == 0.0 && __builtin_signbit(x));
      if (!xneg) return y;
      return __builtin_signbit(y) ? -pi : pi;
    }
    if (x == 0.0) return y < 0.0 ? -pio2 : pio2;
    // an infinite operand
    if (ax == ay) z = 0.5 * pio2;            // pi/4
    else if (ay == __builtin_inf()) z = pio2;
    else z = 0.0;
  }
  if (x > 0.0) return y < 0.0 ? -z : z;
  double r = pi - (z - pi_lo);
  return y < 0.0 ? -r : r;
}
