/* orc_math.h -- math back-end switch for the CPU checker.
 *
 * TEST INFRASTRUCTURE (oracle/).
 *
 * Two builds of the same sources:
 *   liboracle_libm.so      math = this machine's libm, called the way CPython calls it
 *                          (math.sin/cos/sqrt/atan2 -> libm; math.hypot -> CPython's own vector_norm, below; `x ** 2` -> pow(x, 2.0),
 *                          Objects/floatobject.c float_pow).  Bit-identical to the reference when
 *                          run on the glibc the goldens were captured with; this is the build that
 *                          is pinned against tests/golden/.
 *   liboracle_portable.so  math = auv_sim_amd/csrc/auvp_math.h (explicit fp64 ops; the same header
 *                          the HIP kernels compile).  Bit-identical to the GPU; compared with the
 *                          libm build / the goldens exactly on decisions and to 1e-9 on floats.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H
#include <math.h>

#ifdef ORC_PORTABLE_MATH
#include "../auv_sim_amd/csrc/auvp_math.h"
#include "../auv_sim_amd/csrc/auvp_exp.h"
#define ORC_SIN(x) auvp_sin(x)
#define ORC_COS(x) auvp_cos(x)
#define ORC_POW2(x) ((x) * (x))
#define ORC_SQRT(x) auvp_sqrt(x)
#define ORC_ATAN2(y, x) auvp_atan2(y, x)
#define ORC_HYPOT(x, y) auvp_hypot(x, y)
#define ORC_POW_E(z) auvp_pow_e(z)
#define ORC_MATH_NAME "portable"
#else
#define ORC_SIN(x) sin(x)
#define ORC_COS(x) cos(x)
#define ORC_POW2(x) pow((x), 2.0)
#define ORC_SQRT(x) sqrt(x)
#define ORC_ATAN2(y, x) atan2(y, x)
#define ORC_HYPOT(x, y) orc_cpython_hypot(x, y)
#define ORC_POW_E(z) pow(2.718281828459045, (z)) /* math.e ** z */
#define ORC_MATH_NAME "libm"
#endif

/* math.hypot(x, y) as CPython >= 3.10 computes it (Modules/mathmodule.c vector_norm: scaled squares split with Veltkamp's
 * constant, compensated sums, sqrt, one differential correction) -- NOT the C library's hypot(): glibc's differs from it in the
 * last bit on ~0.4 % of arguments (measured in this container: 8 494 of 2 000 000, this restatement 0 of 2 000 000; round 6:
 * tests/golden g2_org_m50_m30 was the first fixture on which the difference reached a stored float).  The "libm" build of the
 * checker stands for "what the reference's interpreter computes", so it takes this one.  Finite, non-subnormal-max arguments
 * (the planners' ranges); inf / NaN as CPython. */
static inline double orc_cpython_hypot(double a, double b) {
  const double T27 = 134217729.0; /* ldexp(1.0, 27) + 1.0 */
  double vec[2], max, x, scale, oldcsum, csum = 1.0, frac1 = 0.0, frac2 = 0.0, frac3 = 0.0, t, hi, lo, h;
  int max_e, i;
  vec[0] = fabs(a); vec[1] = fabs(b);
  max = vec[0] > vec[1] ? vec[0] : vec[1];
  if (isinf(vec[0]) || isinf(vec[1])) return INFINITY;
  if (isnan(a) || isnan(b)) return NAN;
  if (max == 0.0) return max;
  frexp(max, &max_e);
  if (max_e < -1023) return hypot(a, b); /* (CPython divides by the subnormal max here; never reached by the planners) */
  scale = ldexp(1.0, -max_e);
  for (i = 0; i < 2; i++) {
    x = vec[i] * scale;
    t = x * T27; hi = t - (t - x); lo = x - hi;
    x = hi * hi; oldcsum = csum; csum += x; frac1 += (oldcsum - csum) + x;
    x = 2.0 * hi * lo; oldcsum = csum; csum += x; frac2 += (oldcsum - csum) + x;
    frac3 += lo * lo;
  }
  h = sqrt(csum - 1.0 + (frac1 + frac2 + frac3));
  x = h; t = x * T27; hi = t - (t - x); lo = x - hi;
  x = -hi * hi; oldcsum = csum; csum += x; frac1 += (oldcsum - csum) + x;
  x = -2.0 * hi * lo; oldcsum = csum; csum += x; frac2 += (oldcsum - csum) + x;
  x = -lo * lo; oldcsum = csum; csum += x; frac3 += (oldcsum - csum) + x;
  x = csum - 1.0 + (frac1 + frac2 + frac3);
  return (h + x / (2.0 * h)) / scale;
}

/* CPython float floor division a // b for b > 0 (Objects/floatobject.c float_floor_div) */
static inline double orc_floordiv(double vx, double wx) {
  double mod = fmod(vx, wx);
  double div = (vx - mod) / wx;
  if (mod != 0.0) {
    if ((wx < 0) != (mod < 0)) { div -= 1.0; }
  }
  double floordiv;
  if (div != 0.0) {
    floordiv = floor(div);
    if (div - floordiv > 0.5) floordiv += 1.0;
  } else {
    floordiv = copysign(0.0, vx / wx);
  }
  return floordiv;
}

#endif
