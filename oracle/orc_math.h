/* orc_math.h -- math back-end switch for the CPU checker.
 *
 * TEST INFRASTRUCTURE (oracle/).
 *
 * Two builds of the same sources:
 *   liboracle_libm.so      math = this machine's libm, called the way CPython calls it
 *                          (math.sin/cos/sqrt/atan2/hypot -> libm; `x ** 2` -> pow(x, 2.0),
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
#define ORC_HYPOT(x, y) hypot(x, y)
#define ORC_POW_E(z) pow(2.718281828459045, (z)) /* math.e ** z */
#define ORC_MATH_NAME "libm"
#endif

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
