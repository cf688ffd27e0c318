// auvp_math_late.h -- device-only variants of auvp_math.h functions whose constants are materialised where they are used.
// Same operations in the same order as the plain functions: the same doubles (tests compare the kernels that use them with
// the portable checker build bit for bit).
#ifndef AUVP_MATH_LATE_H
#define AUVP_MATH_LATE_H
#include "auvp_math.h"

// atan / atan2 with the argument-reduction constants materialised where they are used (auvp_atan_body.h): hoisted out of the
// step loop they were this kernel's only spill (three 8-byte stores per wave at entry, four reloads per trip)
template <unsigned LO, unsigned HI>
__device__ __forceinline__ double auvp_late_f64() {
  // the two halves as instruction literals inside a volatile asm: nothing loop-invariant is left outside to hoist and spill
  unsigned a, b;
  __asm__ volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(a), "=v"(b) : "i"(LO), "i"(HI));
  return __longlong_as_double(((long long)b << 32) | (long long)a);
}
#define AUVP_LATE_BITS(c) __builtin_bit_cast(unsigned long long, (double)(c))
#define AUVP_ATAN_FN auvp_atan_late
#define AUVP_ATAN2_FN auvp_atan2_late
#define AUVP_ATAN_K(c) auvp_late_f64<(unsigned)(AUVP_LATE_BITS(c) & 0xffffffffull), (unsigned)(AUVP_LATE_BITS(c) >> 32)>()
#include "auvp_atan_body.h"
#undef AUVP_ATAN_FN
#undef AUVP_ATAN2_FN
#undef AUVP_ATAN_K

#endif  // AUVP_MATH_LATE_H
