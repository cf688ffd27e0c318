// valu_issue.hip -- issue cost (cycles per wave64 instruction, one wavefront per SIMD, independent instructions) of the vector
// instructions the fp64 kernels are made of.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define OPS8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define KERNEL(NAME, DECL, BODY8)                                                                          \
  __global__ __launch_bounds__(64) void NAME(long long* out, double seed, int iters) {                     \
    DECL;                                                                                                  \
    long long t0 = __builtin_amdgcn_s_memtime();                                                           \
    for (int i = 0; i < iters; i++) { BODY8 BODY8 BODY8 BODY8 }                                            \
    __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");                                                     \
    long long t1 = __builtin_amdgcn_s_memtime();                                                           \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                       \
    if (seed == 12345.678) out[0] = (long long)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);                    \
  }
#define DECL_D double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7; \
  unsigned u0 = (unsigned)seed, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7
#define A1(op) __asm__ volatile(op " %0, %0, %0\n\t" op " %1, %1, %1\n\t" op " %2, %2, %2\n\t" op " %3, %3, %3\n\t" op " %4, %4, %4\n\t" op " %5, %5, %5\n\t" op " %6, %6, %6\n\t" op " %7, %7, %7" \
  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
#define A3(op) __asm__ volatile(op " %0, %0, %0, %0\n\t" op " %1, %1, %1, %1\n\t" op " %2, %2, %2, %2\n\t" op " %3, %3, %3, %3\n\t" op " %4, %4, %4, %4\n\t" op " %5, %5, %5, %5\n\t" op " %6, %6, %6, %6\n\t" op " %7, %7, %7, %7" \
  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
#define AU(op) __asm__ volatile(op " %0, %0\n\t" op " %1, %1\n\t" op " %2, %2\n\t" op " %3, %3\n\t" op " %4, %4\n\t" op " %5, %5\n\t" op " %6, %6\n\t" op " %7, %7" \
  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
#define ACVT(op) __asm__ volatile(op " %0, %8\n\t" op " %1, %9\n\t" op " %2, %10\n\t" op " %3, %11\n\t" op " %4, %12\n\t" op " %5, %13\n\t" op " %6, %14\n\t" op " %7, %15" \
  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(u7));
#define AI2(op) __asm__ volatile(op " %0, %0, %0\n\t" op " %1, %1, %1\n\t" op " %2, %2, %2\n\t" op " %3, %3, %3\n\t" op " %4, %4, %4\n\t" op " %5, %5, %5\n\t" op " %6, %6, %6\n\t" op " %7, %7, %7" \
  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
#define ALDEXP __asm__ volatile("v_ldexp_f64 %0, %0, %8\n\tv_ldexp_f64 %1, %1, %8\n\tv_ldexp_f64 %2, %2, %8\n\tv_ldexp_f64 %3, %3, %8\n\tv_ldexp_f64 %4, %4, %8\n\tv_ldexp_f64 %5, %5, %8\n\tv_ldexp_f64 %6, %6, %8\n\tv_ldexp_f64 %7, %7, %8" \
  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(u0));
#define ACND __asm__ volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc\n\tv_cndmask_b32 %4, %4, %5, vcc\n\tv_cndmask_b32 %5, %5, %6, vcc\n\tv_cndmask_b32 %6, %6, %7, vcc\n\tv_cndmask_b32 %7, %7, %0, vcc" \
  : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : : "vcc");
#define AMOV64 __asm__ volatile("v_mov_b64 %0, %1\n\tv_mov_b64 %1, %2\n\tv_mov_b64 %2, %3\n\tv_mov_b64 %3, %4\n\tv_mov_b64 %4, %5\n\tv_mov_b64 %5, %6\n\tv_mov_b64 %6, %7\n\tv_mov_b64 %7, %0" \
  : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
#define ACMP __asm__ volatile("v_cmp_gt_f64 vcc, %0, %1\n\tv_cmp_gt_f64 vcc, %1, %2\n\tv_cmp_gt_f64 vcc, %2, %3\n\tv_cmp_gt_f64 vcc, %3, %4\n\tv_cmp_gt_f64 vcc, %4, %5\n\tv_cmp_gt_f64 vcc, %5, %6\n\tv_cmp_gt_f64 vcc, %6, %7\n\tv_cmp_gt_f64 vcc, %7, %0" \
  : : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7) : "vcc");
#define ASNOP __asm__ volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
#define ASMOV __asm__ volatile("s_mov_b32 s90, 1\n\ts_mov_b32 s91, 2\n\ts_mov_b32 s90, 3\n\ts_mov_b32 s91, 4\n\ts_mov_b32 s90, 5\n\ts_mov_b32 s91, 6\n\ts_mov_b32 s90, 7\n\ts_mov_b32 s91, 8" ::: "s90", "s91");

KERNEL(k_add_f64, DECL_D, A1("v_add_f64"))
KERNEL(k_mul_f64, DECL_D, A1("v_mul_f64"))
KERNEL(k_fma_f64, DECL_D, A3("v_fma_f64"))
KERNEL(k_rcp_f64, DECL_D, AU("v_rcp_f64"))
KERNEL(k_rsq_f64, DECL_D, AU("v_rsq_f64"))
KERNEL(k_rndne_f64, DECL_D, AU("v_rndne_f64"))
KERNEL(k_floor_f64, DECL_D, AU("v_floor_f64"))
KERNEL(k_cvt_f64_u32, DECL_D, ACVT("v_cvt_f64_u32"))
KERNEL(k_ldexp_f64, DECL_D, ALDEXP)
KERNEL(k_div_fixup, DECL_D, A3("v_div_fixup_f64"))
KERNEL(k_mov_b64, DECL_D, AMOV64)
KERNEL(k_cmp_f64, DECL_D, ACMP)
KERNEL(k_xor_b32, DECL_D, AI2("v_xor_b32"))
KERNEL(k_lshl_b32, DECL_D, AI2("v_lshlrev_b32"))
KERNEL(k_mul_lo_u32, DECL_D, AI2("v_mul_lo_u32"))
KERNEL(k_cndmask, DECL_D, ACND)
KERNEL(k_s_nop, DECL_D, ASNOP)
KERNEL(k_s_mov, DECL_D, ASMOV)

typedef void (*kern_t)(long long*, double, int);
int main() {
  long long* d;
  const int grid = 1024;  // 256 CUs x 4: one 64-thread workgroup per SIMD
  CHK(hipMalloc(&d, grid * sizeof(long long)));
  struct { const char* n; kern_t k; } ks[] = {
    {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64}, {"v_rcp_f64", k_rcp_f64}, {"v_rsq_f64", k_rsq_f64},
    {"v_rndne_f64", k_rndne_f64}, {"v_floor_f64", k_floor_f64}, {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_ldexp_f64", k_ldexp_f64},
    {"v_div_fixup_f64", k_div_fixup}, {"v_mov_b64", k_mov_b64}, {"v_cmp_gt_f64", k_cmp_f64}, {"v_xor_b32", k_xor_b32},
    {"v_lshlrev_b32", k_lshl_b32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_cndmask_b32", k_cndmask}, {"s_nop 0", k_s_nop}, {"s_mov_b32", k_s_mov}};
  const int iters = 2000;
  for (auto& e : ks) {
    for (int waves : {1, 3}) {  // wavefronts per SIMD issuing the same stream
      std::vector<long long> h(grid * waves);
      long long* dd;
      CHK(hipMalloc(&dd, h.size() * sizeof(long long)));
      hipLaunchKernelGGL(e.k, dim3(grid * waves), dim3(64), 0, 0, dd, 1.5, iters);
      hipLaunchKernelGGL(e.k, dim3(grid * waves), dim3(64), 0, 0, dd, 1.5, iters);
      CHK(hipDeviceSynchronize());
      CHK(hipMemcpy(h.data(), dd, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
      double s = 0;
      for (auto v : h) s += (double)v;
      // s_memtime counts at 100 MHz; shader clock ~2.4 GHz: report both
      const double ticks = s / h.size() / (iters * 32.0);
      printf("%-18s waves/SIMD %d: %.3f memtime ticks per instruction (x 24 = %.1f shader cycles at 2.4 GHz)\n", e.n, waves, ticks, ticks * 24.0);
      CHK(hipFree(dd));
    }
  }
  return 0;
}
