// rows_kernels.hip -- rrt_rows_kernel (rrt_rows_kernel.h: the headline's expansion kernel, four episodes per wavefront) as a
// translation unit of its own, compiled with -mllvm -amdgpu-sched-strategy=max-ilp (__graft_entry__.py).  The kernel is bound
// by fp64 vector issue at three wavefronts per SIMD; the max-ILP scheduling strategy interleaves its independent chains more
// aggressively than the default (occupancy-first) one: 101.3 -> 99.7 ms on the headline batch, bit-identical.  The same flag on
// the rest of the library costs rrt_leaf_kernel 11 % (6.49 -> 7.17 ms), hence the separate unit (profiles/r5_machine_licm.md).
//
// The kernel calls auvp_sincos_sk (auvp_math.h): the Horner steps of sin / cos as one v_fma_f64 each with the constant in a fixed
// scalar pair -- the compiler's v_fmac_f64 form copies the constant into the destination first (18 v_mov per sin / cos):
// 97.1 -> 95.9 ms; the planner and particle-filter kernels measured neutral to -5 % with it and call the plain auvp_sincos.
//
// Entry point for the host side (auvplan.hip): internal, hidden visibility, not part of the C-ABI.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rrt_rows_kernel.h"

extern "C" __attribute__((visibility("hidden"))) hipError_t auvpi_rrt_rows_launch(const auvp::WorldDev* W, const auvp::RrtParamsDev* P,
                                                                                  const auvp::RrtBuffers* B, int n_episodes, int grid,
                                                                                  int block, int lds_max, int lds, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(auvp::rrt_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(auvp::rrt_rows_kernel, dim3(grid), dim3(block), lds, stream, *W, *P, *B, n_episodes);
  return hipGetLastError();
}
