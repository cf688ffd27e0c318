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
#include "rrt_rows_stream_kernel.h"
#include "rrt_stream_kernel.h"

extern "C" __attribute__((visibility("hidden"))) hipError_t auvpi_rrt_rows_launch(const auvp::WorldDev* W, const auvp::RrtParamsDev* P,
                                                                                  const auvp::RrtBuffers* B, int n_episodes, int grid,
                                                                                  int block, int lds_max, int lds, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(auvp::rrt_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(auvp::rrt_rows_kernel, dim3(grid), dim3(block), lds, stream, *W, *P, *B, n_episodes);
  return hipGetLastError();
}

// round 6: the random() streams generated ahead (rrt_stream_kernel.h), and the expansion kernel that reads them
// (rrt_rows_stream_kernel.h: generated from rrt_rows_kernel.h by tools/gen_rows_stream_kernel.py)
extern "C" __attribute__((visibility("hidden"))) hipError_t auvpi_rrt_stream_launch(const auvp::RrtBuffers* B, int n_episodes, hipStream_t stream) {
  const int grid = (n_episodes + auvp::RSTREAM_WAVES - 1) / auvp::RSTREAM_WAVES;
  hipLaunchKernelGGL(auvp::rrt_stream_kernel, dim3(grid), dim3(auvp::RSTREAM_WAVES * 64), 0, stream, *B, n_episodes);
  return hipGetLastError();
}
extern "C" __attribute__((visibility("hidden"))) hipError_t auvpi_rrt_rows_stream_launch(const auvp::WorldDev* W, const auvp::RrtParamsDev* P,
                                                                                         const auvp::RrtBuffers* B, int n_episodes, int grid,
                                                                                         int block, int lds_max, int lds, hipStream_t stream) {
  // (the kernel is a template on the largest workgroup: the four-per-SIMD form -- sixteen wavefronts, 128 registers -- measured
  // slower and is not instantiated: profiles/r6_rows_stream.md)
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, *W, *P, *B, n_episodes);
    return hipGetLastError();
  };
  return go(auvp::rrt_rows_stream_kernel<auvp::RW_WAVES>);
}
