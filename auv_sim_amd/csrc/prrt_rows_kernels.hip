// prrt_rows_kernels.hip -- prrt_rows_kernel (planner_rows_kernel.h: Planner_RRT, four episodes per wavefront: config 5 and
// the batched environment) as a translation unit of its own, compiled with -mllvm -disable-machine-licm (__graft_entry__.py):
// without the machine-LICM pass the kernel needs 152 instead of 160-162 VGPRs (the hoisted loop invariants are re-formed inside
// the step) and config 5's plan launch is 3 % faster (profiles/r5_machine_licm.md); the rest of the library keeps the pass.
//
// Entry point for the host side (planner_rrt_host.h in auvplan.hip): internal, hidden visibility, not part of the C-ABI.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rrt_explore_kernel.h"
#include "planner_rows_kernel.h"

extern "C" __attribute__((visibility("hidden"))) hipError_t auvpi_prrt_rows_launch(int obst_lds, const auvp::WorldDev* W,
                                                                                   const auvp::PrrtParamsDev* P, const auvp::PrrtBuffers* B,
                                                                                   int n_episodes, int* work_counter, int work_base,
                                                                                   int occ_bytes, int grid, int block, int lds,
                                                                                   hipStream_t stream) {
  auto launch = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, *W, *P, *B, n_episodes, work_counter, work_base, occ_bytes);
    return hipGetLastError();
  };
  return obst_lds ? launch(auvp::prrt_rows_kernel<true>) : launch(auvp::prrt_rows_kernel<false>);
}
