// pf_kernels.hip -- the particle-filter kernels (pf_kernel.h) as a translation unit of their own, so that they can be
// compiled with -mllvm -disable-machine-licm (see __graft_entry__.py): LLVM's machine LICM hoists the fp64 literals of
// atan2 / exp / sincos and the per-thread LDS addresses out of the step loop, ~120 loop-invariant scalar registers that then
// spill (v_writelane / scratch) -- with the pass off pf_step_kernel<512, 2> needs 103 VGPRs and no scratch (122 + a 68-byte
// private segment with it; 128 + 132 bytes of real scratch traffic per lane in round 4).  The rest of the library is
// compiled WITH the pass (measured both ways, profiles/r5_machine_licm.md).
//
// Entry points for the host side (pf_host.h, in auvplan.hip): internal, hidden visibility, not part of the C-ABI.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "auvp_types.h"
#include "auvp_wave.h"
#include "pf_kernel.h"

#ifndef AUVP_PF_THREADS
#define AUVP_PF_THREADS 512
#endif

extern "C" {

__attribute__((visibility("hidden"))) int auvpi_pf_threads(void) { return AUVP_PF_THREADS; }

__attribute__((visibility("hidden"))) size_t auvpi_pf_lds_bytes(int N) { return auvp::pf_lds_bytes(N); }

__attribute__((visibility("hidden"))) hipError_t auvpi_pf_create_launch(const auvp::PfDev* D, hipStream_t stream) {
  const size_t lds = auvp::pf_lds_bytes(D->N);
  hipError_t e = hipFuncSetAttribute((const void*)auvp::pf_create_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(auvp::pf_create_kernel, dim3(D->F), dim3(PF_T), lds, stream, *D);
  return hipGetLastError();
}

// threads per filter, measured on MI355X at N = 1000 (4096 filters x 20 steps).  Round 1: 256 -> 11.6 ms, 1024 -> 14.4 ms
// (six barriers per MT19937 regeneration).  Round 4, regeneration without inner barriers: 256 threads (237 registers, one
// wavefront per SIMD and workgroup, two workgroups per CU by LDS) 9.2 ms -- one workgroup alone on a CU takes 0.94 of
// that: the step is a chain of dependent fp64 latencies, not issue bound (0.40 of the VALU issue slots); 512 threads held to
// 128 registers (four wavefronts per SIMD) 7.7 ms; 1024 (one workgroup per CU) 11.5 ms; 384 threads x 3 particles at 168
// registers: 14.1 ms against 7.1 ms -- six wavefronts do not spread evenly over the four SIMDs.  Round 5: no spills, 6.7 ms.
__attribute__((visibility("hidden"))) hipError_t auvpi_pf_step_launch(const auvp::PfDev* D, hipStream_t stream) {
  constexpr int T = AUVP_PF_THREADS, P1 = (1024 + T - 1) / T, P2 = (2048 + T - 1) / T;
  const size_t lds = auvp::pf_lds_bytes(D->N);
  hipError_t e;
  if (D->N <= 1024) {
    e = hipFuncSetAttribute((const void*)auvp::pf_step_kernel<T, P1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((auvp::pf_step_kernel<T, P1>), dim3(D->F), dim3(T), lds, stream, *D);
  } else {
    e = hipFuncSetAttribute((const void*)auvp::pf_step_kernel<T, P2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((auvp::pf_step_kernel<T, P2>), dim3(D->F), dim3(T), lds, stream, *D);
  }
  return hipGetLastError();
}

}  // extern "C"
