"""The slice of rccl.h that csrc/gather_host.h declares LOCALLY (the library binds RCCL at run time and builds without its
headers) against the real header of this ROCm: argument order and convertibility of every entry point it calls, the two data-type
codes, the size of the unique id.  A compile-only check: collectives with more than one rank need more than one GPU, and the pool
has one per box -- this is what can be verified about ncclSend / ncclRecv / ncclBroadcast / ncclAllGather without them.  No GPU."""
import os
import subprocess

import pytest

RCCL_H = "/opt/rocm/include/rccl/rccl.h"

SRC = r'''
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <cstddef>
#include <type_traits>
// the local declarations of auv_sim_amd/csrc/gather_host.h (RcclApi), as function-pointer types
typedef struct ncclComm* auvp_ncclComm_t;
struct auvp_ncclUniqueId { char internal[128]; };
typedef int (*GetUniqueId_t)(auvp_ncclUniqueId*);
typedef int (*CommInitRank_t)(auvp_ncclComm_t*, int, auvp_ncclUniqueId, int);
typedef int (*CommDestroy_t)(auvp_ncclComm_t);
typedef int (*CommCount_t)(auvp_ncclComm_t, int*);
typedef int (*CommUserRank_t)(auvp_ncclComm_t, int*);
typedef int (*AllGather_t)(const void*, void*, size_t, int, auvp_ncclComm_t, hipStream_t);
typedef int (*Broadcast_t)(const void*, void*, size_t, int, int, auvp_ncclComm_t, hipStream_t);
typedef int (*Send_t)(const void*, size_t, int, int, auvp_ncclComm_t, hipStream_t);
typedef int (*Recv_t)(void*, size_t, int, int, auvp_ncclComm_t, hipStream_t);
typedef int (*Group_t)();
typedef const char* (*ErrStr_t)(int);

// same number of parameters, and every parameter of the real prototype has the size and kind (pointer / integer / by-value
// struct) of the local one: what makes calling through the local pointer type ABI-correct on x86-64
template <class A, class B> struct same_abi : std::integral_constant<bool,
    sizeof(A) == sizeof(B) && std::is_pointer<A>::value == std::is_pointer<B>::value &&
    (std::is_integral<A>::value || std::is_enum<A>::value) == (std::is_integral<B>::value || std::is_enum<B>::value) &&
    std::is_class<A>::value == std::is_class<B>::value> {};
template <class F, class G> struct sig_ok : std::false_type {};
template <class R1, class... A, class R2, class... B>
struct sig_ok<R1 (*)(A...), R2 (*)(B...)> {
  template <bool...> struct all;
  template <bool... bs> struct all_true : std::is_same<all<bs...>, all<(bs || true)...>> {};
  static constexpr bool value = sizeof...(A) == sizeof...(B) && same_abi<R1, R2>::value && all_true<same_abi<A, B>::value...>::value;
};
template <class R1, class R2> struct sig_ok<R1 (*)(), R2 (*)()> { static constexpr bool value = same_abi<R1, R2>::value; };

static_assert(sizeof(ncclUniqueId) == sizeof(auvp_ncclUniqueId) && NCCL_UNIQUE_ID_BYTES == 128, "unique id");
static_assert((int)ncclUint8 == 1 && (int)ncclInt64 == 4, "data type codes (AUVP_NCCL_UINT8 / AUVP_NCCL_INT64)");
static_assert((int)ncclSuccess == 0, "0 is success");
static_assert(sig_ok<decltype(&ncclGetUniqueId), GetUniqueId_t>::value, "ncclGetUniqueId");
static_assert(sig_ok<decltype(&ncclCommInitRank), CommInitRank_t>::value, "ncclCommInitRank");
static_assert(sig_ok<decltype(&ncclCommDestroy), CommDestroy_t>::value, "ncclCommDestroy");
static_assert(sig_ok<decltype(&ncclCommCount), CommCount_t>::value, "ncclCommCount");
static_assert(sig_ok<decltype(&ncclCommUserRank), CommUserRank_t>::value, "ncclCommUserRank");
static_assert(sig_ok<decltype(&ncclAllGather), AllGather_t>::value, "ncclAllGather");
static_assert(sig_ok<decltype(&ncclBroadcast), Broadcast_t>::value, "ncclBroadcast");
static_assert(sig_ok<decltype(&ncclSend), Send_t>::value, "ncclSend");
static_assert(sig_ok<decltype(&ncclRecv), Recv_t>::value, "ncclRecv");
static_assert(sig_ok<decltype(&ncclGroupStart), Group_t>::value && sig_ok<decltype(&ncclGroupEnd), Group_t>::value, "group calls");
static_assert(sig_ok<decltype(&ncclGetErrorString), ErrStr_t>::value, "ncclGetErrorString");
// argument ORDER (sizes alone cannot tell `int root` from `int datatype`): the calls as gather_host.h makes them must compile
// against the real prototypes with the real types in those positions
void order(const void* s, void* r, size_t n, ncclComm_t c, hipStream_t st) {
  (void)ncclAllGather(s, r, n, ncclUint8, c, st);
  (void)ncclBroadcast(s, r, n, ncclUint8, /*root*/ 0, c, st);
  (void)ncclSend(s, n, ncclUint8, /*peer*/ 0, c, st);
  (void)ncclRecv(r, n, ncclUint8, /*peer*/ 0, c, st);
}
int main() { return 0; }
'''


@pytest.mark.skipif(not os.path.exists(RCCL_H), reason="no rccl.h in this image")
def test_local_rccl_declarations_match_the_header(tmp_path):
    src = tmp_path / "rccl_abi.cpp"
    src.write_text(SRC)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(src)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_the_checked_declarations_are_the_ones_the_library_uses():
    """the typedefs above are copies: keep them in step with csrc/gather_host.h"""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = open(os.path.join(repo, "auv_sim_amd", "csrc", "gather_host.h")).read()
    for decl in ("int (*GetUniqueId)(auvp_ncclUniqueId*)", "int (*CommInitRank)(auvp_ncclComm_t*, int, auvp_ncclUniqueId, int)",
                 "int (*AllGather)(const void*, void*, size_t, int, auvp_ncclComm_t, hipStream_t)",
                 "int (*Broadcast)(const void*, void*, size_t, int, int, auvp_ncclComm_t, hipStream_t)",
                 "int (*Send)(const void*, size_t, int, int, auvp_ncclComm_t, hipStream_t)",
                 "int (*Recv)(void*, size_t, int, int, auvp_ncclComm_t, hipStream_t)",
                 "enum { AUVP_NCCL_UINT8 = 1, AUVP_NCCL_INT64 = 4 }"):
        assert decl in h, decl
