// auvp_seed.h -- CPython's random.seed(int) on the device: init_by_array over the 32-bit limbs of the seed (Modules/_randommodule.c,
// restated from the MT19937 reference algorithm; the host twin is seed_mt() in auvplan.hip).  Replaces the `random.seed(seed)`
// the reference's drivers call before a planning (path_planning/rrt_dubins.py: module-level `random`; gym_rrt/envs/rrt_env.py).
#ifndef AUVP_SEED_H
#define AUVP_SEED_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace auvp {

// init_by_array is a serial recurrence per generator (1 871 dependent steps), so the work is one thread per generator -- but
// with the state as a column of an LDS tile, m[word * 65] (a read-modify-write of global memory per step cost 4 x as much):
// conflict-free both for the per-thread recurrence (bank = word + thread) and for the coalesced write-out
constexpr int MT_SEED_LDS = 624 * 65 * 4;
// `count` consecutive steps of one of init_by_array's two mixing loops over words [i0, i0 + count) (no wrap inside), eight
// words read ahead of the dependent chain: a step needs the word's OLD value, which no earlier step of the run writes, so the
// LDS reads leave the chain (round 4: the reads in the loop were 2/3 of the kernel's time)
template <bool SECOND>
__device__ __forceinline__ uint32_t mt_seed_run(uint32_t* m, int i0, int count, uint32_t prev, const uint32_t (&key)[2], int klen, int& j) {
  int i = i0, left = count;
  for (; left >= 8; left -= 8, i += 8) {
    uint32_t w[8];
#pragma unroll
    for (int q = 0; q < 8; q++) w[q] = m[(i + q) * 65];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (!SECOND) { prev = (w[q] ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j; j++; if (j >= klen) j = 0; }
      else prev = (w[q] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)(i + q);
      m[(i + q) * 65] = prev;
    }
  }
  for (; left > 0; left--, i++) {
    if (!SECOND) { prev = (m[i * 65] ^ ((prev ^ (prev >> 30)) * 1664525u)) + key[j] + (uint32_t)j; j++; if (j >= klen) j = 0; }
    else prev = (m[i * 65] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
    m[i * 65] = prev;
  }
  return prev;
}

__device__ __forceinline__ void mt_seed_by_array_column(unsigned long long seed, uint32_t* m) {
  const uint32_t key[2] = {(uint32_t)(seed & 0xffffffffull), (uint32_t)(seed >> 32)};
  const int klen = key[1] ? 2 : 1;
  uint32_t prev = 19650218u;
  m[0] = prev;
  for (int i = 1; i < 624; i++) { prev = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i; m[i * 65] = prev; }
  // init_by_array: 624 steps over words 1 .. 623, 1 (the index wraps to 1 after 623; word 0 only ever receives copies and is
  // set last), then 623 steps over words 2 .. 623, 1
  int j = 0;
  prev = 19650218u;  // = m[0]
  prev = mt_seed_run<false>(m, 1, 623, prev, key, klen, j);
  prev = mt_seed_run<false>(m, 1, 1, prev, key, klen, j);
  prev = mt_seed_run<true>(m, 2, 622, prev, key, klen, j);
  prev = mt_seed_run<true>(m, 1, 1, prev, key, klen, j);
  m[0] = 0x80000000u;
}

// random.seed(seeds[e]) for every generator of a batch, 64 per workgroup (2 us per generator on a host core: 1.0 of the 1.4 ms
// a 512-episode Planner_RRT batch took to create, 25 ms of a 12 288-episode RRT.exploring batch).  rng_state (optional):
// the planner's {slot, available, drawn lo, drawn hi} words, zeroed -- a freshly seeded generator has generated nothing yet.
static __global__ __launch_bounds__(64) void mt_seed_kernel(const unsigned long long* __restrict__ seeds, uint32_t* __restrict__ mt,
                                                     int32_t* __restrict__ rng_state, int n) {
  extern __shared__ __align__(16) unsigned char seed_smem[];
  uint32_t* mtl = reinterpret_cast<uint32_t*>(seed_smem);
  const int t = (int)threadIdx.x;
  const int e0 = (int)blockIdx.x * 64;
  const int e = e0 + t;
  mt_seed_by_array_column(e < n ? seeds[e] : 0ull, mtl + t);
  if (e < n && rng_state) {
    int32_t* rs = rng_state + 4 * (size_t)e;
    rs[0] = 0; rs[1] = 0; rs[2] = 0; rs[3] = 0;
  }
  __syncthreads();
  const int n_here = (n - e0) < 64 ? (n - e0) : 64;
  for (int q = 0; q < n_here; q++) {
    uint32_t* dst = mt + (size_t)(e0 + q) * 624;
    for (int w = t; w < 624; w += 64) dst[w] = mtl[w * 65 + q];
  }
}

}  // namespace auvp
#endif
