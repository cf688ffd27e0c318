// rrt_stream_kernel.h -- the random() stream of every episode, generated AHEAD of the expansion kernel (round 6).
//
// rrt_rows_kernel spends ~300 of the 1 284 vector instructions of a trip on CPython's generator: MT19937 refills and -- the
// larger part -- tempering the words and forming the 53-bit random() values, at ~45 % lane utilisation (the four rows of a
// wavefront need their windows at different times: ~5.2 tempering rounds per trip where 2.7 full ones would do), and its 2 496 B
// of generator state per episode are what fills the CU's LDS (48 episodes).  Measured with a stand-in generator
// (profiles/r6_rows_stream.md): without that work the kernel takes 76.7 ms instead of 93.8.
//
// So the stream is produced where it is cheap: ONE WAVEFRONT PER EPISODE, 128 words regenerated per round (two per lane: words
// 227 apart are independent, so any 128 consecutive are), 64 random() values tempered and stored per round -- every lane busy,
// no cross-lane traffic, one coalesced 512-byte store -- into `B.stream[episode][0 .. stream_cap)` in HBM.  The expansion
// kernel (rrt_rows_stream_kernel.h) then reads random() number j of its episode at stream[j]: no generator, no tempering.
// The stream's length is a bound (RRT.exploring draws ~44.8 values per iteration on the bench world, 48.8 at most over short
// runs: the host allocates 46.5 per iteration + 4 096); an episode that runs past it ends with AUVP_ST_STREAM and the batch
// is redone by rrt_rows_kernel (auvplan.hip: stream fallback), like an episode a speculative pipeline gave up on.
//
// The values are CPython's: word q of the stream = output q of MT19937 from the episode's state (B.mt, position B.mt_index:
// 624 = a fresh seed), random() j = (temper(word 2j) >> 5) * 2^26 + (temper(word 2j + 1) >> 6)) / 2^53 (auvp_wave.h).
#ifndef AUVP_RRT_STREAM_KERNEL_H
#define AUVP_RRT_STREAM_KERNEL_H
#include "auvp_types.h"
#include "auvp_wave.h"

namespace auvp {

constexpr int RSTREAM_WAVES = 4;  // episodes per workgroup (one wavefront each; 2 496 B of LDS each)

static __global__ __launch_bounds__(RSTREAM_WAVES * 64) void rrt_stream_kernel(RrtBuffers B, int n_episodes) {
  __shared__ uint32_t st[RSTREAM_WAVES][624];
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const int ep = (int)blockIdx.x * RSTREAM_WAVES + wave;
  if (ep >= n_episodes) return;  // (whole wavefronts: no workgroup barrier below)
  uint32_t* s = st[wave];
  for (int i = lane; i < 624; i += 64) s[i] = B.mt[(size_t)ep * 624 + i];
  int idx = B.mt_index ? B.mt_index[ep] : 624;
  idx = uni(idx < 0 ? 0 : (idx > 624 ? 624 : idx));
  wave_sync();
  double* out = B.stream + (size_t)ep * (size_t)B.stream_cap;
  const long long cap = B.stream_cap;
  // stream word q is state slot (idx + q) mod 624 once it exists; the slots idx .. 623 hold words 0 .. 623 - idx already
  long long have = 624 - idx;  // stream words that exist
  long long emitted = 0;       // random() values stored
  int e0 = idx == 624 ? 0 : idx;  // state slot of stream word 2 * emitted
  int g = 0;                   // state slot the next regenerated word goes to = (idx + have) mod 624 (0 at the start, whatever idx)
  // ---- (i) what the state still holds of the current cycle (a continued stream: idx < 624) goes out first: pairs from LDS
  {
    long long pairs = have >> 1;
    if (pairs > cap) pairs = cap;
    while (pairs > 0) {
      const int n = pairs < 64 ? (int)pairs : 64;
      if (lane < n) {
        int i0 = e0 + 2 * lane;
        i0 = i0 >= 624 ? i0 - 624 : i0;
        const int i1 = i0 + 1 == 624 ? 0 : i0 + 1;
        __builtin_nontemporal_store(py_random_from(mt_temper(s[i0]) >> 5, mt_temper(s[i1]) >> 6), out + emitted + lane);
      }
      emitted += n;
      pairs -= n;
      e0 += 2 * n;
      e0 = e0 >= 624 ? e0 - 624 : e0;
    }
  }
  if ((have & 1) == 0) {
    // ---- (ii) the stream's pairs are the blocks' pairs (always so for a fresh seed): a lane regenerates its two words, tempers
    // them in registers and stores the number -- no second pass over LDS.  x[k] = x[k + 397] ^ twist(x[k], x[k + 1]) in place;
    // blocks of 128 slots that never wrap (the last one of a cycle has 112); every read of a round precedes every write.
    while (emitted < cap) {
      const int len = 624 - g < 128 ? 624 - g : 128;
      const int k = g + 2 * lane;
      const bool mine = k < g + len;           // (len is even: both words or neither)
      const int kc = mine ? k : 0;
      const int k2 = kc + 2 == 624 ? 0 : kc + 2;
      int km = kc + 397;
      km = km >= 624 ? km - 624 : km;           // (k is even, 397 odd: km <= 623 after the wrap, km + 1 may be 624)
      const int km1 = km + 1 == 624 ? 0 : km + 1;
      const uint32_t a0 = s[kc], a1 = s[kc + 1], a2 = s[k2], c0 = s[km], c1 = s[km1];
      const uint32_t y0 = (a0 & 0x80000000u) | (a1 & 0x7fffffffu), y1 = (a1 & 0x80000000u) | (a2 & 0x7fffffffu);
      const uint32_t v0 = c0 ^ (y0 >> 1) ^ ((y0 & 1u) ? 0x9908b0dfu : 0u), v1 = c1 ^ (y1 >> 1) ^ ((y1 & 1u) ? 0x9908b0dfu : 0u);
      wave_sync();
      if (mine) { s[kc] = v0; s[kc + 1] = v1; }
      wave_sync();
      const long long j = emitted + lane;
      if (mine && j < cap) __builtin_nontemporal_store(py_random_from(mt_temper(v0) >> 5, mt_temper(v1) >> 6), out + j);
      emitted += len >> 1;
      g += len;
      g = g >= 624 ? 0 : g;
    }
    return;
  }
  // ---- (iii) a continued stream with an odd number of words left in its cycle: the pairs straddle the blocks' lanes; regenerate
  // a block, then emit every complete pair from LDS (at most one word is left over)
  while (emitted < cap) {
    const int len = 624 - g < 128 ? 624 - g : 128;
    uint32_t v[2];
    int kk[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int k = g + 2 * lane + h;
      kk[h] = k;
      const int kc = k < 624 ? k : 623;  // (lanes past a short block read something harmless)
      const int k1 = kc == 623 ? 0 : kc + 1;
      int km = kc + 397;
      km = km >= 624 ? km - 624 : km;
      const uint32_t a = s[kc], b = s[k1], c = s[km];
      const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
      v[h] = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    wave_sync();
    if (kk[0] < g + len) s[kk[0]] = v[0];
    if (kk[1] < g + len) s[kk[1]] = v[1];
    wave_sync();
    have += len;
    g += len;
    g = g >= 624 ? 0 : g;
    long long pairs = (have >> 1) - emitted;
    if (pairs > cap - emitted) pairs = cap - emitted;
    while (pairs > 0) {
      const int n = pairs < 64 ? (int)pairs : 64;
      if (lane < n) {
        int i0 = e0 + 2 * lane;
        i0 = i0 >= 624 ? i0 - 624 : i0;
        const int i1 = i0 + 1 == 624 ? 0 : i0 + 1;
        __builtin_nontemporal_store(py_random_from(mt_temper(s[i0]) >> 5, mt_temper(s[i1]) >> 6), out + emitted + lane);
      }
      emitted += n;
      pairs -= n;
      e0 += 2 * n;
      e0 = e0 >= 624 ? e0 - 624 : e0;
    }
  }
}

}  // namespace auvp
#endif
