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
// The stream's length is a bound (RRT.exploring draws ~44.8 values per iteration on the bench world, twice that with other
// parameters: the host sets it from what earlier batches with the same parameters drew -- auvplan.hip); an
// episode that runs past it ends with AUVP_ST_STREAM and the batch is redone by rrt_rows_kernel (auvplan.hip: stream
// fallback), like an episode a speculative pipeline gave up on.
//
// The values are CPython's: word q of the stream = output q of MT19937 from the episode's state (B.mt, position B.mt_index:
// 624 = a fresh seed), random() j = (temper(word 2j) >> 5) * 2^26 + (temper(word 2j + 1) >> 6)) / 2^53 (auvp_wave.h).
#ifndef AUVP_RRT_STREAM_KERNEL_H
#define AUVP_RRT_STREAM_KERNEL_H
#include "auvp_types.h"
#include "auvp_wave.h"

namespace auvp {

// episodes per workgroup (one wavefront each; 2.5 KB of LDS each).  Measured on the headline batch (46 GB of numbers): 4 per
// workgroup 10.5 ms, 8: 9.8, 16: 9.9; with plain instead of non-temporal stores in the whole-cycle path 9.3 (4.9 TB/s written)
constexpr int RSTREAM_WAVES = 8;

// One block of a whole cycle at a fixed position G of the state (0, 128, 256, 384, 512): 128 words regenerated in place (112 at
// 512), 64 (56) random() values stored at out[lane].  G is a compile-time constant, so every LDS address is the lane's byte
// offset plus an immediate -- the two blocks with a lane-dependent wrap (128: x[k + 397] crosses the end at lane 50; 512: lane
// 55's x[k + 2] is x[0]) take theirs from addresses the compiler hoists out of the cycle loop.
template <int G>
__device__ __forceinline__ void rstream_block(uint32_t* s, int lane, double* out) {
  const int k = G + 2 * lane;
  const bool mine = G < 512 || lane < 56;
  const int k2 = (G == 512 && lane == 55) ? 0 : k + 2;         // (lanes past a short block read inside the padding: unused)
  const int km = k + 397 >= 624 ? k - 227 : k + 397;           // (k even: km is odd or even alike, never 624)
  const int km1 = km + 1 == 624 ? 0 : km + 1;
  const uint32_t a0 = s[k], a1 = s[k + 1], a2 = s[k2], c0 = s[km], c1 = s[km1];
  uint32_t y0, y1;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(y0) : "v"(0x7fffffffu), "v"(a1), "v"(a0));   // (a0 & 0x80000000) | (a1 & 0x7fffffff)
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(y1) : "v"(0x7fffffffu), "v"(a2), "v"(a1));
  uint32_t m0, m1;                                                                    // bit 0 spread over the word: 0 or ~0
  asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(m0) : "v"(a1));
  asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(m1) : "v"(a2));
  const uint32_t v0 = c0 ^ (y0 >> 1) ^ (m0 & 0x9908b0dfu), v1 = c1 ^ (y1 >> 1) ^ (m1 & 0x9908b0dfu);
  wave_sync();
  if (mine) {
    s[k] = v0;
    s[k + 1] = v1;
  }
  wave_sync();
  if (mine) out[lane] = py_random_from(mt_temper(v0) >> 5, mt_temper(v1) >> 6);
}

static __global__ __launch_bounds__(RSTREAM_WAVES * 64) void rrt_stream_kernel(RrtBuffers B, int n_episodes) {
  __shared__ uint32_t st[RSTREAM_WAVES][640];   // (624 words of state; 16 of padding for the reads of the idle lanes of a short block)
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
    // g is 0 here (the words of the old cycle are out): whole cycles -- 624 words, 312 numbers -- run unrolled with constant
    // addresses (rstream_block: 36 vector instructions per 64 numbers instead of 70) while the stream has room for one ...
    while (emitted + 312 <= cap) {
      double* o = out + emitted;
      rstream_block<0>(s, lane, o);
      rstream_block<128>(s, lane, o + 64);
      rstream_block<256>(s, lane, o + 128);
      rstream_block<384>(s, lane, o + 192);
      rstream_block<512>(s, lane, o + 256);
      emitted += 312;
    }
    // ... and the last, cut one block by block
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
