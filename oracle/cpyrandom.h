/* cpyrandom.h -- CPython 3.10 `random` module stream, restated in C.
 *
 * TEST INFRASTRUCTURE (oracle/): the CPU checker the HIP path is compared against.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * The reference planners draw from the global `random` module (path_planning/rrt_dubins.py:123-129,
 * 259-279,336-339; gym_rrt/envs/rrt_dubins.py:186,223,262-267).  Parity on seeded inputs needs the
 * same stream: MT19937 with init_by_array seeding over the 32-bit limbs of abs(seed), 53-bit
 * random() = (a>>5, b>>6), uniform(a,b) = a + (b-a)*random(), choice() -> _randbelow() by
 * getrandbits(bit_length) rejection.  Pinned by tests/golden/g7_random_kat.json (stdlib outputs).
 */
#ifndef ORC_CPYRANDOM_H
#define ORC_CPYRANDOM_H
#include <stdint.h>

typedef struct {
  uint32_t mt[624];
  int idx;
  uint64_t n_draw32; /* number of 32-bit outputs consumed (diagnostic) */
} cpy_rng;

static void cpy_init_genrand(cpy_rng* r, uint32_t s) {
  r->mt[0] = s;
  for (int i = 1; i < 624; i++)
    r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
  r->idx = 624;
}

static void cpy_init_by_array(cpy_rng* r, const uint32_t* key, int klen) {
  cpy_init_genrand(r, 19650218u);
  int i = 1, j = 0;
  int k = (624 > klen) ? 624 : klen;
  for (; k; k--) {
    r->mt[i] = (r->mt[i] ^ ((r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    i++; j++;
    if (i >= 624) { r->mt[0] = r->mt[623]; i = 1; }
    if (j >= klen) j = 0;
  }
  for (k = 623; k; k--) {
    r->mt[i] = (r->mt[i] ^ ((r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    i++;
    if (i >= 624) { r->mt[0] = r->mt[623]; i = 1; }
  }
  r->mt[0] = 0x80000000u;
  r->idx = 624;
  r->n_draw32 = 0;
}

/* random.seed(n) for a non-negative integer n < 2^64 */
static void cpy_seed_u64(cpy_rng* r, uint64_t seed) {
  uint32_t key[2];
  key[0] = (uint32_t)(seed & 0xffffffffu);
  key[1] = (uint32_t)(seed >> 32);
  cpy_init_by_array(r, key, key[1] ? 2 : 1);
}

static uint32_t cpy_genrand32(cpy_rng* r) {
  if (r->idx >= 624) {
    uint32_t* mt = r->mt;
    int kk;
    for (kk = 0; kk < 624 - 397; kk++) {
      uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < 623; kk++) {
      uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    r->idx = 0;
  }
  uint32_t y = r->mt[r->idx++];
  r->n_draw32++;
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

static double cpy_random(cpy_rng* r) {
  uint32_t a = cpy_genrand32(r) >> 5, b = cpy_genrand32(r) >> 6;
  return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

static double cpy_uniform(cpy_rng* r, double a, double b) { return a + (b - a) * cpy_random(r); }

/* random._randbelow_with_getrandbits(n), 0 < n < 2^32 */
static uint32_t cpy_randbelow(cpy_rng* r, uint32_t n) {
  int k = 0;
  for (uint32_t t = n; t; t >>= 1) k++;
  uint32_t v = cpy_genrand32(r) >> (32 - k);
  while (v >= n) v = cpy_genrand32(r) >> (32 - k);
  return v;
}

#endif
