// Keep-mask streams of the train-mode dropout sites (networks/vit_set.py:28-30, 43/62, 49, 187) — the definition is in
// include/stedm_hip.h ("train-mode dropout"); the parity tests rebuild the same streams in numpy.
#pragma once
#include <stdint.h>

namespace stedm {

struct U4 { uint32_t x, y, z, w; };

// Philox4x32-10 (Salmon et al., SC'11): counter c, key (k0, k1)
__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// elementwise sites: the eight 16-bit uniforms of elements 8 g .. 8 g + 7 (field j = half (j & 1) of word j >> 1)
__device__ __forceinline__ U4 drop_group(uint64_t g, uint32_t site, uint64_t seed) {
  return philox4x32_10(U4{(uint32_t)g, (uint32_t)(g >> 32), site, 0u}, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __forceinline__ uint32_t drop_u16(const U4& r, int j) {
  const uint32_t w = (j >> 1) == 0 ? r.x : (j >> 1) == 1 ? r.y : (j >> 1) == 2 ? r.z : r.w;
  return (w >> (16 * (j & 1))) & 0xFFFFu;
}

// attention site: one xorshift128 stream (Marsaglia 2003) per (query, sample-head, key half), seeded by Philox
__device__ __forceinline__ U4 attn_stream_init(uint32_t q, uint32_t bh, uint32_t site, uint32_t h, uint64_t seed) {
  return philox4x32_10(U4{q, bh, site, h}, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __forceinline__ uint32_t xs128_next(U4& s) {
  const uint32_t t = s.x ^ (s.x << 11);
  s.x = s.y; s.y = s.z; s.z = s.w;
  s.w = s.w ^ (s.w >> 19) ^ t ^ (t >> 8);
  return s.w;
}
// keep bits of one 64-key tile: bit i = (u_i >= thr16); the 16 words drawn are the bit-planes of 32 uniforms, most significant first.
// thr16 is wave-uniform: the plane loop is scalar-branched (bit-sliced comparator, 1-3 vector ops per plane)
__device__ __forceinline__ uint32_t attn_keep_bits(U4& s, uint32_t thr16) {
  uint32_t lt = 0u, eq = ~0u;
#pragma unroll
  for (int k = 15; k >= 0; --k) {
    const uint32_t b = xs128_next(s);
    if (thr16 & (1u << k)) { lt |= eq & ~b; eq &= b; }
    else eq &= ~b;
  }
  return ~lt;
}

}  // namespace stedm
