// Counter-based uniforms and the Gumbel transform used by the sampling kernels.
//
// Everything here is written with single IEEE fp32 multiplies and adds (no fused
// multiply-add, no library log) so that a CPU evaluation of the same expression
// sequence yields bit-identical scores: argmax ties and near-ties then resolve the
// same way on host and device.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BD_HD __host__ __device__ __forceinline__
#else
#define BD_HD static inline
#endif

// ---- Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3") ----
struct bd_philox_out { uint32_t v[4]; };

BD_HD bd_philox_out bd_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                     uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  bd_philox_out o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// 23 random bits -> uniform strictly inside (0,1): (k + 0.5) * 2^-23, exact in fp32.
BD_HD float bd_u01(uint32_t x) {
  return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f;
}

// Stream layout: counter = (position, sequence index, reverse step t, lane),
// lane 0 = selection key of the position, lane 1 + v/4 = token uniforms v..v+3.
BD_HD float bd_uniform_select(uint64_t seed, uint32_t seq, uint32_t t, uint32_t pos) {
  bd_philox_out o = bd_philox4x32_10(pos, seq, t, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  return bd_u01(o.v[0]);
}
// Corruption stream q(x_t | x_0): one block per four neighbouring positions, counter
// (pos / 4, sequence index, t, 2^31), word pos % 4 (lanes below 2^31 belong to the sampler).
BD_HD bd_philox_out bd_corrupt_block(uint64_t seed, uint32_t seq, uint32_t t, uint32_t pos4) {
  return bd_philox4x32_10(pos4, seq, t, 0x80000000u, (uint32_t)seed, (uint32_t)(seed >> 32));
}
BD_HD float bd_uniform_corrupt(uint64_t seed, uint32_t seq, uint32_t t, uint32_t pos) {
  return bd_u01(bd_corrupt_block(seed, seq, t, pos >> 2).v[pos & 3u]);
}
BD_HD float bd_uniform_token(uint64_t seed, uint32_t seq, uint32_t t, uint32_t pos, uint32_t v) {
  bd_philox_out o = bd_philox4x32_10(pos, seq, t, 1u + (v >> 2), (uint32_t)seed, (uint32_t)(seed >> 32));
  return bd_u01(o.v[v & 3u]);
}

// ---- training dropout: keep decisions as a pure function of (seed, step, site, row, column).
// One Philox block, counter (column / 8, row, site, step), serves eight neighbouring columns:
// column c takes 16 bits of it, the low (c even) or high (c odd) half of word (c % 8) / 2, and is
// kept iff that number >= thr = floor(p * 2^16); kept elements are scaled by 1 / (1 - p).
// thr == 0 keeps everything (scale 1): dropout off.
struct bd_drop_args { uint32_t thr, k0, k1, site, step; float scale; };

BD_HD bd_philox_out bd_drop_block(const bd_drop_args& d, uint32_t row, uint32_t col8) {
  return bd_philox4x32_10(col8, row, d.site, d.step, d.k0, d.k1);
}
// factor of column j (0..7) of the block
BD_HD float bd_drop_factor(const bd_drop_args& d, const bd_philox_out& w, int j) {
  const uint32_t h = (w.v[j >> 1] >> (16 * (j & 1))) & 0xffffu;
  return h >= d.thr ? d.scale : 0.0f;
}

// -ln(x) for finite x > 0, fp32 multiplies/adds only (polynomial after Cephes logf).
BD_HD float bd_neglog(float x) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  union { float f; uint32_t u; } b; b.f = x;
  int e = (int)(b.u >> 23) - 127;
  b.u = (b.u & 0x007fffffu) | 0x3f800000u;
  float m = b.f;
  if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
  const float f = m - 1.0f;
  const float z = f * f;
  float y = 7.0376836292E-2f;
  y = y * f; y = y + -1.1514610310E-1f;
  y = y * f; y = y + 1.1676998740E-1f;
  y = y * f; y = y + -1.2420140846E-1f;
  y = y * f; y = y + 1.4249322787E-1f;
  y = y * f; y = y + -1.6668057665E-1f;
  y = y * f; y = y + 2.0000714765E-1f;
  y = y * f; y = y + -2.4999993993E-1f;
  y = y * f; y = y + 3.3333331174E-1f;
  y = y * f; y = y * z;
  const float hz = 0.5f * z;
  y = y - hz;
  float r = f + y;
  const float el = (float)e * 0.693147180559945f;
  r = r + el;
  return -r;
}

// Gumbel(0,1) variate from a uniform: -ln(-ln u), u clamped into (0,1).
BD_HD float bd_gumbel(float u) {
  if (!(u > 2.98023223876953125e-08f)) u = 2.98023223876953125e-08f;
  if (u > 0.99999994f) u = 0.99999994f;
  return bd_neglog(bd_neglog(u));
}
