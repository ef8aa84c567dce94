// Training-step kernels (gfx950): masked cross-entropy forward/backward, LayerNorm backward,
// weight-gradient GEMM, bias column sums, attention backward, embedding backward, global
// gradient norm and the fused AdamW update.
#include "bd_common.h"
#include <stdlib.h>
#include <mutex>
#include "bd_drop.h"
#include "bidiff.h"
#include "bd_prof.h"

namespace {

// =====================================================================================
// Masked cross-entropy. Only rows with a non-zero weight (masked by the corruption AND not
// padding) are read; sums[0] += sum(ce * w), sums[1] += sum(w).
// =====================================================================================
__device__ __forceinline__ float row_weight(const unsigned char* ind, const long long* attn, long long m) {
  if (!ind[m]) return 0.f;
  return attn ? (float)attn[m] : 1.f;
}

constexpr int BD_CE_PART_STRIDE = 32;   // floats between the partial pairs of the sums buffer: one 128-byte line each
constexpr int CE_ROWS = 32;   // rows per workgroup: one pair of same-address atomics per 32 rows
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits,
                                                     const long long* __restrict__ targets,
                                                     const unsigned char* __restrict__ ind,
                                                     const long long* __restrict__ attn,
                                                     float* __restrict__ lse_out, float* __restrict__ sums,
                                                     int M, int V, int n_parts) {
  __shared__ float part[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float lsum = 0.f, wsum = 0.f;
  // The workgroup's rows with a weight are listed first (weights and targets fetched by 32 lanes at once), then dealt
  // to the four waves in turn: a wave's rows are a chain of load - reduce - exp - reduce each, and the kernel lasts as
  // long as its longest chain (rows taken in fixed places, one wave in a thousand met six live rows of eight).
  __shared__ float sw[CE_ROWS];
  __shared__ int st[CE_ROWS], srow[CE_ROWS], n_live;
  const int base = blockIdx.x * CE_ROWS;
  if (wave == 0) {
    float w = 0.f;
    int tg = 0;
    const int row = base + lane;
    if (lane < CE_ROWS && row < M) {
      w = row_weight(ind, attn, row);
      if (w != 0.f) tg = (int)targets[row];
      else lse_out[row] = 0.f;
    }
    const unsigned long long live = __ballot(w != 0.f);
    if (w != 0.f) {
      const int pos = __popcll(live & ((1ull << lane) - 1ull));
      sw[pos] = w; st[pos] = tg; srow[pos] = row;
    }
    if (lane == 0) n_live = __popcll(live);
  }
  __syncthreads();
  const int nl = n_live;
  for (int idx = wave; idx < nl; idx += 4) {
    const int row = srow[idx];
    const float w = sw[idx];
    const float* x = logits + (size_t)row * V;
    const float xt = x[st[idx]];
    float mx = -INFINITY, se = 0.f;
    if (V <= 1024) {          // the row in registers: one pass over memory
      float r[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] = (lane + 64 * q) < V ? x[lane + 64 * q] : -INFINITY;
#pragma unroll
      for (int q = 0; q < 16; ++q) mx = fmaxf(mx, r[q]);
      mx = bd_wave_max(mx);
#pragma unroll
      for (int q = 0; q < 16; ++q) se += __expf(r[q] - mx);     // exp(-inf) = 0 past the row
    } else {
      for (int v = lane; v < V; v += 64) mx = fmaxf(mx, x[v]);
      mx = bd_wave_max(mx);
      for (int v = lane; v < V; v += 64) se += __expf(x[v] - mx);
    }
    se = bd_wave_sum(se);
    const float lse = mx + __logf(se);
    lsum += (lse - xt) * w;
    wsum += w;
    if (lane == 0) lse_out[row] = lse;
  }
  if (lane == 0) { part[wave] = lsum; part[4 + wave] = wsum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float a = (part[0] + part[1]) + (part[2] + part[3]);
    const float b = (part[4] + part[5]) + (part[6] + part[7]);
    // n_parts > 0: the pair goes to one of n_parts pairs behind sums[0..1], each in a cache line of its own
    // (bd_masked_ce_finish adds them up): atomics on one LINE retire one after another, about 75 ns each, and every
    // workgroup ends with this pair
    float* dst = n_parts > 0 ? sums + BD_CE_PART_STRIDE * (1 + blockIdx.x % n_parts) : sums;
    if (b != 0.f) { atomicAdd(&dst[0], a); atomicAdd(&dst[1], b); }
  }
}

// loss = sums[0] / sums[1], an exact zero when no row carried weight
__global__ void ce_finish_kernel(float* __restrict__ sums, float* __restrict__ loss, int n_parts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float a = sums[0], b = sums[1];
    for (int k = 0; k < n_parts; ++k) { a += sums[BD_CE_PART_STRIDE * (1 + k)]; b += sums[BD_CE_PART_STRIDE * (1 + k) + 1]; }
    if (n_parts > 0) { sums[0] = a; sums[1] = b; }      // the backward pass reads the totals here
    loss[0] = b > 0.f ? a / b : 0.f;
  }
}

// dlogits = g * w / count * (softmax - onehot); zero rows where w == 0 or count == 0
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits,
                                                     const long long* __restrict__ targets,
                                                     const unsigned char* __restrict__ ind,
                                                     const long long* __restrict__ attn,
                                                     const float* __restrict__ lse, const float* __restrict__ sums,
                                                     const float* __restrict__ gout, float* __restrict__ dlogits,
                                                     int M, int V) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float cnt = sums[1];
  const float w = row_weight(ind, attn, row);
  float* d = dlogits + (size_t)row * V;
  if (w == 0.f || cnt == 0.f) {
    for (int v = lane; v < V; v += 64) d[v] = 0.f;
    return;
  }
  const float coef = gout[0] * w / cnt;
  const float* x = logits + (size_t)row * V;
  const float l = lse[row];
  const int tgt = (int)targets[row];
  for (int v = lane; v < V; v += 64) d[v] = coef * (__expf(x[v] - l) - (v == tgt ? 1.f : 0.f));
}

// =====================================================================================
// LayerNorm backward: dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma;
// dgamma += sum dy * xhat, dbeta += sum dy. One wave per row, 16 rows per wave, 8 waves per workgroup;
// the parameter gradients are summed over the workgroup's 128 rows in LDS before they go to
// memory, so each address receives one atomic per 128 rows.
// =====================================================================================
template <int NV>
__global__ __launch_bounds__(512) void layernorm_bwd_kernel(const unsigned short* __restrict__ dy,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma,
                                                            const float* dres, float* dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int M, int d, unsigned short* __restrict__ dx_bf16,
                                                            bd_drop_args drop, int rpw, int n_parts, int part_stride) {
  // rpw rows per wave (host: enough workgroups to fill the chip), walked two at a time so that the second row's
  // loads and reductions overlap the first's (a wave that handles its rows one after another spends its time
  // waiting: each row is a load, two dependent wave reductions and a store)
  extern __shared__ float red[];   // [8 waves][2][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = d / 4;
  f32x4 gm[NV], ag[NV], ab[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = lane + 64 * j;
    gm[j] = i < nv ? ((const f32x4*)gamma)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    ag[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ab[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int row0 = (blockIdx.x * 8 + wave) * rpw;
  for (int rr = 0; rr < rpw; rr += 2) {
    f32x4 xh[2][NV], g[2][NV];
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, rstd[2] = {0.f, 0.f};
    bool on[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = row0 + rr + u;
      on[u] = rr + u < rpw && row < M;
      if (!on[u]) continue;
      const float mean = stats[2 * (size_t)row];
      rstd[u] = stats[2 * (size_t)row + 1];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) {
          const f32x4 xv = ((const f32x4*)(x + (size_t)row * d))[i];
          const u32x2 dw = *(const u32x2*)(dy + (size_t)row * d + 4 * i);
          f32x4 dv;
          dv[0] = __uint_as_float(dw[0] << 16); dv[1] = __uint_as_float(dw[0] & 0xffff0000u);
          dv[2] = __uint_as_float(dw[1] << 16); dv[3] = __uint_as_float(dw[1] & 0xffff0000u);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xh[u][j][e] = (xv[e] - mean) * rstd[u];
            g[u][j][e] = dv[e] * gm[j][e];
            s1[u] += g[u][j][e];
            s2[u] += g[u][j][e] * xh[u][j][e];
            ag[j][e] += dv[e] * xh[u][j][e];
            ab[j][e] += dv[e];
          }
        } else {
          xh[u][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          g[u][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    // the four wave sums of the two rows, interleaved
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s1[0] += __shfl_xor(s1[0], o, 64); s2[0] += __shfl_xor(s2[0], o, 64);
      s1[1] += __shfl_xor(s1[1], o, 64); s2[1] += __shfl_xor(s2[1], o, 64);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!on[u]) continue;
      const int row = row0 + rr + u;
      const float m1 = s1[u] / (float)d, m2 = s2[u] / (float)d;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rstd[u] * (g[u][j][e] - m1 - xh[u][j][e] * m2);
          if (dres != nullptr) {
            const f32x4 r = ((const f32x4*)(dres + (size_t)row * d))[i];
            o[0] += r[0]; o[1] += r[1]; o[2] += r[2]; o[3] += r[3];
          }
          if (dx_bf16 == nullptr && drop.thr != 0u) {      // no bf16 copy: the dropout applies to dx itself
            const bd_philox_out w = bd_drop_block(drop, (uint32_t)row, (uint32_t)(i >> 1));
            const int j0 = 4 * (i & 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] *= bd_drop_factor(drop, w, j0 + e);
          }
          ((f32x4*)(dx + (size_t)row * d))[i] = o;
          if (dx_bf16 != nullptr) {      // the next branch's incoming gradient: bf16 copy, optionally dropped
            if (drop.thr != 0u) {
              const bd_philox_out w = bd_drop_block(drop, (uint32_t)row, (uint32_t)(i >> 1));
              const int j0 = 4 * (i & 1);
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] *= bd_drop_factor(drop, w, j0 + e);
            }
            u32x2 pk;
            pk[0] = bd_pack_bf16x2(o[0], o[1]); pk[1] = bd_pack_bf16x2(o[2], o[3]);
            *(u32x2*)(dx_bf16 + (size_t)row * d + 4 * i) = pk;
          }
        }
      }
    }
  }
  // the eight waves' sums side by side in LDS (plain stores), then one thread per column adds them up: LDS atomics
  // from eight waves on the same 2 d addresses serialise
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = lane + 64 * j;
    if (i < nv) {
      ((f32x4*)(red + (size_t)wave * 2 * d))[i] = ag[j];
      ((f32x4*)(red + (size_t)wave * 2 * d + d))[i] = ab[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * d; i += 512) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += red[(size_t)w * 2 * d + i];
    red[i] = t;       // column i is read and written by this thread only
  }
  __syncthreads();
  // n_parts > 1: the sums go to one of n_parts copies of the two rows (part_stride floats apart), which the caller
  // adds up later: atomics on one address retire one after another, and with one copy every address takes one from
  // every workgroup, all at the end of the kernel
  const size_t part = (size_t)(blockIdx.x % n_parts) * part_stride;
  for (int i = threadIdx.x; i < d; i += 512) {
    atomicAdd(dgamma + part + i, red[i]);
    atomicAdd(dbeta + part + i, red[d + i]);
  }
}

// dgamma += sum of the parts' first rows, dbeta += sum of their second rows; the parts are left zeroed
struct LnPartsJob { float* parts; float* dgamma; float* dbeta; };
__global__ __launch_bounds__(256) void ln_parts_finish_kernel(const LnPartsJob* __restrict__ jobs, int n_parts, int d) {
  // one thread per element of a job's two rows; the copies' values are all requested before the first is used
  const LnPartsJob job = jobs[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * d) return;
  float* const src = job.parts + i;
  const size_t stride = 2 * (size_t)d;
  float t = 0.f;
  int k = 0;
  for (; k + 8 <= n_parts; k += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(k + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) { t += v[u]; src[(k + u) * stride] = 0.f; }
  }
  for (; k < n_parts; ++k) { t += src[k * stride]; src[k * stride] = 0.f; }
  float* dst = i < d ? job.dgamma + i : job.dbeta + (i - d);
  *dst += t;
}

// =====================================================================================
// Weight-gradient GEMM: dW[N,K] += dY[M,N]^T * X[M,K]  (bf16 in, fp32 atomic accumulate).
// Both operands are token-major in memory while the MFMA wants the token index contiguous
// per lane: tiles are staged row-major into LDS and read with the transposing LDS read.
// 128 x 128 output tile per workgroup, 64 tokens per step, the token range split over
// gridDim.y workgroups.
// =====================================================================================
__device__ __forceinline__ int tile_off(int row, int ch) {   // 256-byte rows, swizzled 16-byte chunks
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
__device__ __forceinline__ s16x4 lds_read_tr(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

__global__ __launch_bounds__(256, 2) void wgrad_kernel(const unsigned short* __restrict__ dY, int ldy,
                                                       const unsigned short* __restrict__ X, int ldx,
                                                       float* __restrict__ dW, int M, int N, int K,
                                                       int tiles_k, int tiles, int tokens_per_split) {
  constexpr int TB = 64 * 256;   // bytes of one 64-token x 128-column tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TB];
  unsigned char* const Ys = smem;
  unsigned char* const Xs = smem + 2 * TB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int g = lane >> 4, c = lane & 15;
  // Workgroups are dealt round-robin to the 8 XCDs (id % 8). All tiles of one token range go to
  // the SAME XCD, next to each other in time, so that range of dY and X is fetched into that
  // XCD's L2 once and re-read from there by the other tiles.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = slot % tiles, split_id = (slot / tiles) * 8 + xcd;
  const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 128;
  const int m_begin = split_id * tokens_per_split;
  int m_end = m_begin + tokens_per_split;
  m_end = m_end < M ? m_end : M;
  if (m_begin >= m_end) return;
  const int steps = (m_end - m_begin + 63) / 64;

  // staging: thread -> rows (tid/16 + 16 i), chunk tid%16
  const int srow = tid >> 4, sch = tid & 15;
  u32x4 ry[4], rx[4];
  auto gload = [&](int st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m_begin + st * 64 + srow + 16 * i;
      if (m < m_end) {
        ry[i] = *(const u32x4*)(dY + (size_t)m * ldy + n0 + sch * 8);
        rx[i] = *(const u32x4*)(X + (size_t)m * ldx + k0 + sch * 8);
      } else {
        ry[i] = (u32x4){0u, 0u, 0u, 0u};
        rx[i] = (u32x4){0u, 0u, 0u, 0u};
      }
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = tile_off(srow + 16 * i, sch);
      *(u32x4*)(Ys + buf * TB + off) = ry[i];
      *(u32x4*)(Xs + buf * TB + off) = rx[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  gload(0);
  swrite(0);
  __syncthreads();
  const int q4 = c >> 2, p4 = c & 3;
  for (int st = 0; st < steps; ++st) {
    const int buf = st & 1;
    const bool more = st + 1 < steps;
    if (more) gload(st + 1);
    const unsigned char* Yb = Ys + buf * TB;
    const unsigned char* Xb = Xs + buf * TB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int r0 = 32 * kk + 8 * g + q4;
      bf16x8 fy[4], fx[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int chy = 8 * wn + 2 * t + (p4 >> 1), chx = 8 * wk + 2 * t + (p4 >> 1);
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lds_read_tr(Yb + tile_off(r0, chy) + 8 * (p4 & 1));
        u.s.b = lds_read_tr(Yb + tile_off(r0 + 4, chy) + 8 * (p4 & 1));
        fy[t] = u.v;
        u.s.a = lds_read_tr(Xb + tile_off(r0, chx) + 8 * (p4 & 1));
        u.s.b = lds_read_tr(Xb + tile_off(r0 + 4, chx) + 8 * (p4 & 1));
        fx[t] = u.v;
      }
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
          acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[tn], fx[tk], acc[tn][tk], 0, 0, 0);
    }
    if (more) swrite(buf ^ 1);
    __syncthreads();
  }
  // lane (g, c): n = n0 + 64 wn + 16 tn + 4 g + r, k = k0 + 64 wk + 16 tk + c
#pragma unroll
  for (int tn = 0; tn < 4; ++tn)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 64 * wn + 16 * tn + 4 * g + r;
      float* o = dW + (size_t)n * K + k0 + 64 * wk + c;
#pragma unroll
      for (int tk = 0; tk < 4; ++tk) atomicAdd(o + 16 * tk, acc[tn][tk][r]);
    }
}

// -------------------------------------------------------------------------------------
// The same product with the operand tiles streamed by LDS-DMA through a 4-stage ring (three
// 32-token steps in flight per workgroup, two workgroups per CU): the register-staged kernel
// above has one step in flight and is bound by the load latency. Needs M % 32 == 0.
//  * Output rows may be split into up to three segments with their own destinations (the stacked
//    q | k | v gradient goes to three parameters in one launch), and the column sums of dY (bias
//    gradients) come from one extra MFMA per tile against an all-ones operand.
//  * Work split: XCD x (workgroup id % 8) owns the 32-token chunks q = x (mod 8); inside it the
//    chunks are dealt to J groups; workgroup (x, tile, j) walks chunks x + 8 (j + J i). All
//    tiles of a group walk the same chunks at the same time, so a chunk of dY and X enters that
//    XCD's L2 once. One atomic epilogue per workgroup.
// -------------------------------------------------------------------------------------
struct WgradArgs {
  const unsigned short* dY;
  const unsigned short* X;
  float* dW[3];
  float* db[3];
  int ldy, ldx, seg_rows;
  int N, K, tiles_k, tiles, J, n_chunks;
};
constexpr int WG_STAGES = 4;
constexpr int WG_TILE = 32 * 256;        // 32 tokens x 128 bf16 columns
constexpr int WG_STAGE = 2 * WG_TILE;

typedef __attribute__((address_space(1))) void wg_glb_void;
typedef __attribute__((address_space(3))) void wg_lds_void;

// one output tile of one product over the 32-token chunks q0, q0 + qstride, ...
__device__ __forceinline__ void wgrad_dma_tile(const WgradArgs& p, int tile, int q0, int qstride, unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int g = lane >> 4, c = lane & 15;
  const int n0 = (tile / p.tiles_k) * 128, k0 = (tile % p.tiles_k) * 128;
  if (q0 >= p.n_chunks) return;
  const int steps = (p.n_chunks - q0 + qstride - 1) / qstride;

  // LDS-DMA: one wave instruction fills 1 KiB = 4 token rows; lane -> row lane / 16, physical
  // 16-byte chunk lane % 16, which holds global chunk (lane % 16) ^ swizzle(row). For the rows
  // 4 f + lane / 16 with f = wave or wave + 4 the swizzle ((row & 3) << 2) | ((row >> 2) & 3) is
  // ((lane / 16) << 2) | wave: the same for both of the wave's instructions.
  const int ch = (lane & 15) ^ (((lane >> 4) << 2) | wave);
  const unsigned short* gy = p.dY + (size_t)(4 * wave + (lane >> 4)) * p.ldy + n0 + ch * 8;
  const unsigned short* gx = p.X + (size_t)(4 * wave + (lane >> 4)) * p.ldx + k0 + ch * 8;
  auto issue = [&](int i) {
    const size_t m0 = (size_t)(q0 + i * qstride) * 32;
    unsigned char* st = smem + (i % WG_STAGES) * WG_STAGE + wave * 1024;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      __builtin_amdgcn_global_load_lds((wg_glb_void*)(gy + (m0 + 16 * u) * p.ldy), (wg_lds_void*)(st + u * 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((wg_glb_void*)(gx + (m0 + 16 * u) * p.ldx), (wg_lds_void*)(st + WG_TILE + u * 4096), 16, 0, 0);
    }
  };

  f32x4 acc[4][4], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int seg = n0 / p.seg_rows;
  const bool want_bias = (k0 == 0) && (wk == 0) && p.db[seg] != nullptr;   // wave-uniform
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

#pragma unroll
  for (int i = 0; i < WG_STAGES - 1; ++i)
    if (i < steps) issue(i);
  // LDS reads as inline asm with their own wait: the compiler would make every LDS read wait for
  // ALL outstanding LDS-DMA loads (it cannot tell the stages apart), which serialises the ring.
  const int q4 = c >> 2, p4 = c & 3;
  const int r0 = 8 * g + q4;
  unsigned ay[4][2], ax[4][2];   // byte offsets inside a stage, then absolute LDS addresses
  const unsigned lds0 = (unsigned)(size_t)(wg_lds_void*)smem;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int chy = 8 * wn + 2 * t + (p4 >> 1), chx = 8 * wk + 2 * t + (p4 >> 1);
    ay[t][0] = lds0 + tile_off(r0, chy) + 8 * (p4 & 1);
    ay[t][1] = lds0 + tile_off(r0 + 4, chy) + 8 * (p4 & 1);
    ax[t][0] = lds0 + WG_TILE + tile_off(r0, chx) + 8 * (p4 & 1);
    ax[t][1] = lds0 + WG_TILE + tile_off(r0 + 4, chx) + 8 * (p4 & 1);
  }
  for (int st = 0; st < steps; ++st) {
    const int ahead = steps - 1 - st;          // steps already issued beyond this one (capped at 2)
    // my part of this stage has landed; after the barrier everybody's has, and everybody has left
    // the stage of the previous step, which the next issue overwrites
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (st + WG_STAGES - 1 < steps) issue(st + WG_STAGES - 1);
    const unsigned so = (st % WG_STAGES) * WG_STAGE;
    union Frag { struct { s16x4 a, b; } s; bf16x8 v; };
    Frag fy[4], fx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fy[t].s.a) : "v"(ay[t][0] + so));
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fy[t].s.b) : "v"(ay[t][1] + so));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fx[t].s.a) : "v"(ax[t][0] + so));
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fx[t].s.b) : "v"(ax[t][1] + so));
    }
    // every register written by the reads above passes through the wait before its first use
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fy[0].v), "+v"(fy[1].v), "+v"(fy[2].v), "+v"(fy[3].v), "+v"(fx[0].v), "+v"(fx[1].v),
                   "+v"(fx[2].v), "+v"(fx[3].v));
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int tk = 0; tk < 4; ++tk)
        acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[tn].v, fx[tk].v, acc[tn][tk], 0, 0, 0);
    if (want_bias) {
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) accb[tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[tn].v, ones, accb[tn], 0, 0, 0);
    }
  }
  // lane (g, c): n = n0 + 64 wn + 16 tn + 4 g + r, k = k0 + 64 wk + 16 tk + c
  float* const dW = p.dW[seg] - (size_t)seg * p.seg_rows * p.K;
#pragma unroll
  for (int tn = 0; tn < 4; ++tn)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 64 * wn + 16 * tn + 4 * g + r;
      float* o = dW + (size_t)n * p.K + k0 + 64 * wk + c;
#pragma unroll
      for (int tk = 0; tk < 4; ++tk) atomicAdd(o + 16 * tk, acc[tn][tk][r]);
    }
  if (want_bias && c == 0) {   // every column of the ones product holds the same sums
    float* const db = p.db[seg] - (size_t)seg * p.seg_rows;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(db + n0 + 64 * wn + 16 * tn + 4 * g + r, accb[tn][r]);
  }
}

__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(WgradArgs p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[WG_STAGES * WG_STAGE];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = slot % p.tiles, j = slot / p.tiles;
  wgrad_dma_tile(p, tile, xcd + 8 * j, 8 * p.J, smem);
}

// One product, every output tile owned by ONE workgroup that walks all the token chunks (large outputs over a short
// token range; bd_gemm_wgrad_segments chooses). Tiles are dealt so that the workgroups of an XCD (blockIdx % 8) hold
// neighbouring tiles of the list: the same rows of dY, neighbouring columns of X.
__global__ __launch_bounds__(256, 2) void wgrad_single_kernel(WgradArgs p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[WG_STAGES * WG_STAGE];
  const int n_all = gridDim.x, xcd = blockIdx.x & 7, q = n_all >> 3, rr = n_all & 7;
  const int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (blockIdx.x >> 3);
  wgrad_dma_tile(p, tile, 0, 1, smem);
}

// Several products in one launch, every output tile owned by ONE workgroup that walks all the token chunks: the launch
// keeps (sum of tiles) workgroups busy for the length of the token range instead of filling the chip in bursts, which
// is what a stream running beside the data-gradient chain should do; and with a single writer per output element the
// sums do not depend on the order in which workgroups finish.
constexpr int WG_GROUP_MAX = 4;
struct WgradGroupArgs {
  WgradArgs a[WG_GROUP_MAX];
  int first_tile[WG_GROUP_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256, 2) void wgrad_group_kernel(WgradGroupArgs gp) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[WG_STAGES * WG_STAGE];
  const int t = blockIdx.x;
  int i = 0;
#pragma unroll
  for (int k = 1; k < WG_GROUP_MAX; ++k)
    if (k < gp.n && t >= gp.first_tile[k]) i = k;
  // (a run-time index into the argument array, or a copy of one entry, goes through scratch memory: one call per entry)
  if (i == 0) wgrad_dma_tile(gp.a[0], t - gp.first_tile[0], 0, 1, smem);
  else if (i == 1) wgrad_dma_tile(gp.a[1], t - gp.first_tile[1], 0, 1, smem);
  else if (i == 2) wgrad_dma_tile(gp.a[2], t - gp.first_tile[2], 0, 1, smem);
  else wgrad_dma_tile(gp.a[3], t - gp.first_tile[3], 0, 1, smem);
}

// out[n] += sum_m dY[m, n]   (bias gradients). A workgroup owns 64 columns x up to 512 rows:
// thread = (8-column group, row lane); a row lane walks its rows 32 apart, the 32 row lanes are
// summed in LDS, then 64 atomics per workgroup go to memory.
constexpr int CS_ROWS = 512;
__global__ __launch_bounds__(256) void colsum_kernel(const unsigned short* __restrict__ dY, int ld,
                                                     float* __restrict__ out, int M, int N) {
  __shared__ float red[64];
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int col = blockIdx.x * 64 + cg * 8;
  const int m0 = blockIdx.y * CS_ROWS;
  const int m1 = m0 + CS_ROWS < M ? m0 + CS_ROWS : M;
  if (threadIdx.x < 64) red[threadIdx.x] = 0.f;
  __syncthreads();
  if (col < N) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int m = m0 + rl; m < m1; m += 32) {
      const u32x4 w = *(const u32x4*)(dY + (size_t)m * ld + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[2 * e] += __uint_as_float(w[e] << 16);
        s[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
      }
    }
    // the 8 row lanes of a wave that share a column group sit 8 lanes apart
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = s[e];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if ((threadIdx.x & 63) < 8) atomicAdd(red + cg * 8 + e, v);
    }
  }
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < N) atomicAdd(out + blockIdx.x * 64 + threadIdx.x, red[threadIdx.x]);
}

// =====================================================================================
// Attention backward, two kernels that each recompute the probabilities from the saved
// log-sum-exp (no atomics, deterministic):
//   dq kernel : a wave owns 16 queries, sweeps key tiles -> dQ, and delta = rowsum(dO * O)
//   dkv kernel: a wave owns 16 keys, sweeps query tiles  -> dK, dV
// Operand orientation mirrors the forward kernel (everything transposed so that the
// owned index sits on the lane).
// =====================================================================================
constexpr int AT = 128;     // rows per staged tile
constexpr int AROW = 256;   // LDS row stride in bytes

typedef __attribute__((address_space(3))) void bwd_lds;

// A 128-row tile of an operand to LDS by LDS-DMA, all of a wave's pieces in flight at once (a 1-KiB piece = 4 rows x
// 256 B; lane i lands at row i / 16, slot i % 16 and fetches the chunk the swizzle keeps there): one memory latency
// per tile. (Through registers a thread's loads and LDS stores alternate and every piece waits out a latency of its
// own; holding all pieces in registers first spills in these kernels.) Rows at or past `limit` read row limit - 1 instead of zeros (their probabilities are exactly 0, and
// 0 times a finite row is 0); 32-row groups not wanted are not fetched. `src` = the operand's row `0` of this
// sequence and head (wave-uniform), `wave` wave-uniform. The caller waits (s_waitcnt vmcnt(0)) before its barrier.
template <int DH>
__device__ __forceinline__ void stage_rows_dma(unsigned char* dst, const unsigned short* src, size_t row_stride, int first,
                                               int limit, int lane, int wave, unsigned groups) {
  constexpr int CPR = DH / 8;
  const unsigned lds = (unsigned)(size_t)(bwd_lds*)dst;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int piece = 8 * wave + j;                      // rows 4 piece .. 4 piece + 3: inside one 32-row group
    if (!((groups >> (piece >> 3)) & 1u)) continue;
    const int row = 4 * piece + (lane >> 4);
    int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    ch = ch < CPR ? ch : CPR - 1;
    const int r = min(first + row, limit - 1);
    const unsigned voff = (unsigned)(((size_t)r * row_stride + ch * 8) * 2);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 ::"s"(lds + piece * 1024), "v"(voff), "s"(src) : "memory", "m0");
  }
}

template <int DH, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const unsigned short* __restrict__ qkv,
                                                             const unsigned char* __restrict__ keymask,
                                                             const unsigned short* __restrict__ O,
                                                             const unsigned short* __restrict__ dO,
                                                             const float* __restrict__ lse,
                                                             float* __restrict__ delta,
                                                             unsigned short* __restrict__ dqkv, int Ls, int H,
                                                             float scale, bd_drop_args drop, const int* __restrict__ cu) {
  // Ls = row stride of the [B, Ls] key mask, of lse / delta [B, H, Ls] and of the dropout rows; with cu (packed rows:
  // sequence b = rows cu[b] .. cu[b + 1] - 1 of qkv / O / dO / dqkv) L is the sequence's own length
  const int L = cu ? cu[blockIdx.z + 1] - cu[blockIdx.z] : Ls;
  if ((int)blockIdx.x * 64 >= L) return;
  const size_t tb = cu ? (size_t)cu[blockIdx.z] : (size_t)blockIdx.z * Ls;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * AT * AROW + AT];
  unsigned char* const Ks = smem;
  unsigned char* const Vs = smem + AT * AROW;
  unsigned char* const Ms = smem + 2 * AT * AROW;
  constexpr int KS = DH / 32, DT = DH / 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int d = H * DH;
  const size_t row3 = (size_t)3 * d;
  const int q0 = blockIdx.x * 64 + wave * 16;
  int qr = q0 + c;
  const bool qok = qr < L;
  qr = qok ? qr : L - 1;
  const size_t tok = tb + qr;

  bf16x8 qf[KS], dof[KS];
  float dl = 0.f;
  {
    const unsigned short* qp = qkv + tok * row3 + (size_t)h * DH + 8 * g;
    const unsigned short* op = O + tok * d + (size_t)h * DH + 8 * g;
    const unsigned short* dp = dO + tok * d + (size_t)h * DH + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = *(const bf16x8*)(qp + 32 * ks);
      dof[ks] = *(const bf16x8*)(dp + 32 * ks);
      const bf16x8 ov = *(const bf16x8*)(op + 32 * ks);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += (float)dof[ks][e] * (float)ov[e];
    }
  }
  dl += __shfl_xor(dl, 16, 64);
  dl += __shfl_xor(dl, 32, 64);
  const float lq = lse[((size_t)b * H + h) * Ls + qr];
  // delta_i = sum_j P_ij dP_ij. With all keys in one tile (L <= 128) it is summed from the very P and dP that
  // form dS below, in fp32, so that sum_j dS_ij = 0 holds to rounding: rowsum(dO * O) from the bf16-rounded O
  // misses that by 2^-9 of |dO||O|, which is as large as dS itself when the attention is nearly uniform (the
  // q / k gradients of a freshly initialised model came out 13 % off). Longer sequences keep rowsum(dO * O).
  const bool one_tile = L <= AT;
  if (!one_tile && qok && g == 0) delta[((size_t)b * H + h) * Ls + qr] = dl;

  f32x4 acc[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int q4 = c >> 2, p4 = c & 3;

  for (int kt0 = 0; kt0 < L; kt0 += AT) {
    if (kt0 > 0) __syncthreads();
    // the mask load and the K / V pieces of the groups that begin before the end of the sequence are in flight together
    unsigned char mv = 0;
    if (tid < AT) {
      const int key = kt0 + tid;
      if (key < L) mv = keymask ? (keymask[(size_t)b * Ls + key] != 0) : 1;
    }
    const unsigned by_len = (1u << min(4, (L - kt0 + 31) >> 5)) - 1u;
    stage_rows_dma<DH>(Ks, qkv + tb * row3 + d + (size_t)h * DH, row3, kt0, L, lane, wave, by_len);
    stage_rows_dma<DH>(Vs, qkv + tb * row3 + 2 * d + (size_t)h * DH, row3, kt0, L, lane, wave, by_len);
    if (tid < AT) Ms[tid] = mv;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // 32-key groups without a visible key have P = 0 and dS = 0: not multiplied
    unsigned live = 0u;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const u32x4 m0 = *(const u32x4*)(Ms + 32 * s4), m1 = *(const u32x4*)(Ms + 32 * s4 + 16);
      if ((m0[0] | m0[1] | m0[2] | m0[3] | m1[0] | m1[1] | m1[2] | m1[3]) != 0u) live |= 1u << s4;
    }
    float pv[4][8], dv[4][8];
    float psum = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { pv[s4][e] = 0.f; dv[s4][e] = 0.f; }
      if (!((live >> s4) & 1u)) continue;
      bd_philox_out w;      // one block = the lane's eight keys of this group (both halves)
      if (DROP) w = bd_drop_block(drop, (uint32_t)(((size_t)b * H + h) * Ls + qr), (uint32_t)((kt0 + 32 * s4 + 8 * g) >> 3));
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int krow = 32 * s4 + 8 * (c >> 2) + 4 * half + (c & 3);
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 ka = *(const bf16x8*)(Ks + tile_off(krow, 4 * ks + g));
          const bf16x8 va = *(const bf16x8*)(Vs + tile_off(krow, 4 * ks + g));
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, dof[ks], dp, 0, 0, 0);
        }
        const int kb = 32 * s4 + 8 * g + 4 * half;
        const unsigned int m4 = *(const unsigned int*)(Ms + kb);
        if (DROP) {   // dP = (dO V^T) * keep / (1 - p)
#pragma unroll
          for (int r = 0; r < 4; ++r) dp[r] *= bd_drop_factor(drop, w, 4 * half + r);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool vis = ((m4 >> (8 * r)) & 0xffu) != 0u && lq > -INFINITY;
          const float p = vis ? __expf(s[r] * scale - lq) : 0.f;
          pv[s4][4 * half + r] = p;
          dv[s4][4 * half + r] = dp[r];
          psum += p * dp[r];
        }
      }
    }
    if (one_tile) {
      psum += __shfl_xor(psum, 16, 64);
      psum += __shfl_xor(psum, 32, 64);
      dl = psum;
      if (qok && g == 0) delta[((size_t)b * H + h) * Ls + qr] = dl;
    }
    bf16x8 dsf[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      bf16x8 frag;
#pragma unroll
      for (int e = 0; e < 8; ++e) frag[e] = (__bf16)(pv[s4][e] * (dv[s4][e] - dl) * scale);
      dsf[s4] = frag;
    }
    // dQ^T += K^T dS^T
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (!((live >> s4) & 1u)) continue;
        const int r0 = 32 * s4 + 8 * g + q4, ch = 2 * dt + (p4 >> 1);
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lds_read_tr(Ks + tile_off(r0, ch) + 8 * (p4 & 1));
        u.s.b = lds_read_tr(Ks + tile_off(r0 + 4, ch) + 8 * (p4 & 1));
        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, dsf[s4], acc[dt], 0, 0, 0);
      }
  }
  if (qok) {
    unsigned short* op = dqkv + tok * row3 + (size_t)h * DH + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      u32x2 w;
      w[0] = bd_pack_bf16x2(acc[dt][0], acc[dt][1]);
      w[1] = bd_pack_bf16x2(acc[dt][2], acc[dt][3]);
      *(u32x2*)(op + 16 * dt) = w;
    }
  }
}

template <int DH, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const unsigned short* __restrict__ qkv,
                                                              const unsigned char* __restrict__ keymask,
                                                              const unsigned short* __restrict__ dO,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ delta,
                                                              unsigned short* __restrict__ dqkv, int Ls, int H,
                                                              float scale, bd_drop_args drop, const int* __restrict__ cu) {
  const int L = cu ? cu[blockIdx.z + 1] - cu[blockIdx.z] : Ls;      // see attn_bwd_dq_kernel
  if ((int)blockIdx.x * 64 >= L) return;
  const size_t tb = cu ? (size_t)cu[blockIdx.z] : (size_t)blockIdx.z * Ls;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * AT * AROW + 2 * AT * 4];
  unsigned char* const Qs = smem;
  unsigned char* const Ds = smem + AT * AROW;
  float* const Lq = (float*)(smem + 2 * AT * AROW);
  float* const Dl = Lq + AT;
  constexpr int KS = DH / 32, DT = DH / 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int d = H * DH;
  const size_t row3 = (size_t)3 * d;
  const int k0 = blockIdx.x * 64 + wave * 16;
  int kr = k0 + c;
  const bool kok = kr < L;
  kr = kok ? kr : L - 1;
  const size_t tok = tb + kr;
  const bool vis = kok && (keymask ? keymask[(size_t)b * Ls + kr] != 0 : true);
  if (!__syncthreads_or(vis)) {       // no visible key in this workgroup: dK = dV = 0
    if (kok) {
      unsigned short* op = dqkv + tok * row3 + d + (size_t)h * DH + 4 * g;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        *(u32x2*)(op + 16 * dt) = (u32x2){0u, 0u};
        *(u32x2*)(op + d + 16 * dt) = (u32x2){0u, 0u};
      }
    }
    return;
  }
  const bool wave_live = __any(vis);   // a wave of masked keys still stages and synchronises, but multiplies nothing

  bf16x8 kf[KS], vf[KS];
  {
    const unsigned short* kp = qkv + tok * row3 + d + (size_t)h * DH + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = *(const bf16x8*)(kp + 32 * ks);
      vf[ks] = *(const bf16x8*)(kp + d + 32 * ks);
    }
  }
  f32x4 ak[DT], av[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) { ak[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; av[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const int q4 = c >> 2, p4 = c & 3;

  for (int qt0 = 0; qt0 < L; qt0 += AT) {
    if (qt0 > 0) __syncthreads();
    // 32-query groups past the end of the sequence have P = 0 and dS = 0: not staged, not multiplied
    const int ngrp = min(4, (L - qt0 + 31) >> 5);
    const unsigned qgroups = (1u << ngrp) - 1u;
    stage_rows_dma<DH>(Qs, qkv + tb * row3 + (size_t)h * DH, row3, qt0, L, lane, wave, qgroups);
    stage_rows_dma<DH>(Ds, dO + tb * d + (size_t)h * DH, (size_t)d, qt0, L, lane, wave, qgroups);
    if (tid < AT) {
      const int q = qt0 + tid;
      Lq[tid] = q < L ? lse[((size_t)b * H + h) * Ls + q] : INFINITY;
      Dl[tid] = q < L ? delta[((size_t)b * H + h) * Ls + q] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!wave_live) continue;
    // Keep decisions of the lane's key against this lane's 32 queries of the tile (bit i <-> query 8 g + 32 (i >> 3) +
    // (i & 7)). One Philox block per query holds the decisions of eight neighbouring KEYS, and the eight lanes that
    // own those keys (same g, same c >> 3) need the same 32 blocks: each evaluates four of them, reduces every block
    // to its eight keep bits, and the lanes exchange the bytes.
    uint32_t keepbits = 0u;
    if (DROP) {
      const uint32_t prow0 = (uint32_t)(((size_t)b * H + h) * Ls + qt0 + 8 * g);
      const uint32_t kblk = (uint32_t)((k0 + c) >> 3);
      const int j = c & 7;
      uint32_t mine = 0u;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = 4 * j + t;
        if ((i >> 3) >= ngrp) continue;
        const bd_philox_out w = bd_drop_block(drop, prow0 + 32 * (i >> 3) + (i & 7), kblk);
        uint32_t byte = 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          byte |= (uint32_t)(((w.v[e >> 1] >> (16 * (e & 1))) & 0xffffu) >= drop.thr) << e;
        mine |= byte << (8 * t);
      }
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const uint32_t wv = __shfl(mine, (lane & ~7) | jj, 64) >> ((k0 + c) & 7);
#pragma unroll
        for (int t = 0; t < 4; ++t) keepbits |= ((wv >> (8 * t)) & 1u) << (4 * jj + t);
      }
    }
    bf16x8 pf[4], dsf[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      bf16x8 fp, fd;
      if (s4 >= ngrp) continue;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qrow = 32 * s4 + 8 * (c >> 2) + 4 * half + (c & 3);
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 qa = *(const bf16x8*)(Qs + tile_off(qrow, 4 * ks + g));
          const bf16x8 da = *(const bf16x8*)(Ds + tile_off(qrow, 4 * ks + g));
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[ks], dp, 0, 0, 0);
        }
        const int qb = 32 * s4 + 8 * g + 4 * half;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float lq = Lq[qb + r];
          const bool on = vis && lq > -INFINITY && lq < INFINITY;
          const float p = on ? __expf(s[r] * scale - lq) : 0.f;
          const float z = !DROP ? 1.0f : (((keepbits >> (8 * s4 + 4 * half + r)) & 1u) ? drop.scale : 0.f);
          fp[4 * half + r] = (__bf16)(p * z);
          fd[4 * half + r] = (__bf16)(p * (dp[r] * z - Dl[qb + r]) * scale);
        }
      }
      pf[s4] = fp;
      dsf[s4] = fd;
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (s4 >= ngrp) continue;
        const int r0 = 32 * s4 + 8 * g + q4, ch = 2 * dt + (p4 >> 1);
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lds_read_tr(Ds + tile_off(r0, ch) + 8 * (p4 & 1));
        u.s.b = lds_read_tr(Ds + tile_off(r0 + 4, ch) + 8 * (p4 & 1));
        av[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, pf[s4], av[dt], 0, 0, 0);
        u.s.a = lds_read_tr(Qs + tile_off(r0, ch) + 8 * (p4 & 1));
        u.s.b = lds_read_tr(Qs + tile_off(r0 + 4, ch) + 8 * (p4 & 1));
        ak[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, dsf[s4], ak[dt], 0, 0, 0);
      }
  }
  if (kok) {
    unsigned short* op = dqkv + tok * row3 + d + (size_t)h * DH + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      u32x2 w;
      w[0] = bd_pack_bf16x2(ak[dt][0], ak[dt][1]);
      w[1] = bd_pack_bf16x2(ak[dt][2], ak[dt][3]);
      *(u32x2*)(op + 16 * dt) = w;
      w[0] = bd_pack_bf16x2(av[dt][0], av[dt][1]);
      w[1] = bd_pack_bf16x2(av[dt][2], av[dt][3]);
      *(u32x2*)(op + d + 16 * dt) = w;
    }
  }
}

// =====================================================================================
// Embedding backward: scatter-add the residual-stream gradient into the three tables.
// =====================================================================================
// Embedding backward in two passes over dx, each summing in registers along the direction in which rows share a
// destination (one atomic per sum instead of one per row):
//   embed_bwd_seq_kernel  one workgroup per sequence: the timestep row gets the sum over its positions, each token
//                         row the sum of the sequence's rows with that token
//   embed_bwd_pos_kernel  one workgroup per (position, slice of the batch): the position row gets the sum over sequences
// Rows that are exactly zero (padded positions carry no gradient) add nothing.
constexpr int EB_MAXL = 1024;    // longest sequence of embed_bwd_seq_kernel

__global__ __launch_bounds__(1024) void embed_bwd_seq_kernel(const float* __restrict__ dx, const long long* __restrict__ ids,
                                                            const long long* __restrict__ tsteps, float* __restrict__ dtok,
                                                            float* __restrict__ dtim, int Ls, int d, int pad_id,
                                                            int vocab, int n_timesteps, const int* __restrict__ cu) {
  // packed rows (cu given): sequence b = rows cu[b] .. cu[b + 1] - 1 of dx, its ids at ids[b * Ls + l]
  const int L = cu ? cu[blockIdx.x + 1] - cu[blockIdx.x] : Ls;
  // Rows of one sequence sorted by token id (rank sort in LDS), so that rows sharing a token - the mask token is
  // half of a corrupted sequence - are summed in registers and reach the token's row as one atomic per sequence:
  // same-address atomics retire one after another (about 36 ns each), 5000 mask rows of a batch took 180 us.
  __shared__ int sid[EB_MAXL];
  __shared__ short sord[EB_MAXL];
  const int b = blockIdx.x;
  const float* g = dx + (cu ? (size_t)cu[b] : (size_t)b * Ls) * d;
  const int nthr = blockDim.x;
  for (int l = threadIdx.x; l < L; l += nthr) sid[l] = (int)ids[(size_t)b * Ls + l];
  __syncthreads();
  for (int l = threadIdx.x; l < L; l += nthr) {
    const int mine = sid[l];
    int rank = 0;
    for (int j = 0; j < L; ++j) rank += (sid[j] < mine) || (sid[j] == mine && j < l);
    sord[rank] = (short)l;
  }
  __syncthreads();
  // indices outside their tables (the forward made those rows NaN) add nothing and touch no memory
  const long long ts = tsteps[b];
  const bool ts_ok = ts >= 0 && ts < n_timesteps;
  float* c = dtim + (size_t)(ts_ok ? ts : 0) * d;
  for (int i = threadIdx.x; i < d; i += nthr) {
    float tsum = 0.f, run = 0.f;
    int run_id = -1;
    for (int r0 = 0; r0 < L; r0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = r0 + u < L ? g[(size_t)sord[r0 + u] * d + i] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (r0 + u >= L) break;
        const int id = sid[sord[r0 + u]];
        if (id != run_id) {
          if (run != 0.f && run_id != pad_id && run_id >= 0 && run_id < vocab) atomicAdd(dtok + (size_t)run_id * d + i, run);
          run = 0.f;
          run_id = id;
        }
        run += v[u];
        tsum += v[u];
      }
    }
    if (run != 0.f && run_id != pad_id && run_id >= 0 && run_id < vocab) atomicAdd(dtok + (size_t)run_id * d + i, run);
    if (tsum != 0.f && ts_ok) atomicAdd(c + i, tsum);
  }
}

__global__ __launch_bounds__(256) void embed_bwd_pos_varlen_kernel(const float* __restrict__ dx, float* __restrict__ dpos,
                                                                   const int* __restrict__ cu, int B, int d) {
  // packed rows: position l of sequence b is row cu[b] + l when l < its length. One wave per sequence, four
  // sequences of a wave in flight (a loop that visits the sequences one after another waits out a memory latency
  // for each), every wave adds its own sums to the table.
  const int l = blockIdx.x;
  const int per = (B + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = d >> 2;
  for (int i0 = 0; i0 < nv; i0 += 128) {      // two float4 columns per lane and pass
    const int ia = i0 + lane, ib = i0 + 64 + lane;
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
    for (int b = b0 + wave; b < b1; b += 16) {
      f32x4 va[4], vb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int bb = b + 4 * u;
        va[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        vb[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (bb < b1) {
          const int r0 = cu[bb];
          if (l < cu[bb + 1] - r0) {
            const f32x4* row = (const f32x4*)(dx + ((size_t)r0 + l) * d);
            if (ia < nv) va[u] = row[ia];
            if (ib < nv) vb[u] = row[ib];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sa[0] += va[u][0]; sa[1] += va[u][1]; sa[2] += va[u][2]; sa[3] += va[u][3];
        sb[0] += vb[u][0]; sb[1] += vb[u][1]; sb[2] += vb[u][2]; sb[3] += vb[u][3];
      }
    }
    float* out = dpos + (size_t)l * d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (ia < nv && sa[e] != 0.f) atomicAdd(out + 4 * ia + e, sa[e]);
      if (ib < nv && sb[e] != 0.f) atomicAdd(out + 4 * ib + e, sb[e]);
    }
  }
}

__global__ __launch_bounds__(256) void embed_bwd_pos_kernel(const float* __restrict__ dx, float* __restrict__ dpos,
                                                            int B, int L, int d) {
  const int l = blockIdx.x;
  const int per = (B + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  for (int i = threadIdx.x; i < d; i += 256) {
    float sum = 0.f;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = dx[((size_t)(b + u) * L + l) * d + i];
      sum += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (; b < b1; ++b) sum += dx[((size_t)b * L + l) * d + i];
    if (sum != 0.f) atomicAdd(dpos + (size_t)l * d + i, sum);
  }
}

// =====================================================================================
// Global gradient norm and fused AdamW over flat fp32 buffers.
// =====================================================================================
__global__ __launch_bounds__(1024) void sumsq_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
  // few, large workgroups: every workgroup ends with an atomic on the one result, and atomics on one address retire
  // one after another (512 of them were the whole kernel)
  __shared__ float part[16];
  float s = 0.f;
  const long long stride = (long long)gridDim.x * 1024 * 4;
  long long i = ((long long)blockIdx.x * 1024 + threadIdx.x) * 4;
  // four 16-byte loads of a thread in flight (one per sweep of the grid)
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (; i + 3 * stride + 3 < n; i += 4 * stride) {
    const f32x4 a = *(const f32x4*)(x + i), b = *(const f32x4*)(x + i + stride);
    const f32x4 c = *(const f32x4*)(x + i + 2 * stride), e = *(const f32x4*)(x + i + 3 * stride);
    s += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
    s1 += b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
    s2 += c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
    s3 += e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3];
  }
  s = (s + s1) + (s2 + s3);
  for (; i < n; i += stride) {
    if (i + 3 < n) {
      const f32x4 v = *(const f32x4*)(x + i);
      s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    } else {
      for (long long j = i; j < n; ++j) s += x[j] * x[j];
    }
  }
  s = bd_wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += part[w];
    atomicAdd(out, t);
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n,
                                                    float lr, float beta1, float beta2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, const float* __restrict__ gsumsq,
                                                    float max_norm, const int* __restrict__ guard, double beta1d,
                                                    double beta2d) {
  float coef = 1.f;
  if (guard != nullptr) {
    // GradScaler.step semantics: non-finite gradients leave parameters, moments and the step count alone, and the
    // bias corrections follow the count of APPLIED steps, which only the device knows without a host round trip
    if (!isfinite(gsumsq[0])) return;
    const double st = (double)(guard[0] + 1);
    bc1 = (float)(1.0 - pow(beta1d, st));
    bc2_sqrt = (float)sqrt(1.0 - pow(beta2d, st));
  }
  if (gsumsq != nullptr && max_norm > 0.f) {
    const float norm = sqrtf(gsumsq[0]);
    coef = fminf(1.f, max_norm / (norm + 1e-6f));
  }
  const float step_size = lr / bc1;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
    const float vi = v[i] * beta2 + gi * gi * (1.f - beta2);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// =====================================================================================
// Muon: momentum, then the update matrix is pushed towards the nearest semi-orthogonal matrix by
// a quintic Newton-Schulz iteration in bf16 (X <- a X + (b A + c A A) X, A = X X^T), on all nb
// matrices [rows, cols] of a class at once (they sit back to back in the flat buffers).
// X is kept wide: [r, c] = [min, max] of (rows, cols), i.e. transposed when rows > cols.
// =====================================================================================
// buf <- lerp(buf, g, 1 - momentum); u = nesterov ? lerp(g, buf, momentum) : buf (torch's lerp
// formulas); X <- bf16(u) (wide layout); sumsq[b] += bf16(u)^2. g = grad * clip coefficient.
__global__ __launch_bounds__(256) void muon_momentum_kernel(const float* __restrict__ grad, float* __restrict__ buf,
                                                            unsigned short* __restrict__ X, float* __restrict__ sumsq,
                                                            const float* __restrict__ gsumsq, float max_norm,
                                                            float w_buf, float momentum, int nesterov, int rows,
                                                            int cols, int guarded) {
  __shared__ float part[4];
  if (guarded && !isfinite(gsumsq[0])) return;
  float coef = 1.f;
  if (gsumsq != nullptr && max_norm > 0.f) coef = fminf(1.f, max_norm / (sqrtf(gsumsq[0]) + 1e-6f));
  const int b = blockIdx.y;
  const long long n = (long long)rows * cols;
  const bool tr = rows > cols;
  float ss = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float g = grad[b * n + i] * coef;
    float m = buf[b * n + i];
    m = m + w_buf * (g - m);
    buf[b * n + i] = m;
    float u = m;
    if (nesterov) u = momentum < 0.5f ? g + momentum * (m - g) : m - (m - g) * (1.0f - momentum);
    const unsigned short xb = bd_f32_to_bf16(u);
    const float xf = bd_bf16_to_f32(xb);
    ss += xf * xf;
    const int r = (int)(i / cols), c = (int)(i % cols);
    X[b * n + (tr ? (long long)c * rows + r : i)] = xb;
  }
  ss = bd_wave_sum(ss);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sumsq + b, (part[0] + part[1]) + (part[2] + part[3]));
}

// X <- bf16(X / bf16(bf16(norm) + bf16(1e-7)))   (the bf16 tensor arithmetic of x / (x.norm() + eps))
__global__ __launch_bounds__(256) void muon_normalize_kernel(unsigned short* __restrict__ X, const float* __restrict__ sumsq,
                                                             long long n) {
  const int b = blockIdx.y;
  const float nrm = bd_bf16_to_f32(bd_f32_to_bf16(sqrtf(sumsq[b])));
  const float den = bd_bf16_to_f32(bd_f32_to_bf16(nrm + bd_bf16_to_f32(bd_f32_to_bf16(1e-7f))));
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    X[b * n + i] = bd_f32_to_bf16(bd_bf16_to_f32(X[b * n + i]) / den);
}

// out[b][c][r] = in[b][r][c], 64 x 64 tiles through LDS (R, C multiples of 64)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const unsigned short* __restrict__ in,
                                                             unsigned short* __restrict__ out, int R, int C) {
  __shared__ unsigned short tile[64][66];
  const long long base = (long long)blockIdx.z * R * C;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) tile[i][tx] = in[base + (long long)(r0 + i) * C + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 64; i += 4) out[base + (long long)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

// p <- p (1 - lr wd) - lr scale X   (X back in the parameter's own orientation)
__global__ __launch_bounds__(256) void muon_apply_kernel(float* __restrict__ param, const unsigned short* __restrict__ X,
                                                         float decay, float neg_step, int rows, int cols,
                                                         const float* __restrict__ guard_sumsq) {
  if (guard_sumsq != nullptr && !isfinite(guard_sumsq[0])) return;
  const int b = blockIdx.y;
  const long long n = (long long)rows * cols;
  const bool tr = rows > cols;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    const float u = bd_bf16_to_f32(X[b * n + (tr ? (long long)c * rows + r : i)]);
    float pv = param[b * n + i];
    if (decay != 1.0f) pv = pv * decay;
    param[b * n + i] = pv + neg_step * u;
  }
}

// guard[0] += the step was applied, guard[1] += it was skipped
__global__ void guard_update_kernel(int* __restrict__ guard, const float* __restrict__ gsumsq) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const bool ok = isfinite(gsumsq[0]);
    guard[0] += ok ? 1 : 0;
    guard[1] += ok ? 0 : 1;
  }
}

// Many fp32 -> bf16 copies in one launch (the bf16 weight copies and their transposes that the
// training GEMMs read, refreshed after every optimizer step): job j copies src [rows, cols] to
// dst[r * ld + c], or transposed to dst[c * ld + r]. 64 x 64 tiles through LDS; gridDim.y = jobs,
// gridDim.x = tiles of the largest job (workgroups past a job's tile count leave).
__device__ __forceinline__ int cast_job_reblock(const bd_cast_job& j, int i) { return (i / j.group) * j.group_pad + i % j.group; }

__global__ __launch_bounds__(256) void multi_cast_kernel(const bd_cast_job* __restrict__ jobs) {
  __shared__ float tile[64][65];
  const bd_cast_job j = jobs[blockIdx.y];
  const int tc = (j.cols + 63) / 64, tr = (j.rows + 63) / 64;
  if ((int)blockIdx.x >= tc * tr) return;
  const int r0 = (blockIdx.x / tc) * 64, c0 = (blockIdx.x % tc) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  unsigned short* dst = (unsigned short*)j.dst;
  const float sc = j.pad_axis ? j.scale : 1.0f;
  // destination indices of source row r / source column c (head re-blocking moves one of them)
  auto drow = [&](int r) { return j.pad_axis == 1 ? cast_job_reblock(j, r) : r; };
  auto dcol = [&](int c) { return j.pad_axis == 2 ? cast_job_reblock(j, c) : c; };
  // Four elements per thread where the job's shape allows 16-byte reads and 8-byte stores (every weight matrix of the
  // models: dimensions and head groups are multiples of 4); other jobs go element by element below.
  const bool vec = !(j.transpose & 2) && (j.cols & 3) == 0 && (j.rows & 3) == 0 && (j.ld & 3) == 0 &&
                   (j.pad_axis == 0 || ((j.group & 3) == 0 && (j.group_pad & 3) == 0)) &&
                   (((size_t)j.src & 15) == 0) && (((size_t)j.dst & 7) == 0);
  if (vec) {
    const int vx = threadIdx.x & 15, vy = threadIdx.x >> 4;
    if (!j.transpose) {
      const int c = c0 + 4 * vx;
      if (c < j.cols)
        for (int i = vy; i < 64 && r0 + i < j.rows; i += 16) {
          const f32x4 v = *(const f32x4*)(j.src + (size_t)(r0 + i) * j.cols + c);
          u32x2 o;
          o[0] = (unsigned)bd_f32_to_bf16(sc * v[0]) | ((unsigned)bd_f32_to_bf16(sc * v[1]) << 16);
          o[1] = (unsigned)bd_f32_to_bf16(sc * v[2]) | ((unsigned)bd_f32_to_bf16(sc * v[3]) << 16);
          *(u32x2*)(dst + (size_t)drow(r0 + i) * j.ld + dcol(c)) = o;
        }
      return;
    }
    for (int i = vy; i < 64; i += 16) {
      const int c = c0 + 4 * vx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r0 + i < j.rows && c < j.cols) v = *(const f32x4*)(j.src + (size_t)(r0 + i) * j.cols + c);
      tile[i][4 * vx + 0] = sc * v[0]; tile[i][4 * vx + 1] = sc * v[1];
      tile[i][4 * vx + 2] = sc * v[2]; tile[i][4 * vx + 3] = sc * v[3];
    }
    __syncthreads();
    const int r = r0 + 4 * vx;
    if (r < j.rows)
      for (int i = vy; i < 64 && c0 + i < j.cols; i += 16) {
        u32x2 o;
        o[0] = (unsigned)bd_f32_to_bf16(tile[4 * vx + 0][i]) | ((unsigned)bd_f32_to_bf16(tile[4 * vx + 1][i]) << 16);
        o[1] = (unsigned)bd_f32_to_bf16(tile[4 * vx + 2][i]) | ((unsigned)bd_f32_to_bf16(tile[4 * vx + 3][i]) << 16);
        *(u32x2*)(dst + (size_t)dcol(c0 + i) * j.ld + drow(r)) = o;
      }
    return;
  }
  if (j.transpose & 2) {        // plain fp32 copy (stacked biases)
    if (c0 + tx < j.cols)
      for (int i = ty; i < 64 && r0 + i < j.rows; i += 4)
        ((float*)j.dst)[(size_t)drow(r0 + i) * j.ld + dcol(c0 + tx)] = sc * j.src[(size_t)(r0 + i) * j.cols + c0 + tx];
    return;
  }
  if (!j.transpose) {
    if (c0 + tx < j.cols)
      for (int i = ty; i < 64 && r0 + i < j.rows; i += 4)
        dst[(size_t)drow(r0 + i) * j.ld + dcol(c0 + tx)] = bd_f32_to_bf16(sc * j.src[(size_t)(r0 + i) * j.cols + c0 + tx]);
    return;
  }
  for (int i = ty; i < 64; i += 4)
    tile[i][tx] = (r0 + i < j.rows && c0 + tx < j.cols) ? sc * j.src[(size_t)(r0 + i) * j.cols + c0 + tx] : 0.f;
  __syncthreads();
  if (r0 + tx < j.rows)
    for (int i = ty; i < 64 && c0 + i < j.cols; i += 4)
      dst[(size_t)dcol(c0 + i) * j.ld + drow(r0 + tx)] = bd_f32_to_bf16(tile[tx][i]);
}

// dst [rows, cols] += scale * src at the re-blocked index (gradients of zero-padded weight copies back to the
// parameters' own shapes); every destination element belongs to one thread of one job
__global__ __launch_bounds__(256) void multi_unpad_add_kernel(const bd_cast_job* __restrict__ jobs) {
  const bd_cast_job j = jobs[blockIdx.y];
  const int tc = (j.cols + 63) / 64, tr = (j.rows + 63) / 64;
  if ((int)blockIdx.x >= tc * tr) return;
  const int r0 = (blockIdx.x / tc) * 64, c0 = (blockIdx.x % tc) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float sc = (j.pad_axis == 0 && j.scale == 0.f) ? 1.0f : j.scale;
  float* dst = (float*)j.dst;
  const int c = c0 + tx;
  if (c >= j.cols) return;
  const int pc = j.pad_axis == 2 ? cast_job_reblock(j, c) : c;
  for (int i = ty; i < 64 && r0 + i < j.rows; i += 4) {
    const int r = r0 + i;
    const int pr = j.pad_axis == 1 ? cast_job_reblock(j, r) : r;
    dst[(size_t)r * j.cols + c] += sc * j.src[(size_t)pr * j.ld + pc];
  }
}

// x bf16 <- fp32 (flat), used to refresh the bf16 weight copies after an optimizer step
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y,
                                                        long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = bd_f32_to_bf16(x[i]);
}

}  // namespace

// ------------------------------------------------------------------------------------- C ABI
extern "C" int bd_masked_ce_fwd_parts(const float* logits, const long long* targets, const unsigned char* indicator,
                                      const long long* attention_mask, float* lse, float* sums, int n_parts, int M, int V,
                                      void* stream) {
  BD_REQUIRE(logits && targets && indicator && lse && sums, "bd_masked_ce_fwd: null operand");
  BD_REQUIRE(M >= 0 && V > 0 && n_parts >= 0 && n_parts <= 64, "bd_masked_ce_fwd: bad shape");
  if (M == 0) return BD_OK;
  bd_prof_scope prof(stream, (double)M * V * 4.0, BD_PROF_CE);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3((M + CE_ROWS - 1) / CE_ROWS), dim3(256), 0, (hipStream_t)stream, logits, targets,
                     indicator, attention_mask, lse, sums, M, V, n_parts);
  return bd_check_launch("bd_masked_ce_fwd");
}

extern "C" int bd_masked_ce_fwd(const float* logits, const long long* targets, const unsigned char* indicator,
                                const long long* attention_mask, float* lse, float* sums, int M, int V,
                                void* stream) {
  return bd_masked_ce_fwd_parts(logits, targets, indicator, attention_mask, lse, sums, 0, M, V, stream);
}

extern "C" int bd_masked_ce_finish(float* sums, float* loss, int n_parts, void* stream) {
  BD_REQUIRE(sums && loss && n_parts >= 0 && n_parts <= 64, "bd_masked_ce_finish: bad argument");
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, loss, n_parts);
  return bd_check_launch("bd_masked_ce_finish");
}

extern "C" int bd_masked_ce_bwd(const float* logits, const long long* targets, const unsigned char* indicator,
                                const long long* attention_mask, const float* lse, const float* sums,
                                const float* grad_out, float* dlogits, int M, int V, void* stream) {
  BD_REQUIRE(logits && targets && indicator && lse && sums && grad_out && dlogits, "bd_masked_ce_bwd: null operand");
  BD_REQUIRE(M >= 0 && V > 0, "bd_masked_ce_bwd: bad shape");
  if (M == 0) return BD_OK;
  bd_prof_scope prof(stream, (double)M * V * 8.0, BD_PROF_CE);
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, targets,
                     indicator, attention_mask, lse, sums, grad_out, dlogits, M, V);
  return bd_check_launch("bd_masked_ce_bwd");
}

static int layernorm_bwd_impl(const void* dy_bf16, const float* x, const float* stats, const float* gamma,
                              const float* dres, float* dx, float* dgamma, float* dbeta, int M, int d, void* dx_bf16,
                              const bd_dropout* drop, unsigned site, int n_parts, int part_stride, void* stream);

extern "C" int bd_layernorm_bwd_cast(const void* dy_bf16, const float* x, const float* stats, const float* gamma,
                                     const float* dres, float* dx, float* dgamma, float* dbeta, int M, int d,
                                     void* dx_bf16, const bd_dropout* drop, unsigned site, void* stream) {
  return layernorm_bwd_impl(dy_bf16, x, stats, gamma, dres, dx, dgamma, dbeta, M, d, dx_bf16, drop, site, 1, 0, stream);
}

extern "C" int bd_layernorm_bwd_parts(const void* dy_bf16, const float* x, const float* stats, const float* gamma,
                                      const float* dres, float* dx, float* parts, int n_parts, int M, int d,
                                      void* dx_bf16, const bd_dropout* drop, unsigned site, void* stream) {
  BD_REQUIRE(parts != nullptr && n_parts >= 1 && n_parts <= 64, "bd_layernorm_bwd_parts: bad parts (n=%d)", n_parts);
  return layernorm_bwd_impl(dy_bf16, x, stats, gamma, dres, dx, parts, parts + d, M, d, dx_bf16, drop, site, n_parts,
                            2 * d, stream);
}

extern "C" int bd_layernorm_parts_finish(const bd_ln_parts_job* jobs, int n_jobs, int n_parts, int d, void* stream) {
  BD_REQUIRE(jobs != nullptr && n_jobs > 0 && n_parts >= 1 && d > 0, "bd_layernorm_parts_finish: bad argument");
  static_assert(sizeof(bd_ln_parts_job) == sizeof(LnPartsJob), "job layout");
  hipLaunchKernelGGL(ln_parts_finish_kernel, dim3((unsigned)((2 * d + 255) / 256), (unsigned)n_jobs), dim3(256), 0, (hipStream_t)stream,
                     (const LnPartsJob*)jobs, n_parts, d);
  return bd_check_launch("bd_layernorm_parts_finish");
}

static int layernorm_bwd_impl(const void* dy_bf16, const float* x, const float* stats, const float* gamma,
                              const float* dres, float* dx, float* dgamma, float* dbeta, int M, int d, void* dx_bf16,
                              const bd_dropout* drop, unsigned site, int n_parts, int part_stride, void* stream) {
  BD_REQUIRE(dy_bf16 && x && stats && gamma && dx && dgamma && dbeta, "bd_layernorm_bwd: null operand");
  BD_REQUIRE(M > 0 && d > 0 && d % 4 == 0 && d <= 2048, "bd_layernorm_bwd: bad shape M=%d d=%d", M, d);
  bd_drop_args da;
  BD_REQUIRE(bd_drop_make(drop, site, &da), "bd_layernorm_bwd: dropout p=%g outside [0, 1)", (double)drop->p);
  BD_REQUIRE(da.thr == 0u || d % 8 == 0, "bd_layernorm_bwd: dropout needs d %% 8 == 0");
  // Rows per wave: about 320 workgroups of 8 waves, at most 16, an even number above 1. Every workgroup ends with
  // the sum over its eight waves and 2 d atomics, a fixed cost per workgroup, so more, smaller workgroups are slower
  // (10 k rows: 63 us at 1 row per wave, 30 us at 4 or 8) and fewer leave CUs idle. Neither a second launch that
  // sums per-workgroup partial results (40 us against 25) nor a last-workgroup reduction behind a device-scope fence
  // (150 us: the fence writes the XCD's L2 back) beats the atomics.
  int rpw = M / (8 * 320);
  rpw = rpw < 1 ? 1 : (rpw > 16 ? 16 : rpw);
  if (rpw > 1) rpw &= ~1;
  const int nv = (d / 4 + 63) / 64;
  if (nv >= 3) {
    // rows of 768 floats and more: the kernel's registers (and 16 d floats of LDS) leave room for ONE workgroup per CU,
    // so more workgroups than CUs run in two rounds, each with the fixed costs above: an even number of rows per wave
    // that fits the rows into one round (5120 x 1152: 27.6 us with 160 workgroups against 33.0 with 320; 2560 x 1536: 20.5
    // against 25.6)
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t prop;
      ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    static const int wide_rounds = getenv("BD_LNB_WIDE") ? atoi(getenv("BD_LNB_WIDE")) : 1;
    if (wide_rounds) {
      int need = (M + 8 * ncu - 1) / (8 * ncu);
      need = (need + 1) & ~1;
      rpw = need < 2 ? 2 : (need > 16 ? 16 : need);
    }
  }
  dim3 grid((M + 8 * rpw - 1) / (8 * rpw)), block(512);
  const size_t lds = 16 * (size_t)d * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  const unsigned short* dy = (const unsigned short*)dy_bf16;
  unsigned short* ob = (unsigned short*)dx_bf16;
  bd_prof_scope prof(stream, (double)M * d * (2.0 + 4.0 + 4.0 + 4.0 + (dx_bf16 ? 2.0 : 0.0)), BD_PROF_LN);
#define BD_LNB(NV)                                                                                              \
  do {                                                                                                          \
    BD_SET_MAX_LDS((layernorm_bwd_kernel<NV>), lds);                                                            \
    hipLaunchKernelGGL(layernorm_bwd_kernel<NV>, grid, block, lds, s, dy, x, stats, gamma, dres, dx, dgamma, dbeta, M, d, ob, \
                       da, rpw, n_parts, part_stride);                                                          \
  } while (0)
  // NV = 16-byte pieces of a row per lane: the reference's widths 768 / 1152 / 1536 are 3 / 5 (4.5) / 6 of them, and a
  // kernel sized for 8 keeps a third of its row registers empty and spills
  if (nv <= 1) BD_LNB(1);
  else if (nv <= 2) BD_LNB(2);
  else if (nv <= 3) BD_LNB(3);
  else if (nv <= 4) BD_LNB(4);
  else if (nv <= 5) BD_LNB(5);
  else if (nv <= 6) BD_LNB(6);
  else BD_LNB(8);
#undef BD_LNB
  return bd_check_launch("bd_layernorm_bwd");
}

extern "C" int bd_layernorm_bwd(const void* dy_bf16, const float* x, const float* stats, const float* gamma,
                                const float* dres, float* dx, float* dgamma, float* dbeta, int M, int d,
                                void* stream) {
  return bd_layernorm_bwd_cast(dy_bf16, x, stats, gamma, dres, dx, dgamma, dbeta, M, d, nullptr, nullptr, 0u, stream);
}

// Group count J of the LDS-DMA kernel from a small cost model fitted to measurements (microseconds):
// an XCD runs its workgroups in rounds of two per CU; a 32-token step takes 0.42 us with one
// workgroup on the CU (load latency with three steps in flight) up to 0.6 us with two; the atomic
// epilogues of all 8 J token groups drain at the L2 rate of one fp32 add per clock and channel,
// ~0.35 M per microsecond.
static int wgrad_groups(int tiles, int n_chunks, long long out_elems, int cus_per_xcd, double* t_best = nullptr) {
  int best = 1;
  double best_t = 1e30;
  const int per_xcd = (n_chunks + 7) / 8;
  for (int J = 1; J <= 16 && J <= per_xcd; ++J) {
    const int wgs = tiles * J;
    const int rounds = (wgs + 2 * cus_per_xcd - 1) / (2 * cus_per_xcd);
    double w = (double)wgs / cus_per_xcd / rounds;
    w = w < 1.0 ? 1.0 : w;
    const double step_us = 0.42 + 0.18 * (w - 1.0);
    const double t = rounds * ((per_xcd + J - 1) / J) * step_us + 8.0 * J * (double)out_elems / 0.35e6;
    if (t < best_t) { best_t = t; best = J; }
  }
  if (t_best) *t_best = best_t;
  return best;
}
// The same model for ONE workgroup per output tile that walks every token chunk (wgrad_group_kernel): no split of the
// token range, one atomic per output element. It wins where the output is large and the token range short (the wide
// models' projections at a few thousand rows: eight partial sums per element would drain longer than the products take).
static double wgrad_single_owner_us(int tiles, int n_chunks, long long out_elems, int ncu) {
  const int rounds = (tiles + 2 * ncu - 1) / (2 * ncu);
  double w = (double)tiles / ncu / rounds;
  w = w < 1.0 ? 1.0 : w;
  return rounds * (double)n_chunks * (0.42 + 0.18 * (w - 1.0)) + (double)out_elems / 0.35e6;
}

extern "C" int bd_gemm_wgrad_segments(const void* dY, int ldy, const void* X, int ldx, float* const* dW,
                                      float* const* db, int n_seg, int seg_rows, int M, int K, void* stream) {
  BD_REQUIRE(dY && X && dW, "bd_gemm_wgrad: null operand");
  BD_REQUIRE(n_seg >= 1 && n_seg <= 3, "bd_gemm_wgrad: %d segments (1..3)", n_seg);
  const int N = n_seg * seg_rows;
  BD_REQUIRE(M > 0 && seg_rows > 0 && K > 0 && seg_rows % 128 == 0 && K % 128 == 0,
             "bd_gemm_wgrad: rows=%d and K=%d must be multiples of 128", seg_rows, K);
  BD_REQUIRE(ldy >= N && ldx >= K && ldy % 8 == 0 && ldx % 8 == 0, "bd_gemm_wgrad: bad leading dimension");
  for (int i = 0; i < n_seg; ++i) BD_REQUIRE(dW[i] != nullptr, "bd_gemm_wgrad: null destination %d", i);
  hipStream_t s = (hipStream_t)stream;
  static int use_dma = getenv("BD_WGRAD_DMA") ? atoi(getenv("BD_WGRAD_DMA")) : 1;
  bd_prof_scope prof(stream, 2.0 * (double)M * N * K, BD_PROF_WGRAD);
  if (M % 32 == 0 && use_dma) {
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
      else ncu = 256;
    }
    WgradArgs a{};
    a.dY = (const unsigned short*)dY; a.X = (const unsigned short*)X;
    for (int i = 0; i < n_seg; ++i) { a.dW[i] = dW[i]; a.db[i] = db ? db[i] : nullptr; }
    a.ldy = ldy; a.ldx = ldx; a.seg_rows = seg_rows; a.N = N; a.K = K;
    a.tiles_k = K / 128; a.tiles = (N / 128) * a.tiles_k; a.n_chunks = M / 32;
    static int forced = getenv("BD_WGRAD_J") ? atoi(getenv("BD_WGRAD_J")) : 0;
    static int single = getenv("BD_WGRAD_SINGLE") ? atoi(getenv("BD_WGRAD_SINGLE")) : -1;      // 0 never, 1 always, -1 by the model
    double t_split = 0.0;
    a.J = forced > 0 ? forced : wgrad_groups(a.tiles, a.n_chunks, (long long)N * K, ncu / 8 > 0 ? ncu / 8 : 1, &t_split);
    if (single == 1 || (single < 0 && forced <= 0 && a.tiles >= ncu / 2 &&
                        wgrad_single_owner_us(a.tiles, a.n_chunks, (long long)N * K, ncu) < t_split)) {
      a.J = 1;
      hipLaunchKernelGGL(wgrad_single_kernel, dim3((unsigned)a.tiles), dim3(256), 0, s, a);
      return bd_check_launch("bd_gemm_wgrad");
    }
    const long long grid = 8ll * a.tiles * a.J;
    BD_REQUIRE(grid < (1ll << 31), "bd_gemm_wgrad: too many tiles");
    hipLaunchKernelGGL(wgrad_dma_kernel, dim3((unsigned)grid), dim3(256), 0, s, a);
    return bd_check_launch("bd_gemm_wgrad");
  }
  // ragged token count: the register-staged kernel, one segment at a time, bias sums separately
  for (int i = 0; i < n_seg; ++i) {
    const unsigned short* y = (const unsigned short*)dY + (size_t)i * seg_rows;
    const int tiles_k = K / 128, tiles = (seg_rows / 128) * tiles_k;
    int split = (1024 + tiles - 1) / tiles;
    const int max_split = (M + 511) / 512;   // at least 8 token steps per workgroup
    split = split > max_split ? max_split : split;
    split = split < 1 ? 1 : split;
    int per = ((M + split - 1) / split + 63) / 64 * 64;
    split = (M + per - 1) / per;
    const int split8 = (split + 7) / 8 * 8;   // split ids past the last one find an empty token range and leave
    hipLaunchKernelGGL(wgrad_kernel, dim3(tiles * split8), dim3(256), 0, s, y, ldy, (const unsigned short*)X, ldx, dW[i],
                       M, seg_rows, K, tiles_k, tiles, per);
    if (db && db[i])
      hipLaunchKernelGGL(colsum_kernel, dim3((seg_rows + 63) / 64, (M + CS_ROWS - 1) / CS_ROWS), dim3(256), 0, s, y, ldy,
                         db[i], M, seg_rows);
  }
  return bd_check_launch("bd_gemm_wgrad");
}

extern "C" int bd_gemm_wgrad_group(const bd_wgrad_job* jobs, int n_jobs, int M, int single_owner, void* stream) {
  BD_REQUIRE(jobs != nullptr && n_jobs >= 1 && M > 0, "bd_gemm_wgrad_group: bad argument");
  // Off unless BD_WGRAD_GROUP=1: measured on the bench's training step the single-owner launches take 1.56 ms per step
  // on their stream against 0.93 ms of the split ones and leave the data-gradient chain no faster (3.23 vs 2.81 ms
  // per step); what they buy is weight gradients that come out the same bits on every run.
  static int env_grouped = getenv("BD_WGRAD_GROUP") ? atoi(getenv("BD_WGRAD_GROUP")) : 0;
  const int grouped = single_owner < 0 ? env_grouped : single_owner;
  if (M % 32 != 0 || !grouped) {      // one launch per product, its token range split over the chip
    for (int i = 0; i < n_jobs; ++i) {
      const bd_wgrad_job& j = jobs[i];
      int rc = bd_gemm_wgrad_segments(j.dY, j.ldy, j.X, j.ldx, j.dW, j.db[0] || j.db[1] || j.db[2] ? j.db : nullptr, j.n_seg,
                                      j.seg_rows, M, j.K, stream);
      if (rc) return rc;
    }
    return BD_OK;
  }
  hipStream_t s = (hipStream_t)stream;
  for (int base = 0; base < n_jobs; base += WG_GROUP_MAX) {
    WgradGroupArgs g{};
    g.n = n_jobs - base < WG_GROUP_MAX ? n_jobs - base : WG_GROUP_MAX;
    int tiles = 0;
    double flops = 0.0;
    for (int i = 0; i < g.n; ++i) {
      const bd_wgrad_job& j = jobs[base + i];
      BD_REQUIRE(j.dY && j.X, "bd_gemm_wgrad_group: null operand in job %d", base + i);
      BD_REQUIRE(j.n_seg >= 1 && j.n_seg <= 3, "bd_gemm_wgrad_group: %d segments (1..3)", j.n_seg);
      const int N = j.n_seg * j.seg_rows;
      BD_REQUIRE(j.seg_rows > 0 && j.K > 0 && j.seg_rows % 128 == 0 && j.K % 128 == 0,
                 "bd_gemm_wgrad_group: rows=%d and K=%d must be multiples of 128", j.seg_rows, j.K);
      BD_REQUIRE(j.ldy >= N && j.ldx >= j.K && j.ldy % 8 == 0 && j.ldx % 8 == 0, "bd_gemm_wgrad_group: bad leading dimension");
      WgradArgs& a = g.a[i];
      a.dY = (const unsigned short*)j.dY; a.X = (const unsigned short*)j.X;
      for (int k = 0; k < j.n_seg; ++k) {
        BD_REQUIRE(j.dW[k] != nullptr, "bd_gemm_wgrad_group: null destination");
        a.dW[k] = j.dW[k]; a.db[k] = j.db[k];
      }
      a.ldy = j.ldy; a.ldx = j.ldx; a.seg_rows = j.seg_rows; a.N = N; a.K = j.K;
      a.tiles_k = j.K / 128; a.tiles = (N / 128) * a.tiles_k; a.n_chunks = M / 32; a.J = 1;
      g.first_tile[i] = tiles;
      tiles += a.tiles;
      flops += 2.0 * (double)M * N * j.K;
    }
    g.first_tile[g.n] = tiles;
    bd_prof_scope prof(stream, flops, BD_PROF_WGRAD);
    hipLaunchKernelGGL(wgrad_group_kernel, dim3((unsigned)tiles), dim3(256), 0, s, g);
    int rc = bd_check_launch("bd_gemm_wgrad_group");
    if (rc) return rc;
  }
  return BD_OK;
}

extern "C" int bd_gemm_wgrad_bf16(const void* dY, int ldy, const void* X, int ldx, float* dW, int M, int N, int K,
                                  void* stream) {
  BD_REQUIRE(N > 0 && N % 128 == 0, "bd_gemm_wgrad_bf16: N=%d must be a multiple of 128", N);
  return bd_gemm_wgrad_segments(dY, ldy, X, ldx, &dW, nullptr, 1, N, M, K, stream);
}

extern "C" int bd_colsum_bf16(const void* dY, int ld, float* out, int M, int N, void* stream) {
  BD_REQUIRE(dY && out && M > 0 && N > 0 && N % 8 == 0 && ld >= N && ld % 8 == 0, "bd_colsum_bf16: bad argument");
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, (M + CS_ROWS - 1) / CS_ROWS), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)dY, ld, out, M, N);
  return bd_check_launch("bd_colsum_bf16");
}

extern "C" int bd_attention_bwd(const void* qkv, const unsigned char* keymask, const void* out, const void* dout,
                                const float* lse, float* delta, void* dqkv, int B, int L, int H, int head_dim,
                                void* stream) {
  return bd_attention_bwd_drop(qkv, keymask, out, dout, lse, delta, dqkv, B, L, H, head_dim, nullptr, 0u, stream);
}

static int attention_bwd_impl(const void* qkv, const unsigned char* keymask, const void* out, const void* dout,
                              const float* lse, float* delta, void* dqkv, const int* cu, int B, int L, int H, int head_dim,
                              const bd_dropout* drop, unsigned site, void* stream) {
  BD_REQUIRE(qkv && out && dout && lse && delta && dqkv, "bd_attention_bwd: null operand");
  BD_REQUIRE(B > 0 && L > 0 && H > 0 && B <= 65535 && H <= 65535, "bd_attention_bwd: bad shape");
  bd_drop_args da;
  BD_REQUIRE(bd_drop_make(drop, site, &da), "bd_attention_bwd: dropout p=%g outside [0, 1)", (double)drop->p);
  BD_REQUIRE(da.thr == 0u || (long long)B * H * L < (1ll << 32), "bd_attention_bwd: dropout needs B*H*L < 2^32");
  bd_prof_scope prof(stream, 10.0 * (double)B * H * L * L * head_dim, BD_PROF_ATTN_BWD);   // S, dP twice; dQ, dK, dV
  const float scale = 1.0f / sqrtf((float)head_dim);
  dim3 grid((L + 63) / 64, H, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  const unsigned short* q = (const unsigned short*)qkv;
  const unsigned short* o = (const unsigned short*)out;
  const unsigned short* dO = (const unsigned short*)dout;
  unsigned short* dq = (unsigned short*)dqkv;
#define BD_LAUNCH_BWD(DH)                                                                                        \
  if (da.thr != 0u) {                                                                                            \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DH, true>), grid, block, 0, s, q, keymask, o, dO, lse, delta, dq, L, H, scale, da, cu);  \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, true>), grid, block, 0, s, q, keymask, dO, lse, delta, dq, L, H, scale, da, cu);    \
  } else {                                                                                                       \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DH, false>), grid, block, 0, s, q, keymask, o, dO, lse, delta, dq, L, H, scale, da, cu); \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, false>), grid, block, 0, s, q, keymask, dO, lse, delta, dq, L, H, scale, da, cu);   \
  }
  switch (head_dim) {
    case 32: BD_LAUNCH_BWD(32); break;
    case 64: BD_LAUNCH_BWD(64); break;
    case 128: BD_LAUNCH_BWD(128); break;
    default: return bd_fail(BD_ERR_UNSUPPORTED, "bd_attention_bwd: head_dim %d not in {32, 64, 128}", head_dim);
  }
#undef BD_LAUNCH_BWD
  return bd_check_launch("bd_attention_bwd");
}

extern "C" int bd_attention_bwd_drop(const void* qkv, const unsigned char* keymask, const void* out, const void* dout,
                                     const float* lse, float* delta, void* dqkv, int B, int L, int H, int head_dim,
                                     const bd_dropout* drop, unsigned site, void* stream) {
  return attention_bwd_impl(qkv, keymask, out, dout, lse, delta, dqkv, nullptr, B, L, H, head_dim, drop, site, stream);
}

extern "C" int bd_attention_bwd_varlen(const void* qkv, const unsigned char* keymask, const void* out, const void* dout,
                                       const float* lse, float* delta, void* dqkv, const int* cu_seqlens, int B, int L,
                                       int H, int head_dim, const bd_dropout* drop, unsigned site, void* stream) {
  BD_REQUIRE(cu_seqlens != nullptr, "bd_attention_bwd_varlen: null cu_seqlens");
  return attention_bwd_impl(qkv, keymask, out, dout, lse, delta, dqkv, cu_seqlens, B, L, H, head_dim, drop, site, stream);
}

// ---- attention backward for callers that cannot allocate (HIP graph capture, tight serving loops): the
// [B, H, L] fp32 row sums the two kernels exchange live in a file-scope device array instead of a caller
// buffer. Each GPU holds its own copy of the array at its own address, looked up once per device.
constexpr long long BD_STATIC_DELTA_FLOATS = 1ll << 22;      // 16 MiB per GPU, held until the process ends
static __device__ __align__(256) float bd_static_delta[BD_STATIC_DELTA_FLOATS];
constexpr int BD_STATIC_MAX_DEVICES = 64;
static float* g_static_delta_ptr[BD_STATIC_MAX_DEVICES] = {};
static std::mutex g_static_delta_mu;

// The scratch array's address on the current device: looked up once per device (under the mutex), afterwards one
// atomic load. Null on failure (bd_last_error says why).
static float* static_delta_of_current_device(const char* who) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BD_STATIC_MAX_DEVICES) {
    bd_fail(BD_ERR_ARG, "%s: no current device (or device index >= %d)", who, BD_STATIC_MAX_DEVICES);
    return nullptr;
  }
  float* delta = __atomic_load_n(&g_static_delta_ptr[dev], __ATOMIC_ACQUIRE);
  if (delta != nullptr) return delta;
  std::lock_guard<std::mutex> lk(g_static_delta_mu);
  if (g_static_delta_ptr[dev] == nullptr) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(bd_static_delta));
    if (e != hipSuccess || p == nullptr) {
      bd_fail(BD_ERR_LAUNCH, "%s: hipGetSymbolAddress failed: %s", who, hipGetErrorString(e));
      return nullptr;
    }
    __atomic_store_n(&g_static_delta_ptr[dev], (float*)p, __ATOMIC_RELEASE);
  }
  return g_static_delta_ptr[dev];
}

// Also the set-up call: it resolves the scratch address for the current device, so that a later
// bd_attention_bwd_static on that device (under stream capture, say) makes no runtime lookup.
extern "C" long long bd_attention_bwd_static_capacity(void) {
  (void)static_delta_of_current_device("bd_attention_bwd_static_capacity");
  return BD_STATIC_DELTA_FLOATS;
}

extern "C" int bd_attention_bwd_static(const void* qkv, const unsigned char* keymask, const void* out, const void* dout,
                                       const float* lse, void* dqkv, int B, int L, int H, int head_dim, void* stream) {
  BD_REQUIRE(B > 0 && L > 0 && H > 0, "bd_attention_bwd_static: bad shape");
  BD_REQUIRE((long long)B * H * L <= BD_STATIC_DELTA_FLOATS,
             "bd_attention_bwd_static: B*H*L = %lld exceeds the built-in scratch of %lld entries (use bd_attention_bwd "
             "with a caller buffer)", (long long)B * H * L, BD_STATIC_DELTA_FLOATS);
  float* const delta = static_delta_of_current_device("bd_attention_bwd_static");
  if (delta == nullptr) return BD_ERR_LAUNCH;
  // every entry a kernel reads (queries 0..L-1 of each head) is written by the dq kernel of this call first:
  // nothing of an earlier call's contents is consumed, so nothing has to be cleared
  return bd_attention_bwd_drop(qkv, keymask, out, dout, lse, delta, dqkv, B, L, H, head_dim, nullptr, 0u, stream);
}

static int embed_bwd_impl(const float* dx, const long long* ids, const long long* timesteps, float* dtok, float* dpos,
                          float* dtime, const int* cu, int B, int L, int d, int pad_id, int vocab, int max_pos,
                          int n_timesteps, void* stream);

extern "C" int bd_embed_bwd(const float* dx, const long long* ids, const long long* timesteps, float* dtok,
                            float* dpos, float* dtime, int B, int L, int d, int pad_id, int vocab, int max_pos,
                            int n_timesteps, void* stream) {
  return embed_bwd_impl(dx, ids, timesteps, dtok, dpos, dtime, nullptr, B, L, d, pad_id, vocab, max_pos, n_timesteps, stream);
}

extern "C" int bd_embed_bwd_varlen(const float* dx, const long long* ids, const long long* timesteps, float* dtok,
                                   float* dpos, float* dtime, const int* cu_seqlens, int B, int L, int d, int pad_id,
                                   int vocab, int max_pos, int n_timesteps, void* stream) {
  BD_REQUIRE(cu_seqlens != nullptr, "bd_embed_bwd_varlen: null cu_seqlens");
  return embed_bwd_impl(dx, ids, timesteps, dtok, dpos, dtime, cu_seqlens, B, L, d, pad_id, vocab, max_pos, n_timesteps,
                        stream);
}

static int embed_bwd_impl(const float* dx, const long long* ids, const long long* timesteps, float* dtok, float* dpos,
                          float* dtime, const int* cu, int B, int L, int d, int pad_id, int vocab, int max_pos,
                          int n_timesteps, void* stream) {
  BD_REQUIRE(dx && ids && timesteps && dtok && dpos && dtime, "bd_embed_bwd: null operand");
  BD_REQUIRE(vocab > 0 && n_timesteps > 0, "bd_embed_bwd: empty table");
  BD_REQUIRE(L <= max_pos, "bd_embed_bwd: sequence length %d exceeds max_position_embeddings %d", L, max_pos);
  BD_REQUIRE(B > 0 && L > 0 && L <= EB_MAXL && d > 0, "bd_embed_bwd: bad shape B=%d L=%d d=%d", B, L, d);
  bd_prof_scope prof(stream, 2.0 * (double)B * L * d * 4.0, BD_PROF_EMBED);
  // one thread per column: the longest sequence's row loop, not the column loop, sets the kernel's length
  const int seq_threads = d >= 1024 ? 1024 : (d + 63) / 64 * 64;
  hipLaunchKernelGGL(embed_bwd_seq_kernel, dim3((unsigned)B), dim3(seq_threads), 0, (hipStream_t)stream, dx, ids, timesteps,
                     dtok, dtime, L, d, pad_id, vocab, n_timesteps, cu);
  // enough workgroups to fill the chip: the batch is cut into slices when there are few positions
  const int slices = L >= 1024 ? 1 : (1024 + L - 1) / L;
  if (cu)
    hipLaunchKernelGGL(embed_bwd_pos_varlen_kernel, dim3((unsigned)L, (unsigned)(slices < B ? slices : B)), dim3(256), 0,
                       (hipStream_t)stream, dx, dpos, cu, B, d);
  else
    hipLaunchKernelGGL(embed_bwd_pos_kernel, dim3((unsigned)L, (unsigned)(slices < B ? slices : B)), dim3(256), 0,
                       (hipStream_t)stream, dx, dpos, B, L, d);
  return bd_check_launch("bd_embed_bwd");
}

extern "C" int bd_sumsq_f32(const float* x, long long n, float* out, void* stream) {
  BD_REQUIRE(x && out && n >= 0, "bd_sumsq_f32: bad argument");
  bd_prof_scope prof(stream, (double)n * 4.0, BD_PROF_OPT);
  if (n == 0) return BD_OK;
  long long blocks = (n + 16383) / 16384;
  blocks = blocks > 128 ? 128 : blocks;   // one same-address atomic per workgroup: keep them few (43 MB: 19 us with 512
                                          // workgroups of 256 threads, 8.7 us with 128 of 1024)
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, x, n, out);
  return bd_check_launch("bd_sumsq_f32");
}

extern "C" int bd_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                             float lr, float beta1, float beta2, float eps, float weight_decay,
                             float bias_correction1, float bias_correction2_sqrt, const float* grad_sumsq,
                             float max_norm, void* stream) {
  BD_REQUIRE(params && grads && exp_avg && exp_avg_sq && n >= 0, "bd_adamw_step: bad argument");
  BD_REQUIRE(bias_correction1 > 0.f && bias_correction2_sqrt > 0.f, "bd_adamw_step: bias corrections must be > 0");
  bd_prof_scope prof(stream, (double)n * 28.0, BD_PROF_OPT);
  if (n == 0) return BD_OK;
  long long blocks = (n + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grads,
                     exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bias_correction1,
                     bias_correction2_sqrt, grad_sumsq, max_norm, (const int*)nullptr, 0.0, 0.0);
  return bd_check_launch("bd_adamw_step");
}

extern "C" int bd_adamw_step_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                                     float lr, double beta1, double beta2, float eps, float weight_decay,
                                     const int* guard, const float* grad_sumsq, float max_norm, void* stream) {
  BD_REQUIRE(params && grads && exp_avg && exp_avg_sq && n >= 0, "bd_adamw_step_guarded: bad argument");
  BD_REQUIRE(guard && grad_sumsq, "bd_adamw_step_guarded: needs the guard counters and the gradient sum of squares");
  BD_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "bd_adamw_step_guarded: betas outside [0, 1)");
  bd_prof_scope prof(stream, (double)n * 28.0, BD_PROF_OPT);
  if (n == 0) return BD_OK;
  long long blocks = (n + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grads,
                     exp_avg, exp_avg_sq, n, lr, (float)beta1, (float)beta2, eps, weight_decay, 1.f, 1.f, grad_sumsq,
                     max_norm, guard, beta1, beta2);
  return bd_check_launch("bd_adamw_step_guarded");
}

extern "C" int bd_step_guard_update(int* guard, const float* grad_sumsq, void* stream) {
  BD_REQUIRE(guard && grad_sumsq, "bd_step_guard_update: null operand");
  hipLaunchKernelGGL(guard_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, guard, grad_sumsq);
  return bd_check_launch("bd_step_guard_update");
}

extern "C" int bd_cast_f32_to_bf16(const float* x, void* y, long long n, void* stream) {
  BD_REQUIRE(x && y && n >= 0, "bd_cast_f32_to_bf16: bad argument");
  if (n == 0) return BD_OK;
  long long blocks = (n + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                     (unsigned short*)y, n);
  return bd_check_launch("bd_cast_f32_to_bf16");
}

// ------------------------------------------------------------------------------------- Muon
static long long muon_align(long long x) { return (x + 255) / 256 * 256; }

extern "C" long long bd_muon_workspace_bytes(int nb, int rows, int cols) {
  const long long r = rows < cols ? rows : cols, c = rows < cols ? cols : rows;
  return 2 * muon_align(2ll * nb * r * c) + 2 * muon_align(2ll * nb * r * r) + muon_align(4ll * nb);
}

static int muon_step_impl(float* params, const float* grads, float* momentum_buf, void* workspace,
                          long long workspace_bytes, int nb, int rows, int cols, float lr, float weight_decay,
                          float momentum, int nesterov, int ns_steps, const float* grad_sumsq, float max_norm,
                          int guarded, void* stream);

extern "C" int bd_muon_step(float* params, const float* grads, float* momentum_buf, void* workspace,
                            long long workspace_bytes, int nb, int rows, int cols, float lr, float weight_decay,
                            float momentum, int nesterov, int ns_steps, const float* grad_sumsq, float max_norm,
                            void* stream) {
  return muon_step_impl(params, grads, momentum_buf, workspace, workspace_bytes, nb, rows, cols, lr, weight_decay,
                        momentum, nesterov, ns_steps, grad_sumsq, max_norm, 0, stream);
}

extern "C" int bd_muon_step_guarded(float* params, const float* grads, float* momentum_buf, void* workspace,
                                    long long workspace_bytes, int nb, int rows, int cols, float lr,
                                    float weight_decay, float momentum, int nesterov, int ns_steps,
                                    const float* grad_sumsq, float max_norm, void* stream) {
  BD_REQUIRE(grad_sumsq != nullptr, "bd_muon_step_guarded: needs the gradient sum of squares");
  return muon_step_impl(params, grads, momentum_buf, workspace, workspace_bytes, nb, rows, cols, lr, weight_decay,
                        momentum, nesterov, ns_steps, grad_sumsq, max_norm, 1, stream);
}

static int muon_step_impl(float* params, const float* grads, float* momentum_buf, void* workspace,
                          long long workspace_bytes, int nb, int rows, int cols, float lr, float weight_decay,
                          float momentum, int nesterov, int ns_steps, const float* grad_sumsq, float max_norm,
                          int guarded, void* stream) {
  BD_REQUIRE(params && grads && momentum_buf && workspace, "bd_muon_step: null operand");
  BD_REQUIRE(nb > 0 && nb <= 65535 && rows > 0 && cols > 0 && rows % 128 == 0 && cols % 128 == 0,
             "bd_muon_step: %d matrices of %d x %d (sizes must be multiples of 128)", nb, rows, cols);
  BD_REQUIRE(workspace_bytes >= bd_muon_workspace_bytes(nb, rows, cols), "bd_muon_step: workspace too small");
  BD_REQUIRE(momentum >= 0.f && momentum <= 1.f, "bd_muon_step: momentum=%g", (double)momentum);
  bd_prof_scope prof(stream, (double)nb * rows * cols * 28.0, BD_PROF_OPT);
  hipStream_t s = (hipStream_t)stream;
  const int r = rows < cols ? rows : cols, c = rows < cols ? cols : rows;
  const long long n = (long long)rows * cols;
  unsigned char* w = (unsigned char*)workspace;
  unsigned short* X = (unsigned short*)w;  w += muon_align(2ll * nb * n);
  unsigned short* Z = (unsigned short*)w;  w += muon_align(2ll * nb * n);
  unsigned short* A = (unsigned short*)w;  w += muon_align(2ll * nb * r * r);
  unsigned short* P = (unsigned short*)w;  w += muon_align(2ll * nb * r * r);
  float* sumsq = (float*)w;
  if (hipMemsetAsync(sumsq, 0, 4ll * nb, s) != hipSuccess) return bd_fail(BD_ERR_LAUNCH, "bd_muon_step: memset failed");
  const unsigned gx = (unsigned)((n + 1023) / 1024 < 1024 ? (n + 1023) / 1024 : 1024);
  hipLaunchKernelGGL(muon_momentum_kernel, dim3(gx, nb), dim3(256), 0, s, grads, momentum_buf, X, sumsq, grad_sumsq, max_norm,
                     (float)(1.0 - (double)momentum), momentum, nesterov, rows, cols, guarded);
  hipLaunchKernelGGL(muon_normalize_kernel, dim3(gx, nb), dim3(256), 0, s, X, sumsq, n);
  const int steps = ns_steps < 1 ? 1 : ns_steps;
  for (int it = 0; it < steps; ++it) {
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(c / 64, r / 64, nb), dim3(256), 0, s, X, Z, r, c);
    // A = X X^T;  P = b A + c A A;  X = a X + P X  (P X = P Z^T with Z = X^T)
    int rc = bd_gemm_bf16_batched(X, X, A, nullptr, nb, n, n, (long long)r * r, r, r, c, 1.f, 0.f, stream);
    if (rc) return rc;
    rc = bd_gemm_bf16_batched(A, A, P, A, nb, (long long)r * r, (long long)r * r, (long long)r * r, r, r, r, 2.0315f, -4.7750f,
                              stream);
    if (rc) return rc;
    rc = bd_gemm_bf16_batched(P, Z, X, X, nb, (long long)r * r, n, n, r, c, r, 1.f, 3.4445f, stream);
    if (rc) return rc;
  }
  const float scale = rows > cols ? (float)rows / (float)cols : 1.0f;   // max(1, rows / cols)
  const float decay = weight_decay != 0.f ? (float)(1.0 - (double)lr * (double)weight_decay) : 1.0f;
  hipLaunchKernelGGL(muon_apply_kernel, dim3(gx, nb), dim3(256), 0, s, params, X, decay, (float)(-(double)lr * (double)scale),
                     rows, cols, guarded ? grad_sumsq : (const float*)nullptr);
  return bd_check_launch("bd_muon_step");
}

extern "C" int bd_multi_unpad_add_f32(const bd_cast_job* jobs_device, int n_jobs, int max_rows, int max_cols, void* stream) {
  BD_REQUIRE(jobs_device && n_jobs > 0 && n_jobs <= 65535 && max_rows > 0 && max_cols > 0, "bd_multi_unpad_add_f32: bad argument");
  const long long tiles = (long long)((max_rows + 63) / 64) * ((max_cols + 63) / 64);
  BD_REQUIRE(tiles < (1ll << 31), "bd_multi_unpad_add_f32: job too large");
  bd_prof_scope prof(stream, 0.0, BD_PROF_OPT);
  hipLaunchKernelGGL(multi_unpad_add_kernel, dim3((unsigned)tiles, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_device);
  return bd_check_launch("bd_multi_unpad_add_f32");
}

extern "C" int bd_multi_cast_bf16(const bd_cast_job* jobs_device, int n_jobs, int max_rows, int max_cols, void* stream) {
  BD_REQUIRE(jobs_device && n_jobs > 0 && n_jobs <= 65535 && max_rows > 0 && max_cols > 0, "bd_multi_cast_bf16: bad argument");
  const long long tiles = (long long)((max_rows + 63) / 64) * ((max_cols + 63) / 64);
  BD_REQUIRE(tiles < (1ll << 31), "bd_multi_cast_bf16: job too large");
  bd_prof_scope prof(stream, 0.0, BD_PROF_OPT);
  hipLaunchKernelGGL(multi_cast_kernel, dim3((unsigned)tiles, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_device);
  return bd_check_launch("bd_multi_cast_bf16");
}
