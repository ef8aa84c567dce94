// Bandwidth-bound kernels of the denoiser forward and the q(x_t|x_0) corruption step.
#include "bd_common.h"
#include "bd_rng.h"
#include "bd_drop.h"
#include "bd_prof.h"

namespace {

// ---------------------------------------------------------------------------------
// Embedding sum: x[m, :] = tok[ids[m]] + pos[m % L] + time[t[m / L]]   (fp32)
// One wave per token row, float4 per lane.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const long long* __restrict__ ids,
                                                    const long long* __restrict__ tsteps,
                                                    const float* __restrict__ tok,
                                                    const float* __restrict__ pos,
                                                    const float* __restrict__ tim,
                                                    float* __restrict__ x, int M, int L, int d,
                                                    int vocab, int n_timesteps, bd_drop_args drop,
                                                    const int* __restrict__ rowmap) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  // packed rows: row holds logical token rowmap[row] = b * L + l of the [B, L] batch, or nothing (< 0: zeros)
  const int lrow = rowmap ? rowmap[row] : row;
  if (lrow < 0) {
    f32x4* o = (f32x4*)(x + (size_t)row * d);
    for (int i = lane; i < d / 4; i += 64) { f32x4 r; r[0] = 0.f; r[1] = 0.f; r[2] = 0.f; r[3] = 0.f; o[i] = r; }
    return;
  }
  const long long id = ids[lrow];
  const long long t = tsteps[lrow / L];
  if (id < 0 || id >= vocab || t < 0 || t >= n_timesteps) {
    // an index outside its table (nn.Embedding raises there): never read past the table, and make the
    // row NaN so that the logits and the loss of this batch are visibly invalid
    f32x4* o = (f32x4*)(x + (size_t)row * d);
    const float q = __int_as_float(0x7fc00000);
    for (int i = lane; i < d / 4; i += 64) { f32x4 r; r[0] = q; r[1] = q; r[2] = q; r[3] = q; o[i] = r; }
    return;
  }
  const f32x4* a = (const f32x4*)(tok + (size_t)id * d);
  const f32x4* b = (const f32x4*)(pos + (size_t)(lrow % L) * d);
  const f32x4* c = (const f32x4*)(tim + (size_t)t * d);
  f32x4* o = (f32x4*)(x + (size_t)row * d);
  for (int i = lane; i < d / 4; i += 64) {
    const f32x4 va = a[i], vb = b[i], vc = c[i];
    f32x4 r;
    r[0] = (va[0] + vb[0]) + vc[0]; r[1] = (va[1] + vb[1]) + vc[1];
    r[2] = (va[2] + vb[2]) + vc[2]; r[3] = (va[3] + vb[3]) + vc[3];
    if (drop.thr != 0u) {
      const bd_philox_out w = bd_drop_block(drop, (uint32_t)row, (uint32_t)(i >> 1));
      const int j0 = 4 * (i & 1);
      r[0] *= bd_drop_factor(drop, w, j0);     r[1] *= bd_drop_factor(drop, w, j0 + 1);
      r[2] *= bd_drop_factor(drop, w, j0 + 2); r[3] *= bd_drop_factor(drop, w, j0 + 3);
    }
    o[i] = r;
  }
}

// ---------------------------------------------------------------------------------
// Pooling of the final hidden states over the sequence: [B, L, d] bf16 -> [B, d] fp32.
//   mean: sum_l h * m / max(sum_l m, 1e-9)   (plain mean without a mask)
//   max : max_l (h * m + (1 - m) * -1e9)     (plain max without a mask)
//   cls : h[:, 0]
// m = float(attention_mask). A workgroup owns (sequence, 512 columns): lane = 8 columns
// (one 16-byte load per row), the four waves take rows 4 apart and meet in LDS.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_kernel(const unsigned short* __restrict__ h,
                                                   const long long* __restrict__ mask, float* __restrict__ out,
                                                   int L, int d, int mode) {
  __shared__ float red[4][512];
  __shared__ float msum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x, col = blockIdx.y * 512 + lane * 8;
  const bool on = col < d;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = mode == 2 ? -INFINITY : 0.f;
  float ms = 0.f;
  const int l_end = mode == 1 ? 1 : L;
  for (int l = wave; l < l_end; l += 4) {
    const float m = mask ? (float)mask[(size_t)b * L + l] : 1.0f;
    ms += m;
    if (!on) continue;
    const u32x4 w = *(const u32x4*)(h + ((size_t)b * L + l) * d + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
      if (mode == 2) {
        const float fill = (1.0f - m) * -1e9f;
        acc[2 * e] = fmaxf(acc[2 * e], mask ? lo * m + fill : lo);
        acc[2 * e + 1] = fmaxf(acc[2 * e + 1], mask ? hi * m + fill : hi);
      } else {
        acc[2 * e] += lo * m;
        acc[2 * e + 1] += hi * m;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = acc[e];
  if (lane == 0) msum[wave] = ms;
  __syncthreads();
  // thread t finishes columns t and t + 256 of the slab
  for (int cc = threadIdx.x; cc < 512; cc += 256) {
    const int c2 = blockIdx.y * 512 + cc;
    if (c2 >= d) continue;
    float v;
    if (mode == 2) v = fmaxf(fmaxf(red[0][cc], red[1][cc]), fmaxf(red[2][cc], red[3][cc]));
    else v = (red[0][cc] + red[1][cc]) + (red[2][cc] + red[3][cc]);
    if (mode == 0) {
      const float cnt = mask ? (msum[0] + msum[1]) + (msum[2] + msum[3]) : (float)L;
      v = v / fmaxf(cnt, 1e-9f);
    }
    out[(size_t)b * d + c2] = v;
  }
}

// ---------------------------------------------------------------------------------
// Batch assembly from a token store resident in HBM: sequence i of the batch is store row
// indices[i], tokens[offsets[row] .. offsets[row + 1]) (uint16, BOS/EOS already in place);
// ids_out[i, :] = those tokens then pad, attention_mask 1 on the tokens, int64 as the model's
// interface takes them. One wave per sequence.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void collate_kernel(const unsigned short* __restrict__ tokens,
                                                      const long long* __restrict__ offsets,
                                                      const long long* __restrict__ indices, long long n_rows,
                                                      int B, int L, int pad_id, long long* __restrict__ ids_out,
                                                      long long* __restrict__ mask_out, int* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const long long row = indices[i];
  long long start = 0, len = 0;
  if (row < 0 || row >= n_rows) {
    if (lane == 0) atomicOr(err, 1);
  } else {
    start = offsets[row];
    len = offsets[row + 1] - start;
    if (len > L) {
      if (lane == 0) atomicOr(err, 2);
      len = L;
    }
  }
  for (int l = lane; l < L; l += 64) {
    const bool in = l < len;
    ids_out[(size_t)i * L + l] = in ? (long long)tokens[start + l] : (long long)pad_id;
    mask_out[(size_t)i * L + l] = in ? 1 : 0;
  }
}

// out[m, n] = x[m, n] * keep / (1 - p), fp32 or bf16 result; one thread per four columns
template <bool BF16_OUT>
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, void* __restrict__ out,
                                                      long long n4, int N4, bd_drop_args drop) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const uint32_t row = (uint32_t)(i / N4), c4 = (uint32_t)(i % N4);
    f32x4 v = ((const f32x4*)x)[i];
    const bd_philox_out w = bd_drop_block(drop, row, c4 >> 1);
    const int j0 = 4 * (int)(c4 & 1u);
    v[0] *= bd_drop_factor(drop, w, j0);     v[1] *= bd_drop_factor(drop, w, j0 + 1);
    v[2] *= bd_drop_factor(drop, w, j0 + 2); v[3] *= bd_drop_factor(drop, w, j0 + 3);
    if (BF16_OUT) {
      u32x2 o;
      o[0] = bd_pack_bf16x2(v[0], v[1]); o[1] = bd_pack_bf16x2(v[2], v[3]);
      ((u32x2*)out)[i] = o;
    } else {
      ((f32x4*)out)[i] = v;
    }
  }
}

// ---------------------------------------------------------------------------------
// LayerNorm (eps inside the sqrt, affine): fp32 in, bf16 and/or fp32 out.
// One wave per row; the row lives in registers (d <= 64*4*NV).
// ---------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        unsigned short* __restrict__ out_bf16,
                                                        float* __restrict__ out_f32,
                                                        float* __restrict__ stats, int M, int d,
                                                        float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nv = d / 4;
  const f32x4* xr = (const f32x4*)(x + (size_t)row * d);
  f32x4 v[NV];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = lane + 64 * j;
    v[j] = i < nv ? xr[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  }
  const float mean = bd_wave_sum(sum) / (float)d;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = lane + 64 * j;
    if (i < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float dlt = v[j][e] - mean; sq += dlt * dlt; }
    }
  }
  const float var = bd_wave_sum(sq) / (float)d;
  const float rstd = rsqrtf(var + eps);
  if (stats != nullptr && lane == 0) { stats[2 * (size_t)row] = mean; stats[2 * (size_t)row + 1] = rstd; }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = lane + 64 * j;
    if (i < nv) {
      const f32x4 gm = ((const f32x4*)gamma)[i], bt = ((const f32x4*)beta)[i];
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = (v[j][e] - mean) * rstd * gm[e] + bt[e];
      if (out_bf16 != nullptr) {
        u32x2 w;
        w[0] = bd_pack_bf16x2(y[0], y[1]);
        w[1] = bd_pack_bf16x2(y[2], y[3]);
        *(u32x2*)(out_bf16 + (size_t)row * d + 4 * i) = w;
      }
      if (out_f32 != nullptr) ((f32x4*)(out_f32 + (size_t)row * d))[i] = y;
    }
  }
}

// ---------------------------------------------------------------------------------
// q(x_t | x_0): per-token Bernoulli masking. One thread per token, coalesced.
// uniforms == null -> Philox stream keyed by (seed, global sequence index, t, position).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void corrupt_kernel(const long long* __restrict__ ids,
                                                      const long long* __restrict__ attn,
                                                      const long long* __restrict__ tsteps,
                                                      const float* __restrict__ schedule,
                                                      const float* __restrict__ uniforms,
                                                      long long* __restrict__ noisy,
                                                      unsigned char* __restrict__ indicator,
                                                      long long n, int L, int mask_id, int pad_id,
                                                      int bos_id, int eos_id, int has_bos, int has_eos,
                                                      unsigned long long seed, long long seq_offset) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long b = i / L;
  const int l = (int)(i - b * L);
  const long long id = ids[i];
  const long long t = tsteps[b];
  const float p = schedule[t];
  const float u = uniforms ? uniforms[i]
                           : bd_uniform_corrupt(seed, (uint32_t)(b + seq_offset), (uint32_t)t, (uint32_t)l);
  bool m = u < p;
  m = m && (attn ? (attn[i] == 1) : (id != pad_id));
  if (has_bos) m = m && (id != bos_id);
  if (has_eos) m = m && (id != eos_id);
  noisy[i] = m ? (long long)mask_id : id;
  indicator[i] = m ? 1 : 0;
}

// The same for L % 4 == 0: a thread owns four neighbouring positions of one sequence (one Philox
// block, 32-byte loads and stores of the int64 ids, one 4-byte store of the indicator).
typedef __attribute__((ext_vector_type(2))) long long i64x2;
__global__ __launch_bounds__(256) void corrupt4_kernel(const long long* __restrict__ ids,
                                                       const long long* __restrict__ attn,
                                                       const long long* __restrict__ tsteps,
                                                       const float* __restrict__ schedule,
                                                       const float* __restrict__ uniforms,
                                                       long long* __restrict__ noisy,
                                                       unsigned char* __restrict__ indicator,
                                                       long long n4, int L4, int mask_id, int pad_id,
                                                       int bos_id, int eos_id, int has_bos, int has_eos,
                                                       unsigned long long seed, long long seq_offset) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= n4) return;
  const long long b = g / L4;
  const int l4 = (int)(g - b * L4);
  const long long t = tsteps[b];
  const float p = schedule[t];
  long long id[4], am[4];
  *(i64x2*)(id) = ((const i64x2*)ids)[2 * g];
  *(i64x2*)(id + 2) = ((const i64x2*)ids)[2 * g + 1];
  if (attn) {
    *(i64x2*)(am) = ((const i64x2*)attn)[2 * g];
    *(i64x2*)(am + 2) = ((const i64x2*)attn)[2 * g + 1];
  }
  float u[4];
  if (uniforms) {
    const f32x4 uv = ((const f32x4*)uniforms)[g];
    u[0] = uv[0]; u[1] = uv[1]; u[2] = uv[2]; u[3] = uv[3];
  } else {
    const bd_philox_out w = bd_corrupt_block(seed, (uint32_t)(b + seq_offset), (uint32_t)t, (uint32_t)l4);
#pragma unroll
    for (int e = 0; e < 4; ++e) u[e] = bd_u01(w.v[e]);
  }
  unsigned int flags = 0u;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bool m = u[e] < p;
    m = m && (attn ? (am[e] == 1) : (id[e] != pad_id));
    if (has_bos) m = m && (id[e] != bos_id);
    if (has_eos) m = m && (id[e] != eos_id);
    if (m) { id[e] = (long long)mask_id; flags |= 1u << (8 * e); }
  }
  ((i64x2*)noisy)[2 * g] = *(i64x2*)(id);
  ((i64x2*)noisy)[2 * g + 1] = *(i64x2*)(id + 2);
  ((unsigned int*)indicator)[g] = flags;
}

// byte mask from an int64 attention mask (non-zero -> 1)
__global__ __launch_bounds__(256) void mask_to_bytes_kernel(const long long* __restrict__ attn,
                                                            unsigned char* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = attn[i] != 0 ? 1 : 0;
}

}  // namespace

extern "C" int bd_embed(const long long* ids, const long long* timesteps, const float* tok_emb,
                        const float* pos_emb, const float* time_emb, float* x, int B, int L, int d, int vocab,
                        int max_pos, int n_timesteps, void* stream) {
  return bd_embed_drop(ids, timesteps, tok_emb, pos_emb, time_emb, x, B, L, d, vocab, max_pos, n_timesteps, nullptr,
                       0u, stream);
}

extern "C" int bd_pool_hidden(const void* hidden_bf16, const long long* attention_mask, float* out, int B, int L,
                              int d, int mode, void* stream) {
  BD_REQUIRE(hidden_bf16 && out, "bd_pool_hidden: null operand");
  BD_REQUIRE(B > 0 && L > 0 && d > 0 && d % 8 == 0 && B <= 2147483647, "bd_pool_hidden: bad shape B=%d L=%d d=%d", B, L, d);
  BD_REQUIRE(mode >= 0 && mode <= 2, "bd_pool_hidden: unknown pooling mode %d (0 mean, 1 cls, 2 max)", mode);
  hipLaunchKernelGGL(pool_kernel, dim3(B, (d + 511) / 512), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)hidden_bf16, mode == 1 ? nullptr : attention_mask, out, L, d, mode);
  return bd_check_launch("bd_pool_hidden");
}

extern "C" int bd_collate_batch(const void* tokens_u16, const long long* offsets, const long long* indices,
                                long long n_rows, int B, int L, int pad_id, long long* ids_out,
                                long long* attention_mask_out, int* error_flag, void* stream) {
  BD_REQUIRE(tokens_u16 && offsets && indices && ids_out && attention_mask_out && error_flag, "bd_collate_batch: null operand");
  BD_REQUIRE(B > 0 && L > 0 && n_rows > 0, "bd_collate_batch: bad shape B=%d L=%d rows=%lld", B, L, n_rows);
  hipLaunchKernelGGL(collate_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)tokens_u16,
                     offsets, indices, n_rows, B, L, pad_id, ids_out, attention_mask_out, error_flag);
  return bd_check_launch("bd_collate_batch");
}

static int dropout_launch(const float* x, void* out, int M, int N, const bd_dropout* drop, unsigned site, bool bf16,
                          void* stream) {
  BD_REQUIRE(x && out, "bd_dropout: null operand");
  BD_REQUIRE(M > 0 && N > 0 && N % 4 == 0, "bd_dropout: bad shape M=%d N=%d", M, N);
  bd_drop_args da;
  BD_REQUIRE(bd_drop_make(drop, site, &da), "bd_dropout: p=%g outside [0, 1)", (double)drop->p);
  bd_prof_scope prof(stream, (double)M * N * (bf16 ? 6.0 : 8.0), BD_PROF_EMBED);
  const long long n4 = (long long)M * (N / 4);
  const unsigned grid = (unsigned)((n4 + 255) / 256 < 65536 ? (n4 + 255) / 256 : 65536);
  if (bf16) hipLaunchKernelGGL(dropout_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, n4, N / 4, da);
  else hipLaunchKernelGGL(dropout_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, n4, N / 4, da);
  return bd_check_launch("bd_dropout");
}

extern "C" int bd_dropout_f32(const float* x, float* out, int M, int N, const bd_dropout* drop, unsigned site,
                              void* stream) {
  return dropout_launch(x, out, M, N, drop, site, false, stream);
}

extern "C" int bd_dropout_f32_to_bf16(const float* x, void* out, int M, int N, const bd_dropout* drop, unsigned site,
                                      void* stream) {
  return dropout_launch(x, out, M, N, drop, site, true, stream);
}

extern "C" int bd_embed_drop(const long long* ids, const long long* timesteps, const float* tok_emb,
                             const float* pos_emb, const float* time_emb, float* x, int B, int L, int d,
                             int vocab, int max_pos, int n_timesteps, const bd_dropout* drop, unsigned site,
                             void* stream) {
  BD_REQUIRE(ids && timesteps && tok_emb && pos_emb && time_emb && x, "bd_embed: null operand");
  BD_REQUIRE(vocab > 0 && n_timesteps > 0, "bd_embed: empty table (vocab=%d, timesteps=%d)", vocab, n_timesteps);
  BD_REQUIRE(L <= max_pos, "bd_embed: sequence length %d exceeds max_position_embeddings %d", L, max_pos);
  bd_drop_args da;
  BD_REQUIRE(bd_drop_make(drop, site, &da), "bd_embed: dropout p=%g outside [0, 1)", (double)drop->p);
  BD_REQUIRE(B > 0 && L > 0 && d > 0 && d % 4 == 0, "bd_embed: bad shape B=%d L=%d d=%d", B, L, d);
  const long long M = (long long)B * L;
  BD_REQUIRE(M < (1ll << 31), "bd_embed: too many tokens");
  bd_prof_scope prof(stream, (double)M * (d * 4.0 + 8.0), BD_PROF_EMBED);
  hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ids,
                     timesteps, tok_emb, pos_emb, time_emb, x, (int)M, L, d, vocab, n_timesteps, da, (const int*)nullptr);
  return bd_check_launch("bd_embed");
}

extern "C" int bd_embed_rows(const long long* ids, const long long* timesteps, const int* rowmap, const float* tok_emb,
                             const float* pos_emb, const float* time_emb, float* x, int n_rows, int L, int d, int vocab,
                             int max_pos, int n_timesteps, const bd_dropout* drop, unsigned site, void* stream) {
  BD_REQUIRE(ids && timesteps && rowmap && tok_emb && pos_emb && time_emb && x, "bd_embed_rows: null operand");
  BD_REQUIRE(vocab > 0 && n_timesteps > 0, "bd_embed_rows: empty table");
  BD_REQUIRE(L <= max_pos, "bd_embed_rows: sequence length %d exceeds max_position_embeddings %d", L, max_pos);
  bd_drop_args da;
  BD_REQUIRE(bd_drop_make(drop, site, &da), "bd_embed_rows: dropout p=%g outside [0, 1)", (double)drop->p);
  BD_REQUIRE(n_rows > 0 && L > 0 && d > 0 && d % 4 == 0, "bd_embed_rows: bad shape rows=%d L=%d d=%d", n_rows, L, d);
  bd_prof_scope prof(stream, (double)n_rows * (d * 4.0 + 12.0), BD_PROF_EMBED);
  hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ids, timesteps,
                     tok_emb, pos_emb, time_emb, x, n_rows, L, d, vocab, n_timesteps, da, rowmap);
  return bd_check_launch("bd_embed_rows");
}

// rowmap[r] = b * L + (r - cu[b]) for the sequence b whose packed rows contain r, -1 for rows >= cu[B]
static __global__ __launch_bounds__(256) void rowmap_kernel(const int* __restrict__ cu, int B, int L, int n_rows,
                                                     int* __restrict__ rowmap) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  int lo = 0, hi = B;                 // largest b with cu[b] <= r
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cu[mid] <= r) lo = mid; else hi = mid;
  }
  rowmap[r] = r < cu[B] ? lo * L + (r - cu[lo]) : -1;
}

extern "C" int bd_build_rowmap(const int* cu_seqlens, int B, int L, int n_rows, int* rowmap, void* stream) {
  BD_REQUIRE(cu_seqlens && rowmap && B > 0 && L > 0 && n_rows > 0, "bd_build_rowmap: bad argument");
  hipLaunchKernelGGL(rowmap_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cu_seqlens, B,
                     L, n_rows, rowmap);
  return bd_check_launch("bd_build_rowmap");
}

// targets / loss flags of the packed rows in one pass: the flag folds the attention mask in, filler rows get none
static __global__ __launch_bounds__(256) void pack_targets_kernel(const long long* __restrict__ targets,
                                                           const unsigned char* __restrict__ indicator,
                                                           const long long* __restrict__ attn,
                                                           const int* __restrict__ rowmap, int n_rows,
                                                           long long* __restrict__ tg_out,
                                                           unsigned char* __restrict__ ind_out) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  const int src = rowmap[r];
  long long tg = 0;
  unsigned char on = 0;
  if (src >= 0) {
    tg = targets[src];
    on = (indicator[src] != 0 && (attn == nullptr || attn[src] != 0)) ? 1 : 0;
  }
  tg_out[r] = tg;
  ind_out[r] = on;
}

extern "C" int bd_pack_targets(const long long* targets, const unsigned char* indicator, const long long* attention_mask,
                               const int* rowmap, int n_rows, long long* targets_out, unsigned char* indicator_out,
                               void* stream) {
  BD_REQUIRE(targets && indicator && rowmap && targets_out && indicator_out && n_rows > 0, "bd_pack_targets: bad argument");
  hipLaunchKernelGGL(pack_targets_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, targets,
                     indicator, attention_mask, rowmap, n_rows, targets_out, indicator_out);
  return bd_check_launch("bd_pack_targets");
}

extern "C" int bd_layernorm(const float* x, const float* gamma, const float* beta, void* out_bf16,
                            float* out_f32, float* stats, int M, int d, float eps, void* stream) {
  BD_REQUIRE(x && gamma && beta && (out_bf16 || out_f32), "bd_layernorm: null operand");
  BD_REQUIRE(M > 0 && d > 0 && d % 4 == 0 && d <= 4096, "bd_layernorm: bad shape M=%d d=%d", M, d);
  bd_prof_scope prof(stream, (double)M * d * (4.0 + (out_bf16 ? 2.0 : 0.0) + (out_f32 ? 4.0 : 0.0)), BD_PROF_LN);
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  unsigned short* ob = (unsigned short*)out_bf16;
  const int nv = (d / 4 + 63) / 64;
  if (nv <= 1)
    hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, x, gamma, beta, ob, out_f32, stats, M, d, eps);
  else if (nv <= 2)
    hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, gamma, beta, ob, out_f32, stats, M, d, eps);
  else if (nv <= 4)
    hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, x, gamma, beta, ob, out_f32, stats, M, d, eps);
  else if (nv <= 8)
    hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, x, gamma, beta, ob, out_f32, stats, M, d, eps);
  else
    hipLaunchKernelGGL(layernorm_kernel<16>, grid, block, 0, s, x, gamma, beta, ob, out_f32, stats, M, d, eps);
  return bd_check_launch("bd_layernorm");
}

extern "C" int bd_corrupt(const long long* ids, const long long* attention_mask,
                          const long long* timesteps, const float* schedule, int schedule_len,
                          const float* uniforms, long long* noisy_ids, unsigned char* mask_indicator,
                          int B, int L, int mask_id, int pad_id, int bos_id, int eos_id,
                          unsigned long long seed, long long seq_offset, void* stream) {
  BD_REQUIRE(ids && timesteps && schedule && noisy_ids && mask_indicator, "bd_corrupt: null operand");
  BD_REQUIRE(B >= 0 && L >= 0 && schedule_len > 0, "bd_corrupt: bad shape");
  bd_prof_scope prof(stream, (double)B * L * (uniforms ? 29.0 : 25.0), BD_PROF_EMBED);
  const long long n = (long long)B * L;
  if (n == 0) return BD_OK;
  const bool aligned = ((((uintptr_t)ids | (uintptr_t)noisy_ids | (uintptr_t)attention_mask | (uintptr_t)uniforms) & 15u) == 0u) &&
                       (((uintptr_t)mask_indicator & 3u) == 0u);
  if (L % 4 == 0 && aligned)
    hipLaunchKernelGGL(corrupt4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ids, attention_mask, timesteps, schedule, uniforms, noisy_ids, mask_indicator, n / 4, L / 4,
                       mask_id, pad_id, bos_id, eos_id, bos_id >= 0 ? 1 : 0, eos_id >= 0 ? 1 : 0, seed, seq_offset);
  else
    hipLaunchKernelGGL(corrupt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ids, attention_mask, timesteps, schedule, uniforms, noisy_ids, mask_indicator, n, L,
                       mask_id, pad_id, bos_id, eos_id, bos_id >= 0 ? 1 : 0, eos_id >= 0 ? 1 : 0, seed,
                       seq_offset);
  return bd_check_launch("bd_corrupt");
}

extern "C" int bd_mask_to_bytes(const long long* attention_mask, unsigned char* out, long long n,
                                void* stream) {
  BD_REQUIRE(attention_mask && out && n >= 0, "bd_mask_to_bytes: bad argument");
  if (n == 0) return BD_OK;
  hipLaunchKernelGGL(mask_to_bytes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, attention_mask, out, n);
  return bd_check_launch("bd_mask_to_bytes");
}
