// Row-panel bf16 MFMA GEMM for skinny-K layers (K is a few hundred, M is huge):
//     C[M,N] = A[M,K] * W[N,K]^T (+ bias) with a fused epilogue, optionally LayerNorm(A) first.
//
// The Linear layers of the denoiser have K = d or K = F with tiny d, so each output element
// costs only a few hundred MACs and the layers are HBM-bound: what matters is that every
// activation byte is read once and every output byte written once. A 512-thread workgroup owns a
// 128-row panel of A and keeps its MFMA fragments IN REGISTERS (K is consumed in chunks of
// 32*KS32 <= 384), so the LDS only carries the weight stream: W tiles of 128 x 64 flow through
// a two-buffer LDS ring fed by a two-stage register prefetch (loads are issued two stages
// ahead of their use). With LN_A the panel is loaded as fp32 rows, normalised in registers
// (the row is spread over four lanes) and rounded to bf16 on the fly: gamma / beta are folded
// into the weights and the bias by the caller, so no normalised copy of the activations ever
// goes to memory.
//
// Wave layout: 8 waves = 4 (rows) x 2 (columns); a wave owns 32 rows x 64 columns of each
// 128 x 128 output tile; NACC output tiles are accumulated at once (NACC = 3 covers N = 384,
// which lets a K-deep layer stream K once).
#include "bd_common.h"
#include "bd_prof.h"

namespace {

constexpr int PBM = 128, PBN = 128, PBK = 64;
constexpr int W_TILE = PBN * PBK * 2;   // 16 KiB
constexpr int MAX_BIAS = 4096;

struct PanelArgs {
  const void* A;             // bf16 [M, K] or (LN_A) fp32 [M, K]
  const unsigned short* W;   // bf16 [N, K]
  const float* bias;         // [N] or null
  void* out;                 // bf16 [M, ldc] (epi 0/1) or fp32 [M, ldc] (epi 2)
  const float* resid;        // epi 2: residual input (may equal out)
  int M, N, K, ldc;
  int nkc, ngroups;          // K chunks, groups of NACC output tiles
  float eps;
};

__device__ __forceinline__ int pswz_w(int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); }

__device__ __forceinline__ float perf_as(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(1.0f + 0.3275911f * ax);
  float p = 1.061405429f;
  p = p * t + -1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t + -0.284496736f;
  p = p * t + 0.254829592f;
  const float y = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float pgelu(float x) { return 0.5f * x * (1.0f + perf_as(x * 0.70710678118654752f)); }

template <int KS32, int NACC, int EPI, bool LN_A>
__global__ __launch_bounds__(512, 2) void gemm_panel_kernel(PanelArgs p) {
  constexpr int KC = 32 * KS32;      // K chunk held in registers
  constexpr int KS64 = KS32 / 2;     // W stages per (chunk, tile)
  // prefetch depth = number of register sets = number of LDS buffers; the stage position inside a
  // K chunk fixes both indices at compile time, so D must divide NACC * KS64
  constexpr int D = ((NACC * KS64) % 3 == 0) ? 3 : 2;
  static_assert((NACC * KS64) % D == 0, "stage count per chunk must be a multiple of the pipeline depth");
  __shared__ __attribute__((aligned(16))) unsigned char smem[D * W_TILE + MAX_BIAS * 4];
  float* const sbias = (float*)(smem + D * W_TILE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, c = lane & 15;
  const int m0 = blockIdx.x * PBM;
  const int K = p.K;

  for (int i = tid; i < p.N; i += 512) sbias[i] = p.bias ? p.bias[i] : 0.f;

  // ---- W staging: thread -> two 16-byte chunks of a 128 x 64 tile (row = tid/8 + 64 i, chunk tid%8)
  const int srow = tid >> 3, sch = tid & 7;
  int soff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = srow + 64 * i;
    soff[i] = r * 128 + ((sch ^ pswz_w(r)) << 4);
  }
  const int stages_per_group = p.nkc * NACC * KS64;
  const int S = p.ngroups * stages_per_group;
  // weight-tile source for stage (grp, kc, nb, k64): multiplications only, no division
  const unsigned short* const wthread = p.W + (size_t)srow * K + sch * 8;
  auto w_src = [&](int grp, int kc, int nb, int k64, int i) -> const unsigned short* {
    const int n = (grp * NACC + nb) * PBN + 64 * i;
    return wthread + (size_t)n * K + kc * KC + k64 * PBK;
  };
  u32x4 rw[D][2];

  // ---- fragment rows
  int wrow[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) wrow[t] = wn * 64 + 16 * (c >> 2) + 4 * t + (c & 3);
  int arow[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int m = m0 + wm * 32 + 16 * mt + c;
    arow[mt] = m < p.M ? m : p.M - 1;
  }

  bf16x8 afr[2][KS32];
  auto load_a = [&](int kc) {
    if (LN_A) {
      // two sweeps over the fp32 rows (the second one hits L1 / L2): statistics, then normalise
      // straight into the bf16 fragments - no fp32 copy of the panel is kept in registers
      const float* A = (const float*)p.A;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const float* row = A + (size_t)arow[mt] * K + 8 * g;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
          const f32x4 a0 = *(const f32x4*)(row + 32 * ks), a1 = *(const f32x4*)(row + 32 * ks + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { s1 += a0[e] + a1[e]; s2 += a0[e] * a0[e] + a1[e] * a1[e]; }
        }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        const float mean = s1 / (float)KC;
        const float var = fmaxf(s2 / (float)KC - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
          const f32x4 a0 = *(const f32x4*)(row + 32 * ks), a1 = *(const f32x4*)(row + 32 * ks + 4);
          bf16x8 f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[e] = (__bf16)((a0[e] - mean) * rstd);
            f[4 + e] = (__bf16)((a1[e] - mean) * rstd);
          }
          afr[mt][ks] = f;
        }
      }
    } else {
      const unsigned short* A = (const unsigned short*)p.A;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const unsigned short* row = A + (size_t)arow[mt] * K + kc * KC + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) afr[mt][ks] = *(const bf16x8*)(row + 32 * ks);
      }
    }
  };

  f32x4 acc[NACC][4][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[a][t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  // ---- prologue: stages 0 and 1 in registers, stage 0 in LDS
#pragma unroll
  for (int j = 0; j < D; ++j) {
    // the first D stages are (0, 0, nb, k64) with nb * KS64 + k64 = j  (D <= NACC * KS64)
#pragma unroll
    for (int i = 0; i < 2; ++i) rw[j][i] = *(const u32x4*)w_src(0, 0, j / KS64, j % KS64, i);
  }
  load_a(0);
#pragma unroll
  for (int i = 0; i < 2; ++i) *(u32x4*)(smem + soff[i]) = rw[0][i];
  zero_acc();
  __syncthreads();

  int s = 0;
  for (int grp = 0; grp < p.ngroups; ++grp) {
    for (int kc = 0; kc < p.nkc; ++kc) {
      if (p.nkc > 1 && !(grp == 0 && kc == 0)) load_a(kc);
#pragma unroll
      for (int nb = 0; nb < NACC; ++nb) {
#pragma unroll
        for (int k64 = 0; k64 < KS64; ++k64, ++s) {
          const int pos = nb * KS64 + k64;          // compile-time after unrolling
          const int set = pos % D;                  // register set and LDS buffer of stage s
          // 1. prefetch stage s + D into the register set of stage s (already written to LDS)
          if (s + D < S) {
            int pg = grp, pk = kc, ppos = pos + D;
            if (ppos >= NACC * KS64) {
              ppos -= NACC * KS64;
              pk = kc + 1;
              if (pk == p.nkc) { pk = 0; pg = grp + 1; }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) rw[set][i] = *(const u32x4*)w_src(pg, pk, ppos / KS64, ppos % KS64, i);
          }
          // 2. compute stage s from its LDS buffer
          const unsigned char* Wb = smem + set * W_TILE;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int ch = 4 * kk + g;
            bf16x8 fw[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
              fw[t] = *(const bf16x8*)(Wb + wrow[t] * 128 + ((ch ^ pswz_w(wrow[t])) << 4));
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int mt = 0; mt < 2; ++mt)
                acc[nb][t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[t], afr[mt][2 * k64 + kk], acc[nb][t][mt], 0, 0, 0);
          }
          // 3. stage s + 1 (in registers since D - 1 iterations) goes to the next LDS buffer
          if (s + 1 < S) {
            unsigned char* Wn = smem + ((set + 1) % D) * W_TILE;
#pragma unroll
            for (int i = 0; i < 2; ++i) *(u32x4*)(Wn + soff[i]) = rw[(set + 1) % D][i];
          }
          __syncthreads();
        }
      }
    }
    // ---- epilogue of this group: lane (g, c) holds n = nbase + 16 g + 4 t + r for row c of m-tile mt
#pragma unroll
    for (int nb = 0; nb < NACC; ++nb) {
      const int nbase = (grp * NACC + nb) * PBN + wn * 64 + 16 * g;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + wm * 32 + 16 * mt + c;
        float v[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * t + r] = acc[nb][t][mt][r] + sbias[nbase + 4 * t + r];
        if (m < p.M) {
          if (EPI == 0 || EPI == 1) {
            if (EPI == 1) {
#pragma unroll
              for (int j = 0; j < 16; ++j) v[j] = pgelu(v[j]);
            }
            unsigned short* o = (unsigned short*)p.out + (size_t)m * p.ldc + nbase;
            u32x4 lo, hi;
            lo[0] = bd_pack_bf16x2(v[0], v[1]);   lo[1] = bd_pack_bf16x2(v[2], v[3]);
            lo[2] = bd_pack_bf16x2(v[4], v[5]);   lo[3] = bd_pack_bf16x2(v[6], v[7]);
            hi[0] = bd_pack_bf16x2(v[8], v[9]);   hi[1] = bd_pack_bf16x2(v[10], v[11]);
            hi[2] = bd_pack_bf16x2(v[12], v[13]); hi[3] = bd_pack_bf16x2(v[14], v[15]);
            *(u32x4*)(o) = lo;
            *(u32x4*)(o + 8) = hi;
          } else {
            float* o = (float*)p.out + (size_t)m * p.ldc + nbase;
            const float* ri = p.resid + (size_t)m * p.ldc + nbase;
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
              f32x4 x = *(const f32x4*)(ri + 4 * j4);
              x[0] += v[4 * j4 + 0]; x[1] += v[4 * j4 + 1]; x[2] += v[4 * j4 + 2]; x[3] += v[4 * j4 + 3];
              *(f32x4*)(o + 4 * j4) = x;
            }
          }
        }
      }
    }
    zero_acc();
  }
}

template <int KS32, int NACC, int EPI, bool LN_A>
int launch_panel(const PanelArgs& a, hipStream_t s) {
  const int blocks = (a.M + PBM - 1) / PBM;
  void* tok = bd_prof_on() ? bd_prof_begin(s) : nullptr;
  hipLaunchKernelGGL((gemm_panel_kernel<KS32, NACC, EPI, LN_A>), dim3(blocks), dim3(512), 0, s, a);
  if (tok) bd_prof_end(tok, s, 2.0 * (double)a.M * (double)a.N * (double)a.K, BD_PROF_PANEL);
  return bd_check_launch("bd_gemm_panel");
}

template <int KS32, int NACC>
int dispatch_epi(const PanelArgs& a, int epi, bool ln, hipStream_t s) {
  if (ln) {
    if constexpr (NACC == 1) {
      if (epi == 0) return launch_panel<KS32, 1, 0, true>(a, s);
      if (epi == 1) return launch_panel<KS32, 1, 1, true>(a, s);
    }
    return bd_fail(BD_ERR_UNSUPPORTED, "bd_gemm_panel: fused LayerNorm needs epilogue 0 or 1 and a single K chunk");
  }
  switch (epi) {
    case 0: return launch_panel<KS32, NACC, 0, false>(a, s);
    case 1: return launch_panel<KS32, NACC, 1, false>(a, s);
    default: return launch_panel<KS32, NACC, 2, false>(a, s);
  }
}

}  // namespace

namespace {
// path A: one output tile at a time, the whole K in registers (K in {128, 256, 384}), or N = 128 with
//         K a multiple of one of those; path B: three output tiles at a time (N a multiple of 384),
//         K in chunks of 192 (or 128).
int chunk_single(int K) {
  if (K == 384 || K == 256 || K == 128) return K;
  return 0;
}
int chunk_multi_a(int K) { return (K % 384 == 0) ? 384 : ((K % 256 == 0) ? 256 : ((K % 128 == 0) ? 128 : 0)); }
int chunk_b(int K) { return (K % 192 == 0) ? 192 : ((K % 128 == 0) ? 128 : 0); }
}  // namespace

// 1 when bd_gemm_panel supports the shape (else use bd_gemm_bf16).
extern "C" int bd_gemm_panel_supported(int N, int K, int layernorm_a) {
  if (N <= 0 || K <= 0 || N % PBN != 0 || N > MAX_BIAS) return 0;
  const int ntiles = N / PBN;
  if (layernorm_a) return chunk_single(K) != 0;
  if (chunk_single(K)) return 1;
  if (ntiles % 3 == 0 && chunk_b(K)) return 1;
  if (ntiles == 1 && chunk_multi_a(K)) return 1;
  return 0;
}

// epilogue 0: bf16 = acc + bias, 1: bf16 = gelu(acc + bias), 2: fp32 out = resid + acc + bias.
// layernorm_a: A is fp32 [M,K]; rows are normalised ((x - mean) * rstd, eps inside the sqrt) before
// the product - gamma and beta must already be folded into W and bias.
extern "C" int bd_gemm_panel(const void* A, const void* W, const float* bias, void* out, const float* resid,
                             int M, int N, int K, int ldc, int epilogue, int layernorm_a, float eps,
                             void* stream) {
  BD_REQUIRE(A && W && out, "bd_gemm_panel: null operand");
  BD_REQUIRE(M > 0, "bd_gemm_panel: empty M");
  BD_REQUIRE(epilogue >= 0 && epilogue <= 2, "bd_gemm_panel: unknown epilogue %d", epilogue);
  BD_REQUIRE(bd_gemm_panel_supported(N, K, layernorm_a), "bd_gemm_panel: unsupported shape N=%d K=%d ln=%d", N, K,
             layernorm_a);
  BD_REQUIRE(ldc >= N && ldc % 8 == 0, "bd_gemm_panel: bad ldc");
  BD_REQUIRE(epilogue != 2 || resid != nullptr, "bd_gemm_panel: residual epilogue needs resid");
  PanelArgs a;
  a.A = A; a.W = (const unsigned short*)W; a.bias = bias; a.out = out; a.resid = resid;
  a.M = M; a.N = N; a.K = K; a.ldc = ldc; a.eps = eps;
  const int ntiles = N / PBN;
  hipStream_t s = (hipStream_t)stream;
  const bool ln = layernorm_a != 0;
  const bool single = chunk_single(K) != 0;
  const bool path_b = !ln && ntiles % 3 == 0 && chunk_b(K) != 0 && (epilogue == 2 || !single);
  if (path_b) {
    const int kc = chunk_b(K);
    a.nkc = K / kc;
    a.ngroups = ntiles / 3;
    if (kc == 192) return dispatch_epi<6, 3>(a, epilogue, false, s);
    return dispatch_epi<4, 3>(a, epilogue, false, s);
  }
  const int kc = single ? K : chunk_multi_a(K);
  BD_REQUIRE(single || ntiles == 1, "bd_gemm_panel: unsupported shape N=%d K=%d", N, K);
  a.nkc = K / kc;
  a.ngroups = ntiles;
  switch (kc) {
    case 384: return dispatch_epi<12, 1>(a, epilogue, ln, s);
    case 256: return dispatch_epi<8, 1>(a, epilogue, ln, s);
    default: return dispatch_epi<4, 1>(a, epilogue, ln, s);
  }
}
