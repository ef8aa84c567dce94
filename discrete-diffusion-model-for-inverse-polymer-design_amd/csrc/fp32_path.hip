// fp32 compute mode of the denoiser forward: the two kernels the bf16 path has no fp32 counterpart of.
// The reference samples in fp32 (no autocast in src/sampling/sampler.py); this mode reproduces its logits to
// 1e-5 on the GPU at about a twelfth of the bf16 path's speed. Embeddings and LayerNorm are the fp32 kernels
// of elementwise.hip.
//   gemm_f32_kernel       C = A W^T on v_mfma_f32_32x32x2_f32 (f32 operands, f32 accumulation, the vector FMA rate),
//                         128 x 128 tile per workgroup, 64 x 64 per wave, operands staged k-major in LDS,
//                         next k-step's global loads in flight during the MFMAs; bias / erf-GELU / residual epilogues
//   attention_f32_kernel  softmax(Q K^T / sqrt(dh) + key padding) V per (sequence, head, 32 queries): eight lanes
//                         share a query and own dh / 8 features each; two passes over the keys (scores to LDS, then
//                         exp and P V), K and V rows read straight from the qkv rows (L2 resident)
#include "bd_common.h"
#include "bidiff.h"

namespace {

constexpr int GT = 128;          // tile rows / columns
constexpr int GK = 16;           // k-step
constexpr int GLD = GT + 4;      // LDS row pitch of a k-row (floats)

struct GemmF32Args {
  const float* A; const float* W; const float* bias; const float* resid; float* out;
  long long lda, ldw, ldc;
  int M, N, K, epi;
};

__device__ __forceinline__ float gelu_erf_f32(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args p) {
  __shared__ float sA[GK][GLD];
  __shared__ float sW[GK][GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;            // the wave's 64 x 64 quadrant
  const int r = lane & 31, kh = lane >> 5;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  // global -> LDS: thread t moves rows t / 4 and t / 4 + 64, k columns 4 (t % 4) .. + 3 of both operands
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  const float* a0 = p.A + (size_t)min(m0 + lr, p.M - 1) * p.lda + lk;
  const float* a1 = p.A + (size_t)min(m0 + lr + 64, p.M - 1) * p.lda + lk;
  const float* w0 = p.W + (size_t)min(n0 + lr, p.N - 1) * p.ldw + lk;
  const float* w1 = p.W + (size_t)min(n0 + lr + 64, p.N - 1) * p.ldw + lk;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x4 ga0 = *(const f32x4*)a0, ga1 = *(const f32x4*)a1, gw0 = *(const f32x4*)w0, gw1 = *(const f32x4*)w1;
  for (int k0 = 0; k0 < p.K; k0 += GK) {
    __syncthreads();                                   // everyone is done reading the previous step
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sA[lk + e][lr] = ga0[e]; sA[lk + e][lr + 64] = ga1[e];
      sW[lk + e][lr] = gw0[e]; sW[lk + e][lr + 64] = gw1[e];
    }
    __syncthreads();
    if (k0 + GK < p.K) {                               // next step's loads fly during this step's MFMAs
      ga0 = *(const f32x4*)(a0 + k0 + GK); ga1 = *(const f32x4*)(a1 + k0 + GK);
      gw0 = *(const f32x4*)(w0 + k0 + GK); gw1 = *(const f32x4*)(w1 + k0 + GK);
    }
#pragma unroll
    for (int k2 = 0; k2 < GK; k2 += 2) {
      float fa[2], fw[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = sA[k2 + kh][64 * wm + 32 * i + r];
#pragma unroll
      for (int j = 0; j < 2; ++j) fw[j] = sW[k2 + kh][64 * wn + 32 * j + r];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fw[j], acc[i][j], 0, 0, 0);
    }
  }
  // accumulator element 4 q + e of lane (r, kh) = C[row 8 q + 4 kh + e][column r] of its 32 x 32 tile
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + 64 * wn + 32 * j + r;
      if (n >= p.N) continue;
      const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + 64 * wm + 32 * i + 8 * q + 4 * kh + e;
          if (m >= p.M) continue;
          float v = acc[i][j][4 * q + e] + b;
          if (p.epi == 1) v = gelu_erf_f32(v);
          if (p.epi == 2) v += p.resid[(size_t)m * p.ldc + n];
          p.out[(size_t)m * p.ldc + n] = v;
        }
    }
}

constexpr int AQ = 32;           // queries per workgroup

typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int VEC> struct FVec { typedef f32x4 type; };
template <> struct FVec<2> { typedef f32x2_t type; };

template <int DPL>               // head features per lane = dh / 8 (even; 16-byte accesses when it divides by 4, else 8-byte)
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ qkv, const unsigned char* __restrict__ keymask,
                                                            float* __restrict__ out, int B, int L, int H, float scale) {
  extern __shared__ float sS[];                        // [AQ][L] scores, then probabilities
  const int tid = threadIdx.x, part = tid & 7, ql = tid >> 3;
  const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int dh = DPL * 8, d = H * dh;
  const int q = min(qb * AQ + ql, L - 1);              // lanes past the sequence shadow its last query
  const size_t row0 = (size_t)b * L;
  const float* qp = qkv + (row0 + q) * 3 * d + h * dh + part * DPL;
  constexpr int VEC = DPL % 4 == 0 ? 4 : 2;
  typedef typename FVec<VEC>::type vec_t;
  float qv[DPL];
#pragma unroll
  for (int e = 0; e < DPL; e += VEC) {
    const vec_t t = *(const vec_t*)(qp + e);
#pragma unroll
    for (int c = 0; c < VEC; ++c) qv[e + c] = t[c];
  }
  const unsigned char* km = keymask ? keymask + row0 : nullptr;
  float* srow = sS + (size_t)ql * L;
  float mx = -INFINITY;
  for (int k = 0; k < L; ++k) {
    const float* kp = qkv + (row0 + k) * 3 * d + d + h * dh + part * DPL;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < DPL; e += VEC) {
      const vec_t kv = *(const vec_t*)(kp + e);
#pragma unroll
      for (int c = 0; c < VEC; ++c) s += qv[e + c] * kv[c];
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    s = (km == nullptr || km[k]) ? s * scale : -INFINITY;
    if (part == 0) srow[k] = s;
    mx = fmaxf(mx, s);
  }
  __syncthreads();
  float o[DPL];
#pragma unroll
  for (int e = 0; e < DPL; ++e) o[e] = 0.f;
  float sum = 0.f;
  const bool dead = mx == -INFINITY;                   // every key padded: the row's output is 0
  for (int k = 0; k < L; ++k) {
    const float pk = dead ? 0.f : expf(srow[k] - mx);
    sum += pk;
    const float* vp = qkv + (row0 + k) * 3 * d + 2 * d + h * dh + part * DPL;
#pragma unroll
    for (int e = 0; e < DPL; e += VEC) {
      const vec_t vv = *(const vec_t*)(vp + e);
#pragma unroll
      for (int c = 0; c < VEC; ++c) o[e + c] += pk * vv[c];
    }
  }
  if (qb * AQ + ql < L) {
    const float inv = dead ? 0.f : 1.0f / sum;
    float* op = out + (row0 + q) * d + h * dh + part * DPL;
#pragma unroll
    for (int e = 0; e < DPL; e += VEC) {
      vec_t v;
#pragma unroll
      for (int c = 0; c < VEC; ++c) v[c] = o[e + c] * inv;
      *(vec_t*)(op + e) = v;
    }
  }
}

}  // namespace

extern "C" int bd_gemm_f32(const float* A, long long lda, const float* W, long long ldw, const float* bias,
                           const float* resid, float* out, long long ldc, int M, int N, int K, int epilogue,
                           void* stream) {
  BD_REQUIRE(A && W && out, "bd_gemm_f32: null operand");
  BD_REQUIRE(M >= 0 && N > 0 && K > 0 && K % GK == 0, "bd_gemm_f32: M=%d N=%d K=%d (K must be a multiple of %d)", M, N, K, GK);
  BD_REQUIRE(lda % 4 == 0 && ldw % 4 == 0 && lda >= K && ldw >= K && ldc >= N, "bd_gemm_f32: bad leading dimensions");
  BD_REQUIRE((((uintptr_t)A | (uintptr_t)W) & 15u) == 0u, "bd_gemm_f32: operands must be 16-byte aligned");
  BD_REQUIRE(epilogue >= 0 && epilogue <= 2 && (epilogue != 2 || resid), "bd_gemm_f32: epilogue %d", epilogue);
  if (M == 0) return BD_OK;
  GemmF32Args a{A, W, bias, resid, out, lda, ldw, ldc, M, N, K, epilogue};
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((N + GT - 1) / GT, (M + GT - 1) / GT), dim3(256), 0, (hipStream_t)stream, a);
  return bd_check_launch("bd_gemm_f32");
}

extern "C" int bd_attention_f32(const float* qkv, const unsigned char* keymask, float* out, int B, int L, int H,
                                int dh, void* stream) {
  BD_REQUIRE(qkv && out, "bd_attention_f32: null operand");
  BD_REQUIRE(B >= 0 && L > 0 && H > 0, "bd_attention_f32: bad shape");
  BD_REQUIRE(dh >= 16 && dh <= 128 && dh % 16 == 0, "bd_attention_f32: head width %d unsupported (multiples of 16 up to 128)", dh);
  BD_REQUIRE((size_t)AQ * L * 4 <= 64 * 1024, "bd_attention_f32: L=%d too long", L);
  if (B == 0) return BD_OK;
  const dim3 grid((L + AQ - 1) / AQ, H, B);
  const size_t lds = (size_t)AQ * L * 4;
  const float scale = 1.0f / sqrtf((float)dh);
  hipStream_t s = (hipStream_t)stream;
  switch (dh / 8) {
#define BD_ATTN_F32_CASE(DPL) \
    case DPL: hipLaunchKernelGGL(attention_f32_kernel<DPL>, grid, dim3(256), lds, s, qkv, keymask, out, B, L, H, scale); break;
    BD_ATTN_F32_CASE(2) BD_ATTN_F32_CASE(4) BD_ATTN_F32_CASE(6) BD_ATTN_F32_CASE(8)
    BD_ATTN_F32_CASE(10) BD_ATTN_F32_CASE(12) BD_ATTN_F32_CASE(14) BD_ATTN_F32_CASE(16)
#undef BD_ATTN_F32_CASE
    default: return bd_fail(BD_ERR_UNSUPPORTED, "bd_attention_f32: head width %d", dh);
  }
  return bd_check_launch("bd_attention_f32");
}
