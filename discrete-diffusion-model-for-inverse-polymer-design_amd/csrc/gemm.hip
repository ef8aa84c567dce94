// bf16 MFMA GEMM for the denoiser's Linear layers on gfx950:
//     C[M,N] = A[M,K] * W[N,K]^T (+ bias) with a fused epilogue.
// Both operands are K-contiguous (torch.nn.Linear keeps W as [out,in]), which is the
// natural MFMA operand order: every fragment is one 16-byte LDS read.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 each),
// BK = 64, 16x16x32 bf16 MFMA, fp32 accumulate. The weight rows are the MFMA "A"
// operand and the activation rows the "B" operand, so a lane ends up holding
// consecutive output columns of ONE activation row; the weight rows are additionally
// permuted inside each wave so that a lane's 16 accumulators of one m-tile are 16
// consecutive n: the epilogue stores 16-byte vectors straight from registers.
#include "bd_common.h"
#include "bd_prof.h"
#include "bd_drop.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // one operand tile: 128 rows x 128 B

enum Epi : int {
  EPI_BIAS_BF16 = 0,       // out bf16 = acc + bias
  EPI_BIAS_GELU_BF16 = 1,  // out bf16 = gelu(acc + bias)   (erf form, F.gelu default)
  EPI_BIAS_RESID_F32 = 2,  // out fp32 += acc + bias        (residual stream, in place)
  EPI_F32 = 3,             // out fp32 = acc (+ bias if given), columns >= n_real dropped
  EPI_DGELU_BF16 = 4,      // out bf16 = (acc + bias) * gelu'(aux)   (aux = pre-activation, bf16)
  EPI_AXPBY_BF16 = 5,      // out bf16 = rn(rn(beta * aux) + rn(alpha * rn(acc))), aux bf16 laid out like out; every
                           // product and the sum rounded to bf16 as a chain of bf16 tensor ops would
  EPI_GELU_GRAD_BF16 = 6,  // out bf16 = drop(gelu(acc + bias)), aux_out bf16 = gelu'(acc + bias) * (the same dropout
                           // factor): the forward pass leaves the backward pass its multiplier instead of the pre-activation
  EPI_MUL_AUX_BF16 = 7     // out bf16 = (acc + bias) * aux_in   (aux_in bf16 laid out like out; backward of epilogue 6)
};

struct GemmArgs {
  const unsigned short* A;   // [M, K] bf16
  const unsigned short* W;   // [Npad, K] bf16, Npad % 128 == 0 (zero rows past n_real)
  const float* bias;         // [n_real] or null
  void* out;                 // bf16 or fp32, row stride ldc elements
  const void* aux_in;        // EPI 2: residual input fp32 (null = out itself); EPI 4: pre-activation bf16
  void* aux_out;             // EPI 1: optional bf16 copy of the pre-activation (acc + bias)
  int M, Npad, K, n_real, ldc;
  int tiles_n, n_tiles;
  bd_drop_args drop;         // DROP kernels: dropout of the epilogue value at (row m, column n)
  long long sA, sW, sOut;    // batched launch (gridDim.y problems): element strides of A, W and out / aux_in
  float alpha, beta;         // EPI 5
  // LayerNorm deferred across a residual projection (eight-wave kernel only; all null otherwise):
  unsigned short* xb_out;    // EPI 2: bf16 copy of the updated residual rows (row stride ldc)
  float* part_out;           // EPI 2: [Npad / 64][M][2] sum and sum of squares of the updated rows per 64-column block
  const float* ln_stats;     // EPI 0 / 1: [M][2] mean and 1 / std of the rows of A; value = rstd * (acc - mean * colsum[n]) + bias[n]
  const float* ln_colsum;    // [Npad] sums of the weight rows
};

__device__ __forceinline__ int swz_a(int row) { return row & 7; }
__device__ __forceinline__ int swz_w(int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); }

// Phi(x) = 0.5 (1 + erf(x / sqrt 2)) as clamp(0.5 + x P(min(x^2, 16)), 0, 1): x P(x^2) is a degree-7
// odd minimax fit of 0.5 erf(x / sqrt 2) on |x| <= 4, past that the product overshoots +-0.5 and
// the clamp saturates it. |error| < 2e-4, below the bf16 rounding of the activations; no
// transcendental (the same polynomial the fused inference kernels evaluate).
__device__ __forceinline__ float norm_cdf(float x) {
  const float s = fminf(x * x, 16.0f);
  float p = s * -1.903182666e-09f + 1.410586241e-07f;
  p = p * s + -4.565313249e-06f;
  p = p * s + 8.634553992e-05f;
  p = p * s + -1.085383136e-03f;
  p = p * s + 9.789848745e-03f;
  p = p * s + -6.636063135e-02f;
  p = p * s + 3.989269649e-01f;
  return __builtin_amdgcn_fmed3f(x * p + 0.5f, 0.0f, 1.0f);
}
__device__ __forceinline__ float gelu_erf(float x) { return x * norm_cdf(x); }
// d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_grad(float x) {
  return norm_cdf(x) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// keep / (1 - p) factors of the 16 consecutive columns nb .. nb + 15 of row m
__device__ __forceinline__ void drop_factors16(const bd_drop_args& d, int m, int nb, float (&f)[16]) {
#pragma unroll
  for (int j8 = 0; j8 < 2; ++j8) {
    const bd_philox_out w = bd_drop_block(d, (uint32_t)m, (uint32_t)(nb >> 3) + j8);
#pragma unroll
    for (int r = 0; r < 8; ++r) f[8 * j8 + r] = bd_drop_factor(d, w, r);
  }
}

// Epilogue of one output row segment: v = acc + bias for the 16 consecutive columns nb .. nb + 15
// of row m; applies the epilogue's activation / dropout / residual and stores.
template <int EPI, bool DROP>
__device__ __forceinline__ void epilogue_row(const GemmArgs& p, int m, int nb, float (&v)[16]) {
  if (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_GELU_BF16 || EPI == EPI_DGELU_BF16 || EPI == EPI_AXPBY_BF16 ||
      EPI == EPI_GELU_GRAD_BF16 || EPI == EPI_MUL_AUX_BF16) {
    if (EPI == EPI_GELU_GRAD_BF16) {
      float f[16];
      if (DROP) drop_factors16(p.drop, m, nb, f);
      float gp[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float x = v[j], cdf = norm_cdf(x);
        const float fj = DROP ? f[j] : 1.0f;
        gp[j] = (cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x)) * fj;
        v[j] = x * cdf * fj;
      }
      unsigned short* uo = (unsigned short*)p.aux_out + (size_t)m * p.ldc + nb;
      u32x4 lo, hi;
      lo[0] = bd_pack_bf16x2(gp[0], gp[1]);   lo[1] = bd_pack_bf16x2(gp[2], gp[3]);
      lo[2] = bd_pack_bf16x2(gp[4], gp[5]);   lo[3] = bd_pack_bf16x2(gp[6], gp[7]);
      hi[0] = bd_pack_bf16x2(gp[8], gp[9]);   hi[1] = bd_pack_bf16x2(gp[10], gp[11]);
      hi[2] = bd_pack_bf16x2(gp[12], gp[13]); hi[3] = bd_pack_bf16x2(gp[14], gp[15]);
      *(u32x4*)(uo) = lo;
      *(u32x4*)(uo + 8) = hi;
    }
    if (EPI == EPI_MUL_AUX_BF16) {
      const unsigned short* ui = (const unsigned short*)p.aux_in + (size_t)m * p.ldc + nb;
      const u32x4 lo = *(const u32x4*)(ui), hi = *(const u32x4*)(ui + 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] *= __uint_as_float(lo[e] << 16);
        v[2 * e + 1] *= __uint_as_float(lo[e] & 0xffff0000u);
        v[8 + 2 * e] *= __uint_as_float(hi[e] << 16);
        v[8 + 2 * e + 1] *= __uint_as_float(hi[e] & 0xffff0000u);
      }
    }
    if (EPI == EPI_AXPBY_BF16) {
      const unsigned short* xi = (const unsigned short*)p.aux_in + (size_t)m * p.ldc + nb;
      const u32x4 lo = *(const u32x4*)(xi), hi = *(const u32x4*)(xi + 8);
      float x[16];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[2 * e] = __uint_as_float(lo[e] << 16);     x[2 * e + 1] = __uint_as_float(lo[e] & 0xffff0000u);
        x[8 + 2 * e] = __uint_as_float(hi[e] << 16); x[8 + 2 * e + 1] = __uint_as_float(hi[e] & 0xffff0000u);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float prod = bd_bf16_to_f32(bd_f32_to_bf16(v[j]));
        v[j] = bd_bf16_to_f32(bd_f32_to_bf16(p.beta * x[j])) + bd_bf16_to_f32(bd_f32_to_bf16(p.alpha * prod));
      }
    }
    if (EPI == EPI_BIAS_GELU_BF16) {
      if (p.aux_out != nullptr) {
        unsigned short* uo = (unsigned short*)p.aux_out + (size_t)m * p.ldc + nb;
        u32x4 lo, hi;
        lo[0] = bd_pack_bf16x2(v[0], v[1]);   lo[1] = bd_pack_bf16x2(v[2], v[3]);
        lo[2] = bd_pack_bf16x2(v[4], v[5]);   lo[3] = bd_pack_bf16x2(v[6], v[7]);
        hi[0] = bd_pack_bf16x2(v[8], v[9]);   hi[1] = bd_pack_bf16x2(v[10], v[11]);
        hi[2] = bd_pack_bf16x2(v[12], v[13]); hi[3] = bd_pack_bf16x2(v[14], v[15]);
        *(u32x4*)(uo) = lo;
        *(u32x4*)(uo + 8) = hi;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = gelu_erf(v[j]);
    }
    if (DROP && EPI != EPI_GELU_GRAD_BF16) {
      float f[16];
      drop_factors16(p.drop, m, nb, f);
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] *= f[j];
    }
    if (EPI == EPI_DGELU_BF16) {
      const unsigned short* ui = (const unsigned short*)p.aux_in + (size_t)m * p.ldc + nb;
      const u32x4 lo = *(const u32x4*)(ui), hi = *(const u32x4*)(ui + 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] *= gelu_grad(__uint_as_float(lo[e] << 16));
        v[2 * e + 1] *= gelu_grad(__uint_as_float(lo[e] & 0xffff0000u));
        v[8 + 2 * e] *= gelu_grad(__uint_as_float(hi[e] << 16));
        v[8 + 2 * e + 1] *= gelu_grad(__uint_as_float(hi[e] & 0xffff0000u));
      }
    }
    unsigned short* o = (unsigned short*)p.out + (size_t)m * p.ldc + nb;
    u32x4 lo, hi;
    lo[0] = bd_pack_bf16x2(v[0], v[1]);   lo[1] = bd_pack_bf16x2(v[2], v[3]);
    lo[2] = bd_pack_bf16x2(v[4], v[5]);   lo[3] = bd_pack_bf16x2(v[6], v[7]);
    hi[0] = bd_pack_bf16x2(v[8], v[9]);   hi[1] = bd_pack_bf16x2(v[10], v[11]);
    hi[2] = bd_pack_bf16x2(v[12], v[13]); hi[3] = bd_pack_bf16x2(v[14], v[15]);
    *(u32x4*)(o) = lo;
    *(u32x4*)(o + 8) = hi;
  } else if (EPI == EPI_BIAS_RESID_F32) {
    float* o = (float*)p.out + (size_t)m * p.ldc + nb;
    const float* ri = p.aux_in ? (const float*)p.aux_in + (size_t)m * p.ldc + nb : o;
    if (DROP) {
      float f[16];
      drop_factors16(p.drop, m, nb, f);
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] *= f[j];
    }
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      f32x4 x = *(const f32x4*)(ri + 4 * j4);
      x[0] += v[4 * j4 + 0]; x[1] += v[4 * j4 + 1]; x[2] += v[4 * j4 + 2]; x[3] += v[4 * j4 + 3];
      *(f32x4*)(o + 4 * j4) = x;
    }
  } else {  // EPI_F32, ragged N: rows of n_real floats are only 4-byte aligned, the 16-byte stores say so
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    float* o = (float*)p.out + (size_t)m * p.ldc;
    if (nb + 16 <= p.n_real) {
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        f32x4_u x;
        x[0] = v[4 * j4]; x[1] = v[4 * j4 + 1]; x[2] = v[4 * j4 + 2]; x[3] = v[4 * j4 + 3];
        *(f32x4_u*)(o + nb + 4 * j4) = x;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (nb + j < p.n_real) o[nb + j] = v[j];
    }
  }
}

// bias of the 16 consecutive columns nb .. nb + 15 (zeros without a bias)
template <int EPI>
__device__ __forceinline__ void load_bias16(const GemmArgs& p, int nb, float (&bv)[16]) {
#pragma unroll
  for (int j = 0; j < 16; ++j) bv[j] = 0.f;
  if (p.bias != nullptr) {
    if (EPI == EPI_F32) {
#pragma unroll
      for (int j = 0; j < 16; ++j) bv[j] = (nb + j) < p.n_real ? p.bias[nb + j] : 0.f;
    } else {
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const f32x4 t = *(const f32x4*)(p.bias + nb + 4 * j4);
        bv[4 * j4 + 0] = t[0]; bv[4 * j4 + 1] = t[1]; bv[4 * j4 + 2] = t[2]; bv[4 * j4 + 3] = t[3];
      }
    }
  }
}

template <int EPI, bool DROP>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];
  unsigned char* const As = smem;                   // 2 buffers
  unsigned char* const Ws = smem + 2 * TILE_BYTES;  // 2 buffers

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, c = lane & 15;
  if (EPI == EPI_AXPBY_BF16 || EPI == EPI_BIAS_BF16) {   // the epilogues batched launches use
    const long long b = blockIdx.y;
    p.A += b * p.sA;
    p.W += b * p.sW;
    p.out = (unsigned short*)p.out + b * p.sOut;
    if (EPI == EPI_AXPBY_BF16) p.aux_in = (const unsigned short*)p.aux_in + b * p.sOut;
  }

  const int K = p.K;
  const int nk = K / BK;

  // ---- staging: thread -> (row = tid/8 + 32 i, 16-byte chunk = tid%8); tile-independent LDS offsets
  const int srow = tid >> 3, sch = tid & 7;
  int offa[4], offw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = srow + 32 * i;
    offa[i] = r * 128 + ((sch ^ swz_a(r)) << 4);
    offw[i] = r * 128 + ((sch ^ swz_w(r)) << 4);
  }
  // ---- fragment read offsets (per kk add 64 B chunk pairs through the XOR)
  int fa_row[4], fw_row[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    fa_row[t] = wm * 64 + 16 * t + c;                          // activation row (MFMA col)
    fw_row[t] = wn * 64 + 16 * (c >> 2) + 4 * t + (c & 3);     // permuted weight row
  }

  // Persistent workgroups: workgroup w takes the logical ids w, w + gridDim.x, ... (gridDim.x is a
  // multiple of 8 or the tile count itself, so a logical id keeps the workgroup's XCD = id % 8).
  // XCD-aware, bijective remap of a logical id: ids that share an XCD walk a contiguous run of tiles
  // (n fastest), so the A row-panel is re-read from that XCD's L2.
  const unsigned short* ga[4];
  const unsigned short* gw[4];
  int nm0 = 0, nn0 = 0;
  auto setup = [&](int logical) {     // addresses of a tile's operands; sets nm0 / nn0
    const int nwg = p.n_tiles;
    const int q = nwg >> 3, r = nwg & 7, xcd = logical & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (logical >> 3);
    nm0 = (tile / p.tiles_n) * BM;
    nn0 = (tile % p.tiles_n) * BN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = srow + 32 * i;
      int am = nm0 + rr; am = am < p.M ? am : p.M - 1;
      ga[i] = p.A + (size_t)am * K + sch * 8;
      gw[i] = p.W + (size_t)(nn0 + rr) * K + sch * 8;
    }
  };

  u32x4 ra[4], rw[4];
  setup(blockIdx.x);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = *(const u32x4*)(ga[i]);
    rw[i] = *(const u32x4*)(gw[i]);
  }

  for (int logical = blockIdx.x; logical < p.n_tiles; logical += gridDim.x) {
  const int m0 = nm0, n0 = nn0;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *(u32x4*)(As + offa[i]) = ra[i];
    *(u32x4*)(Ws + offw[i]) = rw[i];
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = (kt + 1) < nk;
    if (more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *(const u32x4*)(ga[i] + (kt + 1) * BK);
        rw[i] = *(const u32x4*)(gw[i] + (kt + 1) * BK);
      }
    }
    const unsigned char* Ab = As + buf * TILE_BYTES;
    const unsigned char* Wb = Ws + buf * TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = 4 * kk + g;
      bf16x8 fa[4], fw[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = *(const bf16x8*)(Ab + fa_row[t] * 128 + ((ch ^ swz_a(fa_row[t])) << 4));
        fw[t] = *(const bf16x8*)(Wb + fw_row[t] * 128 + ((ch ^ swz_w(fw_row[t])) << 4));
      }
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[tn], fa[tm], acc[tn][tm], 0, 0, 0);
    }
    if (more) {
      unsigned char* An = As + (buf ^ 1) * TILE_BYTES;
      unsigned char* Wn = Ws + (buf ^ 1) * TILE_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *(u32x4*)(An + offa[i]) = ra[i];
        *(u32x4*)(Wn + offw[i]) = rw[i];
      }
    }
    __syncthreads();
  }

  // the next tile's first operand step is already on its way while this tile's epilogue runs
  if (logical + (int)gridDim.x < p.n_tiles) {
    setup(logical + gridDim.x);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *(const u32x4*)(ga[i]);
      rw[i] = *(const u32x4*)(gw[i]);
    }
  }

  // ---- epilogue: lane (g, c) holds, for m-tile tm, n = nb + 4 tn + r (16 consecutive)
  const int nb = n0 + wn * 64 + 16 * g;
  float bv[16];
  load_bias16<EPI>(p, nb, bv);
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int m = m0 + wm * 64 + 16 * tm + c;
    if (m >= p.M) continue;
    float v[16];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * tn + r] = acc[tn][tm][r] + bv[4 * tn + r];
    epilogue_row<EPI, DROP>(p, m, nb, v);
  }
  }   // persistent tile loop
}

// -------------------------------------------------------------------------------------------------
// Large-tile variant for Npad % 256 == 0: 256 x 256 output tile per workgroup, 4 waves (2 x 2) of
// 128 x 128 each, 32x32x16 MFMAs, fp32 accumulators in 256 registers per lane, one wave per SIMD.
// The 64 x 64 wave tile above reads 1 KiB of LDS per 2 MFMAs and saturates the LDS pipe together
// with the MFMA pipe; a 128 x 128 wave tile reads half as much per flop.
//  * Operands reach LDS by LDS-DMA as whole 64-byte memory requests: a 1-KiB piece is 16 rows x 64 B (BK = 32), so
//    a DMA instruction makes 16 requests and uses all of each (a piece laid out as one MFMA operand, 32 rows x 32 B,
//    makes 32 for half of each, and the CU's request rate then paces the kernel). Inside a row the four 16-byte
//    k-chunks are stored at slot chunk ^ ((row >> 2) & 3): the operand read of an MFMA (32 rows, one k-chunk per
//    half-wave) then covers every bank once per 16 lanes.
//  * Weight rows are permuted inside a 32-row tile (MFMA row rho <-> column 16 ((rho >> 2) & 1) +
//    4 (rho >> 3) + (rho & 3)) so that a lane's 16 accumulators of a tile are 16 consecutive n.
//  * Four 32-KiB stages (BK = 32), two to three of them in flight; counted vmcnt waits (loads return in order), one
//    barrier per stage in the middle of a step. LDS reads are inline asm with their own waits (the compiler would
//    wait for the DMA in flight).
//  * What paces it (8192^3, one box, measured on the two-stage BK = 64 form): 1.07 ms; without the DMA 0.84, without
//    the operand reads 1.00, with neither 0.59 (the matrix cores alone, 1.87 PFLOP/s: the chip's sustained clock
//    under this load). 64 KB per CU per 64 k-values against about 25 B per clock that a CU takes in at best through
//    its vector memory path: 2.6 k cycles against 2.0 k of MFMAs, so the stream, not the issue order, sets the pace;
//    the tile order above keeps the stream in L2 (worth 8 %), the deeper ring is worth 1-6 %. K = 8192 also pays for
//    its power-of-two row stride (all rows of a piece in one memory channel): K = 8320 runs at 1.18 PFLOP/s, 1.05 at 8192.
//  * One tile per workgroup. A persistent form that sends the next tile's first stages off before the epilogue was 7 %
//    slower at K = 1024: loads and stores share vmcnt and do not keep each other's order, so the next tile's first
//    wait also waits for the epilogue's stores, and tiles dealt out statically lose the hardware's load balancing.
// -------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) void g256_glb;
typedef __attribute__((address_space(3))) void g256_lds;
constexpr int G256_STAGE = 32768;   // four stages of BK = 32 (256 weight rows and 256 activation rows x 64 B)
// The same kernel on a 128 x 128 tile (T = 2: waves of 64 x 64) with an EIGHT-stage ring (16 KB per stage, seven
// stages = 224 k-values in flight) serves the products with few tiles and a long K (the training step's N = 384,
// K = 1152 / 1536 products: 240 tiles for 256 CUs). There the register-staged 128 x 128 kernel above has one k-step
// in flight and pays a memory latency per step (0.65 us per 64 k-values, against 0.2 us of MFMAs).
constexpr int G128_STAGES = 8, G128_STAGE = 16384;

template <int NP, int MAXL>
__device__ __forceinline__ void g256_stage_ready(int later_stages) {
  // this wave's pieces of the stage have landed, `later_stages` younger stages (NP pieces each) may stay in flight
  if (MAXL >= 6 && later_stages >= 6) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(6 * NP) : "memory");
  else if (MAXL >= 5 && later_stages == 5) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(5 * NP) : "memory");
  else if (MAXL >= 4 && later_stages == 4) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * NP) : "memory");
  else if (MAXL >= 3 && later_stages == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * NP) : "memory");
  else if (later_stages >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NP) : "memory");
  else if (later_stages == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NP) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
}

// T = 32-row operand tiles per wave and side (4: the 256 x 256 tile described above, 2: 128 x 128), NS = ring stages
// NL > 0: NL more waves (4 or 8, T = 2 only) that do nothing but send the LDS-DMA pieces off. A wave moves one 1-KiB
// piece per 165 clocks at best, however many it keeps in flight (tools_prof/micro/stream_rate.hip), and the four waves
// of a 128 x 128 tile need four pieces each per 8 MFMAs = 256 clocks of matrix pipe: as their own loaders they run
// at a quarter of the MFMA rate. The loader waves take the pieces over (two or one per stage and wave) and meet the
// multiplying waves at the one barrier per step.
template <int EPI, bool DROP, int T, int NS, int NL = 0>
__global__ __launch_bounds__(256 + 64 * NL) __attribute__((amdgpu_waves_per_eu(1, T == 4 ? 1 : 2 + NL / 8))) void gemm256_kernel(GemmArgs p) {
  constexpr int TILE = 64 * T;                 // rows of either operand per workgroup
  constexpr int PART = TILE * 64;              // bytes of one operand's rows in a stage (BK = 32)
  constexpr int STAGE = 2 * PART;
  constexpr int PL = NL ? (TILE / 16) / NL : T;  // pieces of either part that a piece-moving wave owns
  constexpr int NP = 2 * PL;                   // pieces such a wave moves per stage (PL weight + PL activation)
  constexpr int NM = T * T;                    // MFMAs per k16 index and wave
  static_assert(NL == 0 || (T == 2 && (NL == 4 || NL == 8)), "loader waves: 128 x 128 tile, four or eight of them");
  static_assert((NS & (NS - 1)) == 0 && NS >= 4 && NS <= 8, "ring depth: a power of two, 4 .. 8 (buffer index by mask, vmcnt <= 63)");
  static_assert(NP <= NM, "the operand reads and the pieces of a step are placed one per MFMA");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem256[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) & 1, wn = wave & 1;
  const bool loader = NL > 0 && wave >= 4;     // (wave-uniform)
  const int ow = NL ? wave - 4 : wave;         // index among the waves that move pieces
  const int kh = lane >> 5, r = lane & 31;
  if (EPI == EPI_AXPBY_BF16 || EPI == EPI_BIAS_BF16) {
    const long long b = blockIdx.y;
    p.A += b * p.sA;
    p.W += b * p.sW;
    p.out = (unsigned short*)p.out + b * p.sOut;
    if (EPI == EPI_AXPBY_BF16) p.aux_in = (const unsigned short*)p.aux_in + b * p.sOut;
  }
  int tile;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int q = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
    tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
  }
  // Tiles in groups of 8 tile rows, column by column inside a group: the 32 workgroups that an XCD runs at a time
  // (consecutive logical tiles) then cover 8 x 4 tiles and share 12 operand panels in their L2 instead of the 33
  // of a 1 x 32 strip.
  const int tiles_n = (p.Npad + TILE - 1) / TILE, tiles_m = (p.M + TILE - 1) / TILE;
  const int per_group = 8 * tiles_n, group = tile / per_group;
  const int rows_here = min(tiles_m - 8 * group, 8);
  const int in_group = tile - group * per_group;
  const int m0 = (8 * group + in_group % rows_here) * TILE, n0 = (in_group / rows_here) * TILE;
  const int K = p.K, nk = K / 32;

  // NS stages of BK = 32 (TILE weight rows and TILE activation rows x 64 B each), NS - 2 to NS - 1 of them in flight
  // at any time: with two stages of BK = 64 every k-step began with nothing in flight (the whole next
  // stage is issued and awaited inside one step), and the CU's ingest path idled while the queue refilled.
  // DMA: a 1-KiB piece = 16 operand rows x 64 B (one 64-byte memory request per row). Lane i lands at byte 16 i of
  // the piece: row i / 4, 16-byte slot i % 4, and fetches the row's k-chunk (slot ^ ((row >> 2) & 3)), so that the
  // MFMA operand reads below (32 rows x one k-chunk) cover every bank once per 16 lanes. LDS row rho of a 32-row
  // weight tile holds weight row n0 + 32 tile + nperm(rho). Wave w moves pieces T w .. T w + T - 1 of either part.
  // A piece costs three instructions: M0 = LDS destination, then the load with a scalar base (the tile's rows at
  // this k-step) and one 32-bit lane offset per piece number. (The builtin with a per-lane 64-bit pointer makes the
  // compiler emit 64-bit vector adds per piece: 57 cycles of issue per piece instead of 20, on a wave whose SIMD has
  // no other wave to run.)
  unsigned voffw[PL], voffa[PL];
#pragma unroll
  for (int t = 0; t < PL; ++t) {
    const int row = (16 * (PL * ow + t) + (lane >> 2)) & (TILE - 1);   // 0 .. TILE - 1 inside the part
    const int rho = row & 31, tl = row >> 5;
    const int chunk = (lane & 3) ^ ((rho >> 2) & 3);
    const int np = 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
    // Npad is a multiple of 128: the last 256-column tile may be half empty; its missing weight rows read the
    // last row instead (finite values, results never stored)
    const int wr = min(32 * tl + np, p.Npad - 1 - n0);
    voffw[t] = (unsigned)((wr * K + 8 * chunk) * 2);
    int am = m0 + row;
    am = am < p.M ? am : p.M - 1;
    voffa[t] = (unsigned)(((am - m0) * K + 8 * chunk) * 2);
    asm volatile("" : "+v"(voffw[t]), "+v"(voffa[t]));
  }
  const unsigned char* const wtile = (const unsigned char*)(p.W + (size_t)n0 * K);
  const unsigned char* const atile = (const unsigned char*)(p.A + (size_t)m0 * K);
  const unsigned ldsw = (unsigned)(size_t)(g256_lds*)smem256 + (ow & (TILE / 16 / PL - 1)) * (PL * 1024);
  // piece i of stage s: i < PL weight rows, else activation rows
  auto piece = [&](int s, int i) {
    const unsigned st = ldsw + (s & (NS - 1)) * STAGE;
    const unsigned char* src = (i < PL ? wtile : atile) + (size_t)s * 64;
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
                 ::"s"(st), "i"((i % PL) * 1024 + (i < PL ? 0 : PART)), "v"(i < PL ? voffw[i % PL] : voffa[i % PL]), "s"(src)
                 : "memory", "m0", "scc");
  };
  if (loader) {
    // the loader waves' whole life: the ring's stages sent off one per step, each wait and barrier where the
    // multiplying waves have theirs
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
      if (st < nk) {
#pragma unroll
        for (int i = 0; i < NP; ++i) piece(st, i);
      }
    g256_stage_ready<NP, NS - 2>(min(nk - 1, NS - 2));
    for (int s = 0; s < nk; ++s) {
      const bool next = s + 1 < nk, issue = s + NS - 1 < nk;
      if (next) g256_stage_ready<NP, NS - 3>(min(nk - 2 - s, NS - 3));
      if (issue) {
#pragma unroll
        for (int i = 0; i < NP; ++i) piece(s + NS - 1, i);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  f32x16 acc[T][T];
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      if (T == 4) asm volatile("" : "+a"(acc[i][j]));
    }

  // operand addresses of this wave inside a stage: W tiles T wn + j, A tiles T wm + j (2 KiB each: 32 rows x 64 B);
  // lane (r, kh) reads row r, k-chunk 2 q + kh (q = 0, 1), which sits in slot (2 q + kh) ^ ((r >> 2) & 3)
  const unsigned lds0 = (unsigned)(size_t)(g256_lds*)smem256 + r * 64;
  const unsigned wbase = lds0 + (T * wn) * 2048, abase = lds0 + PART + (T * wm) * 2048;
  unsigned qoff[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) qoff[q] = (unsigned)(((2 * q + kh) ^ ((r >> 2) & 3)) * 16);

  bf16x8 fw[2][T], fa[2][T];
  // one operand read of stage s: i < T weight tile i, else activation tile i - T, k16 index q, into register set q
  auto read1 = [&](int s, int q, int i) {
    const unsigned so = (s & (NS - 1)) * STAGE;
    if (i < T)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fw[q][i % T]) : "v"(wbase + so + qoff[q]), "i"((i % T) * 2048));
    else
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[q][i % T]) : "v"(abase + so + qoff[q]), "i"((i % T) * 2048));
  };
  auto wait_reads = [&](int q) {      // every register written by the reads of set q passes through this wait
    if constexpr (T == 4)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fw[q][0]), "+v"(fw[q][1]), "+v"(fw[q][2]), "+v"(fw[q][3]), "+v"(fa[q][0]), "+v"(fa[q][1]),
                     "+v"(fa[q][2]), "+v"(fa[q][3]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fw[q][0]), "+v"(fw[q][1]), "+v"(fa[q][0]), "+v"(fa[q][1]));
  };

  // prologue: stages 0 .. NS - 2 on their way; stage 0 landed; its first operands read
  if (NL == 0) {
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
      if (st < nk) {
#pragma unroll
        for (int i = 0; i < NP; ++i) piece(st, i);
      }
    g256_stage_ready<NP, NS - 2>(min(nk - 1, NS - 2));
  } else {
    __builtin_amdgcn_s_barrier();              // the loaders' stage 0 has landed
  }
#pragma unroll
  for (int i = 0; i < 2 * T; ++i) read1(0, 0, i);

  // Step s: MFMAs of k16 index 0 with the reads of index 1 between them; then stage s + 1 must have landed (barrier in
  // the MIDDLE of the step, so that its first reads can go out between the MFMAs of index 1 and no step begins with
  // an exposed LDS latency); then the MFMAs of index 1 with those reads and the pieces of stage s + NS - 1 between them
  // (its buffer held stage s - 1, which every wave has left behind when it reaches this step's barrier).
  // One wave per SIMD issues in order: whatever is not placed between two MFMAs adds to the step.
  for (int s = 0; s < nk; ++s) {
    wait_reads(0);
#pragma unroll
    for (int jn = 0; jn < T; ++jn)
#pragma unroll
      for (int jm = 0; jm < T; ++jm) {
        const int i = T * jn + jm;
        acc[jn][jm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[0][jn], fa[0][jm], acc[jn][jm], 0, 0, 0);
        if (i < 2 * T) read1(s, 1, i);
        __builtin_amdgcn_sched_barrier(0);
      }
    const bool next = s + 1 < nk, issue = s + NS - 1 < nk;
    if (next) {
      if (NL == 0) g256_stage_ready<NP, NS - 3>(min(nk - 2 - s, NS - 3));      // younger stages may still be in flight
      else __builtin_amdgcn_s_barrier();
    }
    wait_reads(1);
#pragma unroll
    for (int jn = 0; jn < T; ++jn)
#pragma unroll
      for (int jm = 0; jm < T; ++jm) {
        const int i = T * jn + jm;
        acc[jn][jm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[1][jn], fa[1][jm], acc[jn][jm], 0, 0, 0);
        if (i < 2 * T && next) read1(s + 1, 0, i);
        if (NL == 0 && i >= NM - NP && issue) piece(s + NS - 1, i - (NM - NP));
        __builtin_amdgcn_sched_barrier(0);
      }
  }

  // ---- epilogue: tile (jn, jm): lane holds row m = m0 + 32 T wm + 32 jm + r, columns nb .. nb + 15
#pragma unroll
  for (int jn = 0; jn < T; ++jn) {
    const int nb = n0 + 32 * T * wn + 32 * jn + 16 * kh;
    if (nb >= p.Npad) continue;                 // the empty half of a ragged last column tile
    float bv[16];
    load_bias16<EPI>(p, nb, bv);
#pragma unroll
    for (int jm = 0; jm < T; ++jm) {
      const int m = m0 + 32 * T * wm + 32 * jm + r;
      if (m >= p.M) continue;
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = acc[jn][jm][e] + bv[e];
      epilogue_row<EPI, DROP>(p, m, nb, v);
    }
  }
}

// -------------------------------------------------------------------------------------------------
// The same 256 x 256 tile with EIGHT waves, two per SIMD, that take turns (ping-pong): waves 0-3 hold the tile's upper
// 128 activation rows, waves 4-7 the lower 128 (a wave = 128 rows x 64 columns: 8 x 4 accumulators of
// v_mfma_f32_16x16x32_bf16, 128 registers), and through a tile's K-loop the second group runs one barrier behind the
// first, so that on every SIMD one wave issues its 16 MFMAs of a phase while its partner reads operands from LDS and
// sends LDS-DMA pieces off. The kernel is persistent: a workgroup walks several output tiles (see the tile list in the
// kernel), and the stage stream below runs on from a tile's last K-tiles into the next tile's first ones.
// A K-tile (64 k-values) is four phases, one per 64 x 32 quadrant of the wave's output:
//   P1 reads X0 (2 weight tiles) + Y0 (4 activation tiles), P2 reads X1, P3 reads Y1, P4 reads nothing
//   (quadrants (Y0,X0), (Y0,X1), (Y1,X1), (Y1,X0); every fragment is read from LDS once per K-tile).
// LDS: two buffers of 64 KB = [Ylo | Yhi | Xlo | Xhi], 128 rows x 128 B each (chunk c of row r at slot c ^ ((r >> 1) & 7):
// the 16 rows x one k-chunk of an operand read touch every bank once). The buffer of K-tile u is refilled for tile u + 2
// region by region as soon as no wave reads the region any more, one 16-KB stage per phase (2 pieces per thread):
//   tile u: P1 -> weights' upper half of tile u + 1, P2 -> activation rows 64-127 (of both halves) of u + 1,
//           P3 -> activation rows 0-63 of u + 2, P4 -> weights' lower half of u + 2,
// (a tile's first phase issues nothing and its last phase two stages: the next tile begins with a K-tile and a half
// landed, see the prologue), i.e. every stage is issued 4 to 6 phases before its first read, and 3 to 4 stages are in
// flight at any time. A wave
// waits for its own pieces only (vmcnt(6) in P4 for what P1 of the next K-tile reads, vmcnt(8) in P2 for what P3 reads,
// each with exactly that many younger pieces behind it: past the workgroup's last tile the stages are still issued,
// re-reading that tile's first bytes into regions nobody reads, so that the counts hold to the end), and every read of
// a region comes behind a barrier that all waves pass after that wait. Weight rows are permuted inside the 64-row sub-tiles so that a
// lane's 16 accumulators of an activation row are 16 consecutive output columns (epilogue_row as in the kernels above).
constexpr int G8_BUF = 65536;

template <int MH, int XH>
__device__ __forceinline__ void g8_quadrant(f32x4 (&acc)[8][4], const bf16x8 (&X)[2][2], const bf16x8 (&Y)[4][2]) {
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int sx = 0; sx < 2; ++sx)
        acc[4 * MH + t][2 * XH + sx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X[sx][kk], Y[t][kk], acc[4 * MH + t][2 * XH + sx], 0, 0, 0);
}

template <int EPI, bool DROP>
__global__ __launch_bounds__(512) void gemm8p_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem8p[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, c = lane & 15;
  if (EPI == EPI_AXPBY_BF16 || EPI == EPI_BIAS_BF16) {
    const long long b = blockIdx.y;
    p.A += b * p.sA;
    p.W += b * p.sW;
    p.out = (unsigned short*)p.out + b * p.sOut;
    if (EPI == EPI_AXPBY_BF16) p.aux_in = (const unsigned short*)p.aux_in + b * p.sOut;
  }
  // The workgroup's tiles: the logical tile list is cut into 8 contiguous chunks, one per group of workgroups that
  // share an XCD (blockIdx & 7); inside a chunk workgroup j of the gx on that XCD takes tiles j, j + gx, j + 2 gx, ...
  // (a grid of one workgroup per tile gives every workgroup exactly one). Logical tiles run in groups of 8 tile rows,
  // column by column inside a group, so that the tiles an XCD works on at a time share operand panels in its L2.
  const int tiles_n = (p.Npad + 255) / 256, tiles_m = (p.M + 255) / 256;
  int tile, tile_end, tile_step;
  {
    const int n_all = tiles_m * tiles_n, nwg = gridDim.x, xcd = blockIdx.x & 7;
    const int q = n_all >> 3, rr = n_all & 7;
    const int chunk0 = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    tile_end = chunk0 + q + (xcd < rr ? 1 : 0);
    tile_step = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);
    tile = chunk0 + (blockIdx.x >> 3);
  }
  const int K = p.K, nt = K >> 6;
  const int per_group = 8 * tiles_n;

  // ---- LDS-DMA: stage kinds 0 = activation rows 0-63 of both halves, 1 = weights' lower half, 2 = weights' upper half,
  //      3 = activation rows 64-127 of both halves; thread's pieces q = 2 wave + j (j = 0, 1) of the stage's sixteen.
  //      Source of a tile: base pointers of its activation and weight rows and the thread's byte offsets per stage kind.
  struct Src {
    const unsigned char *a, *w;
    unsigned v[4][2];
    int m0, n0;
  };
  auto make_src = [&](int t, Src& s) {
    const int group = t / per_group, rows_here = min(tiles_m - 8 * group, 8), in_group = t - group * per_group;
    const int m0 = (8 * group + in_group % rows_here) * 256, n0 = (in_group / rows_here) * 256;
    s.m0 = m0;
    s.n0 = n0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = 2 * wave + j, rr = lane >> 3, slot = lane & 7;
      const int rho = 8 * (q & 7) + rr;                          // row inside the 64-row part of activation half q >> 3 (= wr)
      const int cha = slot ^ ((rho >> 1) & 7);
      const int am0 = min(m0 + 128 * (q >> 3) + rho, p.M - 1) - m0, am1 = min(m0 + 128 * (q >> 3) + 64 + rho, p.M - 1) - m0;
      s.v[0][j] = (unsigned)((am0 * K + 8 * cha) * 2);
      s.v[3][j] = (unsigned)((am1 * K + 8 * cha) * 2);
      const int rx = 8 * q + rr;                                 // LDS row of the 128-row weight half
      const int i = rx & 15, tn = (rx >> 4) & 3;
      const int nrel = 64 * (rx >> 6) + 16 * (i >> 2) + 4 * tn + (i & 3);
      const int chx = slot ^ ((rx >> 1) & 7);
      const int last = p.Npad - 1 - n0;                          // a half-empty last column tile re-reads the last weight row
      s.v[1][j] = (unsigned)((min(nrel, last) * K + 8 * chx) * 2);
      s.v[2][j] = (unsigned)((min(128 + nrel, last) * K + 8 * chx) * 2);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(s.v[k][0]), "+v"(s.v[k][1]));
    s.w = (const unsigned char*)(p.W + (size_t)n0 * K);
    s.a = (const unsigned char*)(p.A + (size_t)m0 * K);
  };
  Src cur, nxt;                         // the tile being multiplied and the one after it (the stage stream runs two K-tiles ahead)
  make_src(tile, cur);
  bool more = tile + tile_step < tile_end;
  if (more) make_src(tile + tile_step, nxt); else nxt = cur;      // no next tile: the stages past the end re-read this one's first bytes
  int kbase = 0;                        // K-tiles of earlier tiles: buffer parity runs on through the tile sequence

  const unsigned lds0 = (unsigned)(size_t)(g256_lds*)smem8p;
  // LDS destinations of the thread's two pieces per stage kind (wave-uniform)
  const unsigned dstY = lds0 + wr * 16384 + (2 * wc) * 1024, dstX = lds0 + 32768 + (2 * wave) * 1024;
  auto glds = [&](const unsigned char* src, unsigned vo, unsigned dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(vo), "s"(src) : "memory", "m0");
  };
  // stage `kind` of K-tile kt of the current tile; kt >= nt: K-tile kt - nt of the next one (nt >= 2, so kt - nt <= 1)
  auto issue = [&](int kind, int kt) {
    const unsigned bo = (unsigned)((kbase + kt) & 1) * G8_BUF;
    const bool over = kt >= nt;
    const size_t ko = (size_t)(over ? kt - nt : kt) * 128;
    const bool act = kind == 0 || kind == 3;
    const unsigned char* src = (over ? (act ? nxt.a : nxt.w) : (act ? cur.a : cur.w)) + ko;
    const unsigned dst = kind == 0 ? dstY + bo : kind == 3 ? dstY + 8192 + bo : kind == 1 ? dstX + bo : dstX + 16384 + bo;
    glds(src, over ? nxt.v[kind][0] : cur.v[kind][0], dst);
    glds(src, over ? nxt.v[kind][1] : cur.v[kind][1], dst + 1024);
  };

  // ---- operand reads: lane (g, c) reads row 16 tile + c, k-chunk 4 kk + g, which sits in slot (4 kk + g) ^ (c >> 1)
  const unsigned sl0 = (unsigned)((g ^ (c >> 1)) * 16), sl1 = (unsigned)(((4 + g) ^ (c >> 1)) * 16);
  const unsigned xrow = lds0 + 32768 + (wc >> 1) * 16384 + (64 * (wc & 1) + c) * 128;
  const unsigned yrow = lds0 + wr * 16384 + c * 128;
  bf16x8 X0[2][2], X1[2][2], Y[4][2];
  f32x4 acc[8][4];

  // ---- prologue: K-tile 0 whole and the first two stages of K-tile 1; K-tile 0's first three stages landed
  // A tile begins with its K-tile 0 whole and K-tile 1 but for the activation rows 64-127 landed (here: one drain; for the
  // tiles that follow: the drain in front of the previous tile's stores), so that its first K-tile and a half run
  // without a counted wait: such a wait would also wait for the previous tile's stores, which are slow to be
  // acknowledged when every CU writes its tile at once.
  issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0); issue(0, 1); issue(1, 1); issue(2, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

#define G8_RDX(dst, tn, kk, base)                                                                                  \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"((base) + ((kk) ? sl1 : sl0)), "n"((tn) * 2048) : "memory")
#define G8_RDY(dst, tm, kk, base)                                                                                  \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"((base) + ((kk) ? sl1 : sl0)), "n"((tm) * 2048) : "memory")

  for (;;) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (wr == 1) __builtin_amdgcn_s_barrier();          // through a tile's K-loop the lower group runs one barrier behind
    for (int u = 0; u < nt; ++u) {
      const unsigned bo = (unsigned)((kbase + u) & 1) * G8_BUF;
      const unsigned xb = xrow + bo, yb = yrow + bo;
      // ---- P1: X0, Y0 -> quadrant (Y0, X0)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        G8_RDX(X0[0][kk], 0, kk, xb); G8_RDX(X0[1][kk], 1, kk, xb);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        G8_RDY(Y[0][kk], 0, kk, yb); G8_RDY(Y[1][kk], 1, kk, yb); G8_RDY(Y[2][kk], 2, kk, yb); G8_RDY(Y[3][kk], 3, kk, yb);
      }
      if (u > 0) issue(2, u + 1);                          // (K-tile 1's came with the previous tile's last phase)
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(X0[0][0]), "+v"(X0[0][1]), "+v"(X0[1][0]), "+v"(X0[1][1]), "+v"(Y[0][0]), "+v"(Y[0][1]), "+v"(Y[1][0]),
                     "+v"(Y[1][1]), "+v"(Y[2][0]), "+v"(Y[2][1]), "+v"(Y[3][0]), "+v"(Y[3][1]));
      __builtin_amdgcn_s_setprio(1);
      g8_quadrant<0, 0>(acc, X0, Y);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      // ---- P2: X1 -> quadrant (Y0, X1)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        G8_RDX(X1[0][kk], 2, kk, xb); G8_RDX(X1[1][kk], 3, kk, xb);
      }
      issue(3, u + 1);
      // own pieces of rows 64-127 of THIS K-tile (read in P3) landed; K-tile 0's were there before the tile began
      if (u > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X1[0][0]), "+v"(X1[0][1]), "+v"(X1[1][0]), "+v"(X1[1][1]));
      __builtin_amdgcn_s_setprio(1);
      g8_quadrant<0, 1>(acc, X1, Y);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      // ---- P3: Y1 -> quadrant (Y1, X1)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        G8_RDY(Y[0][kk], 4, kk, yb); G8_RDY(Y[1][kk], 5, kk, yb); G8_RDY(Y[2][kk], 6, kk, yb); G8_RDY(Y[3][kk], 7, kk, yb);
      }
      issue(0, u + 2);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(Y[0][0]), "+v"(Y[0][1]), "+v"(Y[1][0]), "+v"(Y[1][1]), "+v"(Y[2][0]), "+v"(Y[2][1]), "+v"(Y[3][0]),
                     "+v"(Y[3][1]));
      __builtin_amdgcn_s_setprio(1);
      g8_quadrant<1, 1>(acc, X1, Y);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      // ---- P4: quadrant (Y1, X0); what P1 of the next K-tile reads has landed when this wait returns
      issue(1, u + 2);
      if (u == nt - 1) {
        // the tile's last K-tile: the weights' upper half of the NEXT tile's K-tile 1 goes out as well (its region was
        // last read two phases ago), and the wait leaves that stage in flight too
        issue(2, u + 2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if (u > 0) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
      g8_quadrant<1, 0>(acc, X0, Y);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
    }

    // ---- the tile's output: accumulator (tm, tn), register j = row m0 + 128 wr + 16 tm + c, column n0 + 64 wc + 16 g + 4 tn + j.
    // The upper group waits for the barrier the lower one is behind, so that both groups write their halves at the same
    // time (one after the other, each would wait for the other's memory latencies). The next tile's first stages are in
    // flight or landed meanwhile. The loads and stores here are older than every later stage piece, and loads complete in
    // order: the counted waits of the K-loop stay sufficient (a store still on its way can only make them wait longer).
    if (wr == 0) __builtin_amdgcn_s_barrier();
    {
      const int nb = cur.n0 + 64 * wc + 16 * g;
      float bv[16];
      load_bias16<EPI>(p, nb < p.Npad ? nb : 0, bv);      // (on its way while the wait below drains the stage pieces)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every stage piece issued so far has landed before the first store goes out
      if (nb < p.Npad) {
        const bool ln_in = (EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_GELU_BF16) && p.ln_stats != nullptr;
        const bool ln_out = EPI == EPI_BIAS_RESID_F32 && !DROP && p.xb_out != nullptr;
        float cs[16];
        f32x2 rowst[8];       // (mean, 1 / std) of the lane's eight rows, all requested at once
        if (ln_in) {
#pragma unroll
          for (int j4 = 0; j4 < 4; ++j4) {
            const f32x4 t = *(const f32x4*)(p.ln_colsum + nb + 4 * j4);
            cs[4 * j4 + 0] = t[0]; cs[4 * j4 + 1] = t[1]; cs[4 * j4 + 2] = t[2]; cs[4 * j4 + 3] = t[3];
          }
#pragma unroll
          for (int tm = 0; tm < 8; ++tm) {
            const int m = cur.m0 + 128 * wr + 16 * tm + c;
            rowst[tm] = *(const f32x2*)(p.ln_stats + 2 * (size_t)(m < p.M ? m : p.M - 1));
          }
        }
        // The residual epilogue reads the rows it updates: the loads of four row groups go out together ahead of their
        // uses (one memory latency per four groups; one group at a time, the eight latencies in a row were most of a
        // tile's output time). Rows past M load the last row instead (no branch around the loads) and are dropped below.
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          f32x4 res[4][4];
          f32x2 sums[4];
          if (EPI == EPI_BIAS_RESID_F32) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int m = cur.m0 + 128 * wr + 16 * (4 * half + t) + c;
              const float* ri = (p.aux_in ? (const float*)p.aux_in : (const float*)p.out) + (size_t)(m < p.M ? m : p.M - 1) * p.ldc + nb;
#pragma unroll
              for (int j4 = 0; j4 < 4; ++j4) res[t][j4] = *(const f32x4*)(ri + 4 * j4);
            }
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int tm = 4 * half + t;
            const int m = cur.m0 + 128 * wr + 16 * tm + c;
            if (m >= p.M) continue;
            float v[16];
            if (ln_in) {
              // A held the raw rows x (bf16), the weights carry gamma: LN(x) W^T = rstd (x W^T - mean colsum(W))
              const float mean = rowst[tm][0], rstd = rowst[tm][1];
#pragma unroll
              for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                  v[4 * tn + j] = rstd * (acc[tm][tn][j] - mean * cs[4 * tn + j]) + bv[4 * tn + j];
            } else {
#pragma unroll
              for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * tn + j] = acc[tm][tn][j] + bv[4 * tn + j];
            }
            if (EPI == EPI_BIAS_RESID_F32) {
              // the updated residual row segment in fp32; with ln_out also its bf16 copy (the next projection's operand)
              // and its share of the row's LayerNorm sums: the four lanes of a row (g = 0 .. 3) hold the wave's 64 columns
              if (DROP) {
                float f[16];
                drop_factors16(p.drop, m, nb, f);
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] *= f[j];
              }
              float s1 = 0.f, s2 = 0.f;
#pragma unroll
              for (int j4 = 0; j4 < 4; ++j4) {
                f32x4 x = res[t][j4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  x[e] += v[4 * j4 + e];
                  v[4 * j4 + e] = x[e];
                  s1 += x[e];
                  s2 += x[e] * x[e];
                }
                res[t][j4] = x;           // stored below, behind the last use of the four groups' loads
              }
              if (ln_out) {
                s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                sums[t][0] = s1; sums[t][1] = s2;
              }
            } else {
              epilogue_row<EPI, DROP>(p, m, nb, v);
            }
          }
          // Loads and stores share one counter: a store between two groups' uses would make the second group's wait for
          // its loads a wait for that store's acknowledgement as well. All four groups' rows go out here, in one run.
          if (EPI == EPI_BIAS_RESID_F32) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int m = cur.m0 + 128 * wr + 16 * (4 * half + t) + c;
              if (m >= p.M) continue;
              float* o = (float*)p.out + (size_t)m * p.ldc + nb;
#pragma unroll
              for (int j4 = 0; j4 < 4; ++j4) *(f32x4*)(o + 4 * j4) = res[t][j4];
              if (ln_out) {
                unsigned short* ob = p.xb_out + (size_t)m * p.ldc + nb;
                u32x4 lo, hi;
                lo[0] = bd_pack_bf16x2(res[t][0][0], res[t][0][1]); lo[1] = bd_pack_bf16x2(res[t][0][2], res[t][0][3]);
                lo[2] = bd_pack_bf16x2(res[t][1][0], res[t][1][1]); lo[3] = bd_pack_bf16x2(res[t][1][2], res[t][1][3]);
                hi[0] = bd_pack_bf16x2(res[t][2][0], res[t][2][1]); hi[1] = bd_pack_bf16x2(res[t][2][2], res[t][2][3]);
                hi[2] = bd_pack_bf16x2(res[t][3][0], res[t][3][1]); hi[3] = bd_pack_bf16x2(res[t][3][2], res[t][3][3]);
                *(u32x4*)(ob) = lo;
                *(u32x4*)(ob + 8) = hi;
                if (g == 0) *(f32x2*)(p.part_out + ((size_t)((cur.n0 >> 6) + wc) * p.M + m) * 2) = sums[t];
              }
            }
          }
        }
      }
    }
    if (!more) break;
    kbase += nt;
    tile += tile_step;
    cur = nxt;
    more = tile + tile_step < tile_end;
    if (more) make_src(tile + tile_step, nxt);
  }
#undef G8_RDX
#undef G8_RDY
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stages past the last tile: no piece may land after the workgroup has gone
}

// Persistent grid: two workgroups per CU (what the kernel's registers and 64 KB of LDS allow), a
// multiple of 8 so that logical tile ids keep their XCD; fewer tiles than that: one workgroup each.
int persistent_grid(int n_tiles) {
  static int slots = 0;
  if (!slots) {
    int dev = 0, ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    slots = 2 * ncu / 8 * 8;
    if (const char* e = getenv("BD_GEMM_SLOTS")) slots = atoi(e) > 0 ? atoi(e) / 8 * 8 : slots;
    if (slots < 8) slots = 8;
  }
  return n_tiles <= slots ? n_tiles : slots;
}

// the large-tile kernel pays when the problem fills the chip with 256 x 256 tiles
bool use_gemm256(const GemmArgs& a) {
  static int mode = getenv("BD_GEMM256") ? atoi(getenv("BD_GEMM256")) : 1;
  if (!mode || a.n_real != a.Npad) return false;
  // widths that are odd multiples of 128 (the reference's "large" size: 1152, 3456) run with a half-empty last
  // column tile once that costs little (an eighth of the columns at most)
  if (a.Npad % 256 != 0 && a.Npad < 1024) return false;
  if (a.K < 768) return false;        // few k-steps: the 128 x 128 kernel with two workgroups per CU hides its epilogue better
  return (((long long)a.M + 255) / 256) * ((a.Npad + 255) / 256) >= 128;
}
// workgroups of the eight-wave kernel: one per CU (128 KB of LDS each), a multiple of 8 so that the workgroups of an
// XCD split the tile list evenly; more tiles than that: every workgroup walks through several (BD_GEMM8P_SLOTS=0: one
// workgroup per tile)
int gemm8p_slots() {
  int dev = 0, ncu = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
  int slots = ncu / 8 * 8;
  if (const char* e = getenv("BD_GEMM8P_SLOTS")) slots = atoi(e) > 0 ? atoi(e) / 8 * 8 : 0x7fffffff;
  return slots < 8 ? 8 : slots;
}
// the eight-wave form of the large tile: same shapes as use_gemm256, K a multiple of 64 (BD_GEMM8P=0: off, 2: every
// shape that fits, for tests)
bool use_gemm8p(const GemmArgs& a) {
  static int mode = getenv("BD_GEMM8P") ? atoi(getenv("BD_GEMM8P")) : 1;
  if (!mode || a.n_real != a.Npad || a.K % 64 != 0 || a.K < 128) return false;
  if (mode == 2) return true;
  if (use_gemm256(a)) return true;
  // wide outputs over a short K (the feed-forward projections of the small model's training step, 10 k rows): one round
  // of 256 x 256 tiles beats two and a half rounds of 128 x 128 ones by 8-10 %
  static int wide = getenv("BD_GEMM8P_WIDE") ? atoi(getenv("BD_GEMM8P_WIDE")) : 1;
  const long long tiles = (((long long)a.M + 255) / 256) * ((a.Npad + 255) / 256);
  return wide && a.K >= 384 && a.Npad >= 1536 && a.Npad % 256 == 0 && tiles >= 192;
}
// the deep-ring 128 x 128 kernel: at most one tile per CU (ahead there, 20-40 % slower beyond)
bool use_gemm128_ring(const GemmArgs& a) {
  static int mode = getenv("BD_GEMM128_RING") ? atoi(getenv("BD_GEMM128_RING")) : 1;
  if (!mode || a.n_real != a.Npad || a.K % 32 != 0) return false;
  if (mode == 2) return true;         // every shape (measurements)
  // one workgroup per CU: with more tiles than CUs the two-per-CU kernel wins. With its loader waves the ring kernel is
  // ahead from K = 384 on (10 240 x 384 x 384: 9.0 against 9.6 us, K = 768: 12.6 against 14.6)
  static int min_k = getenv("BD_GEMM128_RING_MINK") ? atoi(getenv("BD_GEMM128_RING_MINK")) : 384;
  return a.K >= min_k && a.n_tiles <= 256;
}

template <int EPI, bool DROP = false>
int launch(const GemmArgs& a, hipStream_t s) {
  void* tok = bd_prof_on() ? bd_prof_begin(s) : nullptr;
  if (use_gemm8p(a)) {
    const long long nt = (((long long)a.M + 255) / 256) * ((a.Npad + 255) / 256);
    static int slots = gemm8p_slots();
    BD_SET_MAX_LDS((gemm8p_kernel<EPI, DROP>), 2 * G8_BUF);
    hipLaunchKernelGGL((gemm8p_kernel<EPI, DROP>), dim3((unsigned)(nt < slots ? nt : slots)), dim3(512), 2 * G8_BUF, s, a);
  } else if (use_gemm256(a)) {
    const long long nt = (((long long)a.M + 255) / 256) * ((a.Npad + 255) / 256);
    BD_SET_MAX_LDS((gemm256_kernel<EPI, DROP, 4, 4>), 4 * G256_STAGE);
    hipLaunchKernelGGL((gemm256_kernel<EPI, DROP, 4, 4>), dim3((unsigned)nt), dim3(256), 4 * G256_STAGE, s, a);
  } else if (use_gemm128_ring(a)) {
    // four loader waves: 15-22 % faster than the four multiplying waves loading for themselves (N = 384, K = 1536 over
    // 10 240 rows: 19.8 against 24.2 us; N = 1536, K = 6144 over 2560 rows: 63 against 74); eight are no faster
    static int loaders = getenv("BD_GEMM_LOADERS") ? atoi(getenv("BD_GEMM_LOADERS")) : 4;
    if (loaders == 8) {
      BD_SET_MAX_LDS((gemm256_kernel<EPI, DROP, 2, G128_STAGES, 8>), G128_STAGES * G128_STAGE);
      hipLaunchKernelGGL((gemm256_kernel<EPI, DROP, 2, G128_STAGES, 8>), dim3((unsigned)a.n_tiles), dim3(768),
                         G128_STAGES * G128_STAGE, s, a);
    } else if (loaders == 4) {
      BD_SET_MAX_LDS((gemm256_kernel<EPI, DROP, 2, G128_STAGES, 4>), G128_STAGES * G128_STAGE);
      hipLaunchKernelGGL((gemm256_kernel<EPI, DROP, 2, G128_STAGES, 4>), dim3((unsigned)a.n_tiles), dim3(512),
                         G128_STAGES * G128_STAGE, s, a);
    } else {
      BD_SET_MAX_LDS((gemm256_kernel<EPI, DROP, 2, G128_STAGES>), G128_STAGES * G128_STAGE);
      hipLaunchKernelGGL((gemm256_kernel<EPI, DROP, 2, G128_STAGES>), dim3((unsigned)a.n_tiles), dim3(256),
                         G128_STAGES * G128_STAGE, s, a);
    }
  } else {
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, DROP>), dim3(persistent_grid(a.n_tiles)), dim3(256), 0, s, a);
  }
  if (tok) bd_prof_end(tok, s, 2.0 * (double)a.M * (double)a.n_real * (double)a.K);
  return bd_check_launch("bd_gemm_bf16");
}

}  // namespace

// epilogue: 0 bias->bf16, 1 bias+gelu->bf16, 2 fp32 out = resid + acc+bias, 3 fp32 out (ragged n),
//           4 bf16 out = (acc+bias) * gelu'(aux_in)
extern "C" int bd_gemm_bf16_drop(const void* A, const void* W, const float* bias, void* out,
                                 const void* aux_in, void* aux_out, int M, int n_pad, int n_real, int K,
                                 int ldc, int epilogue, const bd_dropout* drop, unsigned site, void* stream) {
  BD_REQUIRE(A && W && out, "bd_gemm_bf16: null operand");
  BD_REQUIRE((epilogue >= 0 && epilogue <= 4) || epilogue == EPI_GELU_GRAD_BF16 || epilogue == EPI_MUL_AUX_BF16,
             "bd_gemm_bf16: unknown epilogue %d", epilogue);
  BD_REQUIRE(M > 0 && K > 0 && n_pad > 0, "bd_gemm_bf16: empty shape M=%d N=%d K=%d", M, n_pad, K);
  BD_REQUIRE(n_pad % BN == 0, "bd_gemm_bf16: weight rows %d must be padded to a multiple of %d", n_pad, BN);
  BD_REQUIRE(K % BK == 0, "bd_gemm_bf16: K=%d must be a multiple of %d", K, BK);
  BD_REQUIRE(n_real > 0 && n_real <= n_pad, "bd_gemm_bf16: n_real=%d out of range", n_real);
  BD_REQUIRE(epilogue == EPI_F32 || (n_real == n_pad && (ldc % 8) == 0),
             "bd_gemm_bf16: vector epilogues need n_real == n_pad and ldc %% 8 == 0");
  BD_REQUIRE(ldc >= n_real, "bd_gemm_bf16: ldc=%d < n=%d", ldc, n_real);
  GemmArgs a;
  a.A = (const unsigned short*)A; a.W = (const unsigned short*)W; a.bias = bias; a.out = out;
  a.aux_in = aux_in; a.aux_out = aux_out;
  a.xb_out = nullptr; a.part_out = nullptr; a.ln_stats = nullptr; a.ln_colsum = nullptr;
  BD_REQUIRE(epilogue != EPI_DGELU_BF16 || aux_in != nullptr, "bd_gemm_bf16: epilogue 4 needs the pre-activation");
  BD_REQUIRE(epilogue != EPI_MUL_AUX_BF16 || aux_in != nullptr, "bd_gemm_bf16: epilogue 7 needs its multiplier");
  BD_REQUIRE(epilogue != EPI_GELU_GRAD_BF16 || aux_out != nullptr, "bd_gemm_bf16: epilogue 6 needs aux_out");
  a.M = M; a.Npad = n_pad; a.K = K; a.n_real = n_real; a.ldc = ldc;
  a.sA = a.sW = a.sOut = 0; a.alpha = 1.f; a.beta = 0.f;
  a.tiles_n = n_pad / BN;
  const long long tiles_m = ((long long)M + BM - 1) / BM;
  const long long nt = tiles_m * a.tiles_n;
  BD_REQUIRE(nt < (1ll << 31), "bd_gemm_bf16: too many tiles");
  a.n_tiles = (int)nt;
  hipStream_t s = (hipStream_t)stream;
  BD_REQUIRE(bd_drop_make(drop, site, &a.drop), "bd_gemm_bf16: dropout p=%g outside [0, 1)", (double)drop->p);
  if (a.drop.thr != 0u) {
    switch (epilogue) {
      case EPI_BIAS_GELU_BF16: return launch<EPI_BIAS_GELU_BF16, true>(a, s);
      case EPI_BIAS_RESID_F32: return launch<EPI_BIAS_RESID_F32, true>(a, s);
      case EPI_DGELU_BF16: return launch<EPI_DGELU_BF16, true>(a, s);
      case EPI_GELU_GRAD_BF16: return launch<EPI_GELU_GRAD_BF16, true>(a, s);
      default: return bd_fail(BD_ERR_ARG, "bd_gemm_bf16: epilogue %d has no dropout", epilogue);
    }
  }
  switch (epilogue) {
    case EPI_BIAS_BF16: return launch<EPI_BIAS_BF16>(a, s);
    case EPI_BIAS_GELU_BF16: return launch<EPI_BIAS_GELU_BF16>(a, s);
    case EPI_BIAS_RESID_F32: return launch<EPI_BIAS_RESID_F32>(a, s);
    case EPI_F32: return launch<EPI_F32>(a, s);
    case EPI_DGELU_BF16: return launch<EPI_DGELU_BF16>(a, s);
    case EPI_GELU_GRAD_BF16: return launch<EPI_GELU_GRAD_BF16>(a, s);
    case EPI_MUL_AUX_BF16: return launch<EPI_MUL_AUX_BF16>(a, s);
    default: return bd_fail(BD_ERR_ARG, "bd_gemm_bf16: unknown epilogue %d", epilogue);
  }
}

// ---- LayerNorm deferred across the residual projections (eight-wave kernel)
namespace {
// (mean, 1 / std) of every row from the per-block sums the residual projection left: part [n_blk][M][2]
__global__ __launch_bounds__(256) void ln_stats_finish_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                              int M, int n_blk, float inv_d, float eps) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < n_blk; ++b) {
    const f32x2 t = *(const f32x2*)(part + ((size_t)b * M + m) * 2);
    s1 += t[0];
    s2 += t[1];
  }
  const float mean = s1 * inv_d;
  const float var = fmaxf(s2 * inv_d - mean * mean, 0.f);
  f32x2 o; o[0] = mean; o[1] = rsqrtf(var + eps);
  *(f32x2*)(stats + 2 * (size_t)m) = o;
}

bool deferred_shape_ok(int M, int N, int K) {
  return M > 0 && N > 0 && N % BN == 0 && K % 64 == 0 && K >= 128 &&
         (((long long)M + 255) / 256) * ((N + 255) / 256) < (1ll << 31);
}

template <int EPI>
int launch8p(GemmArgs& a, hipStream_t s, const char* what) {
  a.tiles_n = a.Npad / BN;
  a.n_tiles = (int)((((long long)a.M + BM - 1) / BM) * a.tiles_n);
  a.sA = a.sW = a.sOut = 0; a.alpha = 1.f; a.beta = 0.f;
  bd_drop_make(nullptr, 0u, &a.drop);
  void* tok = bd_prof_on() ? bd_prof_begin(s) : nullptr;
  const long long nt = (((long long)a.M + 255) / 256) * ((a.Npad + 255) / 256);
  static int slots = gemm8p_slots();
  BD_SET_MAX_LDS((gemm8p_kernel<EPI, false>), 2 * G8_BUF);
  hipLaunchKernelGGL((gemm8p_kernel<EPI, false>), dim3((unsigned)(nt < slots ? nt : slots)), dim3(512), 2 * G8_BUF, s, a);
  if (tok) bd_prof_end(tok, s, 2.0 * (double)a.M * (double)a.n_real * (double)a.K);
  return bd_check_launch(what);
}
}  // namespace

// 1 when the forward would run a projection of this shape on the eight-wave kernel anyway (the deferred-LayerNorm
// entry points always do: the layout of their row sums is that kernel's)
extern "C" int bd_gemm_deferred_ln_supported(int M, int N, int K) {
  if (!deferred_shape_ok(M, N, K)) return 0;
  GemmArgs a{};
  a.M = M; a.Npad = N; a.n_real = N; a.K = K;
  return use_gemm8p(a) ? 1 : 0;
}

extern "C" long long bd_gemm_resid_stats_part_bytes(int M, int N) {
  return M > 0 && N > 0 ? (long long)((N + 63) / 64) * M * 2 * (long long)sizeof(float) : -1;
}

extern "C" int bd_gemm_bf16_resid_stats(const void* A, const void* W, const float* bias, float* x, void* xb, float* part,
                                        float* stats, int M, int N, int K, float eps, void* stream) {
  BD_REQUIRE(A && W && x && xb && part && stats, "bd_gemm_bf16_resid_stats: null operand");
  BD_REQUIRE(deferred_shape_ok(M, N, K), "bd_gemm_bf16_resid_stats: M=%d N=%d K=%d (N a multiple of 128, K of 64, K >= 128)", M, N, K);
  GemmArgs a{};
  a.A = (const unsigned short*)A; a.W = (const unsigned short*)W; a.bias = bias; a.out = x;
  a.M = M; a.Npad = N; a.n_real = N; a.K = K; a.ldc = N;
  a.xb_out = (unsigned short*)xb; a.part_out = part;
  hipStream_t s = (hipStream_t)stream;
  int rc = launch8p<EPI_BIAS_RESID_F32>(a, s, "bd_gemm_bf16_resid_stats");
  if (rc) return rc;
  hipLaunchKernelGGL(ln_stats_finish_kernel, dim3((M + 255) / 256), dim3(256), 0, s, part, stats, M, N / 64, 1.0f / (float)N, eps);
  return bd_check_launch("bd_gemm_bf16_resid_stats (row statistics)");
}

extern "C" int bd_gemm_bf16_deferred_ln(const void* xb, const void* W_ln, const float* bias_ln, const float* colsum,
                                        const float* stats, void* out, int M, int N, int K, int epilogue, void* stream) {
  BD_REQUIRE(xb && W_ln && colsum && stats && out, "bd_gemm_bf16_deferred_ln: null operand");
  BD_REQUIRE(epilogue == EPI_BIAS_BF16 || epilogue == EPI_BIAS_GELU_BF16, "bd_gemm_bf16_deferred_ln: epilogue %d (0 or 1)", epilogue);
  BD_REQUIRE(deferred_shape_ok(M, N, K), "bd_gemm_bf16_deferred_ln: M=%d N=%d K=%d (N a multiple of 128, K of 64, K >= 128)", M, N, K);
  GemmArgs a{};
  a.A = (const unsigned short*)xb; a.W = (const unsigned short*)W_ln; a.bias = bias_ln; a.out = out;
  a.M = M; a.Npad = N; a.n_real = N; a.K = K; a.ldc = N;
  a.ln_stats = stats; a.ln_colsum = colsum;
  hipStream_t s = (hipStream_t)stream;
  return epilogue == EPI_BIAS_BF16 ? launch8p<EPI_BIAS_BF16>(a, s, "bd_gemm_bf16_deferred_ln")
                                   : launch8p<EPI_BIAS_GELU_BF16>(a, s, "bd_gemm_bf16_deferred_ln");
}

// The eight-wave kernel whatever the tile count: a few rows of a product whose full form runs on that kernel come out
// bit for bit as there (an element's sum does not depend on the tile it sits in). Epilogues 0, 1, 2.
extern "C" int bd_gemm_bf16_tile256(const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                                    int epilogue, void* stream) {
  BD_REQUIRE(A && W && out, "bd_gemm_bf16_tile256: null operand");
  BD_REQUIRE(epilogue >= 0 && epilogue <= 2, "bd_gemm_bf16_tile256: epilogue %d (0, 1 or 2)", epilogue);
  BD_REQUIRE(deferred_shape_ok(M, N, K), "bd_gemm_bf16_tile256: M=%d N=%d K=%d (N a multiple of 128, K of 64, K >= 128)", M, N, K);
  GemmArgs a{};
  a.A = (const unsigned short*)A; a.W = (const unsigned short*)W; a.bias = bias; a.out = out;
  a.M = M; a.Npad = N; a.n_real = N; a.K = K; a.ldc = N;
  hipStream_t s = (hipStream_t)stream;
  switch (epilogue) {
    case EPI_BIAS_BF16: return launch8p<EPI_BIAS_BF16>(a, s, "bd_gemm_bf16_tile256");
    case EPI_BIAS_GELU_BF16: return launch8p<EPI_BIAS_GELU_BF16>(a, s, "bd_gemm_bf16_tile256");
    default: return launch8p<EPI_BIAS_RESID_F32>(a, s, "bd_gemm_bf16_tile256");
  }
}

// Batched C_b = A_b W_b^T over `batch` problems of one shape (element strides sA, sW, sOut; a zero
// stride shares the operand), bf16 out with row stride N. aux == null: out = acc. aux given (bf16,
// laid out like out): out = rn(rn(beta * aux) + rn(alpha * rn(acc))).
extern "C" int bd_gemm_bf16_batched(const void* A, const void* W, void* out, const void* aux, int batch,
                                    long long sA, long long sW, long long sOut, int M, int N, int K, float alpha,
                                    float beta, void* stream) {
  BD_REQUIRE(A && W && out, "bd_gemm_bf16_batched: null operand");
  BD_REQUIRE(batch > 0 && batch <= 65535, "bd_gemm_bf16_batched: batch=%d", batch);
  BD_REQUIRE(M > 0 && N > 0 && K > 0 && N % BN == 0 && K % BK == 0,
             "bd_gemm_bf16_batched: N=%d must be a multiple of %d and K=%d of %d", N, BN, K, BK);
  BD_REQUIRE(sA >= 0 && sW >= 0 && sOut >= (long long)M * N && sA % 8 == 0 && sW % 8 == 0 && sOut % 8 == 0,
             "bd_gemm_bf16_batched: bad strides");
  GemmArgs a{};
  a.A = (const unsigned short*)A; a.W = (const unsigned short*)W; a.bias = nullptr; a.out = out;
  a.aux_in = aux; a.aux_out = nullptr;
  a.M = M; a.Npad = N; a.K = K; a.n_real = N; a.ldc = N;
  a.sA = sA; a.sW = sW; a.sOut = sOut; a.alpha = alpha; a.beta = beta;
  a.tiles_n = N / BN;
  const long long nt = (((long long)M + BM - 1) / BM) * a.tiles_n;
  BD_REQUIRE(nt < (1ll << 31), "bd_gemm_bf16_batched: too many tiles");
  a.n_tiles = (int)nt;
  bd_drop_make(nullptr, 0u, &a.drop);
  hipStream_t s = (hipStream_t)stream;
  void* tok = bd_prof_on() ? bd_prof_begin(s) : nullptr;
  const int gx = persistent_grid(a.n_tiles);
  if (aux) hipLaunchKernelGGL((gemm_bf16_kernel<EPI_AXPBY_BF16, false>), dim3(gx, batch), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((gemm_bf16_kernel<EPI_BIAS_BF16, false>), dim3(gx, batch), dim3(256), 0, s, a);
  if (tok) bd_prof_end(tok, s, 2.0 * batch * (double)M * N * K);
  return bd_check_launch("bd_gemm_bf16_batched");
}

extern "C" int bd_gemm_bf16_aux(const void* A, const void* W, const float* bias, void* out,
                                const void* aux_in, void* aux_out, int M, int n_pad, int n_real, int K,
                                int ldc, int epilogue, void* stream) {
  return bd_gemm_bf16_drop(A, W, bias, out, aux_in, aux_out, M, n_pad, n_real, K, ldc, epilogue, nullptr, 0u, stream);
}

extern "C" int bd_gemm_bf16(const void* A, const void* W, const float* bias, void* out, int M, int n_pad,
                            int n_real, int K, int ldc, int epilogue, void* stream) {
  return bd_gemm_bf16_aux(A, W, bias, out, nullptr, nullptr, M, n_pad, n_real, K, ldc, epilogue, stream);
}
