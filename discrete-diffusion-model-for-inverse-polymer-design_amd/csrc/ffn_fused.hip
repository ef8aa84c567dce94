// Fused position-wise feed-forward block of a pre-norm transformer layer:
//     x <- x + W2 * GELU(W1 * LayerNorm(x) + b1) + b2
// in ONE kernel that reads the fp32 residual stream once and writes it once. The hidden
// activations ([tokens, F], the largest tensor of the layer) never exist in memory.
//
// A 256-thread workgroup owns 128 tokens, each wave 32 of them (one 32-wide MFMA column block).
// Everything a wave needs of its tokens lives in registers for the whole kernel:
//   * the residual rows as MFMA accumulators acc[T] (fp32 32x32 tiles, the TOKEN is the MFMA
//     column), initialised from x and accumulated into by the second GEMM,
//   * LayerNorm(x) as bf16 MFMA B operands, built from those same registers.
// The weights are the MFMA A operands. They are streamed through a six-unit LDS ring with
// LDS-DMA loads from a stream the host packed in fragment order (bd_ffn_pack_weights): one
// wave-instruction moves 1 KiB that is, byte for byte, the 64 lanes' A fragment of one MFMA, so
// global reads are whole lines and LDS reads are conflict-free without any swizzle.
//
// Chaining without data movement: because the weight fragments are packed by the host, the k
// order of both GEMMs is free, and it is chosen so that a result tile, rounded to bf16, IS the
// next GEMM's B operand (details at the kernel). The hidden layer is walked in chunks of 32
// units: GEMM1 (d/16 MFMAs), GELU in registers, GEMM2 (d/16 MFMAs) into the residual accumulators.
#include "bd_common.h"
#include "bd_prof.h"
#include "fused_phases.h"

namespace {

using namespace bdf;

struct FfnArgs {
  const float* x;               // [M, d] residual stream in
  float* out;                   // [M, d] residual stream out (may equal x)
  FfnPhase ph;
  int M;
  int stagger_blocks;           // workgroups of the first round (= CUs), which start phase-shifted
  int stagger_units;            // phase step in units of ~1 us
};

// 32x32x16 MFMAs: such an MFMA keeps the SIMD's vector issue port for 8 of its 32 cycles (a
// 16x16x32 one for 8 of 16), which leaves room for the GELU arithmetic, the LDS reads and the
// LDS-DMA issue of a wave that is alone on its SIMD. Layouts (h = lane / 32, c = lane % 32):
//   A operand: row c, k = 8 h + e          B operand: column c, k = 8 h + e        (e = 0..7)
//   D: column c, row 8 (r / 4) + 4 h + r % 4                                        (r = 0..15)
// acc[T][r] = x[token c][32 T + 8 (r / 4) + 4 h + r % 4]; registers 8 kk .. 8 kk + 7 of a D tile
// (kk = 0, 1), rounded to bf16, are the B operand of k-block kk of that tile for the next GEMM
// when the weight fragments use the k order  e -> 8 (2 kk + e / 4) + 4 h + e % 4  (the packing).
// The GELU of chunk j + 1 is spread over GEMM2 of chunk j (first half of the values) and GEMM1 of
// chunk j + 2 (second half), about five scalar instructions per MFMA.
//
// GELU(x) = x * clamp(0.5 + x P(min(x^2, 2.9^2)), 0, 1) in packed f16 (two values per instruction), P a cubic
// minimax fit of (Phi(x) - 0.5) / x; the result is the f16 B operand of the second GEMM (f16 MFMA, W2 stored as
// f16). |error| < 6.4e-3 absolute (rms 1.7e-3): smaller than rounding the exact GELU to bf16. No transcendental.
// (The phase itself lives in fused_phases.h.)
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_fused_kernel(FfnArgs p) {
  constexpr int D = NT * 16;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 128 + wave * 32 + (lane & 31);
  stagger_start(p.stagger_blocks, p.stagger_units);
  f32x16 acc[NT / 2];
  load_rows(acc, p.x + (size_t)(m < p.M ? m : p.M - 1) * D + 4 * (lane >> 5));
  ffn_phase<NT>(p.ph, smem, acc);
  if (m < p.M) store_rows(acc, p.out + (size_t)m * D + 4 * (lane >> 5));
}

// Packs W1 (LayerNorm gamma already folded, bf16 [F, d]) and W2 (bf16 [d, F]) into the fragment
// stream of ffn_fused_kernel: chunk j = 32 hidden units; fragment = 64 lanes x 8 bf16, lane = 32 h + c.
//   W1 fragment kb (16 features), element e:  W1[32 j + c][32 (kb / 2) + 8 (2 (kb % 2) + e / 4) + 4 h + e % 4]
//   W2 fragment 2 T + kk,         element e:  W2[32 T + c][32 j + 8 (2 kk + e / 4) + 4 h + e % 4]   (as f16)
__global__ void ffn_pack_kernel(const unsigned short* __restrict__ w1, const unsigned short* __restrict__ w2,
                                unsigned short* __restrict__ out, int d, int F) {
  const int NT = d / 16, FRAGS = 2 * NT;
  const long total = (long)(F / 32) * FRAGS * 512;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = i & 7, l = (i >> 3) & 63;
    const int c = l & 31, h = l >> 5;
    const long fr = i >> 9;
    const int f = fr % FRAGS, j = fr / FRAGS;
    unsigned short v;
    if (f < NT) {
      const int kb = f;
      const int k = 32 * (kb >> 1) + 8 * (2 * (kb & 1) + (e >> 2)) + 4 * h + (e & 3);
      v = w1[(size_t)(32 * j + c) * d + k];
    } else {
      const int T = (f - NT) >> 1, kk = (f - NT) & 1;
      // the second GEMM runs on the f16 MFMA (its other operand is the f16 GELU output): bf16 -> f16, exact
      // for every bf16 value of normal f16 magnitude (|w| >= 6.1e-5; smaller ones round at 2^-25 absolute)
      const unsigned short b = w2[(size_t)(32 * T + c) * F + 32 * j + 8 * (2 * kk + (e >> 2)) + 4 * h + (e & 3)];
      const _Float16 hv = (_Float16)__uint_as_float((unsigned)b << 16);
      v = __builtin_bit_cast(unsigned short, hv);
    }
    out[i] = v;
  }
}

}  // namespace

extern "C" int bd_ffn_fused_supported(int d, int F) { return (d == 384 || d == 256 || d == 128) && F % 32 == 0 && F >= 96 && F <= 4096; }

extern "C" long bd_ffn_pack_bytes(int d, int F) { return (long)d * F * 4; }

extern "C" int bd_ffn_pack_weights(const void* w1, const void* w2, void* out, int d, int F, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BD_REQUIRE(w1 && w2 && out, "bd_ffn_pack_weights: null pointer");
  BD_REQUIRE(bd_ffn_fused_supported(d, F), "bd_ffn_pack_weights: unsupported d=%d F=%d", d, F);
  hipLaunchKernelGGL(ffn_pack_kernel, dim3(1024), dim3(256), 0, stream, (const unsigned short*)w1,
                     (const unsigned short*)w2, (unsigned short*)out, d, F);
  return bd_check_launch("bd_ffn_pack_weights");
}

template <int NT>
static int launch_ffn(const FfnArgs& a, hipStream_t stream) {
  const size_t lds = ffn_phase_lds(NT, a.ph.F);
  BD_SET_MAX_LDS((ffn_fused_kernel<NT>), lds);
  hipLaunchKernelGGL((ffn_fused_kernel<NT>), dim3((a.M + 127) / 128), dim3(256), lds, stream, a);
  return bd_check_launch("bd_ffn_fused");
}

extern "C" int bd_ffn_fused(const float* x, float* out, const void* wpack, const float* b1, const float* b2, int M,
                            int d, int F, float eps, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BD_REQUIRE(x && out && wpack && b1 && b2, "bd_ffn_fused: null pointer");
  BD_REQUIRE(M > 0, "bd_ffn_fused: M=%d", M);
  BD_REQUIRE(bd_ffn_fused_supported(d, F), "bd_ffn_fused: unsupported d=%d F=%d", d, F);
  FfnArgs a{x, out, FfnPhase{(const unsigned char*)wpack, b1, b2, F, eps}, M, 0, 0};
  {
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
      else ncu = 256;
    }
    a.stagger_units = 3;
    // only worth it when every CU runs several workgroups in a row
    a.stagger_blocks = ((M + 127) / 128 >= 4 * ncu) ? ncu : 0;
  }
  void* tok = bd_prof_on() ? bd_prof_begin(stream) : nullptr;
  int rc;
  switch (d) {
    case 384: rc = launch_ffn<24>(a, stream); break;
    case 256: rc = launch_ffn<16>(a, stream); break;
    default: rc = launch_ffn<8>(a, stream); break;
  }
  if (tok) bd_prof_end(tok, stream, 2.0 * 2.0 * (double)M * d * F, BD_PROF_FFN);
  return rc;
}
