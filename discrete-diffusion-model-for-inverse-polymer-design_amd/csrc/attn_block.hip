// Fused self-attention block of a pre-norm transformer layer, one sequence per workgroup:
//     x <- x + Wo * Attention(LayerNorm(x) Wq^T, LayerNorm(x) Wk^T, LayerNorm(x) Wv^T) + bo
// in ONE kernel (sequence length <= 128). q, k, v, the scores and the attention output never
// exist in memory: the residual rows are read once and written once, the weights stream from
// L2 through LDS.
//
// A 256-thread workgroup owns a sequence, each wave 32 tokens (one 32-wide MFMA block). The
// structure is the one of the fused feed-forward kernel (ffn_fused.hip): weights pre-packed by
// the host in MFMA-fragment order, streamed with LDS-DMA through a ring of units of d/16
// fragments, 32x32x16 MFMAs, results chained from one GEMM into the next without leaving
// registers because the packing fixes the k order so that a result tile, rounded to bf16, IS the
// next operand:
//   per head:  Q^T = Wq x^T (kept as B operands)     K^T = Wk x^T, V = x Wv^T (rounded tiles are
//              the A fragments of the score / value products; they go to LDS, all waves read
//              them)   S = K Q^T   P = softmax over keys (exp2; the log2(e)/sqrt(dh) scale is
//              folded into Wq by the host)   O^T = V^T P^T (rounded tiles = B operands of Wo)
//              rows += Wo[:, head's columns] O^T (the rows stay in the accumulation registers)
//   then:      + bo, store.
// Every q / k / v unit of the stream is one 32x32 output tile: d/16 fragments -> d/16 MFMAs on one
// accumulator; a unit of a head's Wo slice covers d/dh output tiles with dh/16 fragments each. Key padding enters as the initial value (0 / -1e30) of the score accumulators.
// A query whose keys are all masked gives an all-zero attention output (as bd_attention_fwd).
#include "bd_common.h"
#include "bd_prof.h"
#include "fused_phases.h"

namespace {

using namespace bdf;

struct AttnArgs {
  const float* x;               // [B * L, d] residual stream in
  float* out;                   // [B * L, d] residual stream out (may equal x)
  AttnPhase ph;                 // ph.keymask: [B, L] here, advanced to the sequence by the kernel
  int B;
  int stagger_blocks, stagger_units;
};

// (The phase itself lives in fused_phases.h.)
template <int NT, int DH, int H>   // NT = d / 16, H heads of DH (padded) features
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_block_kernel(AttnArgs p) {
  constexpr int D = NT * 16;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seq = blockIdx.x, L = p.ph.L;
  const int tok = 32 * wave + (lane & 31);
  const size_t row = (size_t)seq * L + (tok < L ? tok : L - 1);
  stagger_start(p.stagger_blocks, p.stagger_units);
  AttnPhase ph = p.ph;
  if (ph.keymask != nullptr) ph.keymask += (size_t)seq * L;
  const float* xrow = p.x + row * D + 4 * (lane >> 5);
  f32x16 acc[NT / 2];
  load_rows(acc, xrow);
  attn_phase<NT, DH, H>(ph, smem, acc);
  if (tok < L) store_rows(acc, p.out + row * D + 4 * (lane >> 5));
}

// Packs the q|k|v projection (bf16 [3 H dh, d], LayerNorm gamma folded, q rows scaled by
// log2(e) / sqrt(true head size); dh = the padded head size, padded rows zero) and the output
// projection (bf16 [d, H dh], padded columns zero) into the fragment stream.
// Fragment = 64 lanes x 8 bf16, lane = 32 h + c; unit = d/16 fragments.
//   units of head hd: [q tiles | k tiles | v tiles | Wo slice], dh/32 units each.
//   q / k / v tile dt, fragment kb:
//       W[sec H dh + hd dh + 32 dt + c][32 (kb / 2) + 8 (2 (kb % 2) + e / 4) + 4 h + e % 4]
//   Wo slice: fragment g = (unit within the slice) d/16 + f covers output tile T = g / (dh/16) and
//   k-fragment 2 dt + kk = g % (dh/16) of the head's columns:
//       Wo[32 T + c][hd dh + 32 dt + 8 (2 kk + e / 4) + 4 h + e % 4]
__global__ void attn_pack_kernel(const unsigned short* __restrict__ wqkv, const unsigned short* __restrict__ wo,
                                 unsigned short* __restrict__ out, int d, int H, int DH) {
  const int NT = d / 16, DT = DH / 32, KBO = DH / 16, AW = H * DH;
  const long total = 4l * H * DT * NT * 512;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = i & 7, l = (i >> 3) & 63;
    const int c = l & 31, h = l >> 5;
    const long fr = i >> 9;
    const int f = fr % NT, u = fr / NT;
    unsigned short v;
    const int hd = u / (4 * DT), r = u % (4 * DT), sec = r / DT, dt = r % DT;
    if (sec < 3) {
      const int k = 32 * (f >> 1) + 8 * (2 * (f & 1) + (e >> 2)) + 4 * h + (e & 3);
      v = wqkv[(size_t)(sec * AW + hd * DH + 32 * dt + c) * d + k];
    } else {
      const int g = dt * NT + f, T = g / KBO, rr = g % KBO, kt = rr >> 1, kk = rr & 1;
      v = wo[(size_t)(32 * T + c) * AW + hd * DH + 32 * kt + 8 * (2 * kk + (e >> 2)) + 4 * h + (e & 3)];
    }
    out[i] = v;
  }
}

template <int NT, int DH, int H = NT * 16 / DH>
int launch_attn(const AttnArgs& a, hipStream_t stream) {
  const size_t lds = attn_phase_lds(NT, DH, H);
  BD_SET_MAX_LDS((attn_block_kernel<NT, DH, H>), lds);
  hipLaunchKernelGGL((attn_block_kernel<NT, DH, H>), dim3(a.B), dim3(256), lds, stream, a);
  return bd_check_launch("bd_attn_block_fused");
}

}  // namespace

// Head size the kernels run with: the true one, or the next of 32 / 64 / 128 with zero padding. 0 = no kernel.
static int padded_head(int d, int n_heads) {
  if (n_heads <= 0 || d % n_heads) return 0;
  const int dh = d / n_heads;
  const bool exact = (d == 384 && (dh == 128 || dh == 64)) || (d == 256 && (dh == 128 || dh == 64)) || (d == 128 && (dh == 64 || dh == 32));
  if (exact) return dh;
  if (d == 384 && n_heads == 8) return 64;      // 8 heads of 48: the reference's default backbone
  return 0;
}

extern "C" int bd_attn_block_supported(int d, int n_heads, int L) {
  return padded_head(d, n_heads) != 0 && L >= 1 && L <= 128;
}

extern "C" int bd_attn_block_width(int d, int n_heads) { return n_heads * padded_head(d, n_heads); }

extern "C" long bd_attn_pack_bytes(int d) { return 8l * d * d; }

extern "C" long bd_attn_pack_bytes_heads(int d, int n_heads) { return 8l * bd_attn_block_width(d, n_heads) * d; }

extern "C" int bd_attn_pack_weights(const void* w_qkv, const void* w_o, void* out, int d, int n_heads, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BD_REQUIRE(w_qkv && w_o && out, "bd_attn_pack_weights: null pointer");
  BD_REQUIRE(bd_attn_block_supported(d, n_heads, 128), "bd_attn_pack_weights: unsupported d=%d heads=%d", d, n_heads);
  hipLaunchKernelGGL(attn_pack_kernel, dim3(1024), dim3(256), 0, stream, (const unsigned short*)w_qkv,
                     (const unsigned short*)w_o, (unsigned short*)out, d, n_heads, padded_head(d, n_heads));
  return bd_check_launch("bd_attn_pack_weights");
}

extern "C" int bd_attn_block_fused(const float* x, float* out, const void* wpack, const float* b_q, const float* b_o,
                                   const float* b_o_fold, const unsigned char* keymask, int B, int L, int d, int n_heads,
                                   float eps, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BD_REQUIRE(x && out && wpack && b_q && b_o && b_o_fold, "bd_attn_block_fused: null pointer");
  BD_REQUIRE(B > 0, "bd_attn_block_fused: B=%d", B);
  BD_REQUIRE(bd_attn_block_supported(d, n_heads, L), "bd_attn_block_fused: unsupported d=%d heads=%d L=%d", d, n_heads, L);
  AttnArgs a{x, out, AttnPhase{(const unsigned char*)wpack, b_q, b_o, b_o_fold, keymask, L, eps}, B, 0, 3};
  {
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
      else ncu = 256;
    }
    a.stagger_blocks = (B >= 4 * ncu) ? ncu : 0;   // only worth it when every CU runs several workgroups in a row
  }
  const int dh = padded_head(d, n_heads);
  void* tok = bd_prof_on() ? bd_prof_begin(stream) : nullptr;
  int rc;
  if (d == 384 && n_heads == 8) rc = launch_attn<24, 64, 8>(a, stream);
  else if (d == 384 && dh == 128) rc = launch_attn<24, 128>(a, stream);
  else if (d == 384) rc = launch_attn<24, 64>(a, stream);
  else if (d == 256 && dh == 128) rc = launch_attn<16, 128>(a, stream);
  else if (d == 256) rc = launch_attn<16, 64>(a, stream);
  else if (dh == 64) rc = launch_attn<8, 64>(a, stream);
  else rc = launch_attn<8, 32>(a, stream);
  // algorithmic work: four d x d projections per token and the two L x L attention products
  if (tok) bd_prof_end(tok, stream, 2.0 * (double)B * L * (4.0 * d * d + 2.0 * L * d), BD_PROF_ATTN_BLOCK);
  return rc;
}
