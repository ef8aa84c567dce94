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
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

struct FfnArgs {
  const float* x;               // [M, d] residual stream in
  float* out;                   // [M, d] residual stream out (may equal x)
  const unsigned char* wpack;   // packed weight stream (bd_ffn_pack_weights)
  const float* b1;              // [F] first bias with LayerNorm beta folded in
  const float* b2;              // [d]
  int M, F;
  float eps;
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
// GELU(x) = x * clamp(0.5 + x P(min(x^2, 16)), 0, 1): x P(x^2) is a degree-7 minimax fit of
// 0.5 erf(x / sqrt 2) on |x| <= 4; past that the product overshoots +-0.5 and the clamp saturates
// it. |error| < 2e-4 absolute, below the bf16 rounding of the activations that matter. No
// transcendental and only scalar fp32 FMAs (packed fp32 math is slow next to MFMAs).
struct Gelu2 { float x0, x1, s0, s1, p0, p1; };

template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_fused_kernel(FfnArgs p) {
  constexpr int D = NT * 16;
  constexpr int NT32 = NT / 2;                // 32-feature tiles of the residual row
  constexpr int UNIT = NT * 1024;             // unit 2 j = W1 of chunk j, unit 2 j + 1 = W2 of chunk j
  constexpr int RING = 6;
  constexpr int PER_UNIT = NT / 4;
  constexpr int NF = 2 * NT;                  // fragments (= MFMAs) per chunk
  constexpr int FB = 8;                       // fragment read batch
  static_assert(NT % FB == 0, "fragment batches must tile both GEMMs");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* const sb1 = (float*)(smem + RING * UNIT);     // F + 32 floats
  typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
  typedef const __attribute__((address_space(3))) f32x4* lds_f4;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA targets stay scalar
  const int h = lane >> 5, c = lane & 31;
  const int nch = p.F >> 5;
  const int nunits = 2 * nch;
  const int m = blockIdx.x * 128 + wave * 32 + c;

  auto issue_unit = [&](int u) {
    const unsigned char* src = p.wpack + (size_t)u * UNIT + lane * 16;
    unsigned char* dst = smem + (u % RING) * UNIT;
#pragma unroll
    for (int i = 0; i < PER_UNIT; ++i) {
      const int f = wave + 4 * i;
      __builtin_amdgcn_global_load_lds((glb_void*)(src + f * 1024), (lds_void*)(dst + f * 1024), 16, 0, 0);
    }
  };

  // The first workgroup of every CU starts a little later than its neighbours' (16 phases): all
  // workgroups take the same time, so without this every CU would load its rows, compute and
  // store in lockstep, and the memory system would alternate between overload and idleness.
  if ((int)blockIdx.x < p.stagger_blocks) {
    const int phase = (blockIdx.x >> 3) & 15;
    for (int k = 0; k < phase * p.stagger_units; ++k) __builtin_amdgcn_s_sleep(32);
  }

  f32x16 acc[NT32];
  {
    const float* xr = p.x + (size_t)(m < p.M ? m : p.M - 1) * D + 4 * h;
#pragma unroll
    for (int T = 0; T < NT32; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *(const f32x4*)(xr + 32 * T + 8 * q);
        acc[T][4 * q + 0] = v[0]; acc[T][4 * q + 1] = v[1]; acc[T][4 * q + 2] = v[2]; acc[T][4 * q + 3] = v[3];
      }
    // the rows live in accumulation registers from here on (without this hint the compiler keeps
    // all of them in ordinary VGPRs through the LayerNorm and spills about a hundred)
#pragma unroll
    for (int T = 0; T < NT32; ++T) asm volatile("" : "+a"(acc[T]));
  }
#pragma unroll
  for (int u = 0; u < 5; ++u) issue_unit(u);            // F >= 96: the first five units exist
  for (int i = tid; i < p.F + 32; i += 256) sb1[i] = i < p.F ? p.b1[i] : 0.f;

  // ---- LayerNorm -> bf16 B operands (gamma / beta are folded into W1 and b1)
  bf16x8 xln[NT];
  {
    float s = 0.f;
#pragma unroll
    for (int T = 0; T < NT32; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[T][r];
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int T = 0; T < NT32; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float dv = acc[T][r] - mean; q += dv * dv; }
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q * (1.0f / D) + p.eps);
#pragma unroll
    for (int kb = 0; kb < NT; ++kb) {
      bf16x8 f;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (__bf16)((acc[kb >> 1][8 * (kb & 1) + e] - mean) * rstd);
      xln[kb] = f;
    }
  }

  const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem + lane * 16;
  const unsigned ldsb = (unsigned)(size_t)(lds_void*)sb1 + 16 * h;

  f32x16 hn_cur, hn_next;      // GEMM1 results of chunk j and j + 1 (rows = hidden units of the chunk)
  bf16x8 hb[2], hb0_next;      // GELU(hn_cur) as the two B operands of GEMM2; first half of the next
  Gelu2 gs;

  // GELU part g (0..47): pair g / 6 = registers 2 pair, 2 pair + 1 of the source tile, in six parts
  auto gelu_part = [&](int g, const f32x16& src, bf16x8& dst) {
    const int pair = g / 6, part = g % 6;
    if (part == 0) {
      gs.x0 = src[2 * pair]; gs.x1 = src[2 * pair + 1];
      gs.s0 = fminf(gs.x0 * gs.x0, 16.0f); gs.s1 = fminf(gs.x1 * gs.x1, 16.0f);
      asm volatile("" : "+v"(gs.s0), "+v"(gs.s1));
    } else if (part == 1) {
      gs.p0 = gs.s0 * -1.903182666e-09f + 1.410586241e-07f; gs.p1 = gs.s1 * -1.903182666e-09f + 1.410586241e-07f;
      gs.p0 = gs.p0 * gs.s0 + -4.565313249e-06f;            gs.p1 = gs.p1 * gs.s1 + -4.565313249e-06f;
      asm volatile("" : "+v"(gs.p0), "+v"(gs.p1));
    } else if (part == 2) {
      gs.p0 = gs.p0 * gs.s0 + 8.634553992e-05f;  gs.p1 = gs.p1 * gs.s1 + 8.634553992e-05f;
      gs.p0 = gs.p0 * gs.s0 + -1.085383136e-03f; gs.p1 = gs.p1 * gs.s1 + -1.085383136e-03f;
      asm volatile("" : "+v"(gs.p0), "+v"(gs.p1));
    } else if (part == 3) {
      gs.p0 = gs.p0 * gs.s0 + 9.789848745e-03f;  gs.p1 = gs.p1 * gs.s1 + 9.789848745e-03f;
      gs.p0 = gs.p0 * gs.s0 + -6.636063135e-02f; gs.p1 = gs.p1 * gs.s1 + -6.636063135e-02f;
      asm volatile("" : "+v"(gs.p0), "+v"(gs.p1));
    } else if (part == 4) {
      gs.p0 = gs.p0 * gs.s0 + 3.989269649e-01f;  gs.p1 = gs.p1 * gs.s1 + 3.989269649e-01f;
      gs.p0 = __builtin_amdgcn_fmed3f(gs.x0 * gs.p0 + 0.5f, 0.0f, 1.0f);
      gs.p1 = __builtin_amdgcn_fmed3f(gs.x1 * gs.p1 + 0.5f, 0.0f, 1.0f);
      asm volatile("" : "+v"(gs.p0), "+v"(gs.p1));
    } else {
      const int j0 = 4 * ((pair >> 1) & 1) + 2 * (pair & 1);
      dst[j0] = (__bf16)(gs.x0 * gs.p0);
      dst[j0 + 1] = (__bf16)(gs.x1 * gs.p1);
      asm volatile("" : "+v"(dst));
    }
  };

  // One chunk iteration: GEMM1 of chunk j1 into hn_next, then (unless first) GEMM2 of chunk j1 - 1
  // from hb. Fragment i < NT is W1 k-block i, fragment NT + 2 T + kk is W2 tile T, k-block kk.
  // A wave alone on its SIMD must never drain its LDS queue: the fragment reads run LOOK ahead of
  // their MFMAs and every MFMA waits with an exact count. The compiler only places
  // wait-for-everything on LDS reads here, so the steady-state path issues its reads and waits as
  // inline asm (ASM_READS); the one-off first step keeps compiler-managed loads.
  constexpr int LOOK = 8;
  auto chunk_step = [&](int j1, auto first_tag) {
    constexpr bool first = decltype(first_tag)::value;
    constexpr bool ASM_READS = !first;
    unsigned W1b = lds0 + ((2 * j1) % RING) * UNIT;
    unsigned W2b = lds0 + ((2 * j1 + RING - 1) % RING) * UNIT;
    unsigned bb1 = ldsb + 128 * j1;
    asm volatile("" : "+v"(W1b), "+v"(W2b), "+v"(bb1));
    constexpr int nf = first ? NT : NF;
    bf16x8 wf[NF];
    auto read_frag = [&](int i) {
      if (ASM_READS) {
        if (i < NT) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i]) : "v"(W1b), "i"(i * 1024));
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i]) : "v"(W2b), "i"((i - NT) * 1024));
      } else {
        wf[i] = *(lds_frag)(size_t)(i < NT ? W1b + i * 1024 : W2b + (i - NT) * 1024);
      }
    };
    // bias first (oldest in the queue), then the first LOOK fragments
    f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (ASM_READS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[q]) : "v"(bb1), "i"(32 * q));
      else bq[q] = *(lds_f4)(size_t)(bb1 + 32 * q);
    }
#pragma unroll
    for (int i = 0; i < LOOK; ++i) read_frag(i);
    if (ASM_READS) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]) : "i"(LOOK));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      hn_next[4 * q + 0] = bq[q][0]; hn_next[4 * q + 1] = bq[q][1]; hn_next[4 * q + 2] = bq[q][2]; hn_next[4 * q + 3] = bq[q][3];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      if (i < nf) {
        if (i + LOOK < nf) read_frag(i + LOOK);
        // reads younger than fragment i that may stay in flight
        if (ASM_READS) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(wf[i]) : "i"(i + LOOK < nf ? LOOK : nf - 1 - i));
        if (i < NT) {
          hn_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xln[i], hn_next, 0, 0, 0);
        } else {
          const int T = (i - NT) >> 1, kk = (i - NT) & 1;
          acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], hb[kk], acc[T], 0, 0, 0);
        }
      }
      {
        if (i < NT) {
          // second half of GELU(hn_cur) -> hb[1], beside GEMM1 (skipped before the first chunk)
          if (!first) {
#pragma unroll
            for (int g = 24 + i * 24 / NT; g < 24 + (i + 1) * 24 / NT; ++g) gelu_part(g, hn_cur, hb[1]);
          }
        } else {
          // first half of GELU(hn_next) -> hb0_next, beside GEMM2 (or alone before the first chunk)
#pragma unroll
          for (int g = (i - NT) * 24 / NT; g < (i - NT + 1) * 24 / NT; ++g) gelu_part(g, hn_next, hb0_next);
        }
      }
      __builtin_amdgcn_sched_barrier(0);     // nothing moves between fragment steps
    }
    hn_cur = hn_next;
    hb[0] = hb0_next;
    asm volatile("" : "+v"(hb[0]), "+v"(hb[1]));
  };

  // units 0, 1, 2 landed (3 and 4 may still be in flight), bias table written
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_UNIT) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  hb[1] = xln[0];                    // defined value; overwritten before its first real use
  chunk_step(0, std::true_type{});

  for (int j = 0; j < nch; ++j) {
    // units 2 j + 5 and 2 j + 6 go where units 2 j - 1 and 2 j were read during iteration j - 1
    const int u5 = 2 * j + 5;
    if (u5 < nunits) issue_unit(u5);
    if (u5 + 1 < nunits) issue_unit(u5 + 1);
    // GEMM1 of chunk j + 1 (past the end: a stale unit and zero biases, result unused), GEMM2 of chunk j
    chunk_step(j + 1, std::false_type{});
    // units 2 j + 3 and 2 j + 4 (issued one iteration ago) must have landed for every wave before
    // the next iteration reads them; what this iteration issued stays in flight
    if (u5 + 1 < nunits) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_UNIT) : "memory");
    else if (u5 < nunits) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_UNIT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: + b2, store the new residual rows
  if (m < p.M) {
    float* o = p.out + (size_t)m * D + 4 * h;
#pragma unroll
    for (int T = 0; T < NT32; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *(const f32x4*)(p.b2 + 32 * T + 8 * q + 4 * h);
        f32x4 v;
        v[0] = acc[T][4 * q + 0] + b[0]; v[1] = acc[T][4 * q + 1] + b[1];
        v[2] = acc[T][4 * q + 2] + b[2]; v[3] = acc[T][4 * q + 3] + b[3];
        *(f32x4*)(o + 32 * T + 8 * q) = v;
      }
  }
}

// Packs W1 (LayerNorm gamma already folded, bf16 [F, d]) and W2 (bf16 [d, F]) into the fragment
// stream of ffn_fused_kernel: chunk j = 32 hidden units; fragment = 64 lanes x 8 bf16, lane = 32 h + c.
//   W1 fragment kb (16 features), element e:  W1[32 j + c][32 (kb / 2) + 8 (2 (kb % 2) + e / 4) + 4 h + e % 4]
//   W2 fragment 2 T + kk,         element e:  W2[32 T + c][32 j + 8 (2 kk + e / 4) + 4 h + e % 4]
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
      v = w2[(size_t)(32 * T + c) * F + 32 * j + 8 * (2 * kk + (e >> 2)) + 4 * h + (e & 3)];
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
  const size_t lds = 6 * (size_t)NT * 1024 + ((size_t)a.F + 32) * 4;
  hipFuncSetAttribute((const void*)ffn_fused_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((ffn_fused_kernel<NT>), dim3((a.M + 127) / 128), dim3(256), lds, stream, a);
  return bd_check_launch("bd_ffn_fused");
}

extern "C" int bd_ffn_fused(const float* x, float* out, const void* wpack, const float* b1, const float* b2, int M,
                            int d, int F, float eps, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BD_REQUIRE(x && out && wpack && b1 && b2, "bd_ffn_fused: null pointer");
  BD_REQUIRE(M > 0, "bd_ffn_fused: M=%d", M);
  BD_REQUIRE(bd_ffn_fused_supported(d, F), "bd_ffn_fused: unsupported d=%d F=%d", d, F);
  FfnArgs a{x, out, (const unsigned char*)wpack, b1, b2, M, F, eps, 0, 0};
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
