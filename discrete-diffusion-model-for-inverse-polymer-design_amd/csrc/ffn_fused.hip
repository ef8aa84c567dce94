// Fused position-wise feed-forward block of a pre-norm transformer layer:
//     x <- x + W2 * GELU(W1 * LayerNorm(x) + b1) + b2
// in ONE kernel that reads the fp32 residual stream once and writes it once. The hidden
// activations ([tokens, F], the largest tensor of the layer) never exist in memory.
//
// A 256-thread workgroup owns 128 tokens, each wave 32 of them. Everything a wave needs of its
// tokens lives in registers for the whole kernel:
//   * the residual rows as MFMA accumulators  acc[n-tile][token-tile]  (fp32, 16x16 tiles with
//     the TOKEN as the MFMA column), initialised from x and accumulated into by the second GEMM,
//   * LayerNorm(x) as bf16 MFMA B operands, built from those same registers.
// The weights are the MFMA A operands. They are streamed through a three-buffer LDS ring with
// LDS-DMA loads from a stream the host packed in fragment order (bd_ffn_pack_weights): one
// wave-instruction moves 1 KiB that is, byte for byte, the 64 lanes' A fragment of one MFMA, so
// global reads are whole lines and LDS reads are conflict-free without any swizzle.
//
// Chaining without data movement: a 16x16 MFMA result holds, for token column c, rows 4g..4g+3
// (g = lane / 16); the next MFMA wants, for the same column, k = 8g..8g+7. Because the weight
// fragments are packed by the host, the row order of W1 tiles and the k order of both GEMMs are
// free: the packing is chosen so that two result tiles, converted to bf16, ARE the next B operand.
// The hidden layer is walked in chunks of 32 units: GEMM1 (2 tiles x d/32 k-blocks), GELU in
// registers, GEMM2 (d/16 tiles x 1 k-block) into the residual accumulators.
#include "bd_common.h"
#include "bd_prof.h"
#include <stdlib.h>

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
};

typedef __attribute__((ext_vector_type(2))) float f32x2;

// GELU(x) = x * clamp(0.5 + x P(min(x^2, 16)), 0, 1): x P(x^2) is a degree-7 minimax fit of
// 0.5 erf(x / sqrt 2) on |x| <= 4; past that the product overshoots +-0.5 and the clamp saturates
// it. |error| < 2e-4 absolute, below the bf16 rounding of the activations that matter. No
// transcendental and only scalar fp32 FMAs (packed fp32 math is slow next to MFMAs).
// Evaluated for two values at a time in three parts, so the chains can be spread between MFMAs.
struct GeluState { float x0, x1, s0, s1, p0, p1; };
__device__ __forceinline__ void gelu_part0(GeluState& g, float a, float b) {
  g.x0 = a; g.x1 = b;
  g.s0 = fminf(a * a, 16.0f); g.s1 = fminf(b * b, 16.0f);
  g.p0 = g.s0 * -1.903182666e-09f + 1.410586241e-07f; g.p1 = g.s1 * -1.903182666e-09f + 1.410586241e-07f;
  g.p0 = g.p0 * g.s0 + -4.565313249e-06f;             g.p1 = g.p1 * g.s1 + -4.565313249e-06f;
}
__device__ __forceinline__ void gelu_part1(GeluState& g) {
  g.p0 = g.p0 * g.s0 + 8.634553992e-05f;  g.p1 = g.p1 * g.s1 + 8.634553992e-05f;
  g.p0 = g.p0 * g.s0 + -1.085383136e-03f; g.p1 = g.p1 * g.s1 + -1.085383136e-03f;
  g.p0 = g.p0 * g.s0 + 9.789848745e-03f;  g.p1 = g.p1 * g.s1 + 9.789848745e-03f;
}
__device__ __forceinline__ f32x2 gelu_part2(GeluState& g) {
  g.p0 = g.p0 * g.s0 + -6.636063135e-02f; g.p1 = g.p1 * g.s1 + -6.636063135e-02f;
  g.p0 = g.p0 * g.s0 + 3.989269649e-01f;  g.p1 = g.p1 * g.s1 + 3.989269649e-01f;
  const float t0 = __builtin_amdgcn_fmed3f(g.x0 * g.p0 + 0.5f, 0.0f, 1.0f);
  const float t1 = __builtin_amdgcn_fmed3f(g.x1 * g.p1 + 0.5f, 0.0f, 1.0f);
  return (f32x2){g.x0 * t0, g.x1 * t1};
}

template <int NT, int DBG = 0>  // NT = d / 16 output tiles; d / 32 = NT / 2 k-blocks; DBG: timing ablations
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ffn_fused_kernel(FfnArgs p) {
  constexpr int D = NT * 16;
  constexpr int NKB = NT / 2;
  constexpr int MT = 2;                       // 16-token tiles per wave
  // The stream is a sequence of units of NT fragments (1 KiB each): unit 2 j = W1 of chunk j
  // (2 tiles x NKB k-blocks), unit 2 j + 1 = W2 of chunk j (NT tiles). The LDS ring holds six
  // units: byte o of the stream lives at o mod (6 units).
  constexpr int UNIT = NT * 1024;
  constexpr int RING = 6;
  constexpr int PER_UNIT = NT / 4;            // LDS-DMA instructions per wave and unit
  static_assert(NT % 4 == 0, "fragments must split evenly over the four waves");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* const sb1 = (float*)(smem + RING * UNIT);     // F + 32 floats

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int nch = p.F >> 5;
  const int nunits = 2 * nch;
  const int m0 = blockIdx.x * 128 + wave * 32;

  auto issue_unit = [&](int u) {
    const unsigned char* src = p.wpack + (size_t)u * UNIT + lane * 16;
    unsigned char* dst = smem + (u % RING) * UNIT;
#pragma unroll
    for (int i = 0; i < PER_UNIT; ++i) {
      const int f = wave + 4 * i;
      __builtin_amdgcn_global_load_lds((glb_void*)(src + f * 1024), (lds_void*)(dst + f * 1024), 16, 0, 0);
    }
  };

  // ---- residual rows -> accumulators: lane (g, c) holds x[token c of tile mt][16 nt + 4 g + r]
  f32x4 acc[NT][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + c;
    const int row = m < p.M ? m : p.M - 1;
    const float* xr = p.x + (size_t)row * D + 4 * g;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = *(const f32x4*)(xr + 16 * nt);
  }
  // the first five units start flowing (F >= 96 is checked by the host, so they all exist)
#pragma unroll
  for (int u = 0; u < 5; ++u) issue_unit(u);

  for (int i = tid; i < p.F + 32; i += 256) sb1[i] = i < p.F ? p.b1[i] : 0.f;

  // ---- LayerNorm (gamma / beta live in the packed W1 and in b1) -> bf16 B operands
  bf16x8 xln[NKB][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) s += (acc[nt][mt][0] + acc[nt][mt][1]) + (acc[nt][mt][2] + acc[nt][mt][3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float dv = acc[nt][mt][r] - mean; q += dv * dv; }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q * (1.0f / D) + p.eps);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      bf16x8 f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f[r] = (__bf16)((acc[2 * kb][mt][r] - mean) * rstd);
        f[4 + r] = (__bf16)((acc[2 * kb + 1][mt][r] - mean) * rstd);
      }
      xln[kb][mt] = f;
    }
  }

  // One chunk step = GEMM1 of chunk j1 (24 W1 fragments) followed by GEMM2 of chunk j1 - 1 (24 W2
  // fragments, B operand hb) with the GELU of GEMM1's result spread between GEMM2's MFMAs.
  // Fragment reads run LOOK fragments ahead of their MFMAs; the sched_group_barrier sequence at
  // the end pins that interleave (1 LDS read, 2 MFMA, a few VALU per fragment).
  // Fragment reads are batched: while the MFMAs of one batch of FB fragments run, the reads of the
  // next batch are issued during the FIRST half of it, so that by the time the compiler's
  // wait-for-all (it does not count LDS reads individually here) comes up at the next batch, the
  // youngest read is half a batch old and the one wave of a SIMD does not stall on it.
  constexpr int FB = 8;
  constexpr int NF = 2 * NT;                  // fragments per step
  static_assert(NT % FB == 0, "batches must not straddle the two GEMMs' prologue-only case");
  typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
  typedef const __attribute__((address_space(3))) f32x4* lds_f4;
  auto frag_off = [&](int i) -> int {
    // GEMM1 order: k-block major, tile minor, so the four accumulators are hit round robin
    return i < NT ? ((i & 1) * NKB + (i >> 1)) * 1024 : (i - NT) * 1024;
  };
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem + lane * 16;
  const unsigned ldsb = (unsigned)(size_t)(lds_void*)sb1 + 32 * g;
  auto chunk_step = [&](int j1, bool do_gemm2, bf16x8 (&hb)[MT]) {
    // unit bases as opaque registers, so every fragment read is base + 16-bit immediate
    unsigned W1b = lds0 + ((2 * j1) % RING) * UNIT;
    unsigned W2b = lds0 + ((2 * j1 + RING - 1) % RING) * UNIT;
    unsigned bb1 = ldsb + 128 * j1;
    asm volatile("" : "+v"(W1b), "+v"(W2b), "+v"(bb1));
    const int nf = do_gemm2 ? NF : NT;
    auto read_frag = [&](int i) -> bf16x8 { return *(lds_frag)(size_t)((i < NT ? W1b : W2b) + frag_off(i)); };
    const f32x4 bia0 = *(lds_f4)(size_t)bb1, bia1 = *(lds_f4)(size_t)(bb1 + 16);
    bf16x8 wf[NF];
    bf16x8 hnew[MT];
    GeluState gs;
    if (DBG & 4) { hnew[0] = hb[0]; hnew[1] = hb[1]; }
#pragma unroll
    for (int i = 0; i < FB; ++i) wf[i] = read_frag(i);
    f32x4 hn[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { hn[0][mt] = bia0; hn[1][mt] = bia1; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      if (i < nf) {
        const int q = i % FB, nb = (i / FB + 1) * FB;     // position in the batch, start of the next
        if (q < FB / 2 && nb < nf) {
          wf[nb + 2 * q] = read_frag(nb + 2 * q);
          wf[nb + 2 * q + 1] = read_frag(nb + 2 * q + 1);
        }
        if (i < NT) {
          const int kb = i >> 1, t = i & 1;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            if (!(DBG & 16)) hn[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xln[kb][mt], hn[t][mt], 0, 0, 0);
            else asm volatile("" :: "v"(wf[i]));
        } else {
          const int nt = i - NT;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            if (!(DBG & 8)) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], hb[mt], acc[nt][mt], 0, 0, 0);
            else asm volatile("" :: "v"(wf[i]));
        }
      }
      // GELU piece k = (mt, t, r2), two values, in three parts on consecutive fragments; the two
      // result tiles are, lane for lane, the k = 8 g + (4 t + r) B operand of GEMM2. The pieces
      // sit beside GEMM2's MFMAs when there is a GEMM2.
      if (i >= NT && !(DBG & 4)) {
#pragma unroll
        for (int slot = (i - NT) * 24 / NT; slot < (i - NT + 1) * 24 / NT; ++slot) {
          const int k = slot / 3, part = slot % 3, mt = k >> 2, t = (k >> 1) & 1, r2 = k & 1;
          // the empty asm statements tie each part to its fragment step (plain arithmetic would
          // otherwise drift to the end of the block, behind all the MFMAs)
          if (part == 0) { gelu_part0(gs, hn[t][mt][2 * r2], hn[t][mt][2 * r2 + 1]); asm volatile("" : "+v"(gs.p0), "+v"(gs.p1)); }
          if (part == 1) { gelu_part1(gs); asm volatile("" : "+v"(gs.p0), "+v"(gs.p1)); }
          if (part == 2) {
            const f32x2 v = gelu_part2(gs);
            hnew[mt][4 * t + 2 * r2] = (__bf16)v[0];
            hnew[mt][4 * t + 2 * r2 + 1] = (__bf16)v[1];
            asm volatile("" : "+v"(hnew[mt]));
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);     // nothing moves between fragment steps
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) hb[mt] = hnew[mt];
    // keeps the activation in this step (it would otherwise sink below the barrier that follows)
    asm volatile("" : "+v"(hb[0]), "+v"(hb[1]));
  };

  // units 0, 1, 2 landed (3 and 4 may still be in flight), bias table written
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_UNIT) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf16x8 hb[MT];
  chunk_step(0, false, hb);

  for (int j = 0; j < ((DBG & 32) ? 0 : nch); ++j) {
    // units 2 j + 5 and 2 j + 6 go where units 2 j - 1 and 2 j were read during iteration j - 1
    const int u5 = 2 * j + 5;
    if (!(DBG & 1)) {
    if (u5 < nunits) issue_unit(u5);
    if (u5 + 1 < nunits) issue_unit(u5 + 1);
    }

    // GEMM1 of chunk j + 1 (past the end: a stale unit and zero biases, result unused), GEMM2 of
    // chunk j
    chunk_step(j + 1, true, hb);

    // units 2 j + 3 and 2 j + 4 (issued one iteration ago) must have landed for every wave before
    // the next iteration reads them; what this iteration issued stays in flight
    if (DBG & 2) continue;
    if ((DBG & 1)) {}
    else if (u5 + 1 < nunits) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_UNIT) : "memory");
    else if (u5 < nunits) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_UNIT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: + b2, store the new residual rows
  f32x4 bb[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bb[nt] = *(const f32x4*)(p.b2 + 16 * nt + 4 * g);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + 16 * mt + c;
    if (m < p.M) {
      float* o = p.out + (size_t)m * D + 4 * g;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 v = acc[nt][mt];
        v[0] += bb[nt][0]; v[1] += bb[nt][1]; v[2] += bb[nt][2]; v[3] += bb[nt][3];
        *(f32x4*)(o + 16 * nt) = v;
      }
    }
  }
}

// Packs W1 (LayerNorm gamma already folded, bf16 [F, d]) and W2 (bf16 [d, F]) into the fragment
// stream: chunk j = 32 hidden units; fragment = 64 lanes x 8 bf16.
//   W1 fragment (t, kb), lane (c, g), element e:  W1[32 j + 8 (c / 4) + 4 t + c % 4][16 (2 kb + e / 4) + 4 g + e % 4]
//   W2 fragment nt,      lane (c, g), element e:  W2[16 nt + c][32 j + 8 g + e]
__global__ void ffn_pack_kernel(const unsigned short* __restrict__ w1, const unsigned short* __restrict__ w2,
                                unsigned short* __restrict__ out, int d, int F) {
  const int NT = d / 16, NKB = NT / 2, FRAGS = 2 * NT;
  const long total = (long)(F / 32) * FRAGS * 512;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = i & 7, l = (i >> 3) & 63;
    const int c = l & 15, g = l >> 4;
    const long fr = i >> 9;
    const int f = fr % FRAGS, j = fr / FRAGS;
    unsigned short v;
    if (f < 2 * NKB) {
      const int t = f / NKB, kb = f % NKB;
      const int u = 32 * j + 8 * (c >> 2) + 4 * t + (c & 3);
      const int k = 16 * (2 * kb + (e >> 2)) + 4 * g + (e & 3);
      v = w1[(size_t)u * d + k];
    } else {
      const int nt = f - 2 * NKB;
      v = w2[(size_t)(16 * nt + c) * F + 32 * j + 8 * g + e];
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

template <int NT, int DBG = 0>
static int launch_ffn(const FfnArgs& a, hipStream_t stream) {
  const size_t lds = 6 * (size_t)NT * 1024 + ((size_t)a.F + 32) * 4;
  hipFuncSetAttribute((const void*)ffn_fused_kernel<NT, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((ffn_fused_kernel<NT, DBG>), dim3((a.M + 127) / 128), dim3(256), lds, stream, a);
  return bd_check_launch("bd_ffn_fused");
}

extern "C" int bd_ffn_fused(const float* x, float* out, const void* wpack, const float* b1, const float* b2, int M,
                            int d, int F, float eps, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BD_REQUIRE(x && out && wpack && b1 && b2, "bd_ffn_fused: null pointer");
  BD_REQUIRE(M > 0, "bd_ffn_fused: M=%d", M);
  BD_REQUIRE(bd_ffn_fused_supported(d, F), "bd_ffn_fused: unsupported d=%d F=%d", d, F);
  FfnArgs a{x, out, (const unsigned char*)wpack, b1, b2, M, F, eps};
  void* tok = bd_prof_on() ? bd_prof_begin(stream) : nullptr;
  int rc;
  switch (d) {
    case 384: {
      const char* e = getenv("BD_FFN_DBG");
      const int dbg = e ? atoi(e) : 0;
      switch (dbg) {
        case 1: rc = launch_ffn<24, 1>(a, stream); break;
        case 3: rc = launch_ffn<24, 3>(a, stream); break;
        case 4: rc = launch_ffn<24, 4>(a, stream); break;
        case 8: rc = launch_ffn<24, 8>(a, stream); break;
        case 16: rc = launch_ffn<24, 16>(a, stream); break;
        case 24: rc = launch_ffn<24, 24>(a, stream); break;
        case 28: rc = launch_ffn<24, 28>(a, stream); break;
        case 31: rc = launch_ffn<24, 31>(a, stream); break;
        case 32: rc = launch_ffn<24, 32>(a, stream); break;
        default: rc = launch_ffn<24>(a, stream); break;
      }
      break;
    }
    case 256: rc = launch_ffn<16>(a, stream); break;
    default: rc = launch_ffn<8>(a, stream); break;
  }
  if (tok) bd_prof_end(tok, stream, 2.0 * 2.0 * (double)M * d * F);
  return rc;
}
