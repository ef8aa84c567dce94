// Fused self-attention block of a pre-norm transformer layer, one sequence per workgroup:
//     x <- x + Wo * Attention(LayerNorm(x) Wq^T, LayerNorm(x) Wk^T, LayerNorm(x) Wv^T) + bo
// in ONE kernel (sequence length <= 128). q, k, v, the scores and the attention output never
// exist in memory: the residual rows are read (twice) and written once, the weights stream from
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
//   then:      acc = Wo O^T over all heads, + x + bo, store.
// Every unit of the stream is one 32x32 output tile: d/16 fragments -> d/16 MFMAs on one
// accumulator. Key padding enters as the initial value (0 / -1e30) of the score accumulators.
// A query whose keys are all masked gives an all-zero attention output (as bd_attention_fwd).
#include "bd_common.h"
#include "bd_prof.h"
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
typedef __attribute__((address_space(3))) bf16x8* lds_frag_w;
typedef const __attribute__((address_space(3))) f32x4* lds_f4;
typedef const __attribute__((address_space(3))) float* lds_f1;

struct AttnArgs {
  const float* x;               // [B * L, d] residual stream in
  float* out;                   // [B * L, d] residual stream out (may equal x)
  const unsigned char* wpack;   // packed weight stream (bd_attn_pack_weights)
  const float* bq;              // [d]: q bias with LayerNorm beta folded, scaled like the q rows
  const float* bo;              // [d] output bias
  const float* bo_fold;         // [d] output bias + Wo * (v bias with LayerNorm beta folded)
  const unsigned char* keymask; // [B, L] bytes, non-zero = visible; null = all visible
  int B, L;
  float eps;
  int stagger_blocks, stagger_units;
};

template <int NT, int DH>   // NT = d / 16, DH = head dimension
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_block_kernel(AttnArgs p) {
  constexpr int D = NT * 16;
  constexpr int NT32 = NT / 2;
  constexpr int H = D / DH;
  constexpr int DT = DH / 32;                 // 32-row tiles of a head's q / k / v
  constexpr int KBO = DH / 16;                // 16-wide k-blocks of a head
  constexpr int UNIT = NT * 1024;
  constexpr int RING = 3;
  constexpr int PER_UNIT = NT / 4;
  constexpr int QKV_UNITS = 3 * H * DT;
  constexpr int NUNITS = QKV_UNITS + NT32;
  constexpr int KV_BYTES = 128 * DH * 2;
  constexpr int LOOK = NT >= 16 ? 8 : 4;      // fragment reads in flight
  constexpr int MID = NT / 2;                 // MFMA index of a unit's barrier
  static_assert(MID <= NT - LOOK, "the barrier must precede the first read of the next unit");
  static_assert(NT % 4 == 0 && DH % 32 == 0 && D % DH == 0, "unsupported shape");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const sK = smem + RING * UNIT;
  unsigned char* const sV = sK + KV_BYTES;
  float* const sbias = (float*)(sV + KV_BYTES);        // d floats (q bias)
  float* const smask = sbias + D;                  // 128 floats: 0 (visible) or -1e30

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, c = lane & 31;
  const int seq = blockIdx.x;
  const int tok = 32 * wave + c;                        // token of this lane's MFMA column
  const size_t row = (size_t)seq * p.L + (tok < p.L ? tok : p.L - 1);

  auto issue_unit = [&](int u) {
    const unsigned char* src = p.wpack + (size_t)u * UNIT + lane * 16;
    unsigned char* dst = smem + (u % RING) * UNIT;
#pragma unroll
    for (int i = 0; i < PER_UNIT; ++i) {
      const int f = wave + 4 * i;
      __builtin_amdgcn_global_load_lds((glb_void*)(src + f * 1024), (lds_void*)(dst + f * 1024), 16, 0, 0);
    }
  };

  // phase-shifted start of the first workgroup of every CU (see ffn_fused.hip)
  if ((int)blockIdx.x < p.stagger_blocks) {
    const int phase = (blockIdx.x >> 3) & 15;
    for (int k = 0; k < phase * p.stagger_units; ++k) __builtin_amdgcn_s_sleep(32);
  }

  // ---- LayerNorm of the wave's 32 rows -> bf16 operands xln[kb] (row / column = token c,
  //      k = feature 32 (kb / 2) + 8 (2 (kb % 2) + e / 4) + 4 h + e % 4)
  bf16x8 xln[NT];
  {
    f32x16 xt[NT32];
    const float* xr = p.x + row * D + 4 * h;
#pragma unroll
    for (int T = 0; T < NT32; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *(const f32x4*)(xr + 32 * T + 8 * q);
        xt[T][4 * q + 0] = v[0]; xt[T][4 * q + 1] = v[1]; xt[T][4 * q + 2] = v[2]; xt[T][4 * q + 3] = v[3];
      }
    issue_unit(0);
    issue_unit(1);
    for (int i = tid; i < D; i += 256) sbias[i] = p.bq[i];
    if (tid < 128) {
      const bool vis = tid < p.L && (p.keymask == nullptr || p.keymask[(size_t)seq * p.L + tid] != 0);
      smask[tid] = vis ? 0.f : -1e30f;
    }
    float s = 0.f;
#pragma unroll
    for (int T = 0; T < NT32; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += xt[T][r];
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int T = 0; T < NT32; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float dv = xt[T][r] - mean; q += dv * dv; }
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q * (1.0f / D) + p.eps);
#pragma unroll
    for (int kb = 0; kb < NT; ++kb) {
      bf16x8 f;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (__bf16)((xt[kb >> 1][8 * (kb & 1) + e] - mean) * rstd);
      xln[kb] = f;
    }
  }

  const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem + lane * 16;
  const unsigned ldsK = (unsigned)(size_t)(lds_void*)sK + lane * 16;
  const unsigned ldsV = (unsigned)(size_t)(lds_void*)sV + lane * 16;
  const unsigned ldsB = (unsigned)(size_t)(lds_void*)sbias;
  const unsigned ldsM = (unsigned)(size_t)(lds_void*)smask + 16 * h;

  // The weight stream is consumed as ONE continuous sequence of fragments. One unit = NT fragments
  // -> NT MFMAs on one 32x32 accumulator; fragment i of a unit lives in register wf[i], and its LDS
  // read is issued LOOK MFMAs earlier - across unit boundaries - with an exact wait count before
  // each MFMA (inline asm: the compiler would wait for all outstanding reads each time, and a wave
  // alone on its SIMD must not drain its LDS queue).
  // Ring protocol, one barrier per unit, in its MIDDLE: when every wave has passed the middle of
  // unit u, (a) every wave's LDS-DMA pieces of unit u + 1 have landed (each waits for its own
  // before the barrier), so reading u + 1 may start during the tail of u, and (b) nobody reads
  // unit u - 1 any more, so unit u + 2 is issued into that slot.
  bf16x8 wf[NT];
  auto mid_sync = [&](int u) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (u + 2 < NUNITS) issue_unit(u + 2);
  };
  // WEIGHT_IS_A: tile = W_frag * op (rows = weight rows), else tile = op * W_frag (rows = tokens).
  auto run_unit = [&](int u, auto weight_is_a, auto last_unit, f32x16 t, const bf16x8 (&op)[NT]) -> f32x16 {
    constexpr bool WEIGHT_IS_A = decltype(weight_is_a)::value;
    constexpr bool LAST = decltype(last_unit)::value;
    unsigned bc = lds0 + (u % RING) * UNIT, bn = lds0 + ((u + 1) % RING) * UNIT;
    asm volatile("" : "+v"(bc), "+v"(bn));
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      if (i == MID) mid_sync(u);
      const int j = i + LOOK;
      if (j < NT) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[j]) : "v"(bc), "i"(j * 1024));
      else if (!LAST) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[j - NT]) : "v"(bn), "i"((j - NT) * 1024));
      // reads younger than fragment i that may stay in flight
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(wf[i]) : "i"((j < NT || !LAST) ? LOOK : NT - 1 - i));
      if (WEIGHT_IS_A) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], op[i], t, 0, 0, 0);
      else t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[i], wf[i], t, 0, 0, 0);
    }
    return t;
  };
  // accumulator initialised with per-ROW values table[8 q + 4 h + e] (rows of a D tile)
  auto rows_init = [&](unsigned addr) -> f32x16 {
    f32x16 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *(lds_f4)(size_t)(addr + 32 * q);
      t[4 * q + 0] = b[0]; t[4 * q + 1] = b[1]; t[4 * q + 2] = b[2]; t[4 * q + 3] = b[3];
    }
    return t;
  };
  auto pack_half = [&](const f32x16& t, int kk) -> bf16x8 {
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (__bf16)t[8 * kk + e];
    return f;
  };
  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

  // units 0 and 1 and the tables are staged; start the fragment stream
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    unsigned b0 = lds0;
    asm volatile("" : "+v"(b0));
#pragma unroll
    for (int i = 0; i < LOOK; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i]) : "v"(b0), "i"(i * 1024));
  }

  bf16x8 oall[NT];              // attention output of all heads as B operands of the Wo product
#pragma unroll
  for (int i = 0; i < NT; ++i) oall[i] = xln[0];   // defined values; every entry is overwritten
  bool dead = false;            // every key of the sequence masked (same for all queries and heads)

  int u = 0;
#pragma nounroll
  for (int hd = 0; hd < H; ++hd) {
    // q bias rows of this head's tiles (the k bias shifts all scores of a query equally and drops
    // out of the softmax; the v bias is folded into the output bias by the host)
    f32x16 qinit[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) qinit[dt] = rows_init(ldsB + (hd * DH + 32 * dt + 4 * h) * 4);
    // ---- Q^T tiles -> B operands of the score product
    bf16x8 qb[KBO];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const f32x16 t = run_unit(u, std::true_type{}, std::false_type{}, qinit[dt], xln);
      qb[2 * dt] = pack_half(t, 0);
      qb[2 * dt + 1] = pack_half(t, 1);
      ++u;
    }
    // ---- K^T tiles: rounded halves are the A fragments (row = key) of S = K Q^T
#pragma nounroll
    for (int dt = 0; dt < DT; ++dt) {
      const f32x16 t = run_unit(u, std::true_type{}, std::false_type{}, zero16, xln);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        *(lds_frag_w)(size_t)(ldsK + ((wave * KBO + 2 * dt + kk) << 10)) = pack_half(t, kk);
      ++u;
    }
    // ---- V tiles (rows = tokens): rounded halves are the A fragments (row = dh) of O^T = V^T P^T
#pragma nounroll
    for (int dt = 0; dt < DT; ++dt) {
      const f32x16 t = run_unit(u, std::false_type{}, std::false_type{}, zero16, xln);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        *(lds_frag_w)(size_t)(ldsV + ((dt * 8 + 2 * wave + kk) << 10)) = pack_half(t, kk);
      ++u;
    }
    // K and V of this head are complete and visible to all waves
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- S = K Q^T (rows = keys, columns = this wave's queries), key padding as initial value
    f32x16 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) s[kt] = rows_init(ldsM + 128 * kt);
    {
      bf16x8 kf[4 * KBO];
#pragma unroll
      for (int i = 0; i < LOOK; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[i]) : "v"(ldsK), "i"(i * 1024));
#pragma unroll
      for (int i = 0; i < 4 * KBO; ++i) {
        if (i + LOOK < 4 * KBO)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[i + LOOK]) : "v"(ldsK), "i"((i + LOOK) * 1024));
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(kf[i]) : "i"(i + LOOK < 4 * KBO ? LOOK : 4 * KBO - 1 - i));
        s[i / KBO] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i], qb[i % KBO], s[i / KBO], 0, 0, 0);
      }
    }
    // ---- softmax over the keys of each query column (half of them in the lane c + 32)
    float m = s[0][0];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, s[kt][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    dead = m < -1e29f;                        // every key masked
    float l = 0.f;
    bf16x8 pb[8];                             // P as B operands: k-block 2 kt + kk = 16 keys
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pv = __builtin_amdgcn_exp2f(s[kt][8 * kk + e] - m);
          l += pv;
          f[e] = (__bf16)pv;
        }
        pb[2 * kt + kk] = f;
      }
    l += __shfl_xor(l, 32, 64);
    const float inv = dead ? 0.f : __builtin_amdgcn_rcpf(l);

    // ---- O^T = V^T P^T (rows = dh, columns = queries) -> B operands of the Wo product
    bf16x8 onew[KBO];
    {
      bf16x8 vf[DT * 8];
#pragma unroll
      for (int i = 0; i < LOOK; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vf[i]) : "v"(ldsV), "i"(i * 1024));
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
          const int i = dt * 8 + kb;
          if (i + LOOK < DT * 8)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vf[i + LOOK]) : "v"(ldsV), "i"((i + LOOK) * 1024));
          asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(vf[i]) : "i"(i + LOOK < DT * 8 ? LOOK : DT * 8 - 1 - i));
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i], pb[kb], o, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= inv;
        onew[2 * dt] = pack_half(o, 0);
        onew[2 * dt + 1] = pack_half(o, 1);
      }
    }
    // heads land in order: shift the array down by one head and append
#pragma unroll
    for (int i = 0; i + KBO < NT; ++i) oall[i] = oall[i + KBO];
#pragma unroll
    for (int i = 0; i < KBO; ++i) oall[NT - KBO + i] = onew[i];
  }

  // ---- out = Wo O^T: one unit per 32-feature tile
  f32x16 acc[NT32];
#pragma unroll
  for (int T = 0; T < NT32; ++T) {
    if (T + 1 < NT32) acc[T] = run_unit(u, std::true_type{}, std::false_type{}, zero16, oall);
    else acc[T] = run_unit(u, std::true_type{}, std::true_type{}, zero16, oall);
    ++u;
  }

  // ---- epilogue: + residual + bo, store
  if (tok < p.L) {
    const float* xr = p.x + row * D + 4 * h;
    float* o = p.out + row * D + 4 * h;
#pragma unroll
    for (int T = 0; T < NT32; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 xv = *(const f32x4*)(xr + 32 * T + 8 * q);
        const f32x4 b = *(const f32x4*)((dead ? p.bo : p.bo_fold) + 32 * T + 8 * q + 4 * h);
        f32x4 v;
        v[0] = acc[T][4 * q + 0] + xv[0] + b[0]; v[1] = acc[T][4 * q + 1] + xv[1] + b[1];
        v[2] = acc[T][4 * q + 2] + xv[2] + b[2]; v[3] = acc[T][4 * q + 3] + xv[3] + b[3];
        *(f32x4*)(o + 32 * T + 8 * q) = v;
      }
  }
}

// Packs the q|k|v projection (bf16 [3d, d], LayerNorm gamma folded, q rows scaled by
// log2(e) / sqrt(dh)) and the output projection (bf16 [d, d]) into the fragment stream.
// Fragment = 64 lanes x 8 bf16, lane = 32 h + c; unit = d/16 fragments = one output tile.
//   units of head hd: [q tiles | k tiles | v tiles], tile dt, fragment kb:
//       W[sec d + hd dh + 32 dt + c][32 (kb / 2) + 8 (2 (kb % 2) + e / 4) + 4 h + e % 4]
//   then unit T of Wo, fragment kbg = hd (dh/16) + 2 dt + kk:
//       Wo[32 T + c][hd dh + 32 dt + 8 (2 kk + e / 4) + 4 h + e % 4]
__global__ void attn_pack_kernel(const unsigned short* __restrict__ wqkv, const unsigned short* __restrict__ wo,
                                 unsigned short* __restrict__ out, int d, int H) {
  const int NT = d / 16, DH = d / H, DT = DH / 32, KBO = DH / 16;
  const long total = 2l * NT * NT * 512;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = i & 7, l = (i >> 3) & 63;
    const int c = l & 31, h = l >> 5;
    const long fr = i >> 9;
    const int f = fr % NT, u = fr / NT;
    unsigned short v;
    if (u < 3 * H * DT) {
      const int hd = u / (3 * DT), r = u % (3 * DT), sec = r / DT, dt = r % DT;
      const int k = 32 * (f >> 1) + 8 * (2 * (f & 1) + (e >> 2)) + 4 * h + (e & 3);
      v = wqkv[(size_t)(sec * d + hd * DH + 32 * dt + c) * d + k];
    } else {
      const int T = u - 3 * H * DT;
      const int hd = f / KBO, rr = f % KBO, dt = rr >> 1, kk = rr & 1;
      v = wo[(size_t)(32 * T + c) * d + hd * DH + 32 * dt + 8 * (2 * kk + (e >> 2)) + 4 * h + (e & 3)];
    }
    out[i] = v;
  }
}

template <int NT, int DH>
int launch_attn(const AttnArgs& a, hipStream_t stream) {
  const size_t lds = 3 * (size_t)NT * 1024 + 2 * (size_t)128 * DH * 2 + (size_t)NT * 16 * 4 + 128 * 4;
  hipFuncSetAttribute((const void*)attn_block_kernel<NT, DH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((attn_block_kernel<NT, DH>), dim3(a.B), dim3(256), lds, stream, a);
  return bd_check_launch("bd_attn_block_fused");
}

}  // namespace

extern "C" int bd_attn_block_supported(int d, int n_heads, int L) {
  if (n_heads <= 0 || d % n_heads) return 0;
  const int dh = d / n_heads;
  const bool shape = (d == 384 && dh == 128) || (d == 256 && (dh == 128 || dh == 64)) || (d == 128 && (dh == 64 || dh == 32));
  return shape && L >= 1 && L <= 128;
}

extern "C" long bd_attn_pack_bytes(int d) { return 8l * d * d; }

extern "C" int bd_attn_pack_weights(const void* w_qkv, const void* w_o, void* out, int d, int n_heads, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BD_REQUIRE(w_qkv && w_o && out, "bd_attn_pack_weights: null pointer");
  BD_REQUIRE(bd_attn_block_supported(d, n_heads, 128), "bd_attn_pack_weights: unsupported d=%d heads=%d", d, n_heads);
  hipLaunchKernelGGL(attn_pack_kernel, dim3(1024), dim3(256), 0, stream, (const unsigned short*)w_qkv,
                     (const unsigned short*)w_o, (unsigned short*)out, d, n_heads);
  return bd_check_launch("bd_attn_pack_weights");
}

extern "C" int bd_attn_block_fused(const float* x, float* out, const void* wpack, const float* b_q, const float* b_o,
                                   const float* b_o_fold, const unsigned char* keymask, int B, int L, int d, int n_heads,
                                   float eps, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BD_REQUIRE(x && out && wpack && b_q && b_o && b_o_fold, "bd_attn_block_fused: null pointer");
  BD_REQUIRE(B > 0, "bd_attn_block_fused: B=%d", B);
  BD_REQUIRE(bd_attn_block_supported(d, n_heads, L), "bd_attn_block_fused: unsupported d=%d heads=%d L=%d", d, n_heads, L);
  AttnArgs a{x, out, (const unsigned char*)wpack, b_q, b_o, b_o_fold, keymask, B, L, eps, 0, 3};
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
  const int dh = d / n_heads;
  void* tok = bd_prof_on() ? bd_prof_begin(stream) : nullptr;
  int rc;
  if (d == 384) rc = launch_attn<24, 128>(a, stream);
  else if (d == 256 && dh == 128) rc = launch_attn<16, 128>(a, stream);
  else if (d == 256) rc = launch_attn<16, 64>(a, stream);
  else if (dh == 64) rc = launch_attn<8, 64>(a, stream);
  else rc = launch_attn<8, 32>(a, stream);
  // algorithmic work: four d x d projections per token and the two L x L attention products
  if (tok) bd_prof_end(tok, stream, 2.0 * (double)B * L * (4.0 * d * d + 2.0 * L * d), BD_PROF_ATTN_BLOCK);
  return rc;
}
