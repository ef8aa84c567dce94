// Fused non-causal self-attention with a key-padding mask (gfx950, bf16 MFMA).
//
// Input is the fused QKV projection output [B*L, 3*d] (row = q | k | v, head h at
// columns h*DH .. h*DH+DH-1 of each third); output is [B*L, d] bf16 with heads merged.
// Semantics follow torch SDPA with a boolean mask broadcast over queries: masked keys
// get weight 0, and a query whose keys are ALL masked yields an all-zero output row
// (never NaN).
//
// One workgroup = 4 waves = 64 queries of one (batch, head); each wave owns 16 queries.
// Keys/values are staged 128 at a time into LDS (256-byte rows, XOR-swizzled 16-byte
// chunks). Both products are computed transposed so that every softmax quantity is
// lane-local:
//   S^T = K Q^T      A = K rows from LDS (ds_read_b128), B = Q fragment in registers
//   O^T = V^T P^T    A = V^T through the transposing LDS read, B = P^T straight from
//                    the S^T accumulators (no LDS round trip, no shuffles)
// The K rows fed to the first product are permuted inside each 32-key group so that the
// accumulator registers of two adjacent 16-key tiles are exactly one k-step of the
// second product in natural key order.
#include "bd_common.h"
#include "bd_drop.h"
#include "bd_prof.h"

namespace {

constexpr int KT = 128;        // keys per LDS tile
constexpr int ROWB = 256;      // LDS row stride in bytes (also for DH < 128)
constexpr int MAX_L = 8192;    // longest sequence of the forward kernel (its key-mask bits sit in LDS)

__device__ __forceinline__ int kv_off(int row, int ch) {
  return ROWB * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

typedef __attribute__((address_space(3))) void attn_lds;

__device__ __forceinline__ s16x4 lds_read_tr(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(p));
}

// QB = 16-query blocks per wave (1 or 2): with two, every K / V fragment read from LDS feeds two
// MFMAs, which halves the LDS traffic per flop (the kernel is LDS-bound with one).
template <int DH, bool DROP, int QB>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const unsigned short* __restrict__ qkv,
                                                          const unsigned char* __restrict__ keymask,
                                                          unsigned short* __restrict__ out,
                                                          float* __restrict__ lse_out,
                                                          int Ls, int H, float scale, bd_drop_args drop,
                                                          const int* __restrict__ cu, int n_qblk) {
  // Ls = row stride of the [B, Ls] key mask, of the [B, H, Ls] log-sum-exp and of the dropout rows. With cu
  // (packed rows: sequence b occupies rows cu[b] .. cu[b + 1] - 1 of qkv / out) L is that sequence's own length.
  // Workgroups are handed to the 8 XCDs in turn (blockIdx.x % 8): the logical blocks are dealt so that each XCD works
  // through one contiguous eighth of the list, query block fastest, and the query blocks of one (sequence, head)
  // find its K / V rows in their XCD's L2 instead of fetching them once per XCD.
  int bz, by, bx;
  {
    const int n_all = gridDim.x, xcd = blockIdx.x & 7, q = n_all >> 3, rr = n_all & 7;
    const int lin = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (blockIdx.x >> 3);
    bx = lin % n_qblk;
    const int t = lin / n_qblk;
    by = t % H;
    bz = t / H;
  }
  const int L = cu ? cu[bz + 1] - cu[bz] : Ls;
  if (bx * (64 * QB) >= L) return;                         // no query of this sequence in the block (uniform)
  const size_t tb = cu ? (size_t)cu[bz] : (size_t)bz * Ls;   // first row of the sequence
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KT * ROWB + MAX_L / 8];
  unsigned char* const Ks = smem;
  unsigned char* const Vs = smem + KT * ROWB;
  unsigned long long* const Mb = (unsigned long long*)(smem + 2 * KT * ROWB);   // visible keys of the sequence, one bit each

  constexpr int KS = DH / 32;   // k-steps of the first product
  constexpr int DT = DH / 16;   // output d-tiles of the second product
  constexpr int CPR = DH / 8;   // 16-byte chunks per K/V row

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int b = bz, h = by;
  const int d = H * DH;
  const size_t row3 = (size_t)3 * d;
  const int q0 = bx * (64 * QB) + wave * (16 * QB);   // block qb: queries q0 + 16 qb + c
  // exponents are taken base 2: exp(scale * (s - m)) = 2^(scale2 * s - scale2 * m), one fused multiply-add and the
  // hardware's 2^x per score; the running maximum m stays in raw score units, the log-sum-exp goes out in natural units
  const float scale2 = scale * 1.44269504089f;
  const f32x4 scale4 = {scale2, scale2, scale2, scale2};

  // Q fragments: lane holds Q[q0 + 16 qb + c][32 ks + 8 g .. +7]
  bf16x8 qf[QB][KS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int qr = q0 + 16 * qb + c; qr = qr < L ? qr : L - 1;
    const unsigned short* qp = qkv + (tb + qr) * row3 + (size_t)h * DH + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[qb][ks] = *(const bf16x8*)(qp + 32 * ks);
  }

  f32x4 o[QB][DT];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY; l_run[qb] = 0.f;
#pragma unroll
    for (int i = 0; i < DT; ++i) o[qb][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // K and V rows go to LDS by LDS-DMA, 8 + 8 pieces per wave and tile: a 1-KiB piece is 4 rows x 256 B, lane i lands at
  // row i / 16, slot i % 16 and fetches the chunk that the swizzle keeps there. Rows past the end of the sequence read
  // its last row instead of zeros: their keys are masked, P = 0 exactly, and 0 times a finite row is 0; pieces that lie
  // wholly past the end are not fetched, and their 32-key group is not multiplied (no visible key). Chunks a row does
  // not have (DH < 128) land in slots nobody reads. A wave's eight pieces of a tile are one 32-key group: all fetched
  // or none.
  const unsigned char* const kbase = (const unsigned char*)(qkv + tb * row3 + (size_t)h * DH + d);
  const unsigned char* const vbase = kbase + (size_t)d * 2;
  const unsigned lds_k = (unsigned)(size_t)(attn_lds*)Ks, lds_v = (unsigned)(size_t)(attn_lds*)Vs;
  auto stage = [&](int kt0, const unsigned char* base, unsigned lds) {
    if (kt0 + 32 * wave >= L) return;                    // the wave's group lies past the end
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int piece = 8 * wave + j;                    // rows 4 piece .. 4 piece + 3
      const int row = 4 * piece + (lane >> 4);
      int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
      ch = ch < CPR ? ch : CPR - 1;
      const int key = min(kt0 + row, L - 1);
      const unsigned voff = (unsigned)(((size_t)key * row3 + ch * 8) * 2);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   ::"s"(lds + piece * 1024), "v"(voff), "s"(base) : "memory", "m0");
    }
  };
  // the visible keys of the whole sequence, one bit each (64 per wave and step), before the first piece is issued: a
  // load the compiler waits for later would wait for every piece issued before it as well
  for (int k0 = 64 * wave; k0 < ((L + KT - 1) / KT) * KT; k0 += 256) {      // both words of the last tile
    const int key = k0 + lane;
    bool vis = false;
    if (key < L) vis = keymask ? (keymask[(size_t)b * Ls + key] != 0) : true;
    const unsigned long long bits = __ballot(vis);
    if (lane == 0) Mb[k0 >> 6] = bits;
  }
  // Per tile: K lands -> barrier -> S = K Q^T, softmax -> V lands -> barrier (nobody reads K any more: the next tile's
  // K is fetched behind PV) -> O += V^T P^T -> barrier (nobody reads V any more: the next tile's V is fetched behind
  // the next tile's scores). K pieces are issued before the V pieces of the same tile, so "at most the 8 V pieces
  // outstanding" means K has landed.
  // (the Q fragments pass through here so that their loads are waited for now, not behind the first pieces)
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[qb][ks]));
  stage(0, kbase, lds_k);
  stage(0, vbase, lds_v);
  for (int kt0 = 0; kt0 < L; kt0 += KT) {
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned long long vis0 = Mb[kt0 >> 6], vis1 = Mb[(kt0 >> 6) + 1];
    const bool all_vis = (vis0 & vis1) == ~0ull;       // the common case: no masked key in the tile, no selects below
    bool live[4];
    live[0] = (unsigned)vis0 != 0u; live[1] = (unsigned)(vis0 >> 32) != 0u;
    live[2] = (unsigned)vis1 != 0u; live[3] = (unsigned)(vis1 >> 32) != 0u;

    // ---- S^T tiles: s[qb][tile][r] <-> key_local = 32 (tile>>1) + 8 g + 4 (tile&1) + r, query c of block qb
    f32x4 s[QB][8];
#pragma unroll
    for (int tile = 0; tile < 8; ++tile) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) s[qb][tile] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (live[tile >> 1]) {
        const int krow = 32 * (tile >> 1) + 8 * (c >> 2) + 4 * (tile & 1) + (c & 3);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 a = *(const bf16x8*)(Ks + kv_off(krow, 4 * ks + g));
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
            s[qb][tile] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[qb][ks], s[qb][tile], 0, 0, 0);
        }
      }
    }

    bf16x8 pf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      // ---- masked online softmax along keys (lane-local + the 4 lanes sharing c). The running maximum is kept in raw
      //      score units; 2^(s * scale2 - m * scale2) is one fused multiply-add and one 2^x per score
      if (!all_vis) {
#pragma unroll
        for (int tile = 0; tile < 8; ++tile) {
          const int kb = 32 * (tile >> 1) + 8 * g + 4 * (tile & 1);       // bits kb .. kb + 3 of the tile's 128
          const unsigned m4 = (unsigned)((tile < 4 ? vis0 : vis1) >> (kb & 63)) & 0xfu;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (!((m4 >> r) & 1u)) s[qb][tile][r] = -INFINITY;
        }
      }
      float mloc = -INFINITY;
#pragma unroll
      for (int tile = 0; tile < 8; ++tile)
        mloc = fmaxf(fmaxf(mloc, fmaxf(s[qb][tile][0], s[qb][tile][1])), fmaxf(s[qb][tile][2], s[qb][tile][3]));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float m_new = fmaxf(m_run[qb], mloc);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;           // no visible key so far: every weight 2^(-inf) = 0
      const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_use) * scale2);    // m_run = -inf: 0
      const float msc = m_use * scale2;
      const f32x4 msc4 = {msc, msc, msc, msc};
      f32x4 ps4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tile = 0; tile < 8; ++tile) {
        const f32x4 e = s[qb][tile] * scale4 - msc4;                    // -inf stays -inf (scale2 > 0)
        f32x4 pv;
        pv[0] = __builtin_amdgcn_exp2f(e[0]); pv[1] = __builtin_amdgcn_exp2f(e[1]);
        pv[2] = __builtin_amdgcn_exp2f(e[2]); pv[3] = __builtin_amdgcn_exp2f(e[3]);
        s[qb][tile] = pv;
        ps4 += pv;
      }
      float psum = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
      psum += __shfl_xor(psum, 16, 64);
      psum += __shfl_xor(psum, 32, 64);
      l_run[qb] = l_run[qb] * alpha + psum;
      m_run[qb] = m_new;
      const f32x4 al4 = {alpha, alpha, alpha, alpha};
#pragma unroll
      for (int i = 0; i < DT; ++i) o[qb][i] *= al4;

      // ---- dropout of the probabilities (the normaliser l_run keeps every visible key)
      if (DROP) {
        const uint32_t prow = (uint32_t)(((size_t)b * H + h) * Ls + (q0 + 16 * qb + c));
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {      // tiles 2 s4, 2 s4 + 1 = keys key0 .. key0 + 7: one block
          if (!live[s4]) continue;            // no visible key: the group's probabilities are zeros already
          const int key0 = kt0 + 32 * s4 + 8 * g;
          const bd_philox_out w = bd_drop_block(drop, prow, (uint32_t)(key0 >> 3));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s[qb][2 * s4][r] *= bd_drop_factor(drop, w, r);
            s[qb][2 * s4 + 1][r] *= bd_drop_factor(drop, w, 4 + r);
          }
        }
      }

      // ---- P^T fragments: k-step s4 covers keys 32 s4 + 8 g + j = tiles (2 s4, 2 s4 + 1)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        bf16x8 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t[r] = (__bf16)s[qb][2 * s4][r];
          t[4 + r] = (__bf16)s[qb][2 * s4 + 1][r];
        }
        pf[qb][s4] = t;
      }
    }

    // ---- V has landed, and no wave reads K any more: the next tile's K rows go out
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt0 + KT < L) stage(kt0 + KT, kbase, lds_k);

    // ---- O^T += V^T P^T ; V^T fragment via two transposing reads of 4 keys x 16 d
    const int q4 = c >> 2, p4 = c & 3;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (!live[s4]) continue;
        const int r0 = 32 * s4 + 8 * g + q4;
        const int ch = 2 * dt + (p4 >> 1);
        const s16x4 lo = lds_read_tr(Vs + kv_off(r0, ch) + 8 * (p4 & 1));
        const s16x4 hi = lds_read_tr(Vs + kv_off(r0 + 4, ch) + 8 * (p4 & 1));
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lo; u.s.b = hi;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          o[qb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, pf[qb][s4], o[qb][dt], 0, 0, 0);
      }
    }
    if (kt0 + KT < L) {                                  // no wave reads V any more: the next tile's V rows go out
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      stage(kt0 + KT, vbase, lds_v);
    }
  }

  // ---- normalise and store: lane holds O[q0 + 16 qb + c][16 dt + 4 g + r]
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qrow = q0 + 16 * qb + c;
    const float inv = l_run[qb] > 0.f ? 1.0f / l_run[qb] : 0.f;
    if (qrow < L) {
      unsigned short* op = out + (tb + qrow) * d + (size_t)h * DH + 4 * g;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        u32x2 w;
        w[0] = bd_pack_bf16x2(o[qb][dt][0] * inv, o[qb][dt][1] * inv);
        w[1] = bd_pack_bf16x2(o[qb][dt][2] * inv, o[qb][dt][3] * inv);
        *(u32x2*)(op + 16 * dt) = w;
      }
      if (lse_out != nullptr && g == 0)
        lse_out[((size_t)b * H + h) * Ls + qrow] = l_run[qb] > 0.f ? (m_run[qb] * scale + __logf(l_run[qb])) : -INFINITY;
    }
  }
}

}  // namespace

// qkv [B*L, 3*H*DH] bf16, keymask [B, L] bytes (non-zero = visible) or null,
// out [B*L, H*DH] bf16, lse (optional) [B, H, L] fp32 log-sum-exp of the scaled scores.
template <int DH>
static void launch_attn_fwd(hipStream_t s, const void* qkv, const unsigned char* keymask, void* out, float* lse, int B,
                            int L, int H, float scale, const bd_drop_args& da, const int* cu) {
  const unsigned short* q = (const unsigned short*)qkv;
  unsigned short* o = (unsigned short*)out;
  if (L > 64) {      // two query blocks per wave: 128 queries per workgroup
    const int nq = (L + 127) / 128;
    dim3 grid((unsigned)((long long)nq * H * B));
    if (da.thr != 0u) hipLaunchKernelGGL((attn_fwd_kernel<DH, true, 2>), grid, dim3(256), 0, s, q, keymask, o, lse, L, H, scale, da, cu, nq);
    else hipLaunchKernelGGL((attn_fwd_kernel<DH, false, 2>), grid, dim3(256), 0, s, q, keymask, o, lse, L, H, scale, da, cu, nq);
  } else {
    dim3 grid((unsigned)((long long)H * B));
    if (da.thr != 0u) hipLaunchKernelGGL((attn_fwd_kernel<DH, true, 1>), grid, dim3(256), 0, s, q, keymask, o, lse, L, H, scale, da, cu, 1);
    else hipLaunchKernelGGL((attn_fwd_kernel<DH, false, 1>), grid, dim3(256), 0, s, q, keymask, o, lse, L, H, scale, da, cu, 1);
  }
}

static int attention_fwd_impl(const void* qkv, const unsigned char* keymask, void* out, float* lse, const int* cu,
                              int B, int L, int H, int head_dim, const bd_dropout* drop, unsigned site, void* stream) {
  BD_REQUIRE(qkv && out, "bd_attention_fwd: null operand");
  BD_REQUIRE(B > 0 && L > 0 && H > 0, "bd_attention_fwd: empty shape");
  BD_REQUIRE(B <= 65535 && H <= 65535, "bd_attention_fwd: B and H must be <= 65535");
  BD_REQUIRE(L <= MAX_L, "bd_attention_fwd: L=%d exceeds %d", L, MAX_L);
  BD_REQUIRE((long long)B * H * ((L + 127) / 128) < (1ll << 31), "bd_attention_fwd: too many workgroups");
  bd_drop_args da;
  BD_REQUIRE(bd_drop_make(drop, site, &da), "bd_attention_fwd: dropout p=%g outside [0, 1)", (double)drop->p);
  BD_REQUIRE(da.thr == 0u || (long long)B * H * L < (1ll << 32), "bd_attention_fwd: dropout needs B*H*L < 2^32");
  bd_prof_scope prof(stream, 4.0 * (double)B * H * L * L * head_dim, BD_PROF_ATTN_FWD);
  const float scale = 1.0f / sqrtf((float)head_dim);
  hipStream_t s = (hipStream_t)stream;
  switch (head_dim) {
    case 32: launch_attn_fwd<32>(s, qkv, keymask, out, lse, B, L, H, scale, da, cu); break;
    case 64: launch_attn_fwd<64>(s, qkv, keymask, out, lse, B, L, H, scale, da, cu); break;
    case 128: launch_attn_fwd<128>(s, qkv, keymask, out, lse, B, L, H, scale, da, cu); break;
    default:
      return bd_fail(BD_ERR_UNSUPPORTED, "bd_attention_fwd: head_dim %d not in {32, 64, 128}", head_dim);
  }
  return bd_check_launch("bd_attention_fwd");
}

extern "C" int bd_attention_fwd_drop(const void* qkv, const unsigned char* keymask, void* out, float* lse, int B, int L,
                                     int H, int head_dim, const bd_dropout* drop, unsigned site, void* stream) {
  return attention_fwd_impl(qkv, keymask, out, lse, nullptr, B, L, H, head_dim, drop, site, stream);
}

extern "C" int bd_attention_fwd_varlen(const void* qkv, const unsigned char* keymask, void* out, float* lse,
                                       const int* cu_seqlens, int B, int L, int H, int head_dim, const bd_dropout* drop,
                                       unsigned site, void* stream) {
  BD_REQUIRE(cu_seqlens != nullptr, "bd_attention_fwd_varlen: null cu_seqlens");
  return attention_fwd_impl(qkv, keymask, out, lse, cu_seqlens, B, L, H, head_dim, drop, site, stream);
}

extern "C" int bd_attention_fwd(const void* qkv, const unsigned char* keymask, void* out,
                                float* lse, int B, int L, int H, int head_dim, void* stream) {
  return bd_attention_fwd_drop(qkv, keymask, out, lse, B, L, H, head_dim, nullptr, 0u, stream);
}
