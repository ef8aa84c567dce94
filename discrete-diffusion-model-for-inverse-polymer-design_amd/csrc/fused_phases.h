// Device-side building blocks of the fused inference kernels (ffn_fused.hip, attn_block.hip,
// backbone_mega.hip): the feed-forward phase and the attention phase of a transformer layer, each
// working on residual rows held in registers (see those files for the design notes).
#pragma once
#include "bd_common.h"
#include <type_traits>

namespace bdf {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
typedef __attribute__((address_space(3))) bf16x8* lds_frag_w;
typedef const __attribute__((address_space(3))) f32x4* lds_f4;
typedef const __attribute__((address_space(3))) float* lds_f1;

// four values of the GELU in flight, as two packed f16 pairs: input, its square, the polynomial
struct Gelu4 { unsigned x[2], s[2], p[2]; };
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// bytes of LDS the phases need
__host__ __device__ constexpr size_t ffn_phase_lds(int NT, int F) { return 6 * (size_t)NT * 1024 + ((size_t)F + 64) * 4 + (size_t)NT * 16 * 4; }
__host__ __device__ constexpr size_t attn_phase_lds(int NT, int DH, int H) {
  return 3 * (size_t)NT * 1024 + 2 * (size_t)128 * DH * 2 + (size_t)H * DH * 4 + 128 * 4 + 2 * (size_t)NT * 16 * 4;
}

// Where the single-launch forward keeps both phases' tables: behind everything either phase puts into LDS, so that
// a phase's tables can be fetched while the other phase runs.
__host__ __device__ constexpr size_t phase_tables_base(int NT, int DH) {
  const size_t a = 3 * (size_t)NT * 1024 + 2 * (size_t)128 * DH * 2, f = 6 * (size_t)NT * 1024;
  return a > f ? a : f;
}
__host__ __device__ constexpr size_t ffn_tables_bytes(int NT, int F) { return ((size_t)F + 64 + (size_t)NT * 16) * 4; }
__host__ __device__ constexpr size_t attn_tables_bytes(int NT, int DH, int H) { return ((size_t)H * DH + 128 + 2 * (size_t)NT * 16) * 4; }

// `bytes` (a multiple of 16) from global memory to LDS by LDS-DMA, 1-KiB pieces dealt to the four waves in turn
// (`turn` counts the pieces across calls). The caller waits for its own pieces (vmcnt) and meets the other waves at a
// barrier before anybody reads the table.
__device__ __forceinline__ void fetch_table(const float* src, float* dst, int bytes, int wave, int lane, int& turn) {
  const unsigned lds = (unsigned)(size_t)(lds_void*)dst;
  unsigned lane16 = lane * 16;
  asm volatile("" : "+v"(lane16));        // recomputed at every call: hoisted out of the layer loop the offsets get spilled
  for (int off = 0; off < bytes; off += 1024) {
    if ((turn++ & 3) == wave && off + lane16 < (unsigned)bytes) {
      const unsigned m0v = __builtin_amdgcn_readfirstlane(lds + off);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   ::"s"(m0v), "v"(off + lane16), "s"(src)
                   : "memory", "m0");
    }
  }
}

// Loads / stores of the wave's 32 rows in accumulator layout: acc[T][4 q + e] = row[32 T + 8 q + 4 h + e]
template <int NT32>
__device__ __forceinline__ void load_rows(f32x16 (&acc)[NT32], const float* xrow) {
#pragma unroll
  for (int T = 0; T < NT32; ++T)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *(const f32x4*)(xrow + 32 * T + 8 * q);
      acc[T][4 * q + 0] = v[0]; acc[T][4 * q + 1] = v[1]; acc[T][4 * q + 2] = v[2]; acc[T][4 * q + 3] = v[3];
    }
  // the rows live in accumulation registers from here on (without this hint the compiler keeps
  // all of them in ordinary VGPRs through the LayerNorm and spills about a hundred)
#pragma unroll
  for (int T = 0; T < NT32; ++T) asm volatile("" : "+a"(acc[T]));
}
template <int NT32>
__device__ __forceinline__ void store_rows(const f32x16 (&acc)[NT32], float* orow) {
#pragma unroll
  for (int T = 0; T < NT32; ++T)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
      v[0] = acc[T][4 * q + 0]; v[1] = acc[T][4 * q + 1]; v[2] = acc[T][4 * q + 2]; v[3] = acc[T][4 * q + 3];
      *(f32x4*)(orow + 32 * T + 8 * q) = v;
    }
}
// phase-shifted start of the first workgroup of every CU: all workgroups take the same time, so
// without this every CU would load its rows, compute and store in lockstep, and the memory system
// would alternate between overload and idleness
__device__ __forceinline__ void stagger_start(int stagger_blocks, int stagger_units) {
  if ((int)blockIdx.x < stagger_blocks) {
    const int phase = (blockIdx.x >> 3) & 15;
    for (int k = 0; k < phase * stagger_units; ++k) __builtin_amdgcn_s_sleep(32);
  }
}

// ---------------------------------------------------------------------------------------------
// Feed-forward phase: acc (in) = the wave's 32 residual rows, acc (out) = rows + W2 GELU(W1 LN(rows)
// + b1) + b2. smem: 6 units + (F + 64) + d floats. All four waves of the workgroup call it together;
// nobody may still be using smem when they enter.
struct FfnPhase {
  const unsigned char* wpack;   // packed weight stream (bd_ffn_pack_weights)
  const float* b1;              // [F] first bias with LayerNorm beta folded in
  const float* b2;              // [d]
  int F;
  float eps;
  // the phase's tables ((F + 64) floats of first biases, zeros past F, then d floats of second biases) at a caller's
  // place in LDS, already filled (the single-launch forward fetches them during the phase before); null: behind the
  // ring, filled here
  float* tables = nullptr;
};

template <int NT>
__device__ __forceinline__ void ffn_phase(const FfnPhase& p, unsigned char* smem, f32x16 (&acc)[NT / 2]) {
  // the rows live in accumulation registers (see load_rows)
#pragma unroll
  for (int T = 0; T < NT / 2; ++T) asm volatile("" : "+a"(acc[T]));
  constexpr int D = NT * 16;
  constexpr int NT32 = NT / 2;                // 32-feature tiles of the residual row
  constexpr int UNIT = NT * 1024;             // unit 2 j = W1 of chunk j, unit 2 j + 1 = W2 of chunk j
  constexpr int RING = 6;
  constexpr int PER_UNIT = NT / 4;            // 1-KiB pieces of a unit that one wave moves
  constexpr int NF = 2 * NT;                  // fragments (= MFMAs) per chunk iteration
  constexpr int LOOK = 8;                     // fragment reads in flight ahead of their MFMA (the wait counts and the
                                              // slot table below are written for 8; 6 also works and measures the same)
  constexpr int SYNC = NF - LOOK;             // step of the iteration's barrier
  constexpr int DMA_EVERY = 4;                // one piece per four steps on average: 2 PER_UNIT pieces per iteration
  constexpr int DMA_GROUP = 2;                // issued in groups of this many (4 and 6 measure the same)
  static_assert(NT % 8 == 0 && NF / DMA_EVERY == 2 * PER_UNIT && PER_UNIT >= 2, "unsupported width");
  // hand-over of the hidden tile inside the second GEMM (steps counted from NT): copy out of the accumulator in
  // COPY_STEPS steps, then the next chunk's biases into it from step BIAS_AT on, PER registers per step; the biases are
  // read at step BIAS_READ of the iteration and have landed LOOK steps later
  constexpr int PER = NT >= 16 ? 2 : 4;
  constexpr int COPY_STEPS = 2 + 8 / PER;
  constexpr int BIAS_READ = NT >= 16 ? NT : NT - 4;
  constexpr int BIAS_AT = BIAS_READ + LOOK - NT > COPY_STEPS ? BIAS_READ + LOOK - NT : COPY_STEPS;
  static_assert(BIAS_READ >= 0 && BIAS_AT + 16 / PER <= NT, "copy, then biases, inside the second GEMM");
  float* const sb1 = p.tables ? p.tables : (float*)(smem + RING * UNIT);     // F + 64 floats
  float* const sb2 = sb1 + p.F + 64;                   // D floats: the output bias, staged with the other table (read
                                                       // from memory at the end of the phase its latency stands alone)
  typedef const __attribute__((address_space(3))) bf16x8* lds_frag;
  typedef const __attribute__((address_space(3))) f32x4* lds_f4;

  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));     // per-lane offsets are recomputed in every phase: hoisted out of the caller's layer loop they spill
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA targets stay scalar
  const int h = lane >> 5;
  const int nch = p.F >> 5;
  const int nunits = 2 * nch;

  // this wave's k-th 1-KiB piece of unit u -> ring slot u % RING
  auto issue_piece = [&](int u, int k) {
    const int f = wave + 4 * k;
    __builtin_amdgcn_global_load_lds((glb_void*)(p.wpack + (size_t)u * UNIT + lane * 16 + f * 1024),
                                     (lds_void*)(smem + (u % RING) * UNIT + f * 1024), 16, 0, 0);
  };
  auto issue_unit = [&](int u) {
#pragma unroll
    for (int k = 0; k < PER_UNIT; ++k) issue_piece(u, k);
  };

#pragma unroll
  for (int u = 0; u < 5; ++u) issue_unit(u);            // F >= 96: the first five units exist
  if (p.tables == nullptr) {
    for (int i = tid; i < p.F + 64; i += 256) sb1[i] = i < p.F ? p.b1[i] : 0.f;
    for (int i = tid; i < D; i += 256) sb2[i] = p.b2[i];
  }

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
  u32x4 hb[2], hb0_next;       // GELU(hn_cur) as the two B operands (8 f16 each) of GEMM2; first half of the next
  Gelu4 gs = {};

  // GELU in packed f16, two values per instruction: x = cvt_pkrtz(fp32 pair), s = min(x^2, 2.9^2),
  // GELU(x) = x * clamp(0.5 + x P(s), 0, 1) with P a cubic minimax fit of (Phi(x) - 0.5) / x on |x| <= 2.9 (past
  // that the product overshoots +-0.5 and the clamp saturates it); the result pair IS a dword of the f16 B operand
  // of the second GEMM, whose weight fragments are stored as f16 (bd_ffn_pack_weights; bf16 values are exact in
  // f16). |error| < 6.4e-3 absolute, rms 1.7e-3, against 1.6e-2 / 4.8e-3 for rounding the exact GELU to bf16 (what
  // the unfused path does). 8 packed operations per pair = 4 per value (the fp32 form took 12): the vector work of
  // a feed-forward step drops from 16 to 5 cycles beside its MFMA.
  // GELU part g (0..31) of a 16-value tile: quad g / 8 = registers 4 quad .. 4 quad + 3 of the source tile, in
  // eight parts of ONE packed operation on each of its two pairs.
  constexpr unsigned GC0 = 0x8c018c01u, GC1 = 0x1e081e08u, GC2 = 0xab87ab87u, GC3 = 0x36513651u;   // cubic, high to low
  constexpr unsigned GX2 = 0x48344834u, GHALF = 0x38003800u;                                        // 2.9^2, 0.5
  unsigned gc0v = GC0;
  asm volatile("" : "+v"(gc0v));
  auto gelu_part = [&](int g, const f32x16& src, u32x4& dst) {
    const int quad = g / 8, part = g % 8;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      if (part == 0) {
        asm("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(gs.x[pr]) : "v"(src[4 * quad + 2 * pr]), "v"(src[4 * quad + 2 * pr + 1]));
      } else if (part == 1) {
        asm("v_pk_mul_f16 %0, %1, %1" : "=v"(gs.s[pr]) : "v"(gs.x[pr]));
      } else if (part == 2) {
        asm("v_pk_min_f16 %0, %1, %2" : "=v"(gs.s[pr]) : "v"(gs.s[pr]), "s"(GX2));
      } else if (part == 3) {
        asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(gs.p[pr]) : "v"(gc0v), "v"(gs.s[pr]), "s"(GC1));
      } else if (part == 4) {
        asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(gs.p[pr]) : "v"(gs.p[pr]), "v"(gs.s[pr]), "s"(GC2));
      } else if (part == 5) {
        asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(gs.p[pr]) : "v"(gs.p[pr]), "v"(gs.s[pr]), "s"(GC3));
      } else if (part == 6) {
        asm("v_pk_fma_f16 %0, %1, %2, %3 clamp" : "=v"(gs.p[pr]) : "v"(gs.x[pr]), "v"(gs.p[pr]), "s"(GHALF));
      } else {
        // registers 4 quad + v of a 32x32 tile are rows 8 quad + 4 h + v: operand elements 4 (quad % 2) + v of
        // k-block quad / 2, i.e. dword 2 (quad % 2) + pr of that operand
        unsigned r;
        asm("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(gs.x[pr]), "v"(gs.p[pr]));
        dst[2 * (quad & 1) + pr] = r;
      }
    }
    if (part == 7) asm volatile("" : "+v"(dst));
    asm volatile("" : "+v"(gs.x[0]), "+v"(gs.x[1]), "+v"(gs.s[0]), "+v"(gs.s[1]), "+v"(gs.p[0]), "+v"(gs.p[1]));
  };
  constexpr int GH = 16;       // GELU parts per half tile (two quads)

  // units 0, 1, 2 landed (3 and 4 may still be in flight), bias table written
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_UNIT) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- before the loop: GEMM1 of chunk 0 (unit 0) and the first half of its GELU; compiler-managed reads
  {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *(lds_f4)(size_t)(ldsb + 32 * q);
      hn_next[4 * q + 0] = b[0]; hn_next[4 * q + 1] = b[1]; hn_next[4 * q + 2] = b[2]; hn_next[4 * q + 3] = b[3];
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) hn_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(lds_frag)(size_t)(lds0 + i * 1024), xln[i], hn_next, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < GH; ++g) gelu_part(g, hn_next, hb0_next);
    hn_cur = hn_next;
    hb[0] = hb0_next;
    hb[1] = hb0_next;                // defined value; overwritten before its first real use
    // the hidden tile of iteration 0 starts from chunk 1's biases (inside the loop an iteration sets up the next one's)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *(lds_f4)(size_t)(ldsb + 128 + 32 * q);
      hn_next[4 * q + 0] = b[0]; hn_next[4 * q + 1] = b[1]; hn_next[4 * q + 2] = b[2]; hn_next[4 * q + 3] = b[3];
    }
    asm volatile("" : "+a"(hn_next));          // in its accumulation registers here, not at the head of the loop
    // every wave is done with unit 0: unit 6 takes its slot (the loop issues odd units from 5, even units from 8)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (6 < nunits) issue_unit(6);
  }

  // ---- The loop. Iteration j = GEMM1 of chunk j + 1 (unit 2 j + 2, steps 0 .. NT - 1; past the last chunk a stale
  // unit and zero biases, result unused) and GEMM2 of chunk j (unit 2 j + 1, steps NT .. NF - 1), with the GELU of
  // chunk j's second half and of chunk j + 1's first half spread over the steps.
  // The fragment reads are ONE continuous stream LOOK steps ahead of the MFMAs, across iterations: the last LOOK
  // steps of an iteration already read the next iteration's biases and first fragments, so the LDS pipeline never
  // drains. One barrier per iteration, at step SYNC, before the first read of the next iteration's units. When every
  // wave has reached it,
  //  (a) units 2 j + 3 and 2 j + 4 have landed (each wave waits for its own pieces before the barrier; only the
  //      pieces of unit 2 j + 6, issued last, may still be in flight), so the next iteration may read them, and
  //  (b) nobody reads unit 2 j - 1 (finished in iteration j - 1) or unit 2 j + 2 (finished at step NT - 1) any
  //      more: their slots take units 2 j + 5 (pieces issued from here to step NT - LOOK of the next iteration;
  //      it must land before the next barrier) and 2 j + 8 (issued in the steps after those).
  // Step order: wait for this step's fragment, issue the read LOOK ahead, the step's GELU share, the MFMA: on one
  // wave per SIMD that order hides about four vector operations per step; read - wait - MFMA - vector exposes them.
  // A pair of pieces inside the loop costs four instructions: M0 = LDS base of the wave's part of the unit, a wait
  // state, and two loads whose immediate offsets tell the pieces apart; the unit's bases are scalars.
  // In the loop wave w moves the PER_UNIT consecutive fragments w PER_UNIT .. of a unit (global and LDS layouts of a
  // unit are the same bytes): the instruction's immediate offset moves the global address and the LDS destination
  // together, so four pieces share one lane-offset register.
  constexpr int NVO = (PER_UNIT + 3) / 4;
  unsigned voff[NVO];
#pragma unroll
  for (int k = 0; k < NVO; ++k) {
    voff[k] = (unsigned)(lane * 16 + (wave * PER_UNIT + 4 * k) * 1024);
    asm volatile("" : "+v"(voff[k]));
  }
  const unsigned ldsw = (unsigned)(size_t)(lds_void*)smem + wave * (PER_UNIT * 1024);
  // pieces k and k + 1 (k even: both under one M0 value)
  auto dma_pair = [&](const unsigned char* unit_src, unsigned unit_lds, int k) {
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 offset:%4\n\t"
                 "global_load_lds_dwordx4 %2, %3 offset:%5"
                 ::"s"(unit_lds), "i"((k >> 2) * 4096), "v"(voff[k >> 2]), "s"(unit_src), "i"((k & 3) * 1024),
                   "i"((k & 3) * 1024 + 1024)
                 : "memory", "m0", "scc");
  };
  bf16x8 wf[NF];
  f32x4 bq[4];
  // start the stream for iteration 0: fragments 0 .. LOOK - 1 of unit 2
  {
    unsigned W1b = lds0 + 2 * UNIT;
    asm volatile("" : "+v"(W1b));
#pragma unroll
    for (int i = 0; i < LOOK; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i]) : "v"(W1b), "i"(i * 1024));
  }

  // The ring has six slots and an iteration advances by two units: the slot numbers repeat every three iterations, so
  // the body is instantiated for j % 3 = 0, 1, 2 and every slot offset is a constant (computed from j they cost about
  // 35 scalar and vector instructions per iteration, a tenth of what a wave issues there).
  static_assert(RING == 6, "the three-iteration period below assumes six slots, two units per iteration");
  auto chunk_iteration = [&](const int j, auto phase) {
    constexpr int PH = decltype(phase)::value;                 // j % 3
    unsigned W1b = lds0 + ((2 * PH + 2) % RING) * UNIT;        // this iteration's GEMM1 unit
    unsigned W2b = lds0 + ((2 * PH + 1) % RING) * UNIT;        // this iteration's GEMM2 unit
    unsigned W1n = lds0 + ((2 * PH + 4) % RING) * UNIT;        // the next iteration's GEMM1 unit
    unsigned bbn = ldsb + 128 * (j + 2);                      // the next iteration's biases (zeros past the end)
    asm volatile("" : "+v"(W1b), "+v"(W2b), "+v"(W1n), "+v"(bbn));
    // global and LDS bases of the three units this iteration moves pieces of (scalar, once per iteration)
    const unsigned char* const src_odd = p.wpack + (size_t)(2 * j + 3) * UNIT;
    const unsigned char* const src_even = p.wpack + (size_t)(2 * j + 6) * UNIT;
    const unsigned char* const src_next = p.wpack + (size_t)(2 * j + 5) * UNIT;
    const unsigned lds_odd = ldsw + ((2 * PH + 3) % RING) * UNIT;
    const unsigned lds_even = ldsw + ((2 * PH + 6) % RING) * UNIT;
    const unsigned lds_next = ldsw + ((2 * PH + 5) % RING) * UNIT;
    const bool odd_rest = j >= 1 && 2 * j + 3 < nunits;       // pieces 2.. of unit 2 j + 3 (0 and 1 went out last iteration)
    const bool even_all = j >= 1 && 2 * j + 6 < nunits;       // unit 2 j + 6 (unit 6 went out before the loop)
    const bool odd_next = 2 * j + 5 < nunits;                 // pieces 0 and 1 of unit 2 j + 5
    // Nothing stands between the last MFMA of an iteration and the first of the next: the finished hidden tile is
    // copied out of its accumulator (hn_next -> hn_cur) during the first steps of the second GEMM, and the accumulator
    // takes the next chunk's biases during later ones (as a block at the head of the iteration the 16 reads, 16 writes
    // and the wait state behind them left the matrix core idle for about 130 cycles per iteration).
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      if (i == SYNC) {
        // own pieces of units 2 j + 3 and 2 j + 4 landed; unit 2 j + 6's (or unit 6's, before the loop) may fly
        if (2 * j + 6 < nunits) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_UNIT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      // DMA_GROUP pieces back to back (the stream into the CU runs at 17 B/clk with one piece in flight per wave, 23
      // with two, 24.7 at best: tools_prof/micro/stream_rate.hip); the two pieces of the next odd unit wait for the
      // barrier at step SYNC
      static_assert(DMA_GROUP == 2 && PER_UNIT % 2 == 0 && (SYNC / DMA_EVERY) % 2 == 0, "pieces go out in pairs under one M0");
#pragma unroll
      for (int slot = 0; slot < 2 * PER_UNIT; slot += 2) {
        const int at = slot >= SYNC / DMA_EVERY ? SYNC : (slot / DMA_GROUP) * (DMA_GROUP * DMA_EVERY);
        if (at != i) continue;
        if (slot >= SYNC / DMA_EVERY) {
          if (odd_next) dma_pair(src_next, lds_next, slot - SYNC / DMA_EVERY);
        } else if (slot < PER_UNIT - 2) {
          if (odd_rest) dma_pair(src_odd, lds_odd, slot + 2);
        } else {
          if (even_all) dma_pair(src_even, lds_even, slot - (PER_UNIT - 2));
        }
      }
      // wait for fragment i: younger reads in flight are the LOOK - 1 behind it, plus, in the tail, the four bias reads
      // (the four bias reads go out at step BIAS_READ, in front of that step's fragment read: they are younger than
      // the fragments of steps BIAS_READ + 1 .. BIAS_READ + LOOK - 1 and older than the one of step BIAS_READ + LOOK,
      // whose wait covers them)
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(wf[i]) : "i"(i > BIAS_READ && i < BIAS_READ + LOOK ? LOOK + 3 : LOOK - 1));
      if (i == BIAS_READ) {
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[q]) : "v"(bbn), "i"(32 * q));
      }
      if (i == BIAS_READ + LOOK) asm volatile("" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));     // landed
      {
        // fragment i + LOOK of the stream: GEMM1 unit, GEMM2 unit, then the next iteration's GEMM1 unit
        const unsigned base = i >= SYNC ? W1n : (i + LOOK < NT ? W1b : W2b);
        asm volatile("ds_read_b128 %0, %1 offset:%2"
                     : "=v"(wf[i + LOOK < NF ? i + LOOK : i + LOOK - NF])
                     : "v"(base), "i"((i + LOOK < NT ? i + LOOK : i + LOOK < NF ? i + LOOK - NT : i + LOOK - NF) * 1024));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i < NT) {
        // second half of GELU(hn_cur) -> hb[1], beside GEMM1
#pragma unroll
        for (int g = GH + i * GH / NT; g < GH + (i + 1) * GH / NT; ++g) gelu_part(g, hn_cur, hb[1]);
      } else {
        const int k = i - NT;
        // the finished tile leaves its accumulator: registers 0 .. 7 (first half of the GELU, needed from here on) in
        // the first two steps, 8 .. 15 in the steps behind them, PER registers per step
        if (k < 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) hn_cur[4 * k + e] = hn_next[4 * k + e];
        } else if (k < COPY_STEPS) {
#pragma unroll
          for (int e = 0; e < PER; ++e) hn_cur[8 + PER * (k - 2) + e] = hn_next[8 + PER * (k - 2) + e];
        }
        if (k < COPY_STEPS) asm volatile("" : "+v"(hn_cur));       // the copy happens in this step
        // then the accumulator takes the next chunk's biases, PER registers per step
        if (k >= BIAS_AT && k < BIAS_AT + 16 / PER) {
#pragma unroll
          for (int e = PER * (k - BIAS_AT); e < PER * (k - BIAS_AT + 1); ++e) hn_next[e] = bq[e / 4][e % 4];
          asm volatile("" : "+a"(hn_next));                        // written into the accumulator in this step
        }
        // first half of GELU(the copied tile) -> hb0_next, beside GEMM2
#pragma unroll
        for (int g = k * GH / NT; g < (k + 1) * GH / NT; ++g) gelu_part(g, hn_cur, hb0_next);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i < NT) {
        hn_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xln[i], hn_next, 0, 0, 0);
      } else {
        const int T = (i - NT) >> 1, kk = (i - NT) & 1;
        acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[i]), __builtin_bit_cast(f16x8, hb[kk]),
                                                        acc[T], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);     // nothing moves between fragment steps
    }
    hb[0] = hb0_next;
    asm volatile("" : "+v"(hb[0]), "+v"(hb[1]));
  };
  {
    int j = 0;
#pragma nounroll
    for (; j + 3 <= nch; j += 3) {
      chunk_iteration(j, std::integral_constant<int, 0>{});
      chunk_iteration(j + 1, std::integral_constant<int, 1>{});
      chunk_iteration(j + 2, std::integral_constant<int, 2>{});
    }
    if (j < nch) chunk_iteration(j, std::integral_constant<int, 0>{});
    if (j + 1 < nch) chunk_iteration(j + 1, std::integral_constant<int, 1>{});
  }
  // the reads issued for the iteration after the last one
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // + b2
#pragma unroll
  for (int T = 0; T < NT32; ++T)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *(lds_f4)(size_t)((unsigned)(size_t)(lds_void*)sb2 + (32 * T + 8 * q + 4 * h) * 4);
      acc[T][4 * q + 0] += b[0]; acc[T][4 * q + 1] += b[1]; acc[T][4 * q + 2] += b[2]; acc[T][4 * q + 3] += b[3];
    }
}

// ---------------------------------------------------------------------------------------------
// Attention phase for one sequence: acc (in) = the wave's 32 residual rows (token 32 wave + c of the
// sequence), acc (out) = rows + Wo MHA(LN(rows)) + bias. The rows never leave the accumulation
// registers: every head adds its slice of the output projection to them as soon as its attention
// output exists.
// smem: 3 units + K and V fragments (2 x 256 dh bytes... 128 keys) + H dh + 128 + 2 d floats.
// H heads of DH features each; H DH may exceed d (heads zero-padded to a width the MFMA tiles divide: the padded
// q / k features add nothing to the scores, the padded v features meet zero columns of Wo).
struct AttnPhase {
  const unsigned char* wpack;   // packed weight stream (bd_attn_pack_weights)
  const float* bq;              // [H dh]: q bias with LayerNorm beta folded, scaled like the q rows
  const float* bo;              // [d] output bias
  const float* bo_fold;         // [d] output bias + Wo * (v bias with LayerNorm beta folded)
  const unsigned char* keymask; // this SEQUENCE's byte mask [L] (non-zero = visible) or null
  int L;
  float eps;
  // the phase's tables (H dh floats of q biases, 128 floats of key mask values, 2 d floats of output biases) at a
  // caller's place in LDS, already filled; null: behind the K / V fragments, filled here
  float* tables = nullptr;
  // this WAVE's 128 key mask values in LDS when the workgroup's waves do not all see the same keys (several short
  // sequences sharing the workgroup: a wave's table hides the other sequences' keys); null: the phase's own table
  const float* wave_mask = nullptr;
};

template <int NT, int DH, int H>
__device__ __forceinline__ void attn_phase(const AttnPhase& p, unsigned char* smem, f32x16 (&acc)[NT / 2]) {
  constexpr int D = NT * 16;
  constexpr int NT32 = NT / 2;
  constexpr int AW = H * DH;                  // width of q, k and v
  constexpr int DT = DH / 32;                 // 32-row tiles of a head's q / k / v
  constexpr int KBO = DH / 16;                // 16-wide k-blocks of a head
  constexpr int UNIT = NT * 1024;
  constexpr int RING = 3;
  constexpr int PER_UNIT = NT / 4;
  constexpr int NUNITS = 4 * H * DT;             // per head: DT units each of q, k, v and of the head's Wo slice
  constexpr int WO_TILES = NT / KBO;             // output tiles covered by one unit of a Wo slice
  constexpr int KV_BYTES = 128 * DH * 2;
  constexpr int LOOK = NT >= 16 ? 6 : 4;      // fragment reads in flight (6 measures 0.6 % faster than 8, 4 and 12 slower)
  constexpr int MID = NT / 2;                 // MFMA index of a unit's barrier (a third of the way: no difference)
  static_assert(MID <= NT - LOOK, "the barrier must precede the first read of the next unit");
  static_assert(NT % 4 == 0 && DH % 32 == 0 && NT % KBO == 0 && AW >= D, "unsupported shape");
  unsigned char* const sK = smem + RING * UNIT;
  unsigned char* const sV = sK + KV_BYTES;
  float* const sbias = p.tables ? p.tables : (float*)(sV + KV_BYTES);        // H dh floats (q bias)
  float* const smask = sbias + AW;                 // 128 floats: 0 (visible) or -1e30
  float* const sbo = smask + 128;                  // 2 D floats: output bias without / with the folded v bias

  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));     // as in the feed-forward phase
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;
  // A unit moves as PER_UNIT 1-KiB pieces per wave; the unit's global and LDS bases are scalars.
  // In the loop wave w moves the PER_UNIT consecutive fragments w PER_UNIT .. of a unit (global and LDS layouts of a
  // unit are the same bytes): the instruction's immediate offset moves the global address and the LDS destination
  // together, so four pieces share one lane-offset register.
  constexpr int NVO = (PER_UNIT + 3) / 4;
  unsigned voff[NVO];
#pragma unroll
  for (int k = 0; k < NVO; ++k) {
    voff[k] = (unsigned)(lane * 16 + (wave * PER_UNIT + 4 * k) * 1024);
    asm volatile("" : "+v"(voff[k]));
  }
  const unsigned ldsw = (unsigned)(size_t)(lds_void*)smem + wave * (PER_UNIT * 1024);
  auto dma_piece = [&](const unsigned char* unit_src, unsigned unit_lds, int k) {
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 offset:%4"
                 ::"s"(unit_lds), "i"((k >> 2) * 4096), "v"(voff[k >> 2]), "s"(unit_src), "i"((k & 3) * 1024)
                 : "memory", "m0", "scc");
  };
  // pieces k and k + 1 (k even: both under one M0 value)
  auto dma_pair = [&](const unsigned char* unit_src, unsigned unit_lds, int k) {
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 offset:%4\n\t"
                 "global_load_lds_dwordx4 %2, %3 offset:%5"
                 ::"s"(unit_lds), "i"((k >> 2) * 4096), "v"(voff[k >> 2]), "s"(unit_src), "i"((k & 3) * 1024),
                   "i"((k & 3) * 1024 + 1024)
                 : "memory", "m0", "scc");
  };
  auto issue_unit = [&](int u) {
    const unsigned char* const unit_src = p.wpack + (size_t)u * UNIT;
    const unsigned unit_lds = ldsw + (u % RING) * UNIT;
#pragma unroll
    for (int k = 0; k < PER_UNIT; ++k) dma_piece(unit_src, unit_lds, k);
  };

  // ---- LayerNorm of the wave's 32 rows -> bf16 operands xln[kb] (row / column = token c,
  //      k = feature 32 (kb / 2) + 8 (2 (kb % 2) + e / 4) + 4 h + e % 4)
  bf16x8 xln[NT];
  {
    issue_unit(0);
    issue_unit(1);
    if (p.tables == nullptr) {
      for (int i = tid; i < AW; i += 256) sbias[i] = p.bq[i];
      for (int i = tid; i < 2 * D; i += 256) sbo[i] = i < D ? p.bo[i] : p.bo_fold[i - D];
      if (tid < 128) {
        const bool vis = tid < p.L && (p.keymask == nullptr || p.keymask[tid] != 0);
        smask[tid] = vis ? 0.f : -1e30f;
      }
    }
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
  const unsigned ldsK = (unsigned)(size_t)(lds_void*)sK + lane * 16;
  const unsigned ldsV = (unsigned)(size_t)(lds_void*)sV + lane * 16;
  const unsigned ldsB = (unsigned)(size_t)(lds_void*)sbias;
  const unsigned ldsM = (unsigned)(size_t)(lds_void*)(p.wave_mask ? p.wave_mask : smask) + 16 * h;

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
  auto mid_sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // One unit: NT fragments -> NT MFMAs, step(i, fragment) issues the MFMA of fragment i. The reads of the first LOOK
  // fragments of the following unit go out during the tail (after the last unit they fetch stale bytes nobody uses).
  // LDS offsets of the ring slots of units u, u + 1, u + 2 and the global address of unit u + 2, carried from unit to
  // unit (rotated at the end of stream_unit): computed from u they are three divisions by three, a dependent chain of
  // multiply-high, shift and vector add in front of the first fragment read of every unit.
  static_assert(RING == 3, "the slot rotation below assumes three slots");
  unsigned ro0 = 0, ro1 = UNIT, ro2 = 2 * UNIT;
  const unsigned char* src2 = p.wpack + 2 * (size_t)UNIT;
  auto stream_unit = [&](int u, auto&& step) {
    unsigned bc = lds0 + ro0, bn = lds0 + ro1;
    asm volatile("" : "+v"(bc), "+v"(bn));
    // unit u + 2 goes out in pairs of pieces over the unit's second half: a burst of all pieces of all four waves at
    // the barrier queues up behind the CU's memory path, single pieces leave it half idle (17 B/clk against 23)
    const bool more = u + 2 < NUNITS;
    const unsigned char* const next_src = src2;
    const unsigned next_lds = ldsw + ro2;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      if (i == MID) mid_sync();
      if (i >= MID && (i - MID) % 4 == 0 && (i - MID) / 2 < PER_UNIT) {      // in pairs: two in flight per wave
        if (more) dma_pair(next_src, next_lds, (i - MID) / 2);
      }
      const int j = i + LOOK;
      if (j < NT) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[j]) : "v"(bc), "i"(j * 1024));
      else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[j - NT]) : "v"(bn), "i"((j - NT) * 1024));
      // LOOK younger reads may stay in flight
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(wf[i]) : "i"(LOOK));
      step(i, wf[i]);
    }
    const unsigned t = ro0;
    ro0 = ro1; ro1 = ro2; ro2 = t;
    src2 += UNIT;
  };
  // WEIGHT_IS_A: tile = W_frag * op (rows = weight rows), else tile = op * W_frag (rows = tokens).
  auto run_unit = [&](int u, auto weight_is_a, f32x16 t, const bf16x8 (&op)[NT]) -> f32x16 {
    constexpr bool WEIGHT_IS_A = decltype(weight_is_a)::value;
    stream_unit(u, [&](int i, const bf16x8& w) {
      if (WEIGHT_IS_A) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, op[i], t, 0, 0, 0);
      else t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[i], w, t, 0, 0, 0);
    });
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
      const f32x16 t = run_unit(u, std::true_type{}, qinit[dt], xln);
      qb[2 * dt] = pack_half(t, 0);
      qb[2 * dt + 1] = pack_half(t, 1);
      ++u;
    }
    // ---- K^T tiles: rounded halves are the A fragments (row = key) of S = K Q^T
#pragma nounroll
    for (int dt = 0; dt < DT; ++dt) {
      const f32x16 t = run_unit(u, std::true_type{}, zero16, xln);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        *(lds_frag_w)(size_t)(ldsK + ((wave * KBO + 2 * dt + kk) << 10)) = pack_half(t, kk);
      ++u;
    }
    // ---- V tiles (rows = tokens): rounded halves are the A fragments (row = dh) of O^T = V^T P^T
#pragma nounroll
    for (int dt = 0; dt < DT; ++dt) {
      const f32x16 t = run_unit(u, std::false_type{}, zero16, xln);
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
    // ---- rows += Wo[:, this head's columns] O^T: DT units of WO_TILES output tiles x KBO fragments each
#pragma unroll
    for (int w = 0; w < DT; ++w) {
      stream_unit(u, [&](int i, const bf16x8& wfrag) {
        const int T = w * WO_TILES + i / KBO;
        acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag, onew[i % KBO], acc[T], 0, 0, 0);
      });
      ++u;
    }
  }
  // the reads issued past the last unit
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- + bias
  // (three tiles' worth of bias reads in flight at a time: left alone, the compiler issues all of them first and
  // spills the registers they need)
#pragma unroll
  for (int T = 0; T < NT32; ++T) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *(lds_f4)(size_t)((unsigned)(size_t)(lds_void*)sbo + ((dead ? 0 : D) + 32 * T + 8 * q + 4 * h) * 4);
      acc[T][4 * q + 0] += b[0]; acc[T][4 * q + 1] += b[1]; acc[T][4 * q + 2] += b[2]; acc[T][4 * q + 3] += b[3];
    }
    if (T % 3 == 2) __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace bdf
