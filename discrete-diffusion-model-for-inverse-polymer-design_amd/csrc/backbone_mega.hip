// Whole denoiser forward in ONE launch, one sequence (L <= 128) per workgroup:
//   embeddings -> n_layers x [attention phase, feed-forward phase] -> final LayerNorm -> bf16 hidden.
// Sequences never interact, so a workgroup carries its sequence through every layer without any
// grid-wide synchronisation. The residual rows stay in accumulation registers from the embeddings
// to the final LayerNorm (the attention phase's output layout IS the feed-forward phase's input
// layout, and vice versa, and both phases add to the rows in place). HBM sees the ids, the weights
// and the final hidden states; between layers nothing drains, no kernel boundary, no tail.
#include <cstdlib>
#include "bd_common.h"
#include "bd_prof.h"
#include "bidiff.h"
#include "fused_phases.h"

namespace {

using namespace bdf;

constexpr int MEGA_MAX_LAYERS = 32;

struct MegaLayer {
  const unsigned char* attn_pack;
  const float* bq;
  const float* bo;
  const float* bo_fold;
  const unsigned char* ffn_pack;
  const float* b1;
  const float* b2;
};

struct MegaArgs {
  const long long* ids;          // [B, L]
  const long long* timesteps;    // [B]
  const unsigned char* keymask;  // [B, L] or null
  const float* tok_emb;
  const float* pos_emb;
  const float* time_emb;
  const float* lnf_g;
  const float* lnf_b;
  float* xbuf;                   // unused (kept for the entry point's signature)
  unsigned short* hidden;        // [B * L, d] bf16 final-norm output
  int B, L, F, n_layers;
  float eps;
  int stagger_blocks, stagger_units;
  int tables_by_dma;             // every bias table 16-byte aligned: fetched by LDS-DMA one phase ahead
  int seg_waves;                 // waves per sequence: 4, or 2 / 1 when L <= 64 / 32 (two / four sequences share a workgroup)
  const int* wave_seq;           // [gridDim.x, 4] sequence of each wave (-1: none), equal entries adjacent; null: the rule above
  // Chosen-row mode (row_pos != null): the caller reads the hidden rows of row_pos[seq * row_cap + i] (positions, -1 =
  // none) alone, unless need_all[seq] is set. A workgroup none of whose sequences needs all rows stops behind the last
  // layer's attention phase and leaves the residual rows of the chosen positions in xsel[seq * row_cap + i, :]; the
  // tail kernel runs the last feed-forward block and the final LayerNorm on those rows.
  const int* row_pos;
  unsigned char* need_all;
  float* xsel;
  int row_cap;
  MegaLayer layers[MEGA_MAX_LAYERS];
};

// Clock diagnostics (bd_clock_stamps_set): while a buffer is set, workgroup b < cap leaves s_memtime / s_memrealtime at
// its start in buf[4 b], [4 b + 1] and at its end in [4 b + 2], [4 b + 3]. The buffer lives in a device variable that
// is read where a stamp is taken (not a kernel argument: nothing of it stays in registers across the kernel).
struct ClockStamps { unsigned long long* buf; int cap; };
__device__ ClockStamps g_clock_stamps = {nullptr, 0};

__device__ __forceinline__ void clock_stamp(int which) {
#ifdef BD_AB_NO_CLOCK_STAMPS      // comparison builds only (tools_prof/ab_mega.py)
  return;
#endif
  if (threadIdx.x != 0) return;
  const volatile ClockStamps* cs = &g_clock_stamps;
  unsigned long long* buf = cs->buf;
  if (buf == nullptr || (int)blockIdx.x >= cs->cap) return;
  buf[4 * (size_t)blockIdx.x + which] = __builtin_amdgcn_s_memtime();
  buf[4 * (size_t)blockIdx.x + which + 1] = __builtin_amdgcn_s_memrealtime();
}

// Final LayerNorm (affine) of the wave's 32 rows in accumulator layout -> the lane's half of its bf16 hidden row
// (orow = start of the row, null: nothing stored).
template <int NT>
__device__ __forceinline__ void final_norm_store(const f32x16 (&acc)[NT / 2], const float* lnf_g, const float* lnf_b,
                                                 float eps, unsigned short* orow, int h) {
  constexpr int D = NT * 16;
  constexpr int NT32 = NT / 2;
  float s = 0.f;
#pragma unroll
  for (int T = 0; T < NT32; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[T][r];
  s += __shfl_xor(s, 32, 64);
  const float mean = s * (1.0f / D);
  float q2 = 0.f;
#pragma unroll
  for (int T = 0; T < NT32; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float dv = acc[T][r] - mean; q2 += dv * dv; }
  q2 += __shfl_xor(q2, 32, 64);
  const float rstd = rsqrtf(q2 * (1.0f / D) + eps);
  if (orow != nullptr) {
    unsigned short* o = orow + 4 * h;
#pragma unroll
    for (int T = 0; T < NT32; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 g = *(const f32x4*)(lnf_g + 32 * T + 8 * q + 4 * h);
        const f32x4 b = *(const f32x4*)(lnf_b + 32 * T + 8 * q + 4 * h);
        u32x2 w;
        w[0] = bd_pack_bf16x2((acc[T][4 * q + 0] - mean) * rstd * g[0] + b[0], (acc[T][4 * q + 1] - mean) * rstd * g[1] + b[1]);
        w[1] = bd_pack_bf16x2((acc[T][4 * q + 2] - mean) * rstd * g[2] + b[2], (acc[T][4 * q + 3] - mean) * rstd * g[3] + b[3]);
        *(u32x2*)(o + 32 * T + 8 * q) = w;
      }
  }
}

template <int NT, int DH, int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void backbone_mega_kernel(MegaArgs p) {
  constexpr int D = NT * 16;
  constexpr int NT32 = NT / 2;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, c = lane & 31;
  // Short sequences share a workgroup: the weight stream, which is what a workgroup costs, then serves several of them,
  // and no wave computes rows that only shadow a last row. By rule: sequences of at most 64 (32) tokens in twos
  // (fours). By the caller's table: wave w works on rows 32 (w - first wave of its run) .. of sequence wave_seq[w], so
  // sequences of different lengths fill the four waves (rows beyond a sequence's waves are not computed at all).
  // The sequences' keys sit side by side in the 128-key tiles; each wave's mask table hides the others' keys.
  const int L = p.L, SW = p.seg_waves, NSEG = 4 >> (SW >> 1);      // SW = 4, 2, 1
  const bool shared = p.wave_seq != nullptr || NSEG > 1;
  int sq[4], fw[4];                                      // per wave: sequence (-1: none), first wave of its run
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    sq[w] = p.wave_seq ? p.wave_seq[4 * (size_t)blockIdx.x + w] : (int)blockIdx.x * NSEG + (w >> (SW >> 1));
    if (sq[w] < 0 || sq[w] >= p.B) sq[w] = -1;           // the last workgroup may have fewer sequences than segments
  }
  fw[0] = 0;
#pragma unroll
  for (int w = 1; w < 4; ++w) fw[w] = (sq[w] >= 0 && sq[w] == sq[w - 1]) ? fw[w - 1] : w;
  auto of_wave = [](const int (&a)[4], int w) { return w == 0 ? a[0] : w == 1 ? a[1] : w == 2 ? a[2] : a[3]; };
  bool rows_only = p.row_pos != nullptr;         // workgroup-uniform: the last feed-forward block is the tail kernel's
  if (rows_only) {
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (sq[w] >= 0 && p.need_all[sq[w]] != 0) rows_only = false;
  }
  const int seq_own = of_wave(sq, wave), first = of_wave(fw, wave);
  const int seq = seq_own >= 0 ? seq_own : 0;
  const int tok = 32 * (wave - first) + c;
  const bool valid = seq_own >= 0 && tok < L;
  const int tokc = valid ? tok : L - 1;                 // lanes past the sequence shadow its last row
  const size_t row = (size_t)seq * L + tokc;
  stagger_start(p.stagger_blocks, p.stagger_units);
  clock_stamp(0);

  // ---- embeddings: x = (token + position) + timestep, straight into accumulator layout
  f32x16 acc[NT32];
  {
    const float* te = p.tok_emb + (size_t)p.ids[row] * D + 4 * h;
    const float* pe = p.pos_emb + (size_t)tokc * D + 4 * h;
    const float* me = p.time_emb + (size_t)p.timesteps[seq] * D + 4 * h;
#pragma unroll
    for (int T = 0; T < NT32; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 a = *(const f32x4*)(te + 32 * T + 8 * q);
        const f32x4 b = *(const f32x4*)(pe + 32 * T + 8 * q);
        const f32x4 m = *(const f32x4*)(me + 32 * T + 8 * q);
        acc[T][4 * q + 0] = (a[0] + b[0]) + m[0]; acc[T][4 * q + 1] = (a[1] + b[1]) + m[1];
        acc[T][4 * q + 2] = (a[2] + b[2]) + m[2]; acc[T][4 * q + 3] = (a[3] + b[3]) + m[3];
      }
#pragma unroll
    for (int T = 0; T < NT32; ++T) asm volatile("" : "+a"(acc[T]));
  }

  AttnPhase ap;
  ap.keymask = p.keymask ? p.keymask + (size_t)seq * L : nullptr;      // (read by the phase only when it fills its own tables)
  ap.L = L;
  ap.eps = p.eps;
  FfnPhase fp;
  fp.F = p.F;
  fp.eps = p.eps;

  // Both phases' tables (biases, key mask values) live behind everything the phases put into LDS and are fetched by
  // LDS-DMA one phase ahead: filled at the head of the phase that uses them, their loads from memory (and the wait for
  // the first weight pieces in front of them, loads return in order) stood between the phases twelve times per
  // sequence. The key mask values are the sequence's own and are written once.
  constexpr int AW = H * DH;
  float* const tabF = (float*)(smem + phase_tables_base(NT, DH));       // first biases (F + 64, zeros past F), second biases (d)
  float* const tabA = tabF + p.F + 64 + D;                                // q biases (H dh), key mask values (128), output biases (2 d)
  int turn = 0;
  auto fetch_attn_tables = [&](const MegaLayer& lw) {
    fetch_table(lw.bq, tabA, AW * 4, wave, lane, turn);
    fetch_table(lw.bo, tabA + AW + 128, D * 4, wave, lane, turn);
    fetch_table(lw.bo_fold, tabA + AW + 128 + D, D * 4, wave, lane, turn);
  };
  auto fetch_ffn_tables = [&](const MegaLayer& lw) {
    fetch_table(lw.b1, tabF, p.F * 4, wave, lane, turn);
    fetch_table(lw.b2, tabF + p.F + 64, D * 4, wave, lane, turn);
  };
  if (p.tables_by_dma) {
    fetch_attn_tables(p.layers[0]);
    if (threadIdx.x < 64) tabF[p.F + threadIdx.x] = 0.f;
    if (!shared) {
      if (threadIdx.x < 128) {
        const bool vis = (int)threadIdx.x < L && (ap.keymask == nullptr || ap.keymask[threadIdx.x] != 0);
        tabA[AW + threadIdx.x] = vis ? 0.f : -1e30f;
      }
    } else {
      // one table per wave behind the attention tables: key j of the tiles belongs to wave j / 32's sequence, and a wave
      // sees it if both are in the same run of waves
      float* const tabM = tabA + AW + 128 + 2 * D;
      if (threadIdx.x < 128) {
        const int kw = threadIdx.x >> 5;
        const int ksq = of_wave(sq, kw), kfw = of_wave(fw, kw);
        const int kpos = 32 * (kw - kfw) + ((int)threadIdx.x & 31);
        const bool kvis = ksq >= 0 && kpos < L && (p.keymask == nullptr || p.keymask[(size_t)ksq * L + kpos] != 0);
#pragma unroll
        for (int w = 0; w < 4; ++w) tabM[128 * w + threadIdx.x] = (kvis && kfw == fw[w]) ? 0.f : -1e30f;
      }
      ap.wave_mask = tabM + 128 * wave;
    }
    ap.tables = tabA;
    fp.tables = tabF;
  }

#pragma nounroll
  for (int l = 0; l < p.n_layers; ++l) {
    const MegaLayer& lw = p.layers[l];
    ap.wpack = lw.attn_pack; ap.bq = lw.bq; ap.bo = lw.bo; ap.bo_fold = lw.bo_fold;
    fp.wpack = lw.ffn_pack; fp.b1 = lw.b1; fp.b2 = lw.b2;
    const bool stop_here = rows_only && l + 1 == p.n_layers;
    // every wave has left the previous phase's LDS; no memory operation of this wave is pending
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.tables_by_dma && !stop_here) fetch_ffn_tables(lw);   // nobody reads the feed-forward tables before the next barrier
    attn_phase<NT, DH, H>(ap, smem, acc);
    if (stop_here) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.tables_by_dma && l + 1 < p.n_layers) fetch_attn_tables(p.layers[l + 1]);    // the attention phase is done with its own
    ffn_phase<NT>(fp, smem, acc);
  }

  if (rows_only) {
    // the residual rows of the chosen positions, as they stand behind the last attention block, for the tail kernel
    if (valid) {
      int slot = -1;
      for (int i = 0; i < p.row_cap; ++i)
        if (p.row_pos[(size_t)seq * p.row_cap + i] == tok) slot = seq * p.row_cap + i;
      if (slot >= 0) store_rows(acc, p.xsel + (size_t)slot * D + 4 * h);
    }
    clock_stamp(2);
    return;
  }
  if (p.row_pos != nullptr && threadIdx.x < 4) {
    // this workgroup finishes every row of all its sequences because one of them needs that: the others are marked
    // (2), so that the tail kernel leaves their listed rows alone
    const int w = threadIdx.x;
    int sw = p.wave_seq ? p.wave_seq[4 * (size_t)blockIdx.x + w] : (int)blockIdx.x * NSEG + (w >> (SW >> 1));
    if (sw >= 0 && sw < p.B && p.need_all[sw] == 0) p.need_all[sw] = 2;
  }

  // ---- final LayerNorm (affine) -> bf16 hidden rows
  final_norm_store<NT>(acc, p.lnf_g, p.lnf_b, p.eps, valid ? p.hidden + row * D : nullptr, h);
  clock_stamp(2);
}

// The last feed-forward block and the final LayerNorm for the chosen rows that backbone_mega_kernel left in xsel
// (chosen-row mode): 128 slots per workgroup, a wave = 32 slots, each an independent row (of any sequence). A row goes
// through the same arithmetic as inside the single-launch kernel, so its hidden row carries the same bits.
struct TailArgs {
  const float* xsel;             // [n_slots, d] residual rows behind the last attention block
  const int* row_pos;            // [n_slots] position inside sequence slot / row_cap, -1: empty slot
  const unsigned char* need_all; // [B] non-zero: sequences whose rows the main kernel finished itself
  unsigned short* hidden;        // [B * L, d]
  const float* lnf_g;
  const float* lnf_b;
  FfnPhase ph;
  int n_slots, row_cap, L;
};

template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mega_tail_kernel(TailArgs p) {
  constexpr int D = NT * 16;
  constexpr int NT32 = NT / 2;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int slot = blockIdx.x * 128 + wave * 32 + c;
  int pos = -1, b = 0;
  if (slot < p.n_slots) {
    b = slot / p.row_cap;
    pos = p.need_all[b] != 0 ? -1 : p.row_pos[slot];
  }
  if (!__syncthreads_or(pos >= 0)) return;
  f32x16 acc[NT32];
  {
    const float* xrow = p.xsel + (size_t)(pos >= 0 ? slot : 0) * D + 4 * h;
#pragma unroll
    for (int T = 0; T < NT32; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = *(const f32x4*)(xrow + 32 * T + 8 * q);
        if (pos < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};          // empty slots compute on zeros, not on stale memory
        acc[T][4 * q + 0] = v[0]; acc[T][4 * q + 1] = v[1]; acc[T][4 * q + 2] = v[2]; acc[T][4 * q + 3] = v[3];
      }
#pragma unroll
    for (int T = 0; T < NT32; ++T) asm volatile("" : "+a"(acc[T]));
  }
  ffn_phase<NT>(p.ph, smem, acc);
  final_norm_store<NT>(acc, p.lnf_g, p.lnf_b, p.ph.eps, pos >= 0 ? p.hidden + ((size_t)b * p.L + pos) * D : nullptr, h);
}

template <int NT, int DH, int H = NT * 16 / DH>
int launch_mega(const MegaArgs& a, int n_blocks, hipStream_t stream) {
  const size_t la = attn_phase_lds(NT, DH, H), lf = ffn_phase_lds(NT, a.F);
  const size_t lt = phase_tables_base(NT, DH) + ffn_tables_bytes(NT, a.F) + attn_tables_bytes(NT, DH, H);
  MegaArgs b = a;
  if (lt > 160 * 1024) b.tables_by_dma = 0;              // no room for both phases' tables side by side: filled per phase
  // short sequences share a workgroup only with the prefetched tables (the per-segment mask tables sit behind them)
  const size_t lm = lt + 4 * 128 * sizeof(float);
  if (!b.tables_by_dma || lm > 160 * 1024) {
    b.seg_waves = 4;
    BD_REQUIRE(b.wave_seq == nullptr, "bd_backbone_mega_ragged: this model leaves no room for the per-wave mask tables");
  }
  const bool shared = b.seg_waves < 4 || b.wave_seq != nullptr;
  const size_t lds = b.tables_by_dma ? (shared ? lm : lt) : (la > lf ? la : lf);
  const int nseg = 4 / b.seg_waves, blocks = b.wave_seq ? n_blocks : (b.B + nseg - 1) / nseg;
  if (blocks < 4 * b.stagger_blocks) b.stagger_blocks = 0;
  BD_SET_MAX_LDS((backbone_mega_kernel<NT, DH, H>), lds);
  hipLaunchKernelGGL((backbone_mega_kernel<NT, DH, H>), dim3(blocks), dim3(256), lds, stream, b);
  return bd_check_launch("bd_backbone_mega");
}

template <int NT>
int launch_tail(const TailArgs& t, hipStream_t stream) {
  const size_t lds = ffn_phase_lds(NT, t.ph.F);
  BD_SET_MAX_LDS((mega_tail_kernel<NT>), lds);
  hipLaunchKernelGGL((mega_tail_kernel<NT>), dim3((t.n_slots + 127) / 128), dim3(256), lds, stream, t);
  return bd_check_launch("bd_backbone_forward_rows (tail)");
}

}  // namespace

extern "C" int bd_backbone_mega_supported(const bd_backbone_desc* c, int L) {
  if (c == nullptr || c->n_layers < 1 || c->n_layers > MEGA_MAX_LAYERS) return 0;
  if (!bd_attn_block_supported(c->d, c->n_heads, L) || !bd_ffn_fused_supported(c->d, c->ffn)) return 0;
  // the q | k | v width of the descriptor's weights must be the one the attention stream was packed for
  if ((c->attn_width != 0 ? c->attn_width : c->d) != bd_attn_block_width(c->d, c->n_heads)) return 0;
  if (ffn_phase_lds(c->d / 16, c->ffn) > 160 * 1024) return 0;
  for (int i = 0; i < c->n_layers; ++i) {
    const bd_layer_weights& lw = c->layers[i];
    if (!lw.attn_pack || !lw.b_q_attn || !lw.b_o_fold || !lw.b_o || !lw.ffn_pack || !lw.b_fc1_ln || !lw.b_fc2) return 0;
  }
  return 1;
}

namespace {
bool mega_tables_by_dma(const bd_backbone_desc& c) {
  const char* e = getenv("BD_MEGA_TABLES_DMA");
  if (e && atoi(e) == 0) return false;
  size_t bits = 0;
  for (int i = 0; i < c.n_layers; ++i) {
    const bd_layer_weights& lw = c.layers[i];
    bits |= (size_t)lw.b_q_attn | (size_t)lw.b_o | (size_t)lw.b_o_fold | (size_t)lw.b_fc1_ln | (size_t)lw.b_fc2;
  }
  return (bits & 15) == 0 && c.ffn % 4 == 0;
}

// Chosen rows of a launch (all null / 0: every row is computed)
struct RowChoice {
  const int* row_pos;
  unsigned char* need_all;
  float* xsel;
  int row_cap;
};

int mega_impl(const bd_backbone_desc* desc, const long long* ids, const long long* timesteps,
              const unsigned char* keymask, const int* wave_seq, int n_blocks, int B, int L, float* xbuf,
              void* hidden_out, void* stream_, const RowChoice& rows_sel = RowChoice{nullptr, nullptr, nullptr, 0}) {
  hipStream_t stream = (hipStream_t)stream_;
  BD_REQUIRE(desc && ids && timesteps && hidden_out, "bd_backbone_mega: null operand");
  BD_REQUIRE(B > 0, "bd_backbone_mega: B=%d", B);
  BD_REQUIRE(bd_backbone_mega_supported(desc, L), "bd_backbone_mega: unsupported model shape or L=%d", L);
  const bd_backbone_desc& c = *desc;
  MegaArgs a{};
  a.ids = ids; a.timesteps = timesteps; a.keymask = keymask;
  a.tok_emb = c.tok_emb; a.pos_emb = c.pos_emb; a.time_emb = c.time_emb;
  a.lnf_g = c.lnf_g; a.lnf_b = c.lnf_b;
  a.xbuf = xbuf; a.hidden = (unsigned short*)hidden_out;
  a.B = B; a.L = L; a.F = c.ffn; a.n_layers = c.n_layers; a.eps = c.ln_eps;
  a.wave_seq = wave_seq;
  a.row_pos = rows_sel.row_pos; a.need_all = rows_sel.need_all; a.xsel = rows_sel.xsel; a.row_cap = rows_sel.row_cap;
  for (int i = 0; i < c.n_layers; ++i) {
    const bd_layer_weights& lw = c.layers[i];
    a.layers[i] = MegaLayer{(const unsigned char*)lw.attn_pack, lw.b_q_attn, lw.b_o, lw.b_o_fold,
                            (const unsigned char*)lw.ffn_pack, lw.b_fc1_ln, lw.b_fc2};
  }
  {
    a.tables_by_dma = mega_tables_by_dma(c);
    const char* e = getenv("BD_MEGA_PACK");            // 0: one sequence per workgroup whatever L (for comparisons)
    a.seg_waves = (e && atoi(e) == 0) ? 4 : L <= 32 ? 1 : L <= 64 ? 2 : 4;
  }
  {
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
      else ncu = 256;
    }
    a.stagger_blocks = (B >= 4 * ncu) ? ncu : 0;
    static int units = -1;
    if (units < 0) {
      const char* e = getenv("BD_MEGA_STAGGER");      // tuning knob: phase shift between the CUs' first workgroups
      units = e ? atoi(e) : 3;
      if (units < 0 || units > 64) units = 3;
    }
    a.stagger_units = units;
  }
  const int d = c.d, dh = bd_attn_block_width(d, c.n_heads) / c.n_heads;
  void* tok = bd_prof_on() ? bd_prof_begin(stream) : nullptr;
  int rc;
  if (d == 384 && c.n_heads == 8) rc = launch_mega<24, 64, 8>(a, n_blocks, stream);
  else if (d == 384 && dh == 128) rc = launch_mega<24, 128>(a, n_blocks, stream);
  else if (d == 384) rc = launch_mega<24, 64>(a, n_blocks, stream);
  else if (d == 256 && dh == 128) rc = launch_mega<16, 128>(a, n_blocks, stream);
  else if (d == 256) rc = launch_mega<16, 64>(a, n_blocks, stream);
  else if (dh == 64) rc = launch_mega<8, 64>(a, n_blocks, stream);
  else rc = launch_mega<8, 32>(a, n_blocks, stream);
  // algorithmic work of the layers: projections, attention products, feed-forward
  // (a launch on a wave table is charged the rows its workgroups hold, not the padded batch)
  const double rows = wave_seq ? 128.0 * n_blocks : (double)B * L;
  // (chosen-row mode is charged without the last feed-forward block, which only sequences flagged need_all run)
  const double ffn_flops = 2.0 * rows * 2.0 * d * c.ffn;
  if (tok) bd_prof_end(tok, stream, (double)c.n_layers * 2.0 * rows * (4.0 * d * d + 2.0 * L * d + 2.0 * d * c.ffn) -
                                        (rows_sel.row_pos ? ffn_flops : 0.0),
                       BD_PROF_MEGA);
  if (rc != BD_OK || rows_sel.row_pos == nullptr) return rc;
  // the last feed-forward block and the final LayerNorm of the chosen rows
  const bd_layer_weights& lw = c.layers[c.n_layers - 1];
  TailArgs t{rows_sel.xsel, rows_sel.row_pos, rows_sel.need_all, (unsigned short*)hidden_out, c.lnf_g, c.lnf_b,
             FfnPhase{(const unsigned char*)lw.ffn_pack, lw.b_fc1_ln, lw.b_fc2, c.ffn, c.ln_eps},
             B * rows_sel.row_cap, rows_sel.row_cap, L};
  bd_prof_scope prof(stream, 2.0 * 2.0 * (double)t.n_slots * d * c.ffn, BD_PROF_FFN);
  if (d == 384) return launch_tail<24>(t, stream);
  if (d == 256) return launch_tail<16>(t, stream);
  return launch_tail<8>(t, stream);
}
}  // namespace

extern "C" int bd_backbone_mega(const bd_backbone_desc* desc, const long long* ids, const long long* timesteps,
                                const unsigned char* keymask, int B, int L, float* xbuf, void* hidden_out,
                                void* stream) {
  return mega_impl(desc, ids, timesteps, keymask, nullptr, 0, B, L, xbuf, hidden_out, stream);
}

extern "C" int bd_backbone_mega_ragged_supported(const bd_backbone_desc* c, int L) {
  if (!bd_backbone_mega_supported(c, L) || !mega_tables_by_dma(*c)) return 0;
  const int nt = c->d / 16, aw = bd_attn_block_width(c->d, c->n_heads), dh = aw / c->n_heads;
  const size_t lt = phase_tables_base(nt, dh) + ffn_tables_bytes(nt, c->ffn) + attn_tables_bytes(nt, dh, c->n_heads);
  return lt + 4 * 128 * sizeof(float) <= 160 * 1024;
}

extern "C" int bd_backbone_mega_ragged(const bd_backbone_desc* desc, const long long* ids, const long long* timesteps,
                                       const unsigned char* keymask, const int* wave_seq, int n_blocks, int B, int L,
                                       void* hidden_out, void* stream) {
  BD_REQUIRE(wave_seq != nullptr && n_blocks > 0, "bd_backbone_mega_ragged: no wave table");
  BD_REQUIRE(desc && bd_backbone_mega_ragged_supported(desc, L), "bd_backbone_mega_ragged: unsupported model shape or L=%d", L);
  return mega_impl(desc, ids, timesteps, keymask, wave_seq, n_blocks, B, L, nullptr, hidden_out, stream);
}

extern "C" int bd_backbone_forward_rows_supported(const bd_backbone_desc* c, int L) {
  return bd_backbone_mega_supported(c, L);
}

extern "C" int bd_backbone_forward_rows(const bd_backbone_desc* desc, const long long* ids, const long long* timesteps,
                                        const unsigned char* keymask, const int* wave_seq, int n_blocks, int B, int L,
                                        const int* row_pos, unsigned char* need_all, int row_cap,
                                        void* workspace, long long workspace_bytes, void* hidden_out, void* stream) {
  BD_REQUIRE(desc && row_pos && need_all && workspace && hidden_out, "bd_backbone_forward_rows: null operand");
  BD_REQUIRE(B > 0 && L > 0, "bd_backbone_forward_rows: empty batch");
  BD_REQUIRE(row_cap >= 1 && row_cap <= 32, "bd_backbone_forward_rows: row_cap=%d outside 1..32", row_cap);
  BD_REQUIRE((long long)B * row_cap < (1ll << 31) / desc->d, "bd_backbone_forward_rows: too many row slots");
  BD_REQUIRE(workspace_bytes >= (long long)B * row_cap * desc->d * 4,
             "bd_backbone_forward_rows: workspace %lld < %lld bytes", workspace_bytes, (long long)B * row_cap * desc->d * 4);
  BD_REQUIRE(((size_t)workspace & 15) == 0, "bd_backbone_forward_rows: workspace must be 16-byte aligned");
  if (wave_seq != nullptr) {
    BD_REQUIRE(n_blocks > 0, "bd_backbone_forward_rows: wave table without workgroups");
    BD_REQUIRE(bd_backbone_mega_ragged_supported(desc, L), "bd_backbone_forward_rows: unsupported model shape or L=%d for a wave table", L);
  }
  BD_REQUIRE(bd_backbone_forward_rows_supported(desc, L), "bd_backbone_forward_rows: unsupported model shape or L=%d", L);
  return mega_impl(desc, ids, timesteps, keymask, wave_seq, n_blocks, B, L, nullptr, hidden_out, stream,
                   RowChoice{row_pos, need_all, (float*)workspace, row_cap});
}

extern "C" int bd_clock_stamps_set(unsigned long long* stamps, int n_workgroups) {
  BD_REQUIRE(stamps == nullptr || n_workgroups > 0, "bd_clock_stamps_set: a buffer needs a positive workgroup count");
  const ClockStamps cs{stamps, stamps ? n_workgroups : 0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_clock_stamps), &cs, sizeof(cs)) != hipSuccess)
    return bd_fail(BD_ERR_LAUNCH, "bd_clock_stamps_set: writing the device variable failed");
  return BD_OK;
}
