// One reverse-diffusion unmasking step and the post-loop repairs, one wavefront per
// sequence (gfx950, wave64). The step is the explicit-uniform restatement of the
// reference samplers' inner loop:
//   * k = ktable[n_masked] positions are chosen as the k smallest per-position keys
//     (distribution-identical to randperm(n)[:k]); ties go to the lower position;
//   * each chosen position draws argmax_v(logit[v] + gumbel(u[v])) (distribution-
//     identical to multinomial(softmax(logit))), ties to the lower v; a row with no
//     allowed entry falls back to a uniform pick among the non-special ids;
//   * draws are sequential inside a sequence and apply the same in-step constraint
//     updates as the reference.
// Two logits sources: LOGITS mode edits a full [B,L,V] fp32 tensor in place exactly as
// the reference mutates its `logits` (divide by temperature, -inf edits, boosts), which
// is what the post-loop repairs consume at t = 1; HIDDEN mode computes vocabulary
// logits only for the chosen rows from the final hidden states and the output
// projection, never materialising [B,L,V].
#include "bd_common.h"
#include "bd_rng.h"
#include "bidiff.h"
#include "bd_prof.h"

namespace {

constexpr int MAXL = BD_SAMPLE_MAX_L;
constexpr int MAXV = BD_SAMPLE_MAX_V;
constexpr int MAXD = BD_SAMPLE_MAX_D;
constexpr int MAXR = BD_SAMPLE_MAX_RING;

// per-position flags in LDS
constexpr int PF_MASK0 = 1, PF_FIXED = 2, PF_DYN_BOND = 4, PF_DYN_CLOSE = 8, PF_CLOSE = 16, PF_OPEN = 32,
              PF_BOND = 64, PF_BOOST = 128;

// Working set of one sequence in LDS, carved from dynamic shared memory with the sizes of THIS launch
// (a fixed worst-case layout of 20 KB per wave would cap a CU at 8 sequences; the small model needs 4 KB).
struct Smem {
  int* ids;                  // [L]
  unsigned int* key;         // [L]
  float* col;                // [L]
  float* row;                // [V]
  float* h;                  // [d]
  int* ringcnt;              // [R]
  short* sel;                // [L]
  unsigned char* pf;         // [L]
  unsigned char* ringforbid; // [R]
};
__host__ __device__ inline size_t smem_align4(size_t x) { return (x + 3) & ~(size_t)3; }
__host__ __device__ inline size_t smem_bytes(int L, int V, int d, int R) {
  return 4 * ((size_t)3 * L + V + d + R) + smem_align4(2 * (size_t)L) + smem_align4((size_t)L) + smem_align4((size_t)R) + 16;
}
__device__ __forceinline__ void smem_carve(Smem& s, unsigned char* raw, int L, int V, int d, int R) {
  s.ids = (int*)raw;                  raw += 4 * (size_t)L;
  s.key = (unsigned int*)raw;         raw += 4 * (size_t)L;
  s.col = (float*)raw;                raw += 4 * (size_t)L;
  s.row = (float*)raw;                raw += 4 * (size_t)V;
  s.h = (float*)raw;                  raw += 4 * (size_t)d;
  s.ringcnt = (int*)raw;              raw += 4 * (size_t)R;
  s.sel = (short*)raw;                raw += smem_align4(2 * (size_t)L);
  s.pf = raw;                         raw += smem_align4((size_t)L);
  s.ringforbid = raw;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned int lo = __shfl_xor((unsigned int)v, o, 64);
    const unsigned int hi = __shfl_xor((unsigned int)(v >> 32), o, 64);
    const unsigned long long w = ((unsigned long long)hi << 32) | lo;
    v = w < v ? w : v;
  }
  return v;
}

// (score, index) argmax over the wave, ties to the lower index; idx < 0 = no candidate.
__device__ __forceinline__ void wave_argmax(float& best, int& idx) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    const bool take = (oi >= 0) && (idx < 0 || ob > best || (ob == best && oi < idx));
    if (take) { best = ob; idx = oi; }
  }
}

// Is token v (class bits tf) forbidden at masked position l?
__device__ __forceinline__ bool forbidden_at(const bd_step_args& p, const Smem& s, int l, int v, int tf,
                                             bool cnt_forbid, bool eos_forbid, bool single_forbid) {
  // the specials: everywhere a MASK stands, or (SELFIES sample_conditional) at the non-fixed ones only
  const bool sp = !(p.variant & BD_VB_SPECIAL_UNFIXED) || !(s.pf[l] & PF_FIXED);
  if (sp && (tf & BD_TF_SPECIAL)) return true;
  if (sp && (tf & BD_TF_EOS) && eos_forbid) return true;
  if (!p.use_constraints) return false;
  if (cnt_forbid && (tf & BD_TF_COUNTED)) return true;
  if (single_forbid && v == p.count_id) return true;
  if (p.variant & BD_VB_SYNTAX) {
    const int f = s.pf[l];
    if ((tf & BD_TF_CLOSE) && (f & (PF_CLOSE | PF_DYN_CLOSE))) return true;
    if ((tf & BD_TF_OPEN) && (f & PF_OPEN)) return true;
    if ((tf & BD_TF_BOND) && (f & (PF_BOND | PF_DYN_BOND))) return true;
    if ((tf & BD_TF_RING) && s.ringforbid[p.ring_slot[v]]) return true;
  }
  return false;
}

// Start-of-step ids and position flags of sequence b into LDS.
__device__ __forceinline__ void load_ids_and_flags(const bd_step_args& p, const Smem& s, int b, int lane) {
  const int L = p.L;
  const long long* ids_g = p.ids + (size_t)b * L;
  for (int l = lane; l < L; l += 64) {
    const int id = (int)ids_g[l];
    s.ids[l] = id;
    int f = (id == p.mask_id) ? PF_MASK0 : 0;
    if ((p.variant & BD_VB_FIXED) && p.fixed && p.fixed[(size_t)b * L + l]) f |= PF_FIXED;
    s.pf[l] = (unsigned char)f;
  }
  for (int r = lane; r < p.R; r += 64) { s.ringcnt[r] = 0; s.ringforbid[r] = 0; }
  __syncthreads();
}

// The k = ktable[n_sel] selectable masked positions with the smallest keys, in draw order, into s.sel; returns k.
// A function of the sequence's ids, fixed flags, seed and step alone (the chosen-row kernels repeat it).
__device__ __forceinline__ int choose_positions(const bd_step_args& p, const Smem& s, int b, int lane, int n_sel) {
  const int L = p.L;
  const int CH = (L + 63) / 64;
  const int l0 = lane * CH;
  const uint32_t seq = (uint32_t)(p.seq_offset + b);
  int k_sel = 0;
  if (n_sel > 0) {
    k_sel = p.ktable[n_sel];
    k_sel = k_sel < 1 ? 1 : (k_sel > n_sel ? n_sel : k_sel);
    for (int l = lane; l < L; l += 64) {
      const int f = s.pf[l];
      unsigned int kb = 0xffffffffu;
      if ((f & PF_MASK0) && !(f & PF_FIXED)) {
        const float u = p.u_sel ? p.u_sel[(size_t)b * L + l]
                                : bd_uniform_select(p.seed, seq, (uint32_t)p.t, (uint32_t)l);
        kb = __float_as_uint(u) & 0x7fffffffu;
      }
      s.key[l] = kb;
    }
    __syncthreads();
    for (int round = 0; round < k_sel; ++round) {
      unsigned long long best = ~0ull;
      for (int j = 0; j < CH; ++j) {
        const int l = l0 + j;
        if (l < L) {
          const unsigned int kb = s.key[l];
          if (kb != 0xffffffffu) {
            const unsigned long long ck = ((unsigned long long)kb << 32) | (unsigned int)l;
            best = ck < best ? ck : best;
          }
        }
      }
      best = wave_min_u64(best);
      const int pos = (int)(best & 0xffffffffu);
      if (lane == 0) { s.sel[round] = (short)pos; s.key[pos] = 0xffffffffu; }
      __syncthreads();
    }
    if (p.variant & BD_VB_SORTED) {
      // ascending-position order: rank each chosen position among the chosen
      for (int i = lane; i < k_sel; i += 64) s.key[i] = (unsigned int)s.sel[i];
      __syncthreads();
      for (int i = lane; i < k_sel; i += 64) {
        const unsigned int mine = s.key[i];
        int rank = 0;
        for (int j = 0; j < k_sel; ++j) rank += (s.key[j] < mine);
        s.sel[rank] = (short)mine;
      }
      __syncthreads();
    }
  }
  return k_sel;
}

// ---- chosen-row logits ahead of the draws (HIDDEN mode with a row workspace) -----------------------------
// Which positions a step draws is fixed before its first draw, and a row's raw logits do not depend on the
// draws. So the first row_cap rows of every sequence are gathered into one (rows x V x d) product on the matrix
// cores, where 32 rows share each pass over the output projection, instead of one wave re-reading the whole
// projection for every row it draws (V * d * 2 bytes each: 617 KB at V = 803, d = 384).

// row_pos[b * cap + i] = position of sequence b's i-th draw of this step, or -1.
// need_all (optional) [B]: 1 when the step reads hidden rows of sequence b beyond those: more draws than row_cap, or
// the urgency rule, which reads the boosted token's logit at every position (the condition of sample_step_kernel).
__global__ __launch_bounds__(64) void choose_rows_kernel(bd_step_args p, unsigned char* need_all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem s;
  const int lane = threadIdx.x, b = blockIdx.x, L = p.L;
  smem_carve(s, smem_raw, L, p.V, p.d, p.R > 0 ? p.R : 1);
  load_ids_and_flags(p, s, b, lane);
  int n_sel = 0, n_mask = 0, total = 0;
  for (int l = lane; l < L; l += 64) {
    const int f = s.pf[l];
    n_sel += (f & PF_MASK0) && !(f & PF_FIXED);
    if (f & PF_MASK0) n_mask++;
    else total += p.contrib[s.ids[l]];
  }
  n_sel = bd_wave_sum_i(n_sel);
  const int k_sel = choose_positions(p, s, b, lane, n_sel);
  for (int i = lane; i < p.row_cap; i += 64) p.row_pos[(size_t)b * p.row_cap + i] = i < k_sel ? (int)s.sel[i] : -1;
  if (need_all != nullptr) {
    n_mask = bd_wave_sum_i(n_mask);
    total = bd_wave_sum_i(total);
    bool all = k_sel > p.row_cap;
    if (p.use_constraints && (p.variant & BD_VB_URGENCY)) {
      const int needed = 2 - total;
      if (needed > 0 && n_mask > 0 && n_mask <= 5 * needed) all = true;
    }
    if (lane == 0) need_all[b] = all ? 1 : 0;
  }
}

// row_logits[slot][v] = hidden[b, row_pos[slot], :] . wout[v, :] for the occupied slots. A workgroup = 4 waves =
// 4 x 32 slots over a quarter of the vocabulary (blockIdx.y); every wave keeps its 32 hidden rows as MFMA operands
// in registers for the whole pass. The projection rows of a vocabulary tile (32 x d bf16, contiguous in memory)
// are fetched once per workgroup with full-line loads into a double-buffered LDS tile and read from there as
// operands by the four waves: fetched straight from memory in operand order, every load instruction touches 32
// lines for 32 bytes each and the CU's line rate, not the matrix cores, sets the pace (3 x slower).
template <int NK>   // d / 16
__global__ __launch_bounds__(256) void row_logits_kernel(bd_step_args p, int VS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wtile[];
  constexpr int ROWB = NK * 32 + 16;         // bytes per projection row in LDS (+16: rows r and r + 1 start 4 banks apart)
  constexpr int TILEB = 32 * ROWB;
  constexpr int CHUNKS = 64 * NK;            // 16-byte pieces per tile
  constexpr int PER = (CHUNKS + 255) / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, kh = lane >> 5;
  const int d = p.d, V = p.V;
  const long long n_slots = (long long)p.B * p.row_cap;
  const long long slot = ((long long)blockIdx.x * 4 + wave) * 32 + r;
  int pos = -1;
  long long b = 0;
  if (slot < n_slots) { pos = p.row_pos[slot]; b = slot / p.row_cap; }
  if (!__syncthreads_or(pos >= 0)) return;
  const bool active = __ballot(pos >= 0) != 0ull;
  const unsigned short* hrow = (const unsigned short*)p.hidden + ((size_t)b * p.L + (pos >= 0 ? pos : 0)) * d + 8 * kh;
  bf16x8 hf[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) hf[ks] = *(const bf16x8*)(hrow + 16 * ks);
  const int n_tiles = (V + 31) / 32;
  const int t0 = n_tiles * blockIdx.y / gridDim.y, t1 = n_tiles * (blockIdx.y + 1) / gridDim.y;
  const unsigned char* wout = (const unsigned char*)p.wout;
  float* out = p.row_logits + (size_t)slot * VS;
  u32x4 stage[PER];
  auto fetch = [&](int t) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + 256 * i;
      if (CHUNKS % 256 == 0 || c < CHUNKS) {
        const int row = c / (2 * NK), cc = c - row * (2 * NK);
        stage[i] = *(const u32x4*)(wout + ((size_t)min(32 * t + row, V - 1) * d) * 2 + cc * 16);
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + 256 * i;
      if (CHUNKS % 256 == 0 || c < CHUNKS) {
        const int row = c / (2 * NK), cc = c - row * (2 * NK);
        *(u32x4*)(wtile + buf * TILEB + row * ROWB + cc * 16) = stage[i];
      }
    }
  };
  if (t0 < t1) fetch(t0);
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) & 1;
    stash(buf);
    __syncthreads();
    if (t + 1 < t1) fetch(t + 1);
    if (!active) continue;
    const unsigned char* wfrag = wtile + buf * TILEB + r * ROWB + kh * 16;
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < NK; ks += 2) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(wfrag + 32 * ks), hf[ks], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(wfrag + 32 * ks + 32), hf[ks + 1], acc1, 0, 0, 0);
    }
    // lane: slot r, vocabulary entries 32 t + 8 g + 4 kh + (0..3)
    if (pos >= 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 o = {acc0[4 * g] + acc1[4 * g], acc0[4 * g + 1] + acc1[4 * g + 1], acc0[4 * g + 2] + acc1[4 * g + 2],
                   acc0[4 * g + 3] + acc1[4 * g + 3]};
        *(f32x4*)(out + 32 * t + 8 * g + 4 * kh) = o;
      }
    }
  }
}

__global__ __launch_bounds__(64) void sample_step_kernel(bd_step_args p, int VS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem s;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int L = p.L, V = p.V, d = p.d;
  smem_carve(s, smem_raw, L, V, d, p.R > 0 ? p.R : 1);
  const int CH = (L + 63) / 64;  // contiguous positions per lane
  const int l0 = lane * CH;
  long long* ids_g = p.ids + (size_t)b * L;
  const bool cons = p.use_constraints != 0;
  const bool syntax = cons && (p.variant & BD_VB_SYNTAX);
  const bool logits_mode = p.hidden == nullptr;
  float* lg = p.logits ? p.logits + (size_t)b * L * V : nullptr;
  const uint32_t seq = (uint32_t)(p.seq_offset + b);
  const unsigned short* hidden = (const unsigned short*)p.hidden;
  const unsigned short* wout = (const unsigned short*)p.wout;

  load_ids_and_flags(p, s, b, lane);

  // ---- per-sequence counts: masked, selectable, counted-token total
  int n_mask = 0, n_sel = 0, total = 0;
  for (int j = 0; j < CH; ++j) {
    const int l = l0 + j;
    if (l < L) {
      const int f = s.pf[l];
      if (f & PF_MASK0) { n_mask++; if (!(f & PF_FIXED)) n_sel++; }
      else total += p.contrib[s.ids[l]];
    }
  }
  n_mask = bd_wave_sum_i(n_mask);
  n_sel = bd_wave_sum_i(n_sel);
  total = bd_wave_sum_i(total);
  bool cnt_forbid = cons && total >= 2;
  bool single_forbid = false;  // BD_VB_DYN_SINGLE: the direct placeholder id alone got forbidden in this step
  const bool natural = (p.variant & BD_VB_NATURAL_EOS) != 0;
  bool eos_forbid = true;  // EOS is an ordinary forbidden special unless natural-EOS mode
  if (natural) eos_forbid = p.has_eos[b] != 0;

  // ---- SMILES syntax state from the start-of-step ids
  if (syntax) {
    int dloc = 0, mloc = 0, lastnm = -1;
    for (int j = 0; j < CH; ++j) {
      const int l = l0 + j;
      if (l < L) {
        const int id = s.ids[l];
        dloc += (int)(p.open_id >= 0 && id == p.open_id) - (int)(p.close_id >= 0 && id == p.close_id);
        if (s.pf[l] & PF_MASK0) mloc++; else lastnm = l;
        const int slot = p.ring_slot[id];
        if (slot >= 0) atomicAdd(&s.ringcnt[slot], 1);
      }
    }
    int dinc = dloc, minc = mloc, pmax = lastnm;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int dv = __shfl_up(dinc, o, 64);
      const int pv = __shfl_up(pmax, o, 64);
      const int mv = __shfl_down(minc, o, 64);
      if (lane >= o) { dinc += dv; pmax = pv > pmax ? pv : pmax; }
      if (lane + o < 64) minc += mv;
    }
    int depth = dinc - dloc;             // open - close over positions before this chunk
    int msuf = minc;                     // masks at positions >= l0
    int prev = __shfl_up(pmax, 1, 64);   // last non-mask index before this chunk
    if (lane == 0) prev = -1;
    const bool have_paren = (p.open_id >= 0) || (p.close_id >= 0);
    const bool have_bond = p.n_bond > 0;
    for (int j = 0; j < CH; ++j) {
      const int l = l0 + j;
      if (l < L) {
        const int id = s.ids[l];
        int f = s.pf[l];
        const bool is_m = (f & PF_MASK0) != 0;
        if (!is_m) prev = l;
        if (is_m) {
          if (have_paren) {
            if (p.close_id >= 0 && depth <= 0) f |= PF_CLOSE;
            if (p.open_id >= 0 && depth + 2 > msuf) f |= PF_OPEN;
          }
          if (have_bond) {
            const int pid = prev >= 0 ? s.ids[prev] : -1;
            bool inv = prev < 0;
            if (!inv) {
              inv = (p.bos_id >= 0 && pid == p.bos_id) || (p.eos_id >= 0 && pid == p.eos_id) ||
                    (p.pad_id >= 0 && pid == p.pad_id) || (p.open_id >= 0 && pid == p.open_id) ||
                    (p.close_id >= 0 && pid == p.close_id) || (p.tok_flags[pid] & BD_TF_BOND);
            }
            if (inv) f |= PF_BOND;
            if (p.open_id >= 0 && p.close_id >= 0 && pid == p.open_id) f |= PF_CLOSE;
          }
          s.pf[l] = (unsigned char)f;
          msuf--;
        }
        depth += (int)(p.open_id >= 0 && id == p.open_id) - (int)(p.close_id >= 0 && id == p.close_id);
      }
    }
    __syncthreads();
    int open_cnt = 0;
    for (int r = lane; r < p.R; r += 64) open_cnt += (s.ringcnt[r] == 1);
    open_cnt = bd_wave_sum_i(open_cnt);
    const bool forbid_new = n_mask <= open_cnt;
    for (int r = lane; r < p.R; r += 64) {
      const int cnt = s.ringcnt[r];
      s.ringforbid[r] = (cnt >= 2) || (forbid_new && cnt != 1);
    }
  }
  __syncthreads();

  // ---- LOGITS mode: the whole tensor becomes logits / temperature with the static edits
  if (logits_mode) {
    const int tot = L * V;
    for (int i = lane; i < tot; i += 64) {
      const int l = i / V, v = i - l * V;
      float x = lg[i] / p.temperature;
      if ((s.pf[l] & PF_MASK0) && forbidden_at(p, s, l, v, p.tok_flags[v], cnt_forbid, eos_forbid, false))
        x = -INFINITY;
      lg[i] = x;
    }
    __syncthreads();
  }

  // ---- SELFIES urgency boost of the placeholder when few masked slots remain
  if (cons && (p.variant & BD_VB_URGENCY)) {
    const int needed = 2 - total;
    if (needed > 0 && n_mask > 0 && n_mask <= 5 * needed) {
      const int bid = p.boost_id;
      if (!logits_mode) {
        for (int k = lane; k < d; k += 64) s.h[k] = bd_bf16_to_f32(wout[(size_t)bid * d + k]);
        __syncthreads();
      }
      float cmax = -INFINITY;
      for (int j = 0; j < CH; ++j) {
        const int l = l0 + j;
        if (l < L) {
          float x;
          if (logits_mode) {
            x = lg[(size_t)l * V + bid];
          } else {
            const unsigned short* hr = hidden + ((size_t)b * L + l) * d;
            float acc = 0.f;
            for (int k = 0; k < d; k += 8) {
              const u32x4 w = *(const u32x4*)(hr + k);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                acc += __uint_as_float(w[e] << 16) * s.h[k + 2 * e];
                acc += __uint_as_float(w[e] & 0xffff0000u) * s.h[k + 2 * e + 1];
              }
            }
            if (lg) lg[(size_t)l * V + bid] = acc;  // raw dump for replay
            x = acc / p.temperature;
          }
          s.col[l] = x;
          cmax = fmaxf(cmax, x);
        }
      }
      cmax = bd_wave_max(cmax);
      __syncthreads();
      int kk = needed < n_mask ? needed : n_mask;
      // any masked slot whose softmax-over-positions weight is non-zero in fp32? The reference's softmax keeps
      // denormals: the weight fl(fl(exp(x - max)) / Z) is positive down to about 104 - ln Z below the maximum
      double zloc = 0.0;
      for (int j = 0; j < CH; ++j) {
        const int l = l0 + j;
        if (l < L) zloc += exp((double)(s.col[l] - cmax));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long zb = (unsigned long long)__double_as_longlong(zloc);
        const unsigned int lo = __shfl_xor((unsigned int)zb, o, 64), hi = __shfl_xor((unsigned int)(zb >> 32), o, 64);
        zloc += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
      }
      int anyv = 0;
      for (int j = 0; j < CH; ++j) {
        const int l = l0 + j;
        if (l < L && (s.pf[l] & PF_MASK0)) {
          const float delta = s.col[l] - cmax;
          if (delta >= -87.0f) anyv = 1;
          else if (delta >= -104.5f && rint(exp((double)delta) * 0x1p149) > 0.5 * zloc) anyv = 1;
        }
      }
      anyv = bd_wave_sum_i(anyv);
      if (anyv > 0) {
        for (int round = 0; round < kk; ++round) {
          float best = -INFINITY; int idx = -1;
          for (int j = 0; j < CH; ++j) {
            const int l = l0 + j;
            if (l < L && (s.pf[l] & PF_MASK0) && !(s.pf[l] & PF_BOOST)) {
              const float x = s.col[l];
              if (idx < 0 || x > best) { best = x; idx = l; }
            }
          }
          wave_argmax(best, idx);
          if (idx >= 0 && lane == 0) {
            s.pf[idx] |= PF_BOOST;
            if (logits_mode) lg[(size_t)idx * V + bid] = 100.0f;
          }
          __syncthreads();
        }
      }
    }
  }

  const int k_sel = choose_positions(p, s, b, lane, n_sel);

  // ---- sequential draws
  for (int it = 0; it < k_sel; ++it) {
    const int pos = s.sel[it];
    if (s.ids[pos] != p.mask_id) continue;  // filled by an EOS termination earlier in this step
    const int pflag = s.pf[pos];
    const bool have_row = !logits_mode && p.row_logits != nullptr && it < p.row_cap;
    if (have_row) {
      const float* rl = p.row_logits + ((size_t)b * p.row_cap + it) * VS;
      for (int v = lane; v < V; v += 64) s.row[v] = rl[v];
      __syncthreads();
    }
    if (!logits_mode && !have_row) {
      const unsigned short* hr = hidden + ((size_t)b * L + pos) * d;
      for (int k = lane * 8; k < d; k += 512) {
        const u32x4 w = *(const u32x4*)(hr + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s.h[k + 2 * e] = __uint_as_float(w[e] << 16);
          s.h[k + 2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
        }
      }
      __syncthreads();
    }
    if (!logits_mode && !have_row) {
      // logits of this row: the wave walks the vocabulary, every lane multiplies its 8-feature slices of
      // the weight row with the hidden row (coalesced 16-byte loads, independent across v), a wave sum
      // completes the dot product and lane v % 64 keeps it in s.row[v]
      // four vocabulary rows per round: their loads are in flight together and the four wave sums interleave
      for (int v0 = 0; v0 < V; v0 += 4) {
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = lane * 8; k < d; k += 512) {
          u32x4 w[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) w[u] = *(const u32x4*)(wout + (size_t)min(v0 + u, V - 1) * d + k);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              part[u] += __uint_as_float(w[u][e] << 16) * s.h[k + 2 * e];
              part[u] += __uint_as_float(w[u][e] & 0xffff0000u) * s.h[k + 2 * e + 1];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
          for (int u = 0; u < 4; ++u) part[u] += __shfl_xor(part[u], o, 64);
        if (lane < 4 && v0 + lane < V) s.row[v0 + lane] = lane == 0 ? part[0] : lane == 1 ? part[1] : lane == 2 ? part[2] : part[3];
      }
      __syncthreads();
    }
    float best = -INFINITY; int idx = -1;
    for (int v = lane; v < V; v += 64) {
      float x;
      if (logits_mode) {
        x = lg[(size_t)pos * V + v];
      } else {
        const float acc = s.row[v];
        if (lg) lg[(size_t)pos * V + v] = acc;  // raw dump for replay
        x = acc / p.temperature;
        if ((pflag & PF_BOOST) && v == p.boost_id) x = 100.0f;
        if (forbidden_at(p, s, pos, v, p.tok_flags[v], cnt_forbid, eos_forbid, single_forbid)) x = -INFINITY;
      }
      if (x > -INFINITY) {
        const float u = p.u_tok ? p.u_tok[((size_t)b * L + pos) * V + v]
                                : bd_uniform_token(p.seed, seq, (uint32_t)p.t, (uint32_t)pos, (uint32_t)v);
        const float sc = x + bd_gumbel(u);
        if (idx < 0 || sc > best) { best = sc; idx = v; }
      }
    }
    wave_argmax(best, idx);
    int tok = idx;
    if (tok < 0) {
      const float u0 = p.u_tok ? p.u_tok[((size_t)b * L + pos) * V]
                               : bd_uniform_token(p.seed, seq, (uint32_t)p.t, (uint32_t)pos, 0u);
      int j = (int)(u0 * (float)p.n_ns);
      j = j < 0 ? 0 : (j >= p.n_ns ? p.n_ns - 1 : j);
      tok = p.non_special[j];
    }
    __syncthreads();
    if (lane == 0) s.ids[pos] = tok;
    __syncthreads();

    if (natural && tok == p.eos_id) {
      // the sequence ends here: pad the tail, hide it from attention, stop this row
      for (int l = pos + 1 + lane; l < L; l += 64) {
        s.ids[l] = p.pad_id;
        if (p.attn) p.attn[(size_t)b * L + l] = 0;
        if (p.keymask) p.keymask[(size_t)b * L + l] = 0;
      }
      if (lane == 0) p.has_eos[b] = 1;
      __syncthreads();
      break;
    }

    if (cons) {
      const int tf = p.tok_flags[tok];
      if ((p.variant & BD_VB_DYN_SINGLE) && (tf & BD_TF_COUNTED)) {
        // Group_SELFIES sample_conditional: only the direct placeholder id counts and gets forbidden
        if (tok == p.count_id) {
          int c = 0;
          for (int j = 0; j < CH; ++j) {
            const int l = l0 + j;
            if (l < L && s.ids[l] == p.count_id) c++;
          }
          c = bd_wave_sum_i(c);
          if (c >= 2) {
            single_forbid = true;
            if (logits_mode)
              for (int l = lane; l < L; l += 64)
                if (s.ids[l] == p.mask_id) lg[(size_t)l * V + p.count_id] = -INFINITY;
          }
        }
      } else if (tf & BD_TF_COUNTED) {
        int tot2 = 0;
        for (int j = 0; j < CH; ++j) {
          const int l = l0 + j;
          if (l < L && s.ids[l] != p.mask_id) tot2 += p.contrib[s.ids[l]];
        }
        tot2 = bd_wave_sum_i(tot2);
        if (tot2 >= 2) {
          cnt_forbid = true;
          if (logits_mode) {
            const bool all_masks = (p.variant & BD_VB_DYN_ALL_MASKS) != 0;
            for (int l = lane; l < L; l += 64) {
              if (s.ids[l] == p.mask_id && (all_masks || !(s.pf[l] & PF_FIXED)))
                for (int q = 0; q < p.n_counted; ++q) lg[(size_t)l * V + p.counted_ids[q]] = -INFINITY;
            }
          }
        }
      } else if ((p.variant & BD_VB_SYNTAX) && (tf & BD_TF_BOND)) {
        const int np = pos + 1;
        if (np < L && s.ids[np] == p.mask_id) {
          if (lane == 0) s.pf[np] |= PF_DYN_BOND;
          if (logits_mode)
            for (int q = lane; q < p.n_bond; q += 64) lg[(size_t)np * V + p.bond_ids[q]] = -INFINITY;
        }
      } else if ((p.variant & BD_VB_SYNTAX) && p.open_id >= 0 && tok == p.open_id) {
        const int np = pos + 1;
        if (np < L && s.ids[np] == p.mask_id) {
          if (lane == 0) s.pf[np] |= PF_DYN_CLOSE;
          // the reference indexes column close_id even when it is -1 (= last column)
          if (logits_mode && lane == 0) lg[(size_t)np * V + (p.close_id >= 0 ? p.close_id : V - 1)] = -INFINITY;
        }
      }
      __syncthreads();
    }
  }

  __syncthreads();
  for (int l = lane; l < L; l += 64) ids_g[l] = (long long)s.ids[l];
}

// -----------------------------------------------------------------------------------
// Post-loop repairs on the final ids with the t = 1 logits.
// -----------------------------------------------------------------------------------
struct FixSmem {
  int ids[MAXL];
  unsigned char mark[MAXL];
};

// argmax over the vocabulary row with an exclusion predicate; first maximum wins,
// an all -inf row gives 0 (torch.argmax convention).
template <typename Excl>
__device__ __forceinline__ int row_argmax(const float* row, int V, int lane, Excl excl) {
  float best = -INFINITY; int idx = -1;
  for (int v = lane; v < V; v += 64) {
    float x = row[v];
    if (excl(v)) x = -INFINITY;
    if (idx < 0 || x > best) { best = x; idx = v; }
  }
  wave_argmax(best, idx);
  return idx < 0 ? 0 : idx;
}

// Keep/choose `want` positions with the largest column value among candidates
// (cand[l] != 0), ties to the lower position; chosen positions get mark |= 2.
__device__ __forceinline__ void pick_top_positions(FixSmem& s, const float* lg, int L, int V, int col,
                                                   int want, int lane) {
  const int CH = (L + 63) / 64, l0 = lane * CH;
  for (int round = 0; round < want; ++round) {
    float best = -INFINITY; int idx = -1;
    for (int j = 0; j < CH; ++j) {
      const int l = l0 + j;
      if (l < L && s.mark[l] == 1) {
        const float x = lg[(size_t)l * V + col];
        if (idx < 0 || x > best) { best = x; idx = l; }
      }
    }
    wave_argmax(best, idx);
    if (idx >= 0 && lane == 0) s.mark[idx] = 3;
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void fixup_kernel(bd_fix_args p) {
  __shared__ FixSmem s;
  const int lane = threadIdx.x, b = blockIdx.x;
  const int L = p.L, V = p.V;
  const int CH = (L + 63) / 64, l0 = lane * CH;
  long long* ids_g = p.ids + (size_t)b * L;
  const float* lg = p.logits ? p.logits + (size_t)b * L * V : nullptr;
  for (int l = lane; l < L; l += 64) s.ids[l] = (int)ids_g[l];
  __syncthreads();

  // ---- natural-EOS rows that never drew EOS get one appended
  if (p.natural_eos && p.has_eos && p.has_eos[b] == 0) {
    int last = -1;
    for (int j = 0; j < CH; ++j) {
      const int l = l0 + j;
      if (l < L && s.ids[l] != p.pad_id) last = l;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(last, o, 64); last = v > last ? v : last; }
    if (last >= 0) {
      if (s.ids[last] == p.mask_id) {
        if (lane == 0) s.ids[last] = p.eos_id;
      } else if (last + 1 < L) {
        if (lane == 0) s.ids[last + 1] = p.eos_id;
        for (int l = last + 2 + lane; l < L; l += 64) {
          s.ids[l] = p.pad_id;
          if (p.attn) p.attn[(size_t)b * L + l] = 0;
          if (p.keymask) p.keymask[(size_t)b * L + l] = 0;
        }
      }
    }
    if (lane == 0) p.has_eos[b] = 1;
    __syncthreads();
  }

  if (p.do_fix && lg != nullptr) {
    // ---- exactly `target` star / placeholder tokens
    const int cid = p.count_id;
    int n_c = 0;
    for (int j = 0; j < CH; ++j) { const int l = l0 + j; if (l < L && s.ids[l] == cid) n_c++; }
    n_c = bd_wave_sum_i(n_c);
    if (n_c > p.target) {
      for (int l = lane; l < L; l += 64) s.mark[l] = (s.ids[l] == cid) ? 1 : 0;
      __syncthreads();
      pick_top_positions(s, lg, L, V, cid, p.target, lane);
      for (int l = 0; l < L; ++l) {
        if (s.mark[l] == 1) {  // an extra one: best other token
          const int tok = row_argmax(lg + (size_t)l * V, V, lane, [&](int v) {
            if (v == cid) return true;
            for (int q = 0; q < p.n_excl; ++q) if (v == p.excl[q]) return true;
            return false;
          });
          __syncthreads();
          if (lane == 0) s.ids[l] = tok;
        }
      }
      __syncthreads();
    } else if (n_c < p.target) {
      const int needed = p.target - n_c;
      int nvalid = 0;
      for (int l = lane; l < L; l += 64) {
        const int id = s.ids[l];
        const bool ok = id != p.bos_id && id != p.eos_id && id != p.pad_id && id != cid;
        s.mark[l] = ok ? 1 : 0;
        nvalid += ok;
      }
      nvalid = bd_wave_sum_i(nvalid);
      __syncthreads();
      if (nvalid >= needed) {
        pick_top_positions(s, lg, L, V, cid, needed, lane);
        for (int l = lane; l < L; l += 64) if (s.mark[l] == 3) s.ids[l] = cid;
        __syncthreads();
      }
    }

    if (p.smiles_syntax) {
      auto paren_excl = [&](int v) {
        return v == p.close_id || v == p.open_id || v == p.mask_id || v == p.pad_id || v == p.bos_id ||
               v == p.eos_id || (p.close_id < 0 && v == V - 1) || (p.open_id < 0 && v == V - 1);
      };
      // ---- parentheses: left-to-right drop unmatched ')', right-to-left drop unmatched '('
      if (p.open_id >= 0 || p.close_id >= 0) {
        int depth = 0;
        for (int l = 0; l < L; ++l) {
          const int id = s.ids[l];
          if (id == p.open_id) depth++;
          else if (id == p.close_id) {
            if (depth > 0) depth--;
            else {
              const int tok = row_argmax(lg + (size_t)l * V, V, lane, paren_excl);
              __syncthreads();
              if (lane == 0) s.ids[l] = tok;
              __syncthreads();
            }
          }
        }
        depth = 0;
        for (int l = L - 1; l >= 0; --l) {
          const int id = s.ids[l];
          if (id == p.close_id) depth++;
          else if (id == p.open_id) {
            if (depth > 0) depth--;
            else {
              const int tok = row_argmax(lg + (size_t)l * V, V, lane, paren_excl);
              __syncthreads();
              if (lane == 0) s.ids[l] = tok;
              __syncthreads();
            }
          }
        }
      }
      // ---- ring closures: odd count -> replace the last occurrence
      for (int r = 0; r < p.R; ++r) {
        const int rid = p.ring_ids[r];
        int cnt = 0, last = -1;
        for (int j = 0; j < CH; ++j) {
          const int l = l0 + j;
          if (l < L && s.ids[l] == rid) { cnt++; last = l; }
        }
        cnt = bd_wave_sum_i(cnt);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(last, o, 64); last = v > last ? v : last; }
        if (cnt & 1) {
          const int tok = row_argmax(lg + (size_t)last * V, V, lane, [&](int v) {
            return (p.tok_flags[v] & BD_TF_RING) != 0 || v == p.mask_id || v == p.pad_id || v == p.bos_id ||
                   v == p.eos_id || v == p.open_id || v == p.close_id;
          });
          __syncthreads();
          if (lane == 0) s.ids[last] = tok;
          __syncthreads();
        }
      }
    }
  }
  __syncthreads();
  for (int l = lane; l < L; l += 64) ids_g[l] = (long long)s.ids[l];
}

}  // namespace

extern "C" int bd_sample_step(const bd_step_args* a, void* stream) {
  BD_REQUIRE(a != nullptr, "bd_sample_step: null args");
  const bd_step_args& p = *a;
  BD_REQUIRE(p.ids && p.ktable && p.tok_flags && p.contrib && p.ring_slot && p.non_special,
             "bd_sample_step: null table or ids");
  BD_REQUIRE(p.B >= 0 && p.L > 0 && p.L <= MAXL, "bd_sample_step: L=%d outside 1..%d", p.L, MAXL);
  BD_REQUIRE(p.V > 0 && p.V <= MAXV, "bd_sample_step: V=%d outside 1..%d", p.V, MAXV);
  BD_REQUIRE(p.R >= 0 && p.R <= MAXR, "bd_sample_step: %d ring ids > %d", p.R, MAXR);
  BD_REQUIRE(p.n_ns > 0, "bd_sample_step: empty fallback token list");
  BD_REQUIRE(p.temperature > 0.f, "bd_sample_step: temperature must be > 0");
  BD_REQUIRE(p.t >= 1, "bd_sample_step: t must be >= 1");
  if (p.hidden != nullptr) {
    BD_REQUIRE(p.wout != nullptr, "bd_sample_step: hidden mode needs the output projection");
    BD_REQUIRE(p.d > 0 && p.d <= MAXD && p.d % 8 == 0, "bd_sample_step: d=%d unsupported", p.d);
  } else {
    BD_REQUIRE(p.logits != nullptr, "bd_sample_step: neither logits nor hidden states given");
  }
  if (p.variant & BD_VB_NATURAL_EOS)
    BD_REQUIRE(p.has_eos != nullptr, "bd_sample_step: natural-EOS mode needs has_eos");
  if (p.variant & BD_VB_URGENCY)
    BD_REQUIRE(p.boost_id >= 0 && p.boost_id < p.V, "bd_sample_step: bad boost id");
  if (p.variant & BD_VB_DYN_SINGLE)
    BD_REQUIRE(p.count_id >= 0 && p.count_id < p.V, "bd_sample_step: bad count id");
  if (p.B == 0) return BD_OK;
  const size_t lds = smem_bytes(p.L, p.V, p.d, p.R > 0 ? p.R : 1);
  BD_REQUIRE(lds <= 64 * 1024, "bd_sample_step: working set of %zu bytes per sequence exceeds LDS", lds);
  // ids in and out, fixed mask, and (hidden mode) about one hidden row and its result per chosen position
  bd_prof_scope prof(stream, (double)p.B * (17.0 * p.L + (p.hidden ? 2.0 * p.d + 8.0 : 4.0 * p.L * p.V)), BD_PROF_SAMPLE);
  const int VS = (p.V + 31) & ~31;   // row stride of the chosen-row workspace
  if (p.hidden != nullptr && p.row_logits != nullptr) {
    BD_REQUIRE(p.row_pos != nullptr, "bd_sample_step: row workspace without row_pos");
    BD_REQUIRE(p.row_cap >= 1 && p.row_cap <= 32 && (p.row_cap & (p.row_cap - 1)) == 0,
               "bd_sample_step: row_cap=%d is not a power of two in 1..32", p.row_cap);
    BD_REQUIRE(p.d % 16 == 0 && p.d <= 512 && (p.d / 16) % 8 == 0,
               "bd_sample_step: the chosen-row workspace needs d in {128, 256, 384, 512}, not %d", p.d);
    if (!p.rows_chosen)
      hipLaunchKernelGGL(choose_rows_kernel, dim3(p.B), dim3(64), lds, (hipStream_t)stream, p, (unsigned char*)nullptr);
    const dim3 grid((unsigned)(((long long)p.B * p.row_cap + 127) / 128), 4);
#define BD_ROWS(NK)                                                                            \
  do {                                                                                         \
    const int wl = 2 * 32 * (NK * 32 + 16);                                                    \
    BD_SET_MAX_LDS((row_logits_kernel<NK>), wl);                                               \
    hipLaunchKernelGGL(row_logits_kernel<NK>, grid, dim3(256), wl, (hipStream_t)stream, p, VS); \
  } while (0)
    switch (p.d / 16) {
      case 8: BD_ROWS(8); break;
      case 16: BD_ROWS(16); break;
      case 24: BD_ROWS(24); break;
      default: BD_ROWS(32); break;
    }
#undef BD_ROWS
  }
  hipLaunchKernelGGL(sample_step_kernel, dim3(p.B), dim3(64), lds, (hipStream_t)stream, p, VS);
  return bd_check_launch("bd_sample_step");
}

extern "C" int bd_sample_choose_rows(const bd_step_args* a, int* row_pos, unsigned char* need_all, int row_cap,
                                     void* stream) {
  BD_REQUIRE(a != nullptr && row_pos != nullptr && need_all != nullptr, "bd_sample_choose_rows: null operand");
  bd_step_args p = *a;
  BD_REQUIRE(p.ids && p.ktable && p.contrib, "bd_sample_choose_rows: null table or ids");
  BD_REQUIRE(p.B >= 0 && p.L > 0 && p.L <= MAXL, "bd_sample_choose_rows: L=%d outside 1..%d", p.L, MAXL);
  BD_REQUIRE(p.V > 0 && p.V <= MAXV, "bd_sample_choose_rows: V=%d outside 1..%d", p.V, MAXV);
  BD_REQUIRE(p.R >= 0 && p.R <= MAXR, "bd_sample_choose_rows: %d ring ids > %d", p.R, MAXR);
  BD_REQUIRE(p.d > 0 && p.d <= MAXD, "bd_sample_choose_rows: d=%d unsupported", p.d);
  BD_REQUIRE(p.t >= 1, "bd_sample_choose_rows: t must be >= 1");
  BD_REQUIRE(row_cap >= 1 && row_cap <= 32, "bd_sample_choose_rows: row_cap=%d outside 1..32", row_cap);
  if (p.B == 0) return BD_OK;
  p.row_pos = row_pos;
  p.row_cap = row_cap;
  const size_t lds = smem_bytes(p.L, p.V, p.d, p.R > 0 ? p.R : 1);
  BD_REQUIRE(lds <= 64 * 1024, "bd_sample_choose_rows: working set of %zu bytes per sequence exceeds LDS", lds);
  bd_prof_scope prof(stream, (double)p.B * (9.0 * p.L + 4.0 * row_cap + 1.0), BD_PROF_SAMPLE);
  hipLaunchKernelGGL(choose_rows_kernel, dim3(p.B), dim3(64), lds, (hipStream_t)stream, p, need_all);
  return bd_check_launch("bd_sample_choose_rows");
}

extern "C" int bd_sample_fixup(const bd_fix_args* a, void* stream) {
  BD_REQUIRE(a != nullptr, "bd_sample_fixup: null args");
  const bd_fix_args& p = *a;
  BD_REQUIRE(p.ids, "bd_sample_fixup: null ids");
  BD_REQUIRE(p.B >= 0 && p.L > 0 && p.L <= MAXL, "bd_sample_fixup: L=%d outside 1..%d", p.L, MAXL);
  BD_REQUIRE(p.V > 0 && p.V <= MAXV, "bd_sample_fixup: V=%d outside 1..%d", p.V, MAXV);
  BD_REQUIRE(!p.do_fix || (p.logits && p.count_id >= 0 && p.count_id < p.V),
             "bd_sample_fixup: repairs need logits and a counted token id");
  BD_REQUIRE(!p.smiles_syntax || (p.tok_flags && (p.R == 0 || p.ring_ids)),
             "bd_sample_fixup: SMILES repairs need the token tables");
  BD_REQUIRE(p.n_excl >= 0 && p.n_excl <= 8, "bd_sample_fixup: too many excluded ids");
  if (p.B == 0) return BD_OK;
  bd_prof_scope prof(stream, (double)p.B * 16.0 * p.L, BD_PROF_SAMPLE);
  hipLaunchKernelGGL(fixup_kernel, dim3(p.B), dim3(64), 0, (hipStream_t)stream, p);
  return bd_check_launch("bd_sample_fixup");
}
