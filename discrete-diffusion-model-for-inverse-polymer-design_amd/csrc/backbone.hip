// Host-side driver of the denoiser forward: carves the caller's workspace and enqueues the
// kernel chain of one forward pass on the given stream.
#include "bd_common.h"
#include "bidiff.h"
#include <stdlib.h>

namespace {
inline long long align256(long long x) { return (x + 255) & ~255ll; }

struct Carve {
  long long x, h, qkv, att, ff, part, stats, total;
};

Carve carve(const bd_backbone_desc& c, long long M) {
  Carve w;
  long long off = 0;
  w.x = off;   off += align256(M * c.d * 4);        // residual stream fp32
  w.h = off;   off += align256(M * c.d * 2);        // LayerNorm output bf16
  const long long aw = c.attn_width > 0 ? c.attn_width : c.d;
  w.qkv = off; off += align256(M * 3 * aw * 2);     // fused q|k|v bf16 (heads padded to 32 / 64 / 128)
  w.att = off; off += align256(M * aw * 2);         // attention output bf16
  w.ff = off;  off += align256(M * c.ffn * 2);      // GELU(fc1) bf16
  w.part = off; off += align256((c.d / 64 + 1) * M * 8);   // LayerNorm sums per 64-column block of the residual rows
  w.stats = off; off += align256(M * 8);            // (mean, 1 / std) per row
  w.total = off;
  return w;
}

int check_desc(const bd_backbone_desc* d) {
  BD_REQUIRE(d != nullptr, "backbone: null descriptor");
  BD_REQUIRE(d->d > 0 && d->n_layers >= 0 && d->n_heads > 0 && d->ffn > 0 && d->vocab > 0,
             "backbone: bad dimensions");
  BD_REQUIRE(d->d % d->n_heads == 0, "backbone: hidden %d not divisible by heads %d", d->d, d->n_heads);
  BD_REQUIRE(d->attn_width == 0 || (d->attn_width >= d->d && d->attn_width % (128 * 1) == 0 && d->attn_width % d->n_heads == 0),
             "backbone: attn_width %d must be a multiple of 128 and of the head count, and >= hidden", d->attn_width);
  BD_REQUIRE(d->d % 128 == 0 && d->ffn % 128 == 0,
             "backbone: hidden (%d) and ffn (%d) sizes must be multiples of 128", d->d, d->ffn);
  BD_REQUIRE(d->vocab_pad % 128 == 0 && d->vocab_pad >= d->vocab, "backbone: vocab_pad must be a multiple of 128");
  BD_REQUIRE(d->tok_emb && d->pos_emb && d->time_emb && d->lnf_g && d->lnf_b, "backbone: null table");
  BD_REQUIRE(d->n_layers == 0 || d->layers != nullptr, "backbone: null layer array");
  return BD_OK;
}
}  // namespace

extern "C" long long bd_backbone_workspace_bytes(const bd_backbone_desc* desc, int B, int L) {
  if (check_desc(desc) != BD_OK || B <= 0 || L <= 0) return -1;
  return carve(*desc, (long long)B * L).total;
}

namespace {
// n_run layers (n_run = n_layers: the whole network); resid_out (optional) receives the fp32 residual
// stream after those layers; the final LayerNorm / vocabulary head run only when asked for.
int forward_impl(const bd_backbone_desc* desc, const long long* ids, const long long* timesteps,
                 const unsigned char* keymask, int B, int L, void* workspace, long long workspace_bytes,
                 int n_run, float* resid_out, void* hidden_out, float* logits_out, void* stream) {
  int rc = check_desc(desc);
  if (rc != BD_OK) return rc;
  const bd_backbone_desc& c = *desc;
  BD_REQUIRE(ids && timesteps && workspace, "bd_backbone_forward: null operand");
  BD_REQUIRE(B > 0 && L > 0, "bd_backbone_forward: empty batch");
  BD_REQUIRE(L <= c.max_pos, "bd_backbone_forward: sequence length %d exceeds max_position_embeddings %d", L,
             c.max_pos);
  const long long M = (long long)B * L;
  BD_REQUIRE(M < (1ll << 31), "bd_backbone_forward: too many tokens");
  const Carve w = carve(c, M);
  BD_REQUIRE(workspace_bytes >= w.total, "bd_backbone_forward: workspace %lld < %lld bytes", workspace_bytes,
             w.total);
  BD_REQUIRE(logits_out == nullptr || c.w_out != nullptr, "bd_backbone_forward: logits need w_out");
  char* ws = (char*)workspace;
  float* x = (float*)(ws + w.x);
  void* h = ws + w.h;
  void* qkv = ws + w.qkv;
  void* att = ws + w.att;
  void* ff = ws + w.ff;
  const int d = c.d, F = c.ffn, Mi = (int)M;
  const int aw = c.attn_width > 0 ? c.attn_width : d;   // q / k / v row width (padded heads)

  // Whole forward in one launch when the model fits the fused phases (one sequence per workgroup,
  // residual rows in registers across all layers); BD_MEGA=0 in the environment keeps the
  // per-layer kernels (for comparisons).
  static const bool mega_enabled = []() { const char* e = getenv("BD_MEGA"); return !(e && e[0] == '0'); }();
  if (mega_enabled && n_run == c.n_layers && resid_out == nullptr && (hidden_out != nullptr || logits_out != nullptr) &&
      bd_backbone_mega_supported(desc, L)) {
    void* hf = hidden_out ? hidden_out : h;
    if ((rc = bd_backbone_mega(desc, ids, timesteps, keymask, B, L, x, hf, stream))) return rc;
    if (logits_out != nullptr)
      return bd_gemm_bf16(hf, c.w_out, nullptr, logits_out, Mi, c.vocab_pad, c.vocab, d, c.vocab, 3, stream);
    return BD_OK;
  }
  if ((rc = bd_embed(ids, timesteps, c.tok_emb, c.pos_emb, c.time_emb, x, B, L, d, c.vocab, c.max_pos, c.n_timesteps, stream))) return rc;
  // Fused kernels whenever the shapes allow: the whole attention block in one kernel per sequence
  // and the whole feed-forward block in one kernel per 128-token panel (nothing but the residual
  // stream touches memory). Otherwise LayerNorm folded into the row-panel projection kernel, or
  // plain LayerNorm + tiled GEMMs. The residual projections of the unfused path use the tiled
  // GEMM, which is the faster one there.
  const bool pan_qkv = bd_gemm_panel_supported(3 * aw, d, 1) != 0;
  const bool pan_fc1 = bd_gemm_panel_supported(F, d, 1) != 0;
  const bool fused_ffn = bd_ffn_fused_supported(d, F) != 0;
  const bool fused_attn = bd_attn_block_supported(d, c.n_heads, L) != 0 && aw == bd_attn_block_width(d, c.n_heads);
  // LayerNorm deferred across the residual projections (large models on the eight-wave GEMM): the projection that
  // updates the residual rows also leaves their bf16 copy and their row sums, and the next projection multiplies the
  // raw rows by the gamma-folded weights and normalises in its epilogue, LN(x) W^T = rstd (x W^T - mean colsum(W)).
  // No LayerNorm kernel between the layers then (the first one, behind the embeddings, and the final one remain).
  float* part = (float*)(ws + w.part);
  float* stats = (float*)(ws + w.stats);
  bool defer = !fused_attn && !fused_ffn && bd_gemm_deferred_ln_supported(Mi, d, aw) && bd_gemm_deferred_ln_supported(Mi, d, F) &&
               bd_gemm_deferred_ln_supported(Mi, 3 * aw, d) && bd_gemm_deferred_ln_supported(Mi, F, d);
  for (int i = 0; i < n_run && defer; ++i) {
    const bd_layer_weights& lw = c.layers[i];
    defer = lw.w_qkv_ln && lw.b_qkv_ln && lw.cs_qkv_ln && lw.w_fc1_ln && lw.b_fc1_ln && lw.cs_fc1_ln;
  }
  bool have_stats = false;      // h holds the bf16 copy of x and stats its row statistics
  for (int i = 0; i < n_run; ++i) {
    const bd_layer_weights& lw = c.layers[i];
    if (defer) {
      // ---- x += out_proj(attention(ln1(x)))
      if (have_stats) {
        rc = bd_gemm_bf16_deferred_ln(h, lw.w_qkv_ln, lw.b_qkv_ln, lw.cs_qkv_ln, stats, qkv, Mi, 3 * aw, d, 0, stream);
      } else {
        if ((rc = bd_layernorm(x, lw.ln1_g, lw.ln1_b, h, nullptr, nullptr, Mi, d, c.ln_eps, stream))) return rc;
        rc = bd_gemm_bf16(h, lw.w_qkv, lw.b_qkv, qkv, Mi, 3 * aw, 3 * aw, d, 3 * aw, 0, stream);
      }
      if (rc) return rc;
      if ((rc = bd_attention_fwd(qkv, keymask, att, nullptr, B, L, c.n_heads, aw / c.n_heads, stream))) return rc;
      if ((rc = bd_gemm_bf16_resid_stats(att, lw.w_o, lw.b_o, x, h, part, stats, Mi, d, aw, c.ln_eps, stream))) return rc;
      // ---- x += fc2(gelu(fc1(ln2(x))))
      if ((rc = bd_gemm_bf16_deferred_ln(h, lw.w_fc1_ln, lw.b_fc1_ln, lw.cs_fc1_ln, stats, ff, Mi, F, d, 1, stream))) return rc;
      if (i + 1 < n_run) {
        if ((rc = bd_gemm_bf16_resid_stats(ff, lw.w_fc2, lw.b_fc2, x, h, part, stats, Mi, d, F, c.ln_eps, stream))) return rc;
        have_stats = true;
      } else {
        if ((rc = bd_gemm_bf16(ff, lw.w_fc2, lw.b_fc2, x, Mi, d, d, F, d, 2, stream))) return rc;
      }
      continue;
    }
    // ---- x += out_proj(attention(ln1(x)))
    if (fused_attn && lw.attn_pack && lw.b_q_attn && lw.b_o_fold) {
      rc = bd_attn_block_fused(x, x, lw.attn_pack, lw.b_q_attn, lw.b_o, lw.b_o_fold, keymask, B, L, d, c.n_heads, c.ln_eps, stream);
      if (rc) return rc;
    } else {
      if (pan_qkv && lw.w_qkv_ln && lw.b_qkv_ln) {
        rc = bd_gemm_panel(x, lw.w_qkv_ln, lw.b_qkv_ln, qkv, nullptr, Mi, 3 * aw, d, 3 * aw, 0, 1, c.ln_eps, stream);
      } else {
        if ((rc = bd_layernorm(x, lw.ln1_g, lw.ln1_b, h, nullptr, nullptr, Mi, d, c.ln_eps, stream))) return rc;
        rc = bd_gemm_bf16(h, lw.w_qkv, lw.b_qkv, qkv, Mi, 3 * aw, 3 * aw, d, 3 * aw, 0, stream);
      }
      if (rc) return rc;
      if ((rc = bd_attention_fwd(qkv, keymask, att, nullptr, B, L, c.n_heads, aw / c.n_heads, stream))) return rc;
      if ((rc = bd_gemm_bf16(att, lw.w_o, lw.b_o, x, Mi, d, d, aw, d, 2, stream))) return rc;
    }
    // ---- x += fc2(gelu(fc1(ln2(x))))
    if (fused_ffn && lw.ffn_pack && lw.b_fc1_ln) {
      if ((rc = bd_ffn_fused(x, x, lw.ffn_pack, lw.b_fc1_ln, lw.b_fc2, Mi, d, F, c.ln_eps, stream))) return rc;
    } else {
      if (pan_fc1 && lw.w_fc1_ln && lw.b_fc1_ln) {
        rc = bd_gemm_panel(x, lw.w_fc1_ln, lw.b_fc1_ln, ff, nullptr, Mi, F, d, F, 1, 1, c.ln_eps, stream);
      } else {
        if ((rc = bd_layernorm(x, lw.ln2_g, lw.ln2_b, h, nullptr, nullptr, Mi, d, c.ln_eps, stream))) return rc;
        rc = bd_gemm_bf16(h, lw.w_fc1, lw.b_fc1, ff, Mi, F, F, d, F, 1, stream);
      }
      if (rc) return rc;
      if ((rc = bd_gemm_bf16(ff, lw.w_fc2, lw.b_fc2, x, Mi, d, d, F, d, 2, stream))) return rc;
    }
  }
  if (resid_out != nullptr) {
    if (hipMemcpyAsync(resid_out, x, (size_t)M * d * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
      return bd_fail(BD_ERR_LAUNCH, "bd_backbone_forward_layers: copy of the residual stream failed");
  }
  if (hidden_out == nullptr && logits_out == nullptr) return BD_OK;
  void* hf = hidden_out ? hidden_out : h;
  if ((rc = bd_layernorm(x, c.lnf_g, c.lnf_b, hf, nullptr, nullptr, Mi, d, c.ln_eps, stream))) return rc;
  if (logits_out != nullptr) {
    if ((rc = bd_gemm_bf16(hf, c.w_out, nullptr, logits_out, Mi, c.vocab_pad, c.vocab, d, c.vocab, 3, stream)))
      return rc;
  }
  return BD_OK;
}
}  // namespace

extern "C" int bd_backbone_forward(const bd_backbone_desc* desc, const long long* ids,
                                   const long long* timesteps, const unsigned char* keymask, int B,
                                   int L, void* workspace, long long workspace_bytes, void* hidden_out,
                                   float* logits_out, void* stream) {
  if (desc == nullptr) return bd_fail(BD_ERR_ARG, "backbone: null descriptor");
  return forward_impl(desc, ids, timesteps, keymask, B, L, workspace, workspace_bytes, desc->n_layers, nullptr,
                      hidden_out, logits_out, stream);
}

extern "C" int bd_backbone_forward_ragged(const bd_backbone_desc* desc, const long long* ids,
                                          const long long* timesteps, const unsigned char* keymask,
                                          const int* wave_seq, int n_blocks, int B, int L, void* workspace,
                                          long long workspace_bytes, void* hidden_out, float* logits_out,
                                          void* stream) {
  if (desc == nullptr) return bd_fail(BD_ERR_ARG, "backbone: null descriptor");
  int rc = check_desc(desc);
  if (rc != BD_OK) return rc;
  const bd_backbone_desc& c = *desc;
  BD_REQUIRE(ids && timesteps && workspace && wave_seq, "bd_backbone_forward_ragged: null operand");
  BD_REQUIRE(B > 0 && L > 0 && n_blocks > 0, "bd_backbone_forward_ragged: empty batch");
  BD_REQUIRE(L <= c.max_pos, "bd_backbone_forward_ragged: sequence length %d exceeds max_position_embeddings %d", L,
             c.max_pos);
  BD_REQUIRE(hidden_out != nullptr || logits_out != nullptr, "bd_backbone_forward_ragged: no output");
  const long long M = (long long)B * L;
  BD_REQUIRE(M < (1ll << 31), "bd_backbone_forward_ragged: too many tokens");
  const Carve w = carve(c, M);
  BD_REQUIRE(workspace_bytes >= w.total, "bd_backbone_forward_ragged: workspace %lld < %lld bytes", workspace_bytes,
             w.total);
  BD_REQUIRE(logits_out == nullptr || c.w_out != nullptr, "bd_backbone_forward_ragged: logits need w_out");
  void* hf = hidden_out ? hidden_out : (char*)workspace + w.h;
  if (hidden_out == nullptr) {
    // rows no wave computes must still be finite operands of the vocabulary head
    if (hipMemsetAsync(hf, 0, (size_t)M * c.d * 2, (hipStream_t)stream) != hipSuccess)
      return bd_fail(BD_ERR_LAUNCH, "bd_backbone_forward_ragged: clearing the hidden rows failed");
  }
  if ((rc = bd_backbone_mega_ragged(desc, ids, timesteps, keymask, wave_seq, n_blocks, B, L, hf, stream))) return rc;
  if (logits_out != nullptr)
    return bd_gemm_bf16(hf, c.w_out, nullptr, logits_out, (int)M, c.vocab_pad, c.vocab, c.d, c.vocab, 3, stream);
  return BD_OK;
}

// ---- the forward of a reverse step that reads few rows, for models on the per-layer kernels ---------------------
namespace {
// slot s = (sequence s / cap, position row_pos[s]); empty slots (-1) take the sequence's first row and are not written back
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const unsigned short* __restrict__ att,
                                                          const int* __restrict__ row_pos, float* __restrict__ xc,
                                                          unsigned short* __restrict__ attc, int L, int cap, int d, int aw) {
  const int s = blockIdx.x;
  const int pos = row_pos[s];
  const size_t row = (size_t)(s / cap) * L + (pos < 0 ? 0 : pos);
  for (int i = threadIdx.x; i < d / 4; i += 256) ((f32x4*)(xc + (size_t)s * d))[i] = ((const f32x4*)(x + row * d))[i];
  for (int i = threadIdx.x; i < aw / 8; i += 256) ((u32x4*)(attc + (size_t)s * aw))[i] = ((const u32x4*)(att + row * aw))[i];
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(const unsigned short* __restrict__ hc, const int* __restrict__ row_pos,
                                                           unsigned short* __restrict__ hidden, int L, int cap, int d) {
  const int s = blockIdx.x;
  const int pos = row_pos[s];
  if (pos < 0) return;
  const size_t row = (size_t)(s / cap) * L + pos;
  for (int i = threadIdx.x; i < d / 8; i += 256) ((u32x4*)(hidden + row * d))[i] = ((const u32x4*)(hc + (size_t)s * d))[i];
}

struct RowsCarve { long long xc, attc, hc, ffc, hfc, total; };
RowsCarve rows_carve(const bd_backbone_desc& c, long long Mc) {
  RowsCarve w;
  long long off = 0;
  const long long aw = c.attn_width > 0 ? c.attn_width : c.d;
  w.xc = off;   off += align256(Mc * c.d * 4);
  w.attc = off; off += align256(Mc * aw * 2);
  w.hc = off;   off += align256(Mc * c.d * 2);
  w.ffc = off;  off += align256(Mc * c.ffn * 2);
  w.hfc = off;  off += align256(Mc * c.d * 2);
  w.total = off;
  return w;
}
}  // namespace

// 1 when the model runs on the per-layer kernels with all four projections of a block on the eight-wave GEMM at
// B * L rows (the chosen rows' products then run on the same kernel and come out bit for bit as in the full forward)
extern "C" int bd_backbone_forward_rows_unfused_supported(const bd_backbone_desc* desc, int B, int L) {
  if (check_desc(desc) != BD_OK || B <= 0 || L <= 0 || L > desc->max_pos || desc->n_layers < 1) return 0;
  const bd_backbone_desc& c = *desc;
  const long long M = (long long)B * L;
  if (M >= (1ll << 31)) return 0;
  const int d = c.d, F = c.ffn, aw = c.attn_width > 0 ? c.attn_width : d, Mi = (int)M;
  if (bd_backbone_mega_supported(desc, L)) return 0;
  if (bd_ffn_fused_supported(d, F) || (bd_attn_block_supported(d, c.n_heads, L) && aw == bd_attn_block_width(d, c.n_heads))) return 0;
  if (bd_gemm_panel_supported(3 * aw, d, 1) || bd_gemm_panel_supported(F, d, 1)) return 0;
  for (int i = 0; i < c.n_layers; ++i)
    if (c.layers[i].cs_qkv_ln || c.layers[i].cs_fc1_ln) return 0;      // deferred LayerNorm: other roundings in the full forward
  return bd_gemm_deferred_ln_supported(Mi, d, aw) && bd_gemm_deferred_ln_supported(Mi, d, F) &&
         bd_gemm_deferred_ln_supported(Mi, 3 * aw, d) && bd_gemm_deferred_ln_supported(Mi, F, d);
}

extern "C" long long bd_backbone_rows_workspace_bytes(const bd_backbone_desc* desc, int B, int row_cap) {
  if (check_desc(desc) != BD_OK || B <= 0 || row_cap < 1 || row_cap > 32) return -1;
  return rows_carve(*desc, (long long)B * row_cap).total;
}

extern "C" int bd_backbone_forward_rows_unfused(const bd_backbone_desc* desc, const long long* ids,
                                                const long long* timesteps, const unsigned char* keymask, int B, int L,
                                                const int* row_pos, int row_cap, void* workspace, long long workspace_bytes,
                                                void* rows_workspace, long long rows_workspace_bytes, void* hidden_out,
                                                void* stream) {
  if (desc == nullptr) return bd_fail(BD_ERR_ARG, "backbone: null descriptor");
  BD_REQUIRE(row_pos && rows_workspace && hidden_out, "bd_backbone_forward_rows_unfused: null operand");
  BD_REQUIRE(row_cap >= 1 && row_cap <= 32, "bd_backbone_forward_rows_unfused: row_cap=%d outside 1..32", row_cap);
  BD_REQUIRE(bd_backbone_forward_rows_unfused_supported(desc, B, L),
             "bd_backbone_forward_rows_unfused: model shape or batch %d x %d not on the per-layer large-tile kernels", B, L);
  const bd_backbone_desc& c = *desc;
  const long long Mc = (long long)B * row_cap;
  const RowsCarve rw = rows_carve(c, Mc);
  BD_REQUIRE(rows_workspace_bytes >= rw.total, "bd_backbone_forward_rows_unfused: rows workspace %lld < %lld bytes",
             rows_workspace_bytes, rw.total);
  BD_REQUIRE(((size_t)rows_workspace & 255) == 0, "bd_backbone_forward_rows_unfused: rows workspace must be 256-byte aligned");
  // every layer but the last for every row (the residual rows stay in the workspace)
  int rc = forward_impl(desc, ids, timesteps, keymask, B, L, workspace, workspace_bytes, c.n_layers - 1, nullptr, nullptr,
                        nullptr, stream);
  if (rc) return rc;
  const long long M = (long long)B * L;
  const Carve w = carve(c, M);
  char* ws = (char*)workspace;
  float* x = (float*)(ws + w.x);
  void* h = ws + w.h;
  void* qkv = ws + w.qkv;
  void* att = ws + w.att;
  const int d = c.d, F = c.ffn, Mi = (int)M, Mci = (int)Mc;
  const int aw = c.attn_width > 0 ? c.attn_width : d;
  const bd_layer_weights& lw = c.layers[c.n_layers - 1];
  // the last layer: attention needs every row's keys and values; behind it only the chosen rows go on
  if ((rc = bd_layernorm(x, lw.ln1_g, lw.ln1_b, h, nullptr, nullptr, Mi, d, c.ln_eps, stream))) return rc;
  if ((rc = bd_gemm_bf16(h, lw.w_qkv, lw.b_qkv, qkv, Mi, 3 * aw, 3 * aw, d, 3 * aw, 0, stream))) return rc;
  if ((rc = bd_attention_fwd(qkv, keymask, att, nullptr, B, L, c.n_heads, aw / c.n_heads, stream))) return rc;
  char* rs = (char*)rows_workspace;
  float* xc = (float*)(rs + rw.xc);
  void* attc = rs + rw.attc;
  void* hc = rs + rw.hc;
  void* ffc = rs + rw.ffc;
  void* hfc = rs + rw.hfc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)Mc), dim3(256), 0, s, x, (const unsigned short*)att, row_pos, xc,
                     (unsigned short*)attc, L, row_cap, d, aw);
  if ((rc = bd_check_launch("bd_backbone_forward_rows_unfused (gather)"))) return rc;
  if ((rc = bd_gemm_bf16_tile256(attc, lw.w_o, lw.b_o, xc, Mci, d, aw, 2, stream))) return rc;
  if ((rc = bd_layernorm(xc, lw.ln2_g, lw.ln2_b, hc, nullptr, nullptr, Mci, d, c.ln_eps, stream))) return rc;
  if ((rc = bd_gemm_bf16_tile256(hc, lw.w_fc1, lw.b_fc1, ffc, Mci, F, d, 1, stream))) return rc;
  if ((rc = bd_gemm_bf16_tile256(ffc, lw.w_fc2, lw.b_fc2, xc, Mci, d, F, 2, stream))) return rc;
  if ((rc = bd_layernorm(xc, c.lnf_g, c.lnf_b, hfc, nullptr, nullptr, Mci, d, c.ln_eps, stream))) return rc;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)Mc), dim3(256), 0, s, (const unsigned short*)hfc, row_pos,
                     (unsigned short*)hidden_out, L, row_cap, d);
  return bd_check_launch("bd_backbone_forward_rows_unfused (scatter)");
}

extern "C" int bd_backbone_forward_layers(const bd_backbone_desc* desc, const long long* ids,
                                          const long long* timesteps, const unsigned char* keymask, int B,
                                          int L, void* workspace, long long workspace_bytes, int n_run,
                                          float* resid_out, void* stream) {
  if (desc == nullptr) return bd_fail(BD_ERR_ARG, "backbone: null descriptor");
  BD_REQUIRE(n_run >= 0 && n_run <= desc->n_layers, "bd_backbone_forward_layers: n_run=%d of %d layers", n_run,
             desc->n_layers);
  BD_REQUIRE(resid_out != nullptr, "bd_backbone_forward_layers: null output");
  return forward_impl(desc, ids, timesteps, keymask, B, L, workspace, workspace_bytes, n_run, resid_out, nullptr,
                      nullptr, stream);
}
