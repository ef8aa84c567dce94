/*
 * bidiff.h - C ABI of libbidiff_hip.so: the Bi_Diffusion reverse-sampling / training-step
 * hot path as hand-written HIP kernels for gfx950 (MI355X).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm allocates);
 *     the library keeps no caller pointer after a call returns and allocates nothing
 *     except where a function says so;
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); calls only
 *     enqueue work, they never synchronise the device;
 *   - return value: 0 = ok, negative = error (BD_ERR_*); bd_last_error() gives the text;
 *   - token ids, attention masks and timesteps are int64, as in the reference's Python API;
 *     bf16 tensors are passed as void*.
 *
 * The reference (htang7415/Discrete-Diffusion-Model-for-inverse-polymer-design) has no
 * FFI: the interface each entry point stands behind is a Python method, cited per function
 * as <variant>/src/...py:line (Bi_Diffusion_SMILES unless stated).
 */
#ifndef BIDIFF_H
#define BIDIFF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BD_ERR_ARG (-1)
#define BD_ERR_LAUNCH (-2)
#define BD_ERR_UNSUPPORTED (-3)

/* Library identification / diagnostics. */
const char* bd_version(void);
const char* bd_last_error(void);
/* 1 when a HIP device is usable by this process, else 0 (never raises). */
int bd_device_available(void);

/* ---------------------------------------------------------------- denoiser forward */

/* x[b,l,:] = tok_emb[ids[b,l]] + pos_emb[l] + time_emb[timesteps[b]]   (fp32)
 * vocab / max_pos / n_timesteps are the row counts of the three tables: L > max_pos is rejected; a token id or
 * timestep outside its table (nn.Embedding raises there) reads nothing and makes that row NaN.
 * replaces src/model/backbone.py:234-246 (three nn.Embedding lookups and their sum). */
int bd_embed(const long long* ids, const long long* timesteps, const float* tok_emb,
             const float* pos_emb, const float* time_emb, float* x, int B, int L, int d, int vocab,
             int max_pos, int n_timesteps, void* stream);

/* LayerNorm over the last dim (eps 1e-5 in the reference), fp32 in; writes bf16 and/or
 * fp32 output and optionally per-row (mean, rstd) for the backward pass.
 * replaces nn.LayerNorm at src/model/backbone.py:115,117,194 (calls :126,128,254). */
int bd_layernorm(const float* x, const float* gamma, const float* beta, void* out_bf16,
                 float* out_f32, float* stats, int M, int d, float eps, void* stream);

/* ---- fp32 compute mode (logits within 1e-5 of the reference's fp32 forward; the reference samples without
 * autocast, src/sampling/sampler.py). Embeddings and LayerNorm are bd_embed / bd_layernorm (fp32 output).
 *
 * out[M,N] = A[M,K] W[N,K]^T (+ bias[N]) in fp32 on the f32-input MFMA; epilogue 0: that, 1: erf-GELU of it,
 * 2: resid + it (out may be resid). K must be a multiple of 16, lda / ldw multiples of 4, A and W 16-byte aligned.
 * replaces nn.Linear at src/model/backbone.py:26-29 (q/k/v/out projections), :93-94 (feed-forward), :197 (head)
 * and nn.GELU at :95 when the model computes in fp32. */
int bd_gemm_f32(const float* A, long long lda, const float* W, long long ldw, const float* bias,
                const float* resid, float* out, long long ldc, int M, int N, int K, int epilogue,
                void* stream);
/* out[b,l,h*dh:(h+1)*dh] = softmax(q k^T / sqrt(dh) + key padding) v in fp32; qkv rows are [q | k | v] of width
 * 3 H dh; keymask uint8 [B,L] (0 = padded key) or null; rows whose keys are all padded come out as zeros.
 * dh a multiple of 16 up to 128, L <= 512. replaces F.scaled_dot_product_attention at src/model/backbone.py:60-66. */
int bd_attention_f32(const float* qkv, const unsigned char* keymask, float* out, int B, int L, int H,
                     int dh, void* stream);

/* C[M,n] = A[M,K] (bf16) * W[n_pad,K]^T (bf16) with a fused epilogue
 *   0: out bf16 = acc + bias          (q/k/v projections fused as one [3d,d] weight)
 *   1: out bf16 = gelu_erf(acc + bias)
 *   2: out fp32 += acc + bias         (residual stream, in place)
 *   3: out fp32  = acc (+ bias), only columns < n_real are written (vocabulary head)
 * n_pad % 128 == 0 (weight rows past n_real are zero), K % 64 == 0.
 * replaces nn.Linear at src/model/backbone.py:49-51,78 (attention), :96-99 (FFN),
 * :255 (output_proj). */
int bd_gemm_bf16(const void* A, const void* W, const float* bias, void* out, int M, int n_pad,
                 int n_real, int K, int ldc, int epilogue, void* stream);

/* Row-panel GEMM for skinny-K layers: the 128-row activation panel lives in registers, only the
 * weight stream goes through LDS; optional fused LayerNorm of A (fp32 rows; gamma / beta folded into
 * W / bias by the caller). epilogue 0: bf16 = acc + bias; 1: bf16 = gelu(acc + bias); 2: fp32 out =
 * resid + acc + bias (resid may alias out). bd_gemm_panel_supported tells whether (N, K) is handled
 * (N % 128 == 0, K a multiple of 128 / 256 / 384 chunk; several chunks need N in {128, 384}).
 * Same reference call sites as bd_gemm_bf16 (+ nn.LayerNorm, src/model/backbone.py:126,128). */
int bd_gemm_panel_supported(int N, int K, int layernorm_a);
int bd_gemm_panel(const void* A, const void* W, const float* bias, void* out, const float* resid, int M,
                  int N, int K, int ldc, int epilogue, int layernorm_a, float eps, void* stream);

/* Fused feed-forward block of one pre-norm layer, in place on the fp32 residual stream:
 *     out = x + W2 * GELU(W1' * normalise(x) + b1') + b2        (x, out: fp32 [M, d]; may alias)
 * normalise = LayerNorm without affine; the caller folds gamma into W1 (W1' = W1 * gamma) and beta
 * into b1 (b1' = b1 + W1 * beta). The [M, F] hidden activations are never written to memory.
 * The GELU runs in packed f16 and feeds the second product on the f16 matrix instruction (W2 is stored as f16 in
 * the stream; every bf16 weight of normal f16 magnitude converts exactly). Limits that follow, which the CALLER
 * checks before it packs (DiffusionBackbone.packed does; layers outside them run bd_gemm_bf16 with the GELU
 * epilogue instead): first-product outputs W1' * normalise(x) + b1' inside +-65504 (beyond that the conversion
 * saturates: the block returns the result for an activation of +-65504); W2 entries below 2^-14 in magnitude lose
 * bits (absolute error at most 2^-25 each, or the entry itself where the matrix instruction drops f16 subnormals).
 * wpack is the weight stream made by bd_ffn_pack_weights from W1' (bf16 [F, d]) and W2 (bf16 [d, F]);
 * it holds bd_ffn_pack_bytes(d, F) bytes. Supported: d in {128, 256, 384}, F % 32 == 0, F <= 4096.
 * Replaces TransformerBlock.forward's  x + ffn(ln2(x))  (src/model/backbone.py:128, FeedForward :85-96). */
int bd_ffn_fused_supported(int d, int F);
long bd_ffn_pack_bytes(int d, int F);
int bd_ffn_pack_weights(const void* w1_folded, const void* w2, void* wpack, int d, int F, void* stream);
int bd_ffn_fused(const float* x, float* out, const void* wpack, const float* b1_folded, const float* b2, int M,
                 int d, int F, float eps, void* stream);

/* Fused self-attention block of one pre-norm layer, in place on the fp32 residual stream, one
 * workgroup per sequence (L <= 128):
 *     out = x + Wo * MHA(normalise(x)) + b_o          (x, out: fp32 [B*L, d]; may alias)
 * q, k, v, scores and the attention output never reach memory. wpack is the weight stream made
 * by bd_attn_pack_weights from the stacked q|k|v projection (bf16 [3d, d], LayerNorm gamma folded
 * in, the q rows also scaled by log2(e)/sqrt(head_dim)) and the output projection (bf16 [d, d]);
 * it holds bd_attn_pack_bytes(d) bytes. Biases (fp32 [d] each, LayerNorm beta folded in: b + W beta):
 * b_q scaled like the q rows; the k bias is not needed (it shifts all scores of a query equally);
 * the v bias enters through b_o_fold = b_o + Wo * b_v. keymask: byte mask [B, L] or null; a
 * sequence whose keys are all masked gets out = x + b_o (zero attention output, as
 * bd_attention_fwd + out-proj). Supported: (d, head_dim) in {(384,128), (384,64), (256,128), (256,64),
 * (128,64), (128,32)}, L <= 128.
 * Replaces TransformerBlock.forward's  x + attn(ln1(x))  (src/model/backbone.py:126,
 * MultiHeadAttention.forward :38-72). */
int bd_attn_block_supported(int d, int n_heads, int L);
/* Width H * dh' of q, k and v inside the fused block: dh' is the head size d / n_heads, or the next of 32 / 64 / 128
 * when the kernels have no tiling for it (8 heads of 48 at d = 384 run as 8 heads of 64). The caller zero-pads the
 * q | k | v rows ([3 H dh', d], head by head) and the columns of the output projection ([d, H dh']) and multiplies the
 * q rows by log2(e) / sqrt(d / n_heads) (the true head size); the q bias has H dh' entries. 0 = unsupported. */
int bd_attn_block_width(int d, int n_heads);
long bd_attn_pack_bytes(int d);                       /* heads of 32 / 64 / 128 (width = d) */
long bd_attn_pack_bytes_heads(int d, int n_heads);    /* any supported head count: 8 * width * d */
int bd_attn_pack_weights(const void* w_qkv_folded_scaled, const void* w_o, void* wpack, int d, int n_heads,
                         void* stream);
int bd_attn_block_fused(const float* x, float* out, const void* wpack, const float* b_q_folded_scaled,
                        const float* b_o, const float* b_o_fold, const unsigned char* keymask, int B, int L,
                        int d, int n_heads, float eps, void* stream);

/* Non-causal multi-head attention on the fused qkv rows [B*L, 3*H*head_dim] (bf16) with a
 * key-padding byte mask [B,L] (non-zero = visible; null = all visible). A query whose keys
 * are all masked gives an all-zero output row. out [B*L, H*head_dim] bf16; lse optional
 * [B,H,L] fp32. head_dim in {32, 64, 128}, any L (other head sizes: zero-pad, see
 * bd_backbone_desc.attn_width).
 * replaces F.scaled_dot_product_attention at src/model/backbone.py:54-67. */
int bd_attention_fwd(const void* qkv, const unsigned char* keymask, void* out, float* lse, int B,
                     int L, int H, int head_dim, void* stream);

/* out[i] = attention_mask[i] != 0 (the .to(torch.bool) of src/model/backbone.py:59). */
int bd_mask_to_bytes(const long long* attention_mask, unsigned char* out, long long n, void* stream);

/* Weights of one pre-LN transformer block (device pointers; bf16 matrices are row-major
 * [out, in] exactly as torch.nn.Linear stores them, q/k/v stacked into one [3d, d]). */
typedef struct bd_layer_weights {
  const float* ln1_g; const float* ln1_b;
  const void* w_qkv;  const float* b_qkv;   /* [3d, d] bf16, [3d] fp32 */
  const void* w_o;    const float* b_o;     /* [d, d],  [d]            */
  const float* ln2_g; const float* ln2_b;
  const void* w_fc1;  const float* b_fc1;   /* [F, d],  [F]            */
  const void* w_fc2;  const float* b_fc2;   /* [d, F],  [d]            */
  /* optional LayerNorm-folded copies for the fused path (null = unfused):
   * w' = w * gamma (per input column), b' = b + w * beta                 */
  const void* w_qkv_ln; const float* b_qkv_ln;
  const void* w_fc1_ln; const float* b_fc1_ln;
  /* optional weight stream of the fused feed-forward kernel (bd_ffn_pack_weights of w_fc1_ln and
   * w_fc2; needs b_fc1_ln); null = separate fc1 / fc2 GEMMs                                   */
  const void* ffn_pack;
  /* optional weight stream and bias of the fused attention block (bd_attn_pack_weights); null =
   * separate projection / attention / out-proj kernels                                        */
  const void* attn_pack; const float* b_q_attn; const float* b_o_fold;
  /* optional row sums (fp32, over the input columns) of the bf16 values of w_qkv_ln [3 * attn_width] and w_fc1_ln [F]:
   * with them (and the folded copies above) large models run LayerNorm deferred across the residual projections,
   * see bd_gemm_bf16_deferred_ln; null = LayerNorm kernels                                       */
  const float* cs_qkv_ln; const float* cs_fc1_ln;
} bd_layer_weights;

typedef struct bd_backbone_desc {
  int d, n_layers, n_heads, ffn, vocab, vocab_pad, max_pos, n_timesteps;
  float ln_eps;
  const float* tok_emb;   /* [vocab, d]        fp32 */
  const float* pos_emb;   /* [max_pos, d]      fp32 */
  const float* time_emb;  /* [n_timesteps, d]  fp32 */
  const float* lnf_g; const float* lnf_b;
  const void* w_out;      /* [vocab_pad, d] bf16, rows >= vocab are zero */
  const bd_layer_weights* layers; /* HOST array of n_layers entries */
  /* width of the q, k, v and attention-output rows: n_heads * (head_dim padded to 32 / 64 / 128);
   * 0 = d. When larger than d (e.g. 8 heads of 48 at d = 384 -> 8 x 64 = 512) the caller pads the
   * weights with zero rows / columns: w_qkv [3*attn_width, d], b_qkv [3*attn_width], w_o [d, attn_width],
   * and pre-scales the q rows by sqrt(padded / true head_dim) (the kernel scales by 1/sqrt(padded)). */
  int attn_width;
} bd_backbone_desc;

/* LayerNorm deferred across a residual projection, for the shapes the eight-wave 256 x 256 GEMM takes (N a multiple of
 * 128, K a multiple of 64, K >= 128; bd_gemm_deferred_ln_supported additionally tells whether the plain bd_gemm_bf16
 * would pick that kernel for the shape: bd_backbone_forward defers only then, and only when every layer carries cs_qkv_ln
 * and cs_fc1_ln; measured, the forward takes the same time either way).
 * bd_gemm_bf16_resid_stats: x [M, N] fp32 += A [M, K] W[N, K]^T + bias (the block's residual projection, epilogue 2 of
 * bd_gemm_bf16), and on the way out xb [M, N] bf16 = the updated rows rounded once, part (scratch of
 * bd_gemm_resid_stats_part_bytes(M, N) bytes) their sums and sums of squares per 64-column block, stats [M, 2] fp32 =
 * (mean, 1 / sqrt(biased variance + eps)) of every updated row (a second, small launch).
 * bd_gemm_bf16_deferred_ln: out [M, N] bf16 = f(LN(x) W^T + b) computed as rstd * (xb W_ln^T - mean * colsum) + bias_ln
 * with W_ln = W * gamma per input column (bf16), bias_ln = b + W beta, colsum [N] = fp32 row sums of the bf16 values of
 * W_ln; epilogue 0: f = identity, 1: f = GELU (erf form). The rows enter the matrix cores as bf16(x) instead of
 * bf16(LN(x)): the rounding error of an element is relative to |x| instead of |x - mean|, the same when the row mean
 * is small against the row's spread (residual rows of a pre-LN transformer), larger by |mean| / std otherwise.
 * Together they replace  x = x + proj(...); h = norm(x); y = linear(h)  of TransformerBlock.forward
 * (src/model/backbone.py:120-129) without a LayerNorm kernel between the two projections. */
int bd_gemm_deferred_ln_supported(int M, int N, int K);
long long bd_gemm_resid_stats_part_bytes(int M, int N);
int bd_gemm_bf16_resid_stats(const void* A, const void* W, const float* bias, float* x, void* xb, float* part,
                             float* stats, int M, int N, int K, float eps, void* stream);
int bd_gemm_bf16_deferred_ln(const void* xb, const void* W_ln, const float* bias_ln, const float* colsum,
                             const float* stats, void* out, int M, int N, int K, int epilogue, void* stream);

/* Bytes of device scratch bd_backbone_forward needs for a [B, L] batch. */
long long bd_backbone_workspace_bytes(const bd_backbone_desc* desc, int B, int L);

/* Whole denoiser forward (inference, no dropout): embeddings -> n_layers x
 * [LN -> QKV -> attention -> out-proj + residual -> LN -> fc1 + GELU -> fc2 + residual]
 * -> final LN (-> vocabulary projection). Writes hidden_out [B*L, d] bf16 (final-norm
 * output; may be null) and/or logits_out [B*L, vocab] fp32 (may be null). keymask: byte
 * mask [B, L] or null. workspace: caller-allocated device scratch.
 * replaces DiffusionBackbone.forward, src/model/backbone.py:215-257 (hidden_out with
 * logits_out == null: get_hidden_states(layer_idx=-1), :259-299). */
int bd_backbone_forward(const bd_backbone_desc* desc, const long long* ids,
                        const long long* timesteps, const unsigned char* keymask, int B, int L,
                        void* workspace, long long workspace_bytes, void* hidden_out,
                        float* logits_out, void* stream);

/* The same forward as ONE kernel launch, one sequence per workgroup (two / four when L <= 64 / 32: the
 * sequences sit side by side in the key tiles behind per-wave mask tables, rows keep their bits; environment
 * BD_MEGA_PACK=0 turns the sharing off), when bd_backbone_mega_supported
 * (every layer carries the fused-kernel weight streams, L <= 128, n_layers <= 32): embeddings, all
 * layers and the final LayerNorm; the residual rows stay in registers from the embeddings to the
 * final LayerNorm. xbuf is not used any more (any non-null pointer; kept so that the signature
 * stays), hidden_out [B*L, d] bf16 is the final-norm output. bd_backbone_forward
 * takes this route by itself (environment BD_MEGA=0 turns it off). */
int bd_backbone_mega_supported(const bd_backbone_desc* desc, int L);
int bd_backbone_mega(const bd_backbone_desc* desc, const long long* ids, const long long* timesteps,
                     const unsigned char* keymask, int B, int L, float* xbuf, void* hidden_out, void* stream);

/* The single-launch forward for batches whose sequences are shorter than the padded length L (the
 * reverse loop on per-sample lengths, src/sampling/sampler.py:747-812, pads a batch to its longest row): the
 * caller deals the sequences to the four 32-row waves of each workgroup, wave_seq [n_blocks, 4] int32 on
 * the device holding the sequence of each wave (-1: none), the waves of one sequence adjacent and in row
 * order. A sequence with w waves gets its rows 0 .. 32 w - 1 computed (w = ceil(attended length / 32));
 * hidden_out rows beyond them are NOT written, rows inside them equal bd_backbone_mega's bit for bit.
 * Every sequence must appear in exactly one run of waves. bd_backbone_forward_ragged adds the
 * vocabulary head over all B * L rows like bd_backbone_forward (hidden_out or logits_out may be null;
 * the workspace is the one of bd_backbone_forward). */
int bd_backbone_mega_ragged_supported(const bd_backbone_desc* desc, int L);
int bd_backbone_mega_ragged(const bd_backbone_desc* desc, const long long* ids, const long long* timesteps,
                            const unsigned char* keymask, const int* wave_seq, int n_blocks, int B, int L,
                            void* hidden_out, void* stream);
int bd_backbone_forward_ragged(const bd_backbone_desc* desc, const long long* ids, const long long* timesteps,
                               const unsigned char* keymask, const int* wave_seq, int n_blocks, int B, int L,
                               void* workspace, long long workspace_bytes, void* hidden_out, float* logits_out,
                               void* stream);

/* The forward for a reverse step that reads few hidden rows: hidden_out [B*L, d] bf16 receives the final-norm rows
 * of the positions row_pos[b * row_cap + i] (>= 0) of every sequence b, and all rows of the sequences with
 * need_all[b] != 0 (bd_sample_choose_rows writes both lists); any other row of hidden_out either keeps what it
 * held or receives its final-norm row too. The rows written carry the bits bd_backbone_mega writes. Every layer runs
 * for every row except the last feed-forward block and the final LayerNorm, which only the listed rows need (the
 * last layer's keys and values need every row of the layers below).
 * need_all is in / out: a sequence that shares a workgroup with a flagged one has all its rows finished as well and
 * is marked with 2 (bd_sample_choose_rows rewrites the list for the next step).
 * wave_seq / n_blocks: as for bd_backbone_mega_ragged, or null / 0. workspace: B * row_cap * d floats.
 * Serves the reverse loop at t > 1 (src/sampling/sampler.py:661-718: the step reads the logits of the positions
 * it draws, :686-693; their choice does not depend on the logits, :676-686). */
int bd_backbone_forward_rows_supported(const bd_backbone_desc* desc, int L);
int bd_backbone_forward_rows(const bd_backbone_desc* desc, const long long* ids, const long long* timesteps,
                             const unsigned char* keymask, const int* wave_seq, int n_blocks, int B, int L,
                             const int* row_pos, unsigned char* need_all, int row_cap,
                             void* workspace, long long workspace_bytes, void* hidden_out, void* stream);

/* Embeddings + the first n_run blocks only; resid_out [B*L, d] fp32 receives the residual stream
 * (no final LayerNorm). Replaces get_hidden_states(layer_idx = n_run - 1 >= 0),
 * src/model/backbone.py:259-299. */
int bd_backbone_forward_layers(const bd_backbone_desc* desc, const long long* ids,
                               const long long* timesteps, const unsigned char* keymask, int B, int L,
                               void* workspace, long long workspace_bytes, int n_run, float* resid_out,
                               void* stream);

/* Pooling of the final hidden states over the sequence: hidden [B, L, d] bf16 -> out [B, d] fp32.
 * mode 0 = mean: sum(h * m) / max(sum m, 1e-9) with m = float(attention_mask) (plain mean when the
 * mask is null); 1 = cls: h[:, 0]; 2 = max over h * m + (1 - m) * -1e9 (plain max without a mask).
 * replaces DiffusionBackbone.get_pooled_output, src/model/backbone.py:301-335. */
int bd_pool_hidden(const void* hidden_bf16, const long long* attention_mask, float* out, int B, int L,
                   int d, int mode, void* stream);

/* bd_gemm_bf16 with the extra operands some epilogues take:
 *   epilogue 1: aux_out (optional) receives the bf16 pre-activation acc + bias (saved for backward)
 *   epilogue 2: aux_in (optional) is the fp32 residual input; out = aux_in + acc + bias
 *   epilogue 4: out bf16 = (acc + bias) * gelu'(aux_in), aux_in = bf16 pre-activation
 *   epilogue 6: out bf16 = gelu(acc + bias), aux_out (required) = bf16 gelu'(acc + bias): the multiplier of the
 *               backward pass instead of the pre-activation
 *   epilogue 7: out bf16 = (acc + bias) * aux_in, aux_in = the bf16 multiplier epilogue 6 left
 * (no single reference line: the pieces autograd records for backbone.py:96-99,126-128). */
int bd_gemm_bf16_aux(const void* A, const void* W, const float* bias, void* out, const void* aux_in,
                     void* aux_out, int M, int n_pad, int n_real, int K, int ldc, int epilogue,
                     void* stream);

/* ---------------------------------------------------------------- q(x_t | x_0) */

/* should_mask = (u < schedule[t[b]]) & (attention_mask == 1, or ids != pad when the mask is
 * null) & ids != bos & ids != eos (bos/eos < 0 = "None"); noisy = mask_id where should_mask.
 * uniforms [B,L] fp32 or null; null draws u from Philox4x32-10 with key = seed: word l % 4 of the
 * block with counter (l / 4, seq_offset + b, t[b], 2^31).
 * replaces DiscreteMaskingDiffusion.forward_process, src/model/diffusion.py:84-127. */
int bd_corrupt(const long long* ids, const long long* attention_mask, const long long* timesteps,
               const float* schedule, int schedule_len, const float* uniforms,
               long long* noisy_ids, unsigned char* mask_indicator, int B, int L, int mask_id,
               int pad_id, int bos_id, int eos_id, unsigned long long seed, long long seq_offset,
               void* stream);

/* ---------------------------------------------------------------- training input */

/* Batch assembly from a pre-tokenised store resident in device memory: tokens (uint16, all sequences
 * back to back with BOS / EOS in place), offsets [n_rows + 1] (int64), indices [B] (int64 store rows of
 * this batch). Writes ids_out [B, L] = tokens then pad_id and attention_mask_out [B, L] = 1 on tokens
 * (int64). error_flag (device int, caller zeroes) gets bit 0 for an index out of range and bit 1 for a
 * sequence longer than L (it is cut to L).
 * replaces PolymerDataset.__getitem__ + collate_fn / dynamic_collate_fn, src/data/dataset.py:66-84,200-240
 * (tokenisation itself happens once, on the host, when the store is built). */
int bd_collate_batch(const void* tokens_u16, const long long* offsets, const long long* indices,
                     long long n_rows, int B, int L, int pad_id, long long* ids_out,
                     long long* attention_mask_out, int* error_flag, void* stream);

/* ---------------------------------------------------------------- training step */

/* Training dropout (nn.Dropout / SDPA dropout_p, src/model/backbone.py:30,65,93,98,100,185,247).
 * Keep decisions are a pure function of (seed, step, site, row, column): one Philox4x32-10 block
 * with counter (column / 8, row, site, step) and key = seed serves eight neighbouring columns; column
 * c takes the low (c even) or high (c odd) 16 bits of word (c % 8) / 2 and is kept iff that number
 * >= floor(p * 2^16), then scaled by 1 / (1 - p). Forward and backward kernels regenerate the
 * decisions, nothing is stored; a caller bumps `step` once per forward pass. A null bd_dropout or
 * p < 2^-16 is the identity.
 * Sites of the denoiser (rows x columns): 0 = embedding sum [B*L, d]; for block i:
 * 4 i + 1 = attention probabilities [(b H + h) L + query, key], 4 i + 2 = GELU output [B*L, F],
 * 4 i + 3 = fc2 output [B*L, d]. */
typedef struct bd_dropout {
  float p;
  unsigned step;
  unsigned long long seed;
} bd_dropout;

/* out = x * keep / (1 - p) over [M, N] fp32 (out may alias x); N % 4 == 0. */
int bd_dropout_f32(const float* x, float* out, int M, int N, const bd_dropout* drop, unsigned site,
                   void* stream);
/* The same with a bf16 result. */
int bd_dropout_f32_to_bf16(const float* x, void* out_bf16, int M, int N, const bd_dropout* drop,
                           unsigned site, void* stream);

/* bd_embed followed by dropout of the sum (src/model/backbone.py:247). */
int bd_embed_drop(const long long* ids, const long long* timesteps, const float* tok_emb,
                  const float* pos_emb, const float* time_emb, float* x, int B, int L, int d,
                  int vocab, int max_pos, int n_timesteps, const bd_dropout* drop, unsigned site, void* stream);

/* bd_gemm_bf16_aux with dropout in the epilogue (row = m, column = n of the output):
 *   epilogue 1: out = drop(gelu(acc + bias))            (src/model/backbone.py:97-98)
 *   epilogue 2: out = aux_in + drop(acc + bias)         (src/model/backbone.py:99-100,128)
 *   epilogue 4: out = (acc + bias) * keep / (1 - p) * gelu'(aux_in)    (backward of epilogue 1)
 *   epilogue 6: out = drop(gelu(acc + bias)), aux_out = gelu'(acc + bias) * keep / (1 - p) with the same decisions
 *               (epilogue 7 without dropout is then the whole backward of src/model/backbone.py:97-98)
 * other epilogues reject a non-null drop with p > 0. */
int bd_gemm_bf16_drop(const void* A, const void* W, const float* bias, void* out, const void* aux_in,
                      void* aux_out, int M, int n_pad, int n_real, int K, int ldc, int epilogue,
                      const bd_dropout* drop, unsigned site, void* stream);

/* Masked cross-entropy (fp32): weight w[m] = indicator[m] * attention_mask[m] (1 without a mask);
 * forward adds sum(ce * w) to sums[0] and sum(w) to sums[1] (caller zeroes sums) and stores the
 * per-row log-sum-exp; backward writes dlogits = grad_out[0] * w / sums[1] * (softmax - onehot),
 * all zeros when sums[1] == 0. Rows with w == 0 are never read.
 * replaces DiscreteMaskingDiffusion._compute_loss, src/model/diffusion.py:169-209, and its autograd. */
int bd_masked_ce_fwd(const float* logits, const long long* targets, const unsigned char* indicator,
                     const long long* attention_mask, float* lse, float* sums, int M, int V, void* stream);
/* bd_masked_ce_fwd with the two sums spread over n_parts pairs, each in a 128-byte line of its own: sums is
 * [32 * (1 + n_parts)] floats, zeroed by the caller; workgroup w adds to sums[32 * (1 + w % n_parts)] and the float
 * behind it (a cache line then takes n_parts times fewer atomics, which retire one after another). bd_masked_ce_finish adds the pairs into sums[0] and sums[1], where
 * bd_masked_ce_bwd reads them, and writes loss[0] = sums[0] / sums[1], exactly 0 when sums[1] == 0 (the reference
 * returns a zero that is still attached to the graph when no position is both masked and attended,
 * src/model/diffusion.py:203-205). n_parts = 0: sums is the plain pair of bd_masked_ce_fwd. */
int bd_masked_ce_fwd_parts(const float* logits, const long long* targets, const unsigned char* indicator,
                           const long long* attention_mask, float* lse, float* sums, int n_parts, int M, int V,
                           void* stream);
int bd_masked_ce_finish(float* sums, float* loss, int n_parts, void* stream);
int bd_masked_ce_bwd(const float* logits, const long long* targets, const unsigned char* indicator,
                     const long long* attention_mask, const float* lse, const float* sums,
                     const float* grad_out, float* dlogits, int M, int V, void* stream);

/* LayerNorm backward from the saved (mean, rstd): dx = dres + dLN(dy) (dres may be null or alias dx),
 * dgamma += ..., dbeta += ... (autograd of nn.LayerNorm, src/model/backbone.py:126,128,254). */
int bd_layernorm_bwd(const void* dy_bf16, const float* x, const float* stats, const float* gamma,
                     const float* dres, float* dx, float* dgamma, float* dbeta, int M, int d, void* stream);

/* The same, also writing dx rounded to bf16 (dx_bf16 [M, d]; what the next branch's GEMMs read),
 * with optional dropout of that copy: dx_bf16 = bf16(dx * keep / (1 - p)) at (row m, column n) of `site`
 * (the backward of the dropout after fc2, src/model/backbone.py:100). With dx_bf16 == NULL and a dropout given,
 * dx itself is dropped (the backward of the embedding dropout, src/model/backbone.py:247, where no GEMM follows). */
int bd_layernorm_bwd_cast(const void* dy_bf16, const float* x, const float* stats, const float* gamma,
                          const float* dres, float* dx, float* dgamma, float* dbeta, int M, int d,
                          void* dx_bf16, const bd_dropout* drop, unsigned site, void* stream);

/* bd_layernorm_bwd_cast with the gamma / beta sums spread over n_parts copies: workgroup w adds to rows
 * parts[w % n_parts][0][d] (gamma) and parts[w % n_parts][1][d] (beta) of a caller-owned, zeroed [n_parts, 2, d] buffer
 * (one address then takes n_parts times fewer atomics, which retire one after another). bd_layernorm_parts_finish adds
 * the copies of several such buffers to their dgamma / dbeta destinations in one launch and leaves them zeroed.
 * (autograd of nn.LayerNorm, src/model/backbone.py:126,128,254) */
typedef struct bd_ln_parts_job { float* parts; float* dgamma; float* dbeta; } bd_ln_parts_job;
int bd_layernorm_bwd_parts(const void* dy_bf16, const float* x, const float* stats, const float* gamma,
                           const float* dres, float* dx, float* parts, int n_parts, int M, int d,
                           void* dx_bf16, const bd_dropout* drop, unsigned site, void* stream);
int bd_layernorm_parts_finish(const bd_ln_parts_job* jobs, int n_jobs, int n_parts, int d, void* stream);

/* Several weight-gradient products of one token range handed over together (a layer's four: the stacked q | k | v
 * projection, the output projection and the two feed-forward layers). Default: one bd_gemm_wgrad_segments launch per
 * job (token range split over the chip, fp32 atomics). With single_owner = 1 (-1: BD_WGRAD_GROUP in the environment,
 * default 0) and M % 32 == 0:
 * ONE launch for up to four jobs in which every 128 x 128 output tile is owned by one workgroup that walks all M
 * tokens, so the sums do not depend on the order in which workgroups finish (one writer per element: the same bits on
 * every run); measured 15 % slower per training step, hence opt-in. Each job is a bd_gemm_wgrad_segments call's
 * arguments (db entries may be null). Replaces the weight half of loss.backward() like bd_gemm_wgrad_segments
 * (src/training/trainer_backbone.py:437-441). */
typedef struct bd_wgrad_job {
  const void* dY; const void* X;      /* bf16 [M, ldy], [M, ldx] */
  float* dW[3]; float* db[3];         /* fp32 destinations of the n_seg row segments, accumulated into */
  int ldy, ldx, n_seg, seg_rows, K;
} bd_wgrad_job;
int bd_gemm_wgrad_group(const bd_wgrad_job* jobs, int n_jobs, int M, int single_owner, void* stream);

/* dW[N,K] += dY[M,N]^T * X[M,K] (bf16 in, fp32 accumulate with atomics); N, K multiples of 128.
 * (autograd of nn.Linear weights, src/model/backbone.py:25-28,91-92,195). */
int bd_gemm_wgrad_bf16(const void* dY, int ldy, const void* X, int ldx, float* dW, int M, int N, int K,
                       void* stream);

/* The same for up to three row segments of dY at once, each with its own destination:
 * dW[s][seg_rows, K] += dY[:, s*seg_rows : (s+1)*seg_rows]^T * X, and, where db and db[s] are given,
 * db[s][seg_rows] += column sums of that slice (the bias gradient). One launch for the stacked
 * q | k | v gradient of src/model/backbone.py:25-27. dW / db are host arrays of device pointers. */
int bd_gemm_wgrad_segments(const void* dY, int ldy, const void* X, int ldx, float* const* dW,
                           float* const* db, int n_seg, int seg_rows, int M, int K, void* stream);

/* out[n] += sum_m dY[m,n] (bias gradients). */
int bd_colsum_bf16(const void* dY, int ld, float* out, int M, int N, void* stream);

/* Attention backward: dqkv [B*L, 3*H*head_dim] bf16 (dQ | dK | dV) from qkv, the forward output,
 * its gradient and the saved log-sum-exp; delta [B,H,L] fp32 is scratch written by the call.
 * (autograd of F.scaled_dot_product_attention, src/model/backbone.py:54-67). */
int bd_attention_bwd(const void* qkv, const unsigned char* keymask, const void* out, const void* dout,
                     const float* lse, float* delta, void* dqkv, int B, int L, int H, int head_dim,
                     void* stream);

/* bd_attention_bwd without the scratch argument, for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc). No counterpart in the reference. The
 * [B, H, L] fp32 row sums live in a file-scope device array of bd_attention_bwd_static_capacity() = 2^22 entries
 * (16 MiB, held per GPU until the process ends); a larger B*H*L is rejected with BD_ERR_ARG. The call allocates
 * nothing (with the event profiler off). Calls that overlap on one GPU - other streams or threads - share that
 * array and race; calls on one stream, one after another, are safe.
 * bd_attention_bwd_static_capacity() is also the set-up call: it looks the array's address up for the CURRENT
 * device (once per device). Call it on every device before the first bd_attention_bwd_static there that runs under
 * stream capture; without it that first call makes the lookup itself (a runtime query, not a stream operation). */
long long bd_attention_bwd_static_capacity(void);
int bd_attention_bwd_static(const void* qkv, const unsigned char* keymask, const void* out, const void* dout,
                            const float* lse, void* dqkv, int B, int L, int H, int head_dim, void* stream);

/* bd_attention_fwd / bd_attention_bwd with dropout of the attention probabilities
 * (F.scaled_dot_product_attention(dropout_p), src/model/backbone.py:59-66): the softmax is
 * normalised over all visible keys, then each probability is dropped or scaled. `out` given to
 * the backward is the forward's (dropped) output. */
int bd_attention_fwd_drop(const void* qkv, const unsigned char* keymask, void* out, float* lse, int B,
                          int L, int H, int head_dim, const bd_dropout* drop, unsigned site, void* stream);
int bd_attention_bwd_drop(const void* qkv, const unsigned char* keymask, const void* out, const void* dout,
                          const float* lse, float* delta, void* dqkv, int B, int L, int H, int head_dim,
                          const bd_dropout* drop, unsigned site, void* stream);

/* ---- packed rows: the training step without the trailing padding of each sequence.
 * The reference pads every sequence of a batch to a common length (collate_fn / dynamic_collate_fn,
 * src/data/dataset.py:200-240) and runs the denoiser on all B * L rows; rows behind a sequence's last attended
 * position receive no attention as keys, carry no loss and get no gradient (src/model/diffusion.py:190-207), so the
 * training step may leave them out. Packed layout: sequence b occupies rows cu_seqlens[b] .. cu_seqlens[b + 1] - 1
 * (int32 [B + 1], cu_seqlens[0] = 0) of every token-major tensor; rows from cu_seqlens[B] up to the (padded) row
 * count are fillers. Key masks, log-sum-exp / delta buffers and the dropout rows of the attention probabilities keep
 * their [B, L] / [B, H, L] layout; the token-row dropout sites count rows in the packed order. */

/* rowmap[r] = b * L + l for packed row r (the logical token it holds), -1 for filler rows. */
int bd_build_rowmap(const int* cu_seqlens, int B, int L, int n_rows, int* rowmap, void* stream);
/* Targets and loss flags of the packed rows: targets_out[r] = targets[rowmap[r]] (0 for fillers), indicator_out[r] = 1
 * where rowmap[r] >= 0, indicator[rowmap[r]] != 0 and attention_mask[rowmap[r]] != 0 (attention_mask may be null):
 * the valid-position product of src/model/diffusion.py:190-200 on the rows that are computed. */
int bd_pack_targets(const long long* targets, const unsigned char* indicator, const long long* attention_mask,
                    const int* rowmap, int n_rows, long long* targets_out, unsigned char* indicator_out, void* stream);
/* bd_embed_drop on packed rows: x[r] = embedding sum of logical token rowmap[r] (ids [B, L], timesteps [B]); filler
 * rows become zeros. (src/model/backbone.py:234-247) */
int bd_embed_rows(const long long* ids, const long long* timesteps, const int* rowmap, const float* tok_emb,
                  const float* pos_emb, const float* time_emb, float* x, int n_rows, int L, int d, int vocab,
                  int max_pos, int n_timesteps, const bd_dropout* drop, unsigned site, void* stream);
/* bd_attention_fwd_drop / bd_attention_bwd_drop on packed qkv / out / dout / dqkv rows; keymask [B, L], lse and delta
 * [B, H, L] as in the padded form. (src/model/backbone.py:54-67 and its autograd) */
int bd_attention_fwd_varlen(const void* qkv, const unsigned char* keymask, void* out, float* lse,
                            const int* cu_seqlens, int B, int L, int H, int head_dim, const bd_dropout* drop,
                            unsigned site, void* stream);
int bd_attention_bwd_varlen(const void* qkv, const unsigned char* keymask, const void* out, const void* dout,
                            const float* lse, float* delta, void* dqkv, const int* cu_seqlens, int B, int L, int H,
                            int head_dim, const bd_dropout* drop, unsigned site, void* stream);
/* bd_embed_bwd with dx in packed rows (ids [B, L] and timesteps [B] stay logical). */
int bd_embed_bwd_varlen(const float* dx, const long long* ids, const long long* timesteps, float* dtok, float* dpos,
                        float* dtime, const int* cu_seqlens, int B, int L, int d, int pad_id, int vocab, int max_pos,
                        int n_timesteps, void* stream);

/* Scatter-add of the embedding-sum gradient into the three tables (the pad row of the token table
 * receives nothing, as nn.Embedding(padding_idx) does). Table sizes as in bd_embed: L > max_pos is rejected,
 * rows whose index lies outside its table add nothing. (autograd of src/model/backbone.py:234-246). */
int bd_embed_bwd(const float* dx, const long long* ids, const long long* timesteps, float* dtok, float* dpos,
                 float* dtime, int B, int L, int d, int pad_id, int vocab, int max_pos, int n_timesteps,
                 void* stream);

/* out[0] += sum x^2 (global gradient norm of clip_grad_norm_, src/training/trainer_backbone.py:446-449). */
int bd_sumsq_f32(const float* x, long long n, float* out, void* stream);

/* Fused AdamW over flat fp32 buffers with torch.optim.AdamW's update order; when grad_sumsq is
 * given the gradients are first scaled by min(1, max_norm / (sqrt(grad_sumsq[0]) + 1e-6)).
 * replaces optimizer.step() of shared/optimizer_setup.py:290-294 (AdamW branch). */
int bd_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                  float bias_correction2_sqrt, const float* grad_sumsq, float max_norm, void* stream);

/* The update under GradScaler.step's rule (src/training/trainer_backbone.py:129,439-453: scaler.step skips
 * optimizer.step when the unscaled gradients hold an inf or NaN, so parameters, moments and the per-parameter step
 * count stay as they are): when grad_sumsq[0] is not finite the launch changes nothing. guard[0] = steps applied
 * so far; the bias corrections are computed on the device from guard[0] + 1 (betas in double, as torch evaluates
 * 1 - beta ** step), so no host round trip decides them. bd_step_guard_update, launched after every update of the
 * step has been enqueued, adds the step to guard[0] (applied) or guard[1] (skipped). The scale factor itself is
 * not kept: multiplying the loss by a power of two and dividing the gradients by it is exact in fp32. */
int bd_adamw_step_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                          float lr, double beta1, double beta2, float eps, float weight_decay, const int* guard,
                          const float* grad_sumsq, float max_norm, void* stream);
int bd_step_guard_update(int* guard, const float* grad_sumsq, void* stream);

/* Batched C_b = A_b W_b^T over `batch` problems of one shape (element strides sA, sW, sOut; bf16 in and
 * out, row stride N, N % 128 == 0, K % 64 == 0). aux == null: out = acc rounded. aux given (bf16, laid
 * out like out): out = rn(rn(beta * aux) + rn(alpha * rn(acc))), every step rounded to bf16 as the chain
 * of bf16 tensor ops in shared/optimizer_setup.py:60-63 is. */
int bd_gemm_bf16_batched(const void* A, const void* W, void* out, const void* aux, int batch, long long sA,
                         long long sW, long long sOut, int M, int N, int K, float alpha, float beta,
                         void* stream);

/* One Muon update of `nb` matrices [rows, cols] stored back to back in params / grads / momentum_buf
 * (fp32): momentum (lerp, optional Nesterov), normalisation, `ns_steps` quintic Newton-Schulz iterations
 * in bf16 on the matrix cores, p <- p (1 - lr wd) - lr max(1, rows / cols) X. grad_sumsq / max_norm as
 * in bd_adamw_step (global-norm clip folded in). rows, cols multiples of 128.
 * replaces the "muon" group of MuonAdamW.step and _zeropower_via_newtonschulz5,
 * shared/optimizer_setup.py:43-67,137-167. */
long long bd_muon_workspace_bytes(int nb, int rows, int cols);
int bd_muon_step(float* params, const float* grads, float* momentum_buf, void* workspace,
                 long long workspace_bytes, int nb, int rows, int cols, float lr, float weight_decay,
                 float momentum, int nesterov, int ns_steps, const float* grad_sumsq, float max_norm,
                 void* stream);
/* The same, leaving parameters and momentum alone when grad_sumsq[0] is not finite (see bd_adamw_step_guarded). */
int bd_muon_step_guarded(float* params, const float* grads, float* momentum_buf, void* workspace,
                         long long workspace_bytes, int nb, int rows, int cols, float lr, float weight_decay,
                         float momentum, int nesterov, int ns_steps, const float* grad_sumsq, float max_norm,
                         void* stream);

/* Many fp32 -> bf16 copies in one launch: job j copies src [rows, cols] (contiguous) to dst[r * ld + c],
 * or, with transpose == 1, to dst[c * ld + r]; transpose == 2 copies fp32 to fp32 dst[r * ld + c] (stacked
 * biases). jobs_device is a device array; max_rows / max_cols bound
 * every job's shape. (The bf16 weight copies autocast makes per GEMM in the reference's training step,
 * trainer_backbone.py:434-436, plus the transposes the data-gradient GEMMs read.)
 * Head re-blocking (heads whose size the attention kernels do not tile, e.g. 48, run zero-padded to 64): with
 * pad_axis == 1 the source ROW index r, with pad_axis == 2 the source COLUMN index c, is replaced by
 * (i / group) * group_pad + i % group before the store above, and the value is multiplied by scale; the
 * destination's padding entries are never written (the caller zeroes them once). pad_axis == 0: no re-blocking,
 * group / group_pad / scale ignored. */
typedef struct bd_cast_job {
  const float* src;
  void* dst;
  int rows, cols, ld, transpose;
  int pad_axis, group, group_pad;
  float scale;
} bd_cast_job;
int bd_multi_cast_bf16(const bd_cast_job* jobs_device, int n_jobs, int max_rows, int max_cols, void* stream);
/* The way back for gradients, fp32 to fp32: dst [rows, cols] (contiguous) += scale * src[r' * ld + c'] with the
 * re-blocked index r' (pad_axis 1) or c' (pad_axis 2) as above (pad_axis 0, scale 0 is read as 1: a plain add);
 * transpose is ignored. */
int bd_multi_unpad_add_f32(const bd_cast_job* jobs_device, int n_jobs, int max_rows, int max_cols, void* stream);

/* y (bf16) = x (fp32), flat. */
int bd_cast_f32_to_bf16(const float* x, void* y, long long n, void* stream);

/* ---------------------------------------------------------------- reverse step */

#define BD_SAMPLE_MAX_L 512
#define BD_SAMPLE_MAX_V 1024
#define BD_SAMPLE_MAX_D 2048
#define BD_SAMPLE_MAX_RING 128

/* token class bits, tok_flags[v] */
#define BD_TF_BOND 1      /* - = # / \                                  */
#define BD_TF_RING 2      /* ring digit or %nn                          */
#define BD_TF_OPEN 4      /* (                                          */
#define BD_TF_CLOSE 8     /* )                                          */
#define BD_TF_SPECIAL 16  /* always forbidden at MASK positions         */
#define BD_TF_COUNTED 32  /* star / placeholder-bearing token           */
#define BD_TF_EOS 64      /* EOS (forbidden unless natural-EOS mode lets it through) */

/* sampler variant bits */
#define BD_VB_FIXED 1        /* honour fixed_mask when choosing positions (SMILES / BPE)   */
#define BD_VB_SYNTAX 2       /* SMILES parenthesis / ring / bond constraints               */
#define BD_VB_URGENCY 4      /* SELFIES placeholder urgency boost                          */
#define BD_VB_SORTED 8       /* draw the chosen positions left to right (SELFIES)          */
#define BD_VB_NATURAL_EOS 16 /* SELFIES allow_natural_eos=True                              */
/* sample_conditional of the SELFIES / Group_SELFIES samplers (positions are chosen among the non-fixed masks
 * like SMILES, but the in-step edits differ from their sample()):                                         */
#define BD_VB_DYN_ALL_MASKS 32   /* the in-step counted-token update edits every MASK position, fixed or
                                    not (Bi_Diffusion_SELFIES sampler.py:708-709, Group :685-686)          */
#define BD_VB_DYN_SINGLE 64      /* that update is triggered by count_id alone, counts count_id alone and
                                    forbids count_id alone (Group sampler.py:679-687)                      */
#define BD_VB_SPECIAL_UNFIXED 128 /* special / EOS columns are forbidden at non-fixed MASK positions only
                                    (Bi_Diffusion_SELFIES sampler.py:639-664)                              */

typedef struct bd_step_args {
  long long* ids;               /* [B,L] in/out                                             */
  const unsigned char* fixed;   /* [B,L] fixed_mask bytes or null                           */
  long long* attn;              /* [B,L] in/out, touched only in natural-EOS mode, or null  */
  unsigned char* keymask;       /* [B,L] byte mirror of attn, same rule, or null            */
  unsigned char* has_eos;       /* [B] in/out, natural-EOS mode only                        */
  float* logits;                /* LOGITS mode: [B,L,V] raw logits in, edited logits out.
                                   HIDDEN mode: optional [B,L,V] dump of the raw rows used  */
  const void* hidden;           /* HIDDEN mode: [B,L,d] bf16 final-norm output, else null   */
  const void* wout;             /* HIDDEN mode: [>=V,d] bf16 output projection              */
  const float* u_sel;           /* [B,L] selection keys or null (Philox)                    */
  const float* u_tok;           /* [B,L,V] token uniforms or null (Philox)                  */
  const int* ktable;            /* [L+1]: max(1, int(n * (1.0 / t))) for n = 0..L           */
  const unsigned char* tok_flags; /* [V]                                                    */
  const int* contrib;           /* [V] star / placeholder contribution of each token        */
  const int* ring_slot;         /* [V] index into the ring-id list or -1                    */
  const int* counted_ids;       /* [n_counted] tokens with contrib > 0                      */
  const int* bond_ids;          /* [n_bond]                                                 */
  const int* non_special;       /* [n_ns] fallback pool                                     */
  int n_counted, n_bond, n_ns, R;
  int B, L, V, d;
  int mask_id, pad_id, bos_id, eos_id, open_id, close_id, boost_id;
  int count_id;                 /* the direct star / placeholder id (BD_VB_DYN_SINGLE), else unused  */
  int variant, use_constraints, t;
  float temperature;
  unsigned long long seed;
  long long seq_offset;
  /* HIDDEN mode, optional workspace: the raw logits of each sequence's first row_cap draws of the step are
   * computed ahead of the draws as one product on the matrix cores (32 rows share each pass over wout); draws
   * beyond row_cap compute their row one by one as without the workspace. row_cap a power of two <= 32,
   * d in {128, 256, 384, 512}. Which path a draw takes depends on its rank in its own sequence only, so a
   * sequence's result does not depend on the batch it is in.                                               */
  float* row_logits;            /* [B * row_cap, (V + 31) / 32 * 32] or null                           */
  int* row_pos;                 /* [B * row_cap]                                                        */
  int row_cap;
  int rows_chosen;              /* non-zero: row_pos already holds this step's choice (bd_sample_choose_rows ran
                                   on these ids with this row_cap); zero: bd_sample_step fills it itself  */
} bd_step_args;

/* One reverse step t for a batch: static constraints, choice of k positions per sequence,
 * sequential Gumbel-max draws with in-step constraint updates, ids updated in place.
 * replaces the body of the `for t in steps` loop after the backbone call:
 * src/sampling/sampler.py:667-718 (SMILES / SMILES_BPE),
 * Bi_Diffusion_SELFIES/src/sampling/sampler.py:306-393,
 * Bi_Diffusion_Group_SELFIES/src/sampling/sampler.py:363-418. */
int bd_sample_step(const bd_step_args* args, void* stream);

/* The positions step t will draw, known before the forward pass of that step: row_pos[b * row_cap + i] = position
 * of sequence b's i-th draw (draw order), -1 past its last draw. need_all[b] = 1 when the step reads hidden rows of
 * sequence b beyond those (it draws more than row_cap positions, or the SELFIES urgency rule applies, which reads the
 * placeholder logit of every position), else 0. Reads args->ids, fixed, ktable, contrib, u_sel / seed, seq_offset,
 * t, variant, use_constraints and the token ids; row_pos / row_cap of the struct are ignored.
 * The reference chooses with torch.randperm after the forward, from ids the forward does not change:
 * src/sampling/sampler.py:676-686 (the choice does not depend on the logits),
 * Bi_Diffusion_SELFIES/src/sampling/sampler.py:133-152 (urgency condition). */
int bd_sample_choose_rows(const bd_step_args* args, int* row_pos, unsigned char* need_all, int row_cap, void* stream);

typedef struct bd_fix_args {
  long long* ids;               /* [B,L] in/out                                             */
  const float* logits;          /* [B,L,V] final logits (after the t = 1 step's edits)      */
  long long* attn;              /* natural-EOS finalisation only, or null                   */
  unsigned char* keymask;       /* same                                                      */
  unsigned char* has_eos;       /* same                                                      */
  const unsigned char* tok_flags; /* [V] (SMILES repairs)                                   */
  const int* ring_ids;          /* [R]                                                       */
  int excl[8];                  /* ids excluded (besides count_id) when replacing an extra  */
  int n_excl, R;
  int B, L, V;
  int mask_id, pad_id, bos_id, eos_id, open_id, close_id;
  int count_id, target;         /* star / placeholder id and the wanted count (2)           */
  int do_fix, smiles_syntax, natural_eos;
} bd_fix_args;

/* Post-loop repairs: natural-EOS completion, star / placeholder count, parenthesis balance,
 * ring-closure pairing.
 * replaces src/sampling/sampler.py:447-632 (called :720-723),
 * Bi_Diffusion_SELFIES/src/sampling/sampler.py:156-220, 395-414,
 * Bi_Diffusion_Group_SELFIES/src/sampling/sampler.py:247-315. */
int bd_sample_fixup(const bd_fix_args* args, void* stream);

/* Token ids -> text: out[b, :] = the texts of row b's tokens joined in position order, out_len[b] = its length
 * in bytes (a value above `stride` means the row did not fit and was cut). text / offsets [V + 1]: the tokens'
 * UTF-8 texts back to back; a token that decode skips (the special tokens) has an empty text, ids outside 0..V-1
 * decode to nothing ('[UNK]' is a special token).
 * replaces tokenizer.batch_decode(ids.cpu().tolist(), skip_special_tokens=True), src/sampling/sampler.py:724 with
 * src/data/tokenizer.py:207-224,110-121 (same loop in the SMILES_BPE and SELFIES tokenizers). */
int bd_decode_ids(const long long* ids, int B, int L, const unsigned char* text, const int* offsets, int V,
                  unsigned char* out, int* out_len, int stride, void* stream);

/* Host-side evaluation of the device uniform / Gumbel functions (testing aid; the product
 * never needs it): out[i] = gumbel(u[i]); Philox words for a counter/key. */
void bd_host_gumbel(const float* u, float* out, long long n);
void bd_host_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                    uint32_t out[4]);
/* Host evaluation of the training-dropout decision stream: out[r * cols + c] = keep / (1 - p). */
int bd_host_dropout_factors(const bd_dropout* drop, unsigned site, long long rows, long long cols, float* out);
float bd_host_uniform_select(unsigned long long seed, uint32_t seq, uint32_t t, uint32_t pos);
float bd_host_uniform_token(unsigned long long seed, uint32_t seq, uint32_t t, uint32_t pos,
                            uint32_t v);

/* ---------------------------------------------------------------- event profiling */

/* When enabled, every bd_gemm_bf16 launch is bracketed by HIP events on its stream so that
 * bench.py can report the kernel's average duration. bd_prof_read synchronises the recorded
 * events and returns launches / total milliseconds / total flops since the last reset. */
int bd_prof_enable(int on);
int bd_prof_read(long long* launches, double* total_ms, double* total_flops, int reset);
/* Same, for one kernel family: 0 tiled GEMM, 1 row-panel GEMM, 2 fused feed-forward block,
 * 3 fused attention block, 4 whole-forward kernel, 5 weight-gradient GEMM, 6 attention forward,
 * 7 attention backward; and, with algorithmic BYTES in place of flops, 8 LayerNorm forward / backward,
 * 9 masked cross-entropy, 10 embedding / corruption / dropout / casts, 11 optimizer, 12 reverse step and
 * repairs; -1 = all families together. */
int bd_prof_read_kind(int kind, long long* launches, double* total_ms, double* total_flops, int reset);

/* The chosen-row forward for models on the per-layer kernels (d >= 512 or L > 128: no single-launch forward). Every
 * layer but the last runs for every row; in the last layer attention still needs every row's keys and values, and
 * behind it only the rows in row_pos [B * row_cap] (int32 positions, -1 = empty slot; bd_sample_choose_rows writes
 * them) go through the output projection, the feed-forward block and the final LayerNorm, as B * row_cap gathered
 * rows. hidden_out [B*L, d] bf16 receives those rows, its other rows keep what they held. The rows written equal
 * bd_backbone_forward's bit for bit: bd_backbone_forward_rows_unfused_supported answers 1 only where the full forward
 * runs all four projections of a block on the eight-wave GEMM at B * L rows without fused or panel kernels and
 * without the deferred LayerNorm (cs_*_ln null), and the gathered rows' products run on that same kernel
 * (bd_gemm_bf16_tile256: the eight-wave kernel whatever the tile count; epilogues 0, 1, 2 of bd_gemm_bf16, N a
 * multiple of 128, K a multiple of 64, K >= 128). There is no need_all list here: a caller whose step flagged a
 * sequence (bd_sample_choose_rows) runs bd_backbone_forward for that step. rows_workspace: scratch of
 * bd_backbone_rows_workspace_bytes(desc, B, row_cap) bytes, 256-byte aligned, beside the forward's own workspace.
 * Replaces the same call sites as bd_backbone_forward_rows (src/sampling/sampler.py:654-700). */
int bd_backbone_forward_rows_unfused_supported(const bd_backbone_desc* desc, int B, int L);
long long bd_backbone_rows_workspace_bytes(const bd_backbone_desc* desc, int B, int row_cap);
int bd_backbone_forward_rows_unfused(const bd_backbone_desc* desc, const long long* ids, const long long* timesteps,
                                     const unsigned char* keymask, int B, int L, const int* row_pos, int row_cap,
                                     void* workspace, long long workspace_bytes, void* rows_workspace,
                                     long long rows_workspace_bytes, void* hidden_out, void* stream);
int bd_gemm_bf16_tile256(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int epilogue,
                         void* stream);

/* Clock diagnostics of the single-launch forward (bd_backbone_mega and the entry points that use it): while a
 * buffer is set, workgroup b < n_workgroups of every such launch writes four 64-bit words to stamps[4 b ..]:
 * the shader-clock counter and the 100 MHz real-time counter when it starts, and both again when it ends. The
 * sustained clock of a launch is (end - start) of the first counter over (end - start) of the second, times
 * 100 MHz, e.g. as the median over workgroups: it tells a box that holds a lower clock under this load from a
 * slower kernel. stamps = null (the initial state) turns the stamps off; process-wide, not thread-safe. */
int bd_clock_stamps_set(unsigned long long* stamps, int n_workgroups);

#ifdef __cplusplus
}
#endif
#endif /* BIDIFF_H */
