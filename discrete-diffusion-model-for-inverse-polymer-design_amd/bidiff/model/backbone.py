"""Bidirectional pre-LN transformer denoiser, executed by the HIP library.

Drop-in for the reference's ``src/model/backbone.py`` (``DiffusionBackbone``, :132-377):
same constructor arguments, sub-module / parameter names (hence the same ``state_dict`` keys
and checkpoint layout), the same seeded initialisation, and the same public methods. The
modules below hold the parameters; ``DiffusionBackbone`` runs the whole stack in ``libbidiff_hip.so`` through
``bd_backbone_forward`` (inference) or the training-step kernels (autograd path) and never calls the
block-level modules, whose own ``forward`` (inference only, ``_blocks.py``) serves callers that apply a
single block or projection.
"""
import ctypes as C
from typing import Optional

import math

import torch
import torch.nn as nn

from .. import _lib


class MultiHeadAttention(nn.Module):
    """q/k/v/out projections (reference backbone.py:9-78)."""

    def __init__(self, hidden_size: int, num_heads: int, dropout: float = 0.1):
        super().__init__()
        if hidden_size % num_heads != 0:
            raise AssertionError("hidden_size must be divisible by num_heads")
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.head_dim = hidden_size // num_heads
        self.q_proj = nn.Linear(hidden_size, hidden_size)
        self.k_proj = nn.Linear(hidden_size, hidden_size)
        self.v_proj = nn.Linear(hidden_size, hidden_size)
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """out_proj(softmax(q k^T / sqrt(dh) + key padding) v) of x [B, L, d]; attention_mask [B, L], non-zero =
        visible key; rows whose keys are all padded come out as the output bias (reference backbone.py:32-78)."""
        from ._blocks import attention_forward
        return attention_forward(self, x, attention_mask)


class FeedForward(nn.Module):
    """fc1 -> GELU -> fc2 (reference backbone.py:81-101)."""

    def __init__(self, hidden_size: int, ffn_hidden_size: int, dropout: float = 0.1):
        super().__init__()
        self.fc1 = nn.Linear(hidden_size, ffn_hidden_size)
        self.fc2 = nn.Linear(ffn_hidden_size, hidden_size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """fc2(gelu(fc1(x))) (reference backbone.py:95-101)."""
        from ._blocks import ffn_forward
        return ffn_forward(self, x)


class TransformerBlock(nn.Module):
    """One pre-norm block (reference backbone.py:104-129)."""

    def __init__(self, hidden_size: int, num_heads: int, ffn_hidden_size: int, dropout: float = 0.1):
        super().__init__()
        self.norm1 = nn.LayerNorm(hidden_size)
        self.attn = MultiHeadAttention(hidden_size, num_heads, dropout)
        self.norm2 = nn.LayerNorm(hidden_size)
        self.ffn = FeedForward(hidden_size, ffn_hidden_size, dropout)

    def forward(self, x: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x + attn(norm1(x)), then + ffn(norm2(.)) (reference backbone.py:120-129)."""
        from ._blocks import block_forward
        return block_forward(self, x, attention_mask)


F16_MAX = 65504.0
F16_MIN_NORMAL = 2.0 ** -14


def _fits_f16_feed_forward(w1ln, b1ln, w2, d) -> bool:
    """May a layer's feed-forward block run in the fused kernel, whose GELU and second product are f16?

    The first product's outputs must stay inside f16: a LayerNorm output row has norm <= sqrt(d), so
    |fc1 output j| <= |row j of W1 * gamma| * sqrt(d) + |bias j|, and the largest such bound must be below 65504 (past it
    the packed conversion saturates and GELU(x) stops growing with x). W2 is stored as f16 there: exact for bf16
    values of normal f16 size, while entries below 2^-14 = 6.1e-5 lose bits (counted as lost altogether here); a row
    may lose at most 2^-9 of its absolute sum that way, the rounding bf16 itself brings. One device -> host copy per
    layer, when the weights are packed."""
    bound = w1ln.float().norm(dim=1) * math.sqrt(d) + b1ln.float().abs()
    w = w2.detach().to(torch.bfloat16).float().abs()
    lost = torch.where((w < F16_MIN_NORMAL) | (w > F16_MAX), w, torch.zeros_like(w)).sum(dim=1)
    ok = (bound.max() < F16_MAX) & torch.isfinite(w).all() & (lost <= w.sum(dim=1) * 2.0 ** -9).all()
    return bool(ok.item())


class _Packed:
    """bf16 copies of the matrices + the C descriptor pointing at them."""

    def __init__(self):
        self.key = None
        self.tensors = []
        self.layers = None
        self.desc = None
        self.w_out = None


class DiffusionBackbone(nn.Module):
    """Transformer denoiser for discrete masking diffusion (reference backbone.py:132-377)."""

    def __init__(
        self,
        vocab_size: int,
        hidden_size: int = 768,
        num_layers: int = 12,
        num_heads: int = 12,
        ffn_hidden_size: int = 3072,
        max_position_embeddings: int = 256,
        num_diffusion_steps: int = 100,
        dropout: float = 0.1,
        pad_token_id: int = 0,
    ):
        super().__init__()
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.ffn_hidden_size = ffn_hidden_size
        self.max_position_embeddings = max_position_embeddings
        self.num_diffusion_steps = num_diffusion_steps
        self.dropout = dropout
        self.pad_token_id = pad_token_id

        self.token_embedding = nn.Embedding(vocab_size, hidden_size, padding_idx=pad_token_id)
        self.position_embedding = nn.Embedding(max_position_embeddings, hidden_size)
        self.timestep_embedding = nn.Embedding(num_diffusion_steps + 1, hidden_size)
        self.embedding_dropout = nn.Dropout(dropout)
        self.layers = nn.ModuleList(
            [TransformerBlock(hidden_size, num_heads, ffn_hidden_size, dropout) for _ in range(num_layers)]
        )
        self.final_norm = nn.LayerNorm(hidden_size)
        self.output_proj = nn.Linear(hidden_size, vocab_size, bias=False)
        self._init_weights()
        self._packed = _Packed()
        self._ws = None
        # "bf16": bf16 MFMA GEMMs with fp32 accumulation (logits within 2e-2 of the reference's fp32 forward);
        # "fp32": every product in fp32 on the f32-input MFMA (within 1e-5), inference only, about 12x slower
        self.compute_dtype = "bf16"
        self._packed32 = None
        # Training dropout: keep decisions are a function of (dropout_seed, forward-pass counter,
        # site, element). dropout_seed None = drawn from torch's CPU generator at the first
        # training forward (so torch.manual_seed fixes it) and offset by the process rank.
        self.dropout_seed: Optional[int] = None
        self._drop_step = 0

    # ------------------------------------------------------------------ init
    def _init_weights(self):
        """Xavier-uniform matrices, zero biases, N(0, 0.02) embeddings with a zero pad row,
        unit LayerNorms - visited in ``self.modules()`` order so that a given torch seed
        produces the reference's weights (reference backbone.py:200-213)."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, mean=0.0, std=0.02)
                if m.padding_idx is not None:
                    nn.init.zeros_(m.weight[m.padding_idx])
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _next_dropout(self):
        """bd_dropout of the next training forward pass, or None when dropout is off (eval mode or
        p == 0; reference: nn.Dropout and SDPA ``dropout_p if self.training``, backbone.py:30,65)."""
        if not self.training or self.dropout <= 0.0:
            return None
        if not self.dropout < 1.0:
            raise ValueError(f"dropout must be in [0, 1), got {self.dropout}")
        if self.dropout_seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                seed = (seed + 0x9E3779B97F4A7C15 * torch.distributed.get_rank()) % (2 ** 64)
            self.dropout_seed = seed
        drop = _lib.Dropout(float(self.dropout), self._drop_step & 0xFFFFFFFF, int(self.dropout_seed) % (2 ** 64))
        self._drop_step += 1
        return drop

    # ------------------------------------------------------------------ packing
    def _params(self):
        """The module's parameters as a tuple, cached: walking the module tree costs about 0.2 ms, and the
        training step asks five times. The set of Parameter objects is fixed after construction; ``_apply``
        (``.to()``, ``.cuda()``, ...) may replace them and drops the cache."""
        pc = self.__dict__.get("_param_cache")
        if pc is None:
            pc = tuple(self.parameters())
            self.__dict__["_param_cache"] = pc
        return pc

    def _apply(self, fn, *args, **kwargs):
        self.__dict__["_param_cache"] = None
        return super()._apply(fn, *args, **kwargs)

    # True: large models (per-layer kernels on the eight-wave GEMM) run LayerNorm deferred across the residual
    # projections (bd_gemm_bf16_deferred_ln) instead of a LayerNorm kernel in front of every projection. Off by default:
    # measured on the medium / large sizes and the 24-layer stress model it takes the same time (what the LayerNorm
    # kernels cost moves into the projections' epilogues), and it rounds the rows before the mean is taken out.
    defer_layernorm = False

    def _pack_key(self):
        return (_lib.weights_generation(), self.defer_layernorm) + tuple((p.data_ptr(), p._version) for p in self._params())

    def packed(self) -> _Packed:
        """bf16 weight copies and the bd_backbone_desc; rebuilt when any parameter changed."""
        key = self._pack_key()
        pk = self._packed
        if pk.key == key:
            return pk
        dev = self.output_proj.weight.device
        if dev.type != "cuda":
            raise RuntimeError("bidiff DiffusionBackbone: move the model to a ROCm device first "
                               "(no CPU execution path)")
        d, F, V = self.hidden_size, self.ffn_hidden_size, self.vocab_size
        keep = []

        def bf(w):
            t = w.detach().to(torch.bfloat16).contiguous()
            keep.append(t)
            return t

        def f32(w):
            t = w.detach().to(torch.float32).contiguous()
            keep.append(t)
            return t

        layers = (_lib.LayerWeights * max(1, self.num_layers))()
        L_ = _lib.lib()
        # head sizes the attention kernels do not have (e.g. 48) are zero-padded to 32 / 64 / 128: extra
        # q / k components add nothing to the scores, extra v components meet zero columns of out_proj;
        # the q rows absorb sqrt(padded / true) because the kernel scales by 1 / sqrt(padded)
        H = self.num_heads
        dh = d // H
        dhp = dh if dh in (32, 64, 128) else (32 if dh < 32 else 64 if dh < 64 else 128)
        if dh > 128:
            raise NotImplementedError("bidiff attention kernels support head_dim <= 128")
        aw = H * dhp
        if aw % 128:
            raise NotImplementedError(f"bidiff: num_heads * padded head_dim = {aw} must be a multiple of 128")

        def pad_rows(w):       # [H*dh, ...] -> [H*dhp, ...]
            if dhp == dh:
                return w
            out = w.new_zeros((H, dhp) + tuple(w.shape[1:]))
            out[:, :dh] = w.view((H, dh) + tuple(w.shape[1:]))
            return out.view((H * dhp,) + tuple(w.shape[1:]))

        qfix = math.sqrt(dhp / dh)
        with torch.no_grad():
            for i, blk in enumerate(self.layers):
                a = blk.attn
                wq32 = torch.cat([pad_rows(a.q_proj.weight.float() * qfix), pad_rows(a.k_proj.weight.float()),
                                  pad_rows(a.v_proj.weight.float())], dim=0)
                bq32 = torch.cat([pad_rows(a.q_proj.bias.float() * qfix), pad_rows(a.k_proj.bias.float()),
                                  pad_rows(a.v_proj.bias.float())], dim=0)
                wqkv = bf(wq32)
                bqkv = f32(bq32)
                wo32 = a.out_proj.weight.float()
                if dhp != dh:
                    wo32 = pad_rows(wo32.t().contiguous()).t().contiguous()       # zero input columns
                lw = layers[i]
                lw.ln1_g = f32(blk.norm1.weight).data_ptr(); lw.ln1_b = f32(blk.norm1.bias).data_ptr()
                lw.w_qkv = wqkv.data_ptr(); lw.b_qkv = bqkv.data_ptr()
                lw.w_o = bf(wo32).data_ptr(); lw.b_o = f32(a.out_proj.bias).data_ptr()
                lw.ln2_g = f32(blk.norm2.weight).data_ptr(); lw.ln2_b = f32(blk.norm2.bias).data_ptr()
                lw.w_fc1 = bf(blk.ffn.fc1.weight).data_ptr(); lw.b_fc1 = f32(blk.ffn.fc1.bias).data_ptr()
                lw.w_fc2 = bf(blk.ffn.fc2.weight).data_ptr(); lw.b_fc2 = f32(blk.ffn.fc2.bias).data_ptr()
                # LayerNorm folded into the following projection: LN(x) W^T + b =
                # ((x - mean) * rstd) (W * gamma)^T + (b + W beta)   (fp32 fold, then one bf16 rounding)
                wqln = bf(wq32 * blk.norm1.weight.float()[None, :])
                lw.w_qkv_ln = wqln.data_ptr()
                lw.b_qkv_ln = f32(bqkv + wq32 @ blk.norm1.bias.float()).data_ptr()
                # row sums of the folded weights AS ROUNDED (the deferred LayerNorm subtracts mean * these from what
                # the matrix cores accumulated over the same bf16 values)
                if self.defer_layernorm:
                    lw.cs_qkv_ln = f32(wqln.float().sum(dim=1)).data_ptr()
                # weight stream of the fused attention block: q rows (and bias) also carry the
                # softmax scale log2(e) / sqrt(head_dim), see attn_block.hip
                if L_.bd_attn_block_supported(d, self.num_heads, 1) and L_.bd_attn_block_width(d, self.num_heads) == aw:
                    # the q rows already carry sqrt(padded / true) for the kernels that scale by 1 / sqrt(padded);
                    # this stream wants log2(e) / sqrt(true head size) in total
                    sc = math.log2(math.e) / math.sqrt(dh) / qfix
                    wa = (wq32 * blk.norm1.weight.float()[None, :]).clone()
                    ba = (bqkv + wq32 @ blk.norm1.bias.float()).clone()
                    wa[:aw] *= sc
                    ba[:aw] *= sc
                    wab = bf(wa)
                    apack = torch.empty(L_.bd_attn_pack_bytes_heads(d, self.num_heads), dtype=torch.uint8, device=dev)
                    keep.append(apack)
                    _lib.check(L_.bd_attn_pack_weights(wab.data_ptr(), lw.w_o, apack.data_ptr(), d, self.num_heads,
                                                       _lib.stream_handle()), "bd_attn_pack_weights")
                    lw.attn_pack = apack.data_ptr()
                    lw.b_q_attn = f32(ba[:aw]).data_ptr()
                    # v bias: softmax rows sum to one, so it passes through attention unchanged
                    lw.b_o_fold = f32(a.out_proj.bias.float() + wo32 @ ba[2 * aw:]).data_ptr()
                w1 = blk.ffn.fc1.weight.float()
                w1ln = bf(w1 * blk.norm2.weight.float()[None, :])
                lw.w_fc1_ln = w1ln.data_ptr()
                if self.defer_layernorm:
                    lw.cs_fc1_ln = f32(w1ln.float().sum(dim=1)).data_ptr()
                lw.b_fc1_ln = f32(blk.ffn.fc1.bias.float() + w1 @ blk.norm2.bias.float()).data_ptr()
                # weight stream of the fused feed-forward kernel (fragment order, see ffn_fused.hip); it computes GELU
                # and the second product in f16, which not every weight set fits: such layers keep the bf16 kernels
                if L_.bd_ffn_fused_supported(d, F) and _fits_f16_feed_forward(w1 * blk.norm2.weight.float()[None, :],
                                                                           blk.ffn.fc1.bias.float() + w1 @ blk.norm2.bias.float(),
                                                                           blk.ffn.fc2.weight, d):
                    pack = torch.empty(L_.bd_ffn_pack_bytes(d, F), dtype=torch.uint8, device=dev)
                    keep.append(pack)
                    _lib.check(L_.bd_ffn_pack_weights(w1ln.data_ptr(), lw.w_fc2, pack.data_ptr(), d, F,
                                                      _lib.stream_handle()), "bd_ffn_pack_weights")
                    lw.ffn_pack = pack.data_ptr()
            vpad = ((V + 127) // 128) * 128
            w_out = torch.zeros(vpad, d, dtype=torch.bfloat16, device=dev)
            w_out[:V] = self.output_proj.weight.detach().to(torch.bfloat16)
            keep.append(w_out)
            desc = _lib.BackboneDesc()
            desc.d, desc.n_layers, desc.n_heads, desc.ffn = d, self.num_layers, self.num_heads, F
            desc.vocab, desc.vocab_pad = V, vpad
            desc.max_pos, desc.n_timesteps = self.max_position_embeddings, self.num_diffusion_steps + 1
            desc.ln_eps = float(self.final_norm.eps)
            desc.tok_emb = f32(self.token_embedding.weight).data_ptr()
            desc.pos_emb = f32(self.position_embedding.weight).data_ptr()
            desc.time_emb = f32(self.timestep_embedding.weight).data_ptr()
            desc.lnf_g = f32(self.final_norm.weight).data_ptr()
            desc.lnf_b = f32(self.final_norm.bias).data_ptr()
            desc.w_out = w_out.data_ptr()
            desc.layers = C.cast(layers, C.POINTER(_lib.LayerWeights))
            desc.attn_width = aw
        pk.key, pk.tensors, pk.layers, pk.desc, pk.w_out = key, keep, layers, desc, w_out
        return pk

    def _workspace(self, pk, B, L, device):
        need = _lib.lib().bd_backbone_workspace_bytes(C.byref(pk.desc), B, L)
        if need < 0:
            _lib.check(-1, "bd_backbone_workspace_bytes")
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws, need

    # ------------------------------------------------------------------ native forward
    def _check_inputs(self, input_ids, timesteps):
        _lib.require_device(input_ids, "DiffusionBackbone.forward")
        if input_ids.dim() != 2:
            raise ValueError("input_ids must be [batch, seq_len]")
        if timesteps.dim() == 2:
            timesteps = timesteps.squeeze(-1)
        if timesteps.shape[0] != input_ids.shape[0]:
            raise ValueError("timesteps must have one entry per sequence")
        if input_ids.shape[1] > self.max_position_embeddings:
            # nn.Embedding raises the same error type for a position past the table (reference backbone.py:238-243)
            raise IndexError(f"sequence length {input_ids.shape[1]} exceeds max_position_embeddings "
                             f"{self.max_position_embeddings}")
        return timesteps.to(device=input_ids.device, dtype=torch.long).contiguous()

    @staticmethod
    def _keymask_bytes(attention_mask, like):
        if attention_mask is None:
            return None
        if attention_mask.dtype == torch.uint8 and attention_mask.is_contiguous():
            return attention_mask
        return (attention_mask != 0).to(torch.uint8).contiguous()

    def _fp32_weights(self):
        """Stacked q|k|v weight and bias (one GEMM for the three projections), rebuilt when a parameter changed."""
        key = self._pack_key()
        pk = self._packed32
        if pk is not None and pk[0] == key:
            return pk[1]
        layers = []
        for blk in self.layers:
            a = blk.attn
            layers.append((torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0).detach().float().contiguous(),
                           torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0).detach().float().contiguous()))
        self._packed32 = (key, layers)
        return layers

    def _forward_fp32(self, ids, timesteps, keymask, want_hidden, want_logits, upto=None):
        """The forward in fp32: the reference's operation sequence (backbone.py:215-257 with the blocks of
        :120-129), one kernel per operation. Returns (final-norm hidden fp32 [B,L,d] or the residual stream after
        ``upto`` blocks, logits fp32 or None)."""
        L_ = _lib.lib()
        st = _lib.stream_handle()
        B, L = ids.shape
        M, d, F, V, H = B * L, self.hidden_size, self.ffn_hidden_size, self.vocab_size, self.num_heads
        dev = ids.device
        if d % 16 or F % 16:
            raise ValueError("bidiff fp32 mode needs hidden and ffn sizes that are multiples of 16")
        for p_ in self._params():
            if p_.dtype != torch.float32 or not p_.is_contiguous():
                raise RuntimeError("bidiff fp32 mode expects contiguous fp32 parameters")

        def gemm(A, W, bias, out, N, K, epi=0, resid=None):
            _lib.check(L_.bd_gemm_f32(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(bias), _lib.ptr(resid), _lib.ptr(out), N,
                                      M, N, K, epi, st), "bd_gemm_f32")

        def ln(x, norm, out):
            _lib.check(L_.bd_layernorm(_lib.ptr(x), _lib.ptr(norm.weight), _lib.ptr(norm.bias), None, _lib.ptr(out), None,
                                       M, d, float(norm.eps), st), "bd_layernorm")

        x = torch.empty(M, d, device=dev)
        _lib.check(L_.bd_embed(_lib.ptr(ids), _lib.ptr(timesteps), _lib.ptr(self.token_embedding.weight),
                               _lib.ptr(self.position_embedding.weight), _lib.ptr(self.timestep_embedding.weight),
                               _lib.ptr(x), B, L, d, self.vocab_size, self.max_position_embeddings,
                               self.num_diffusion_steps + 1, st), "bd_embed")
        h = torch.empty(M, d, device=dev)
        qkv = torch.empty(M, 3 * d, device=dev)
        att = torch.empty(M, d, device=dev)
        u = torch.empty(M, F, device=dev)
        n_blocks = self.num_layers if upto is None else upto
        for blk, (wqkv, bqkv) in list(zip(self.layers, self._fp32_weights()))[:n_blocks]:
            ln(x, blk.norm1, h)
            gemm(h, wqkv, bqkv, qkv, 3 * d, d)
            _lib.check(L_.bd_attention_f32(_lib.ptr(qkv), _lib.ptr(keymask), _lib.ptr(att), B, L, H, d // H, st),
                       "bd_attention_f32")
            gemm(att, blk.attn.out_proj.weight, blk.attn.out_proj.bias, x, d, d, epi=2, resid=x)
            ln(x, blk.norm2, h)
            gemm(h, blk.ffn.fc1.weight, blk.ffn.fc1.bias, u, F, d, epi=1)
            gemm(u, blk.ffn.fc2.weight, blk.ffn.fc2.bias, x, d, F, epi=2, resid=x)
        if upto is not None:
            return x.view(B, L, d), None
        ln(x, self.final_norm, h)
        logits = None
        if want_logits:
            logits = torch.empty(M, V, device=dev)
            gemm(h, self.output_proj.weight, None, logits, V, d)
            logits = logits.view(B, L, V)
        return (h.view(B, L, d) if want_hidden else None), logits

    def ragged_wave_table(self, lengths, L, device=None):
        """Deal sequences of the given attended lengths (host integers, one per sequence of a [B, L] batch) to the
        32-row waves of the single-launch forward's workgroups: a sequence takes ceil(length / 32) adjacent waves,
        first-fit by decreasing size. Returns the int32 [n_blocks, 4] table for ``run_forward(wave_table=...)``, or
        None when this model / L has no such launch or the table would not save a tenth of the workgroups."""
        if self.compute_dtype != "bf16" or L > 128:
            return None
        pk = self.packed()
        if _lib.lib().bd_backbone_mega_ragged_supported(C.byref(pk.desc), int(L)) != 1:
            return None
        need = [max(1, min(4, -(-min(int(n), L) // 32))) for n in lengths]
        B = len(need)
        per_block = 4 if L <= 32 else 2 if L <= 64 else 1            # what the launch does by itself
        table, free = [], {1: [], 2: [], 3: []}
        for i in sorted(range(B), key=lambda k: -need[k]):
            w = need[i]
            blk = None
            for f in range(w, 4):
                if free[f]:
                    blk, have = free[f].pop(), f
                    break
            if blk is None:
                table.append([-1, -1, -1, -1])
                blk, have = len(table) - 1, 4
            row = table[blk]
            row[4 - have:4 - have + w] = [i] * w
            if have - w > 0:
                free[have - w].append(blk)
        if len(table) * 10 > -(-B // per_block) * 9:
            return None
        dev = device if device is not None else self.token_embedding.weight.device
        return torch.tensor(table, dtype=torch.int32).to(dev)

    def wave_table_for_mask(self, mask):
        """``ragged_wave_table`` for a [B, L] attention / key mask on the device (its last non-zero column + 1 is a
        sequence's length; one device -> host copy), or None when nothing would be saved."""
        B, L = mask.shape
        if not (32 < L <= 128) or self.compute_dtype != "bf16" or B == 0:
            return None
        lens = ((mask != 0) * torch.arange(1, L + 1, device=mask.device)).amax(1)
        if int(lens.min().item()) > (L - 1) // 32 * 32:          # nobody needs fewer waves than the padded length
            return None
        return self.ragged_wave_table(lens.tolist(), L, mask.device)

    def run_forward(self, input_ids, timesteps, keymask, want_hidden=True, want_logits=False,
                    hidden_out=None, logits_out=None, wave_table=None):
        """One inference forward on the current stream. keymask: uint8 [B,L] or None.
        Returns (hidden bf16 [B,L,d] or None, logits fp32 [B,L,V] or None).
        wave_table (int32 [n_blocks, 4] on the device, from ``ragged_wave_table``): the single-launch forward with
        sequences of different lengths dealt to the waves of its workgroups; rows beyond a sequence's waves are
        not computed (hidden rows there keep what ``hidden_out`` held)."""
        timesteps = self._check_inputs(input_ids, timesteps)
        ids = input_ids.to(torch.long).contiguous()
        B, L = ids.shape
        if self.compute_dtype == "fp32":
            # fp32 mode: hidden comes back in fp32 (not the bf16 rows of the fused path)
            hidden, logits = self._forward_fp32(ids, timesteps, keymask, want_hidden, want_logits)
            if hidden_out is not None and hidden is not None:
                hidden_out.copy_(hidden)
                hidden = hidden_out
            if logits_out is not None and logits is not None:
                logits_out.copy_(logits)
                logits = logits_out
            return hidden, logits
        if self.compute_dtype != "bf16":
            raise ValueError(f"compute_dtype must be 'bf16' or 'fp32', got {self.compute_dtype!r}")
        pk = self.packed()
        ws, need = self._workspace(pk, B, L, ids.device)
        if want_hidden and hidden_out is None:
            hidden_out = torch.empty(B, L, self.hidden_size, dtype=torch.bfloat16, device=ids.device)
        if want_logits and logits_out is None:
            logits_out = torch.empty(B, L, self.vocab_size, dtype=torch.float32, device=ids.device)
        if wave_table is not None:
            if wave_table.dtype != torch.int32 or wave_table.dim() != 2 or wave_table.shape[1] != 4 \
                    or not wave_table.is_contiguous() or wave_table.device != ids.device:
                raise ValueError("wave_table must be a contiguous int32 [n_blocks, 4] tensor on the inputs' device")
            rc = _lib.lib().bd_backbone_forward_ragged(
                C.byref(pk.desc), _lib.ptr(ids), _lib.ptr(timesteps), _lib.ptr(keymask), _lib.ptr(wave_table),
                int(wave_table.shape[0]), B, L, _lib.ptr(ws), need, _lib.ptr(hidden_out) if want_hidden else None,
                _lib.ptr(logits_out) if want_logits else None, _lib.stream_handle())
            _lib.check(rc, "bd_backbone_forward_ragged")
            return (hidden_out if want_hidden else None), (logits_out if want_logits else None)
        rc = _lib.lib().bd_backbone_forward(
            C.byref(pk.desc), _lib.ptr(ids), _lib.ptr(timesteps), _lib.ptr(keymask), B, L,
            _lib.ptr(ws), need, _lib.ptr(hidden_out) if want_hidden else None,
            _lib.ptr(logits_out) if want_logits else None, _lib.stream_handle())
        _lib.check(rc, "bd_backbone_forward")
        return (hidden_out if want_hidden else None), (logits_out if want_logits else None)

    def rows_forward_supported(self, L) -> bool:
        """Does ``run_forward_rows`` exist for this model and sequence length (the single-launch forward does)?"""
        if self.compute_dtype != "bf16":
            return False
        return _lib.lib().bd_backbone_forward_rows_supported(C.byref(self.packed().desc), int(L)) == 1

    def rows_forward_mode(self, B, L):
        """Which chosen-row forward this model has for a [B, L] batch: "single_launch" (``need_all`` handled on the
        device), "per_layer" (large models on the per-layer kernels: the caller runs the full forward for a step that
        flagged a sequence in ``need_all``), or None."""
        if self.compute_dtype != "bf16":
            return None
        desc = C.byref(self.packed().desc)
        if _lib.lib().bd_backbone_forward_rows_supported(desc, int(L)) == 1:
            return "single_launch"
        if _lib.lib().bd_backbone_forward_rows_unfused_supported(desc, int(B), int(L)) == 1:
            return "per_layer"
        return None

    def run_forward_rows(self, input_ids, timesteps, keymask, row_pos, need_all, row_cap, hidden_out, wave_table=None):
        """The inference forward for a reverse step that reads few rows: ``hidden_out`` [B, L, d] bf16 receives the
        final-norm rows at positions ``row_pos`` [B * row_cap] (int32, -1 = none) and every row of the sequences
        flagged in ``need_all`` [B] (uint8); its other rows keep what they held (``bd_sample_choose_rows`` writes
        both lists). The rows written equal ``run_forward``'s bit for bit."""
        timesteps = self._check_inputs(input_ids, timesteps)
        ids = input_ids.to(torch.long).contiguous()
        B, L = ids.shape
        pk = self.packed()
        ws, need = self._workspace(pk, B, L, ids.device)
        if hidden_out.dtype != torch.bfloat16 or not hidden_out.is_contiguous() or tuple(hidden_out.shape) != (B, L, self.hidden_size):
            raise ValueError("hidden_out must be a contiguous bf16 [B, L, hidden] tensor")
        if row_pos.dtype != torch.int32 or row_pos.numel() != B * row_cap or need_all.dtype != torch.uint8 \
                or need_all.numel() != B:
            raise ValueError("row_pos must be int32 [B * row_cap] and need_all uint8 [B]")
        if _lib.lib().bd_backbone_forward_rows_supported(C.byref(pk.desc), L) != 1:
            # per-layer kernels: chosen rows only behind the last layer's attention (no need_all list: see rows_forward_mode)
            rneed = _lib.lib().bd_backbone_rows_workspace_bytes(C.byref(pk.desc), B, int(row_cap))
            if rneed < 0:
                _lib.check(-1, "bd_backbone_rows_workspace_bytes")
            rws = self.__dict__.get("_rows_ws")
            if rws is None or rws.numel() < rneed or rws.device != ids.device:
                rws = torch.empty(rneed, dtype=torch.uint8, device=ids.device)
                self.__dict__["_rows_ws"] = rws
            rc = _lib.lib().bd_backbone_forward_rows_unfused(
                C.byref(pk.desc), _lib.ptr(ids), _lib.ptr(timesteps), _lib.ptr(keymask), B, L, _lib.ptr(row_pos),
                int(row_cap), _lib.ptr(ws), need, _lib.ptr(rws), rneed, _lib.ptr(hidden_out), _lib.stream_handle())
            _lib.check(rc, "bd_backbone_forward_rows_unfused")
            return hidden_out
        n_blocks = 0
        if wave_table is not None:
            if wave_table.dtype != torch.int32 or wave_table.dim() != 2 or wave_table.shape[1] != 4 \
                    or not wave_table.is_contiguous() or wave_table.device != ids.device:
                raise ValueError("wave_table must be a contiguous int32 [n_blocks, 4] tensor on the inputs' device")
            n_blocks = int(wave_table.shape[0])
        rc = _lib.lib().bd_backbone_forward_rows(
            C.byref(pk.desc), _lib.ptr(ids), _lib.ptr(timesteps), _lib.ptr(keymask), _lib.ptr(wave_table), n_blocks,
            B, L, _lib.ptr(row_pos), _lib.ptr(need_all), int(row_cap), _lib.ptr(ws), need, _lib.ptr(hidden_out),
            _lib.stream_handle())
        _lib.check(rc, "bd_backbone_forward_rows")
        return hidden_out

    # ------------------------------------------------------------------ reference API
    def forward(self, input_ids: torch.Tensor, timesteps: torch.Tensor,
                attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Logits [batch, seq_len, vocab] (fp32, computed with bf16 MFMA GEMMs).

        With autograd recording and parameters that require grad this runs the training
        path (activations saved for the HIP backward kernels); otherwise the inference path.
        """
        _lib.require_device(input_ids, "DiffusionBackbone.forward")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._params()):
            from ..training.autograd import backbone_forward_train
            return backbone_forward_train(self, input_ids, timesteps, attention_mask)
        keymask = self._keymask_bytes(attention_mask, input_ids)
        _, logits = self.run_forward(input_ids, timesteps, keymask, want_hidden=False, want_logits=True)
        return logits

    def get_hidden_states(self, input_ids, timesteps, attention_mask=None, layer_idx: int = -1):
        """Hidden states [batch, seq_len, hidden] (reference backbone.py:259-299): ``layer_idx == -1``
        gives the final-norm output of the whole stack, ``layer_idx = i`` the residual stream after
        block i (no final norm)."""
        if layer_idx == -1 and torch.is_grad_enabled() and any(p.requires_grad for p in self._params()):
            # trainable backbone under autograd (a property head fine-tuning the last blocks): the training path
            from ..training.autograd import backbone_forward_train
            return backbone_forward_train(self, input_ids, timesteps, attention_mask, hidden_states=True)
        keymask = self._keymask_bytes(attention_mask, input_ids)
        if layer_idx != -1:
            if not 0 <= layer_idx < self.num_layers:
                raise IndexError(f"layer_idx {layer_idx} out of range for {self.num_layers} layers")
            timesteps = self._check_inputs(input_ids, timesteps)
            B, L = input_ids.shape
            if self.compute_dtype == "fp32":
                return self._forward_fp32(input_ids.to(torch.long).contiguous(), timesteps, keymask, False, False,
                                          upto=layer_idx + 1)[0]
            pk = self.packed()
            ws, need = self._workspace(pk, B, L, input_ids.device)
            ids = input_ids.to(torch.long).contiguous()
            out = torch.empty(B, L, self.hidden_size, dtype=torch.float32, device=input_ids.device)
            rc = _lib.lib().bd_backbone_forward_layers(
                C.byref(pk.desc), _lib.ptr(ids), _lib.ptr(timesteps), _lib.ptr(keymask), B, L, _lib.ptr(ws), need,
                layer_idx + 1, _lib.ptr(out), _lib.stream_handle())
            _lib.check(rc, "bd_backbone_forward_layers")
            return out
        hidden, _ = self.run_forward(input_ids, timesteps, keymask, want_hidden=True, want_logits=False)
        return hidden.float()

    def get_pooled_output(self, input_ids, timesteps, attention_mask=None, pooling: str = "mean"):
        """Masked mean / cls / max pooling of the final hidden states, [batch, hidden] fp32 (reference
        backbone.py:301-335). Frozen backbone or no_grad: one pooling kernel over the bf16 hidden rows the
        forward wrote. With trainable backbone parameters under autograd (PropertyPredictor with
        finetune_last_layers > 0) the hidden states come from the training path and the pooling is
        the reference's torch expression, so gradients reach the unfrozen blocks."""
        modes = {"mean": 0, "cls": 1, "max": 2}
        if pooling not in modes:
            raise ValueError(f"Unknown pooling method: {pooling}")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._params()):
            hs = self.get_hidden_states(input_ids, timesteps, attention_mask)
            if pooling == "cls":
                return hs[:, 0]
            if attention_mask is None:
                return hs.mean(dim=1) if pooling == "mean" else hs.max(dim=1)[0]
            m = attention_mask.unsqueeze(-1).float()
            if pooling == "mean":
                return (hs * m).sum(dim=1) / m.sum(dim=1).clamp(min=1e-9)
            return (hs * m + (1 - m) * (-1e9)).max(dim=1)[0]
        keymask = self._keymask_bytes(attention_mask, input_ids)
        B, L = input_ids.shape
        # pooling under a mask reads attended rows only (and row 0): rows of 32-row blocks past a sequence's length are
        # not computed, short sequences share workgroups of the forward (ragged_pooling = False: every padded row)
        table = None
        if attention_mask is not None and getattr(self, "ragged_pooling", True):
            table = self.wave_table_for_mask(attention_mask)
        if table is not None:
            hidden = torch.zeros(B, L, self.hidden_size, dtype=torch.bfloat16, device=input_ids.device)
            self.run_forward(input_ids, timesteps, keymask, want_hidden=True, want_logits=False, hidden_out=hidden,
                             wave_table=table)
        else:
            hidden, _ = self.run_forward(input_ids, timesteps, keymask, want_hidden=True, want_logits=False)
        if self.compute_dtype == "fp32":        # the pooling kernel reads bf16 rows; fp32 rows are pooled by torch
            if pooling == "cls":
                return hidden[:, 0, :].contiguous()
            if attention_mask is None:
                return hidden.mean(dim=1) if pooling == "mean" else hidden.max(dim=1)[0]
            m = attention_mask.unsqueeze(-1).to(hidden.dtype)
            if pooling == "mean":
                return (hidden * m).sum(dim=1) / m.sum(dim=1).clamp(min=1)
            return hidden.masked_fill(m == 0, float("-inf")).max(dim=1)[0]
        am = None if attention_mask is None else attention_mask.to(torch.long).contiguous()
        out = torch.empty(B, self.hidden_size, dtype=torch.float32, device=input_ids.device)
        rc = _lib.lib().bd_pool_hidden(_lib.ptr(hidden), _lib.ptr(am), _lib.ptr(out), B, L, self.hidden_size,
                                       modes[pooling], _lib.stream_handle())
        _lib.check(rc, "bd_pool_hidden")
        return out

    def config_dict(self):
        return {
            "vocab_size": self.vocab_size,
            "hidden_size": self.hidden_size,
            "num_layers": self.num_layers,
            "num_heads": self.num_heads,
            "ffn_hidden_size": self.ffn_hidden_size,
            "max_position_embeddings": self.max_position_embeddings,
            "num_diffusion_steps": self.num_diffusion_steps,
            "dropout": self.dropout,
            "pad_token_id": self.pad_token_id,
        }

    def save_pretrained(self, path: str):
        """{'state_dict', 'config'} checkpoint, the reference's layout (backbone.py:337-356)."""
        torch.save({"state_dict": self.state_dict(), "config": self.config_dict()}, path)

    @classmethod
    def load_pretrained(cls, path: str, device: str = "cpu", **kwargs) -> "DiffusionBackbone":
        """Load a save_pretrained checkpoint. Tensors only: the file is read with
        ``weights_only=True`` (the reference passes False, backbone.py:370)."""
        ckpt = torch.load(path, map_location=device, weights_only=True)
        cfg = dict(ckpt["config"])
        cfg.update(kwargs)
        model = cls(**cfg)
        model.load_state_dict(ckpt["state_dict"])
        return model
