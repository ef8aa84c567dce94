"""Forward of the block-level modules on their own (``MultiHeadAttention``, ``FeedForward``, ``TransformerBlock``).

The reference's modules are callable (src/model/backbone.py:32-78, :95-101, :120-129); ``DiffusionBackbone`` here
never calls them (it runs the whole stack in fused launches), but a caller that applies a single block or projection
gets the same arithmetic from the same kernels: bf16 operands on the matrix cores with fp32 accumulation, fp32
LayerNorm / residual / softmax, exactly the unfused path of the backbone. Inference only: under autograd with
trainable parameters these raise (training runs through ``DiffusionBackbone`` / ``DiscreteMaskingDiffusion``), and
dropout is not applied (``eval()`` semantics)."""
import torch

from .. import _lib


def _pad_rows(w: torch.Tensor, mult: int = 128) -> torch.Tensor:
    n = w.shape[0]
    n_pad = (n + mult - 1) // mult * mult
    if n_pad == n:
        return w.contiguous()
    out = torch.zeros(n_pad, w.shape[1], dtype=w.dtype, device=w.device)
    out[:n] = w
    return out


class _Bf16Copies:
    """bf16 (row-padded) copies of a module's matrices, rebuilt when a parameter changed."""

    def __init__(self):
        self.key, self.t = None, None

    def get(self, params, build):
        key = (_lib.weights_generation(),) + tuple((p.data_ptr(), p._version, p.device) for p in params)
        if key != self.key:
            with torch.no_grad():
                self.t = build()
            self.key = key
        return self.t


def _check(x: torch.Tensor, what: str, params):
    _lib.require_device(x, what)
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
        raise NotImplementedError(f"{what}: the block-level forward is inference-only (call it under torch.no_grad(); "
                                  f"training runs through DiffusionBackbone)")
    if x.dim() != 3:
        raise ValueError(f"{what}: expected [batch, seq_len, hidden], got {tuple(x.shape)}")


def _gemm(a_bf16, w_bf16, bias, out, M, n_real, K, epilogue):
    n_pad = w_bf16.shape[0]
    if K % 64 != 0:
        raise NotImplementedError(f"bidiff GEMM needs an inner size that is a multiple of 64, got {K}")
    _lib.check(_lib.lib().bd_gemm_bf16(_lib.ptr(a_bf16), _lib.ptr(w_bf16), _lib.ptr(bias), _lib.ptr(out), M, n_pad, n_real, K,
                                       out.shape[-1], epilogue, _lib.stream_handle()), "bd_gemm_bf16")


def _cast(x_f32):
    y = torch.empty(x_f32.shape, dtype=torch.bfloat16, device=x_f32.device)
    _lib.check(_lib.lib().bd_cast_f32_to_bf16(_lib.ptr(x_f32), _lib.ptr(y), x_f32.numel(), _lib.stream_handle()), "bd_cast_f32_to_bf16")
    return y


def _layernorm(x_f32, norm, M, d):
    y = torch.empty(M, d, dtype=torch.bfloat16, device=x_f32.device)
    _lib.check(_lib.lib().bd_layernorm(_lib.ptr(x_f32), _lib.ptr(norm.weight), _lib.ptr(norm.bias), _lib.ptr(y), None, None, M, d,
                                       float(norm.eps), _lib.stream_handle()), "bd_layernorm")
    return y


def _keymask(attention_mask, B, L, device):
    if attention_mask is None:
        return None
    if tuple(attention_mask.shape) != (B, L):
        raise ValueError(f"attention_mask must be [batch, seq_len] = {(B, L)}, got {tuple(attention_mask.shape)}")
    return (attention_mask.to(device) != 0).to(torch.uint8).contiguous()


def _attn_weights(mha):
    def build():
        wqkv = torch.cat([mha.q_proj.weight, mha.k_proj.weight, mha.v_proj.weight], 0).detach()
        bqkv = torch.cat([mha.q_proj.bias, mha.k_proj.bias, mha.v_proj.bias], 0).detach().float().contiguous()
        return (_pad_rows(wqkv.to(torch.bfloat16)), bqkv, _pad_rows(mha.out_proj.weight.detach().to(torch.bfloat16)),
                mha.out_proj.bias.detach().float().contiguous())
    if not hasattr(mha, "_bf16"):
        object.__setattr__(mha, "_bf16", _Bf16Copies())
    return mha._bf16.get(list(mha.parameters()), build)


def _ffn_weights(ffn):
    def build():
        return (_pad_rows(ffn.fc1.weight.detach().to(torch.bfloat16)), ffn.fc1.bias.detach().float().contiguous(),
                _pad_rows(ffn.fc2.weight.detach().to(torch.bfloat16)), ffn.fc2.bias.detach().float().contiguous())
    if not hasattr(ffn, "_bf16"):
        object.__setattr__(ffn, "_bf16", _Bf16Copies())
    return ffn._bf16.get(list(ffn.parameters()), build)


def _attention_core(mha, a_bf16, keymask, B, L, out_f32, epilogue):
    """out_f32 (epilogue 3: = / epilogue 2: +=) out_proj(attention(q, k, v of a_bf16))."""
    d, H, dh = mha.hidden_size, mha.num_heads, mha.head_dim
    if dh not in (32, 64, 128):
        raise NotImplementedError(f"the stand-alone attention forward supports head sizes 32, 64 and 128, got {dh} "
                                  f"(DiffusionBackbone runs other head sizes zero-padded)")
    M = B * L
    wqkv, bqkv, wo, bo = _attn_weights(mha)
    if 3 * d % 8 != 0:
        raise NotImplementedError("hidden size must be a multiple of 8")
    qkv = torch.empty(M, 3 * d, dtype=torch.bfloat16, device=a_bf16.device)
    _gemm(a_bf16, wqkv, bqkv, qkv, M, 3 * d, d, 0)
    att = torch.empty(M, d, dtype=torch.bfloat16, device=a_bf16.device)
    _lib.check(_lib.lib().bd_attention_fwd(_lib.ptr(qkv), _lib.ptr(keymask), _lib.ptr(att), None, B, L, H, dh,
                                           _lib.stream_handle()), "bd_attention_fwd")
    _gemm(att, wo, bo, out_f32, M, d, d, epilogue)


def _ffn_core(ffn, a_bf16, M, out_f32, epilogue):
    w1, b1, w2, b2 = _ffn_weights(ffn)
    d, F = ffn.fc1.in_features, ffn.fc1.out_features
    h = torch.empty(M, F, dtype=torch.bfloat16, device=a_bf16.device)
    _gemm(a_bf16, w1, b1, h, M, F, d, 1)
    _gemm(h, w2, b2, out_f32, M, d, F, epilogue)


def attention_forward(mha, x, attention_mask=None):
    params = list(mha.parameters())
    _check(x, "MultiHeadAttention.forward", params)
    B, L, d = x.shape
    xf = x.detach().to(torch.float32).contiguous()
    out = torch.empty(B * L, d, dtype=torch.float32, device=x.device)
    _attention_core(mha, _cast(xf.view(B * L, d)), _keymask(attention_mask, B, L, x.device), B, L, out, 3)
    return out.view(B, L, d)


def ffn_forward(ffn, x):
    params = list(ffn.parameters())
    _check(x, "FeedForward.forward", params)
    B, L, d = x.shape
    xf = x.detach().to(torch.float32).contiguous()
    out = torch.empty(B * L, d, dtype=torch.float32, device=x.device)
    _ffn_core(ffn, _cast(xf.view(B * L, d)), B * L, out, 3)
    return out.view(B, L, d)


def block_forward(block, x, attention_mask=None):
    params = list(block.parameters())
    _check(x, "TransformerBlock.forward", params)
    B, L, d = x.shape
    M = B * L
    out = x.detach().to(torch.float32).contiguous().clone().view(M, d)      # the residual stream, updated in place
    _attention_core(block.attn, _layernorm(out, block.norm1, M, d), _keymask(attention_mask, B, L, x.device), B, L, out, 2)
    _ffn_core(block.ffn, _layernorm(out, block.norm2, M, d), M, out, 2)
    return out.view(B, L, d)
