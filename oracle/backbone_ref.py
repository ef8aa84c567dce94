"""TEST INFRASTRUCTURE - fp32 PyTorch restatement of the denoiser forward and the loss.

Functional (takes a state_dict with the reference's key names), CPU or any torch device.
Follows Bi_Diffusion_SMILES/src/model/backbone.py:
  embeddings :234-247, block :120-129, attention :46-78 (SDPA with a boolean key mask,
  a fully masked query row gives an all-zero attention output), FFN :95-101 (erf GELU),
  final norm + vocabulary projection :254-255;
and Bi_Diffusion_SMILES/src/model/diffusion.py: schedule :60-71, loss :169-209.
Dropout is the identity (eval mode / dropout=0) unless a ``DropoutStream`` is passed: then the
four dropout sites of the reference (embedding sum :247, attention probabilities :65, GELU output
:98, fc2 output :100) multiply by the factors of the counter-based decision stream that the device
kernels regenerate (``sampler_ref.dropout_mask``), so both sides drop the same elements.
"""
import math

import torch
import torch.nn.functional as F


class DropoutStream:
    """Dropout factors (0 or 1 / (1 - p)) per site: 0 = embedding sum [B*L, d]; block i:
    4 i + 1 = attention probabilities [(b H + h) L + query, key], 4 i + 2 = GELU output [B*L, F],
    4 i + 3 = fc2 output [B*L, d]."""

    def __init__(self, p, seed, step=0, rowmap=None):
        """``rowmap`` (packed-row training path): int array, rowmap[r] = logical token row b * L + l that physical
        row r holds (-1: filler). The token-row sites then count rows in the packed order: logical row
        rowmap[r] takes the factors of stream row r; rows that are left out keep factor 1 (they influence nothing)."""
        self.p, self.seed, self.step = float(p), int(seed), int(step)
        self.rowmap = None if rowmap is None else [int(v) for v in rowmap]

    def factors(self, site, rows, cols, like):
        from . import sampler_ref
        if self.rowmap is not None and site % 4 != 1:
            import numpy as np
            phys = sampler_ref.dropout_mask(self.seed, self.step, site, self.p, len(self.rowmap), cols)
            m = np.ones((rows, cols), dtype=np.float32)
            rm = np.asarray(self.rowmap)
            keep = rm >= 0
            m[rm[keep]] = phys[keep]
            return torch.from_numpy(m).to(device=like.device, dtype=like.dtype)
        m = sampler_ref.dropout_mask(self.seed, self.step, site, self.p, rows, cols)
        return torch.from_numpy(m).to(device=like.device, dtype=like.dtype)


def attention(x, sd, prefix, num_heads, attention_mask, dropout=None, site=0):
    B, L, d = x.shape
    dh = d // num_heads

    def proj(name):
        return F.linear(x, sd[f"{prefix}.{name}.weight"], sd[f"{prefix}.{name}.bias"])
    q = proj("q_proj").view(B, L, num_heads, dh).transpose(1, 2)
    k = proj("k_proj").view(B, L, num_heads, dh).transpose(1, 2)
    v = proj("v_proj").view(B, L, num_heads, dh).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-2, -1)) * (1.0 / math.sqrt(dh))
    if attention_mask is not None:
        vis = attention_mask.to(torch.bool)[:, None, None, :]
        scores = scores.masked_fill(~vis, float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    probs = torch.nan_to_num(probs, nan=0.0)      # all keys masked -> zero output row
    if dropout is not None:
        probs = probs * dropout.factors(site, B * num_heads * L, L, probs).view(B, num_heads, L, L)
    out = torch.matmul(probs, v).transpose(1, 2).contiguous().view(B, L, d)
    return F.linear(out, sd[f"{prefix}.out_proj.weight"], sd[f"{prefix}.out_proj.bias"])


def hidden_states(sd, cfg, input_ids, timesteps, attention_mask=None, prefix="", eps=1e-5, layer_idx=-1,
                  dropout=None):
    """Hidden states [B, L, d] (fp32): final-norm output for ``layer_idx == -1``, otherwise the residual
    stream after block ``layer_idx`` without the final norm (reference backbone.py:291-299)."""
    p = prefix
    B, L = input_ids.shape
    d = cfg["hidden_size"]
    if timesteps.dim() == 2:
        timesteps = timesteps.squeeze(-1)
    pos = torch.arange(L, device=input_ids.device)
    x = (sd[p + "token_embedding.weight"][input_ids] + sd[p + "position_embedding.weight"][pos][None]
         + sd[p + "timestep_embedding.weight"][timesteps][:, None, :])
    if dropout is not None:
        x = x * dropout.factors(0, B * L, d, x).view(B, L, d)
    for i in range(cfg["num_layers"] if layer_idx == -1 else layer_idx + 1):
        lp = f"{p}layers.{i}"
        h = F.layer_norm(x, (d,), sd[f"{lp}.norm1.weight"], sd[f"{lp}.norm1.bias"], eps)
        x = x + attention(h, sd, f"{lp}.attn", cfg["num_heads"], attention_mask, dropout, 4 * i + 1)
        h = F.layer_norm(x, (d,), sd[f"{lp}.norm2.weight"], sd[f"{lp}.norm2.bias"], eps)
        h = F.linear(h, sd[f"{lp}.ffn.fc1.weight"], sd[f"{lp}.ffn.fc1.bias"])
        h = F.gelu(h)
        if dropout is not None:
            h = h * dropout.factors(4 * i + 2, B * L, h.shape[-1], h).view_as(h)
        h = F.linear(h, sd[f"{lp}.ffn.fc2.weight"], sd[f"{lp}.ffn.fc2.bias"])
        if dropout is not None:
            h = h * dropout.factors(4 * i + 3, B * L, d, h).view_as(h)
        x = x + h
    if layer_idx != -1:
        return x
    return F.layer_norm(x, (d,), sd[p + "final_norm.weight"], sd[p + "final_norm.bias"], eps)


def pooled_output(sd, cfg, input_ids, timesteps, attention_mask=None, pooling="mean", prefix=""):
    """Masked mean / cls / max pooling of the final hidden states (reference backbone.py:301-335)."""
    hs = hidden_states(sd, cfg, input_ids, timesteps, attention_mask, prefix)
    if pooling == "mean":
        if attention_mask is None:
            return hs.mean(dim=1)
        m = attention_mask.unsqueeze(-1).float()
        return (hs * m).sum(dim=1) / m.sum(dim=1).clamp(min=1e-9)
    if pooling == "cls":
        return hs[:, 0]
    if pooling == "max":
        if attention_mask is not None:
            m = attention_mask.unsqueeze(-1).float()
            hs = hs * m + (1 - m) * (-1e9)
        return hs.max(dim=1)[0]
    raise ValueError(f"Unknown pooling method: {pooling}")


def property_head(head_sd, pooled):
    """The regression MLP in eval mode: Linear -> ReLU per hidden layer, then Linear(., 1)
    (reference property_head.py:32-47,60-69; state_dict keys ``mlp.<3 i>.weight/bias``)."""
    n = len([k for k in head_sd if k.endswith(".weight")])
    x = pooled
    for i in range(n):
        x = F.linear(x, head_sd[f"mlp.{3 * i}.weight"], head_sd[f"mlp.{3 * i}.bias"])
        if i < n - 1:
            x = F.relu(x)
    return x


def logits(sd, cfg, input_ids, timesteps, attention_mask=None, prefix="", dropout=None):
    h = hidden_states(sd, cfg, input_ids, timesteps, attention_mask, prefix, dropout=dropout)
    return F.linear(h, sd[prefix + "output_proj.weight"])


def mask_schedule(num_steps, beta_min=0.05, beta_max=0.95, force_clean_t0=False):
    """fp32 evaluation in the reference's operand order (diffusion.py:67-70)."""
    steps = torch.arange(num_steps + 1, dtype=torch.float32)
    s = beta_min + (beta_max - beta_min) * steps / num_steps
    if force_clean_t0:
        s[0] = 0.0
    return s


def masked_ce(logits_t, targets, mask_indicator, attention_mask=None):
    """Mean CE over masked & non-pad positions; exactly 0 when there is none (diffusion.py:187-209)."""
    V = logits_t.shape[-1]
    ce = F.cross_entropy(logits_t.reshape(-1, V).float(), targets.reshape(-1), reduction="none").view_as(targets)
    valid = mask_indicator.float()
    if attention_mask is not None:
        valid = valid * attention_mask.float()
    cnt = valid.sum()
    if cnt.item() == 0:
        return logits_t.sum() * 0.0
    return (ce * valid).sum() / cnt
