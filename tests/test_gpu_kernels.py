"""GPU parity of the individual HIP kernels, called through the C ABI, against fp32 PyTorch
references on the same (bf16-rounded) inputs."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from bidiff import _lib

pytestmark = pytest.mark.gpu


def _s():
    return _lib.stream_handle()


def gemm(A, W, bias, out, n_real, epi):
    M, K = A.shape
    rc = _lib.lib().bd_gemm_bf16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), M, W.shape[0], n_real, K,
                                 out.shape[1], epi, _s())
    _lib.check(rc, "gemm")
    torch.cuda.synchronize()


def test_gemm_layout_exact_integers(gpu_device):
    """Small-integer operands make every product exact: any fragment / permutation mistake shows
    as a hard mismatch (asymmetric operands, A != identity)."""
    dev = gpu_device
    M, N, K = 200, 256, 128
    g = torch.Generator().manual_seed(0)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    W = torch.randint(-3, 4, (N, K), generator=g).float()
    bias = torch.randint(-5, 6, (N,), generator=g).float()
    ref = A @ W.t() + bias
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    gemm(A.to(dev).bfloat16(), W.to(dev).bfloat16(), bias.to(dev), out, N, 0)
    assert torch.equal(out.float().cpu(), ref.bfloat16().float())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 384, 384), (1000, 1152, 384), (257, 384, 1536),
                                   (64, 1536, 384), (4096, 384, 384)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_epilogues(gpu_device, M, N, K, epi):
    dev = gpu_device
    g = torch.Generator().manual_seed(M + N + K + epi)
    A = (torch.randn(M, K, generator=g) * 0.7).bfloat16()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, generator=g) * 0.3
    acc = A.float() @ W.float().t() + bias
    if epi == 0:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ref = acc
    elif epi == 1:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ref = torch.nn.functional.gelu(acc)
    else:
        x0 = torch.randn(M, N, generator=g)
        out = x0.clone().to(dev)
        ref = x0 + acc
    gemm(A.to(dev), W.to(dev), bias.to(dev), out, N, epi)
    got = out.float().cpu()
    tol = 2e-2 if epi != 2 else 1e-4
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < tol, err


@pytest.mark.parametrize("V", [39, 43, 233, 803])
def test_gemm_ragged_vocab_head(gpu_device, V):
    dev = gpu_device
    M, K = 333, 384
    g = torch.Generator().manual_seed(V)
    A = torch.randn(M, K, generator=g).bfloat16()
    Wr = (torch.randn(V, K, generator=g) / math.sqrt(K)).bfloat16()
    vpad = (V + 127) // 128 * 128
    W = torch.zeros(vpad, K, dtype=torch.bfloat16)
    W[:V] = Wr
    out = torch.full((M, V), float("nan"), device=dev)
    gemm(A.to(dev), W.to(dev), None, out, V, 3)
    ref = A.float() @ Wr.float().t()
    got = out.cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 1e-4


def ref_attention(qkv, mask, B, L, Hh, dh):
    d = Hh * dh
    x = qkv.float().view(B, L, 3, Hh, dh)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    if mask is not None:
        s = s.masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
    p = torch.nan_to_num(torch.softmax(s, dim=-1), nan=0.0)
    return (p @ v).transpose(1, 2).reshape(B * L, d)


@pytest.mark.parametrize("dh,Hh", [(32, 4), (64, 2), (64, 6), (128, 1), (128, 3)])
@pytest.mark.parametrize("L", [24, 128, 150, 512])
def test_attention_key_padding(gpu_device, dh, Hh, L):
    dev = gpu_device
    B = 5
    d = Hh * dh
    g = torch.Generator().manual_seed(dh * 1000 + L)
    qkv = torch.randn(B * L, 3 * d, generator=g).bfloat16()
    lengths = [L, max(1, L // 3), 1, 0, L - 5]
    mask = torch.zeros(B, L, dtype=torch.uint8)
    for b, n in enumerate(lengths):
        mask[b, :n] = 1
    mask[4, 2] = 0  # a hole in the middle
    out = torch.full((B * L, d), float("nan"), dtype=torch.bfloat16, device=dev)
    d_qkv, d_mask = qkv.to(dev), mask.to(dev)   # keep the device copies alive across the launch
    rc = _lib.lib().bd_attention_fwd(_lib.ptr(d_qkv), _lib.ptr(d_mask), _lib.ptr(out), None, B, L, Hh,
                                     dh, _s())
    _lib.check(rc, "attention")
    torch.cuda.synchronize()
    got = out.float().cpu()
    ref = ref_attention(qkv, mask, B, L, Hh, dh)
    assert torch.isfinite(got).all()
    assert got.view(B, L, d)[3].abs().max().item() == 0.0, "fully padded sequence must give exact zeros"
    assert (got - ref).abs().max().item() < 2e-2
    # no mask at all
    rc = _lib.lib().bd_attention_fwd(_lib.ptr(d_qkv), None, _lib.ptr(out), None, B, L, Hh, dh, _s())
    _lib.check(rc, "attention")
    torch.cuda.synchronize()
    assert (out.float().cpu() - ref_attention(qkv, None, B, L, Hh, dh)).abs().max().item() < 2e-2


@pytest.mark.parametrize("M,d", [(7, 128), (1000, 384), (513, 768), (64, 1024), (33, 2048)])
def test_layernorm(gpu_device, M, d):
    dev = gpu_device
    g = torch.Generator().manual_seed(d + M)
    x = torch.randn(M, d, generator=g) * 3 + 1.5
    gm = torch.randn(d, generator=g)
    bt = torch.randn(d, generator=g)
    ob = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
    of = torch.empty(M, d, device=dev)
    st = torch.empty(M, 2, device=dev)
    dx, dg, db = x.to(dev), gm.to(dev), bt.to(dev)
    rc = _lib.lib().bd_layernorm(_lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(ob),
                                 _lib.ptr(of), _lib.ptr(st), M, d, 1e-5, _s())
    _lib.check(rc, "layernorm")
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x, (d,), gm, bt, 1e-5)
    assert (of.cpu() - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    assert (ob.float().cpu() - ref).abs().max().item() < 2e-2 * ref.abs().max().item()
    assert (st[:, 0].cpu() - x.mean(1)).abs().max().item() < 1e-5


def test_embed(gpu_device):
    dev = gpu_device
    B, L, d, V, T = 4, 10, 128, 39, 21
    g = torch.Generator().manual_seed(1)
    tok, pos, tim = torch.randn(V, d, generator=g), torch.randn(64, d, generator=g), torch.randn(T, d, generator=g)
    ids = torch.randint(0, V, (B, L), generator=g)
    t = torch.randint(0, T, (B,), generator=g)
    x = torch.empty(B * L, d, device=dev)
    dv = [a.to(dev) for a in (ids, t, tok, pos, tim)]
    rc = _lib.lib().bd_embed(_lib.ptr(dv[0]), _lib.ptr(dv[1]), _lib.ptr(dv[2]), _lib.ptr(dv[3]),
                             _lib.ptr(dv[4]), _lib.ptr(x), B, L, d, V, 64, T, _s())
    _lib.check(rc, "embed")
    torch.cuda.synchronize()
    ref = tok[ids] + pos[:L][None] + tim[t][:, None]
    assert torch.equal(x.cpu().view(B, L, d), ref)
    # a sequence longer than the position table is rejected; an id or timestep outside its table reads
    # nothing and marks its row (nn.Embedding raises for all three)
    assert _lib.lib().bd_embed(_lib.ptr(dv[0]), _lib.ptr(dv[1]), _lib.ptr(dv[2]), _lib.ptr(dv[3]),
                               _lib.ptr(dv[4]), _lib.ptr(x), B, L, d, V, L - 1, T, _s()) != 0
    bad = ids.clone()
    bad[1, 3] = V
    bad[2, 0] = -1
    tb = t.clone()
    tb[3] = T
    bad_d, tb_d = bad.to(dev), tb.to(dev)
    _lib.check(_lib.lib().bd_embed(_lib.ptr(bad_d), _lib.ptr(tb_d), _lib.ptr(dv[2]), _lib.ptr(dv[3]),
                                   _lib.ptr(dv[4]), _lib.ptr(x), B, L, d, V, 64, T, _s()), "embed")
    got = x.cpu().view(B, L, d)
    nan_rows = torch.isnan(got).all(-1)
    want = torch.zeros(B, L, dtype=torch.bool)
    want[1, 3] = want[2, 0] = True
    want[3] = True
    assert torch.equal(nan_rows, want) and torch.equal(got[~want], ref[~want])


def test_bad_arguments_are_rejected(gpu_device):
    lib = _lib.lib()
    a = torch.zeros(8, 64, dtype=torch.bfloat16, device=gpu_device)
    assert lib.bd_gemm_bf16(_lib.ptr(a), _lib.ptr(a), None, _lib.ptr(a), 8, 100, 100, 64, 100, 0, _s()) < 0
    assert b"multiple" in lib.bd_last_error()
    assert lib.bd_attention_fwd(_lib.ptr(a), None, _lib.ptr(a), None, 1, 8, 1, 48, _s()) < 0
    assert lib.bd_gemm_bf16(None, _lib.ptr(a), None, _lib.ptr(a), 8, 128, 128, 64, 128, 0, _s()) < 0


def panel(A, W, bias, out, resid, epi, ln):
    M, K = A.shape
    N = W.shape[0]
    rc = _lib.lib().bd_gemm_panel(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(resid), M, N, K,
                                  out.shape[1], epi, int(ln), 1e-5, _s())
    _lib.check(rc, "panel gemm")
    torch.cuda.synchronize()


def test_panel_gemm_layout_exact_integers(gpu_device):
    dev = gpu_device
    g = torch.Generator().manual_seed(0)
    for (M, N, K) in [(200, 384, 384), (130, 128, 256), (77, 384, 1536), (300, 1152, 384)]:
        A = torch.randint(-3, 4, (M, K), generator=g).float()
        W = torch.randint(-3, 4, (N, K), generator=g).float()
        bias = torch.randint(-5, 6, (N,), generator=g).float()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        panel(A.to(dev).bfloat16(), W.to(dev).bfloat16(), bias.to(dev), out, None, 0, False)
        assert torch.equal(out.float().cpu(), (A @ W.t() + bias).bfloat16().float()), (M, N, K)


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (300, 384, 384), (1000, 1152, 384), (257, 384, 1536),
                                   (64, 1536, 384), (4096, 384, 384), (500, 128, 256), (333, 256, 128),
                                   (700, 768, 768), (200, 128, 768), (150, 384, 256)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_panel_gemm_epilogues(gpu_device, M, N, K, epi):
    dev = gpu_device
    assert _lib.lib().bd_gemm_panel_supported(N, K, 0) == 1
    g = torch.Generator().manual_seed(M + N + K + epi)
    A = (torch.randn(M, K, generator=g) * 0.7).bfloat16()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, generator=g) * 0.3
    acc = A.float() @ W.float().t() + bias
    if epi == 2:
        x0 = torch.randn(M, N, generator=g)
        out = x0.clone().to(dev)
        ref = x0 + acc
        panel(A.to(dev), W.to(dev), bias.to(dev), out, out, 2, False)
    else:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ref = acc if epi == 0 else torch.nn.functional.gelu(acc)
        panel(A.to(dev), W.to(dev), bias.to(dev), out, None, epi, False)
    tol = 2e-2 if epi != 2 else 1e-4
    assert (out.float().cpu() - ref).abs().max().item() / ref.abs().max().item() < tol


@pytest.mark.parametrize("M,N,K", [(300, 1152, 384), (129, 1536, 384), (70, 384, 128), (512, 256, 256)])
@pytest.mark.parametrize("epi", [0, 1])
def test_panel_gemm_fused_layernorm(gpu_device, M, N, K, epi):
    """LN(x) W^T + b with gamma / beta folded into the weights on the host."""
    dev = gpu_device
    assert _lib.lib().bd_gemm_panel_supported(N, K, 1) == 1
    g = torch.Generator().manual_seed(M + N + K + epi)
    x = torch.randn(M, K, generator=g) * 2 + 0.7
    gm = 1 + 0.2 * torch.randn(K, generator=g)
    bt = 0.3 * torch.randn(K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g) * 0.3
    ref = torch.nn.functional.layer_norm(x, (K,), gm, bt, 1e-5) @ W.t() + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    Wf = (W * gm[None, :]).bfloat16()
    bf = bias + W @ bt
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    panel(x.to(dev), Wf.to(dev), bf.to(dev), out, None, epi, True)
    assert (out.float().cpu() - ref).abs().max().item() / ref.abs().max().item() < 2e-2


def test_panel_gemm_unsupported_shapes_are_reported(gpu_device):
    lib = _lib.lib()
    assert lib.bd_gemm_panel_supported(256, 768, 0) == 0      # K in several chunks needs N = 128 or a multiple of 384
    assert lib.bd_gemm_panel_supported(768, 768, 0) == 1
    assert lib.bd_gemm_panel_supported(1152, 768, 1) == 0     # fused LayerNorm needs a single chunk
    assert lib.bd_gemm_panel_supported(100, 384, 0) == 0
    a = torch.zeros(256, 768, dtype=torch.bfloat16, device=gpu_device)
    assert lib.bd_gemm_panel(_lib.ptr(a), _lib.ptr(a), None, _lib.ptr(a), None, 8, 256, 768, 256, 0, 0, 1e-5, _s()) < 0


# ---------------------------------------------------------------- fused feed-forward block
def _ffn_case(gpu_device, M, d, F, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, d, generator=g) * 1.5 + 0.3
    gm = 1 + 0.2 * torch.randn(d, generator=g)
    bt = 0.3 * torch.randn(d, generator=g)
    W1 = torch.randn(F, d, generator=g) / math.sqrt(d)
    b1 = 0.3 * torch.randn(F, generator=g)
    W2 = torch.randn(d, F, generator=g) / math.sqrt(F)
    b2 = 0.3 * torch.randn(d, generator=g)
    lib = _lib.lib()
    w1f = (W1 * gm).bfloat16().to(gpu_device)
    w2b = W2.bfloat16().to(gpu_device)
    b1f = (b1 + W1 @ bt).to(gpu_device)
    b2d = b2.to(gpu_device)
    pack = torch.empty(lib.bd_ffn_pack_bytes(d, F), dtype=torch.uint8, device=gpu_device)
    _lib.check(lib.bd_ffn_pack_weights(_lib.ptr(w1f), _lib.ptr(w2b), _lib.ptr(pack), d, F, _s()), "pack")
    xd = x.to(gpu_device)
    out = torch.empty_like(xd)
    _lib.check(lib.bd_ffn_fused(_lib.ptr(xd), _lib.ptr(out), _lib.ptr(pack), _lib.ptr(b1f), _lib.ptr(b2d), M, d, F,
                                1e-5, _s()), "ffn")
    torch.cuda.synchronize()
    # fp32 reference of the module chain: x + fc2(gelu(fc1(LayerNorm(x))))
    ref = x + torch.nn.functional.gelu(torch.nn.functional.layer_norm(x, (d,), gm, bt, 1e-5) @ W1.t() + b1) @ W2.t() + b2
    return out.cpu(), ref, (xd, pack, b1f, b2d)


@pytest.mark.gpu
@pytest.mark.parametrize("M,d,F", [(128, 384, 1536), (300, 384, 1536), (1, 384, 1536), (257, 256, 1024),
                                   (1000, 128, 512), (64, 128, 96), (4096, 384, 1536)])
def test_ffn_fused_matches_fp32_reference(gpu_device, M, d, F):
    assert _lib.lib().bd_ffn_fused_supported(d, F) == 1
    out, ref, _ = _ffn_case(gpu_device, M, d, F, seed=M + d + F)
    # bf16 operands (LayerNorm output, weights, hidden activations), fp32 accumulation and residual
    err = (out - ref).abs().max().item()
    assert err < 2e-2 * ref.abs().max().item(), err
    assert torch.isfinite(out).all()


@pytest.mark.gpu
def test_ffn_fused_in_place_and_rows_independent(gpu_device):
    M, d, F = 300, 384, 1536
    out, _, (xd, pack, b1f, b2d) = _ffn_case(gpu_device, M, d, F, seed=5)
    lib = _lib.lib()
    x2 = xd.clone()
    _lib.check(lib.bd_ffn_fused(_lib.ptr(x2), _lib.ptr(x2), _lib.ptr(pack), _lib.ptr(b1f), _lib.ptr(b2d), M, d, F, 1e-5,
                                _s()), "ffn in place")
    torch.cuda.synchronize()
    assert torch.equal(x2.cpu(), out)                       # aliasing out == x gives the same bits
    sub = xd[37:38].clone()
    o1 = torch.empty_like(sub)
    _lib.check(lib.bd_ffn_fused(_lib.ptr(sub), _lib.ptr(o1), _lib.ptr(pack), _lib.ptr(b1f), _lib.ptr(b2d), 1, d, F, 1e-5,
                                _s()), "ffn one row")
    torch.cuda.synchronize()
    assert torch.equal(o1.cpu()[0], out[37])                # a row does not depend on its panel mates


@pytest.mark.gpu
def test_ffn_fused_unsupported_shapes_are_refused(gpu_device):
    lib = _lib.lib()
    assert lib.bd_ffn_fused_supported(512, 2048) == 0
    assert lib.bd_ffn_fused_supported(384, 1000) == 0
    assert lib.bd_ffn_fused_supported(384, 64) == 0
    a = torch.zeros(1024, device=gpu_device)
    assert lib.bd_ffn_fused(_lib.ptr(a), _lib.ptr(a), _lib.ptr(a), _lib.ptr(a), _lib.ptr(a), 2, 512, 2048, 1e-5, _s()) < 0


# ---------------------------------------------------------------- fused attention block
def _attn_block_case(gpu_device, B, L, d, H, masked, seed, keep=None, in_place=False):
    g = torch.Generator().manual_seed(seed)
    dh = d // H
    x = torch.randn(B * L, d, generator=g) * 1.5 + 0.3
    gm = 1 + 0.2 * torch.randn(d, generator=g)
    bt = 0.3 * torch.randn(d, generator=g)
    Wqkv = torch.randn(3 * d, d, generator=g) / math.sqrt(d)
    bqkv = 0.3 * torch.randn(3 * d, generator=g)
    Wo = torch.randn(d, d, generator=g) / math.sqrt(d)
    bo = 0.3 * torch.randn(d, generator=g)
    km = None
    if masked:
        km = torch.ones(B, L, dtype=torch.uint8)
        for b in range(B):
            km[b, int(torch.randint(1, L + 1, (1,), generator=g)):] = 0
        km[B - 1] = 0                                       # a fully masked sequence
    if keep is not None:                                    # the same draw, only these sequences submitted
        x = x.view(B, L, d)[keep].reshape(-1, d).contiguous()
        km = km[keep].contiguous() if km is not None else None
        B = len(keep)
    lib = _lib.lib()
    scale = math.log2(math.e) / math.sqrt(dh)
    wf = Wqkv * gm[None, :]
    bf = bqkv + Wqkv @ bt
    wf[:d] *= scale
    bf[:d] *= scale
    # heads the kernels have no tiling for run zero-padded: rows of q | k | v and columns of Wo, head by head
    aw = lib.bd_attn_block_width(d, H)
    dhp = aw // H
    Wo_p = Wo
    if dhp != dh:
        wp = torch.zeros(3, H, dhp, d); wp[:, :, :dh] = wf.view(3, H, dh, d); wf = wp.view(3 * aw, d)
        bp = torch.zeros(3, H, dhp); bp[:, :, :dh] = bf.view(3, H, dh); bf = bp.view(3 * aw)
        Wo_p = torch.zeros(d, H, dhp); Wo_p[:, :, :dh] = Wo.view(d, H, dh); Wo_p = Wo_p.view(d, aw)
    wqkv_b = wf.bfloat16().to(gpu_device)
    wo_b = Wo_p.bfloat16().to(gpu_device)
    assert lib.bd_attn_pack_bytes_heads(d, H) == 8 * aw * d and (aw != d or lib.bd_attn_pack_bytes(d) == 8 * d * d)
    pack = torch.empty(lib.bd_attn_pack_bytes_heads(d, H), dtype=torch.uint8, device=gpu_device)
    _lib.check(lib.bd_attn_pack_weights(_lib.ptr(wqkv_b), _lib.ptr(wo_b), _lib.ptr(pack), d, H, _s()), "attn pack")
    xd = x.to(gpu_device)
    out = xd if in_place else torch.empty_like(xd)
    bfd = bf[:aw].contiguous().to(gpu_device)
    bod = bo.to(gpu_device)
    bofd = (bo + Wo_p @ bf[2 * aw:]).to(gpu_device)
    kmd = km.to(gpu_device) if km is not None else None
    _lib.check(lib.bd_attn_block_fused(_lib.ptr(xd), _lib.ptr(out), _lib.ptr(pack), _lib.ptr(bfd), _lib.ptr(bod),
                                       _lib.ptr(bofd), _lib.ptr(kmd) if kmd is not None else None, B, L, d, H, 1e-5, _s()), "attn block")
    torch.cuda.synchronize()
    # fp32 reference of the module chain: x + out_proj(softmax(q k^T / sqrt(dh) + mask) v), inputs LayerNorm(x)
    xs = x.view(B, L, d)
    qkv = torch.nn.functional.layer_norm(xs, (d,), gm, bt, 1e-5) @ Wqkv.t() + bqkv
    q, k, v = [t.view(B, L, H, dh).transpose(1, 2) for t in qkv.split(d, dim=-1)]
    sc = q @ k.transpose(-1, -2) / math.sqrt(dh)
    if km is not None:
        sc = sc.masked_fill(km[:, None, None, :] == 0, float("-inf"))
    pr = torch.nan_to_num(torch.softmax(sc, dim=-1), nan=0.0)   # all keys masked -> zero output row
    ref = xs + (pr @ v).transpose(1, 2).reshape(B, L, d) @ Wo.t() + bo
    return out.cpu().view(B, L, d), ref, xs, bo


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,d,H,masked", [(3, 128, 384, 3, False), (4, 128, 384, 3, True), (5, 100, 384, 3, True),
                                           (2, 1, 384, 3, False), (2, 128, 256, 4, False), (3, 77, 256, 2, True),
                                           (3, 128, 128, 2, False), (3, 50, 128, 4, True), (300, 128, 384, 3, True),
                                           (4, 128, 384, 6, True), (3, 90, 384, 6, False),
                                           (4, 128, 384, 8, True), (3, 70, 384, 8, False)])   # 8 heads of 48, run as 8 x 64
def test_attn_block_fused_matches_fp32_reference(gpu_device, B, L, d, H, masked):
    assert _lib.lib().bd_attn_block_supported(d, H, L) == 1
    out, ref, xs, bo = _attn_block_case(gpu_device, B, L, d, H, masked, seed=B + L + d + H)
    assert torch.isfinite(out).all()
    # bf16 operands (LayerNorm output, weights, q, k, v, probabilities, attention output), fp32 accumulation
    err = (out - ref).abs().max().item()
    assert err < 2e-2 * ref.abs().max().item(), err
    if masked:   # the fully masked sequence: attention contributes exactly nothing
        assert torch.allclose(out[B - 1], xs[B - 1] + bo, atol=1e-6)


@pytest.mark.gpu
def test_attn_block_fused_sequences_independent_and_in_place(gpu_device):
    B, L, d, H = 6, 96, 384, 3
    out, _, xs, _ = _attn_block_case(gpu_device, B, L, d, H, True, seed=11)
    out2, _, _, _ = _attn_block_case(gpu_device, B, L, d, H, True, seed=11)
    assert torch.equal(out, out2)                            # deterministic
    # a sequence's rows depend on no other sequence of the launch: any subset, in any order, gives the same bits
    keep = [4, 1, 5]
    sub, _, _, _ = _attn_block_case(gpu_device, B, L, d, H, True, seed=11, keep=keep)
    assert torch.equal(sub, out[keep])
    # output written over the input
    inp, _, _, _ = _attn_block_case(gpu_device, B, L, d, H, True, seed=11, in_place=True)
    assert torch.equal(inp, out)


@pytest.mark.gpu
def test_attn_block_unsupported_shapes_are_refused(gpu_device):
    lib = _lib.lib()
    assert lib.bd_attn_block_supported(384, 3, 129) == 0
    assert lib.bd_attn_block_supported(384, 12, 128) == 0    # head_dim 32 at d = 384
    assert lib.bd_attn_block_supported(384, 4, 128) == 0     # head_dim 96
    assert lib.bd_attn_block_width(384, 8) == 512 and lib.bd_attn_block_width(384, 3) == 384   # 8 x 48 runs as 8 x 64
    assert lib.bd_attn_block_width(384, 4) == 0
    assert lib.bd_attn_block_supported(768, 6, 128) == 0
    a = torch.zeros(1024, device=gpu_device)
    assert lib.bd_attn_block_fused(_lib.ptr(a), _lib.ptr(a), _lib.ptr(a), _lib.ptr(a), _lib.ptr(a), _lib.ptr(a), None, 1, 200, 384, 3,
                                   1e-5, _s()) < 0


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(16411, 512, 768), (32768, 1536, 1024), (8200, 1024, 1024), (33000, 256, 832),
                                   (6700, 1152, 1152), (2600, 3456, 1152)])    # odd multiples of 128: half-empty last column tile
def test_large_tile_gemm_all_epilogues(gpu_device, M, N, K):
    """Shapes that take the 256 x 256-tile kernel (>= 128 tiles; N a multiple of 256, or of 128 from 1024 on): every epilogue against fp32 torch,
    ragged last row tile included; exact for small-integer operands."""
    import ctypes as C
    from oracle import sampler_ref as ora
    dev = gpu_device
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev).bfloat16()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    s = _lib.stream_handle()
    lib = _lib.lib()
    pre = A.float() @ W.float().t() + bias

    def run(epi, out, aux_in=None, aux_out=None, drop=None, b=bias):
        d = None if drop is None else C.byref(drop)
        _lib.check(lib.bd_gemm_bf16_drop(_lib.ptr(A), _lib.ptr(W), _lib.ptr(b), _lib.ptr(out), _lib.ptr(aux_in),
                                         _lib.ptr(aux_out), M, N, N, K, N, epi, d, 3, s), "gemm")

    def rel(a, b):
        return float((a.float() - b.float()).norm() / b.float().norm())

    o0 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    run(0, o0)
    assert torch.isfinite(o0.float()).all() and rel(o0, pre) < 4e-3
    u = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    o1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    run(1, o1, aux_out=u)
    assert rel(u, pre) < 4e-3 and rel(o1, torch.nn.functional.gelu(pre)) < 6e-3
    resid = torch.randn(M, N, generator=g).to(dev)
    o2 = torch.empty(M, N, device=dev)
    run(2, o2, aux_in=resid)
    assert rel(o2 - resid, pre) < 4e-3
    o2b = resid.clone()
    run(2, o2b)                       # in place on the residual stream
    assert torch.equal(o2b, o2)
    o3 = torch.empty(M, N, device=dev)
    run(3, o3)
    assert rel(o3, pre) < 1e-3 or rel(o3, pre) < 4e-3
    ur = u.float().requires_grad_(True)
    torch.nn.functional.gelu(ur).sum().backward()
    o4 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    run(4, o4, aux_in=u, b=None)
    assert rel(o4, (A.float() @ W.float().t()) * ur.grad) < 1e-2
    drop = _lib.Dropout(0.1, 5, 77)
    f = torch.from_numpy(ora.dropout_mask(77, 5, 3, 0.1, M, N)).to(dev)
    o1d = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    run(1, o1d, drop=drop)
    assert bool((o1d[f == 0] == 0).all()) and rel(o1d, torch.nn.functional.gelu(pre) * f) < 6e-3
    # exact arithmetic on small integers (any mis-addressed fragment shows up as a wrong integer)
    Ai = torch.randint(-3, 4, (M, K), generator=g).float()
    Wi = torch.randint(-3, 4, (N, K), generator=g).float()
    A, W = Ai.to(dev).bfloat16(), Wi.to(dev).bfloat16()
    oi = torch.empty(M, N, device=dev)
    run(3, oi, b=None)
    assert torch.equal(oi.cpu(), Ai @ Wi.t())


# ---------------------------------------------------------------- LayerNorm deferred across a residual projection
@pytest.mark.parametrize("M,d,Kp,Nc,row_mean", [(1000, 768, 3072, 2304, 0.0), (777, 1152, 1152, 4608, 0.0),
                                                 (512, 1024, 4096, 3072, 2.0)])
def test_deferred_layernorm_projections(gpu_device, M, d, Kp, Nc, row_mean):
    """bd_gemm_bf16_resid_stats + bd_gemm_bf16_deferred_ln against the plain chain (epilogue-2 GEMM, LayerNorm, GEMM):
    the updated residual rows bit for bit, their bf16 copy = one rounding of them, the row statistics against torch, and
    the next projection's output against fp32 torch at the bound of the plain bf16 chain (looser when the rows carry a
    mean of row_mean standard deviations: the rows are rounded before the mean is taken out)."""
    dev = gpu_device
    lib = _lib.lib()
    g = torch.Generator().manual_seed(M + d)
    A = (torch.randn(M, Kp, generator=g) * 0.5).bfloat16().to(dev)
    Wp = (torch.randn(d, Kp, generator=g) / math.sqrt(Kp)).bfloat16().to(dev)
    bp = (torch.randn(d, generator=g) * 0.2).to(dev)
    x0 = (torch.randn(M, d, generator=g) * 1.5 + row_mean * 1.5).to(dev)
    gm = (1 + 0.2 * torch.randn(d, generator=g)).to(dev)
    bt = (0.3 * torch.randn(d, generator=g)).to(dev)
    Wc = (torch.randn(Nc, d, generator=g) / math.sqrt(d)).to(dev)
    bc = (torch.randn(Nc, generator=g) * 0.3).to(dev)
    # plain chain
    x_plain = x0.clone()
    gemm(A, Wp, bp, x_plain, d, 2)
    # deferred chain
    x = x0.clone()
    xb = torch.full((M, d), float("nan"), dtype=torch.bfloat16, device=dev)
    part = torch.empty(lib.bd_gemm_resid_stats_part_bytes(M, d) // 4, dtype=torch.float32, device=dev)
    stats = torch.empty(M, 2, dtype=torch.float32, device=dev)
    _lib.check(lib.bd_gemm_bf16_resid_stats(_lib.ptr(A), _lib.ptr(Wp), _lib.ptr(bp), _lib.ptr(x), _lib.ptr(xb), _lib.ptr(part),
                                            _lib.ptr(stats), M, d, Kp, 1e-5, _s()), "resid_stats")
    torch.cuda.synchronize()
    assert torch.equal(x, x_plain)
    assert torch.equal(xb, x.bfloat16())
    mean = x.double().mean(1)
    rstd = 1.0 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-5)
    assert float((stats[:, 0].double() - mean).abs().max()) < 1e-5 * (1 + abs(row_mean) * 1.5)
    assert float(((stats[:, 1].double() - rstd) / rstd).abs().max()) < 1e-4
    Wln = (Wc * gm[None, :]).bfloat16()
    bln = (bc + Wc @ bt).contiguous()
    cs = Wln.float().sum(1).contiguous()
    for epi in (0, 1):
        out = torch.full((M, Nc), float("nan"), dtype=torch.bfloat16, device=dev)
        _lib.check(lib.bd_gemm_bf16_deferred_ln(_lib.ptr(xb), _lib.ptr(Wln), _lib.ptr(bln), _lib.ptr(cs), _lib.ptr(stats),
                                                _lib.ptr(out), M, Nc, d, epi, _s()), "deferred_ln")
        ref = torch.nn.functional.layer_norm(x, (d,), gm, bt, 1e-5) @ Wc.t() + bc
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        # the plain chain on the same inputs, for the scale of the bf16 error
        h = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
        _lib.check(lib.bd_layernorm(_lib.ptr(x), _lib.ptr(gm), _lib.ptr(bt), _lib.ptr(h), None, None, M, d, 1e-5, _s()), "ln")
        plain = torch.empty(M, Nc, dtype=torch.bfloat16, device=dev)
        gemm(h, Wc.bfloat16().contiguous(), bc, plain, Nc, epi)
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        e_def = float((out.float() - ref).norm() / ref.norm())
        e_plain = float((plain.float() - ref).norm() / ref.norm())
        assert e_plain < 6e-3
        assert e_def < (6e-3 if row_mean == 0.0 else 2e-2), (e_def, e_plain)


def test_deferred_layernorm_arguments(gpu_device):
    lib = _lib.lib()
    a = torch.zeros(256, 256, dtype=torch.bfloat16, device=gpu_device)
    f = torch.zeros(256, 256, dtype=torch.float32, device=gpu_device)
    assert lib.bd_gemm_bf16_resid_stats(_lib.ptr(a), _lib.ptr(a), None, _lib.ptr(f), _lib.ptr(a), _lib.ptr(f), _lib.ptr(f),
                                        256, 200, 256, 1e-5, _s()) < 0
    assert lib.bd_gemm_bf16_deferred_ln(_lib.ptr(a), _lib.ptr(a), None, _lib.ptr(f), _lib.ptr(f), _lib.ptr(a), 256, 256, 96, 0,
                                        _s()) < 0
    assert lib.bd_gemm_bf16_deferred_ln(_lib.ptr(a), _lib.ptr(a), None, _lib.ptr(f), _lib.ptr(f), _lib.ptr(a), 256, 256, 256, 2,
                                        _s()) < 0
    assert lib.bd_gemm_deferred_ln_supported(100000, 768, 768) == 1
    if os.environ.get("BD_GEMM8P", "1") != "2":        # (2 forces the eight-wave kernel on every shape)
        assert lib.bd_gemm_deferred_ln_supported(1000, 768, 768) == 0      # too few tiles for the large-tile kernel to pay
    assert lib.bd_gemm_deferred_ln_supported(100000, 768, 96) == 0      # K below 128
