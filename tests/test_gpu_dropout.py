"""GPU parity of training dropout. The kernels draw their keep decisions from a counter-based
stream (Philox block per (site, row, 8 columns)); the CPU oracle evaluates the same stream, so
both sides drop the SAME elements and the comparison is as tight as without dropout. The keep
pattern itself is checked bit-exactly; values to the bf16 / fp32 bars of the dropout-free tests."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from tests.golden import replay_inputs as RI
from oracle import backbone_ref as BR
from oracle import sampler_ref as ora
from bidiff import _lib
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
from bidiff.training import autograd as AG

pytestmark = pytest.mark.gpu


def _s():
    return _lib.stream_handle()


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def factors(drop, site, rows, cols):
    return torch.from_numpy(ora.dropout_mask(drop.seed, drop.step, site, drop.p, rows, cols))


@pytest.mark.parametrize("M,N,p", [(77, 128, 0.1), (1000, 384, 0.1), (5, 4, 0.5), (300, 1536, 0.25)])
def test_dropout_kernels_follow_the_decision_stream_exactly(gpu_device, M, N, p):
    dev = gpu_device
    drop = _lib.Dropout(p, 3, 0xDEADBEEFCAFE1234)
    x = torch.randn(M, N, generator=torch.Generator().manual_seed(M + N))
    f = factors(drop, 9, M, N)
    assert abs(float((f == 0).float().mean()) - p) < 4 * math.sqrt(p * (1 - p) / (M * N)) + 1e-9 or M * N < 100
    xd = x.to(dev)
    out = torch.empty_like(xd)
    _lib.check(_lib.lib().bd_dropout_f32(_lib.ptr(xd), _lib.ptr(out), M, N, C.byref(drop), 9, _s()), "dropout")
    assert torch.equal(out.cpu(), x * f), "fp32 dropout must equal x * factor bit for bit"
    ob = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.lib().bd_dropout_f32_to_bf16(_lib.ptr(xd), _lib.ptr(ob), M, N, C.byref(drop), 9, _s()), "dropout")
    assert torch.equal(ob.cpu(), (x * f).bfloat16())
    # in place, and another step / site / seed gives another pattern
    _lib.check(_lib.lib().bd_dropout_f32(_lib.ptr(xd), _lib.ptr(xd), M, N, C.byref(drop), 9, _s()), "dropout")
    assert torch.equal(xd.cpu(), x * f)
    if M * N >= 1000:
        for other in (_lib.Dropout(p, 4, drop.seed), _lib.Dropout(p, 3, drop.seed + 1)):
            assert not torch.equal(factors(other, 9, M, N), f)
        assert not torch.equal(factors(drop, 10, M, N), f)


def test_dropout_off_is_the_identity_and_bad_p_is_rejected(gpu_device):
    dev = gpu_device
    x = torch.randn(64, 128, device=dev)
    out = torch.empty_like(x)
    for drop in (None, C.byref(_lib.Dropout(0.0, 1, 2))):
        _lib.check(_lib.lib().bd_dropout_f32(_lib.ptr(x), _lib.ptr(out), 64, 128, drop, 0, _s()), "dropout")
        assert torch.equal(out, x)
    for p in (1.0, -0.1, 1.5):
        rc = _lib.lib().bd_dropout_f32(_lib.ptr(x), _lib.ptr(out), 64, 128, C.byref(_lib.Dropout(p, 1, 2)), 0, _s())
        assert rc != 0 and b"outside" in _lib.lib().bd_last_error()
    rc = _lib.lib().bd_dropout_f32(_lib.ptr(x), _lib.ptr(out), 64, 126, C.byref(_lib.Dropout(0.1, 1, 2)), 0, _s())
    assert rc != 0


def test_embedding_dropout(gpu_device):
    dev = gpu_device
    B, L, d, V, T = 3, 20, 128, 39, 10
    g = torch.Generator().manual_seed(4)
    tok, pos, tim = torch.randn(V, d, generator=g), torch.randn(64, d, generator=g), torch.randn(T + 1, d, generator=g)
    ids = torch.randint(0, V, (B, L), generator=g)
    t = torch.randint(1, T + 1, (B,), generator=g)
    drop = _lib.Dropout(0.1, 7, 99)
    x = torch.empty(B * L, d, device=dev)
    args = [a.to(dev) for a in (ids, t, tok, pos, tim)]
    _lib.check(_lib.lib().bd_embed_drop(*[_lib.ptr(a) for a in args], _lib.ptr(x), B, L, d, V, 64, T + 1, C.byref(drop), 0, _s()), "embed")
    want = ((tok[ids] + pos[:L][None]) + tim[t][:, None, :]).view(B * L, d) * factors(drop, 0, B * L, d)
    assert torch.equal(x.cpu(), want)


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (1000, 1536, 384)])
def test_gemm_epilogues_with_dropout(gpu_device, M, N, K):
    dev = gpu_device
    g = torch.Generator().manual_seed(2)
    A = torch.randn(M, K, generator=g).to(dev).bfloat16()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    drop = _lib.Dropout(0.1, 2, 12345)
    f = factors(drop, 6, M, N).to(dev)
    pre = A.float() @ W.float().t() + bias
    # epilogue 1: drop(gelu(.)), the saved pre-activation is not dropped
    u = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    a = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    AG._gemm(A, W, bias, a, M, N, N, K, N, 1, aux_out=u, drop=drop, site=6)
    assert rel(u.float(), pre) < 1e-2
    assert bool((a[f == 0] == 0).all())
    assert rel(a.float(), torch.nn.functional.gelu(pre) * f) < 1e-2
    # epilogue 2: resid + drop(acc + bias)
    resid = torch.randn(M, N, generator=g).to(dev)
    out = torch.empty(M, N, device=dev)
    AG._gemm(A, W, bias, out, M, N, N, K, N, 2, aux_in=resid, drop=drop, site=6)
    assert bool((out[f == 0] == resid[f == 0]).all()), "dropped elements pass the residual through untouched"
    assert rel(out - resid, pre * f) < 1e-2
    # epilogue 4: (acc + bias) * factor * gelu'(u)
    o4 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    AG._gemm(A, W, None, o4, M, N, N, K, N, 4, aux_in=u, drop=drop, site=6)
    ur = u.float().requires_grad_(True)
    torch.nn.functional.gelu(ur).sum().backward()
    assert bool((o4[f == 0] == 0).all())
    assert rel(o4.float(), (A.float() @ W.float().t()) * f * ur.grad) < 1e-2
    # epilogue 6: drop(gelu(.)) and the backward multiplier gelu'(.) * factor; epilogue 7 multiplies by it
    for dr, ff in ((drop, f), (None, torch.ones_like(f))):
        a6 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        t6 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        AG._gemm(A, W, bias, a6, M, N, N, K, N, 6, aux_out=t6, drop=dr, site=6)
        pr = pre.clone().requires_grad_(True)
        torch.nn.functional.gelu(pr).sum().backward()
        assert rel(a6.float(), torch.nn.functional.gelu(pre) * ff) < 1e-2
        assert bool((t6[ff == 0] == 0).all()) and bool((a6[ff == 0] == 0).all())
        assert rel(t6.float(), pr.grad * ff) < 1e-2
        o7 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        AG._gemm(A, W, None, o7, M, N, N, K, N, 7, aux_in=t6)
        assert rel(o7.float(), (A.float() @ W.float().t()) * t6.float()) < 1e-2
    rc = _lib.lib().bd_gemm_bf16_drop(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(a), None, None, M, N, N, K, N, 6,
                                      None, 0, _s())
    assert rc != 0, "epilogue 6 without aux_out"
    # an epilogue without a dropout form refuses
    rc = _lib.lib().bd_gemm_bf16_drop(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(a), None, None, M, N, N, K, N, 0,
                                      C.byref(drop), 6, _s())
    assert rc != 0


def ref_attention_drop(qkv, mask, B, L, Hh, dh, f):
    x = qkv.float().view(B, L, 3, Hh, dh)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    s = s.masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
    p = torch.nan_to_num(torch.softmax(s, dim=-1), nan=0.0) * f.view(B, Hh, L, L)
    return (p @ v).transpose(1, 2).reshape(B * L, Hh * dh)


@pytest.mark.parametrize("dh,Hh,L", [(32, 2, 24), (64, 2, 128), (128, 1, 128), (64, 3, 150), (128, 3, 300)])
def test_attention_forward_backward_with_dropout(gpu_device, dh, Hh, L):
    dev = gpu_device
    B = 4
    d = Hh * dh
    g = torch.Generator().manual_seed(dh + L)
    qkv = torch.randn(B * L, 3 * d, generator=g).bfloat16()
    dout = torch.randn(B * L, d, generator=g).bfloat16()
    lengths = [L, max(2, L // 3), 0, L - 3]
    mask = torch.zeros(B, L, dtype=torch.uint8)
    for b, n in enumerate(lengths):
        mask[b, :n] = 1
    drop = _lib.Dropout(0.1, 11, 0x1234567890ABCDEF)
    f = factors(drop, 5, B * Hh * L, L)
    dq, dm, ddo = qkv.to(dev), mask.to(dev), dout.to(dev)
    out = torch.empty(B * L, d, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, Hh, L, device=dev)
    _lib.check(_lib.lib().bd_attention_fwd_drop(_lib.ptr(dq), _lib.ptr(dm), _lib.ptr(out), _lib.ptr(lse), B, L, Hh, dh,
                                                C.byref(drop), 5, _s()), "fwd")
    dqkv = torch.full((B * L, 3 * d), float("nan"), dtype=torch.bfloat16, device=dev)
    delta = torch.empty(B, Hh, L, device=dev)
    _lib.check(_lib.lib().bd_attention_bwd_drop(_lib.ptr(dq), _lib.ptr(dm), _lib.ptr(out), _lib.ptr(ddo), _lib.ptr(lse),
                                                _lib.ptr(delta), _lib.ptr(dqkv), B, L, Hh, dh, C.byref(drop), 5, _s()),
               "bwd")
    torch.cuda.synchronize()
    qr = qkv.float().requires_grad_(True)
    ref = ref_attention_drop(qr, mask, B, L, Hh, dh, f)
    assert rel(out.float().cpu(), ref.detach()) < 2e-2
    # the log-sum-exp is the one of the undropped softmax
    from tests.test_gpu_kernels import ref_attention
    nodrop = torch.empty_like(out)
    lse0 = torch.empty_like(lse)
    _lib.check(_lib.lib().bd_attention_fwd(_lib.ptr(dq), _lib.ptr(dm), _lib.ptr(nodrop), _lib.ptr(lse0), B, L, Hh, dh, _s()),
               "fwd")
    assert torch.equal(lse, lse0)
    assert rel(nodrop.float().cpu(), ref_attention(qkv, mask, B, L, Hh, dh)) < 2e-2
    ref.backward(dout.float())
    got = dqkv.float().cpu()
    assert torch.isfinite(got).all()
    for name, sl in (("dQ", slice(0, d)), ("dK", slice(d, 2 * d)), ("dV", slice(2 * d, 3 * d))):
        assert rel(got[:, sl], qr.grad[:, sl]) < 2e-2, name
    assert got.view(B, L, 3 * d)[2].abs().max().item() == 0.0, "fully padded sequence: zero gradients"


def _setup(cfg, dev, B=6, L=24):
    torch.manual_seed(7)
    bb = DiffusionBackbone(**cfg)
    RI.perturb_state_dict(bb.state_dict(), 11)
    model = DiscreteMaskingDiffusion(bb, num_steps=cfg["num_diffusion_steps"], mask_token_id=1, pad_token_id=0,
                                     bos_token_id=2, eos_token_id=3)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ids, attn, t = RI.backbone_inputs(cfg, 33, B=B, L=L)
    return model.to(dev), sd, torch.from_numpy(ids), torch.from_numpy(attn), torch.from_numpy(t)


@pytest.mark.parametrize("name,heads", [("tiny_h2", None), ("tiny_h1", None), ("small", 8)])
def test_training_step_with_dropout_matches_oracle_autograd(gpu_device, name, heads):
    """Loss and every parameter gradient of a training forward/backward WITH dropout 0.1 at all four
    sites against fp32 autograd of the CPU restatement fed the same decision stream. ("small", 8) is
    the head size 48 zero-padded to 64 on the device.)"""
    dev = gpu_device
    if name == "small":
        cfg = dict(RI.SMALL_CFG, num_heads=heads, num_layers=2, dropout=0.1)
    else:
        cfg = dict(RI.BACKBONE_CFGS[name], dropout=0.1)
    model, sd, ids, attn, t = _setup(cfg, dev)
    model.train()
    bb = model.backbone
    bb.dropout_seed = 0xABCDEF0123456789
    bb._drop_step = 5
    u = torch.rand(ids.shape, generator=torch.Generator().manual_seed(1))
    noisy, ind = model.forward_process(ids.to(dev), t.to(dev), attn.to(dev), uniforms=u.to(dev))
    logits = bb(noisy, t.to(dev), attn.to(dev))
    assert bb._drop_step == 6
    loss = model._compute_loss(logits, ids.to(dev), ind, attn.to(dev))
    loss.backward()
    stream = BR.DropoutStream(0.1, bb.dropout_seed, 5)
    ref_sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and k != "mask_schedule") for k, v in sd.items()}
    rl = BR.logits(ref_sd, cfg, noisy.cpu(), t, attn, prefix="backbone.", dropout=stream)
    assert rel(logits.detach().cpu(), rl.detach()) < 2e-2
    # the dropout really is on: the dropout-free logits differ visibly
    rl0 = BR.logits(ref_sd, cfg, noisy.cpu(), t, attn, prefix="backbone.")
    assert rel(rl0.detach(), rl.detach()) > 5e-2
    rloss = BR.masked_ce(rl, ids, ind.cpu(), attn)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) < 2e-2 * abs(rloss.item())
    for k, p in model.named_parameters():
        gref = ref_sd[k].grad
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        if k.endswith("k_proj.bias"):
            # with dropout the key-bias gradient is still zero in exact arithmetic (the softmax is
            # shift invariant and dropout acts after it)
            qb = ref_sd[k.replace("k_proj", "q_proj")].grad.norm().item()
            assert p.grad.norm().item() < 0.1 * qb, k
            continue
        assert rel(p.grad.cpu(), gref) < 1e-1, (k, rel(p.grad.cpu(), gref))
        cos = torch.nn.functional.cosine_similarity(p.grad.cpu().flatten(), gref.flatten(), dim=0).item()
        assert cos > 0.995, (k, cos)


def test_dropout_stream_advances_per_forward_and_eval_mode_is_clean(gpu_device):
    dev = gpu_device
    cfg = dict(RI.BACKBONE_CFGS["tiny_h2"], dropout=0.1)
    model, sd, ids, attn, t = _setup(cfg, dev)
    bb = model.backbone
    args = (ids.to(dev), t.to(dev), attn.to(dev))
    model.train()
    torch.manual_seed(123)
    a = bb(*args).detach().clone()
    seed = bb.dropout_seed
    assert seed is not None and bb._drop_step == 1
    b = bb(*args).detach().clone()
    assert not torch.equal(a, b), "a new forward pass draws new decisions"
    bb._drop_step = 0
    assert torch.equal(bb(*args).detach(), a), "same (seed, step) -> same decisions"
    bb.dropout_seed, bb._drop_step = None, 0
    torch.manual_seed(123)
    assert torch.equal(bb(*args).detach(), a) and bb.dropout_seed == seed, "torch.manual_seed fixes the stream"
    model.eval()
    with torch.no_grad():
        e1 = bb(*args)
    ref = BR.logits({k: v for k, v in sd.items()}, cfg, ids, t, attn, prefix="backbone.")
    assert rel(e1.cpu(), ref) < 2e-2
    # p = 0 in training mode: the dropout-free network
    bb.dropout = 0.0
    model.train()
    z = bb(*args).detach()
    assert rel(z.cpu(), ref) < 2e-2
