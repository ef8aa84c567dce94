"""GPU parity of the training-step kernels and of the whole training step (forward, backward,
clip + AdamW) against fp32 PyTorch autograd of the CPU oracle and the reference golden vectors."""
import math
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import replay_inputs as RI
from oracle import backbone_ref as BR
from bidiff import _lib
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
from bidiff.training import autograd as AG
from bidiff.training.optimizer import FlatAdamW

pytestmark = pytest.mark.gpu
BLOB = np.load(os.path.join(H.GOLDEN, "model.npz"))


def _s():
    return _lib.stream_handle()


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def test_masked_ce_matches_reference_golden(gpu_device):
    dev = gpu_device
    lg = torch.from_numpy(BLOB["ce/logits"]).to(dev).requires_grad_(True)
    tg, mi, am = (torch.from_numpy(BLOB[k]).to(dev) for k in ("ce/targets", "ce/ind", "ce/attn"))
    loss = AG.masked_cross_entropy(lg, tg, mi, am)
    loss.backward()
    assert abs(loss.item() - float(BLOB["ce/loss"])) < 1e-5
    assert np.abs(lg.grad.cpu().numpy() - BLOB["ce/dlogits"]).max() < 1e-6
    lg2 = lg.detach().clone().requires_grad_(True)
    zero = AG.masked_cross_entropy(lg2, tg, torch.zeros_like(mi), am)
    zero.backward()
    assert zero.item() == 0.0 and float(lg2.grad.abs().max()) == 0.0
    assert abs(AG.masked_cross_entropy(lg.detach(), tg, mi, None).item() - float(BLOB["ce/loss_noattn"])) < 1e-5
    # upstream gradient scaling (loss / grad_accum)
    lg3 = lg.detach().clone().requires_grad_(True)
    (AG.masked_cross_entropy(lg3, tg, mi, am) / 4).backward()
    assert np.abs(lg3.grad.cpu().numpy() - BLOB["ce/dlogits"] / 4).max() < 1e-6


@pytest.mark.parametrize("V", [39, 803])
def test_masked_ce_large(gpu_device, V):
    dev = gpu_device
    g = torch.Generator().manual_seed(V)
    M = 4000
    lg = (torch.randn(M, V, generator=g) * 3).to(dev).requires_grad_(True)
    tg = torch.randint(0, V, (M,), generator=g).to(dev)
    mi = (torch.rand(M, generator=g) < 0.4).to(dev)
    am = (torch.rand(M, generator=g) < 0.8).long().to(dev)
    loss = AG.masked_cross_entropy(lg.view(40, 100, V), tg.view(40, 100), mi.view(40, 100), am.view(40, 100))
    loss.backward()
    ref_lg = lg.detach().clone().requires_grad_(True)
    ref = BR.masked_ce(ref_lg.view(40, 100, V), tg.view(40, 100), mi.view(40, 100), am.view(40, 100))
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1, abs(ref.item()))
    assert (lg.grad - ref_lg.grad).abs().max().item() < 1e-7 + 1e-4 * ref_lg.grad.abs().max().item()


@pytest.mark.parametrize("M,d", [(50, 128), (1000, 384), (300, 1024)])
def test_layernorm_backward(gpu_device, M, d):
    dev = gpu_device
    g = torch.Generator().manual_seed(M + d)
    x = (torch.randn(M, d, generator=g) * 2 + 0.5).to(dev)
    gm = (1 + 0.2 * torch.randn(d, generator=g)).to(dev)
    bt = torch.randn(d, generator=g).to(dev)
    dy = torch.randn(M, d, generator=g).to(dev).bfloat16()
    dres = torch.randn(M, d, generator=g).to(dev)
    st = torch.empty(M, 2, device=dev)
    hb = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.lib().bd_layernorm(_lib.ptr(x), _lib.ptr(gm), _lib.ptr(bt), _lib.ptr(hb), None, _lib.ptr(st), M, d, 1e-5, _s()), "ln")
    dx = dres.clone()
    dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    _lib.check(_lib.lib().bd_layernorm_bwd(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(st), _lib.ptr(gm), _lib.ptr(dx), _lib.ptr(dx),
                                           _lib.ptr(dg), _lib.ptr(db), M, d, _s()), "ln bwd")
    torch.cuda.synchronize()
    xr, gr, br = x.clone().requires_grad_(True), gm.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-5).backward(dy.float())
    assert rel(dx, xr.grad + dres) < 1e-5
    assert rel(dg, gr.grad) < 1e-4 and rel(db, br.grad) < 1e-4


@pytest.mark.parametrize("M,N,K", [(512, 128, 128), (5000, 384, 1536), (3333, 1152, 384), (70, 896, 384)])
def test_wgrad_and_colsum(gpu_device, M, N, K):
    dev = gpu_device
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g).to(dev).bfloat16()
    X = torch.randn(M, K, generator=g).to(dev).bfloat16()
    dW = torch.ones(N, K, device=dev)          # accumulates on top of existing content
    bs = torch.ones(N, device=dev)
    _lib.check(_lib.lib().bd_gemm_wgrad_bf16(_lib.ptr(dY), N, _lib.ptr(X), K, _lib.ptr(dW), M, N, K, _s()), "wgrad")
    _lib.check(_lib.lib().bd_colsum_bf16(_lib.ptr(dY), N, _lib.ptr(bs), M, N, _s()), "colsum")
    torch.cuda.synchronize()
    ref = dY.float().t() @ X.float() + 1
    assert rel(dW, ref) < 1e-5
    assert rel(bs, dY.float().sum(0) + 1) < 1e-5


@pytest.mark.parametrize("M,seg,nseg,K,extra", [(4096, 384, 3, 384, 0), (32768, 384, 3, 384, 0), (32768, 1536, 1, 384, 0),
                                                 (640, 128, 2, 1536, 64), (96, 256, 3, 128, 0), (70, 128, 3, 256, 8),
                                                 (32, 128, 1, 128, 0),
                                                 # large outputs over few tokens (the wide models): one workgroup per tile
                                                 (2560, 1536, 3, 1536, 0), (1024, 6144, 1, 1536, 0), (608, 1536, 1, 6144, 128)])
def test_wgrad_segments_with_fused_bias_sums(gpu_device, M, seg, nseg, K, extra):
    """One launch for up to three row segments with separate destinations plus the bias gradients;
    token counts that are multiples of 32 take the LDS-DMA kernel, others the register-staged one;
    `extra` = unused columns to the right of the stacked gradient (leading dimension > N)."""
    import ctypes as C
    dev = gpu_device
    N = seg * nseg
    g = torch.Generator().manual_seed(M + N + K)
    dYf = torch.randn(M, N + extra, generator=g).to(dev).bfloat16()
    X = torch.randn(M, K, generator=g).to(dev).bfloat16()
    dWs = [torch.full((seg, K), float(i), device=dev) for i in range(nseg)]     # accumulate on top
    dbs = [torch.full((seg,), float(-i), device=dev) for i in range(nseg)]
    pw = (C.c_void_p * nseg)(*[t.data_ptr() for t in dWs])
    pb = (C.c_void_p * nseg)(*[t.data_ptr() for t in dbs])
    if nseg == 2:           # a missing bias destination is allowed per segment
        pb[1] = None
    _lib.check(_lib.lib().bd_gemm_wgrad_segments(_lib.ptr(dYf), N + extra, _lib.ptr(X), K, pw, pb, nseg, seg, M, K, _s()),
               "wgrad")
    torch.cuda.synchronize()
    ref = dYf[:, :N].float().t() @ X.float()
    cs = dYf[:, :N].float().sum(0)
    for i in range(nseg):
        assert rel(dWs[i], ref[i * seg:(i + 1) * seg] + i) < 1e-5, i
        if nseg == 2 and i == 1:
            assert torch.equal(dbs[i], torch.full((seg,), float(-i), device=dev))
        else:
            assert rel(dbs[i], cs[i * seg:(i + 1) * seg] - i) < 1e-5, i
    # without any bias destination
    dW2 = [torch.zeros(seg, K, device=dev) for _ in range(nseg)]
    pw2 = (C.c_void_p * nseg)(*[t.data_ptr() for t in dW2])
    _lib.check(_lib.lib().bd_gemm_wgrad_segments(_lib.ptr(dYf), N + extra, _lib.ptr(X), K, pw2, None, nseg, seg, M, K, _s()),
               "wgrad")
    for i in range(nseg):
        assert rel(dW2[i], ref[i * seg:(i + 1) * seg]) < 1e-5


@pytest.mark.parametrize("M,N,K", [(200, 256, 128), (256, 256, 128), (1600, 2048, 1536)])
def test_wgrad_exact_integers(gpu_device, M, N, K):
    """Small integers: every fp32 sum is exact, whatever the order of the atomics (the last shape takes the launch
    with one workgroup per output tile)."""
    dev = gpu_device
    g = torch.Generator().manual_seed(0)
    dY = torch.randint(-3, 4, (M, N), generator=g).float()
    X = torch.randint(-3, 4, (M, K), generator=g).float()
    dW = torch.zeros(N, K, device=dev)
    a, b = dY.to(dev).bfloat16(), X.to(dev).bfloat16()
    _lib.check(_lib.lib().bd_gemm_wgrad_bf16(_lib.ptr(a), N, _lib.ptr(b), K, _lib.ptr(dW), M, N, K, _s()), "wgrad")
    torch.cuda.synchronize()
    assert torch.equal(dW.cpu(), dY.t() @ X)


@pytest.mark.parametrize("single_owner", [0, 1])
def test_wgrad_group_of_a_layers_products(gpu_device, single_owner):
    """bd_gemm_wgrad_group on a layer's four products (the stacked q | k | v one in three segments with bias sums),
    split launches (0) and the single-owner launch (1): exact on small integers, accumulating into the destinations;
    the single-owner launch gives the same bits on random data twice in a row."""
    dev = gpu_device
    M, d, F = 1184, 256, 512
    g = torch.Generator().manual_seed(3)
    shapes = [(3 * d, d, 3), (d, d, 1), (F, d, 1), (d, F, 1)]          # (N, K, segments)

    def run(make):
        jobs = (_lib.WgradJob * len(shapes))()
        keep, refs = [], []
        for j, (N, K, nseg) in zip(jobs, shapes):
            dY, X = make(M, N), make(M, K)
            a, b = dY.to(dev).bfloat16(), X.to(dev).bfloat16()
            seg = N // nseg
            dW = [torch.ones(seg, K, device=dev) for _ in range(nseg)]
            db = [torch.zeros(seg, device=dev) for _ in range(nseg)]
            keep += [a, b] + dW + db
            j.dY, j.X = a.data_ptr(), b.data_ptr()
            for k in range(nseg):
                j.dW[k], j.db[k] = dW[k].data_ptr(), db[k].data_ptr()
            j.ldy, j.ldx, j.n_seg, j.seg_rows, j.K = N, K, nseg, seg, K
            refs.append((a.float().cpu(), b.float().cpu(), dW, db, seg))
        _lib.check(_lib.lib().bd_gemm_wgrad_group(jobs, len(shapes), M, single_owner, _s()), "wgrad group")
        torch.cuda.synchronize()
        return refs

    for a, b, dW, db, seg in run(lambda m, n: torch.randint(-3, 4, (m, n), generator=g).float()):
        want = a.t() @ b + 1.0
        for k, (w, bsum) in enumerate(zip(dW, db)):
            assert torch.equal(w.cpu(), want[k * seg:(k + 1) * seg])
            assert torch.equal(bsum.cpu(), a[:, k * seg:(k + 1) * seg].sum(0))
    if single_owner:
        g.manual_seed(9)
        first = [[w.clone() for w in dW] for _, _, dW, _, _ in run(lambda m, n: torch.randn(m, n, generator=g))]
        g.manual_seed(9)
        again = [[w.clone() for w in dW] for _, _, dW, _, _ in run(lambda m, n: torch.randn(m, n, generator=g))]
        for x, y in zip(first, again):
            for u, v in zip(x, y):
                assert torch.equal(u, v)


def test_dgelu_epilogue_and_preactivation_save(gpu_device):
    dev = gpu_device
    M, N, K = 300, 256, 128
    g = torch.Generator().manual_seed(2)
    A = torch.randn(M, K, generator=g).to(dev).bfloat16()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    u = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    a = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    AG._gemm(A, W, bias, a, M, N, N, K, N, 1, aux_out=u)
    pre = A.float() @ W.float().t() + bias
    assert rel(u.float(), pre) < 1e-2 and rel(a.float(), torch.nn.functional.gelu(pre)) < 1e-2
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    AG._gemm(A, W, None, out, M, N, N, K, N, 4, aux_in=u)
    ur = u.float().requires_grad_(True)
    torch.nn.functional.gelu(ur).sum().backward()
    assert rel(out.float(), (A.float() @ W.float().t()) * ur.grad) < 1e-2


@pytest.mark.parametrize("dh,Hh,L", [(32, 2, 24), (64, 2, 128), (128, 1, 128), (64, 3, 150), (128, 3, 300)])
def test_attention_backward(gpu_device, dh, Hh, L):
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
    dq, dm, ddo = qkv.to(dev), mask.to(dev), dout.to(dev)
    out = torch.empty(B * L, d, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, Hh, L, device=dev)
    _lib.check(_lib.lib().bd_attention_fwd(_lib.ptr(dq), _lib.ptr(dm), _lib.ptr(out), _lib.ptr(lse), B, L, Hh, dh, _s()), "fwd")
    dqkv = torch.full((B * L, 3 * d), float("nan"), dtype=torch.bfloat16, device=dev)
    delta = torch.empty(B, Hh, L, device=dev)
    _lib.check(_lib.lib().bd_attention_bwd(_lib.ptr(dq), _lib.ptr(dm), _lib.ptr(out), _lib.ptr(ddo), _lib.ptr(lse),
                                           _lib.ptr(delta), _lib.ptr(dqkv), B, L, Hh, dh, _s()), "bwd")
    torch.cuda.synchronize()
    from tests.test_gpu_kernels import ref_attention
    qr = qkv.float().requires_grad_(True)
    ref_attention(qr, mask, B, L, Hh, dh).backward(dout.float())
    got = dqkv.float().cpu()
    assert torch.isfinite(got).all()
    for name, sl in (("dQ", slice(0, d)), ("dK", slice(d, 2 * d)), ("dV", slice(2 * d, 3 * d))):
        assert rel(got[:, sl], qr.grad[:, sl]) < 2e-2, name
    assert got.view(B, L, 3 * d)[2].abs().max().item() == 0.0, "fully padded sequence: zero gradients"


@pytest.mark.parametrize("B,L,d,V", [(8, 20, 128, 39), (5, 300, 384, 7), (3, 1, 64, 39)])
def test_embedding_backward(gpu_device, B, L, d, V):
    """Rows of a sequence are merged per token before the atomics: few distinct tokens over 300 positions (more
    positions than threads in the rank sort), a single position, and the small default."""
    dev = gpu_device
    T = 21
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(0, V, (B, L), generator=g)
    t = torch.randint(0, T, (B,), generator=g)
    dx = torch.randn(B * L, d, generator=g)
    dtok, dpos, dtim = torch.zeros(V, d, device=dev), torch.zeros(max(64, L), d, device=dev), torch.zeros(T, d, device=dev)
    a, b, c = ids.to(dev), t.to(dev), dx.to(dev)
    _lib.check(_lib.lib().bd_embed_bwd(_lib.ptr(c), _lib.ptr(a), _lib.ptr(b), _lib.ptr(dtok), _lib.ptr(dpos), _lib.ptr(dtim),
                                       B, L, d, 0, V, max(64, L), T, _s()), "embed bwd")
    torch.cuda.synchronize()
    tok = torch.nn.Embedding(V, d, padding_idx=0)
    pos, tim = torch.nn.Embedding(max(64, L), d), torch.nn.Embedding(T, d)
    (tok(ids) + pos(torch.arange(L))[None] + tim(t)[:, None]).backward(dx.view(B, L, d))
    assert rel(dtok.cpu(), tok.weight.grad) < 1e-5 and dtok[0].abs().max().item() == 0.0
    assert rel(dpos.cpu(), pos.weight.grad) < 1e-5 and rel(dtim.cpu(), tim.weight.grad) < 1e-5


def test_fused_adamw_with_clip_matches_torch(gpu_device):
    dev = gpu_device
    g = torch.Generator().manual_seed(6)
    shapes = [(37, 5), (128,), (64, 64), (3,)]
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]
    rs = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    opt = FlatAdamW(ps, lr=3e-3, weight_decay=0.05)
    ref = torch.optim.AdamW(rs, lr=3e-3, weight_decay=0.05)
    for step in range(5):
        for p, r in zip(ps, rs):
            gr = torch.randn(*r.shape, generator=g) * (10.0 if step % 2 == 0 else 0.01)
            r.grad = gr.clone()
            p.grad.copy_(gr.to(dev))
        torch.nn.utils.clip_grad_norm_(rs, 1.0)
        ref.step()
        opt.step(max_norm=1.0)
        opt.zero_grad()
        for p, r in zip(ps, rs):
            assert (p.detach().cpu() - r.detach()).abs().max().item() < 2e-6
    sd = opt.state_dict()
    assert set(sd["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 5.0


@pytest.mark.parametrize("bad", [float("inf"), float("nan")])
def test_guarded_adamw_follows_gradscaler_skip_rule(gpu_device, bad):
    """skip_nonfinite = GradScaler.unscale_ + clip_grad_norm_ + GradScaler.step around torch.optim.AdamW
    (reference trainer_backbone.py:441-453): a step with a non-finite gradient changes nothing and does not count
    for the bias corrections of the steps after it."""
    dev = gpu_device
    g = torch.Generator().manual_seed(16)
    shapes = [(37, 5), (128,), (64, 64)]
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]
    rs = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    opt = FlatAdamW(ps, lr=3e-3, weight_decay=0.05)
    opt.skip_nonfinite = True
    ref = torch.optim.AdamW(rs, lr=3e-3, weight_decay=0.05)
    scaler = torch.amp.GradScaler("cpu", enabled=True)
    for step in range(7):
        scaler.scale(torch.ones(1))
        for i, (p, r) in enumerate(zip(ps, rs)):
            gr = torch.randn(*r.shape, generator=g) * (10.0 if step % 2 == 0 else 0.01)
            if step in (1, 4) and i == 2:
                gr[3, 7] = bad
            r.grad = gr * scaler.get_scale()
            p.grad.copy_(gr.to(dev))
        before = [p.detach().clone() for p in ps]
        scaler.unscale_(ref)
        torch.nn.utils.clip_grad_norm_(rs, 1.0)
        scaler.step(ref)
        scaler.update()
        opt.step(max_norm=1.0)
        opt.zero_grad()
        for p, r, b in zip(ps, rs, before):
            assert torch.isfinite(p).all()
            assert (p.detach().cpu() - r.detach()).abs().max().item() < 2e-6
            if step in (1, 4):
                assert torch.equal(p.detach(), b)
    assert opt.skipped_steps() == 2
    sd = opt.state_dict()
    assert float(sd["state"][0]["step"]) == 5.0 == float(ref.state[rs[0]]["step"])
    # a restored optimizer continues the count of applied steps
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt2 = FlatAdamW(ps2, lr=3e-3, weight_decay=0.05)
    opt2.skip_nonfinite = True
    opt2.load_state_dict(sd)
    for p, p2, r in zip(ps, ps2, rs):
        gr = torch.randn(*r.shape, generator=g)
        r.grad = gr.clone(); p.grad.copy_(gr.to(dev)); p2.grad.copy_(gr.to(dev))
    ref.step(); opt.step(); opt2.step()
    for p, p2, r in zip(ps, ps2, rs):
        assert (p.detach().cpu() - r.detach()).abs().max().item() < 2e-6
        assert torch.equal(p.detach(), p2.detach())


def _grad_setup(name, dev, B=6, L=24):
    cfg = RI.BACKBONE_CFGS[name]
    torch.manual_seed(7)
    bb = DiffusionBackbone(**cfg)
    RI.perturb_state_dict(bb.state_dict(), 11)
    model = DiscreteMaskingDiffusion(bb, num_steps=cfg["num_diffusion_steps"], mask_token_id=1, pad_token_id=0,
                                     bos_token_id=2, eos_token_id=3)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ids, attn, t = RI.backbone_inputs(cfg, 33, B=B, L=L)
    return cfg, model.to(dev), sd, torch.from_numpy(ids), torch.from_numpy(attn), torch.from_numpy(t)


@pytest.mark.parametrize("name", ["tiny_h2", "tiny_h1"])
def test_training_forward_backward_matches_oracle_autograd(gpu_device, name):
    """loss and every parameter gradient of one training forward/backward vs fp32 autograd of the
    CPU restatement (same corruption uniforms)."""
    dev = gpu_device
    cfg, model, sd, ids, attn, t = _grad_setup(name, dev)
    model.train()
    u = torch.rand(ids.shape, generator=torch.Generator().manual_seed(1))
    noisy, ind = model.forward_process(ids.to(dev), t.to(dev), attn.to(dev), uniforms=u.to(dev))
    logits = model.backbone(noisy, t.to(dev), attn.to(dev))
    assert logits.requires_grad
    loss = model._compute_loss(logits, ids.to(dev), ind, attn.to(dev))
    loss.backward()
    # oracle
    ref_sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and k != "mask_schedule") for k, v in sd.items()}
    rl = BR.logits(ref_sd, cfg, noisy.cpu(), t, attn, prefix="backbone.")
    rloss = BR.masked_ce(rl, ids, ind.cpu(), attn)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) < 2e-2 * abs(rloss.item())
    worst = 0.0
    for k, p in model.named_parameters():
        gref = ref_sd[k].grad
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        if k.endswith("k_proj.bias"):
            # shifting every key by the same vector does not change the softmax: the true gradient
            # is zero (fp32 autograd returns rounding noise), so bound it absolutely instead
            qb = ref_sd[k.replace("k_proj", "q_proj")].grad.norm().item()
            assert p.grad.norm().item() < 0.1 * qb, k   # bf16 rounding of the summed dK rows
            continue
        e = rel(p.grad.cpu(), gref)
        worst = max(worst, e)
        assert e < 1e-1, (k, e)    # bf16 products: a few percent on the smallest gradients
        cos = torch.nn.functional.cosine_similarity(p.grad.cpu().flatten(), gref.flatten(), dim=0).item()
        assert cos > 0.995, (k, cos)
    # position rows never used and the pad row get exactly zero
    assert model.backbone.token_embedding.weight.grad[0].abs().max().item() == 0.0
    assert model.backbone.position_embedding.weight.grad[ids.shape[1]:].abs().max().item() == 0.0


@pytest.mark.parametrize("hidden,heads,ffn", [(768, 6, 3072), (1152, 9, 4608), (1536, 12, 6144)])
def test_training_at_medium_and_large_widths_matches_oracle_autograd(gpu_device, hidden, heads, ffn):
    """The reference's medium, large and xl model sizes (configs/config.yaml model_sizes), two layers of each: loss and
    every parameter gradient against fp32 autograd of the CPU restatement. 7168 rows, so that the products of the
    large width (1152 and 3456: odd multiples of 128) take the large-tile kernel with its half-empty last column
    tile, in the forward pass and in the data-gradient products."""
    dev = gpu_device
    cfg = dict(RI.SMALL_CFG, hidden_size=hidden, num_heads=heads, ffn_hidden_size=ffn, num_layers=2, dropout=0.0)
    torch.manual_seed(7)
    bb = DiffusionBackbone(**cfg)
    RI.perturb_state_dict(bb.state_dict(), 11)
    model = DiscreteMaskingDiffusion(bb, num_steps=cfg["num_diffusion_steps"], mask_token_id=1, pad_token_id=0,
                                     bos_token_id=2, eos_token_id=3)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ids, attn, t = (torch.from_numpy(x) for x in RI.backbone_inputs(cfg, 33, B=56, L=128))
    model = model.to(dev).train()
    u = torch.rand(ids.shape, generator=torch.Generator().manual_seed(1))
    noisy, ind = model.forward_process(ids.to(dev), t.to(dev), attn.to(dev), uniforms=u.to(dev))
    loss = model._compute_loss(model.backbone(noisy, t.to(dev), attn.to(dev)), ids.to(dev), ind, attn.to(dev))
    loss.backward()
    ref_sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and k != "mask_schedule") for k, v in sd.items()}
    rloss = BR.masked_ce(BR.logits(ref_sd, cfg, noisy.cpu(), t, attn, prefix="backbone."), ids, ind.cpu(), attn)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) < 2e-2 * abs(rloss.item())
    for k, p in model.named_parameters():
        gref = ref_sd[k].grad
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        if k.endswith("k_proj.bias"):          # true gradient zero (see the test above)
            continue
        e = rel(p.grad.cpu(), gref)
        cos = torch.nn.functional.cosine_similarity(p.grad.cpu().flatten(), gref.flatten(), dim=0).item()
        assert e < 1e-1 and cos > 0.995, (k, e, cos)


def test_training_with_padded_heads_matches_oracle_autograd(gpu_device):
    """8 heads of 48 (the reference's default block): the kernels run with heads zero-padded to 64;
    loss and all parameter gradients must still match fp32 autograd of the CPU restatement."""
    dev = gpu_device
    cfg = dict(RI.SMALL_CFG, num_heads=8, num_layers=1, dropout=0.0)
    torch.manual_seed(7)
    bb = DiffusionBackbone(**cfg)
    RI.perturb_state_dict(bb.state_dict(), 11)
    model = DiscreteMaskingDiffusion(bb, num_steps=cfg["num_diffusion_steps"], mask_token_id=1, pad_token_id=0,
                                     bos_token_id=2, eos_token_id=3)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ids, attn, t = (torch.from_numpy(x) for x in RI.backbone_inputs(cfg, 33, B=6, L=24))
    model = model.to(dev).train()
    u = torch.rand(ids.shape, generator=torch.Generator().manual_seed(1))
    noisy, ind = model.forward_process(ids.to(dev), t.to(dev), attn.to(dev), uniforms=u.to(dev))
    loss = model._compute_loss(model.backbone(noisy, t.to(dev), attn.to(dev)), ids.to(dev), ind, attn.to(dev))
    loss.backward()
    ref_sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and k != "mask_schedule") for k, v in sd.items()}
    rloss = BR.masked_ce(BR.logits(ref_sd, cfg, noisy.cpu(), t, attn, prefix="backbone."), ids, ind.cpu(), attn)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) < 2e-2 * abs(rloss.item())
    for k, p in model.named_parameters():
        if k.endswith("k_proj.bias"):
            continue        # true gradient is zero (softmax shift invariance)
        cos = torch.nn.functional.cosine_similarity(p.grad.cpu().flatten(), ref_sd[k].grad.flatten(), dim=0).item()
        assert rel(p.grad.cpu(), ref_sd[k].grad) < 1e-1 and cos > 0.995, (k, cos)


def test_training_path_rejects_sequences_longer_than_the_position_table(gpu_device):
    """L = max_position_embeddings + 1 raises on the training path and in fp32 mode, like the reference's
    nn.Embedding does (no read past the table, no gradient written past it)."""
    dev = gpu_device
    cfg, model, sd, ids, attn, t = _grad_setup("tiny_h2", dev)
    L = cfg["max_position_embeddings"] + 1
    long_ids = torch.full((2, L), 7, dtype=torch.long, device=dev)
    tt = torch.ones(2, dtype=torch.long, device=dev)
    model.train()
    with pytest.raises(IndexError, match="max_position_embeddings"):
        model.backbone(long_ids, tt, torch.ones_like(long_ids))
    model.eval()
    with torch.no_grad(), pytest.raises(IndexError, match="max_position_embeddings"):
        model.backbone(long_ids, tt, torch.ones_like(long_ids))
    model.backbone.compute_dtype = "fp32"
    with torch.no_grad(), pytest.raises(IndexError, match="max_position_embeddings"):
        model.backbone(long_ids, tt, torch.ones_like(long_ids))
    # the C entry points reject it too
    x = torch.empty(2 * L, cfg["hidden_size"], device=dev)
    bb = model.backbone
    rc = _lib.lib().bd_embed_bwd(_lib.ptr(x), _lib.ptr(long_ids), _lib.ptr(tt), _lib.ptr(bb.token_embedding.weight),
                                 _lib.ptr(bb.position_embedding.weight), _lib.ptr(bb.timestep_embedding.weight), 2, L,
                                 cfg["hidden_size"], 0, cfg["vocab_size"], cfg["max_position_embeddings"],
                                 cfg["num_diffusion_steps"] + 1, _s())
    assert rc != 0 and b"max_position_embeddings" in _lib.lib().bd_last_error()


def test_backward_into_bound_gradients_equals_autograd_accumulation(gpu_device):
    """With .grad tensors already bound (FlatAdamW) the backward kernels add straight into them;
    the result must equal what autograd's own accumulation gives, including over two micro-steps."""
    dev = gpu_device
    cfg, model, sd, ids, attn, t = _grad_setup("tiny_h2", dev)
    model.train()
    u = torch.rand(ids.shape, generator=torch.Generator().manual_seed(1))

    def fwd_bwd():
        noisy, ind = model.forward_process(ids.to(dev), t.to(dev), attn.to(dev), uniforms=u.to(dev))
        logits = model.backbone(noisy, t.to(dev), attn.to(dev))
        model._compute_loss(logits, ids.to(dev), ind, attn.to(dev)).backward()

    fwd_bwd()                                              # .grad is None: autograd accumulates
    ref = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    opt = FlatAdamW(model.parameters(), lr=1e-3)          # binds .grad to the flat buffer
    opt.zero_grad()
    fwd_bwd()
    fwd_bwd()                                              # second micro-step accumulates on top
    for k, p in model.named_parameters():
        # atomic accumulation order differs between launches: compare with a small tolerance
        assert rel(p.grad.cpu(), 2.0 * ref[k].cpu()) < 1e-4 or ref[k].abs().max().item() < 1e-12, k
    assert opt.flat_grad.data_ptr() == next(iter(model.parameters())).grad.data_ptr()


def test_optimizer_picks_up_gradients_unbound_by_model_zero_grad(gpu_device):
    """model.zero_grad() (torch's default sets .grad to None) unbinds the flat buffer: autograd then creates fresh
    .grad tensors. The step must consume those (copied into the flat buffer and re-bound), not the stale buffer,
    and a parameter left without a gradient raises."""
    dev = gpu_device
    cfg, model, sd, ids, attn, t = _grad_setup("tiny_h2", dev)
    model.train()
    u = torch.rand(ids.shape, generator=torch.Generator().manual_seed(1))

    def fwd_bwd():
        noisy, ind = model.forward_process(ids.to(dev), t.to(dev), attn.to(dev), uniforms=u.to(dev))
        model._compute_loss(model.backbone(noisy, t.to(dev), attn.to(dev)), ids.to(dev), ind, attn.to(dev)).backward()

    opt = FlatAdamW(model.parameters(), lr=1e-3)
    opt.zero_grad()
    fwd_bwd()
    ref = opt.flat_grad.clone()
    opt.flat_grad.fill_(123.0)                 # stale content the step must not consume
    model.zero_grad()                          # .grad = None everywhere
    fwd_bwd()
    assert all(p.grad is not None and p.grad.data_ptr() != opt.flat_grad.data_ptr() for p in list(model.parameters())[1:])
    opt.sync_grads()
    assert rel(opt.flat_grad.cpu(), ref.cpu()) < 1e-4
    base = opt.flat_grad.data_ptr()
    assert all(p.grad.data_ptr() == base + 4 * off for p, off in zip(opt._params, opt._offsets))
    before = opt.flat_param.clone()
    opt.step(max_norm=1.0)
    assert not torch.equal(before, opt.flat_param)
    opt.zero_grad()
    next(iter(model.parameters())).grad = None
    with pytest.raises(RuntimeError, match="no gradient"):
        opt.step()


@pytest.mark.parametrize("heads", [3, 8])
def test_tail_hook_fires_when_the_upper_layers_gradients_are_complete(gpu_device, heads):
    """The data-parallel exchange starts the all-reduce of flat_grad[split:] from a hook inside the backward pass:
    what the hook sees there (a copy enqueued on the hook's stream) must already be the final gradients of the
    layers >= split, the final norm and the vocabulary head, while the head of the buffer is still being written.
    With 8 heads of 48 the attention projections' gradients are accumulated zero-padded and must have joined the
    flat buffer for those layers before the hook, and exactly once."""
    from bidiff.parallel import tail_split_offset
    dev = gpu_device
    cfg = dict(RI.SMALL_CFG, vocab_size=39, num_layers=4, dropout=0.1, num_heads=heads)
    torch.manual_seed(7)
    bb = DiffusionBackbone(**cfg)
    model = DiscreteMaskingDiffusion(bb, num_steps=cfg["num_diffusion_steps"], mask_token_id=1, pad_token_id=0,
                                     bos_token_id=2, eos_token_id=3).to(dev).train()
    ids, attn, t = (torch.from_numpy(x).to(dev) for x in RI.backbone_inputs(cfg, 33, B=64, L=128))
    opt = FlatAdamW(model.parameters(), lr=1e-3)
    off = tail_split_offset(opt, bb, 2)
    assert off is not None and 0 < off < opt.flat_grad.numel()
    first_tail = next(i for i, o in enumerate(opt._offsets) if o == off)
    assert opt._params[first_tail] is bb.layers[2].norm1.weight
    seen = []

    def hook():
        seen.append((opt.flat_grad[off:].clone(), opt.flat_grad[:off].clone()))

    AG.set_tail_ready_hook(hook, 2)
    try:
        for _ in range(3):
            opt.zero_grad()
            seen.clear()
            model(ids, attn)["loss"].backward()
            torch.cuda.synchronize()
            assert len(seen) == 1
            assert torch.equal(seen[0][0], opt.flat_grad[off:]), "tail complete when the hook fires"
            assert not torch.equal(seen[0][1], opt.flat_grad[:off]), "head still in progress"
        # the same backward pass without the hook gives the same gradients (nothing added twice, nothing left out)
        bb.dropout = 0.0               # no dropout decisions: the two passes differ in nothing but the hook
        grads = []
        for hk in (hook, None):
            AG.set_tail_ready_hook(hk, 2)
            opt.zero_grad()
            torch.manual_seed(5)       # same timesteps and corruption
            model(ids, attn)["loss"].backward()
            torch.cuda.synchronize()
            grads.append(opt.flat_grad.clone())
        assert rel(grads[0].cpu(), grads[1].cpu()) < 1e-4
        AG.set_tail_ready_hook(hook, 2)
        # gradients not bound to the flat buffer: autograd accumulates after the backward pass, no early hook
        model.zero_grad()
        seen.clear()
        model(ids, attn)["loss"].backward()
        assert seen == []
    finally:
        AG.set_tail_ready_hook(None)


def test_training_steps_follow_fp32_reference_trajectory(gpu_device):
    """Five optimizer steps (corruption, forward, backward, clip 1.0, AdamW) on the GPU next to the
    same five steps of the fp32 CPU restatement with torch.optim.AdamW: losses stay within bf16
    tolerance and the parameters end up close."""
    dev = gpu_device
    cfg, model, sd, ids, attn, t = _grad_setup("tiny_h2", dev, B=16, L=24)
    model.train()
    opt = FlatAdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    ref_params = {k: torch.nn.Parameter(v.clone()) for k, v in sd.items() if k != "mask_schedule"}
    ropt = torch.optim.AdamW(ref_params.values(), lr=1e-3, weight_decay=0.01)
    g = torch.Generator().manual_seed(3)
    losses, rlosses = [], []
    for step in range(5):
        tt = torch.randint(1, cfg["num_diffusion_steps"] + 1, (ids.shape[0],), generator=g)
        u = torch.rand(ids.shape, generator=g)
        noisy, ind = model.forward_process(ids.to(dev), tt.to(dev), attn.to(dev), uniforms=u.to(dev))
        loss = model._compute_loss(model.backbone(noisy, tt.to(dev), attn.to(dev)), ids.to(dev), ind, attn.to(dev))
        loss.backward()
        opt.step(max_norm=1.0)
        opt.zero_grad()
        losses.append(loss.item())
        rl = BR.masked_ce(BR.logits(ref_params, cfg, noisy.cpu(), tt, attn, prefix="backbone."), ids, ind.cpu(), attn)
        ropt.zero_grad()
        rl.backward()
        torch.nn.utils.clip_grad_norm_(ref_params.values(), 1.0)
        ropt.step()
        rlosses.append(rl.item())
    for a, b in zip(losses, rlosses):
        assert abs(a - b) < 3e-2 * abs(b), (losses, rlosses)
    for k, p in model.named_parameters():
        assert (p.detach().cpu() - ref_params[k].detach()).abs().max().item() < 5e-3, k


def test_trainer_refuses_fp32_training(gpu_device, tmp_path):
    """use_amp false means an fp32 step in the reference (autocast(enabled=use_amp), trainer_backbone.py:434); the
    training kernels compute in bf16, so the trainer says so instead of training in narrower arithmetic."""
    from bidiff.training import BackboneTrainer
    cfg, model, sd, ids, attn, t = _grad_setup("tiny_h2", gpu_device, B=4, L=24)
    tc = {"learning_rate": 1e-3, "weight_decay": 0.01, "warmup_steps": 1, "max_steps": 2, "gradient_clip_norm": 1.0,
          "eval_every": 10, "save_every": 10}
    for opt in ({"use_amp": False}, {}):
        with pytest.raises(NotImplementedError, match="use_amp"):
            BackboneTrainer(model, [], [], {"optimization": opt, "training_backbone": tc}, device=str(gpu_device),
                            output_dir=str(tmp_path))


def test_trainer_runs_and_checkpoints(gpu_device, tmp_path):
    from bidiff.training import BackboneTrainer
    cfg, model, sd, ids, attn, t = _grad_setup("tiny_h2", gpu_device, B=8, L=24)
    batches = [{"input_ids": ids, "attention_mask": attn} for _ in range(6)]
    conf = {"optimization": {"use_amp": True, "gradient_accumulation_steps": 2},
            "training_backbone": {"learning_rate": 1e-3, "weight_decay": 0.01, "warmup_steps": 2, "max_steps": 6,
                                  "gradient_clip_norm": 1.0, "eval_every": 2, "save_every": 100, "num_epochs": 2},
            "checkpointing": {"save_best_only": True}}
    tr = BackboneTrainer(model, batches, batches[:2], conf, device=str(gpu_device), output_dir=str(tmp_path))
    hist = tr.train()
    assert len(hist["train_losses"]) == 12 and all(math.isfinite(x) for x in hist["train_losses"])
    assert hist["train_losses"][-1] < hist["train_losses"][0]
    ck = torch.load(tmp_path / "checkpoints" / "backbone_best.pt", map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "global_step", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict",
                       "val_loss", "best_val_loss", "config"}
    assert list(ck["model_state_dict"].keys()) == list(sd.keys())
    head = (tmp_path / "metrics" / "backbone_loss_curve.csv").read_text().splitlines()[0]
    assert head == "step,train_loss,train_bpb,learning_rate"
    assert (tmp_path / "metrics" / "backbone_val_loss.csv").read_text().splitlines()[0] == "step,val_loss,val_bpb"
    assert (tmp_path / "metrics" / "backbone_epoch_loss_curve.csv").read_text().splitlines()[0] == \
        "epoch,train_loss,val_loss,train_bpb,val_bpb"
    # save_best_only=True: no final checkpoint, whatever save_last says (reference trainer_backbone.py:631)
    assert not (tmp_path / "checkpoints" / "backbone_last.pt").exists()
    tr.load_checkpoint(str(tmp_path / "checkpoints" / "backbone_best.pt"))
    # a periodic checkpoint of the reference has no best_val_loss key
    del ck["best_val_loss"], ck["val_loss"], ck["config"]
    torch.save(ck, tmp_path / "periodic.pt")
    tr.load_checkpoint(str(tmp_path / "periodic.pt"))
    assert tr.best_val_loss == float("inf")


def test_multi_cast_jobs(gpu_device):
    """One launch, many fp32 -> bf16 copies: plain, transposed, into sub-blocks with a leading dimension,
    ragged shapes, and the fp32 pass-through used for stacked biases."""
    dev = gpu_device
    g = torch.Generator().manual_seed(0)
    a = torch.randn(384, 384, generator=g).to(dev)
    b = torch.randn(803, 128, generator=g).to(dev)
    c = torch.randn(70, 33, generator=g).to(dev)
    bias = torch.randn(384, generator=g).to(dev)
    stack = torch.full((3 * 384, 384), 7.0, dtype=torch.bfloat16, device=dev)
    stackT = torch.full((384, 3 * 384), 7.0, dtype=torch.bfloat16, device=dev)
    bT = torch.zeros(128, 896, dtype=torch.bfloat16, device=dev)
    cc = torch.zeros(70, 40, dtype=torch.bfloat16, device=dev)
    cT = torch.zeros(33, 70, dtype=torch.bfloat16, device=dev)
    bst = torch.zeros(3 * 384, device=dev)
    jobs = [_lib.CastJob(a.data_ptr(), stack.data_ptr() + 2 * 384 * 384, 384, 384, 384, 0),          # rows 384..767
            _lib.CastJob(a.data_ptr(), stackT.data_ptr() + 2 * 384, 384, 384, 3 * 384, 1),          # columns 384..767
            _lib.CastJob(b.data_ptr(), bT.data_ptr(), 803, 128, 896, 1),
            _lib.CastJob(c.data_ptr(), cc.data_ptr(), 70, 33, 40, 0),
            _lib.CastJob(c.data_ptr(), cT.data_ptr(), 70, 33, 70, 1),
            _lib.CastJob(bias.data_ptr(), bst.data_ptr() + 4 * 768, 1, 384, 3 * 384, 2)]
    arr = (_lib.CastJob * len(jobs))(*jobs)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    _lib.check(_lib.lib().bd_multi_cast_bf16(_lib.ptr(table), len(jobs), 803, 384, _s()), "multi_cast")
    torch.cuda.synchronize()
    assert torch.equal(stack[384:768], a.bfloat16()) and (stack[:384] == 7).all() and (stack[768:] == 7).all()
    assert torch.equal(stackT[:, 384:768], a.t().bfloat16()) and (stackT[:, :384] == 7).all() and (stackT[:, 768:] == 7).all()
    assert torch.equal(bT[:, :803], b.t().bfloat16()) and (bT[:, 803:] == 0).all()
    assert torch.equal(cc[:, :33], c.bfloat16()) and (cc[:, 33:] == 0).all()
    assert torch.equal(cT, c.t().bfloat16())
    assert torch.equal(bst[768:], bias) and (bst[:768] == 0).all()


def test_cast_jobs_reblock_heads_and_unpad_gradients(gpu_device):
    """Head re-blocking of the cast jobs (8 heads of 48 stored as 8 zero-padded heads of 64, with a scale): rows for the
    q / k / v projections, columns for the output projection, straight and transposed, and the fp32 bias; then the way
    back for gradients, bd_multi_unpad_add_f32, which ADDS scale * (the real entries of the padded gradient)."""
    dev = gpu_device
    H, dh, dhp, d = 8, 48, 64, 384
    aw = H * dhp
    g = torch.Generator().manual_seed(2)
    wq = torch.randn(H * dh, d, generator=g).to(dev)
    wo = torch.randn(d, H * dh, generator=g).to(dev)
    bq = torch.randn(H * dh, generator=g).to(dev)
    f = (dhp / dh) ** 0.5
    wq_p = torch.zeros(aw, d, dtype=torch.bfloat16, device=dev); wq_pT = torch.zeros(d, aw, dtype=torch.bfloat16, device=dev)
    wo_p = torch.zeros(d, aw, dtype=torch.bfloat16, device=dev); wo_pT = torch.zeros(aw, d, dtype=torch.bfloat16, device=dev)
    bq_p = torch.zeros(aw, device=dev)
    J = _lib.CastJob
    jobs = [J(wq.data_ptr(), wq_p.data_ptr(), H * dh, d, d, 0, 1, dh, dhp, f),
            J(wq.data_ptr(), wq_pT.data_ptr(), H * dh, d, aw, 1, 1, dh, dhp, f),
            J(wo.data_ptr(), wo_p.data_ptr(), d, H * dh, aw, 0, 2, dh, dhp, 1.0),
            J(wo.data_ptr(), wo_pT.data_ptr(), d, H * dh, d, 1, 2, dh, dhp, 1.0),
            J(bq.data_ptr(), bq_p.data_ptr(), 1, H * dh, aw, 2, 2, dh, dhp, f)]
    arr = (J * len(jobs))(*jobs)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    _lib.check(_lib.lib().bd_multi_cast_bf16(_lib.ptr(table), len(jobs), 384, 384, _s()), "multi_cast")
    pad_rows = lambda w: torch.nn.functional.pad(w.view(H, dh, -1), (0, 0, 0, dhp - dh)).reshape(aw, -1)
    want_q = pad_rows(wq * f).bfloat16()
    want_o = pad_rows(wo.t().contiguous()).t().bfloat16()
    assert torch.equal(wq_p, want_q) and torch.equal(wq_pT, want_q.t())
    assert torch.equal(wo_p, want_o) and torch.equal(wo_pT, want_o.t())
    assert torch.equal(bq_p, pad_rows((bq * f)[:, None])[:, 0])
    # gradients back: dst += scale * real entries; the padding entries of the source are ignored
    gq_p = torch.randn(aw, d, generator=g).to(dev); go_p = torch.randn(d, aw, generator=g).to(dev); gb_p = torch.randn(aw, generator=g).to(dev)
    gq = torch.ones(H * dh, d, device=dev); go = torch.full((d, H * dh), 2.0, device=dev); gb = torch.zeros(H * dh, device=dev)
    plain_src = torch.randn(70, 33, generator=g).to(dev); plain_dst = torch.ones(70, 33, device=dev)
    jobs = [J(gq_p.data_ptr(), gq.data_ptr(), H * dh, d, d, 0, 1, dh, dhp, f),
            J(go_p.data_ptr(), go.data_ptr(), d, H * dh, aw, 0, 2, dh, dhp, 1.0),
            J(gb_p.data_ptr(), gb.data_ptr(), 1, H * dh, aw, 0, 2, dh, dhp, f),
            J(plain_src.data_ptr(), plain_dst.data_ptr(), 70, 33, 33, 0, 0, 0, 0, 0.0)]
    arr = (J * len(jobs))(*jobs)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    _lib.check(_lib.lib().bd_multi_unpad_add_f32(_lib.ptr(table), len(jobs), 384, 384, _s()), "unpad")
    unpad_rows = lambda w: w.view(H, dhp, -1)[:, :dh].reshape(H * dh, -1)
    assert torch.allclose(gq, 1.0 + f * unpad_rows(gq_p), atol=1e-6)
    assert torch.allclose(go, 2.0 + unpad_rows(go_p.t().contiguous()).t(), atol=1e-6)
    assert torch.allclose(gb, f * unpad_rows(gb_p[:, None])[:, 0], atol=1e-6)
    assert torch.allclose(plain_dst, 1.0 + plain_src, atol=1e-6)


def test_training_weight_copies_follow_the_parameters(gpu_device):
    """The persistent bf16 copies are refreshed (one launch) after an optimizer step and equal fresh casts."""
    dev = gpu_device
    cfg, model, sd, ids, attn, t = _grad_setup("tiny_h2", dev)
    model.train()
    bb = model.backbone
    opt = FlatAdamW(model.parameters(), lr=1e-2)
    for _ in range(2):
        out = model(ids.to(dev), attn.to(dev))
        out["loss"].backward()
        opt.step()
        opt.zero_grad()
    tw = AG._train_weights(bb)
    assert tw.jobs is not None
    blk = bb.layers[1]
    w = tw.layers[1]
    want = torch.cat([blk.attn.q_proj.weight, blk.attn.k_proj.weight, blk.attn.v_proj.weight], 0)
    assert torch.equal(w["wqkv"], want.bfloat16()) and torch.equal(w["wqkvT"], want.t().bfloat16())
    assert torch.equal(w["bqkv"], torch.cat([blk.attn.q_proj.bias, blk.attn.k_proj.bias, blk.attn.v_proj.bias]))
    assert torch.equal(w["w1T"], blk.ffn.fc1.weight.t().bfloat16()) and torch.equal(w["w2"], blk.ffn.fc2.weight.bfloat16())
    assert torch.equal(w["woT"], blk.attn.out_proj.weight.t().bfloat16())
    V = cfg["vocab_size"]
    assert torch.equal(tw.wout[:V], bb.output_proj.weight.bfloat16()) and (tw.wout[V:] == 0).all()
    assert torch.equal(tw.woutT[:, :V], bb.output_proj.weight.t().bfloat16()) and (tw.woutT[:, V:] == 0).all()
    same = AG._train_weights(bb)
    assert same is tw


def test_pack_targets_gathers_targets_and_folds_the_masks(gpu_device):
    """bd_pack_targets: packed row r takes the target of logical token rowmap[r]; its loss flag is on where the token
    was masked AND attended (diffusion.py:190-200), off for filler rows."""
    dev = gpu_device
    rs = np.random.RandomState(5)
    B, L, n_rows = 7, 19, 96
    tg = rs.randint(0, 800, size=B * L).astype(np.int64)
    ind = rs.rand(B * L) < 0.5
    am = (rs.rand(B * L) < 0.8).astype(np.int64)
    rowmap = np.full(n_rows, -1, dtype=np.int32)
    real = rs.permutation(B * L)[:80]
    rowmap[:80] = np.sort(real)
    lib = _lib.lib()
    tg_d, ind_d = torch.from_numpy(tg).to(dev), torch.from_numpy(ind).to(dev).view(torch.uint8)
    am_d, rm_d = torch.from_numpy(am).to(dev), torch.from_numpy(rowmap).to(dev)
    for with_mask in (True, False):
        out_t = torch.full((n_rows,), -7, dtype=torch.long, device=dev)
        out_i = torch.full((n_rows,), 9, dtype=torch.uint8, device=dev)
        rc = lib.bd_pack_targets(_lib.ptr(tg_d), _lib.ptr(ind_d), _lib.ptr(am_d) if with_mask else None, _lib.ptr(rm_d),
                                 n_rows, _lib.ptr(out_t), _lib.ptr(out_i), _lib.stream_handle())
        assert rc == 0, lib.bd_last_error()
        want_t = np.where(rowmap >= 0, tg[np.maximum(rowmap, 0)], 0)
        want_i = (rowmap >= 0) & ind[np.maximum(rowmap, 0)] & ((am[np.maximum(rowmap, 0)] != 0) if with_mask else True)
        assert np.array_equal(out_t.cpu().numpy(), want_t)
        assert np.array_equal(out_i.cpu().numpy(), want_i.astype(np.uint8))


def test_masked_ce_finish_divides_and_keeps_the_empty_case_at_zero(gpu_device):
    dev = gpu_device
    lib = _lib.lib()
    for sums, want in (([6.0, 4.0], 1.5), ([0.0, 0.0], 0.0), ([3.0, 0.0], 0.0)):
        s_ = torch.tensor(sums, device=dev)
        out = torch.full((), -1.0, device=dev)
        assert lib.bd_masked_ce_finish(_lib.ptr(s_), _lib.ptr(out), 0, _lib.stream_handle()) == 0
        assert float(out) == want
    # pairs behind the totals: added into sums[0..1], where the backward pass reads them
    s_ = torch.zeros(32 * 4, device=dev)
    s_[32], s_[33], s_[64], s_[65] = 1.0, 2.0, 5.0, 2.0
    out = torch.full((), -1.0, device=dev)
    assert lib.bd_masked_ce_finish(_lib.ptr(s_), _lib.ptr(out), 3, _lib.stream_handle()) == 0
    assert float(out) == 1.5 and s_[:2].tolist() == [6.0, 4.0]


@pytest.mark.parametrize("n_parts", [1, 8, 13])
def test_layernorm_bwd_parts_sum_to_the_single_copy_result(gpu_device, n_parts):
    """bd_layernorm_bwd_parts spreads the gamma / beta sums over copies; bd_layernorm_parts_finish ADDS their total to
    the destinations (several jobs in one launch) and leaves the copies zeroed. dx does not depend on the split."""
    import ctypes as C
    dev = gpu_device
    lib = _lib.lib()
    P, st_ = _lib.ptr, _lib.stream_handle
    M, d = 2048 + 24, 384
    g = torch.Generator().manual_seed(11)
    dy = torch.randn(M, d, generator=g).to(dev).bfloat16()
    x = torch.randn(M, d, generator=g).to(dev)
    stats = torch.stack([x.mean(1), 1 / (x.var(1, unbiased=False) + 1e-5).sqrt()], 1).contiguous()
    gm = torch.randn(d, generator=g).to(dev)
    dres = torch.randn(M, d, generator=g).to(dev)
    dg0, db0 = torch.randn(d, generator=g).to(dev), torch.randn(d, generator=g).to(dev)
    # single copy
    dx1, dg1, db1 = dres.clone(), dg0.clone(), db0.clone()
    assert lib.bd_layernorm_bwd_cast(P(dy), P(x), P(stats), P(gm), P(dx1), P(dx1), P(dg1), P(db1), M, d, None, None, 0, st_()) == 0
    # two jobs through the copies: the same LayerNorm twice, into two destinations
    parts = torch.zeros(2, n_parts, 2, d, device=dev)
    dx2 = [dres.clone(), dres.clone()]
    dst = [(dg0.clone(), db0.clone()), (torch.zeros(d, device=dev), torch.zeros(d, device=dev))]
    for k in range(2):
        assert lib.bd_layernorm_bwd_parts(P(dy), P(x), P(stats), P(gm), P(dx2[k]), P(dx2[k]), P(parts[k]), n_parts, M, d,
                                          None, None, 0, st_()) == 0, lib.bd_last_error()
    arr = (AG._LnPartsJob * 2)(*[AG._LnPartsJob(parts[k].data_ptr(), dst[k][0].data_ptr(), dst[k][1].data_ptr()) for k in range(2)])
    jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    assert lib.bd_layernorm_parts_finish(P(jobs), 2, n_parts, d, st_()) == 0, lib.bd_last_error()
    torch.cuda.synchronize()
    assert torch.equal(dx1, dx2[0]) and torch.equal(dx1, dx2[1])
    scale = float((dg1 - dg0).abs().max())
    assert float((dst[0][0] - dg1).abs().max()) < 1e-4 * scale and float((dst[0][1] - db1).abs().max()) < 1e-4 * scale
    assert float((dst[1][0] - (dg1 - dg0)).abs().max()) < 1e-4 * scale
    assert float(parts.abs().max()) == 0.0


def test_partial_sum_entry_points_reject_bad_arguments(gpu_device):
    """The entry points with partial-sum copies refuse copy counts outside 1..64 (0..64 for the cross-entropy, where 0
    is the plain pair) and null buffers, with a message, instead of launching."""
    dev = gpu_device
    lib = _lib.lib()
    P, st_ = _lib.ptr, _lib.stream_handle
    M, d, V = 64, 128, 40
    dy = torch.zeros(M, d, device=dev, dtype=torch.bfloat16); x = torch.zeros(M, d, device=dev)
    stats = torch.ones(M, 2, device=dev); gm = torch.ones(d, device=dev); dx = torch.zeros(M, d, device=dev)
    parts = torch.zeros(2, 2, d, device=dev)
    for n_parts in (0, 65):
        assert lib.bd_layernorm_bwd_parts(P(dy), P(x), P(stats), P(gm), None, P(dx), P(parts), n_parts, M, d, None, None, 0, st_()) != 0
        assert b"parts" in lib.bd_last_error()
    assert lib.bd_layernorm_bwd_parts(P(dy), P(x), P(stats), P(gm), None, P(dx), None, 2, M, d, None, None, 0, st_()) != 0
    assert lib.bd_layernorm_parts_finish(None, 1, 2, d, st_()) != 0
    lg = torch.zeros(M, V, device=dev); tg = torch.zeros(M, dtype=torch.long, device=dev)
    ind = torch.ones(M, dtype=torch.uint8, device=dev); lse = torch.zeros(M, device=dev); sums = torch.zeros(32 * 3, device=dev)
    assert lib.bd_masked_ce_fwd_parts(P(lg), P(tg), P(ind), None, P(lse), P(sums), 65, M, V, st_()) != 0
    assert lib.bd_masked_ce_fwd_parts(P(lg), P(tg), P(ind), None, P(lse), None, 2, M, V, st_()) != 0
    assert lib.bd_masked_ce_finish(P(sums), None, 2, st_()) != 0
    assert lib.bd_pack_targets(P(tg), P(ind), None, None, M, P(tg), P(ind), st_()) != 0
    # and the good calls still go through
    assert lib.bd_masked_ce_fwd_parts(P(lg), P(tg), P(ind), None, P(lse), P(sums), 2, M, V, st_()) == 0
    loss = torch.zeros((), device=dev)
    assert lib.bd_masked_ce_finish(P(sums), P(loss), 2, st_()) == 0
    assert abs(float(loss) - math.log(V)) < 1e-4 and abs(float(sums[1]) - M) < 1e-3
