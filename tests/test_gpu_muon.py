"""GPU parity of the Muon path: batched bf16 GEMM with the rounded axpby epilogue, bd_muon_step
against the CPU oracle evaluated with bf16 tensors (the arithmetic the reference runs on a GPU)
and fp32 (what it runs on CPU, the golden vectors), and FlatMuonAdamW over a model."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import replay_inputs as RI
from oracle import optimizer_ref as OR
from bidiff import _lib
from bidiff.model.backbone import DiffusionBackbone
from bidiff.training.optimizer import FlatMuonAdamW
from bidiff.training.optimizer_setup import build_step1_backbone_optimizer
from tests.test_oracle_optimizer_golden import BLOB, oracle_trajectory

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def cos(a, b):
    return float(torch.nn.functional.cosine_similarity(torch.as_tensor(a).double().flatten(),
                                                       torch.as_tensor(b).double().flatten(), dim=0))


@pytest.mark.parametrize("nb,M,N,K", [(3, 128, 128, 128), (2, 384, 384, 1536), (5, 200, 256, 64)])
def test_batched_gemm_and_rounded_axpby(gpu_device, nb, M, N, K):
    dev = gpu_device
    g = torch.Generator().manual_seed(nb + M + N + K)
    A = torch.randn(nb, M, K, generator=g).bfloat16()
    W = (torch.randn(nb, N, K, generator=g) / K ** 0.5).bfloat16()
    X = torch.randn(nb, M, N, generator=g).bfloat16()
    Ad, Wd, Xd = A.to(dev), W.to(dev), X.to(dev)
    out = torch.empty(nb, M, N, dtype=torch.bfloat16, device=dev)
    s = _lib.stream_handle()
    _lib.check(_lib.lib().bd_gemm_bf16_batched(_lib.ptr(Ad), _lib.ptr(Wd), _lib.ptr(out), None, nb, M * K, N * K, M * N,
                                               M, N, K, 1.0, 0.0, s), "gemm")
    ref = torch.bmm(A.float(), W.float().transpose(1, 2))
    assert rel(out.float().cpu(), ref) < 4e-3
    # shared W (stride 0) and the rounded axpby epilogue, in place on the aux operand
    a, b = 2.0315, -4.7750
    _lib.check(_lib.lib().bd_gemm_bf16_batched(_lib.ptr(Ad), _lib.ptr(Wd), _lib.ptr(Xd), _lib.ptr(Xd), nb, M * K, 0, M * N,
                                               M, N, K, a, b, s), "gemm")
    prod = torch.matmul(A.float(), W[0].float().t()).bfloat16()
    want = ((b * X).float() + (a * prod).float()).bfloat16()        # each bf16 op rounds
    assert rel(Xd.float().cpu(), want.float()) < 6e-3
    frac_exact = (Xd.cpu() == want).float().mean().item()
    assert frac_exact > 0.9      # same roundings except where the fp32 sums differ in the last bit
    assert _lib.lib().bd_gemm_bf16_batched(_lib.ptr(Ad), _lib.ptr(Wd), _lib.ptr(out), None, nb, M * K, N * K, M * N, M,
                                           N + 1, K, 1.0, 0.0, s) != 0


@pytest.mark.parametrize("nb,rows,cols", [(4, 128, 128), (2, 384, 1536), (2, 1536, 384), (3, 256, 128)])
def test_muon_step_matches_oracle(gpu_device, nb, rows, cols):
    dev = gpu_device
    g = torch.Generator().manual_seed(rows + 3 * cols)
    p0 = torch.randn(nb, rows, cols, generator=g) * 0.05
    buf0 = torch.randn(nb, rows, cols, generator=g) * 0.01
    grads = [torch.randn(nb, rows, cols, generator=g) * 0.02 for _ in range(2)]
    lr, wd, mom = 6e-3, 0.01, 0.95
    pd, bd = p0.to(dev).clone(), buf0.to(dev).clone()
    ws = torch.empty(int(_lib.lib().bd_muon_workspace_bytes(nb, rows, cols)), dtype=torch.uint8, device=dev)
    ref16 = [(p0[i].clone(), buf0[i].clone()) for i in range(nb)]
    ref32 = [(p0[i].clone(), buf0[i].clone()) for i in range(nb)]
    for gr in grads:
        gd = gr.to(dev)
        _lib.check(_lib.lib().bd_muon_step(_lib.ptr(pd), _lib.ptr(gd), _lib.ptr(bd), _lib.ptr(ws), ws.numel(), nb, rows,
                                           cols, lr, wd, mom, 1, 5, None, 0.0, _lib.stream_handle()), "muon")
        for i in range(nb):
            OR.muon_update(ref16[i][0], gr[i], ref16[i][1], lr, wd, mom, True, 5, torch.bfloat16)
            OR.muon_update(ref32[i][0], gr[i], ref32[i][1], lr, wd, mom, True, 5, torch.float32)
    torch.cuda.synchronize()
    for i in range(nb):
        # momentum buffers are plain fp32 arithmetic
        assert torch.allclose(bd[i].cpu(), ref16[i][1], rtol=1e-5, atol=1e-8)
        up_got = pd[i].cpu() - p0[i]
        up16, up32 = ref16[i][0] - p0[i], ref32[i][0] - p0[i]
        assert rel(up_got, up16) < 3e-2 and cos(up_got, up16) > 0.9995, (i, rel(up_got, up16))
        assert rel(up_got, up32) < 1e-1 and cos(up_got, up32) > 0.995, (i, rel(up_got, up32))
    # plain momentum (no Nesterov), one Newton-Schulz step, no weight decay, with the clip coefficient
    pd2, bd2 = p0.to(dev).clone(), buf0.to(dev).clone()
    gsq = torch.tensor([25.0], device=dev)     # norm 5 -> coefficient 1 / (5 + 1e-6)
    _lib.check(_lib.lib().bd_muon_step(_lib.ptr(pd2), _lib.ptr(grads[0].to(dev)), _lib.ptr(bd2), _lib.ptr(ws), ws.numel(), nb,
                                       rows, cols, lr, 0.0, 0.5, 0, 1, _lib.ptr(gsq), 1.0, _lib.stream_handle()), "muon")
    for i in range(nb):
        p, b = p0[i].clone(), buf0[i].clone()
        OR.muon_update(p, grads[0][i] * (1.0 / (5.0 + 1e-6)), b, lr, 0.0, 0.5, False, 1, torch.bfloat16)
        assert torch.allclose(bd2[i].cpu(), b, rtol=1e-5, atol=1e-8)
        assert rel(pd2[i].cpu() - p0[i], p - p0[i]) < 3e-2
    assert _lib.lib().bd_muon_step(_lib.ptr(pd), _lib.ptr(pd), _lib.ptr(bd), _lib.ptr(ws), 16, nb, rows, cols, lr, wd, mom,
                                   1, 5, None, 0.0, _lib.stream_handle()) != 0
    assert _lib.lib().bd_muon_step(_lib.ptr(pd), _lib.ptr(pd), _lib.ptr(bd), _lib.ptr(ws), ws.numel(), nb, rows, 100, lr, wd,
                                   mom, 1, 5, None, 0.0, _lib.stream_handle()) != 0


def test_flat_muon_adamw_follows_reference_trajectory(gpu_device, tmp_path):
    dev = gpu_device
    cfg = RI.BACKBONE_CFGS["tiny_h2"]
    torch.manual_seed(7)
    bb = DiffusionBackbone(**cfg)
    RI.perturb_state_dict(bb.state_dict(), 11)
    start = {n: p.detach().clone() for n, p in bb.named_parameters()}
    bb = bb.to(dev)
    opt, info = build_step1_backbone_optimizer(bb, RI.MUON_OPT_CFG, 3e-4, 0.01)
    assert isinstance(opt, FlatMuonAdamW) and info["type"] == "muon_adamw"
    assert info["num_muon_tensors"] == int(BLOB["info/num_muon_tensors"])
    assert [g["kind"] for g in opt.param_groups] == ["adamw", "muon"]
    snaps16, _ = oracle_trajectory(torch.bfloat16)
    named = dict(bb.named_parameters())
    for step in range(RI.OPT_STEPS):
        grads = RI.optimizer_grads(bb, step)
        for n, p in named.items():
            assert p.grad is not None and p.grad.data_ptr() >= opt.flat_grad.data_ptr()
            p.grad.copy_(torch.from_numpy(grads[n]))
        opt.step(max_norm=1.0)
        opt.zero_grad()
        if step in (0, RI.OPT_STEPS - 1):
            for n in RI.OPT_WATCH:
                got = named[n].detach().cpu()
                want32 = torch.from_numpy(BLOB[f"muon_adamw/step{step}/{n}"])
                if named[n].ndim == 2 and "layers." in n:        # Muon: bf16 Newton-Schulz vs the fp32 golden
                    assert rel(got - start[n], want32 - start[n]) < 1e-1, (step, n)
                    assert cos(got - start[n], want32 - start[n]) > 0.995, (step, n)
                    assert rel(got - start[n], snaps16[step][n] - start[n]) < 5e-2, (step, n)   # measured 0.019-0.022: the clip norm's atomics move last bits, bf16 Newton-Schulz amplifies them
                else:                                            # AdamW group: fp32 arithmetic
                    assert torch.allclose(got, want32, rtol=1e-5, atol=2e-6), (step, n)
    # state_dict layout of the reference: group 0 adamw (step / exp_avg / exp_avg_sq), group 1 muon (momentum_buffer)
    sd = opt.state_dict()
    assert [int(i) for i in BLOB["state/muon_param_index"]] == sd["param_groups"][1]["params"][:3]
    i0 = sd["param_groups"][1]["params"][0]
    assert set(sd["state"][i0]) == {"momentum_buffer"}
    assert torch.allclose(sd["state"][i0]["momentum_buffer"].cpu(), torch.from_numpy(BLOB["state/momentum_buffer0"]),
                          rtol=1e-5, atol=1e-7)
    a0 = sd["param_groups"][0]["params"][0]
    assert set(sd["state"][a0]) == {"step", "exp_avg", "exp_avg_sq"} and int(sd["state"][a0]["step"]) == RI.OPT_STEPS
    # resume: a fresh optimizer loaded from the checkpoint continues identically
    path = str(tmp_path / "opt.pt")
    torch.save({"opt": sd, "model": bb.state_dict()}, path)
    torch.manual_seed(7)
    bb2 = DiffusionBackbone(**cfg).to(dev)
    ck = torch.load(path, weights_only=True)
    bb2.load_state_dict(ck["model"])
    opt2, _ = build_step1_backbone_optimizer(bb2, RI.MUON_OPT_CFG, 3e-4, 0.01)
    opt2.load_state_dict(ck["opt"])
    grads = RI.optimizer_grads(bb, 7)
    before = {n: p.detach().clone() for n, p in bb.named_parameters()}
    for m, o in ((bb, opt), (bb2, opt2)):
        for n, p in m.named_parameters():
            p.grad.copy_(torch.from_numpy(grads[n]))
        o.step(max_norm=1.0)
    for (n, p), (_, q) in zip(bb.named_parameters(), bb2.named_parameters()):
        # the global-norm reduction adds block partials with atomics (order-dependent last bits), so
        # the two runs agree to rounding (fp32 for AdamW, bf16 inside Newton-Schulz), not bit for bit
        if p.ndim == 2 and "layers." in n:
            assert rel(p.detach() - before[n], q.detach() - before[n]) < 5e-2, n
        else:
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), n


def test_guarded_muon_adamw_skips_nonfinite_steps(gpu_device):
    """skip_nonfinite on the mixed optimizer: a NaN anywhere in the gradients leaves both groups' parameters and
    states alone; the next finite step equals the step of an optimizer that never saw the bad one."""
    dev = gpu_device
    torch.manual_seed(3)
    mk = lambda: [torch.nn.Parameter(t.clone().to(dev)) for t in base]
    base = [torch.randn(128, 256), torch.randn(256, 128), torch.randn(128, 256), torch.randn(77)]
    grads = [[torch.randn_like(t) for t in base] for _ in range(3)]
    opts = []
    for _ in range(2):
        ps = mk()
        opt = FlatMuonAdamW([dict(params=ps[:3], kind="muon", lr=0.02, weight_decay=0.01, momentum=0.95),
                             dict(params=ps[3:], kind="adamw", lr=1e-3, betas=(0.9, 0.95), weight_decay=0.0)])
        opt.skip_nonfinite = True
        opts.append((opt, ps))
    (a, pa), (b, pb) = opts
    for step, gs in enumerate(grads):
        for p, g in zip(pa, gs):
            p.grad.copy_(g.to(dev))
        a.step(); a.zero_grad()
        if step == 0:          # optimizer b sees one poisoned step in between
            for p, g in zip(pb, gs):
                p.grad.copy_(g.to(dev))
            b.step(); b.zero_grad()
            snap = [p.detach().clone() for p in pb]
            mom = b.flat_m.clone()
            for p, g in zip(pb, gs):
                p.grad.copy_(g.to(dev))
            pb[1].grad[5, 5] = float("nan")
            b.step(); b.zero_grad()
            assert all(torch.equal(p.detach(), s) for p, s in zip(pb, snap)) and torch.equal(b.flat_m, mom)
        else:
            for p, g in zip(pb, gs):
                p.grad.copy_(g.to(dev))
            b.step(); b.zero_grad()
    # no clipping here: the gradient norm is summed with atomics, and a last-bit difference of the clip factor is
    # amplified by the bf16 Newton-Schulz iterations; without it the two trajectories are the same bits
    assert all(torch.equal(x.detach(), y.detach()) for x, y in zip(pa, pb))
    assert b.skipped_steps() == 1 and a.skipped_steps() == 0
    assert b.state_dict()["state"][3]["step"] == 3


@pytest.mark.parametrize("size", ["tiny", "medium_width", "large_width"])
def test_training_steps_with_muon_reduce_the_loss(gpu_device, size):
    """End to end: the diffusion training step with the Muon + AdamW optimizer learns a fixed batch; also at the
    reference's medium and large widths (two layers: shape classes 768^2 / 3072 x 768 and 1152^2 / 4608 x 1152)."""
    from bidiff.model.diffusion import DiscreteMaskingDiffusion
    dev = gpu_device
    cfg = dict(RI.BACKBONE_CFGS["tiny_h2"])
    if size == "medium_width":
        cfg.update(hidden_size=768, num_heads=6, ffn_hidden_size=3072, num_layers=2)
    elif size == "large_width":
        cfg.update(hidden_size=1152, num_heads=9, ffn_hidden_size=4608, num_layers=2)
    torch.manual_seed(0)
    bb = DiffusionBackbone(**cfg)
    model = DiscreteMaskingDiffusion(bb, num_steps=cfg["num_diffusion_steps"], mask_token_id=1, pad_token_id=0,
                                     bos_token_id=2, eos_token_id=3).to(dev).train()
    opt, _ = build_step1_backbone_optimizer(model, RI.MUON_OPT_CFG, 1e-3, 0.01)
    ids, attn, _ = RI.backbone_inputs(cfg, 3, B=16, L=32)
    ids, attn = torch.from_numpy(ids).to(dev), torch.from_numpy(attn).to(dev)
    t = torch.full((16,), 15, device=dev, dtype=torch.long)
    u = torch.rand(16, 32, generator=torch.Generator().manual_seed(2)).to(dev)
    losses = []
    for _ in range(30):
        noisy, ind = model.forward_process(ids, t, attn, uniforms=u)
        loss = model._compute_loss(model.backbone(noisy, t, attn), ids, ind, attn)
        loss.backward()
        opt.step(max_norm=1.0)
        opt.zero_grad()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and losses[-1] < 0.6 * losses[0], losses[::5]
