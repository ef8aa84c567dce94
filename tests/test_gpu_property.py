"""GPU parity of the pooling kernel and of PropertyPredictor (backbone forward + pooling + head)
against the reference's golden vectors and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import replay_inputs as RI
from oracle import backbone_ref as BR
from bidiff import _lib
from bidiff.model.property_head import PropertyPredictor
from tests.test_oracle_property_golden import BLOB, setup

pytestmark = pytest.mark.gpu
BF16_TOL = 2e-2


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12))


@pytest.mark.parametrize("B,L,d", [(3, 24, 128), (5, 150, 384), (2, 7, 1024), (4, 128, 640)])
def test_pool_kernel_matches_torch(gpu_device, B, L, d):
    """The pooling kernel alone on given bf16 hidden states: exact for cls, fp32-sum accurate for
    mean, exact for max (a maximum of exactly representable products)."""
    dev = gpu_device
    g = torch.Generator().manual_seed(B * L + d)
    h = torch.randn(B, L, d, generator=g).bfloat16()
    lens = [L, max(1, L // 3), 1, L - 1, 0][:B]
    m = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate(lens):
        m[b, :n] = 1
    hf = h.float()
    mf = m.unsqueeze(-1).float()
    want = {0: (hf * mf).sum(1) / mf.sum(1).clamp(min=1e-9), 1: hf[:, 0], 2: (hf * mf + (1 - mf) * (-1e9)).max(1)[0]}
    want_nomask = {0: hf.mean(1), 1: hf[:, 0], 2: hf.max(1)[0]}
    hd, md = h.to(dev), m.to(dev)
    for mode in (0, 1, 2):
        out = torch.full((B, d), float("nan"), device=dev)
        _lib.check(_lib.lib().bd_pool_hidden(_lib.ptr(hd), _lib.ptr(md), _lib.ptr(out), B, L, d, mode,
                                             _lib.stream_handle()), "pool")
        if mode == 0:
            assert torch.allclose(out.cpu(), want[0], rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(out.cpu(), want[mode]), mode
        _lib.check(_lib.lib().bd_pool_hidden(_lib.ptr(hd), None, _lib.ptr(out), B, L, d, mode, _lib.stream_handle()), "pool")
        if mode == 0:
            assert torch.allclose(out.cpu(), want_nomask[0], rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(out.cpu(), want_nomask[mode]), mode
    assert _lib.lib().bd_pool_hidden(_lib.ptr(hd), None, _lib.ptr(out), B, L, d, 3, _lib.stream_handle()) != 0
    assert _lib.lib().bd_pool_hidden(_lib.ptr(hd), None, _lib.ptr(out), B, L, 12, 0, _lib.stream_handle()) != 0


@pytest.mark.parametrize("name", ["tiny_h2", "tiny_h4"])
def test_property_predictor_matches_reference_golden(gpu_device, name):
    dev = gpu_device
    cfg, bb, head, ids, attn, t = setup(name)
    bb, head = bb.to(dev), head.to(dev)
    ti, ta, tt = ids.to(dev), attn.to(dev), t.to(dev)
    labels = torch.from_numpy(BLOB[f"{name}/labels"]).to(dev)
    for pooling in ("mean", "max", "cls"):
        pp = PropertyPredictor(bb, head, freeze_backbone=True, pooling=pooling, default_timestep=1)
        pred = pp.predict(ti, ta)
        assert pred.shape == (5,) and not pp.training
        assert rel(pred.cpu().numpy(), BLOB[f"{name}/{pooling}/predict"]) < BF16_TOL
        with torch.no_grad():
            assert rel(pp(ti, None).cpu().numpy(), BLOB[f"{name}/{pooling}/forward_nomask"]) < BF16_TOL
            assert rel(pp(ti, ta, tt).cpu().numpy(), BLOB[f"{name}/{pooling}/forward_t"]) < BF16_TOL
        out = pp.compute_loss(ti, labels, ta)
        assert abs(out["loss"].item() - float(BLOB[f"{name}/{pooling}/loss"])) < 5e-2 * float(BLOB[f"{name}/{pooling}/loss"])
    # head training on the frozen backbone: gradients of the MSE loss (eval mode = dropout off)
    pp = PropertyPredictor(bb, head, freeze_backbone=True, pooling="mean", default_timestep=1).eval()
    pp.zero_grad()
    pp.compute_loss(ti, labels, ta)["loss"].backward()
    for k, p in head.named_parameters():
        assert p.grad is not None
        assert rel(p.grad.cpu().numpy(), BLOB[f"{name}/grad/{k}"]) < 5e-2, k
    assert all(p.grad is None for p in bb.parameters())


def test_pooled_output_at_bench_shape_matches_oracle_and_guards(gpu_device):
    """d = 384 small backbone through the single-launch forward + pooling kernel vs the fp32 oracle;
    an unfrozen backbone under autograd takes the training path."""
    dev = gpu_device
    from bidiff.model.backbone import DiffusionBackbone
    cfg = dict(RI.SMALL_CFG, num_layers=2, dropout=0.0)
    torch.manual_seed(5)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 11)
    sd = {k: v.clone() for k, v in bb.state_dict().items()}
    ids, attn, t = RI.backbone_inputs(cfg, 9, B=6, L=128)
    ids, attn, t = torch.from_numpy(ids), torch.from_numpy(attn), torch.from_numpy(t)
    bb = bb.to(dev)
    for pooling in ("mean", "max", "cls"):
        with torch.no_grad():
            got = bb.get_pooled_output(ids.to(dev), t.to(dev), attn.to(dev), pooling).cpu()
            want = BR.pooled_output(sd, cfg, ids, t, attn, pooling)
        assert rel(got.numpy(), want.numpy()) < BF16_TOL, pooling
    # an unfrozen backbone under autograd: the training path, pooled by the reference's torch expression
    bb.train()
    got = bb.get_pooled_output(ids.to(dev), t.to(dev), attn.to(dev), "mean")
    assert got.requires_grad and rel(got.detach().cpu().numpy(), BR.pooled_output(sd, cfg, ids, t, attn, "mean").numpy()) < BF16_TOL
    with pytest.raises(ValueError):
        with torch.no_grad():
            bb.get_pooled_output(ids.to(dev), t.to(dev), attn.to(dev), "median")


def test_pooled_output_of_ragged_batches_skips_padded_blocks_with_the_same_bits(gpu_device):
    """Under a mask the pooling reads attended rows only, so the forward computes just the 32-row blocks each
    sequence needs (short sequences share workgroups): same pooled values, bit for bit, as with every padded row."""
    from bidiff.model.backbone import DiffusionBackbone
    cfg = dict(RI.SMALL_CFG, num_layers=2, dropout=0.0)
    torch.manual_seed(6)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 12)
    bb = bb.to(gpu_device)
    rs = np.random.RandomState(3)
    B, L = 50, 120
    lens = rs.randint(1, L + 1, size=B)
    lens[0] = L
    attn = torch.from_numpy((np.arange(L)[None] < lens[:, None]).astype(np.int64)).to(gpu_device)
    ids = torch.from_numpy(rs.randint(5, cfg["vocab_size"], size=(B, L)).astype(np.int64)).to(gpu_device) * attn
    t = torch.from_numpy(rs.randint(1, cfg["num_diffusion_steps"] + 1, size=B).astype(np.int64)).to(gpu_device)
    assert bb.wave_table_for_mask(attn) is not None
    for pooling in ("mean", "max", "cls"):
        with torch.no_grad():
            bb.ragged_pooling = True
            a = bb.get_pooled_output(ids, t, attn, pooling)
            bb.ragged_pooling = False
            b = bb.get_pooled_output(ids, t, attn, pooling)
        assert torch.isfinite(a).all() and torch.equal(a, b), pooling
    bb.ragged_pooling = True


@pytest.mark.parametrize("name,n_ft", [("tiny_h2", 1), ("tiny_h2", 2), ("tiny_h4", 1)])
def test_finetune_last_layers_gradients_match_reference(gpu_device, name, n_ft):
    """PropertyPredictor(freeze_backbone=True, finetune_last_layers=N): the reference unfreezes the last N blocks
    and the final norm (property_head.py:102-112). Same set of trainable parameters, and the gradients of the MSE
    loss for all of them (backbone blocks through the training-step backward kernels, which stop below the lowest
    unfrozen block) against the vectors the reference's autograd produced."""
    import json
    dev = gpu_device
    meta = json.load(open(os.path.join(H.GOLDEN, "property_meta.json")))
    cfg, bb, head, ids, attn, t = setup(name)
    bb, head = bb.to(dev), head.to(dev)
    ti, ta = ids.to(dev), attn.to(dev)
    labels = torch.from_numpy(BLOB[f"{name}/labels"]).to(dev)
    sd_cpu = {"backbone." + k: v.detach().cpu().clone() for k, v in bb.state_dict().items()}
    hsd_cpu = {k: v.detach().cpu().clone() for k, v in head.state_dict().items()}
    t1 = torch.full((ids.shape[0],), 1, dtype=torch.long)
    for pooling in ("mean", "max"):
        # yardstick: how far the CPU restatement under bf16 autocast (the reference's training precision) lands from
        # the reference's fp32 gradients on these five short sequences
        ac = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd_cpu.items()}
        ach = {k: v.clone().requires_grad_(True) for k, v in hsd_cpu.items()}
        with torch.autocast("cpu", dtype=torch.bfloat16):
            pred = BR.property_head(ach, BR.pooled_output(ac, cfg, ids, t1, attn, pooling, prefix="backbone."))
        torch.nn.functional.mse_loss(pred.float(), torch.from_numpy(BLOB[f"{name}/labels"]).unsqueeze(-1)).backward()
        pp = PropertyPredictor(bb, head, freeze_backbone=True, finetune_last_layers=n_ft, pooling=pooling,
                               default_timestep=1).eval()
        names = [k for k, p in pp.named_parameters() if p.requires_grad]
        assert names == meta[f"{name}_ft{n_ft}_trainable"]
        pp.zero_grad()
        out = pp.compute_loss(ti, labels, ta)
        out["loss"].backward()
        for k, p in pp.named_parameters():
            if not p.requires_grad:
                assert p.grad is None, k
                continue
            want = BLOB[f"{name}/ft{n_ft}/{pooling}/{k}"]
            if k.endswith("k_proj.bias"):      # true gradient is zero (softmax shift invariance)
                assert np.linalg.norm(p.grad.cpu().numpy()) < 0.1 * np.linalg.norm(BLOB[f"{name}/ft{n_ft}/{pooling}/{k.replace('k_proj', 'q_proj')}"]) + 1e-9
                continue
            e = rel(p.grad.cpu().numpy(), want)
            g_ac = (ac[k] if k.startswith("backbone.") else ach[k[len("property_head."):]]).grad
            e_ac = rel(g_ac.float().numpy(), want)
            # the tiny-shape bar of the training-step tests (1e-1: five short sequences, and max pooling routes one
            # row per feature, so little averages out; the full-size training-step test holds 2e-2), or 1.5 x what
            # bf16 autocast itself loses on the tensor
            assert e < max(1e-1, 1.5 * e_ac), (k, pooling, e, e_ac)
    # one optimizer step moves the unfrozen blocks and the head, nothing else
    before = {k: p.detach().clone() for k, p in pp.named_parameters()}
    opt = torch.optim.AdamW([p for p in pp.parameters() if p.requires_grad], lr=1e-3)
    opt.zero_grad()
    pp.compute_loss(ti, labels, ta)["loss"].backward()
    opt.step()
    for k, p in pp.named_parameters():
        assert torch.equal(before[k], p.detach()) != p.requires_grad or k.endswith("k_proj.bias"), k
