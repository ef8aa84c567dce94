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
    an unfrozen backbone under autograd is refused (no backward through the pooled features)."""
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
    with pytest.raises(NotImplementedError):
        bb.get_pooled_output(ids.to(dev), t.to(dev), attn.to(dev), "mean")
    with pytest.raises(ValueError):
        with torch.no_grad():
            bb.get_pooled_output(ids.to(dev), t.to(dev), attn.to(dev), "median")
