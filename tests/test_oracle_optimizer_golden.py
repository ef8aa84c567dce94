"""CPU: the oracle's Newton-Schulz / Muon / AdamW restatement against vectors captured from the
reference's MuonAdamW (tests/golden/gen_optimizer_fixtures.py), and the host-side parameter
grouping of the product."""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import replay_inputs as RI
from oracle import optimizer_ref as OR
from bidiff.model.backbone import DiffusionBackbone
from bidiff.training.optimizer_setup import build_muon_param_groups, build_step1_backbone_optimizer

BLOB = np.load(os.path.join(H.GOLDEN, "optimizer.npz"))


@pytest.mark.parametrize("shape", RI.NS_SHAPES)
def test_newton_schulz_matches_reference(shape):
    g = torch.from_numpy(RI.ns_input(shape))
    for steps in (1, 5):
        got = OR.newton_schulz5(g, steps).numpy()
        assert np.allclose(got, BLOB[f"ns/{shape[0]}x{shape[1]}/steps{steps}"], rtol=1e-5, atol=1e-6)
    # five steps push the singular values into a band around 1 (the smallest ones of a square
    # random matrix start near 0 and are still on their way)
    sv = torch.linalg.svdvals(OR.newton_schulz5(g, 5))
    assert sv.max() < 1.3 and sv.median() > 0.6


def reference_split(model):
    groups, info = build_muon_param_groups(model, 0.01, 3e-4, RI.MUON_OPT_CFG["optimizer"]["muon_adamw"])
    return groups, info


def test_parameter_split_matches_reference():
    bb = DiffusionBackbone(**RI.BACKBONE_CFGS["tiny_h2"])
    groups, info = reference_split(bb)
    assert info["using_muon"] and info["num_muon_tensors"] == int(BLOB["info/num_muon_tensors"])
    assert info["num_adamw_tensors"] == int(BLOB["info/num_adamw_tensors"])
    assert abs(info["muon_lr"] - float(BLOB["info/muon_lr"])) < 1e-12
    assert [g["kind"] for g in groups] == ["adamw", "muon"]
    assert all(p.ndim == 2 for p in groups[1]["params"])
    names = {id(p): n for n, p in bb.named_parameters()}
    assert all("layers." in names[id(p)] for p in groups[1]["params"])
    assert not any("embedding" in names[id(p)] or "output_proj" in names[id(p)] for p in groups[1]["params"])
    # frozen parameters are skipped; a model without eligible matrices falls back to AdamW
    for p in bb.layers.parameters():
        p.requires_grad_(False)
    g2, i2 = reference_split(bb)
    assert g2 == [] and i2["reason"] == "no_muon_eligible_parameters"
    with pytest.raises(ValueError):
        build_muon_param_groups(DiffusionBackbone(**RI.BACKBONE_CFGS["tiny_h2"]), 0.01, 3e-4, {"adamw_betas": [0.9]})


def oracle_trajectory(work_dtype=torch.float32):
    cfg = RI.BACKBONE_CFGS["tiny_h2"]
    torch.manual_seed(7)
    bb = DiffusionBackbone(**cfg)
    RI.perturb_state_dict(bb.state_dict(), 11)
    groups, info = reference_split(bb)
    params = {n: p.detach().clone() for n, p in bb.named_parameters()}
    kind = {}
    for g in groups:
        for p in g["params"]:
            kind[[n for n, q in bb.named_parameters() if q is p][0]] = g
    state = {n: {} for n in params}
    snaps = {}
    for step in range(RI.OPT_STEPS):
        grads = {n: torch.from_numpy(v) for n, v in RI.optimizer_grads(bb, step).items()}
        coef = OR.clip_coef(list(grads.values()), 1.0)
        for n, p in params.items():
            g = kind[n]
            grad = grads[n] * coef
            st = state[n]
            if g["kind"] == "adamw":
                if not st:
                    st.update(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
                OR.adamw_update(p, grad, st["exp_avg"], st["exp_avg_sq"], step + 1, g["lr"], g["betas"], g["eps"],
                                g["weight_decay"])
            else:
                if not st:
                    st.update(buf=torch.zeros_like(p))
                OR.muon_update(p, grad, st["buf"], g["lr"], g["weight_decay"], g["momentum"], g["nesterov"],
                               g["backend_steps"], work_dtype)
        snaps[step] = {n: params[n].clone() for n in RI.OPT_WATCH}
    return snaps, state


def test_muon_adamw_trajectory_matches_reference():
    snaps, state = oracle_trajectory()
    for step in (0, RI.OPT_STEPS - 1):
        for n in RI.OPT_WATCH:
            want = BLOB[f"muon_adamw/step{step}/{n}"]
            assert np.allclose(snaps[step][n].numpy(), want, rtol=2e-5, atol=2e-6), (step, n)
    first = [n for n in state if "buf" in state[n]][0]
    assert np.allclose(state[first]["buf"].numpy(), BLOB["state/momentum_buffer0"], rtol=1e-5, atol=1e-7)


def test_optimizer_builder_dispatch_without_device():
    """Config dispatch is host logic; constructing the flat optimizers needs a device and says so."""
    bb = DiffusionBackbone(**RI.BACKBONE_CFGS["tiny_h2"])
    with pytest.raises(RuntimeError):
        build_step1_backbone_optimizer(bb, {}, 3e-4, 0.01)
    with pytest.raises(RuntimeError):
        build_step1_backbone_optimizer(bb, RI.MUON_OPT_CFG, 3e-4, 0.01)
