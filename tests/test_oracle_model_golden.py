"""CPU restatements (oracle/) and the host-side module logic against golden vectors captured
from the imported reference (tests/golden/gen_model_fixtures.py)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import replay_inputs as RI
from oracle import backbone_ref as BR
from oracle import sampler_ref as ora
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion

BLOB = np.load(os.path.join(H.GOLDEN, "model.npz"))
META = json.load(open(os.path.join(H.GOLDEN, "model_meta.json")))


def _sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def test_seeded_init_matches_reference_bit_for_bit():
    """Same module registration order and init distributions: torch.manual_seed(42) reproduces the
    reference's 103 tensors (names, shapes and bytes)."""
    torch.manual_seed(42)
    bb = DiffusionBackbone(**RI.SMALL_CFG)
    model = DiscreteMaskingDiffusion(bb, num_steps=100, mask_token_id=1, pad_token_id=0, bos_token_id=2,
                                     eos_token_id=3)
    sd = model.state_dict()
    want = META["init_small_seed42"]
    assert list(sd.keys()) == sorted(want.keys(), key=list(sd.keys()).index)
    assert set(sd.keys()) == set(want.keys())
    assert len(sd) == 103
    for k, v in sd.items():
        assert list(v.shape) == want[k]["shape"], k
        assert _sha(v) == want[k]["sha256"], k
    assert sum(p.numel() for p in bb.parameters()) == META["n_params_small"] == 10814592


def tiny_model(name):
    cfg = RI.BACKBONE_CFGS[name]
    torch.manual_seed(7)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 11)
    return cfg, bb


@pytest.mark.parametrize("name", list(RI.BACKBONE_CFGS))
def test_oracle_backbone_matches_reference_logits(name):
    cfg, bb = tiny_model(name)
    sd = bb.state_dict()
    for k, v in sd.items():
        assert _sha(v) == META[f"{name}_weights_sha"][k], k
    L = 24 if name != "tiny_h4" else 150
    ids, attn, t = RI.backbone_inputs(cfg, 21, B=5, L=L)
    ti, ta, tt = torch.from_numpy(ids), torch.from_numpy(attn), torch.from_numpy(t)
    with torch.no_grad():
        got = BR.logits(sd, cfg, ti, tt, ta).numpy()
        got_nm = BR.logits(sd, cfg, ti, tt, None).numpy()
        got_2d = BR.logits(sd, cfg, ti, tt.unsqueeze(-1), ta).numpy()
    for g, key in ((got, "logits_masked"), (got_nm, "logits_nomask"), (got_2d, "logits_t2d")):
        ref = BLOB[f"{name}/{key}"]
        assert np.isfinite(g).all()
        err = np.abs(g - ref).max() / np.abs(ref).max()
        assert err < 1e-5, (key, err)
    if name == "tiny_h2":
        with torch.no_grad():
            hs = BR.hidden_states(sd, cfg, ti, tt, ta).numpy()
        ref = BLOB[f"{name}/hidden_masked"]
        assert np.abs(hs - ref).max() / np.abs(ref).max() < 1e-5


def test_mask_schedule_bit_exact():
    for T in (50, 100):
        for clean in (False, True):
            ref = BLOB[f"schedule/T{T}_clean{int(clean)}"]
            o = BR.mask_schedule(T, force_clean_t0=clean).numpy()
            m = DiscreteMaskingDiffusion(torch.nn.Identity(), num_steps=T, force_clean_t0=clean)
            assert np.array_equal(o.view(np.uint32), ref.view(np.uint32))
            assert np.array_equal(m.mask_schedule.numpy().view(np.uint32), ref.view(np.uint32))
            assert "mask_schedule" in m.state_dict()


@pytest.mark.parametrize("tag,bos,eos,use_attn", [("a", 2, 3, True), ("b", 2, 3, False), ("c", None, None, True)])
def test_oracle_forward_process_bit_exact(tag, bos, eos, use_attn):
    ids, attn, t = BLOB["fp/ids"], BLOB["fp/attn"], BLOB["fp/t"]
    sched = BLOB["schedule/T50_clean0"]
    noisy, ind = ora.corrupt(ids, attn if use_attn else None, t, sched, BLOB[f"fp/{tag}_u"], 1, 0, bos, eos)
    np.testing.assert_array_equal(noisy, BLOB[f"fp/{tag}_noisy"])
    np.testing.assert_array_equal(ind, BLOB[f"fp/{tag}_ind"])


def test_oracle_masked_ce_and_gradient():
    lg = torch.from_numpy(BLOB["ce/logits"]).requires_grad_(True)
    tg, mi, am = (torch.from_numpy(BLOB[k]) for k in ("ce/targets", "ce/ind", "ce/attn"))
    loss = BR.masked_ce(lg, tg, mi, am)
    loss.backward()
    assert abs(loss.item() - float(BLOB["ce/loss"])) < 1e-6
    assert np.abs(lg.grad.numpy() - BLOB["ce/dlogits"]).max() < 1e-7
    zero = BR.masked_ce(lg.detach(), tg, torch.zeros_like(mi), am)
    assert zero.item() == 0.0 == float(BLOB["ce/loss_zero"])
    assert float(BLOB["ce/dlogits_zero_absmax"]) == 0.0
    assert abs(BR.masked_ce(lg.detach(), tg, mi, None).item() - float(BLOB["ce/loss_noattn"])) < 1e-6


def test_oracle_training_forward_replays_reference_rng_order():
    """t = randint(1, T+1) first, then the corruption uniforms (diffusion.py:150, :107)."""
    cfg, bb = tiny_model("tiny_h2")
    sd = bb.state_dict()
    ids, attn, _ = RI.backbone_inputs(cfg, 33, B=5, L=24)
    torch.manual_seed(77)
    t = torch.randint(1, 21, (5,))
    u = torch.rand(5, 24)
    np.testing.assert_array_equal(t.numpy(), BLOB["fwd/timesteps"])
    noisy, ind = ora.corrupt(ids, attn, t.numpy(), BR.mask_schedule(20).numpy(), u.numpy(), 1, 0, 2, 3)
    np.testing.assert_array_equal(noisy, BLOB["fwd/noisy_ids"])
    np.testing.assert_array_equal(ind, BLOB["fwd/mask_indicator"])
    with torch.no_grad():
        lg = BR.logits(sd, cfg, torch.from_numpy(noisy), t, torch.from_numpy(attn))
        loss = BR.masked_ce(lg, torch.from_numpy(ids), torch.from_numpy(ind), torch.from_numpy(attn))
    assert np.abs(lg.numpy() - BLOB["fwd/logits"]).max() / np.abs(BLOB["fwd/logits"]).max() < 1e-5
    assert abs(loss.item() - float(BLOB["fwd/loss"])) < 1e-5


def test_forward_result_dict_builds_deferred_entries_on_first_access():
    """The training forward returns the reference's dict; entries that cost a pass over memory are built when first
    read, and every way of reading a dict sees them."""
    from bidiff.model.diffusion import _Outputs
    calls = []

    def build():
        calls.append(1)
        return "L"
    out = _Outputs({"loss": 1.0, "timesteps": 2}, {"logits": build})
    assert "logits" in out and "loss" in out and "nope" not in out and len(out) == 3
    assert out["loss"] == 1.0 and calls == []
    assert out.get("nope") is None and out.get("nope", 5) == 5 and calls == []
    assert out["logits"] == "L" and out["logits"] == "L" and out.get("logits") == "L" and calls == [1]
    out2 = _Outputs({"loss": 1.0}, {"logits": build})
    assert sorted(out2.keys()) == ["logits", "loss"] and dict(out2.items())["logits"] == "L" and calls == [1, 1]
    out3 = _Outputs({"loss": 1.0}, {"logits": build})
    assert sorted(k for k in out3) == ["logits", "loss"] and "L" in out3.values()
    with pytest.raises(KeyError):
        out3["missing"]
    assert _Outputs({"loss": 1.0}, {"logits": build}) == {"loss": 1.0, "logits": "L"}
    assert dict(_Outputs({"loss": 1.0}, {"logits": build})) == {"loss": 1.0, "logits": "L"}


@pytest.mark.parametrize("V", [39, 803])
def test_oracle_at_the_benchmark_architecture(V):
    """d 384, 6 layers, 3 heads of 128, L = 128 (SMILES V = 39, SMILES_BPE V = 803): the restatement's logits and its
    replay of the training forward against vectors from the reference at that very size (model_small.npz)."""
    blob = np.load(os.path.join(H.GOLDEN, "model_small.npz"))
    cfg, ids, attn, t, stride = RI.small_arch_case(V)
    torch.manual_seed(42)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 5)
    sd = bb.state_dict()
    for k, a in (("ids", ids), ("attn", attn), ("t", t)):
        np.testing.assert_array_equal(a, blob[f"v{V}/{k}"])
    want = blob[f"v{V}/logits"]
    with torch.no_grad():
        lg = BR.logits(sd, cfg, torch.from_numpy(ids), torch.from_numpy(t), torch.from_numpy(attn)).numpy()[:, ::stride]
    assert np.abs(lg - want).max() / np.abs(want).max() < 1e-5
    # training forward: t = randint first, then the corruption uniforms
    ids2 = blob[f"v{V}/fwd_ids"]
    torch.manual_seed(78)
    tt = torch.randint(1, 101, (2,))
    u = torch.rand(2, 128)
    np.testing.assert_array_equal(tt.numpy(), blob[f"v{V}/fwd_timesteps"])
    noisy, ind = ora.corrupt(ids2, attn, tt.numpy(), BR.mask_schedule(100).numpy(), u.numpy(), 1, 0, 2, 3)
    np.testing.assert_array_equal(noisy, blob[f"v{V}/fwd_noisy_ids"])
    np.testing.assert_array_equal(ind, blob[f"v{V}/fwd_mask_indicator"])
    with torch.no_grad():
        lg = BR.logits(sd, cfg, torch.from_numpy(noisy), tt, torch.from_numpy(attn))
        loss = BR.masked_ce(lg, torch.from_numpy(ids2), torch.from_numpy(ind), torch.from_numpy(attn))
    want = blob[f"v{V}/fwd_logits"]
    assert np.abs(lg.numpy()[:, ::stride] - want).max() / np.abs(want).max() < 1e-5
    assert abs(loss.item() - float(blob[f"v{V}/fwd_loss"])) < 1e-5
