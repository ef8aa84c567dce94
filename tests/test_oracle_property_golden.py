"""CPU: the oracle's pooling + regression head against vectors captured from the reference's
PropertyPredictor (tests/golden/gen_property_fixtures.py), and the product's PropertyHead
initialisation / checkpoint layout (host logic, no GPU)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import replay_inputs as RI
from oracle import backbone_ref as BR
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.property_head import PropertyHead, PropertyPredictor

BLOB = np.load(os.path.join(H.GOLDEN, "property.npz"))
META = json.load(open(os.path.join(H.GOLDEN, "property_meta.json")))


def sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def setup(name):
    cfg = RI.BACKBONE_CFGS[name]
    torch.manual_seed(7)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 11)
    torch.manual_seed(3)
    head = PropertyHead(cfg["hidden_size"], RI.PROPERTY_HIDDEN, dropout=0.1)
    RI.perturb_state_dict(head.state_dict(), 13)
    L = 24 if name != "tiny_h4" else 150
    ids, attn, t = RI.backbone_inputs(cfg, 21, B=5, L=L)
    return cfg, bb, head, torch.from_numpy(ids), torch.from_numpy(attn), torch.from_numpy(t)


@pytest.mark.parametrize("name", ["tiny_h2", "tiny_h4"])
def test_head_initialisation_reproduces_the_reference(name):
    _, _, head, _, _, _ = setup(name)
    want = META[f"{name}_head_sha"]
    assert set(want) == set(head.state_dict())
    for k, v in head.state_dict().items():
        assert sha(v) == want[k], k


@pytest.mark.parametrize("name", ["tiny_h2", "tiny_h4"])
def test_oracle_property_predictions_match_reference(name):
    cfg, bb, head, ids, attn, t = setup(name)
    sd, hsd = bb.state_dict(), head.state_dict()
    t1 = torch.full((ids.shape[0],), 1, dtype=torch.long)
    labels = torch.from_numpy(BLOB[f"{name}/labels"])
    with torch.no_grad():
        for pooling in ("mean", "max", "cls"):
            pred = BR.property_head(hsd, BR.pooled_output(sd, cfg, ids, t1, attn, pooling))
            assert np.allclose(pred.squeeze(-1).numpy(), BLOB[f"{name}/{pooling}/predict"], rtol=1e-5, atol=1e-6)
            nm = BR.property_head(hsd, BR.pooled_output(sd, cfg, ids, t1, None, pooling))
            assert np.allclose(nm.numpy(), BLOB[f"{name}/{pooling}/forward_nomask"], rtol=1e-5, atol=1e-6)
            ft = BR.property_head(hsd, BR.pooled_output(sd, cfg, ids, t, attn, pooling))
            assert np.allclose(ft.numpy(), BLOB[f"{name}/{pooling}/forward_t"], rtol=1e-5, atol=1e-6)
            loss = torch.nn.functional.mse_loss(pred, labels.unsqueeze(-1))
            assert np.allclose(loss.numpy(), BLOB[f"{name}/{pooling}/loss"], rtol=1e-5)
    # head gradients of the MSE loss on frozen mean-pooled features
    hg = {k: v.clone().requires_grad_(True) for k, v in hsd.items()}
    with torch.no_grad():
        pooled = BR.pooled_output(sd, cfg, ids, t1, attn, "mean")
    torch.nn.functional.mse_loss(BR.property_head(hg, pooled), labels.unsqueeze(-1)).backward()
    for k, v in hg.items():
        assert np.allclose(v.grad.numpy(), BLOB[f"{name}/grad/{k}"], rtol=1e-4, atol=1e-6), k


def test_predictor_checkpoint_layout_and_guards(tmp_path):
    cfg, bb, head, ids, attn, t = setup("tiny_h2")
    pp = PropertyPredictor(bb, head, freeze_backbone=True, pooling="max", default_timestep=1)
    assert not any(p.requires_grad for p in bb.parameters()) and all(p.requires_grad for p in head.parameters())
    path = str(tmp_path / "head.pt")
    pp.save_pretrained(path, include_backbone=True)
    ck = torch.load(path, weights_only=True)
    assert set(ck) == {"property_head_state_dict", "pooling", "default_timestep", "input_size", "hidden_sizes",
                       "backbone_state_dict"}
    assert ck["pooling"] == "max" and ck["hidden_sizes"] == RI.PROPERTY_HIDDEN and ck["input_size"] == cfg["hidden_size"]
    head2 = PropertyHead(cfg["hidden_size"], RI.PROPERTY_HIDDEN)
    pp2 = PropertyPredictor(bb, head2)
    pp2.load_property_head(path)
    for k, v in head.state_dict().items():
        assert torch.equal(v, head2.state_dict()[k])
    # the trainer-style layout with an _orig_mod. prefix
    torch.save({"model_state_dict": {"_orig_mod.property_head." + k: v for k, v in head.state_dict().items()}}, path)
    head3 = PropertyHead(cfg["hidden_size"], RI.PROPERTY_HIDDEN)
    PropertyPredictor(bb, head3).load_property_head(path)
    assert torch.equal(head3.state_dict()["mlp.0.weight"], head.state_dict()["mlp.0.weight"])
    torch.save({"something": 1}, path)
    with pytest.raises(KeyError):
        pp2.load_property_head(path)
    # finetune_last_layers unfreezes exactly what the reference does (property_head.py:106-113)
    ft = PropertyPredictor(bb, head, freeze_backbone=True, finetune_last_layers=1)
    assert [k for k, p_ in ft.named_parameters() if p_.requires_grad] == META["tiny_h2_ft1_trainable"]
    PropertyPredictor(bb, head, freeze_backbone=True)
    assert not any(p_.requires_grad for p_ in bb.parameters())
    # no CPU execution path: the product refuses host tensors instead of falling back
    with pytest.raises(RuntimeError):
        pp.predict(ids, attn)
