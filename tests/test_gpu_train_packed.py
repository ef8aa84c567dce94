"""The training step on packed rows (DiscreteMaskingDiffusion.skip_padded_rows): the rows behind each sequence's
last attended position are left out of every kernel. They receive no attention as keys, carry no loss and get no
gradient in the reference either (diffusion.py:190-207, backbone.py:54-67), so loss and all parameter gradients must
equal the all-rows computation and the fp32 CPU restatement; the returned logits keep their values at the rows that
were computed and are zero elsewhere."""
import numpy as np
import pytest
import torch

from tests.golden import replay_inputs as RI
from oracle import backbone_ref as BR
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
from bidiff.training import autograd as AG
from bidiff.training.optimizer import FlatAdamW

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def setup(dev, B, L, p, heads=2):
    cfg = dict(vocab_size=39, hidden_size=128, num_layers=2, num_heads=heads, ffn_hidden_size=256,
               max_position_embeddings=64, num_diffusion_steps=20, dropout=p, pad_token_id=0)
    torch.manual_seed(7)
    bb = DiffusionBackbone(**cfg)
    RI.perturb_state_dict(bb.state_dict(), 11)
    model = DiscreteMaskingDiffusion(bb, num_steps=20, mask_token_id=1, pad_token_id=0, bos_token_id=2, eos_token_id=3)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    rs = np.random.RandomState(5)
    lens = rs.randint(2, L + 1, size=B)
    lens[0], lens[1], lens[2] = L, 0, 2              # a full row, an empty row, a minimal one
    ids = rs.randint(5, 39, size=(B, L)).astype(np.int64)
    attn = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
    attn[3, 1] = 0                                   # a hole inside a sequence: stays a row, masked as a key
    ids[attn == 0] = 0
    for b in range(B):
        if lens[b] >= 2:
            ids[b, 0], ids[b, lens[b] - 1] = 2, 3
    t = rs.randint(1, 21, size=B).astype(np.int64)
    u = rs.random_sample((B, L)).astype(np.float32)
    return cfg, model.to(dev).train(), sd, torch.from_numpy(ids), torch.from_numpy(attn), torch.from_numpy(t), torch.from_numpy(u), lens


def run(model, dev, ids, attn, t, u, packed, lengths=None, seed=0x1234, step=4):
    bb = model.backbone
    bb.dropout_seed, bb._drop_step = seed, step
    for p_ in model.parameters():
        p_.grad = None
    noisy, ind = model.forward_process(ids.to(dev), t.to(dev), attn.to(dev), uniforms=u.to(dev))
    if packed:
        loss, logits_fn = AG.packed_training_loss(model, noisy, t.to(dev), attn.to(dev), ids.to(dev), ind, lengths)
        logits = logits_fn()
    else:
        logits = bb(noisy, t.to(dev), attn.to(dev))
        loss = model._compute_loss(logits, ids.to(dev), ind, attn.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    return noisy, ind, loss.item(), logits.detach(), {k: p_.grad.detach().clone() for k, p_ in model.named_parameters()}


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_packed_rows_match_all_rows_and_oracle(gpu_device, p):
    dev = gpu_device
    B, L = 12, 40
    cfg, model, sd, ids, attn, t, u, lens = setup(dev, B, L, p)
    noisy, ind, loss_p, logits_p, g_p = run(model, dev, ids, attn, t, u, packed=True)
    # layout the run used: lengths computed from the mask (row 3 keeps its hole as a row)
    want_len = np.array([0 if not a.any() else int(np.nonzero(a)[0][-1]) + 1 for a in attn.numpy()])
    assert np.array_equal(AG.RowPacking.lengths_from_mask(attn.to(dev)), want_len)
    pk = AG.RowPacking(want_len, B, L, dev)
    assert pk.n_real == int(want_len.sum()) and pk.n_rows % 32 == 0 and pk.n_rows >= pk.n_real
    rowmap = pk.rowmap.cpu().numpy()
    kept = np.zeros(B * L, dtype=bool)
    kept[rowmap[rowmap >= 0]] = True
    assert kept.sum() == pk.n_real and np.array_equal(kept.reshape(B, L), np.arange(L)[None] < want_len[:, None])
    # the lengths hint gives the same logits bit for bit (the loss is a sum of atomic adds: equal to rounding)
    _, _, loss_h, logits_h, g_h = run(model, dev, ids, attn, t, u, packed=True, lengths=want_len.tolist())
    assert abs(loss_h - loss_p) <= 1e-6 * abs(loss_p) and torch.equal(logits_h, logits_p)
    # fp32 restatement (with dropout: the same decision stream, token-row sites counted in packed order)
    stream = BR.DropoutStream(p, 0x1234, 4, rowmap=rowmap) if p > 0 else None
    ref_sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and k != "mask_schedule") for k, v in sd.items()}
    rl = BR.logits(ref_sd, cfg, noisy.cpu(), t, attn, prefix="backbone.", dropout=stream)
    rloss = BR.masked_ce(rl, ids, ind.cpu(), attn)
    rloss.backward()
    assert abs(loss_p - rloss.item()) < 2e-2 * abs(rloss.item())
    k2 = torch.from_numpy(kept.reshape(B, L))
    err = (logits_p.cpu()[k2] - rl.detach()[k2]).abs().max().item() / rl.detach()[k2].abs().max().item()
    assert err < 2e-2, err
    assert logits_p.cpu()[~k2].abs().max().item() == 0.0
    for k, g in g_p.items():
        if k.endswith("k_proj.bias"):
            continue
        e = rel(g.cpu(), ref_sd[k].grad)
        cos = torch.nn.functional.cosine_similarity(g.cpu().flatten(), ref_sd[k].grad.flatten(), dim=0).item()
        assert e < 1e-1 and cos > 0.995, (k, e, cos)
    if p == 0.0:
        # against the all-rows computation on the device: same arithmetic per row, sums in another order
        _, _, loss_u, logits_u, g_u = run(model, dev, ids, attn, t, u, packed=False)
        assert abs(loss_u - loss_p) < 1e-5 * abs(loss_u)
        assert (logits_u.cpu()[k2] - logits_p.cpu()[k2]).abs().max().item() < 1e-4 * logits_u.abs().max().item()
        for k in g_p:
            assert rel(g_p[k], g_u[k]) < 2e-2 or g_u[k].abs().max().item() < 1e-10, (k, rel(g_p[k], g_u[k]))


def test_packed_training_steps_and_module_forward(gpu_device):
    """model(ids, mask) takes the packed path by default; a few optimizer steps lower the loss, the flag turns it
    off, and batches whose sequences are all full length or all empty are handled."""
    dev = gpu_device
    B, L = 16, 32
    cfg, model, sd, ids, attn, t, u, lens = setup(dev, B, L, 0.1)
    opt = FlatAdamW(model.parameters(), lr=2e-3)
    losses = []
    torch.manual_seed(0)
    for _ in range(8):
        out = model(ids.to(dev), attn.to(dev))
        assert out["logits"].shape == (B, L, 39) and not out["logits"].requires_grad
        out["loss"].backward()
        opt.step(max_norm=1.0)
        opt.zero_grad()
        losses.append(out["loss"].item())
    assert losses[-1] < losses[0] and all(np.isfinite(losses))
    model.skip_padded_rows = False
    out = model(ids.to(dev), attn.to(dev))
    assert out["logits"].requires_grad
    model.skip_padded_rows = True
    full = torch.ones_like(attn)
    out = model(ids.clamp(min=5).to(dev), full.to(dev), lengths=[L] * B)
    out["loss"].backward()
    assert np.isfinite(out["loss"].item())
    none = torch.zeros_like(attn)
    out = model(ids.to(dev), none.to(dev))
    out["loss"].backward()
    assert out["loss"].item() == 0.0
    with pytest.raises(ValueError):
        model(ids.to(dev), attn.to(dev), lengths=[L + 1] * B)


def test_forward_outputs_build_the_padded_logits_on_demand(gpu_device):
    """model(ids, attn) on the packed path returns the reference's dict (loss, logits, noisy_ids, mask_indicator,
    timesteps); the [B, L, V] logits are written only when somebody looks at them, and every way of looking does."""
    dev = gpu_device
    B, L = 12, 40
    cfg, model, sd, ids, attn, t, u, lens = setup(dev, B, L, 0.0)
    torch.manual_seed(3)
    out = model(ids.to(dev), attn.to(dev))
    assert "logits" in out and len(out) == 5
    assert not dict.__contains__(out, "logits")          # not built yet
    out["loss"].backward()
    assert not dict.__contains__(out, "logits")
    lg = out["logits"]
    assert lg.shape == (B, L, cfg["vocab_size"]) and not lg.requires_grad
    assert out.get("logits") is lg and out.get("nothing", 7) == 7
    torch.manual_seed(3)
    out2 = model(ids.to(dev), attn.to(dev))
    assert set(out2.keys()) == {"loss", "logits", "noisy_ids", "mask_indicator", "timesteps"}
    assert torch.equal(out2["logits"], lg) and torch.equal(out2["noisy_ids"], out["noisy_ids"])
    # rows behind a sequence's last attended position are zeros, the others are the denoiser's logits
    model.skip_padded_rows = False
    torch.manual_seed(3)
    ref = model(ids.to(dev), attn.to(dev))
    model.skip_padded_rows = True
    assert isinstance(ref["logits"], torch.Tensor)
    assert abs(float(ref["loss"].detach()) - float(out["loss"].detach())) < 2e-3
    for b in range(B):
        assert torch.all(lg[b, lens[b]:] == 0)
        if lens[b]:
            assert (lg[b, :lens[b]] - ref["logits"][b, :lens[b]]).abs().max() < 6e-2
