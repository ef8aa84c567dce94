"""End-to-end GPU run of the product samplers (denoiser + reverse loop + repairs) with every
step recorded and replayed through the CPU oracle; plus output invariants at full batch size."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import replay_inputs as RI
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
from bidiff.sampling import core
from oracle import sampler_ref as ora

pytestmark = pytest.mark.gpu


def build_model(tok, device, T, heads=2, layers=2, d=128, F=256):
    torch.manual_seed(3)
    bb = DiffusionBackbone(tok.vocab_size, d, layers, heads, F, 512, T, 0.0, tok.pad_token_id)
    RI.perturb_state_dict(bb.state_dict(), 4)
    with torch.no_grad():
        bb.output_proj.weight.mul_(6.0)   # spread the logits so every constraint class gets drawn
    return DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=tok.mask_token_id, pad_token_id=tok.pad_token_id,
                                    bos_token_id=tok.bos_token_id, eos_token_id=tok.eos_token_id).to(device).eval()


def replay(records, tb, T, L, temperature, use_constraints, fixed, natural=False, extra=0):
    ktab = core.unmask_count_table(T, L)
    last = None
    for r in records:
        t = r["t"]
        has_eos = None if r["has_eos_before"] is None else r["has_eos_before"].numpy()
        o = ora.sample_step(tb, r["ids_before"].numpy(), r["raw_logits"].numpy(), ktab[t], t, temperature=temperature,
                            use_constraints=use_constraints, fixed=fixed, seed=r["seed"], seq_offset=r["seq_offset"],
                            natural_eos=natural, attn=r["attn_before"].numpy() if natural else None,
                            has_eos=has_eos if (natural or has_eos is not None) else None, variant_extra=extra)
        np.testing.assert_array_equal(o["ids"], r["ids_after"].numpy(), err_msg=f"step t={t} ({r['mode']})")
        last = (o, r)
    return last


@pytest.mark.parametrize("variant,kw", [("smiles", {}), ("smiles", {"lengths": True}), ("selfies", {}),
                                        ("selfies", {"natural": True}), ("group", {}),
                                        ("bpe", {}), ("bpe", {"lengths": True})])   # V = 803: chosen-row workspace
def test_sampler_end_to_end_replayed_by_oracle(gpu_device, variant, kw):
    T, B, L = 16, 24, 40
    smp0, tok = H.make_sampler(variant)
    model = build_model(tok, gpu_device, T)
    smp, _ = H.make_sampler(variant, model=model, num_steps=T, temperature=0.9, use_constraints=True,
                            device=str(gpu_device))
    smp._core.seed = 424242
    smp._core.record = []
    tb = H.oracle_tables(smp)
    if kw.get("lengths"):
        lengths = [int(x) for x in np.random.RandomState(1).randint(2, L + 1, size=B)]
        lengths[0] = L
        ids, strings = smp.sample_with_lengths(lengths, max_length=L, show_progress=False, seq_offset=7)
    elif kw.get("natural"):
        ids, strings = smp.sample(B, L, show_progress=False, allow_natural_eos=True, seq_offset=7)
    else:
        ids, strings = smp.sample(B, L, show_progress=False, seq_offset=7)
    rec = smp._core.record
    assert len(rec) == T and rec[-1]["mode"] == "logits" and rec[0]["mode"] == "hidden"
    fixed = (rec[0]["ids_before"].numpy() != tb.mask_id).astype(np.uint8) if variant in ("smiles", "bpe") else None
    natural = bool(kw.get("natural"))
    assert (smp._core.row_workspace is None) and ((tok.vocab_size > 128) == (variant in ("group", "bpe")))
    o, r = replay(rec, tb, T, rec[0]["ids_before"].shape[1], 0.9, True, fixed, natural)
    # post-loop repairs on the oracle side
    of = ora.sample_fixup(tb, o["ids"], o["logits"], natural_eos=natural, attn=o["attn"], has_eos=o["has_eos"])
    np.testing.assert_array_equal(ids.cpu().numpy(), of["ids"])
    out = ids.cpu().numpy()
    assert ids.dtype == torch.long and ids.is_cuda and len(strings) == B
    assert strings == tok.batch_decode(out.tolist(), skip_special_tokens=True)      # decoded on the device
    assert (out != tok.mask_token_id).all()
    cid = tb.count_id
    if variant != "group" and not kw.get("lengths") and not natural:   # very short rows cannot hold two
        assert ((out == cid).sum(1) == 2).all(), "exactly two stars / placeholders after the repairs"


@pytest.mark.parametrize("variant,natural,maskfix", [("smiles", False, False), ("selfies", False, False),
                                                     ("selfies", True, False), ("selfies", False, True),
                                                     ("group", False, False), ("group", False, True)])
def test_sample_conditional_replayed_by_oracle(gpu_device, variant, natural, maskfix):
    """sample_conditional of the three sampler classes (SMILES sampler.py:913-960, SELFIES :553-737,
    Group :593-696): prefix / suffix stay, every step replays bit-exactly through the oracle with the
    flavour's variant bits, and the repaired ids match."""
    T, B, L = 12, 16, 36
    smp0, tok = H.make_sampler(variant)
    model = build_model(tok, gpu_device, T)
    smp, _ = H.make_sampler(variant, model=model, num_steps=T, temperature=1.1, use_constraints=True,
                            device=str(gpu_device))
    smp._core.seed = 99
    smp._core.record = []
    tb = H.oracle_tables(smp)
    v = tok.vocab
    counted = {"smiles": "*", "selfies": "[I+3]", "group": "[IH0+3]"}[variant]
    plain = {"smiles": "C", "selfies": "[C]", "group": "[C]"}[variant]
    pre = [v[plain], v[counted]] + ([v["[MASK]"]] if maskfix else [])
    suf = None if natural else ([v["[MASK]"], v[plain]] if maskfix else [v[plain]])
    kw = {"allow_natural_eos": True} if natural else {}
    ids, strings = smp.sample_conditional(
        B, L, prefix_ids=torch.tensor(pre).repeat(B, 1),
        suffix_ids=None if suf is None else torch.tensor(suf).repeat(B, 1), show_progress=False, **kw)
    rec = smp._core.record
    assert len(rec) == T
    fixed = rec[0]["fixed"].numpy()
    extra = rec[0]["variant_extra"]
    assert extra == RI.CONDITIONAL_EXTRA[variant]
    o, r = replay(rec, tb, T, L, 1.1, True, fixed, natural, extra)
    of = ora.sample_fixup(tb, o["ids"], o["logits"], natural_eos=natural, attn=o["attn"], has_eos=o["has_eos"])
    out = ids.cpu().numpy()
    np.testing.assert_array_equal(out, of["ids"])
    # what was fixed stays unless a post-loop repair rewrote it (the reference's repairs ignore fixed_mask)
    before_fix = rec[-1]["ids_after"].numpy()
    start = rec[0]["ids_before"].numpy()
    assert (before_fix[fixed != 0] == start[fixed != 0]).all()
    assert (before_fix[fixed == 0] != tok.mask_token_id).all()
    assert len(strings) == B


@pytest.mark.parametrize("variant,natural", [("smiles", False), ("selfies", False), ("selfies", True), ("group", False)])
def test_sampling_is_independent_of_batch_split(gpu_device, variant, natural):
    """Sequences are independent units keyed by their global index: one batch of 32 equals two
    batches of 16 with offsets 0 and 16 (what a multi-process shard does), for every sampler class
    and with sampled EOS."""
    T, L = 8, 32
    smp0, tok = H.make_sampler(variant)
    model = build_model(tok, gpu_device, T)
    kw = {"allow_natural_eos": True} if natural else {}
    outs = []
    for split in (1, 2):
        smp, _ = H.make_sampler(variant, model=model, num_steps=T, device=str(gpu_device))
        smp._core.seed = 77
        n = 32 // split
        parts = [smp.sample(n, L, show_progress=False, seq_offset=i * n, **kw)[0] for i in range(split)]
        outs.append(torch.cat(parts).cpu())
    assert torch.equal(outs[0], outs[1])


def test_group_variable_length_and_rejection(gpu_device):
    """Group sampler methods the reference's step2 script calls (sample_variable_length, Group sampler.py:466-513)
    and the rejection wrapper (:515-591) with the validity callback on the host."""
    T = 6
    smp0, tok = H.make_sampler("group")
    model = build_model(tok, gpu_device, T)
    smp, _ = H.make_sampler("group", model=model, num_steps=T, device=str(gpu_device))
    np.random.seed(5)
    all_ids, strings = smp.sample_variable_length(20, length_range=(10, 30), batch_size=64, samples_per_length=8,
                                                  show_progress=False)
    assert len(strings) == 20 and [x.shape[0] for x in all_ids] == [8, 8, 4]
    assert all(10 <= x.shape[1] <= 30 for x in all_ids)
    assert all((x != tok.mask_token_id).all() for x in all_ids)
    smp.count_stars_fn = lambda s: s.count("[IH0+3]")
    smp.check_validity_fn = lambda s: True
    none, good = smp.sample_with_rejection(10, 24, batch_size=16, max_attempts=4, show_progress=False)
    assert none == [] and len(good) <= 10 and all(g.count("[IH0+3]") == 2 for g in good)
    smp.check_validity_fn = None
    import sys
    if "src.utils.chemistry" not in sys.modules:
        with pytest.raises(RuntimeError, match="validity"):
            smp.sample_with_rejection(2, 24, batch_size=4, max_attempts=1, show_progress=False)


def test_full_batch_invariants(gpu_device):
    """Benchmark-size batch of the small architecture, fewer steps: no MASK left, BOS/EOS kept,
    exactly two stars, balanced parentheses, paired ring closures."""
    T, B, L = 10, 4096, 128
    smp0, tok = H.make_sampler("smiles")
    torch.manual_seed(42)
    bb = DiffusionBackbone(tok.vocab_size, 384, 6, 3, 1536, 256, T, 0.1, 0)
    model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=1, pad_token_id=0, bos_token_id=2,
                                     eos_token_id=3).to(gpu_device).eval()
    smp, _ = H.make_sampler("smiles", model=model, num_steps=T, device=str(gpu_device))
    ids, strings = smp.sample(B, L, show_progress=False)
    out = ids.cpu().numpy()
    v = tok.vocab
    assert (out != 1).all() and (out[:, 0] == 2).all() and (out[:, -1] == 3).all()
    assert ((out == v["*"]).sum(1) == 2).all()
    depth = np.cumsum((out == v["("]).astype(int) - (out == v[")"]).astype(int), axis=1)
    assert (depth >= 0).all() and (depth[:, -1] == 0).all()
    for r in smp._core.tt.ring_ids:
        assert (((out == r).sum(1)) % 2 == 0).all()
