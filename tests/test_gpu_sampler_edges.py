"""Edges of the sampler classes' public API on the GPU: the smallest and the longest sequences, more reverse
steps than masked positions, ragged last batches, and the argument errors the reference raises
(Bi_Diffusion_SMILES/src/sampling/sampler.py:42-43, :749-750, :835-836)."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_gpu_sampler_e2e import build_model, replay
from oracle import sampler_ref as ora

pytestmark = pytest.mark.gpu


def _sampler(variant, dev, T, **kw):
    _, tok = H.make_sampler(variant)
    model = build_model(tok, dev, T)
    smp, _ = H.make_sampler(variant, model=model, num_steps=T, device=str(dev), **kw)
    return smp, tok


@pytest.mark.parametrize("variant", ["smiles", "selfies", "group", "bpe"])
@pytest.mark.parametrize("B,L", [(1, 2), (1, 3), (3, 5), (1, 128)])
def test_smallest_batches_and_sequences(gpu_device, variant, B, L):
    """One sequence, and sequences with no (L = 2: BOS, EOS), one and three free positions, every step replayed by
    the oracle. T = 6 is more steps than free positions: the late steps find nothing to draw."""
    T = 6
    smp, tok = _sampler(variant, gpu_device, T)
    smp._core.seed = 5
    smp._core.record = []
    ids, strings = smp.sample(B, L, show_progress=False)
    assert ids.shape == (B, L) and ids.dtype == torch.long and len(strings) == B
    out = ids.cpu().numpy()
    assert (out != tok.mask_token_id).all()
    assert (out[:, 0] == tok.bos_token_id).all() and (out[:, -1] == tok.eos_token_id).all()
    tb = H.oracle_tables(smp)
    rec = smp._core.record
    assert len(rec) == T
    fixed = (rec[0]["ids_before"].numpy() != tb.mask_id).astype(np.uint8) if variant in ("smiles", "bpe") else None
    o, _ = replay(rec, tb, T, L, 1.0, True, fixed)
    of = ora.sample_fixup(tb, o["ids"], o["logits"], natural_eos=False, attn=o["attn"], has_eos=o["has_eos"])
    np.testing.assert_array_equal(out, of["ids"])


@pytest.mark.parametrize("variant,L", [("smiles", 200), ("selfies", 129), ("smiles", 512)])
def test_sequences_longer_than_the_fused_kernels_take(gpu_device, variant, L):
    """L > 128 leaves the single-launch forward for the per-operation kernels (LayerNorm, tiled GEMMs, stand-alone
    attention); L = 512 is the position table's last row. Steps replayed by the oracle, repairs included."""
    T, B = 5, 6
    smp, tok = _sampler(variant, gpu_device, T)
    smp._core.seed = 17
    smp._core.record = []
    ids, strings = smp.sample(B, L, show_progress=False)
    out = ids.cpu().numpy()
    assert out.shape == (B, L) and (out != tok.mask_token_id).all()
    tb = H.oracle_tables(smp)
    rec = smp._core.record
    fixed = (rec[0]["ids_before"].numpy() != tb.mask_id).astype(np.uint8) if variant == "smiles" else None
    o, _ = replay(rec, tb, T, L, 1.0, True, fixed)
    of = ora.sample_fixup(tb, o["ids"], o["logits"], natural_eos=False, attn=o["attn"], has_eos=o["has_eos"])
    np.testing.assert_array_equal(out, of["ids"])
    with pytest.raises((RuntimeError, ValueError, IndexError)):
        smp.sample(2, 513, show_progress=False)          # beyond max_position_embeddings, as the reference's embedding


def test_ragged_last_batch_equals_one_batch(gpu_device):
    """sample_batch(10, batch_size=4) draws batches of 4, 4 and 2; numbered globally they are the rows of sample(10)."""
    T, L = 6, 24
    smp, tok = _sampler("smiles", gpu_device, T)
    smp._core.seed = 9
    whole, s_whole = smp.sample(10, L, show_progress=False)
    parts, s_parts = smp.sample_batch(10, L, batch_size=4, show_progress=False)
    assert [p.shape[0] for p in parts] == [4, 4, 2]
    assert torch.equal(torch.cat(parts), whole) and s_parts == s_whole
    lengths = [2, 24, 3, 7, 24, 11, 5]
    parts, s_parts = smp.sample_batch(7, L, batch_size=3, show_progress=False, lengths=lengths)
    assert [tuple(p.shape) for p in parts] == [(3, 24), (3, 24), (1, 5)]      # each batch as long as its longest row
    whole, s_whole = smp.sample_with_lengths(lengths, max_length=L, show_progress=False)
    for i, ln in enumerate(lengths):
        row = parts[i // 3][i % 3]
        assert torch.equal(row[:ln], whole[i, :ln])
        assert (row[ln:] == tok.pad_token_id).all()
    assert s_parts == s_whole
    none_ids, none_strings = smp.sample_batch(0, L, batch_size=4, show_progress=False)
    assert none_ids == [] and none_strings == []
    empty, es = smp.sample_with_lengths([], show_progress=False)
    assert empty.numel() == 0 and es == []


def test_argument_errors_of_the_reference(gpu_device):
    smp, tok = _sampler("smiles", gpu_device, 4)
    with pytest.raises(ValueError, match="lengths must match num_samples"):
        smp.sample_batch(3, 16, lengths=[5, 6], show_progress=False)
    with pytest.raises(ValueError, match="smaller than required"):
        smp.sample_with_lengths([5, 40], max_length=32, show_progress=False)
    for variant in ("smiles", "selfies", "group"):
        with pytest.raises(ValueError, match="temperature"):
            H.make_sampler(variant, model=smp.diffusion_model, temperature=0.0, device=str(gpu_device))
    # lengths below 2 are raised to 2 (BOS + EOS), as the reference's max(2, int(length))
    ids, strings = smp.sample_with_lengths([0, 1, 2], show_progress=False)
    assert ids.shape == (3, 2) and (ids[:, 0] == tok.bos_token_id).all() and (ids[:, 1] == tok.eos_token_id).all()
    assert strings == ["", "", ""]


@pytest.mark.parametrize("variant", ["smiles", "bpe", "group"])
@pytest.mark.parametrize("d,heads,F", [(768, 6, 3072), (1152, 9, 4608)])
def test_samplers_on_the_medium_and_large_widths(gpu_device, variant, d, heads, F):
    """The reference's medium / large model widths (two layers): per-operation kernels for the denoiser, chosen-row
    logits from hidden rows of that width (V = 39 in the step kernel, V = 233 / 803 through the row workspace), every
    step replayed by the oracle."""
    T, B, L = 6, 10, 48
    _, tok = H.make_sampler(variant)
    model = build_model(tok, gpu_device, T, heads=heads, layers=2, d=d, F=F)
    smp, _ = H.make_sampler(variant, model=model, num_steps=T, device=str(gpu_device))
    smp._core.seed = 23
    smp._core.record = []
    ids, strings = smp.sample(B, L, show_progress=False)
    out = ids.cpu().numpy()
    assert (out != tok.mask_token_id).all() and len(strings) == B
    tb = H.oracle_tables(smp)
    rec = smp._core.record
    fixed = (rec[0]["ids_before"].numpy() != tb.mask_id).astype(np.uint8) if variant in ("smiles", "bpe") else None
    o, _ = replay(rec, tb, T, L, 1.0, True, fixed)
    of = ora.sample_fixup(tb, o["ids"], o["logits"], natural_eos=False, attn=o["attn"], has_eos=o["has_eos"])
    np.testing.assert_array_equal(out, of["ids"])


@pytest.mark.parametrize("variant", ["smiles", "selfies", "group", "bpe"])
def test_ragged_batches_sample_the_same_ids(gpu_device, variant):
    """Per-sample lengths padded to the longest row: with the sequences dealt to the forward's waves by their own
    lengths (the default) the samplers return the ids of the launch that computes every padded row."""
    T, L = 8, 100
    smp, tok = _sampler(variant, gpu_device, T)
    rs = np.random.RandomState(11)
    lengths = [int(x) for x in rs.randint(2, L + 1, size=45)]
    lengths[3] = L
    smp._core.seed = 31
    assert smp._core.ragged is True
    if variant in ("smiles", "bpe"):
        run = lambda: smp.sample_with_lengths(lengths, max_length=L, show_progress=False)
    else:
        def run():
            ids = torch.full((len(lengths), L), tok.pad_token_id, dtype=torch.long, device=gpu_device)
            attn = torch.zeros_like(ids)
            for i, n in enumerate(lengths):
                ids[i, :n] = tok.mask_token_id
                ids[i, 0] = tok.bos_token_id
                ids[i, n - 1] = tok.eos_token_id
                attn[i, :n] = 1
            return smp._core.run(ids, attn, None), None
    a, sa = run()
    smp._core.ragged = False
    b, sb = run()
    assert torch.equal(a, b) and sa == sb
    assert (a != tok.mask_token_id).all()


@pytest.mark.parametrize("variant", ["smiles", "bpe", "group", "selfies"])
def test_variable_length_groups_in_one_loop_equal_the_group_by_group_loop(gpu_device, variant):
    """sample_variable_length draws the groups' lengths first and runs up to batch_size sequences of different
    lengths through one reverse loop; with a fixed seed the ids and strings are those of the reference's shape of
    the loop (one reverse loop per group of samples_per_length sequences, sampler.py:868-911)."""
    T = 8
    smp, tok = _sampler(variant, gpu_device, T)
    smp._core.seed = 77
    kw = dict(length_range=(20, 100), batch_size=64, samples_per_length=16, show_progress=False)
    np.random.seed(9)
    smp.merge_variable_length = True
    a_ids, a_str = smp.sample_variable_length(150, **kw)
    np.random.seed(9)
    smp.merge_variable_length = False
    b_ids, b_str = smp.sample_variable_length(150, **kw)
    assert [tuple(x.shape) for x in a_ids] == [tuple(x.shape) for x in b_ids]
    assert [x.shape[0] for x in a_ids] == [16] * 9 + [6] and len(a_str) == 150
    for x, y in zip(a_ids, b_ids):
        assert torch.equal(x, y)
        assert (x != tok.mask_token_id).all() and (x[:, 0] == tok.bos_token_id).all() and (x[:, -1] == tok.eos_token_id).all()
    assert a_str == b_str


def test_selfies_batches_by_length_share_reverse_loops(gpu_device):
    """SELFIES sample_batch(lengths=...) forms one group of batches per distinct length (reference SELFIES
    sampler.py:421-502); here up to batch_size rows of different lengths share a loop. Same id tensors in the same
    order and the same strings in the caller's order as the loop per distinct length, with a fixed seed."""
    T = 8
    smp, tok = _sampler("selfies", gpu_device, T)
    smp._core.seed = 41
    rs = np.random.RandomState(2)
    lengths = [int(x) for x in rs.randint(12, 90, size=70)]
    smp.merge_variable_length = True
    a_ids, a_str = smp.sample_batch(70, 128, batch_size=32, show_progress=False, lengths=lengths)
    smp.merge_variable_length = False
    b_ids, b_str = smp.sample_batch(70, 128, batch_size=32, show_progress=False, lengths=lengths)
    assert [tuple(x.shape) for x in a_ids] == [tuple(x.shape) for x in b_ids]
    assert all(torch.equal(x, y) for x, y in zip(a_ids, b_ids)) and a_str == b_str and len(a_str) == 70
