"""GPU parity of the reverse step and the post-loop repairs (bd_sample_step / bd_sample_fixup
through the C ABI). Bar: bit-exact ids (and -inf / value pattern of the edited logits) against
the reference golden vectors and against the CPU oracle on identical logits and uniforms."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import replay_inputs as RI
from bidiff.sampling import core
from oracle import sampler_ref as ora

pytestmark = pytest.mark.gpu
ALL = [(v, name) for v in ("smiles", "selfies", "group") for name in RI.CASES[v]]


def _same_logits(a, b):
    assert a.shape == b.shape
    assert np.array_equal(np.isneginf(a), np.isneginf(b)), "-inf pattern"
    fin = ~np.isneginf(b)
    assert np.array_equal(a[fin].view(np.uint32), b[fin].view(np.uint32)), "finite values"


@pytest.mark.parametrize("variant,name", ALL)
def test_hip_step_replays_reference_golden(gpu_device, variant, name):
    """LOGITS mode with explicit uniforms reproduces every step of the reference loop."""
    dev = gpu_device
    case, golden = H.load_cases(variant)[name]
    smp, tok = H.make_sampler(variant, num_steps=case["T"], temperature=case["temperature"],
                              use_constraints=case["use_constraints"])
    tb = H.oracle_tables(smp)
    dt = H.hip_tables(tb, dev)
    T, B, L, V = case["T"], case["B"], case["L"], tok.vocab_size
    ktab = core.unmask_count_table(T, L)
    ids = torch.from_numpy(golden["ids_before"][0].copy()).to(dev)
    attn = torch.from_numpy(golden["attn_before"][0].copy()).to(dev)
    keymask = (attn != 0).to(torch.uint8)
    fixed_np, extra = H.case_fixed_and_variant(variant, case, golden["ids_before"][0], tb.mask_id)
    fixed = None if fixed_np is None else torch.from_numpy(fixed_np).to(dev)
    natural = bool(case.get("natural_eos", False))
    has_eos = None
    if variant == "selfies":
        has_eos = torch.zeros(B, dtype=torch.uint8, device=dev) if natural else torch.ones(B, dtype=torch.uint8, device=dev)
    logits = None
    for i, t in enumerate(range(T, 0, -1)):
        logits = torch.from_numpy(RI.step_logits(case, t, B, L, V)).to(dev)
        H.hip_step(tb, dt, ids, ktab[t], t, logits=logits, temperature=case["temperature"],
                   use_constraints=case["use_constraints"], fixed=fixed,
                   u_sel=torch.from_numpy(RI.step_u_sel(case, t, B, L)).to(dev),
                   u_tok=torch.from_numpy(RI.step_u_tok(case, t, B, L, V)).to(dev),
                   natural_eos=natural, attn=attn if natural else None, keymask=keymask if natural else None,
                   has_eos=has_eos if natural else None, variant_extra=extra)
        if i + 1 < T:
            np.testing.assert_array_equal(ids.cpu().numpy(), golden["ids_before"][i + 1], err_msg=f"after t={t}")
            np.testing.assert_array_equal(attn.cpu().numpy(), golden["attn_before"][i + 1])
            assert torch.equal(keymask, (attn != 0).to(torch.uint8))
    if "final_logits" in golden:
        _same_logits(logits.cpu().numpy(), golden["final_logits"])
    H.hip_fixup(tb, dt, ids, logits, do_fix=case["use_constraints"], natural_eos=natural,
                attn=attn if natural else None, keymask=keymask if natural else None,
                has_eos=has_eos if natural else None)
    if not case.get("tie_hazard"):
        np.testing.assert_array_equal(ids.cpu().numpy(), golden["final_ids"])
    else:
        # tied top-k in the reference repair: compare with the oracle's lowest-index rule instead
        _, o_final, _ = H.oracle_replay(variant, case, golden)
        np.testing.assert_array_equal(ids.cpu().numpy(), o_final)


def _random_state(rs, tb, B, L, frac_masked, variant):
    ids = rs.randint(5, tb.V, size=(B, L)).astype(np.int64)
    ids[rs.random_sample((B, L)) < frac_masked] = tb.mask_id
    ids[:, 0] = tb.bos_id
    ids[:, -1] = tb.eos_id
    fixed = None
    if variant in ("smiles", "bpe"):
        fixed = (rs.random_sample((B, L)) < 0.05).astype(np.uint8)
        fixed[:, 0] = 1
        fixed[:, -1] = 1
    return ids, fixed


@pytest.mark.parametrize("variant,B,L,t,T", [("smiles", 64, 128, 100, 100), ("smiles", 32, 128, 1, 100),
                                             ("smiles", 8, 512, 7, 100), ("bpe", 16, 128, 3, 50),
                                             ("selfies", 64, 128, 2, 100), ("selfies", 16, 300, 1, 100),
                                             ("group", 32, 128, 5, 100), ("group", 8, 128, 1, 100)])
def test_hip_step_philox_matches_oracle(gpu_device, variant, B, L, t, T):
    """Counter-based uniforms generated inside the kernel: same ids and edited logits as the oracle
    regenerating the same Philox numbers on the CPU."""
    dev = gpu_device
    smp, tok = H.make_sampler(variant, num_steps=T)
    tb = H.oracle_tables(smp)
    dt = H.hip_tables(tb, dev)
    rs = np.random.RandomState(B * 7 + L + t)
    frac = 0.9 if t > 50 else (0.3 if t > 1 else 0.1)
    ids, fixed = _random_state(rs, tb, B, L, frac, variant)
    raw = (rs.standard_normal((B, L, tb.V)) * 3).astype(np.float32)
    ktab = core.unmask_count_table(T, L)
    seed, off = 0x1234567890ABCDEF, 1000
    has_eos = np.ones(B, dtype=np.uint8) if variant == "selfies" else None
    o = ora.sample_step(tb, ids, raw, ktab[t], t, temperature=0.9, fixed=fixed, seed=seed, seq_offset=off,
                        has_eos=has_eos)
    d_ids = torch.from_numpy(ids.copy()).to(dev)
    d_lg = torch.from_numpy(raw.copy()).to(dev)
    H.hip_step(tb, dt, d_ids, ktab[t], t, logits=d_lg, temperature=0.9,
               fixed=None if fixed is None else torch.from_numpy(fixed).to(dev), seed=seed, seq_offset=off,
               has_eos=None if has_eos is None else torch.from_numpy(has_eos).to(dev))
    np.testing.assert_array_equal(d_ids.cpu().numpy(), o["ids"])
    _same_logits(d_lg.cpu().numpy(), o["logits"])
    if t == 1:
        assert (d_ids.cpu().numpy()[fixed == 0] != tb.mask_id).all() if fixed is not None else (d_ids != tb.mask_id).all()
        of = ora.sample_fixup(tb, o["ids"], o["logits"])
        H.hip_fixup(tb, dt, d_ids, d_lg)
        np.testing.assert_array_equal(d_ids.cpu().numpy(), of["ids"])


@pytest.mark.parametrize("row_cap", [0, 1, 4, 32])
@pytest.mark.parametrize("variant,L,d", [("smiles", 128, 384), ("selfies", 40, 128), ("group", 128, 256), ("bpe", 64, 384),
                                         ("bpe", 128, 512)])
def test_hip_hidden_mode_matches_oracle_on_dumped_logits(gpu_device, variant, L, d, row_cap):
    """HIDDEN mode (vocabulary logits computed in-kernel for the chosen rows only): the raw rows it
    used are dumped and replayed through the oracle, which must draw the same tokens; the dumped
    rows themselves must equal hidden @ W_out^T. ``row_cap`` > 0: the first row_cap rows of each sequence come
    from the product computed ahead of the draws, the rest (t = 2 draws 5 rows per sequence) one by one."""
    dev = gpu_device
    B, T = 48, 100
    smp, tok = H.make_sampler(variant, num_steps=T)
    tb = H.oracle_tables(smp)
    dt = H.hip_tables(tb, dev)
    rs = np.random.RandomState(L + d)
    ktab = core.unmask_count_table(T, L)
    for t, frac in ((60, 0.8), (2, 0.08)):
        ids, fixed = _random_state(rs, tb, B, L, frac, variant)
        hidden = torch.from_numpy(rs.standard_normal((B, L, d)).astype(np.float32)).to(dev).bfloat16()
        wout = torch.from_numpy((rs.standard_normal((tb.V, d)) / np.sqrt(d) * 3).astype(np.float32)).to(dev).bfloat16()
        dump = torch.zeros(B, L, tb.V, device=dev)
        d_ids = torch.from_numpy(ids.copy()).to(dev)
        has_eos = np.ones(B, dtype=np.uint8) if variant == "selfies" else None
        H.hip_step(tb, dt, d_ids, ktab[t], t, hidden=hidden, wout=wout, dump=dump, temperature=1.1,
                   fixed=None if fixed is None else torch.from_numpy(fixed).to(dev), seed=99, seq_offset=5,
                   has_eos=None if has_eos is None else torch.from_numpy(has_eos).to(dev), row_cap=row_cap)
        raw = dump.cpu().numpy()
        o = ora.sample_step(tb, ids, raw, ktab[t], t, temperature=1.1, fixed=fixed, seed=99, seq_offset=5,
                            has_eos=has_eos)
        np.testing.assert_array_equal(d_ids.cpu().numpy(), o["ids"])
        full = (hidden.float() @ wout.float().t()).cpu().numpy()
        used = raw != 0
        assert used.any()
        assert np.abs(raw[used] - full[used]).max() < 1e-3 * np.abs(full).max()
        changed = (d_ids.cpu().numpy() != ids)
        assert used.any(axis=-1)[changed].all(), "every drawn position must have a dumped row"


@pytest.mark.parametrize("variant,L,d", [("bpe", 128, 384), ("selfies", 64, 256)])
def test_hidden_mode_row_workspace_is_batch_independent(gpu_device, variant, L, d):
    """A sequence's drawn tokens and the rows it used are the same bits whatever batch it is submitted in (which
    path computes a row depends on the row's rank inside its own sequence only)."""
    dev = gpu_device
    B, T, t = 40, 100, 3
    smp, tok = H.make_sampler(variant, num_steps=T)
    tb = H.oracle_tables(smp)
    dt = H.hip_tables(tb, dev)
    rs = np.random.RandomState(7)
    ktab = core.unmask_count_table(T, L)
    ids, fixed = _random_state(rs, tb, B, L, 0.2, variant)       # about 8 draws per sequence at t = 3
    hidden = torch.from_numpy(rs.standard_normal((B, L, d)).astype(np.float32)).to(dev).bfloat16()
    wout = torch.from_numpy((rs.standard_normal((tb.V, d)) / np.sqrt(d) * 3).astype(np.float32)).to(dev).bfloat16()
    has_eos = np.ones(B, dtype=np.uint8) if variant == "selfies" else None

    def run(lo, hi, cap):
        dump = torch.zeros(hi - lo, L, tb.V, device=dev)
        d_ids = torch.from_numpy(ids[lo:hi].copy()).to(dev)
        H.hip_step(tb, dt, d_ids, ktab[t], t, hidden=hidden[lo:hi].contiguous(), wout=wout, dump=dump,
                   fixed=None if fixed is None else torch.from_numpy(fixed[lo:hi].copy()).to(dev), seed=5, seq_offset=100 + lo,
                   has_eos=None if has_eos is None else torch.from_numpy(has_eos[lo:hi].copy()).to(dev), row_cap=cap)
        return d_ids.cpu(), dump.cpu()

    whole_ids, whole_dump = run(0, B, 4)
    for lo, hi in ((0, 1), (13, 30), (37, 40)):
        part_ids, part_dump = run(lo, hi, 4)
        assert torch.equal(part_ids, whole_ids[lo:hi])
        assert torch.equal(part_dump, whole_dump[lo:hi])
    # without the workspace the rows are summed in another order: same tokens are not guaranteed, close rows are
    _, plain_dump = run(0, B, 0)
    used = (whole_dump != 0) & (plain_dump != 0)
    assert (whole_dump[used] - plain_dump[used]).abs().max() < 1e-3 * plain_dump.abs().max()


def test_step_rejects_bad_arguments(gpu_device):
    smp, tok = H.make_sampler("smiles")
    tb = H.oracle_tables(smp)
    dt = H.hip_tables(tb, gpu_device)
    ids = torch.full((2, 600), tb.mask_id, dtype=torch.long, device=gpu_device)
    lg = torch.zeros(2, 600, tb.V, device=gpu_device)
    with pytest.raises(RuntimeError, match="outside"):
        H.hip_step(tb, dt, ids, np.ones(601), 1, logits=lg)
    with pytest.raises(RuntimeError, match="temperature"):
        H.hip_step(tb, dt, ids[:, :16].contiguous(), np.ones(17), 1, logits=lg[:, :16].contiguous(), temperature=0.0)
