"""Distributional pins of the two random draws of a reverse step (the reference draws with torch.randperm and
torch.multinomial, sampler.py:680-693; the restatement with per-position keys and Gumbel-max, fed by the
counter-based Philox uniforms). Bit-exact replay tests cannot see a biased uniform or a biased Gumbel
transform; these can: over thousands of sequences that share ONE logits row, with the constraints off,
  * the drawn tokens follow softmax(logits / temperature) over the allowed tokens (chi-square, p > 1e-3),
  * the chosen positions are uniform over the masked positions, for k = 1 and for k = 4 of 14 (chi-square on the
    per-position counts; sampling without replacement only makes the statistic smaller),
  * every sequence gets exactly k(t) positions.
The CPU test runs the C oracle, the GPU test the HIP kernel (Philox path, B = 4096); both must also agree bit for bit."""
import numpy as np
import pytest
from scipy import stats

from tests import helpers as H
from bidiff.sampling import core
from oracle import sampler_ref as ora

L, T = 16, 100


def _setup(B, seed_logits=5):
    smp, tok = H.make_sampler("smiles", num_steps=T, use_constraints=False)
    tb = H.oracle_tables(smp)
    rs = np.random.RandomState(seed_logits)
    row = (rs.standard_normal(tb.V) * 1.2).astype(np.float32)
    logits = np.broadcast_to(row, (B, L, tb.V)).copy()
    ids = np.full((B, L), tb.mask_id, dtype=np.int64)
    ids[:, 0], ids[:, -1] = tb.bos_id, tb.eos_id
    fixed = (ids != tb.mask_id).astype(np.uint8)
    return smp, tb, row, logits, ids, fixed


def _check(tb, row, ids0, ids1, k, temperature):
    B = ids0.shape[0]
    changed = ids0 != ids1
    assert (changed.sum(1) == k).all(), "every sequence unmasks exactly k positions"
    assert not changed[:, 0].any() and not changed[:, -1].any()
    # positions: uniform over the 14 masked slots
    pos_counts = changed[:, 1:-1].sum(0).astype(np.float64)
    exp = np.full(L - 2, B * k / (L - 2))
    chi_pos = float(((pos_counts - exp) ** 2 / exp).sum())
    p_pos = stats.chi2.sf(chi_pos, L - 3)
    assert p_pos > 1e-3, (chi_pos, p_pos, pos_counts)
    # tokens: softmax over the allowed ids (MASK / PAD / BOS / EOS are forbidden at masked positions)
    allowed = np.array([v for v in range(tb.V) if not (tb.tok_flags[v] & 16)])
    z = (row[allowed] / np.float32(temperature)).astype(np.float64)
    prob = np.exp(z - z.max())
    prob /= prob.sum()
    drawn = ids1[changed]
    assert np.isin(drawn, allowed).all()
    counts = np.array([(drawn == v).sum() for v in allowed], dtype=np.float64)
    expc = prob * drawn.size
    assert expc.min() > 5, "bins large enough for the chi-square approximation"
    chi_tok = float(((counts - expc) ** 2 / expc).sum())
    p_tok = stats.chi2.sf(chi_tok, len(allowed) - 1)
    assert p_tok > 1e-3, (chi_tok, p_tok)
    return p_pos, p_tok


@pytest.mark.parametrize("t,temperature", [(100, 1.0), (3, 0.8)])
def test_oracle_draws_follow_softmax_and_uniform_positions(t, temperature):
    B = 3000
    smp, tb, row, logits, ids, fixed = _setup(B)
    ktab = core.unmask_count_table(T, L)
    k = int(ktab[t][L - 2])
    assert k == (1 if t == 100 else 4)
    o = ora.sample_step(tb, ids, logits, ktab[t], t, temperature=temperature, use_constraints=False, fixed=fixed,
                        seed=0xC0FFEE, seq_offset=0)
    _check(tb, row, ids, o["ids"], k, temperature)
    # another range of global sequence indices is an independent sample: it passes too and differs
    o2 = ora.sample_step(tb, ids, logits, ktab[t], t, temperature=temperature, use_constraints=False, fixed=fixed,
                         seed=0xC0FFEE, seq_offset=B)
    _check(tb, row, ids, o2["ids"], k, temperature)
    assert (o["ids"] != o2["ids"]).any()


def test_gumbel_transform_and_uniforms_are_unbiased():
    """The shared uniform -> Gumbel arithmetic: uniforms from the Philox stream are uniform (Kolmogorov-Smirnov),
    and -ln(-ln u) evaluated by the library's polynomial has the Gumbel mean (Euler's constant) and variance."""
    lib = ora.lib()
    u = np.array([lib.ora_uniform_token(12345, s, 7, p, v) for s in range(40) for p in range(8) for v in range(39)],
                 dtype=np.float32)
    assert stats.kstest(u.astype(np.float64), "uniform").pvalue > 1e-3
    g = ora.gumbel(u).astype(np.float64)
    exact = -np.log(-np.log(u.astype(np.float64)))
    assert np.abs(g - exact).max() < 2e-5 * (1 + np.abs(exact).max())
    n = g.size
    assert abs(g.mean() - 0.5772156649) < 4 * (np.pi / np.sqrt(6)) / np.sqrt(n)
    assert abs(g.var() - np.pi ** 2 / 6) < 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("t,temperature", [(100, 1.0), (3, 0.8)])
def test_hip_draws_follow_softmax_and_uniform_positions(gpu_device, t, temperature):
    import torch
    dev = gpu_device
    B = 4096
    smp, tb, row, logits, ids, fixed = _setup(B)
    dt = H.hip_tables(tb, dev)
    ktab = core.unmask_count_table(T, L)
    k = int(ktab[t][L - 2])
    d_ids = torch.from_numpy(ids.copy()).to(dev)
    d_lg = torch.from_numpy(logits.copy()).to(dev)
    H.hip_step(tb, dt, d_ids, ktab[t], t, logits=d_lg, temperature=temperature, use_constraints=False,
               fixed=torch.from_numpy(fixed).to(dev), seed=0xBADC0DE, seq_offset=12345)
    got = d_ids.cpu().numpy()
    _check(tb, row, ids, got, k, temperature)
    o = ora.sample_step(tb, ids, logits, ktab[t], t, temperature=temperature, use_constraints=False, fixed=fixed,
                        seed=0xBADC0DE, seq_offset=12345)
    np.testing.assert_array_equal(got, o["ids"])
