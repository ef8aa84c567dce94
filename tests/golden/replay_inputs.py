"""Deterministic inputs shared by the fixture generators and the tests that replay them.

All arrays come from ``numpy.random.RandomState`` (legacy MT19937 stream, frozen by NumPy's
compatibility policy), so a fixture only has to store seeds and expected outputs.
"""
import numpy as np

from bidiff.utils import synthetic

_SM = synthetic.smiles_vocab()
_SF = synthetic.selfies_vocab()
_GR, _ = synthetic.group_selfies_vocab()


def _b(vocab, **kw):
    return {vocab[k]: v for k, v in kw.items()}


def _smiles_bias(scale_syntax=2.5):
    s = scale_syntax
    return {_SM["("]: s, _SM[")"]: s, _SM["*"]: s, _SM["-"]: s * 0.6, _SM["="]: s, _SM["#"]: s * 0.5,
            _SM["1"]: s, _SM["2"]: s * 0.8, _SM["%10"]: s * 0.5, _SM["[EOS]"]: 1.0, _SM["[PAD]"]: 1.0}


CASES = {
    "smiles": {
        "on_small": dict(kind="sample", B=8, L=24, T=12, temperature=1.0, use_constraints=True, seed=101,
                         scale=2.0, bias=_smiles_bias()),
        "off_temp": dict(kind="sample", B=6, L=20, T=9, temperature=0.7, use_constraints=False, seed=102,
                         scale=2.0, bias=_smiles_bias()),
        "lengths": dict(kind="lengths", B=6, L=24, T=8, temperature=1.3, use_constraints=True, seed=103,
                        scale=2.0, bias=_smiles_bias(), lengths=[5, 9, 24, 2, 17, 12]),
        "conditional": dict(kind="conditional", B=4, L=16, T=6, temperature=1.0, use_constraints=True, seed=104,
                            scale=2.0, bias=_smiles_bias(), prefix=[_SM["C"], _SM["("]], suffix=[_SM[")"]]),
        "long70": dict(kind="sample", B=3, L=70, T=5, temperature=1.0, use_constraints=True, seed=105,
                       scale=2.0, bias=_smiles_bias(2.0)),
        "long130": dict(kind="sample", B=2, L=130, T=4, temperature=0.9, use_constraints=True, seed=106,
                        scale=1.5, bias=_smiles_bias(3.0)),
        # only parentheses are finite: rows where both are forbidden take the fallback pool
        "fallback": dict(kind="sample", B=5, L=14, T=7, temperature=1.0, use_constraints=True, seed=107,
                         scale=1.0, bias={}, only=[_SM["("], _SM[")"]], tie_hazard=True),
        "t49": dict(kind="sample", B=2, L=100, T=49, temperature=1.0, use_constraints=True, seed=108,
                    scale=2.0, bias=_smiles_bias()),
    },
    "selfies": {
        "on_small": dict(kind="sample", B=8, L=20, T=10, temperature=1.0, use_constraints=True, seed=201,
                         scale=2.0, bias={_SF["[I+3]"]: 1.0}),
        "on_rare_ph": dict(kind="sample", B=8, L=18, T=9, temperature=1.0, use_constraints=True, seed=202,
                           scale=1.0, bias={_SF["[I+3]"]: -6.0}),
        "natural_eos": dict(kind="sample", B=8, L=22, T=8, temperature=1.1, use_constraints=True, seed=203,
                            scale=2.0, bias={_SF["[I+3]"]: 0.5, _SF["[EOS]"]: 2.5}, natural_eos=True),
        "off": dict(kind="sample", B=5, L=16, T=6, temperature=0.8, use_constraints=False, seed=204,
                    scale=2.0, bias={_SF["[I+3]"]: 2.0}),
        "long70": dict(kind="sample", B=3, L=70, T=5, temperature=1.0, use_constraints=True, seed=205,
                       scale=2.0, bias={_SF["[I+3]"]: -2.0}),
        # urgency boost when the masked slots' placeholder logits sit far below the column maximum: the reference
        # boosts iff some masked slot's fp32 softmax-over-positions weight is non-zero (denormals included), which
        # holds down to about 104 - ln Z below the maximum. "col_ramp" subtracts lo..hi from the placeholder column
        # over the positions of the second half (the first half, unmasked early by "first_half_early", holds the maximum).
        "urgency_85": dict(kind="sample", B=6, L=16, T=6, temperature=1.0, use_constraints=True, seed=209,
                           scale=0.5, bias={_SF["[I+3]"]: -3.0}, col_ramp=(_SF["[I+3]"], 82.0, 86.0), first_half_early=True),
        "urgency_95": dict(kind="sample", B=6, L=16, T=6, temperature=1.0, use_constraints=True, seed=210,
                           scale=0.5, bias={_SF["[I+3]"]: -3.0}, col_ramp=(_SF["[I+3]"], 90.0, 108.0), first_half_early=True),
        "urgency_110": dict(kind="sample", B=6, L=16, T=6, temperature=1.0, use_constraints=True, seed=211,
                            scale=0.5, bias={_SF["[I+3]"]: -3.0}, col_ramp=(_SF["[I+3]"], 106.0, 115.0), first_half_early=True),
        # sample_conditional: prefix / suffix fixed, positions chosen among the non-fixed masks
        "conditional": dict(kind="conditional", B=6, L=18, T=7, temperature=1.0, use_constraints=True, seed=206,
                            scale=2.0, bias={_SF["[I+3]"]: 2.5}, prefix=[_SF["[C]"], _SF["[I+3]"]],
                            suffix=[_SF["[=O]"]]),
        # a MASK inside the fixed prefix: it is never drawn, but it counts as masked in the constraint, keeps
        # its special-token logits and is edited by the in-step placeholder update
        "conditional_maskfix": dict(kind="conditional", B=6, L=16, T=6, temperature=0.9, use_constraints=True,
                                    seed=207, scale=2.0, bias={_SF["[I+3]"]: 3.0},
                                    prefix=[_SF["[C]"], _SF["[MASK]"], _SF["[N]"]], suffix=[_SF["[MASK]"], _SF["[O]"]]),
        "conditional_natural": dict(kind="conditional", B=8, L=20, T=8, temperature=1.1, use_constraints=True,
                                    seed=208, scale=2.0, bias={_SF["[I+3]"]: 0.5, _SF["[EOS]"]: 2.5},
                                    prefix=[_SF["[C]"]], suffix=None, natural_eos=True),
    },
    "group": {
        "on_small": dict(kind="sample", B=8, L=20, T=10, temperature=1.0, use_constraints=True, seed=301,
                         scale=2.0, bias={_GR["[IH0+3]"]: 3.0, _GR["[:/0G0]"]: 3.0, _GR["[:/0G4]"]: 3.0,
                                          _GR["[UNK]"]: 4.0}),
        "off": dict(kind="sample", B=5, L=16, T=6, temperature=1.2, use_constraints=False, seed=302,
                    scale=2.0, bias={_GR["[IH0+3]"]: 3.0, _GR["[:/0G4]"]: 3.0}),
        "long70": dict(kind="sample", B=3, L=70, T=5, temperature=1.0, use_constraints=True, seed=303,
                       scale=2.0, bias={_GR["[IH0+3]"]: 2.0, _GR["[:/0G1]"]: 2.0}),
        # sample_conditional: the in-step update looks at the direct placeholder id alone
        "conditional": dict(kind="conditional", B=8, L=18, T=7, temperature=1.0, use_constraints=True, seed=304,
                            scale=2.0, bias={_GR["[IH0+3]"]: 4.0, _GR["[:/0G0]"]: 3.5, _GR["[:/0G4]"]: 3.0},
                            prefix=[_GR["[C]"]], suffix=[_GR["[O]"]]),
        "conditional_maskfix": dict(kind="conditional", B=8, L=16, T=5, temperature=1.0, use_constraints=True,
                                    seed=305, scale=2.0, bias={_GR["[IH0+3]"]: 4.5, _GR["[:/0G1]"]: 3.5},
                                    prefix=[_GR["[C]"], _GR["[MASK]"]], suffix=[_GR["[MASK]"], _GR["[O]"]]),
    },
}


# variant bits of the step kernel / oracle that the sample_conditional flavours add (include/bidiff.h)
CONDITIONAL_EXTRA = {"smiles": 0, "selfies": 1 | 32 | 128, "group": 1 | 32 | 64}


def conditional_fixed(case, natural=False):
    """fixed_mask of a sample_conditional case: BOS, the prefix, the suffix and (unless EOS is sampled) EOS."""
    B, L = case["B"], case["L"]
    fixed = np.zeros((B, L), dtype=np.uint8)
    fixed[:, 0] = 1
    if not natural:
        fixed[:, -1] = 1
    if case.get("prefix"):
        fixed[:, 1:1 + len(case["prefix"])] = 1
    if case.get("suffix"):
        fixed[:, -1 - len(case["suffix"]):-1] = 1
    return fixed


def step_logits(case, t, B, L, V):
    rs = np.random.RandomState(case["seed"] * 1000 + t)
    x = rs.standard_normal((B, L, V)) * case["scale"]
    for tok, bias in case.get("bias", {}).items():
        x[:, :, tok] += bias
    if case.get("col_ramp"):
        tok, lo, hi = case["col_ramp"]
        half = L // 2
        x[:, half:, tok] -= np.linspace(lo, hi, L - half)[None, :]
    x = x.astype(np.float32)
    if case.get("only"):
        keep = np.zeros(V, dtype=bool)
        keep[list(case["only"])] = True
        x[:, :, ~keep] = -np.inf
    return x


def step_u_sel(case, t, B, L):
    rs = np.random.RandomState(case["seed"] * 1000 + 500 + t)
    u = rs.random_sample((B, L)).astype(np.float32)
    if case.get("first_half_early"):      # smaller keys in the first half: those positions are unmasked first
        u[:, :L // 2] *= np.float32(0.25)
        u[:, L // 2:] = np.float32(0.25) + u[:, L // 2:] * np.float32(0.75)
    return u


def step_u_tok(case, t, B, L, V):
    rs = np.random.RandomState(case["seed"] * 1000 + 700 + t)
    return rs.random_sample((B, L, V)).astype(np.float32)


def _neglog_np(x):
    """-ln(x) with fp32 multiplies/adds only: the numpy twin of the library's bd_neglog."""
    x = np.asarray(x, dtype=np.float32)
    bits = x.view(np.uint32)
    e = (bits >> np.uint32(23)).astype(np.int32) - 127
    m = ((bits & np.uint32(0x007FFFFF)) | np.uint32(0x3F800000)).view(np.float32)
    big = m > np.float32(1.41421356)
    m = np.where(big, m * np.float32(0.5), m).astype(np.float32)
    e = np.where(big, e + 1, e)
    f = (m - np.float32(1.0)).astype(np.float32)
    z = (f * f).astype(np.float32)
    y = np.full_like(f, np.float32(7.0376836292e-2))
    for c in (-1.1514610310e-1, 1.1676998740e-1, -1.2420140846e-1, 1.4249322787e-1, -1.6668057665e-1,
              2.0000714765e-1, -2.4999993993e-1, 3.3333331174e-1):
        y = (y * f).astype(np.float32)
        y = (y + np.float32(c)).astype(np.float32)
    y = (y * f).astype(np.float32)
    y = (y * z).astype(np.float32)
    hz = (np.float32(0.5) * z).astype(np.float32)
    y = (y - hz).astype(np.float32)
    r = (f + y).astype(np.float32)
    el = (e.astype(np.float32) * np.float32(0.693147180559945)).astype(np.float32)
    r = (r + el).astype(np.float32)
    return (-r).astype(np.float32)


def gumbel_np(u):
    u = np.array(u, dtype=np.float32, copy=True)
    lo = np.float32(2.98023223876953125e-08)
    u = np.where(u > lo, u, lo).astype(np.float32)
    u = np.minimum(u, np.float32(0.99999994)).astype(np.float32)
    return _neglog_np(_neglog_np(u))


# ---------------------------------------------------------------- denoiser fixtures
BACKBONE_CFGS = {
    "tiny_h2": dict(vocab_size=39, hidden_size=128, num_layers=2, num_heads=2, ffn_hidden_size=256,
                    max_position_embeddings=64, num_diffusion_steps=20, dropout=0.0, pad_token_id=0),
    "tiny_h1": dict(vocab_size=39, hidden_size=128, num_layers=2, num_heads=1, ffn_hidden_size=256,
                    max_position_embeddings=64, num_diffusion_steps=20, dropout=0.0, pad_token_id=0),
    "tiny_h4": dict(vocab_size=43, hidden_size=128, num_layers=1, num_heads=4, ffn_hidden_size=128,
                    max_position_embeddings=160, num_diffusion_steps=10, dropout=0.0, pad_token_id=0),
}
SMALL_CFG = dict(vocab_size=39, hidden_size=384, num_layers=6, num_heads=3, ffn_hidden_size=1536,
                 max_position_embeddings=256, num_diffusion_steps=100, dropout=0.1, pad_token_id=0)


def small_arch_case(V):
    """Config and inputs of the benchmark-architecture golden (model_small.npz): two rows of 128 tokens, the second
    cut to 90; weights = seeded init (torch.manual_seed(42)) + perturb_state_dict(., 5)."""
    cfg = dict(SMALL_CFG, vocab_size=V, dropout=0.0)
    ids, attn, t = (x[[0, 5]].copy() for x in backbone_inputs(cfg, 61, B=6, L=128))
    attn[1, 90:] = 0
    ids[1, 90:] = 0
    ids[1, 89] = 3
    return cfg, ids, attn, t, (1 if V == 39 else 4)


PROPERTY_HIDDEN = [256, 64]      # the reference default of PropertyHead (property_head.py:17)


def perturb_state_dict(sd, seed):
    """Make biases and LayerNorm affine parameters non-trivial (the seeded init leaves them at
    0 / 1): in-place, key order of the state_dict, frozen RandomState stream."""
    import torch
    rs = np.random.RandomState(seed)
    for k, v in sd.items():
        if not v.dtype.is_floating_point or k.endswith("mask_schedule"):
            continue
        noise = torch.from_numpy(rs.standard_normal(tuple(v.shape)).astype(np.float32))
        if k.endswith("bias"):
            v.add_(0.1 * noise)
        elif "norm" in k and k.endswith("weight"):
            v.add_(0.1 * noise)
        else:
            v.add_(0.02 * noise)
    return sd


def backbone_inputs(cfg, seed, B=5, L=24):
    rs = np.random.RandomState(seed)
    ids = rs.randint(5, cfg["vocab_size"], size=(B, L)).astype(np.int64)
    lengths = [L, 10, 1, 0, 17][:B] + [L] * max(0, B - 5)
    attn = np.zeros((B, L), dtype=np.int64)
    for b, n in enumerate(lengths):
        attn[b, :n] = 1
        ids[b, n:] = 0
        if n >= 1:
            ids[b, 0] = 2
        if n >= 2:
            ids[b, n - 1] = 3
    ids[0, 3:9] = 1
    t = rs.randint(1, cfg["num_diffusion_steps"] + 1, size=(B,)).astype(np.int64)
    return ids, attn, t


# ---- optimizer cases (Muon + AdamW)
NS_SHAPES = [(128, 128), (128, 256), (256, 128)]
OPT_STEPS = 3
OPT_WATCH = ["layers.0.attn.q_proj.weight", "layers.1.attn.out_proj.weight", "layers.0.ffn.fc1.weight",
             "layers.1.ffn.fc2.weight", "layers.0.ffn.fc1.bias", "layers.1.norm2.weight", "token_embedding.weight",
             "output_proj.weight"]
MUON_OPT_CFG = {"optimizer": {"type": "muon_adamw", "muon_adamw": {"enabled": True, "muon_lr_multiplier": 20.0,
                                                                   "adamw_betas": [0.9, 0.95], "backend_steps": 5}}}


def ns_input(shape):
    rs = np.random.RandomState(1000 + shape[0] + 7 * shape[1])
    return (rs.standard_normal(shape) * 0.02).astype(np.float32)


def optimizer_grads(model, step):
    """Synthetic gradients per parameter name (frozen stream per step); scaled so that the global norm
    exceeds 1 and the clip is active."""
    rs = np.random.RandomState(500 + step)
    return {n: (rs.standard_normal(tuple(p.shape)) * 0.01).astype(np.float32) for n, p in model.named_parameters()}


# ---- training input pipeline cases
SAMPLER_CASES = [  # (n, batch_size, drop_last, shuffle, seed, bucket_multiplier, replicas, rank, epoch)
    (1000, 32, False, True, 42, 50, 1, 0, 0), (1000, 32, True, True, 42, 3, 1, 0, 2), (1003, 16, False, True, 7, 4, 4, 1, 1),
    (1003, 16, True, True, 7, 4, 4, 3, 0), (257, 64, False, False, 1, 2, 2, 0, 5), (10, 4, False, True, 3, 50, 3, 2, 0)]
POLYMER_MAX_LENGTH = 40


def sampler_lengths(n):
    return np.clip(np.round(np.random.RandomState(n).normal(38.5, 15.0, size=n)), 12, 128).astype(np.int64)


def polymer_strings(n=120, seed=4):
    """Synthetic p-SMILES-like strings over the reference tokenizer's alphabet (two-letter atoms, brackets,
    ring digits, %nn, bonds, stars); some exceed POLYMER_MAX_LENGTH tokens and get truncated."""
    rs = np.random.RandomState(seed)
    parts = ["C", "c", "N", "O", "S", "F", "Cl", "Br", "Si", "(", ")", "=", "#", "1", "2", "3", "%10", "%11", "[NH]", "[Si]",
             "[O-]", "/", "\\", "-", "n", "o", "s", "P", "I"]
    out = []
    for _ in range(n):
        k = int(rs.randint(3, 60))
        out.append("*" + "".join(parts[int(i)] for i in rs.randint(0, len(parts), size=k)) + "*")
    return out
