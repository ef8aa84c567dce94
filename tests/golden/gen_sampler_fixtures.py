"""Golden vectors for the reverse-sampling loop, captured from the imported reference.

Run in this build container only (the reference does not travel):
    python tests/golden/gen_sampler_fixtures.py --variant smiles|selfies|group

The reference sampler is executed unchanged except that its two random primitives are fed
explicit uniforms (that is the restatement the oracle and the HIP kernel implement):
  torch.randperm(n)            -> stable argsort of the per-position keys u_sel[t, i, masked_pos]
  torch.multinomial(probs, 1)  -> argmax_v logits[i, pos, v] + gumbel(u_tok[t, i, pos, v]) over
                                  finite-or-allowed entries; the fallback branch picks
                                  non_special[int(u_tok[t, i, pos, 0] * n)]
and its backbone is a stand-in that returns seeded pseudo-random logits, so the fixture pins
the sampler logic alone (constraints, ordering, in-step updates, repairs).

Stored per case: the configuration and seeds (inputs are regenerated from
numpy.random.RandomState, whose stream is frozen), the ids at the start of every step, the
final ids after the post-loop repairs and the edited last-step logits.
"""
import argparse
import hashlib
import os
import sys
import importlib.util
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "discrete-diffusion-model-for-inverse-polymer-design_amd"))
sys.path.insert(0, ROOT)
from tests.golden import replay_inputs as RI  # noqa: E402  (shared input generators)

REF = "/root/reference"
VARIANT_DIR = {"smiles": "Bi_Diffusion_SMILES", "selfies": "Bi_Diffusion_SELFIES",
               "group": "Bi_Diffusion_Group_SELFIES"}


def import_reference_sampler(variant):
    vdir = os.path.join(REF, VARIANT_DIR[variant])
    if variant == "group":
        import types
        for name, path in (("src", vdir + "/src"), ("src.utils", vdir + "/src/utils")):
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
        chem = types.ModuleType("src.utils.chemistry")
        chem.count_stars = lambda s: s.count("*")
        chem.check_validity = lambda s: True
        sys.modules["src.utils.chemistry"] = chem
    sys.path[:0] = [vdir, REF]
    from src.sampling.sampler import ConstrainedSampler
    return ConstrainedSampler, vdir


def reference_tokenizer(variant, vdir):
    from bidiff.utils import synthetic
    if variant == "smiles":
        spec = importlib.util.spec_from_file_location("ref_tok", vdir + "/src/data/tokenizer.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.PSmilesTokenizer(vocab=synthetic.smiles_vocab(), max_length=256)
    if variant == "selfies":
        return synthetic.selfies_tokenizer()
    return synthetic.group_selfies_tokenizer()


class FakeDiffusion:
    """diffusion_model stand-in: .eval() and .backbone(ids, timesteps, attention_mask)."""

    def __init__(self, case, V, log):
        self.case, self.V, self.log = case, V, log
        self.backbone = self._backbone

    def eval(self):
        return self

    def _backbone(self, ids, timesteps, attention_mask):
        t = int(timesteps[0].item())
        B, L = ids.shape
        self.log["ids_before"][t] = ids.clone().numpy()
        self.log["attn_before"][t] = attention_mask.clone().numpy()
        return torch.from_numpy(RI.step_logits(self.case, t, B, L, self.V))


def run_case(variant, Sampler, tok, case):
    V = tok.vocab_size
    T = case["T"]
    log = {"ids_before": {}, "attn_before": {}, "final_logits": None}
    model = FakeDiffusion(case, V, log)
    smp = Sampler(model, tok, num_steps=T, temperature=case["temperature"],
                  use_constraints=case["use_constraints"], device="cpu")
    B, L = case["B"], case["L"]
    u_sel = {t: RI.step_u_sel(case, t, B, L) for t in range(1, T + 1)}
    u_tok = {t: RI.step_u_tok(case, t, B, L, V) for t in range(1, T + 1)}
    non_special = list(smp.non_special_token_ids)

    def fake_randperm(n, *a, **k):
        fr = sys._getframe(1)
        while "masked_pos" not in fr.f_locals:
            fr = fr.f_back
        i, t = fr.f_locals["i"], fr.f_locals["t"]
        mp = fr.f_locals["masked_pos"].numpy()
        assert len(mp) == n
        keys = u_sel[t][i, mp]
        return torch.from_numpy(np.argsort(keys, kind="stable").astype(np.int64))

    def fake_multinomial(probs_row, num, *a, **k):
        fr = sys._getframe(1)             # _sample_token_with_fallback
        fell_back = "fallback" in fr.f_locals
        fr2 = fr.f_back                    # the sampling loop
        i, t = fr2.f_locals["i"], fr2.f_locals["t"]
        pos = int(fr2.f_locals["pos"])
        if fell_back:
            u0 = np.float32(u_tok[t][i, pos, 0])
            j = int(u0 * np.float32(len(non_special)))
            j = min(max(j, 0), len(non_special) - 1)
            return torch.tensor([non_special[j]])
        row = fr2.f_locals["logits"][i, pos].numpy().astype(np.float32)
        sc = np.full(V, -np.inf, dtype=np.float32)
        ok = row > -np.inf
        sc[ok] = row[ok] + RI.gumbel_np(u_tok[t][i, pos][ok])
        assert ok.any()
        return torch.tensor([int(np.argmax(sc))])

    # grab the logits handed to the first repair function = final_logits
    fix_name = "_fix_star_count" if variant == "smiles" else "_fix_placeholder_count"
    orig_fix = getattr(smp, fix_name)

    def spy_fix(ids, logits, *a, **k):
        log["final_logits"] = logits.clone().numpy()
        log["ids_before_fix"] = ids.clone().numpy()
        return orig_fix(ids, logits, *a, **k)
    setattr(smp, fix_name, spy_fix)

    kind = case["kind"]
    with mock.patch("torch.randperm", fake_randperm), mock.patch("torch.multinomial", fake_multinomial):
        if kind == "sample":
            kw = {}
            if variant == "selfies":
                kw["allow_natural_eos"] = bool(case.get("natural_eos", False))
            ids, strings = smp.sample(B, L, show_progress=False, **kw)
        elif kind == "lengths":
            ids, strings = smp.sample_with_lengths(case["lengths"], max_length=L, show_progress=False)
        elif kind == "conditional":
            kw = {}
            if variant == "selfies":
                kw["allow_natural_eos"] = bool(case.get("natural_eos", False))
            pre, suf = case.get("prefix"), case.get("suffix")
            ids, strings = smp.sample_conditional(
                B, L, prefix_ids=None if not pre else torch.tensor(pre).repeat(B, 1),
                suffix_ids=None if not suf else torch.tensor(suf).repeat(B, 1), show_progress=False, **kw)
        else:
            raise ValueError(kind)
    out = {
        "final_ids": ids.numpy().astype(np.int64),
        "ids_before": np.stack([log["ids_before"][t] for t in range(T, 0, -1)]).astype(np.int64),
        "attn_before": np.stack([log["attn_before"][t] for t in range(T, 0, -1)]).astype(np.int64),
    }
    if log["final_logits"] is not None:
        out["final_logits"] = log["final_logits"].astype(np.float32)
        out["ids_before_fix"] = log["ids_before_fix"].astype(np.int64)
    out["strings_sha"] = np.frombuffer(hashlib.sha256("\n".join(strings).encode()).digest(), dtype=np.uint8)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", required=True, choices=list(VARIANT_DIR))
    args = ap.parse_args()
    Sampler, vdir = import_reference_sampler(args.variant)
    tok = reference_tokenizer(args.variant, vdir)
    cases = RI.CASES[args.variant]
    blob = {}
    for name, case in cases.items():
        res = run_case(args.variant, Sampler, tok, case)
        for k, v in res.items():
            blob[f"{name}/{k}"] = v
        print(args.variant, name, "final ids sha", hashlib.sha256(res["final_ids"].tobytes()).hexdigest()[:16])
    path = os.path.join(HERE, f"sampler_{args.variant}.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
