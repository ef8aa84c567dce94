"""Golden vectors for id -> string decoding, captured from the imported reference tokenizers.

    python tests/golden/gen_decode_fixtures.py        # writes tests/golden/decode.json

The reference's SMILES and SMILES_BPE ``PSmilesTokenizer`` classes (src/data/tokenizer.py) are instantiated on this
repository's synthetic vocabularies; ``batch_decode(rows, skip_special_tokens=True)`` is what the samplers call
(src/sampling/sampler.py:724). Rows: random ids, special tokens in the middle, ids outside the vocabulary, an empty
row, a row of specials only. The SELFIES tokenizer imports the ``selfies`` package, which is not installed here; its
decode is the same loop (selfies_tokenizer.py:163-180,61-77) and is covered by the restatement only."""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("BIDIFF_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "discrete-diffusion-model-for-inverse-polymer-design_amd"))
from bidiff.utils import synthetic  # noqa: E402


def ref_tokenizer(variant_dir, vocab):
    spec = importlib.util.spec_from_file_location("ref_tok_" + variant_dir, os.path.join(REF, variant_dir, "src/data/tokenizer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.PSmilesTokenizer(vocab=vocab, max_length=256)


def rows_for(V, rs):
    rows = [rs.randint(0, V, size=n).tolist() for n in (1, 7, 64, 128, 200)]
    rows.append([2] + rs.randint(5, V, size=30).tolist() + [3] + [0] * 10)      # BOS ... EOS PAD PAD
    rows.append([5, 1, 6, 1, 1, 7, 4, 8])                                        # MASK and UNK in the middle
    rows.append([V, V + 100, 5, -1 if False else V + 1, 6])                      # ids outside the vocabulary
    rows.append([])
    rows.append([0, 1, 2, 3, 4])
    rows.append(list(range(V)))
    return rows


def main():
    rs = np.random.RandomState(20260227)
    out = {}
    for name, vdir, vocab in (("smiles", "Bi_Diffusion_SMILES", synthetic.smiles_vocab()),
                              ("bpe", "Bi_Diffusion_SMILES_BPE", synthetic.bpe_vocab(803))):
        tok = ref_tokenizer(vdir, vocab)
        rows = rows_for(len(vocab), rs)
        out[name] = {"vocab": vocab, "special_tokens": list(tok.SPECIAL_TOKENS), "rows": rows,
                     "expected": tok.batch_decode(rows, skip_special_tokens=True)}
    with open(os.path.join(HERE, "decode.json"), "w") as fh:
        json.dump(out, fh)
    print({k: len(v["rows"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
