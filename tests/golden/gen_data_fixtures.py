"""Golden vectors for the training input pipeline, captured from the imported reference
(Bi_Diffusion_SMILES/src/data/{samplers,dataset,tokenizer}.py). Run in the build container only:
    python tests/golden/gen_data_fixtures.py
src/data/__init__.py pulls in rdkit (absent here), so the three modules are loaded under an empty
``src.data`` package object instead of through that __init__; nothing of them is changed.
"""
import importlib
import os
import sys
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "discrete-diffusion-model-for-inverse-polymer-design_amd"))
sys.path.insert(0, ROOT)
from tests.golden import replay_inputs as RI  # noqa: E402

V = "/root/reference/Bi_Diffusion_SMILES"
for name, path in (("src", V + "/src"), ("src.data", V + "/src/data")):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
samplers = importlib.import_module("src.data.samplers")
tokmod = importlib.import_module("src.data.tokenizer")
dataset = importlib.import_module("src.data.dataset")


def main():
    blob = {}
    for ci, (n, bs, drop, shuf, seed, mult, reps, rank, epoch) in enumerate(RI.SAMPLER_CASES):
        smp = samplers.LengthBucketBatchSampler(RI.sampler_lengths(n).tolist(), bs, drop_last=drop, shuffle=shuf, seed=seed,
                                                bucket_size_multiplier=mult, num_replicas=reps, rank=rank)
        smp.set_epoch(epoch)
        batches = list(smp)
        blob[f"sampler/{ci}/flat"] = np.asarray([i for b in batches for i in b], dtype=np.int64)
        blob[f"sampler/{ci}/sizes"] = np.asarray([len(b) for b in batches], dtype=np.int64)
        blob[f"sampler/{ci}/len"] = np.asarray(len(smp), dtype=np.int64)
    strings = RI.polymer_strings()
    tok = tokmod.PSmilesTokenizer(max_length=RI.POLYMER_MAX_LENGTH)
    tok.build_vocab(strings)
    blob["vocab_tokens"] = np.asarray(sorted(tok.vocab, key=tok.vocab.get))
    df = pd.DataFrame({"p_smiles": strings})
    ds_dyn = dataset.PolymerDataset(df, tok, pad_to_max_length=False)
    ds_fix = dataset.PolymerDataset(df, tok, pad_to_max_length=True)
    blob["lengths"] = np.asarray([int(ds_dyn[i]["input_ids"].shape[0]) for i in range(len(ds_dyn))], dtype=np.int64)
    smp = samplers.LengthBucketBatchSampler(blob["lengths"].tolist(), 16, seed=11, bucket_size_multiplier=2)
    for bi, batch in enumerate(smp):
        out = dataset.dynamic_collate_fn([ds_dyn[i] for i in batch], tok.pad_token_id)
        blob[f"dyn/{bi}/indices"] = np.asarray(batch, dtype=np.int64)
        blob[f"dyn/{bi}/input_ids"] = out["input_ids"].numpy()
        blob[f"dyn/{bi}/attention_mask"] = out["attention_mask"].numpy()
        if bi < 2:
            out = dataset.collate_fn([ds_fix[i] for i in batch])
            blob[f"fix/{bi}/input_ids"] = out["input_ids"].numpy()
            blob[f"fix/{bi}/attention_mask"] = out["attention_mask"].numpy()
    blob["n_dyn_batches"] = np.asarray(bi + 1)
    np.savez_compressed(os.path.join(HERE, "data.npz"), **blob)
    print("wrote", len(blob), "arrays; vocab", len(tok.vocab), "max len", int(blob["lengths"].max()))


if __name__ == "__main__":
    main()
