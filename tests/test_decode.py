"""id -> string decoding: the restatement against the reference tokenizers' golden rows (CPU), the device kernel
against both (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import decode_ref

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "decode.json")) as fh:
    GOLD = json.load(fh)


@pytest.mark.parametrize("name", sorted(GOLD))
def test_oracle_decode_matches_reference_rows(name):
    g = GOLD[name]
    id_to_token = {v: k for k, v in g["vocab"].items()}
    assert decode_ref.batch_decode(g["rows"], id_to_token, g["special_tokens"]) == g["expected"]
    # the special tokens never reach the string, asked for or not (detokenize filters them again)
    assert decode_ref.batch_decode(g["rows"], id_to_token, g["special_tokens"], skip_special_tokens=False) == g["expected"]


def _padded(rows, pad):
    L = max(1, max(len(r) for r in rows))
    a = np.full((len(rows), L), pad, dtype=np.int64)
    for i, r in enumerate(rows):
        a[i, :len(r)] = r
    return a


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLD))
def test_device_decode_matches_reference_rows(gpu_device, name):
    from bidiff.sampling.decode import DeviceDecoder
    g = GOLD[name]
    id_to_token = {v: k for k, v in g["vocab"].items()}
    dec = DeviceDecoder(id_to_token, g["special_tokens"], len(g["vocab"]), gpu_device)
    ids = torch.from_numpy(_padded(g["rows"], 0)).to(gpu_device)       # [PAD] = 0 decodes to nothing
    assert dec.batch_decode(ids) == g["expected"]


@pytest.mark.gpu
@pytest.mark.parametrize("variant,B,L", [("smiles", 4096, 128), ("bpe", 512, 128), ("selfies", 300, 77), ("bpe", 3, 512)])
def test_device_decode_matches_tokenizer_at_batch_sizes(gpu_device, variant, B, L):
    from bidiff.sampling.decode import DeviceDecoder
    from bidiff.utils import synthetic
    tok = {"smiles": synthetic.smiles_tokenizer, "bpe": synthetic.bpe_tokenizer, "selfies": synthetic.selfies_tokenizer}[variant]()
    dec = DeviceDecoder.for_tokenizer(tok, gpu_device)
    assert dec is not None
    rs = np.random.RandomState(B + L)
    ids = rs.randint(0, tok.vocab_size + 2, size=(B, L)).astype(np.int64)       # specials and out-of-range ids included
    ids[0] = tok.pad_token_id                                                    # an empty string
    got = dec.batch_decode(torch.from_numpy(ids).to(gpu_device))
    want = decode_ref.batch_decode(ids.tolist(), tok.id_to_token, tok.SPECIAL_TOKENS)
    assert got == want
    assert got == tok.batch_decode(ids.tolist(), skip_special_tokens=True)
    assert got[0] == ""


def test_decoder_is_refused_for_a_tokenizer_with_its_own_detokenize():
    """A tokenizer whose decode is not the plain join (Group SELFIES runs its grammar) keeps its own batch_decode."""
    from bidiff.sampling.decode import DeviceDecoder
    from bidiff.utils import synthetic

    class Grammar(synthetic.SyntheticTokenizer):
        def batch_decode(self, ids_list, skip_special_tokens=True):
            return [s.upper() for s in super().batch_decode(ids_list, skip_special_tokens)]

    tok = Grammar(synthetic.smiles_vocab())
    assert DeviceDecoder.for_tokenizer(tok, "cpu") is None

    class NoTable:
        vocab_size = 5
    assert DeviceDecoder.for_tokenizer(NoTable(), "cpu") is None
