"""CPU restatement of the reference tokenizers' ``decode`` / ``batch_decode`` (test infrastructure only: nothing
in the product imports it).

Follows Bi_Diffusion_SMILES/src/data/tokenizer.py:207-224 (decode: id -> token through ``id_to_token`` with
'[UNK]' for unknown ids, special tokens skipped) and :110-121 (detokenize: special tokens filtered, ''.join);
the same loops stand in Bi_Diffusion_SMILES_BPE/src/data/tokenizer.py:250-258,119-121 and
Bi_Diffusion_SELFIES/src/data/selfies_tokenizer.py:163-180,61-77. Pinned by tests/golden/decode.json, written by
tests/golden/gen_decode_fixtures.py from the imported reference tokenizer."""
from typing import Dict, Iterable, List, Sequence


def decode(ids: Iterable[int], id_to_token: Dict[int, str], special_tokens: Sequence[str],
           skip_special_tokens: bool = True) -> str:
    tokens = []
    for i in ids:
        tok = id_to_token.get(int(i), "[UNK]")
        if skip_special_tokens and tok in special_tokens:
            continue
        tokens.append(tok)
    # detokenize filters the special tokens once more, so they never reach the string
    return "".join(t for t in tokens if t not in special_tokens)


def batch_decode(ids_list, id_to_token, special_tokens, skip_special_tokens: bool = True) -> List[str]:
    return [decode(ids, id_to_token, special_tokens, skip_special_tokens) for ids in ids_list]
