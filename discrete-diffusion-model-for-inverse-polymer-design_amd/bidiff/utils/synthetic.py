"""Synthetic vocabularies and duck-typed tokenizers for benchmarks, smoke runs and tests.

The reference ships no data, vocabulary file or checkpoint, and its SELFIES / Group-SELFIES
tokenizers need packages that are not installable here, so throughput runs use random-init
weights and these stand-ins. They expose exactly what the samplers read from a tokenizer:
``vocab`` (token -> id), ``vocab_size``, ``SPECIAL_TOKENS``, the five special-token ids,
``get_star_token_id()`` / ``get_placeholder_token_id()``, ``group_smiles`` and
``batch_decode``. Ids follow the reference's convention: the five specials first
(``[PAD]=0 [MASK]=1 [BOS]=2 [EOS]=3 [UNK]=4``), then the remaining tokens in sorted order.
"""
from typing import Dict, List, Optional

SPECIAL_TOKENS = ["[PAD]", "[MASK]", "[BOS]", "[EOS]", "[UNK]"]

# 34 p-SMILES tokens -> V = 39 (every constraint class is populated)
SMILES_TOKENS = [
    "#", "%10", "%11", "(", ")", "*", "-", "/", "1", "2", "3", "4", "5", "6", "=", "\\",
    "B", "Br", "C", "Cl", "F", "I", "N", "O", "P", "S", "Si", "c", "n", "o", "s",
    "[C@@H]", "[C@H]", "[nH]",
]


def _vocab(tokens: List[str]) -> Dict[str, int]:
    vocab = {tok: i for i, tok in enumerate(SPECIAL_TOKENS)}
    for tok in sorted(set(tokens)):
        if tok not in vocab:
            vocab[tok] = len(vocab)
    return vocab


def smiles_vocab() -> Dict[str, int]:
    return _vocab(SMILES_TOKENS)


def bpe_vocab(size: int = 803) -> Dict[str, int]:
    """SMILES base tokens plus synthetic merged tokens up to ``size`` entries."""
    toks = list(SMILES_TOKENS)
    i = 0
    while len(SPECIAL_TOKENS) + len(set(toks)) < size:
        toks.append(f"C{i}c")   # merged-token stand-ins; never parenthesis / bond / ring / star
        i += 1
    return _vocab(toks)


def selfies_vocab(size: int = 43) -> Dict[str, int]:
    toks = ["[I+3]", "[C]", "[=C]", "[#C]", "[N]", "[=N]", "[O]", "[=O]", "[S]", "[F]", "[Cl]", "[Br]",
            "[Branch1]", "[Branch2]", "[=Branch1]", "[Ring1]", "[Ring2]", "[P]", "[Si]", "[NH1]"]
    i = 0
    while len(SPECIAL_TOKENS) + len(set(toks)) < size:
        toks.append(f"[X{i}]")
        i += 1
    return _vocab(toks)


def group_selfies_vocab(size: int = 233, n_groups: int = 24):
    """Returns (vocab, group_smiles): group g < n_groups has a '[:/0G<g>]' token; groups 0..3
    carry one '[I+3]' each and group 4 carries two."""
    toks = ["[IH0+3]", "[C]", "[=C]", "[N]", "[O]", "[=O]", "[S]", "[F]", "[Branch]", "[Ring1]", "[pop]"]
    group_smiles = []
    for g in range(n_groups):
        toks.append(f"[:/0G{g}]")
        if g < 4:
            group_smiles.append("C[I+3]")
        elif g == 4:
            group_smiles.append("[I+3]CC[I+3]")
        else:
            group_smiles.append("c1ccccc1")
    i = 0
    while len(SPECIAL_TOKENS) + len(set(toks)) < size:
        toks.append(f"[Y{i}]")
        i += 1
    return _vocab(toks), group_smiles


class SyntheticTokenizer:
    """Minimal tokenizer exposing the attributes the samplers use; decode joins tokens."""

    SPECIAL_TOKENS = SPECIAL_TOKENS

    def __init__(self, vocab: Dict[str, int], star_token: Optional[str] = "*",
                 placeholder_token: Optional[str] = None, group_smiles: Optional[List[str]] = None,
                 max_length: int = 128):
        self.vocab = dict(vocab)
        self.id_to_token = {v: k for k, v in self.vocab.items()}
        self.max_length = max_length
        self._star = star_token
        self._placeholder = placeholder_token
        self.group_smiles = group_smiles or []

    @property
    def vocab_size(self) -> int:
        return len(self.vocab)

    @property
    def pad_token_id(self) -> int:
        return self.vocab["[PAD]"]

    @property
    def mask_token_id(self) -> int:
        return self.vocab["[MASK]"]

    @property
    def bos_token_id(self) -> int:
        return self.vocab["[BOS]"]

    @property
    def eos_token_id(self) -> int:
        return self.vocab["[EOS]"]

    @property
    def unk_token_id(self) -> int:
        return self.vocab["[UNK]"]

    def get_star_token_id(self) -> int:
        return self.vocab.get(self._star, self.unk_token_id)

    def get_placeholder_token_id(self):
        if self._placeholder is None:
            return None
        return self.vocab.get(self._placeholder)

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        out = []
        for i in ids:
            tok = self.id_to_token.get(int(i), "[UNK]")
            if skip_special_tokens and tok in self.SPECIAL_TOKENS:
                continue
            out.append(tok)
        return "".join(out)

    def batch_decode(self, ids_list, skip_special_tokens: bool = True) -> List[str]:
        return [self.decode(ids, skip_special_tokens) for ids in ids_list]


def smiles_tokenizer() -> SyntheticTokenizer:
    return SyntheticTokenizer(smiles_vocab(), star_token="*")


def bpe_tokenizer(size: int = 803) -> SyntheticTokenizer:
    return SyntheticTokenizer(bpe_vocab(size), star_token="*")


def selfies_tokenizer(size: int = 43) -> SyntheticTokenizer:
    return SyntheticTokenizer(selfies_vocab(size), star_token=None, placeholder_token="[I+3]")


def group_selfies_tokenizer(size: int = 233) -> SyntheticTokenizer:
    vocab, groups = group_selfies_vocab(size)
    return SyntheticTokenizer(vocab, star_token=None, placeholder_token="[IH0+3]", group_smiles=groups)
