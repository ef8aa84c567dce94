"""Pre-tokenised training set kept resident in HBM.

The reference tokenises every sample with Python regexes inside ``PolymerDataset.__getitem__``
(src/data/dataset.py:66-84) on each access; at GPU step times of a few milliseconds that is the
bottleneck. Here the strings are tokenised ONCE on the host with the reference's own tokenizer
(``encode(..., padding=False)``: BOS + tokens + EOS, cut to ``max_length`` with a final EOS,
tokenizer.py:174-188), stored as one flat uint16 array plus int64 offsets (2 bytes per token: a
million polymers of ~40 tokens is 80 MB of a 288 GB device), uploaded once, and every batch is
assembled on the device by ``bd_collate_batch`` from the batch's row indices.
"""
from typing import Iterable, Optional

import numpy as np
import torch


class TokenStore:
    def __init__(self, tokens: np.ndarray, offsets: np.ndarray, pad_token_id: int, max_length: int):
        tokens = np.ascontiguousarray(tokens, dtype=np.uint16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if offsets.ndim != 1 or offsets.size < 2 or offsets[0] != 0 or offsets[-1] != tokens.size:
            raise ValueError("TokenStore: offsets must start at 0, end at len(tokens) and hold at least one row")
        if np.any(np.diff(offsets) <= 0):
            raise ValueError("TokenStore: every row needs at least one token")
        self.tokens, self.offsets = tokens, offsets
        self.pad_token_id, self.max_length = int(pad_token_id), int(max_length)
        self.lengths = np.diff(offsets).astype(np.int64)
        if int(self.lengths.max()) > self.max_length:
            raise ValueError("TokenStore: a row is longer than max_length")
        self._dev = None

    def __len__(self) -> int:
        return int(self.offsets.size - 1)

    @classmethod
    def from_strings(cls, strings: Iterable[str], tokenizer, max_length: Optional[int] = None) -> "TokenStore":
        """Tokenise with the given (reference) tokenizer; ``max_length`` overrides the tokenizer's, as
        ``PolymerDataset(max_length=...)`` does (dataset.py:41-42)."""
        if max_length:
            tokenizer.max_length = max_length
        if tokenizer.vocab_size > 65535:
            raise ValueError("TokenStore holds uint16 token ids: vocabulary too large")
        rows, offsets = [], [0]
        for s in strings:
            ids = tokenizer.encode(s, add_special_tokens=True, padding=False, return_attention_mask=False)["input_ids"]
            rows.append(np.asarray(ids, dtype=np.uint16))
            offsets.append(offsets[-1] + len(ids))
        if not rows:
            raise ValueError("TokenStore: no samples")
        return cls(np.concatenate(rows), np.asarray(offsets, dtype=np.int64), tokenizer.pad_token_id,
                   tokenizer.max_length)

    def row(self, i: int) -> np.ndarray:
        return self.tokens[self.offsets[i]:self.offsets[i + 1]]

    def save(self, path: str) -> None:
        np.savez(path, tokens=self.tokens, offsets=self.offsets,
                 meta=np.asarray([self.pad_token_id, self.max_length], dtype=np.int64))

    @classmethod
    def load(cls, path: str) -> "TokenStore":
        z = np.load(path, allow_pickle=False)
        return cls(z["tokens"], z["offsets"], int(z["meta"][0]), int(z["meta"][1]))

    def to_device(self, device):
        """(tokens uint16 viewed as int16 storage, offsets int64) on the device; uploaded once."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("bidiff TokenStore: batches are assembled on a ROCm device (no CPU path)")
        if self._dev is None or self._dev[0] != device:
            tok = torch.from_numpy(self.tokens.view(np.int16)).to(device)
            off = torch.from_numpy(self.offsets).to(device)
            self._dev = (device, tok, off)
        return self._dev[1], self._dev[2]
