"""Device-side ``batch_decode``: token ids -> strings without the per-id Python loop.

The reference samplers end with ``tokenizer.batch_decode(ids.cpu().tolist(), skip_special_tokens=True)``
(src/sampling/sampler.py:724): B * L dictionary lookups on the host, the serial tail of a 4096-sequence batch.
``DeviceDecoder`` uploads the vocabulary's texts once and lets ``bd_decode_ids`` join them per sequence; the host
only slices one byte buffer into B strings. It applies to tokenizers whose decode is the plain join of the
non-special tokens (SMILES, SMILES_BPE, SELFIES); ``for_tokenizer`` checks that on probe rows against the
tokenizer's own ``batch_decode`` and returns None otherwise (Group SELFIES, whose detokenize runs the group
grammar on the host)."""
import logging
import warnings
from typing import List, Optional

import numpy as np
import torch

from .. import _lib

_HOST_DECODE_LOGGED = set()


def _log_host_decode_once(name):
    """One line per tokenizer class whose decode is not the plain join (its own batch_decode keeps running on the host)."""
    if name not in _HOST_DECODE_LOGGED:
        _HOST_DECODE_LOGGED.add(name)
        logging.getLogger("bidiff").info("decode of %s stays on the host: not a plain join of the token texts", name)


class DeviceDecoder:
    def __init__(self, id_to_token, special_tokens, vocab_size: int, device):
        self.device = torch.device(device)
        self.V = int(vocab_size)
        special = set(special_tokens)
        texts = []
        for i in range(self.V):
            tok = id_to_token.get(i, "[UNK]")
            texts.append(b"" if tok in special else tok.encode("utf-8"))
        offs = np.zeros(self.V + 1, dtype=np.int32)
        offs[1:] = np.cumsum([len(t) for t in texts])
        blob = b"".join(texts) or b"\0"
        self.max_token_bytes = max(1, max(len(t) for t in texts))
        self._offs = torch.from_numpy(offs).to(self.device)
        self._text = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(self.device)

    @classmethod
    def for_tokenizer(cls, tokenizer, device) -> Optional["DeviceDecoder"]:
        """A decoder for ``tokenizer`` if its decode is the plain join (checked on probe rows), else None."""
        id_to_token = getattr(tokenizer, "id_to_token", None)
        special = getattr(tokenizer, "SPECIAL_TOKENS", None)
        if not isinstance(id_to_token, dict) or special is None:
            return None
        V = int(tokenizer.vocab_size)
        special = set(special)
        plain = lambda row: "".join(t for t in (id_to_token.get(int(i), "[UNK]") for i in row) if t not in special)
        # probe rows of ids the tokenizer knows (an id outside 0..V-1 decodes to nothing in the kernel; what a
        # tokenizer does with one - skip it, print [UNK], raise - is its own business and no sampler produces one)
        probes = [list(range(V)), list(range(V - 1, -1, -1)), [0, 1, 2, 3, 4 % V, V - 1, V - 1], []]
        try:
            same = tokenizer.batch_decode(probes, skip_special_tokens=True) == [plain(r) for r in probes]
        except Exception as exc:       # a tokenizer that cannot decode its own vocabulary: say so once, decode on the host
            warnings.warn(f"bidiff: device decode disabled, the tokenizer's batch_decode raised on probe rows: {exc!r}")
            return None
        if not same:
            _log_host_decode_once(type(tokenizer).__name__)
            return None
        return cls(id_to_token, special, V, device)

    def batch_decode(self, ids: torch.Tensor) -> List[str]:
        _lib.require_device(ids, "DeviceDecoder.batch_decode")
        ids = ids.to(torch.long).contiguous()
        B, L = ids.shape
        if B == 0:
            return []
        stride = L * self.max_token_bytes
        out = torch.empty(B, stride, dtype=torch.uint8, device=ids.device)
        lens = torch.empty(B, dtype=torch.int32, device=ids.device)
        _lib.check(_lib.lib().bd_decode_ids(_lib.ptr(ids), B, L, _lib.ptr(self._text), _lib.ptr(self._offs), self.V,
                                            _lib.ptr(out), _lib.ptr(lens), stride, _lib.stream_handle()), "bd_decode_ids")
        lens_h = lens.cpu().numpy()
        width = int(lens_h.max())
        buf = out[:, :max(width, 1)].cpu().numpy()
        return [buf[b, :lens_h[b]].tobytes().decode("utf-8") for b in range(B)]


def decode_batch(owner, ids: torch.Tensor) -> List[str]:
    """``owner.tokenizer.batch_decode(ids, skip_special_tokens=True)``, on the device when the tokenizer's decode
    is the plain join (decided once per owner), through the tokenizer itself otherwise."""
    if not ids.is_cuda:               # nothing is decided (or remembered) from host tensors
        return owner.tokenizer.batch_decode(ids.tolist(), skip_special_tokens=True)
    dec = getattr(owner, "_device_decoder", False)
    if dec is False:
        dec = DeviceDecoder.for_tokenizer(owner.tokenizer, ids.device)
        owner._device_decoder = dec
    if dec is None:
        return owner.tokenizer.batch_decode(ids.cpu().tolist(), skip_special_tokens=True)
    return dec.batch_decode(ids)
