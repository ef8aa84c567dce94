"""ConstrainedSampler for SELFIES token sequences.

Drop-in for ``Bi_Diffusion_SELFIES/src/sampling/sampler.py``: the only chemistry constraint
is "exactly two ``[I+3]`` placeholders" (forbid a third, boost the placeholder when few
masked slots remain), chosen positions are drawn left to right, and EOS is either fixed at
the last position (default) or sampled (``allow_natural_eos=True``).
"""
from collections import defaultdict
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import core
from .core import run_length_pieces
from .decode import decode_batch
from .sampler import _progress


class ConstrainedSampler:
    def __init__(self, diffusion_model, tokenizer, num_steps: int = 100, temperature: float = 1.0,
                 use_constraints: bool = True, device: str = "cuda"):
        self.diffusion_model = diffusion_model
        self.tokenizer = tokenizer
        self.num_steps = num_steps
        self.temperature = temperature
        self.use_constraints = use_constraints
        self.device = device
        if self.temperature <= 0:
            raise ValueError(f"temperature must be > 0, got {self.temperature}")
        self.mask_id = tokenizer.mask_token_id
        self.pad_id = tokenizer.pad_token_id
        self.bos_id = tokenizer.bos_token_id
        self.eos_id = tokenizer.eos_token_id
        self.placeholder_id = tokenizer.get_placeholder_token_id()
        special = {self.mask_id, self.pad_id, self.bos_id, self.eos_id}
        self.non_special_token_ids = [i for i in range(tokenizer.vocab_size) if i not in special]
        if not self.non_special_token_ids:
            self.non_special_token_ids = list(range(tokenizer.vocab_size))
        self._core = core.ReverseSampler(diffusion_model, self._build_tables(), num_steps, temperature,
                                         use_constraints, device)

    def _build_tables(self) -> core.TokenTables:
        V = int(self.diffusion_model.backbone.vocab_size)
        flags = np.zeros(V, dtype=np.uint8)
        for tok in (self.mask_id, self.pad_id, self.bos_id):
            if 0 <= tok < V:
                flags[tok] |= core.TF_SPECIAL
        if 0 <= self.eos_id < V:
            flags[self.eos_id] |= core.TF_EOS
        contrib = np.zeros(V, dtype=np.int32)
        contrib[self.placeholder_id] = 1
        return core.TokenTables(
            V=V, mask_id=int(self.mask_id), pad_id=int(self.pad_id), bos_id=int(self.bos_id),
            eos_id=int(self.eos_id), boost_id=int(self.placeholder_id), count_id=int(self.placeholder_id),
            variant=core.VB_URGENCY | core.VB_SORTED, tok_flags=flags, contrib=contrib,
            non_special=list(self.non_special_token_ids),
            fix_excl=[int(self.mask_id), int(self.pad_id)], fix_syntax=False).finish()

    def sample(self, batch_size: int, seq_length: int, show_progress: bool = True,
               allow_natural_eos: bool = False, seq_offset: int = 0) -> Tuple[torch.Tensor, List[str]]:
        """All positions but BOS (and, by default, a fixed final EOS) start masked
        (reference SELFIES sampler.py:243-419)."""
        self.diffusion_model.eval()
        ids = torch.full((batch_size, seq_length), self.mask_id, dtype=torch.long, device=self.device)
        ids[:, 0] = self.bos_id
        has_eos = torch.zeros(batch_size, dtype=torch.uint8, device=self.device)
        if not allow_natural_eos:
            ids[:, -1] = self.eos_id
            has_eos[:] = 1
        attention_mask = torch.ones_like(ids)
        self._core.num_steps = int(self.num_steps)
        self._core.temperature = float(self.temperature)
        self._core.use_constraints = bool(self.use_constraints)
        ids = self._core.run(ids, attention_mask, None, natural_eos=allow_natural_eos, seq_offset=seq_offset,
                             progress=_progress(show_progress, "Sampling"), has_eos=has_eos)
        strings = decode_batch(self, ids)
        return ids, strings

    def _sample_lengths(self, lengths: List[int], seq_offset: int = 0):
        """``sample`` (fixed final EOS) for rows of different lengths in one batch: row i is BOS, lengths[i] - 2
        masks, EOS, then PAD outside the attention mask."""
        self.diffusion_model.eval()
        n, width = len(lengths), max(lengths)
        lens = torch.tensor(lengths, dtype=torch.long)
        pos = torch.arange(width).unsqueeze(0)
        ids = torch.full((n, width), self.mask_id, dtype=torch.long)
        ids[pos >= lens.unsqueeze(1)] = self.pad_id
        ids[:, 0] = self.bos_id
        ids[torch.arange(n), lens - 1] = self.eos_id
        attn = (pos < lens.unsqueeze(1)).long()
        self._core.num_steps = int(self.num_steps)
        self._core.temperature = float(self.temperature)
        self._core.use_constraints = bool(self.use_constraints)
        ids = self._core.run(ids.to(self.device), attn.to(self.device), None, natural_eos=False, seq_offset=seq_offset,
                             has_eos=torch.ones(n, dtype=torch.uint8, device=self.device))
        return ids, decode_batch(self, ids)

    def sample_batch(self, num_samples: int, seq_length: int, batch_size: int = 256,
                     show_progress: bool = True, lengths: Optional[List[int]] = None):
        """Fixed-length batches, or - given ``lengths`` - one group of batches per distinct
        length with the strings returned in the caller's order (reference :421-502)."""
        all_ids, all_strings = [], []
        if lengths is None:
            while len(all_strings) < num_samples:
                cur = min(batch_size, num_samples - len(all_strings))
                ids, s = self.sample(cur, seq_length, show_progress=False, seq_offset=len(all_strings))
                all_ids.append(ids)
                all_strings.extend(s)
            return all_ids, all_strings
        groups = defaultdict(list)
        for i, ln in enumerate(lengths):
            groups[ln].append(i)
        result = [None] * len(lengths)
        if getattr(self, "merge_variable_length", True):
            # the same batches in the same order (one per distinct length and batch_size rows), but up to batch_size
            # rows of different lengths share a reverse loop instead of one loop per distinct length
            pieces, order = [], []
            for ln in sorted(groups):
                idxs = groups[ln]
                order.extend(idxs)
                for k in range(0, len(idxs), batch_size):
                    pieces.append((min(batch_size, len(idxs) - k), int(ln)))
            all_ids, strings = run_length_pieces(pieces, batch_size, self._sample_lengths)
            for glob, st in zip(order, strings):
                result[glob] = st
            return all_ids, result
        done = 0
        for ln in sorted(groups):
            idxs = groups[ln]
            got = []
            while len(got) < len(idxs):
                cur = min(batch_size, len(idxs) - len(got))
                ids, s = self.sample(cur, ln, show_progress=False, seq_offset=done)
                all_ids.append(ids)
                got.extend(s)
                done += cur
            for local, glob in enumerate(idxs):
                result[glob] = got[local]
        return all_ids, result

    def sample_variable_length(self, num_samples: int, length_range: Tuple[int, int] = (20, 100),
                               batch_size: int = 256, samples_per_length: int = 16,
                               show_progress: bool = True):
        lo, hi = length_range
        inner = min(batch_size, samples_per_length)
        all_ids, all_strings = [], []
        if getattr(self, "merge_variable_length", True):
            pieces, n = [], 0                      # the groups' lengths first: same np.random calls, same order
            while n < num_samples:
                cur = min(inner, num_samples - n)
                pieces.append((cur, int(np.random.randint(lo, hi + 1))))
                n += cur
            return run_length_pieces(pieces, batch_size, self._sample_lengths)
        while len(all_strings) < num_samples:
            cur = min(inner, num_samples - len(all_strings))
            ids, s = self.sample(cur, int(np.random.randint(lo, hi + 1)), show_progress=False,
                                 seq_offset=len(all_strings))
            all_ids.append(ids)
            all_strings.extend(s)
        return all_ids, all_strings

    def sample_conditional(self, batch_size: int, seq_length: int,
                           prefix_ids: Optional[torch.Tensor] = None,
                           suffix_ids: Optional[torch.Tensor] = None, show_progress: bool = True,
                           allow_natural_eos: bool = False, seq_offset: int = 0) -> Tuple[torch.Tensor, List[str]]:
        """Prefix after BOS and suffix before the final EOS stay fixed; a suffix forces the fixed-EOS
        mode (reference SELFIES sampler.py:553-737). Positions are chosen among the non-fixed masks,
        the special-token forbids apply to those positions only (:639-664) and the in-step
        placeholder update edits every remaining MASK (:703-710)."""
        self.diffusion_model.eval()
        natural = bool(allow_natural_eos) and suffix_ids is None
        ids = torch.full((batch_size, seq_length), self.mask_id, dtype=torch.long, device=self.device)
        ids[:, 0] = self.bos_id
        has_eos = torch.zeros(batch_size, dtype=torch.uint8, device=self.device)
        fixed = torch.zeros_like(ids, dtype=torch.bool)
        fixed[:, 0] = True
        if not natural:
            ids[:, -1] = self.eos_id
            fixed[:, -1] = True
            has_eos[:] = 1
        if prefix_ids is not None:
            n = prefix_ids.shape[1]
            ids[:, 1:1 + n] = prefix_ids.to(self.device)
            fixed[:, 1:1 + n] = True
        if suffix_ids is not None:
            n = suffix_ids.shape[1]
            ids[:, -1 - n:-1] = suffix_ids.to(self.device)
            fixed[:, -1 - n:-1] = True
        self._core.num_steps = int(self.num_steps)
        self._core.temperature = float(self.temperature)
        self._core.use_constraints = bool(self.use_constraints)
        extra = core.VB_FIXED | core.VB_DYN_ALL_MASKS | core.VB_SPECIAL_UNFIXED
        ids = self._core.run(ids, torch.ones_like(ids), fixed, natural_eos=natural, seq_offset=seq_offset,
                             progress=_progress(show_progress, "Sampling"), has_eos=has_eos, variant_extra=extra)
        strings = decode_batch(self, ids)
        return ids, strings
