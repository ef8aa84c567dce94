"""ConstrainedSampler for p-SMILES / SMILES-BPE token sequences.

Drop-in for ``Bi_Diffusion_SMILES/src/sampling/sampler.py`` (identical in
``Bi_Diffusion_SMILES_BPE``): same constructor, attributes and ``sample*`` methods, returning
``(ids LongTensor on the device, list[str])``. The reverse loop, the SMILES syntax
constraints and the post-loop repairs run on the device (see ``core.ReverseSampler``);
decoding stays with the host tokenizer.
"""
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import core
from .core import run_length_pieces
from .decode import decode_batch


def _progress(show, desc):
    if not show:
        return None
    try:
        from tqdm import tqdm
    except Exception:  # tqdm is optional
        return None
    return lambda it: tqdm(it, desc=desc)


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

        vocab = tokenizer.vocab
        self.mask_id = tokenizer.mask_token_id
        self.pad_id = tokenizer.pad_token_id
        self.bos_id = tokenizer.bos_token_id
        self.eos_id = tokenizer.eos_token_id
        self.star_id = tokenizer.get_star_token_id()

        # token classes used by the syntax constraints (reference sampler.py:52-79)
        self.open_paren_id = vocab.get("(", -1)
        self.close_paren_id = vocab.get(")", -1)
        self.bond_ids = {vocab.get(b, -1) for b in ["-", "=", "#", "/", "\\"]} - {-1}
        self.ring_digit_ids = {vocab.get(str(dg), -1) for dg in range(10)} - {-1}
        self.ring_percent_ids = {v for k, v in vocab.items() if k.startswith("%")}
        self.bracket_token_ids = {
            v for k, v in vocab.items()
            if k.startswith("[") and k.endswith("]") and k not in tokenizer.SPECIAL_TOKENS
        }
        special = {self.mask_id, self.pad_id, self.bos_id, self.eos_id}
        self.non_special_token_ids = [i for i in range(tokenizer.vocab_size) if i not in special]
        if not self.non_special_token_ids:
            self.non_special_token_ids = list(range(tokenizer.vocab_size))

        self._core = core.ReverseSampler(diffusion_model, self._build_tables(), num_steps, temperature,
                                         use_constraints, device)

    # ------------------------------------------------------------------ tables
    def _build_tables(self) -> core.TokenTables:
        V = int(self.diffusion_model.backbone.vocab_size)
        flags = np.zeros(V, dtype=np.uint8)
        for tok in (self.mask_id, self.pad_id, self.bos_id, self.eos_id):
            if 0 <= tok < V:
                flags[tok] |= core.TF_SPECIAL
        contrib = np.zeros(V, dtype=np.int32)
        if 0 <= self.star_id < V:
            contrib[self.star_id] = 1
        tt = core.TokenTables(
            V=V, mask_id=int(self.mask_id), pad_id=int(self.pad_id), bos_id=int(self.bos_id),
            eos_id=int(self.eos_id), open_id=int(self.open_paren_id), close_id=int(self.close_paren_id),
            count_id=int(self.star_id), variant=core.VB_FIXED | core.VB_SYNTAX, tok_flags=flags,
            contrib=contrib, ring_ids=sorted(self.ring_digit_ids | self.ring_percent_ids),
            bond_ids=sorted(self.bond_ids), non_special=list(self.non_special_token_ids),
            fix_excl=[int(self.mask_id), int(self.pad_id)], fix_syntax=True)
        return tt.finish()

    # ------------------------------------------------------------------ core loop
    def _sample_from_ids(self, ids, attention_mask, fixed_mask, show_progress: bool = True,
                         seq_offset: int = 0) -> Tuple[torch.Tensor, List[str]]:
        """T reverse steps from the given partially masked ids, then the star / parenthesis /
        ring repairs and decoding (reference sampler.py:634-726)."""
        self._core.num_steps = int(self.num_steps)
        self._core.temperature = float(self.temperature)
        self._core.use_constraints = bool(self.use_constraints)
        ids = self._core.run(ids, attention_mask, fixed_mask, seq_offset=seq_offset,
                             progress=_progress(show_progress, "Sampling"))
        smiles = decode_batch(self, ids)
        return ids, smiles

    # ------------------------------------------------------------------ initial states
    def initial_state_lengths(self, lengths: List[int], max_length: Optional[int] = None):
        """ids / attention_mask / fixed_mask for per-sample lengths that include BOS and EOS
        (reference sampler.py:747-773)."""
        lengths = [max(2, int(l)) for l in lengths]
        seq_length = max(lengths)
        if max_length is not None and seq_length > max_length:
            raise ValueError(f"Max length {max_length} is smaller than required {seq_length}")
        n = len(lengths)
        lens = torch.tensor(lengths, dtype=torch.long)
        pos = torch.arange(seq_length).unsqueeze(0)
        ids = torch.full((n, seq_length), self.mask_id, dtype=torch.long)
        ids[pos >= lens.unsqueeze(1)] = self.pad_id
        ids[:, 0] = self.bos_id
        ids[torch.arange(n), lens - 1] = self.eos_id
        attention_mask = (pos < lens.unsqueeze(1)).long()
        fixed_mask = ids != self.mask_id
        return ids.to(self.device), attention_mask.to(self.device), fixed_mask.to(self.device)

    def initial_state(self, batch_size: int, seq_length: int):
        ids = torch.full((batch_size, seq_length), self.mask_id, dtype=torch.long, device=self.device)
        ids[:, 0] = self.bos_id
        ids[:, -1] = self.eos_id
        return ids, torch.ones_like(ids), ids != self.mask_id

    # ------------------------------------------------------------------ reference API
    def sample_with_lengths(self, lengths: List[int], max_length: Optional[int] = None,
                            show_progress: bool = True, seq_offset: int = 0):
        if not lengths:
            return torch.empty((0, 0), dtype=torch.long, device=self.device), []
        ids, am, fm = self.initial_state_lengths(lengths, max_length)
        return self._sample_from_ids(ids, am, fm, show_progress, seq_offset)

    def sample(self, batch_size: int, seq_length: int, show_progress: bool = True, seq_offset: int = 0):
        self.diffusion_model.eval()
        ids, am, fm = self.initial_state(batch_size, seq_length)
        return self._sample_from_ids(ids, am, fm, show_progress, seq_offset)

    def sample_batch(self, num_samples: int, seq_length: int, batch_size: int = 256,
                     show_progress: bool = True, lengths: Optional[List[int]] = None,
                     seq_offset: int = 0):
        """Independent batches; ``seq_offset`` numbers the sequences globally so that a job
        split over several processes draws the same random streams as a single process."""
        if lengths is not None and len(lengths) != num_samples:
            raise ValueError("lengths must match num_samples")
        all_ids, all_smiles = [], []
        idx = 0
        while idx < num_samples:
            cur = min(batch_size, num_samples - idx)
            if lengths is None:
                ids, smiles = self.sample(cur, seq_length, show_progress=False, seq_offset=seq_offset + idx)
            else:
                ids, smiles = self.sample_with_lengths(lengths[idx:idx + cur], max_length=seq_length,
                                                       show_progress=False, seq_offset=seq_offset + idx)
            all_ids.append(ids)
            all_smiles.extend(smiles)
            idx += cur
        return all_ids, all_smiles

    def sample_variable_length(self, num_samples: int, length_range: Tuple[int, int] = (20, 100),
                               batch_size: int = 256, samples_per_length: int = 16,
                               show_progress: bool = True):
        lo, hi = length_range
        inner = min(batch_size, samples_per_length)
        all_ids, all_smiles = [], []
        if not getattr(self, "merge_variable_length", True):
            while len(all_smiles) < num_samples:
                cur = min(inner, num_samples - len(all_smiles))
                seq_length = int(np.random.randint(lo, hi + 1))
                ids, smiles = self.sample(cur, seq_length, show_progress=False, seq_offset=len(all_smiles))
                all_ids.append(ids)
                all_smiles.extend(smiles)
            return all_ids, all_smiles
        # The reference runs one reverse loop per group of `samples_per_length` sequences (sampler.py:868-911): a
        # handful of workgroups per launch. Here the groups' lengths are drawn first (the same np.random calls in the
        # same order), and up to `batch_size` sequences of different lengths go through ONE loop as a batch padded to
        # its longest row, the forward dealing them to its waves by length. A row's draws depend on its global number
        # and its positions only, so with a fixed seed this returns what the group-by-group loop returns.
        pieces, n = [], 0
        while n < num_samples:
            cur = min(inner, num_samples - n)
            pieces.append((cur, max(2, int(np.random.randint(lo, hi + 1)))))
            n += cur
        return run_length_pieces(pieces, batch_size, lambda lengths, off: self.sample_with_lengths(
            lengths, show_progress=False, seq_offset=off))

    def sample_conditional(self, batch_size: int, seq_length: int,
                           prefix_ids: Optional[torch.Tensor] = None,
                           suffix_ids: Optional[torch.Tensor] = None, show_progress: bool = True):
        """Prefix after BOS and suffix before EOS are fixed (reference sampler.py:913-960)."""
        ids = torch.full((batch_size, seq_length), self.mask_id, dtype=torch.long, device=self.device)
        ids[:, 0] = self.bos_id
        ids[:, -1] = self.eos_id
        fixed = torch.zeros_like(ids, dtype=torch.bool)
        fixed[:, 0] = True
        fixed[:, -1] = True
        if prefix_ids is not None:
            n = prefix_ids.shape[1]
            ids[:, 1:1 + n] = prefix_ids.to(self.device)
            fixed[:, 1:1 + n] = True
        if suffix_ids is not None:
            n = suffix_ids.shape[1]
            ids[:, -1 - n:-1] = suffix_ids.to(self.device)
            fixed[:, -1 - n:-1] = True
        return self._sample_from_ids(ids, torch.ones_like(ids), fixed, show_progress)
