"""ConstrainedSampler for Group-SELFIES token sequences.

Drop-in for ``Bi_Diffusion_Group_SELFIES/src/sampling/sampler.py``: the placeholder
constraint sums per-token contributions (the direct placeholder token counts 1, a group
token counts the ``[I+3]`` occurrences of its group SMILES), UNK is forbidden together with
the other special tokens, and chosen positions are drawn in random order.
"""
import re
from typing import Callable, Dict, List, Optional, Tuple

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
        self.unk_id = tokenizer.unk_token_id
        self.placeholder_id = tokenizer.get_placeholder_token_id()
        if self.placeholder_id is None:
            for token, tid in tokenizer.vocab.items():
                if "I" in token and "+3" in token:
                    self.placeholder_id = tid
                    break
        if self.placeholder_id is None:
            raise ValueError("Placeholder token not found in tokenizer vocabulary; rebuild tokenizer "
                             "with the placeholder settings.")
        self.star_id = self.placeholder_id
        self.special_token_ids = {self.mask_id, self.pad_id, self.bos_id, self.eos_id, self.unk_id} - {None}
        self.non_special_token_ids = [i for i in range(tokenizer.vocab_size) if i not in self.special_token_ids]
        if not self.non_special_token_ids:
            self.non_special_token_ids = list(range(tokenizer.vocab_size))
        self.placeholder_contributions = self._build_placeholder_token_map()
        # host-side chemistry used by sample_with_rejection only (the reference imports count_stars /
        # check_validity from its utils.chemistry, Group sampler.py:10): resolved at first use, or
        # set by the caller
        self.count_stars_fn: Optional[Callable[[str], int]] = None
        self.check_validity_fn: Optional[Callable[[str], bool]] = None
        self._core = core.ReverseSampler(diffusion_model, self._build_tables(), num_steps, temperature,
                                         use_constraints, device)

    def _build_placeholder_token_map(self) -> Dict[int, int]:
        """token id -> number of placeholder atoms it decodes to (reference Group sampler :113-144)."""
        contrib = {self.placeholder_id: 1}
        groups = getattr(self.tokenizer, "group_smiles", None)
        if groups:
            for token, tid in self.tokenizer.vocab.items():
                m = re.match(r"\[:/(\d+)G(\d+)\]", token)
                if m:
                    gi = int(m.group(2))
                    if gi < len(groups):
                        cnt = groups[gi].count("[I+3]")
                        if cnt > 0:
                            contrib[tid] = cnt
        return contrib

    def _build_tables(self) -> core.TokenTables:
        V = int(self.diffusion_model.backbone.vocab_size)
        flags = np.zeros(V, dtype=np.uint8)
        for tok in self.special_token_ids:
            if tok is not None and 0 <= tok < V:
                flags[tok] |= core.TF_SPECIAL
        contrib = np.zeros(V, dtype=np.int32)
        for tid, cnt in self.placeholder_contributions.items():
            contrib[tid] = cnt
        return core.TokenTables(
            V=V, mask_id=int(self.mask_id), pad_id=int(self.pad_id), bos_id=int(self.bos_id),
            eos_id=int(self.eos_id), count_id=int(self.placeholder_id), variant=0, tok_flags=flags,
            contrib=contrib, non_special=list(self.non_special_token_ids),
            fix_excl=sorted(int(t) for t in self.special_token_ids if t is not None and t >= 0),
            fix_syntax=False).finish()

    def sample(self, batch_size: int, seq_length: int, show_progress: bool = True,
               seq_offset: int = 0) -> Tuple[torch.Tensor, List[str]]:
        self.diffusion_model.eval()
        ids = torch.full((batch_size, seq_length), self.mask_id, dtype=torch.long, device=self.device)
        ids[:, 0] = self.bos_id
        ids[:, -1] = self.eos_id
        self._core.num_steps = int(self.num_steps)
        self._core.temperature = float(self.temperature)
        self._core.use_constraints = bool(self.use_constraints)
        ids = self._core.run(ids, torch.ones_like(ids), None, seq_offset=seq_offset,
                             progress=_progress(show_progress, "Sampling"))
        strings = decode_batch(self, ids)
        return ids, strings

    def sample_batch(self, num_samples: int, seq_length: int, batch_size: int = 256,
                     show_progress: bool = True, seq_offset: int = 0):
        all_ids, all_strings = [], []
        while len(all_strings) < num_samples:
            cur = min(batch_size, num_samples - len(all_strings))
            ids, s = self.sample(cur, seq_length, show_progress=False, seq_offset=seq_offset + len(all_strings))
            all_ids.append(ids)
            all_strings.extend(s)
        return all_ids, all_strings

    def sample_variable_length(self, num_samples: int, length_range: Tuple[int, int] = (20, 100),
                               batch_size: int = 256, samples_per_length: int = 16,
                               show_progress: bool = True):
        """Small batches, each with its own sequence length drawn from numpy's global generator like
        the reference does (Group sampler.py:466-513)."""
        lo, hi = length_range
        inner = min(batch_size, samples_per_length)
        all_ids, all_strings = [], []
        if not getattr(self, "merge_variable_length", True):
            while len(all_strings) < num_samples:
                cur = min(inner, num_samples - len(all_strings))
                seq_length = int(np.random.randint(lo, hi + 1))
                ids, s = self.sample(cur, seq_length, show_progress=False, seq_offset=len(all_strings))
                all_ids.append(ids)
                all_strings.extend(s)
            return all_ids, all_strings
        # The groups' lengths are drawn first (the same np.random calls in the same order); up to `batch_size` sequences
        # of different lengths then share ONE reverse loop as a batch padded to its longest row (padding: PAD tokens
        # outside the attention mask), which the forward deals to its waves by length instead of a launch of a handful
        # of workgroups per group. With a fixed seed the rows are those of the group-by-group loop.
        pieces, n = [], 0
        while n < num_samples:
            cur = min(inner, num_samples - n)
            pieces.append((cur, int(np.random.randint(lo, hi + 1))))
            n += cur
        return run_length_pieces(pieces, batch_size, self._sample_lengths)

    def _sample_lengths(self, lengths: List[int], seq_offset: int = 0):
        """``sample`` for rows of different lengths in one batch: row i is BOS, lengths[i] - 2 masks, EOS, then PAD
        outside the attention mask."""
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
        ids = self._core.run(ids.to(self.device), attn.to(self.device), None, seq_offset=seq_offset)
        return ids, decode_batch(self, ids)

    def _chemistry(self):
        """(count_stars, check_validity): caller-provided callables, else the host package's
        ``src.utils.chemistry`` (RDKit; what the reference imports), else a plain '*' count and no
        validity check available."""
        cs, cv = self.count_stars_fn, self.check_validity_fn
        if cs is None or cv is None:
            try:
                from src.utils.chemistry import count_stars, check_validity  # host side, RDKit
                cs = cs or count_stars
                cv = cv or check_validity
            except Exception:
                cs = cs or (lambda smiles: smiles.count("*"))
        return cs, cv

    def sample_with_rejection(self, target_samples: int, seq_length: int, batch_size: int = 256,
                              max_attempts: int = 10, target_stars: int = 2, require_validity: bool = True,
                              show_progress: bool = True):
        """Oversample and keep the strings with exactly ``target_stars`` stars that (optionally) pass the
        host validity check; the acceptance rate steers the next round's size
        (Group sampler.py:515-591). Returns ``([], strings)`` like the reference."""
        count_stars, check_validity = self._chemistry()
        if require_validity and check_validity is None:
            raise RuntimeError("sample_with_rejection(require_validity=True) needs a validity check: set "
                               "sampler.check_validity_fn or make src.utils.chemistry importable (RDKit)")
        valid: List[str] = []
        acceptance = 0.5
        attempt = 0
        while len(valid) < target_samples and attempt < max_attempts:
            remaining = target_samples - len(valid)
            factor = max(1.5, 1.0 / max(acceptance, 0.1))
            n_gen = min(int(remaining * factor), batch_size * 10)
            _, strings = self.sample_batch(n_gen, seq_length, batch_size, show_progress=False,
                                           seq_offset=attempt * batch_size * 10)
            new_valid = 0
            for smi in strings:
                if len(valid) >= target_samples:
                    break
                if count_stars(smi) == target_stars and (not require_validity or check_validity(smi)):
                    valid.append(smi)
                    new_valid += 1
            if strings:
                acceptance = 0.7 * acceptance + 0.3 * (new_valid / len(strings))
            attempt += 1
        if len(valid) < target_samples:
            print(f"Warning: Only found {len(valid)}/{target_samples} valid samples "
                  f"after {max_attempts} attempts (acceptance rate: {acceptance:.2%})")
        return [], valid[:target_samples]

    def sample_conditional(self, batch_size: int, seq_length: int,
                           prefix_ids: Optional[torch.Tensor] = None,
                           suffix_ids: Optional[torch.Tensor] = None, show_progress: bool = True,
                           seq_offset: int = 0) -> Tuple[torch.Tensor, List[str]]:
        """Prefix after BOS and suffix before EOS stay fixed; positions are chosen among the non-fixed
        masks and the in-step update looks at the direct placeholder id alone
        (Group sampler.py:593-696, update :679-687)."""
        self.diffusion_model.eval()
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
        self._core.num_steps = int(self.num_steps)
        self._core.temperature = float(self.temperature)
        self._core.use_constraints = bool(self.use_constraints)
        extra = core.VB_FIXED | core.VB_DYN_ALL_MASKS | core.VB_DYN_SINGLE
        ids = self._core.run(ids, torch.ones_like(ids), fixed, seq_offset=seq_offset,
                             progress=_progress(show_progress, "Sampling"), variant_extra=extra)
        strings = decode_batch(self, ids)
        return ids, strings

    # aliases the reference keeps for older callers (Group sampler.py:697-718) have no device
    # counterpart here: the constraint and the repair run inside bd_sample_step / bd_sample_fixup.
