"""ConstrainedSampler for Group-SELFIES token sequences.

Drop-in for ``Bi_Diffusion_Group_SELFIES/src/sampling/sampler.py``: the placeholder
constraint sums per-token contributions (the direct placeholder token counts 1, a group
token counts the ``[I+3]`` occurrences of its group SMILES), UNK is forbidden together with
the other special tokens, and chosen positions are drawn in random order.
"""
import re
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import core
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
        strings = self.tokenizer.batch_decode(ids.cpu().tolist(), skip_special_tokens=True)
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
