"""Host side of the reverse-sampling loop shared by the three sampler flavours.

The device work of one reverse step is two calls: ``bd_backbone_forward`` and
``bd_sample_step``. This module builds the immutable per-sampler token tables, the
``max(1, int(n * (1.0 / t)))`` unmask-count table (Python float64 arithmetic, as the
reference evaluates it) and drives the T-step loop plus the post-loop repairs.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import _lib

TF_BOND, TF_RING, TF_OPEN, TF_CLOSE, TF_SPECIAL, TF_COUNTED, TF_EOS = 1, 2, 4, 8, 16, 32, 64
VB_FIXED, VB_SYNTAX, VB_URGENCY, VB_SORTED, VB_NATURAL_EOS = 1, 2, 4, 8, 16
VB_DYN_ALL_MASKS, VB_DYN_SINGLE, VB_SPECIAL_UNFIXED = 32, 64, 128


def unmask_count_table(num_steps: int, max_len: int) -> np.ndarray:
    """k[t][n] = max(1, int(n * (1.0 / t))) for t = 0..T (row 0 unused), n = 0..max_len.

    Evaluated with Python floats exactly like the samplers do (sampler.py:677,684): this is
    not n // t (e.g. n = t = 49 gives 0 -> 1, n = 98, t = 49 gives 1)."""
    tab = np.ones((num_steps + 1, max_len + 1), dtype=np.int32)
    for t in range(1, num_steps + 1):
        p = 1.0 / t
        for n in range(max_len + 1):
            tab[t, n] = max(1, int(n * p))
    return tab


@dataclass
class TokenTables:
    """Integer views of the vocabulary the step kernel needs (host copies)."""
    V: int
    mask_id: int
    pad_id: int
    bos_id: int
    eos_id: int
    open_id: int = -1
    close_id: int = -1
    boost_id: int = -1
    count_id: int = -1
    variant: int = 0
    tok_flags: np.ndarray = None
    contrib: np.ndarray = None
    ring_slot: np.ndarray = None
    ring_ids: List[int] = field(default_factory=list)
    bond_ids: List[int] = field(default_factory=list)
    counted_ids: List[int] = field(default_factory=list)
    non_special: List[int] = field(default_factory=list)
    fix_excl: List[int] = field(default_factory=list)
    fix_syntax: bool = False

    def finish(self):
        V = self.V
        self.tok_flags = np.zeros(V, dtype=np.uint8) if self.tok_flags is None else self.tok_flags
        self.contrib = np.zeros(V, dtype=np.int32) if self.contrib is None else self.contrib
        self.ring_slot = np.full(V, -1, dtype=np.int32)
        for slot, rid in enumerate(self.ring_ids):
            self.ring_slot[rid] = slot
            self.tok_flags[rid] |= TF_RING
        for bid in self.bond_ids:
            self.tok_flags[bid] |= TF_BOND
        if 0 <= self.open_id < V:
            self.tok_flags[self.open_id] |= TF_OPEN
        if 0 <= self.close_id < V:
            self.tok_flags[self.close_id] |= TF_CLOSE
        self.counted_ids = [int(v) for v in np.nonzero(self.contrib > 0)[0]]
        for v in self.counted_ids:
            self.tok_flags[v] |= TF_COUNTED
        return self


class DeviceTables:
    """TokenTables uploaded once per (sampler, device)."""

    def __init__(self, tt: TokenTables, device):
        def up(a, dt):
            a = np.asarray(a, dtype=dt)
            if a.size == 0:
                a = np.zeros(1, dtype=dt)
            return torch.from_numpy(a.copy()).to(device)
        self.tok_flags = up(tt.tok_flags, np.uint8)
        self.contrib = up(tt.contrib, np.int32)
        self.ring_slot = up(tt.ring_slot, np.int32)
        self.ring_ids = up(tt.ring_ids, np.int32)
        self.bond_ids = up(tt.bond_ids, np.int32)
        self.counted_ids = up(tt.counted_ids, np.int32)
        self.non_special = up(tt.non_special, np.int32)


def fill_step_args(a: _lib.StepArgs, tt: TokenTables, dt: DeviceTables):
    a.tok_flags = dt.tok_flags.data_ptr()
    a.contrib = dt.contrib.data_ptr()
    a.ring_slot = dt.ring_slot.data_ptr()
    a.counted_ids = dt.counted_ids.data_ptr()
    a.bond_ids = dt.bond_ids.data_ptr()
    a.non_special = dt.non_special.data_ptr()
    a.n_counted, a.n_bond, a.n_ns, a.R = len(tt.counted_ids), len(tt.bond_ids), len(tt.non_special), len(tt.ring_ids)
    a.V = tt.V
    a.mask_id, a.pad_id, a.bos_id, a.eos_id = tt.mask_id, tt.pad_id, tt.bos_id, tt.eos_id
    a.open_id, a.close_id, a.boost_id, a.count_id = tt.open_id, tt.close_id, tt.boost_id, tt.count_id


class ReverseSampler:
    """T-step reverse loop on the device for one batch of partially masked sequences."""

    def __init__(self, diffusion_model, tables: TokenTables, num_steps: int, temperature: float,
                 use_constraints: bool, device):
        self.model = diffusion_model
        self.tt = tables
        self.num_steps = int(num_steps)
        self.temperature = float(temperature)
        self.use_constraints = bool(use_constraints)
        self.device = torch.device(device)
        self._dt: Optional[DeviceTables] = None
        self._ktab: Dict[int, torch.Tensor] = {}
        self.seed: Optional[int] = None      # fixed Philox seed (tests); None = draw from torch's generator
        self.record = None                   # list collecting per-step snapshots when not None
        # Chosen-row workspace of the step kernel (raw logits of the rows a step draws, computed for the whole batch
        # as one product ahead of the draws): None = when the vocabulary is large enough to pay (V > 128).
        self.row_workspace: Optional[bool] = None
        self.row_cap_max = 16
        # Sequences shorter than the batch's padded length share workgroups of the single-launch forward (False: every
        # sequence computes all L rows, padding included).
        self.ragged = True
        # The positions a step draws are chosen BEFORE its forward pass (the choice depends on the ids alone), and the
        # forward finishes only those rows: the last feed-forward block and the final LayerNorm run for the chosen
        # rows instead of all B * L (False: every step computes every hidden row). Same ids either way.
        self.choose_first = True

    def _tables(self):
        if self._dt is None:
            self._dt = DeviceTables(self.tt, self.device)
        return self._dt

    def _ktable(self, L):
        if L not in self._ktab:
            self._ktab[L] = torch.from_numpy(unmask_count_table(self.num_steps, L)).to(self.device)
        return self._ktab[L]

    def run(self, ids, attention_mask, fixed_mask=None, natural_eos=False, seq_offset=0,
            progress=None, has_eos=None, variant_extra=0):
        """Mutates ``ids`` (and, in natural-EOS mode, ``attention_mask``) in place; returns the
        final ids after the post-loop repairs. ``variant_extra``: VB_* bits OR-ed into the sampler's
        variant for this call (the sample_conditional flavours of the SELFIES samplers)."""
        tt, dt = self.tt, self._tables()
        backbone = self.model.backbone
        self.model.eval()
        _lib.require_device(ids, "sampling")
        B, L = ids.shape
        if B == 0:
            return ids
        if L > 512 or tt.V > 1024:
            raise ValueError("bidiff sampling supports seq_len <= 512 and vocab <= 1024")
        dev = ids.device
        ids = ids.contiguous()
        attn = attention_mask.to(torch.long).contiguous()
        keymask = (attn != 0).to(torch.uint8).contiguous()
        fixed = None if fixed_mask is None else fixed_mask.to(torch.uint8).contiguous()
        variant = tt.variant | (VB_NATURAL_EOS if natural_eos else 0) | int(variant_extra)
        if natural_eos and has_eos is None:
            has_eos = torch.zeros(B, dtype=torch.uint8, device=dev)
        seed = self.seed
        if seed is None:
            seed = int(torch.randint(0, 2**62, (1,)).item())
        ktab = self._ktable(L)
        d = backbone.hidden_size
        # Batches padded to their longest row: deal the sequences to the forward's waves by their own lengths (one
        # device -> host copy per call). Lengths only shrink during a run (natural EOS), so the table stays valid;
        # rows beyond a sequence's waves are never drawn, read or repaired, and stay zero.
        wave_table = None
        if self.ragged and hasattr(backbone, "wave_table_for_mask"):
            wave_table = backbone.wave_table_for_mask(attn)
        hidden = (torch.zeros if wave_table is not None else torch.empty)(B, L, d, dtype=torch.bfloat16, device=dev)
        logits = None
        pk = backbone.packed()
        lib = _lib.lib()
        stream = _lib.stream_handle()
        final_logits = None

        a = _lib.StepArgs()
        fill_step_args(a, tt, dt)
        a.ids = ids.data_ptr()
        a.fixed = None if fixed is None else fixed.data_ptr()
        a.attn = attn.data_ptr() if natural_eos else None
        a.keymask = keymask.data_ptr() if natural_eos else None
        a.has_eos = has_eos.data_ptr() if natural_eos else None
        a.u_sel = None
        a.u_tok = None
        a.B, a.L, a.d = B, L, d
        a.variant, a.use_constraints = variant, int(self.use_constraints)
        a.temperature = self.temperature
        a.seed = seed
        a.seq_offset = int(seq_offset)

        use_rows = self.row_workspace if self.row_workspace is not None else tt.V > 128
        use_rows = bool(use_rows) and d in (128, 256, 384, 512)
        rows_mode = None
        if bool(self.choose_first) and getattr(backbone, "compute_dtype", "bf16") == "bf16" \
                and hasattr(backbone, "rows_forward_mode"):
            rows_mode = backbone.rows_forward_mode(B, L)
        choose_first = rows_mode is not None
        if use_rows or choose_first:
            # A sequence that starts with n masks draws about n / T of them per step (k = max(1, int(n_t / t)) keeps
            # n_t near n t / T), at most ceil(L / T): the row list covers that many rows per sequence, rounded up to
            # a power of two; the rare draws beyond it compute their row one by one (and, with the rows chosen ahead
            # of the forward, make it finish every row of their sequence).
            row_cap = 1
            while row_cap < min(self.row_cap_max, -(-L // self.num_steps)):
                row_cap *= 2
            row_pos = torch.empty(B * row_cap, dtype=torch.int32, device=dev)
        if use_rows:
            vs = (tt.V + 31) // 32 * 32
            row_logits = torch.empty(B * row_cap, vs, dtype=torch.float32, device=dev)
        if choose_first:
            need_all = torch.empty(B, dtype=torch.uint8, device=dev)

        steps = range(self.num_steps, 0, -1)
        if progress is not None:
            steps = progress(steps)
        dump = None
        if self.record is not None:
            dump = torch.zeros(B, L, tt.V, dtype=torch.float32, device=dev)
        for t in steps:
            ts = torch.full((B,), t, device=dev, dtype=torch.long)
            # fp32 compute mode: full fp32 logits every step (the chosen-row shortcut reads bf16 hidden rows)
            full = (t == 1 and self.use_constraints) or getattr(backbone, "compute_dtype", "bf16") == "fp32"
            a.t = t
            a.ktable = ktab[t].data_ptr()
            if self.record is not None:
                ids_before = ids.clone()
                attn_before = attn.clone()
                eos_before = None if has_eos is None else has_eos.clone()
            if full:
                if logits is None:
                    logits = torch.empty(B, L, tt.V, dtype=torch.float32, device=dev)
                backbone.run_forward(ids, ts, keymask, want_hidden=False, want_logits=True, logits_out=logits,
                                     **({} if wave_table is None else {"wave_table": wave_table}))
                raw = logits.clone() if self.record is not None else None
                a.hidden, a.wout, a.logits = None, None, logits.data_ptr()
                a.row_logits, a.row_pos, a.row_cap = None, None, 0
                final_logits = logits
            else:
                if choose_first:
                    _lib.check(lib.bd_sample_choose_rows(C.byref(a), row_pos.data_ptr(), need_all.data_ptr(), row_cap,
                                                         stream), "bd_sample_choose_rows")
                    if rows_mode == "per_layer" and bool(need_all.any().item()):
                        # a draw beyond the row list (rare): this step computes every hidden row. The read-back waits
                        # for the device, which costs nothing beside these models' forward of tens of milliseconds
                        backbone.run_forward(ids, ts, keymask, want_hidden=True, want_logits=False, hidden_out=hidden)
                    else:
                        backbone.run_forward_rows(ids, ts, keymask, row_pos, need_all, row_cap, hidden, wave_table=wave_table)
                else:
                    backbone.run_forward(ids, ts, keymask, want_hidden=True, want_logits=False, hidden_out=hidden,
                                         **({} if wave_table is None else {"wave_table": wave_table}))
                a.hidden, a.wout = hidden.data_ptr(), pk.w_out.data_ptr()
                if use_rows:
                    a.row_logits, a.row_pos, a.row_cap = row_logits.data_ptr(), row_pos.data_ptr(), row_cap
                    a.rows_chosen = int(choose_first)
                if dump is not None:
                    dump.zero_()
                    a.logits = dump.data_ptr()
                else:
                    a.logits = None
            _lib.check(lib.bd_sample_step(C.byref(a), stream), "bd_sample_step")
            if self.record is not None:
                self.record.append({
                    "t": t, "mode": "logits" if full else "hidden",
                    "ids_before": ids_before.cpu(), "attn_before": attn_before.cpu(),
                    "has_eos_before": None if eos_before is None else eos_before.cpu(),
                    "raw_logits": (raw if full else dump).cpu().clone(),
                    "ids_after": ids.cpu().clone(), "seed": seed, "seq_offset": int(seq_offset),
                    "logits_after": logits.cpu().clone() if full else None,
                    "variant_extra": int(variant_extra), "fixed": None if fixed is None else fixed.cpu().clone(),
                })

        # post-loop repairs
        do_fix = self.use_constraints and final_logits is not None and tt.count_id >= 0
        if do_fix or natural_eos:
            f = _lib.FixArgs()
            f.ids = ids.data_ptr()
            f.logits = final_logits.data_ptr() if do_fix else None
            f.attn = attn.data_ptr() if natural_eos else None
            f.keymask = keymask.data_ptr() if natural_eos else None
            f.has_eos = has_eos.data_ptr() if natural_eos else None
            f.tok_flags = dt.tok_flags.data_ptr()
            f.ring_ids = dt.ring_ids.data_ptr()
            excl = list(tt.fix_excl)[:8]
            for i, v in enumerate(excl):
                f.excl[i] = int(v)
            f.n_excl, f.R = len(excl), len(tt.ring_ids)
            f.B, f.L, f.V = B, L, tt.V
            f.mask_id, f.pad_id, f.bos_id, f.eos_id = tt.mask_id, tt.pad_id, tt.bos_id, tt.eos_id
            f.open_id, f.close_id = tt.open_id, tt.close_id
            f.count_id, f.target = tt.count_id, 2
            f.do_fix = int(do_fix)
            f.smiles_syntax = int(do_fix and tt.fix_syntax)
            f.natural_eos = int(natural_eos)
            _lib.check(lib.bd_sample_fixup(C.byref(f), stream), "bd_sample_fixup")
        if natural_eos:
            attention_mask.copy_(attn)
        return ids


def plain_sample_step(model, noisy_ids, t, attention_mask, temperature):
    """DiscreteMaskingDiffusion.sample_step: resample every MASK position from
    softmax(logits / temperature) with no constraint (reference diffusion.py:211-250)."""
    backbone = model.backbone
    dev = noisy_ids.device
    _lib.require_device(noisy_ids, "sample_step")
    B, L = noisy_ids.shape
    V = backbone.vocab_size
    tt = TokenTables(V=V, mask_id=int(model.mask_token_id), pad_id=int(model.pad_token_id), bos_id=-1,
                     eos_id=-1, non_special=list(range(V))).finish()
    dtab = DeviceTables(tt, dev)
    ids = noisy_ids.to(torch.long).clone().contiguous()
    ts = torch.full((B,), int(t), device=dev, dtype=torch.long)
    keymask = None if attention_mask is None else (attention_mask != 0).to(torch.uint8).contiguous()
    with torch.no_grad():
        _, logits = backbone.run_forward(ids, ts, keymask, want_hidden=False, want_logits=True)
    ktab = torch.arange(L + 1, dtype=torch.int32, device=dev).clamp_(min=1)
    a = _lib.StepArgs()
    fill_step_args(a, tt, dtab)
    a.ids = ids.data_ptr()
    a.logits = logits.data_ptr()
    a.ktable = ktab.data_ptr()
    a.B, a.L, a.d = B, L, backbone.hidden_size
    a.variant, a.use_constraints, a.t = 0, 0, max(1, int(t))
    a.temperature = float(temperature)
    a.seed = int(torch.randint(0, 2**62, (1,)).item())
    a.seq_offset = 0
    _lib.check(_lib.lib().bd_sample_step(C.byref(a), _lib.stream_handle()), "bd_sample_step")
    return ids


def run_length_pieces(pieces, batch_size, run_lengths, offset0=0):
    """Reverse loops for consecutive pieces (rows, length) of equal-length sequences, merged: up to ``batch_size``
    rows of different lengths share one loop (``run_lengths(lengths, seq_offset) -> (ids, strings)``, a batch padded
    to its longest row that the forward deals to its waves by length). Rows keep the global numbers they have when
    every piece runs as a loop of its own, so with a fixed seed the results are the same. Returns the pieces' id
    tensors [rows, length] and the strings of all rows in order."""
    all_ids, all_strings, g, done = [], [], 0, int(offset0)
    while g < len(pieces):
        take, rows = [], 0
        while g < len(pieces) and (not take or rows + pieces[g][0] <= batch_size):
            take.append(pieces[g])
            rows += pieces[g][0]
            g += 1
        ids, strings = run_lengths([ln for cur, ln in take for _ in range(cur)], done)
        off = 0
        for cur, ln in take:
            all_ids.append(ids[off:off + cur, :ln].contiguous())
            off += cur
        all_strings.extend(strings)
        done += rows
    return all_ids, all_strings
