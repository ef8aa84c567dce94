"""TEST INFRASTRUCTURE - ctypes front end of the C oracle (oracle/sampler_ref.c).

numpy arrays in, numpy arrays out; the C code mutates ``ids`` and ``logits`` in place like
the reference mutates its tensors.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None
VP = C.c_void_p


class StepArgs(C.Structure):
    _fields_ = [
        ("ids", VP), ("fixed", VP), ("attn", VP), ("keymask", VP), ("has_eos", VP),
        ("logits", VP), ("hidden", VP), ("wout", VP), ("u_sel", VP), ("u_tok", VP),
        ("ktable", VP), ("tok_flags", VP), ("contrib", VP), ("ring_slot", VP),
        ("counted_ids", VP), ("bond_ids", VP), ("non_special", VP),
        ("n_counted", C.c_int), ("n_bond", C.c_int), ("n_ns", C.c_int), ("R", C.c_int),
        ("B", C.c_int), ("L", C.c_int), ("V", C.c_int), ("d", C.c_int),
        ("mask_id", C.c_int), ("pad_id", C.c_int), ("bos_id", C.c_int), ("eos_id", C.c_int),
        ("open_id", C.c_int), ("close_id", C.c_int), ("boost_id", C.c_int), ("count_id", C.c_int),
        ("variant", C.c_int), ("use_constraints", C.c_int), ("t", C.c_int),
        ("temperature", C.c_float),
        ("seed", C.c_ulonglong),
        ("seq_offset", C.c_longlong),
    ]


class FixArgs(C.Structure):
    _fields_ = [
        ("ids", VP), ("logits", VP), ("attn", VP), ("keymask", VP), ("has_eos", VP),
        ("tok_flags", VP), ("ring_ids", VP),
        ("excl", C.c_int * 8),
        ("n_excl", C.c_int), ("R", C.c_int),
        ("B", C.c_int), ("L", C.c_int), ("V", C.c_int),
        ("mask_id", C.c_int), ("pad_id", C.c_int), ("bos_id", C.c_int), ("eos_id", C.c_int),
        ("open_id", C.c_int), ("close_id", C.c_int),
        ("count_id", C.c_int), ("target", C.c_int),
        ("do_fix", C.c_int), ("smiles_syntax", C.c_int), ("natural_eos", C.c_int),
    ]


def build():
    """Compile the C oracle (gcc) if the shared object is missing or stale."""
    src = os.path.join(_HERE, "sampler_ref.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        h = C.CDLL(_SO)
        h.ora_sample_step.restype = C.c_int
        h.ora_sample_step.argtypes = [C.POINTER(StepArgs)]
        h.ora_sample_fixup.restype = C.c_int
        h.ora_sample_fixup.argtypes = [C.POINTER(FixArgs)]
        h.ora_gumbel.restype = None
        h.ora_gumbel.argtypes = [VP, VP, C.c_longlong]
        h.ora_philox.restype = None
        h.ora_philox.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
        h.ora_uniform_select.restype = C.c_float
        h.ora_uniform_select.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        h.ora_uniform_token.restype = C.c_float
        h.ora_uniform_token.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        h.ora_dropout_mask.restype = None
        h.ora_dropout_mask.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, C.c_longlong, C.c_longlong, VP]
        h.ora_corrupt.restype = C.c_int
        h.ora_corrupt.argtypes = [VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_uint64, C.c_longlong]
        _lib = h
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(VP)


def _arr(a, dt):
    a = np.ascontiguousarray(np.asarray(a, dtype=dt))
    if a.size == 0:
        a = np.zeros(1, dtype=dt)
    return a


def gumbel(u):
    u = np.ascontiguousarray(u, dtype=np.float32)
    out = np.empty_like(u)
    lib().ora_gumbel(_p(u), _p(out), u.size)
    return out


def philox(c, k):
    out = (C.c_uint32 * 4)()
    lib().ora_philox(c[0], c[1], c[2], c[3], k[0], k[1], out)
    return [int(x) for x in out]


def dropout_mask(seed, step, site, p, rows, cols):
    """fp32 [rows, cols] of dropout factors (0 or 1 / (1 - p)) of one site of the decision stream."""
    out = np.empty((rows, cols), dtype=np.float32)
    lib().ora_dropout_mask(int(seed), int(step), int(site), float(p), rows, cols, _p(out))
    return out


def corrupt(ids, attn, t, schedule, uniforms, mask_id, pad_id, bos_id, eos_id, seed=0, seq_offset=0):
    """forward_process restatement -> (noisy_ids int64, mask_indicator bool)."""
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    B, L = ids.shape
    attn_a = None if attn is None else np.ascontiguousarray(attn, dtype=np.int64)
    t = np.ascontiguousarray(t, dtype=np.int64)
    schedule = np.ascontiguousarray(schedule, dtype=np.float32)
    un = None if uniforms is None else np.ascontiguousarray(uniforms, dtype=np.float32)
    noisy = np.empty_like(ids)
    ind = np.empty((B, L), dtype=np.uint8)
    rc = lib().ora_corrupt(_p(ids), _p(attn_a), _p(t), _p(schedule), _p(un), _p(noisy), _p(ind), B, L,
                           mask_id, pad_id, -1 if bos_id is None else bos_id,
                           -1 if eos_id is None else eos_id, seed, seq_offset)
    assert rc == 0
    return noisy, ind.astype(bool)


class Tables:
    """Host token tables in the layout both the oracle and the HIP kernel take
    (built from the product's ``core.TokenTables`` or by hand in tests)."""

    def __init__(self, V, mask_id, pad_id, bos_id, eos_id, tok_flags, contrib, ring_ids, bond_ids,
                 non_special, open_id=-1, close_id=-1, boost_id=-1, count_id=-1, variant=0,
                 fix_excl=(), fix_syntax=False):
        self.V, self.mask_id, self.pad_id, self.bos_id, self.eos_id = V, mask_id, pad_id, bos_id, eos_id
        self.open_id, self.close_id, self.boost_id, self.count_id = open_id, close_id, boost_id, count_id
        self.variant, self.fix_excl, self.fix_syntax = variant, list(fix_excl), fix_syntax
        self.tok_flags = _arr(tok_flags, np.uint8)
        self.contrib = _arr(contrib, np.int32)
        self.ring_ids_list = [int(r) for r in ring_ids]
        self.bond_ids_list = [int(b) for b in bond_ids]
        self.ring_ids = _arr(ring_ids, np.int32)
        self.bond_ids = _arr(bond_ids, np.int32)
        self.non_special_list = [int(v) for v in non_special]
        self.non_special = _arr(non_special, np.int32)
        self.counted_list = [int(v) for v in np.nonzero(np.asarray(contrib) > 0)[0]]
        self.counted_ids = _arr(self.counted_list, np.int32)
        slot = np.full(V, -1, dtype=np.int32)
        for s, r in enumerate(self.ring_ids_list):
            slot[r] = s
        self.ring_slot = slot

    @classmethod
    def from_token_tables(cls, tt):
        return cls(tt.V, tt.mask_id, tt.pad_id, tt.bos_id, tt.eos_id, tt.tok_flags, tt.contrib, tt.ring_ids,
                   tt.bond_ids, tt.non_special, tt.open_id, tt.close_id, tt.boost_id, tt.count_id, tt.variant,
                   tt.fix_excl, tt.fix_syntax)


def sample_step(tb, ids, logits, ktable_row, t, temperature=1.0, use_constraints=True, fixed=None,
                u_sel=None, u_tok=None, seed=0, seq_offset=0, natural_eos=False, attn=None, has_eos=None,
                variant_extra=0):
    """One reverse step on copies; returns dict(ids, logits[, attn, has_eos]). ``variant_extra``: the
    sample_conditional bits (1 fixed, 32 / 64 / 128) OR-ed into the sampler's variant."""
    ids = np.array(ids, dtype=np.int64, order="C", copy=True)
    B, L = ids.shape
    logits = np.array(logits, dtype=np.float32, order="C", copy=True)
    V = logits.shape[-1]
    a = StepArgs()
    keep = [ids, logits]

    def hold(x, dt):
        if x is None:
            return None
        y = np.array(x, dtype=dt, order="C", copy=True)
        keep.append(y)
        return y
    fixed_a = hold(fixed, np.uint8)
    attn_a = hold(attn, np.int64)
    eos_a = hold(has_eos, np.uint8)
    us = hold(u_sel, np.float32)
    ut = hold(u_tok, np.float32)
    kt = hold(ktable_row, np.int32)
    a.ids, a.fixed, a.attn, a.keymask, a.has_eos = _p(ids), _p(fixed_a), _p(attn_a), None, _p(eos_a)
    a.logits, a.hidden, a.wout, a.u_sel, a.u_tok = _p(logits), None, None, _p(us), _p(ut)
    a.ktable = _p(kt)
    a.tok_flags, a.contrib, a.ring_slot = _p(tb.tok_flags), _p(tb.contrib), _p(tb.ring_slot)
    a.counted_ids, a.bond_ids, a.non_special = _p(tb.counted_ids), _p(tb.bond_ids), _p(tb.non_special)
    a.n_counted, a.n_bond, a.n_ns, a.R = len(tb.counted_list), len(tb.bond_ids_list), len(tb.non_special_list), len(tb.ring_ids_list)
    a.B, a.L, a.V, a.d = B, L, V, 0
    a.mask_id, a.pad_id, a.bos_id, a.eos_id = tb.mask_id, tb.pad_id, tb.bos_id, tb.eos_id
    a.open_id, a.close_id, a.boost_id, a.count_id = tb.open_id, tb.close_id, tb.boost_id, tb.count_id
    a.variant = tb.variant | (16 if natural_eos else 0) | int(variant_extra)
    a.use_constraints, a.t = int(use_constraints), int(t)
    a.temperature = float(temperature)
    a.seed, a.seq_offset = int(seed), int(seq_offset)
    rc = lib().ora_sample_step(C.byref(a))
    assert rc == 0, "oracle step rejected its arguments"
    return {"ids": ids, "logits": logits, "attn": attn_a, "has_eos": eos_a}


def sample_fixup(tb, ids, logits, do_fix=True, natural_eos=False, attn=None, has_eos=None):
    ids = np.array(ids, dtype=np.int64, order="C", copy=True)
    B, L = ids.shape
    lg = None if logits is None else np.ascontiguousarray(logits, dtype=np.float32)
    attn_a = None if attn is None else np.array(attn, dtype=np.int64, order="C", copy=True)
    eos_a = None if has_eos is None else np.array(has_eos, dtype=np.uint8, order="C", copy=True)
    f = FixArgs()
    f.ids, f.logits, f.attn, f.keymask, f.has_eos = _p(ids), _p(lg), _p(attn_a), None, _p(eos_a)
    f.tok_flags, f.ring_ids = _p(tb.tok_flags), _p(tb.ring_ids)
    for i, v in enumerate(tb.fix_excl[:8]):
        f.excl[i] = int(v)
    f.n_excl, f.R = len(tb.fix_excl[:8]), len(tb.ring_ids_list)
    f.B, f.L, f.V = B, L, tb.V
    f.mask_id, f.pad_id, f.bos_id, f.eos_id = tb.mask_id, tb.pad_id, tb.bos_id, tb.eos_id
    f.open_id, f.close_id = tb.open_id, tb.close_id
    f.count_id, f.target = tb.count_id, 2
    f.do_fix = int(do_fix and lg is not None and tb.count_id >= 0)
    f.smiles_syntax = int(bool(f.do_fix) and tb.fix_syntax)
    f.natural_eos = int(natural_eos)
    rc = lib().ora_sample_fixup(C.byref(f))
    assert rc == 0
    return {"ids": ids, "attn": attn_a, "has_eos": eos_a}
