"""The C-ABI library loads on a CPU-only host and exports every symbol include/bidiff.h
declares; the host-side RNG helpers (pure CPU code in the library) agree bit for bit with the
oracle and with the published Philox4x32-10 known-answer vectors."""
import ctypes as C
import os
import re

import numpy as np

from tests import helpers as H
from bidiff import _lib
from oracle import sampler_ref as ora
from tests.golden import replay_inputs as RI

HEADER = os.path.join(H.ROOT, "include", "bidiff.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 19
    handle = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in bidiff.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "python binding table out of sync with the header"
    assert b"gfx950" in _lib.lib().bd_version()


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for c, k, want in kat:
        out = (C.c_uint32 * 4)()
        _lib.lib().bd_host_philox(c[0], c[1], c[2], c[3], k[0], k[1], out)
        assert [int(x) for x in out] == want
        assert ora.philox(c, k) == want


def test_gumbel_transform_bitwise_across_library_oracle_numpy():
    u = np.random.RandomState(3).random_sample(200000).astype(np.float32)
    u[:4] = [0.0, 1.0, 1e-30, 0.99999999]
    out = np.empty_like(u)
    _lib.lib().bd_host_gumbel(u.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), u.size)
    o = ora.gumbel(u)
    n = RI.gumbel_np(u)
    assert np.array_equal(out.view(np.uint32), o.view(np.uint32))
    assert np.array_equal(out.view(np.uint32), n.view(np.uint32))
    ref = -np.log(-np.log(np.clip(u.astype(np.float64), 2.98023223876953125e-08, 0.99999994)))
    assert np.abs(out[4:] - ref[4:]).max() < 2e-6


def test_dropout_decision_stream_library_equals_oracle():
    """The library's host evaluation of the training-dropout stream (the code the kernels run) and
    the oracle's agree bit for bit, ragged column counts included; the keep rate is 1 - p."""
    for (p, step, site, seed, rows, cols) in [(0.1, 0, 0, 1, 50, 128), (0.1, 7, 13, 2 ** 63 + 5, 33, 30),
                                               (0.5, 2 ** 32 - 1, 4 * 31 + 3, 2 ** 64 - 1, 9, 5), (0.0, 1, 1, 1, 4, 8)]:
        d = _lib.Dropout(p, step, seed)
        out = np.empty((rows, cols), dtype=np.float32)
        assert _lib.lib().bd_host_dropout_factors(C.byref(d), site, rows, cols, out.ctypes.data_as(C.c_void_p)) == 0
        want = ora.dropout_mask(seed, step, site, p, rows, cols)
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
        assert set(np.unique(out)) <= {np.float32(0.0), np.float32(1.0) / (np.float32(1.0) - np.float32(p))}
    big = ora.dropout_mask(42, 3, 2, 0.1, 2000, 512)
    assert abs(float((big == 0).mean()) - 0.1) < 2e-3
    assert abs(float(big.mean()) - 1.0) < 2e-3
    # rows and 4-column groups are independent draws: no repeated row pattern
    assert len({r.tobytes() for r in (big[:200] == 0)}) == 200
    assert _lib.lib().bd_host_dropout_factors(C.byref(_lib.Dropout(1.0, 0, 0)), 0, 1, 4,
                                              np.empty(4, np.float32).ctypes.data_as(C.c_void_p)) == -1


def test_uniform_streams_match_oracle():
    lib, olib = _lib.lib(), ora.lib()
    for seed in (0, 42, 2**61 + 12345):
        for seq, t, pos, v in [(0, 1, 0, 0), (7, 100, 127, 38), (4095, 50, 511, 802), (123456, 3, 17, 5)]:
            assert lib.bd_host_uniform_select(seed, seq, t, pos) == olib.ora_uniform_select(seed, seq, t, pos)
            a = lib.bd_host_uniform_token(seed, seq, t, pos, v)
            assert a == olib.ora_uniform_token(seed, seq, t, pos, v)
            assert 0.0 < a < 1.0


def test_unmask_count_table_is_python_float_arithmetic():
    from bidiff.sampling.core import unmask_count_table
    tab = unmask_count_table(100, 512)
    odd = [(49, 49), (98, 49), (147, 49), (196, 49), (294, 49), (343, 49), (392, 49), (98, 98), (196, 98),
           (294, 98), (392, 98)]
    for n, t in odd:
        assert int(n * (1.0 / t)) != n // t
        assert tab[t, n] == max(1, int(n * (1.0 / t)))
    assert tab[1, 126] == 126 and tab[100, 126] == 1 and tab[26, 126 - 74] == 2
