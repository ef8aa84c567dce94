"""The forward that finishes only the rows a reverse step reads (bd_sample_choose_rows + bd_backbone_forward_rows):
the positions a step draws are fixed by the ids before the denoiser runs (reference sampler.py:676-686: randperm over
the masked positions, no logits involved), so the last feed-forward block and the final LayerNorm run for those rows
alone. Checked here: the row list equals the oracle's choice, the rows written equal the full forward's bit for bit,
and the samplers return the same ids with the shortcut on and off."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_gpu_sampler_e2e import build_model
from bidiff import _lib
from bidiff.sampling import core
from oracle import sampler_ref as ora

pytestmark = pytest.mark.gpu


def _choose(smp, ids, fixed, t, T, row_cap, seed=11, seq_offset=3, variant_extra=0, u_sel=None):
    """bd_sample_choose_rows on the sampler's tables: (row_pos [B, cap] int32, need_all [B] uint8) on the device."""
    c = smp._core
    tt, dt = c.tt, c._tables()
    B, L = ids.shape
    a = _lib.StepArgs()
    core.fill_step_args(a, tt, dt)
    ktab = torch.from_numpy(core.unmask_count_table(T, L)).to(ids.device)
    a.ids = ids.data_ptr()
    a.fixed = None if fixed is None else fixed.data_ptr()
    a.u_sel = None if u_sel is None else u_sel.data_ptr()
    a.ktable = ktab[t].data_ptr()
    a.B, a.L, a.d = B, L, c.model.backbone.hidden_size
    a.variant, a.use_constraints, a.t = tt.variant | variant_extra, 1, t
    a.temperature, a.seed, a.seq_offset = 1.0, seed, seq_offset
    row_pos = torch.full((B * row_cap,), -7, dtype=torch.int32, device=ids.device)
    need_all = torch.full((B,), 9, dtype=torch.uint8, device=ids.device)
    _lib.check(_lib.lib().bd_sample_choose_rows(C.byref(a), row_pos.data_ptr(), need_all.data_ptr(), row_cap,
                                                _lib.stream_handle()), "bd_sample_choose_rows")
    torch.cuda.synchronize()
    return row_pos.view(B, row_cap), need_all, ktab


def _partially_masked(tok, B, L, rng, frac, lengths=None):
    ids = rng.randint(5, tok.vocab_size, size=(B, L)).astype(np.int64)
    ids[:, 0] = tok.bos_token_id
    attn = np.ones((B, L), dtype=np.int64)
    for b in range(B):
        n = L if lengths is None else int(lengths[b])
        ids[b, n - 1] = tok.eos_token_id
        ids[b, n:] = tok.pad_token_id
        attn[b, n:] = 0
        free = np.arange(1, n - 1)
        pick = free[rng.rand(free.size) < frac[b % len(frac)]]
        ids[b, pick] = tok.mask_token_id
    return ids, attn


@pytest.mark.parametrize("variant", ["smiles", "selfies", "group"])
@pytest.mark.parametrize("L,lengths", [(128, False), (64, False), (32, False), (96, True), (20, False)])
def test_chosen_rows_equal_the_full_forward(gpu_device, variant, L, lengths):
    """Row list against the oracle's choice (explicit keys), then the rows the forward writes against the full
    forward: the same bits at the listed positions and for every row of a flagged sequence; any other row is kept as
    it was or finished too.
    L = 64 / 32 / 20: several sequences share a workgroup; lengths: a wave table deals ragged sequences."""
    dev = gpu_device
    T, B, cap = 12, 37, 2
    _, tok = H.make_sampler(variant)
    model = build_model(tok, dev, T, heads=3, layers=2, d=384, F=1536)
    smp, _ = H.make_sampler(variant, model=model, num_steps=T, device=str(dev))
    bb = model.backbone
    rng = np.random.RandomState(L + len(variant))
    lens = rng.randint(8, L + 1, size=B) if lengths else None
    if lengths:
        lens[0] = L
    # mask fractions from "nearly done" (few masks: the SELFIES urgency rule reads every row) to "all masked"
    ids_np, attn_np = _partially_masked(tok, B, L, rng, frac=[0.03, 0.1, 0.5, 1.0], lengths=lens)
    ids = torch.from_numpy(ids_np).to(dev)
    keymask = torch.from_numpy((attn_np != 0).astype(np.uint8)).to(dev)
    fixed = (ids != tok.mask_token_id).to(torch.uint8) if variant == "smiles" else None
    u_sel = torch.from_numpy(rng.rand(B, L).astype(np.float32)).to(dev)
    t = 5
    row_pos, need_all, ktab = _choose(smp, ids, fixed, t, T, cap, u_sel=u_sel)

    # the oracle's choice: the k smallest keys among the selectable masks, in draw order
    tb = H.oracle_tables(smp)
    rp, na = row_pos.cpu().numpy(), need_all.cpu().numpy()
    kt = core.unmask_count_table(T, L)[t]
    for b in range(B):
        selectable = (ids_np[b] == tok.mask_token_id)
        if fixed is not None:
            selectable &= fixed[b].cpu().numpy() == 0
        pos = np.nonzero(selectable)[0]
        n = pos.size
        k = 0 if n == 0 else min(n, max(1, int(kt[n])))
        keys = u_sel[b].cpu().numpy()[pos].view(np.uint32) & 0x7FFFFFFF
        order = pos[np.lexsort((pos, keys))][:k]
        if tb.variant & core.VB_SORTED:
            order = np.sort(order)
        want = np.full(cap, -1)
        want[:min(k, cap)] = order[:cap]
        np.testing.assert_array_equal(rp[b], want, err_msg=f"sequence {b}")
        n_mask = int((ids_np[b] == tok.mask_token_id).sum())
        total = int(sum(tb.contrib[v] for v in ids_np[b] if v != tok.mask_token_id))
        urgent = bool(tb.variant & core.VB_URGENCY) and 2 - total > 0 and 0 < n_mask <= 5 * (2 - total)
        assert na[b] == int(k > cap or urgent), f"sequence {b}: need_all {na[b]}, k {k}, urgent {urgent}"
    assert na.sum() > 0 and (na == 0).sum() > 0, "the case should hold both kinds of sequences"

    ts = torch.full((B,), t, device=dev, dtype=torch.long)
    table = bb.wave_table_for_mask(keymask) if lengths else None
    if lengths:
        assert table is not None
    kw = {} if table is None else {"wave_table": table}
    full = torch.zeros(B, L, 384, dtype=torch.bfloat16, device=dev)
    bb.run_forward(ids, ts, keymask, want_hidden=True, hidden_out=full, **kw)
    sentinel = 0x7FC1        # a NaN pattern no computation writes
    got = torch.full((B, L, 384), sentinel, dtype=torch.int16, device=dev).view(torch.bfloat16)
    bb.run_forward_rows(ids, ts, keymask, row_pos.reshape(-1), need_all, cap, got, wave_table=table)
    torch.cuda.synchronize()
    g, f = got.view(torch.int16).cpu().numpy(), full.view(torch.int16).cpu().numpy()
    n_rows = n_kept = 0
    for b in range(B):
        n_valid = L if lens is None else int(lens[b])
        if na[b]:
            rows = range((n_valid + 31) // 32 * 32 if table is not None else L)
            rows = [r for r in rows if r < L]
        else:
            rows = [int(r) for r in rp[b] if r >= 0]
        for r in rows:
            np.testing.assert_array_equal(g[b, r], f[b, r], err_msg=f"sequence {b} row {r}")
        # any other row keeps what the buffer held, or (a sequence sharing a workgroup with a flagged one) is finished too
        for r in set(range(L)) - set(rows):
            untouched = (g[b, r] == np.int16(sentinel)).all()
            assert untouched or np.array_equal(g[b, r], f[b, r]), f"sequence {b} row {r}: neither kept nor finished"
            n_kept += int(untouched)
        n_rows += len(rows)
    assert n_rows > 0 and n_kept > 0


@pytest.mark.parametrize("variant,kw", [("smiles", {}), ("smiles", {"lengths": True}), ("selfies", {}),
                                        ("selfies", {"natural": True}), ("group", {}), ("bpe", {}),
                                        ("bpe", {"lengths": True}), ("smiles", {"constraints": False})])
def test_samplers_draw_the_same_ids_with_rows_chosen_first(gpu_device, variant, kw):
    """Whole reverse loops at the benchmark architecture (d 384, 3 heads of 128), the shortcut on and off."""
    dev = gpu_device
    T, B, L = 20, 21, 128
    _, tok = H.make_sampler(variant)
    model = build_model(tok, dev, T, heads=3, layers=3, d=384, F=1536)
    outs = []
    for first in (True, False):
        smp, _ = H.make_sampler(variant, model=model, num_steps=T, temperature=0.9, device=str(dev),
                                use_constraints=kw.get("constraints", True))
        smp._core.seed = 99
        smp._core.choose_first = first
        if kw.get("lengths"):
            lengths = [int(x) for x in np.random.RandomState(2).randint(2, L + 1, size=B)]
            ids, strings = smp.sample_with_lengths(lengths, max_length=L, show_progress=False, seq_offset=5)
        elif kw.get("natural"):
            ids, strings = smp.sample(B, L, show_progress=False, allow_natural_eos=True, seq_offset=5)
        else:
            ids, strings = smp.sample(B, L, show_progress=False, seq_offset=5)
        outs.append((ids.cpu(), strings))
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    assert (outs[0][0] != tok.mask_token_id).all()


def test_rows_forward_argument_errors(gpu_device):
    dev = gpu_device
    _, tok = H.make_sampler("smiles")
    model = build_model(tok, dev, 4, heads=3, layers=1, d=384, F=1536)
    bb = model.backbone
    B, L, cap = 3, 16, 2
    ids = torch.full((B, L), tok.mask_token_id, dtype=torch.long, device=dev)
    ts = torch.ones(B, dtype=torch.long, device=dev)
    rp = torch.full((B * cap,), -1, dtype=torch.int32, device=dev)
    na = torch.zeros(B, dtype=torch.uint8, device=dev)
    hid = torch.empty(B, L, 384, dtype=torch.bfloat16, device=dev)
    with pytest.raises(ValueError):
        bb.run_forward_rows(ids, ts, None, rp[:-1], na, cap, hid)
    with pytest.raises(ValueError):
        bb.run_forward_rows(ids, ts, None, rp, na, cap, hid.float())
    lib = _lib.lib()
    pk = bb.packed()
    ws, need = bb._workspace(pk, B, L, dev)
    args = [C.byref(pk.desc), _lib.ptr(ids), _lib.ptr(ts), None, None, 0, B, L, _lib.ptr(rp), _lib.ptr(na)]
    assert lib.bd_backbone_forward_rows(*args, 0, _lib.ptr(ws), need, _lib.ptr(hid), _lib.stream_handle()) != 0
    assert b"row_cap" in lib.bd_last_error()
    assert lib.bd_backbone_forward_rows(*args, cap, _lib.ptr(ws), 16, _lib.ptr(hid), _lib.stream_handle()) != 0
    assert b"workspace" in lib.bd_last_error()
    assert lib.bd_backbone_forward_rows(*args, cap, _lib.ptr(ws), need, None, _lib.stream_handle()) != 0
    # L beyond the single-launch forward: not supported, the reverse loop then keeps the full forward
    assert not bb.rows_forward_supported(200)
    assert bb.rows_forward_supported(128)


# ---------------------------------------------------------------- large models on the per-layer kernels
@pytest.mark.parametrize("d,heads,F,B,L,cap,ragged", [(768, 6, 3072, 88, 128, 2, False), (1024, 8, 4096, 24, 512, 8, True)])
def test_per_layer_chosen_rows_equal_the_full_forward(gpu_device, d, heads, F, B, L, cap, ragged):
    """Models without the single-launch forward (the reference's medium size; the stress shape with L = 512): the chosen
    rows go through the last layer's output projection, feed-forward block and the final LayerNorm as gathered rows on
    the same GEMM kernel as the full forward. Same bits at the listed positions, every other row untouched."""
    dev = gpu_device
    T = 12
    _, tok = H.make_sampler("smiles")
    model = build_model(tok, dev, T, heads=heads, layers=2, d=d, F=F)
    smp, _ = H.make_sampler("smiles", model=model, num_steps=T, device=str(dev))
    bb = model.backbone
    assert bb.rows_forward_mode(B, L) == "per_layer"
    if os.environ.get("BD_GEMM8P", "1") != "2":        # (2 forces the eight-wave kernel on every shape)
        assert bb.rows_forward_mode(2, L) is None      # too few rows for the large-tile kernels: full forward
    rng = np.random.RandomState(d + L)
    lens = rng.randint(L // 4, L + 1, size=B) if ragged else None
    ids_np, attn_np = _partially_masked(tok, B, L, rng, frac=[0.05, 0.3, 0.6, 1.0], lengths=lens)
    ids = torch.from_numpy(ids_np).to(dev)
    keymask = torch.from_numpy((attn_np != 0).astype(np.uint8)).to(dev)
    fixed = (ids != tok.mask_token_id).to(torch.uint8)
    t = 5
    row_pos, need_all, _ = _choose(smp, ids, fixed, t, T, cap)
    rp = row_pos.cpu().numpy()
    assert (rp >= 0).sum() > B // 2 and (rp < 0).sum() > 0
    ts = torch.full((B,), t, device=dev, dtype=torch.long)
    full = torch.zeros(B, L, d, dtype=torch.bfloat16, device=dev)
    bb.run_forward(ids, ts, keymask, want_hidden=True, hidden_out=full)
    sentinel = 0x7FC1
    got = torch.full((B, L, d), sentinel, dtype=torch.int16, device=dev).view(torch.bfloat16)
    bb.run_forward_rows(ids, ts, keymask, row_pos.reshape(-1), need_all, cap, got)
    torch.cuda.synchronize()
    g, f = got.view(torch.int16).cpu().numpy(), full.view(torch.int16).cpu().numpy()
    listed = np.zeros((B, L), dtype=bool)
    for b in range(B):
        for r in rp[b]:
            if r >= 0:
                listed[b, r] = True
    np.testing.assert_array_equal(g[listed], f[listed])
    assert (g[~listed] == np.int16(sentinel)).all()
    # the deferred LayerNorm rounds differently in the full forward: the shortcut steps aside
    bb.defer_layernorm = True
    assert bb.rows_forward_mode(B, L) is None
    bb.defer_layernorm = False


def test_per_layer_sampler_draws_the_same_ids_with_rows_chosen_first(gpu_device):
    """A whole reverse loop on a medium-size model (per-layer kernels), the shortcut on and off; T < L / 16 makes some
    steps draw more rows than the list holds, which sends those steps through the full forward."""
    dev = gpu_device
    T, B, L = 6, 88, 128
    _, tok = H.make_sampler("smiles")
    model = build_model(tok, dev, T, heads=6, layers=2, d=768, F=3072)
    outs = []
    for first in (True, False):
        smp, _ = H.make_sampler("smiles", model=model, num_steps=T, temperature=0.9, device=str(dev))
        smp._core.seed = 5
        smp._core.choose_first = first
        ids, strings = smp.sample(B, L, show_progress=False, seq_offset=2)
        outs.append((ids.cpu(), strings))
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    assert (outs[0][0] != tok.mask_token_id).all()


def test_per_layer_rows_forward_argument_errors(gpu_device):
    dev = gpu_device
    _, tok = H.make_sampler("smiles")
    model = build_model(tok, dev, 4, heads=6, layers=1, d=768, F=3072)
    bb = model.backbone
    B, L, cap = 88, 128, 2
    lib = _lib.lib()
    pk = bb.packed()
    ws, need = bb._workspace(pk, B, L, dev)
    rneed = lib.bd_backbone_rows_workspace_bytes(C.byref(pk.desc), B, cap)
    assert rneed > 0 and lib.bd_backbone_rows_workspace_bytes(C.byref(pk.desc), B, 0) < 0
    rws = torch.empty(rneed, dtype=torch.uint8, device=dev)
    ids = torch.full((B, L), tok.mask_token_id, dtype=torch.long, device=dev)
    ts = torch.ones(B, dtype=torch.long, device=dev)
    rp = torch.full((B * cap,), -1, dtype=torch.int32, device=dev)
    hid = torch.empty(B, L, 768, dtype=torch.bfloat16, device=dev)
    st = _lib.stream_handle()
    call = lambda b, cap_, rbytes, out: lib.bd_backbone_forward_rows_unfused(
        C.byref(pk.desc), _lib.ptr(ids), _lib.ptr(ts), None, b, L, _lib.ptr(rp), cap_, _lib.ptr(ws), need, _lib.ptr(rws), rbytes,
        out, st)
    assert call(B, cap, rneed, _lib.ptr(hid)) == 0            # every slot empty: nothing written, no fault
    assert call(B, 0, rneed, _lib.ptr(hid)) != 0 and b"row_cap" in lib.bd_last_error()
    assert call(B, cap, 16, _lib.ptr(hid)) != 0 and b"rows workspace" in lib.bd_last_error()
    assert call(B, cap, rneed, None) != 0
    assert call(2, cap, rneed, _lib.ptr(hid)) != 0 and b"large-tile" in lib.bd_last_error()
    torch.cuda.synchronize()
    # the forced large-tile product: shape rules of the eight-wave kernel
    a = torch.zeros(64, 256, dtype=torch.bfloat16, device=dev)
    o = torch.zeros(64, 256, dtype=torch.bfloat16, device=dev)
    assert lib.bd_gemm_bf16_tile256(_lib.ptr(a), _lib.ptr(a), None, _lib.ptr(o), 64, 256, 256, 0, st) == 0
    assert lib.bd_gemm_bf16_tile256(_lib.ptr(a), _lib.ptr(a), None, _lib.ptr(o), 64, 256, 96, 0, st) != 0
    assert lib.bd_gemm_bf16_tile256(_lib.ptr(a), _lib.ptr(a), None, _lib.ptr(o), 64, 256, 256, 3, st) != 0
    torch.cuda.synchronize()
