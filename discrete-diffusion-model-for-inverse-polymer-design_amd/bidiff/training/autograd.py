"""Autograd bridge of the training step: forward passes that save what the HIP backward
kernels need, and backward passes that chain those kernels.

``backbone_forward_train`` is what ``DiffusionBackbone.forward`` runs when autograd is
recording (reference call site: ``DiscreteMaskingDiffusion.forward``, diffusion.py:156, under
``BackboneTrainer._train_step``, trainer_backbone.py:434-439); ``masked_cross_entropy`` is
``_compute_loss`` (diffusion.py:169-209). In training mode with ``dropout > 0`` the four dropout
sites of the reference (embedding sum backbone.py:247, attention probabilities :65, GELU output :98,
fc2 output :100) are applied inside the kernels from a counter-based decision stream that the
backward kernels regenerate; no mask tensor exists.
"""
import ctypes as C
import os

import torch

from .. import _lib


_STREAM = [None]


def _enter():
    """Called on entry of every autograd forward / backward of this module: looks the current HIP stream up once
    (about 9 us through torch) for the fifty-odd launches that follow."""
    _STREAM[0] = _lib.stream_handle()


def _s():
    return _STREAM[0]


_SIDE = {}


def _side_stream(dev):
    """Second HIP stream of the device for the weight-gradient GEMMs (``BD_ASYNC_WGRAD=0`` turns it off)."""
    if os.environ.get("BD_ASYNC_WGRAD", "1") == "0":
        return None
    if dev not in _SIDE:
        _SIDE[dev] = torch.cuda.Stream(device=dev)
    return _SIDE[dev]


# (callable, layer index): called once per backward pass, in direct-accumulation mode only, when every gradient of
# the layers >= that index, the final norm and the vocabulary head has been enqueued - on the stream whose order
# covers all of them (the weight-gradient stream when it is in use). The data-parallel trainer starts the
# all-reduce of that part of the flat gradient buffer there (parallel.GradientExchange).
_TAIL_HOOK = [None, -1]


def set_tail_ready_hook(fn, layer_index: int = -1, model=None):
    """``model`` (a DiffusionBackbone): the hook belongs to that backbone's backward passes alone, so a second trainer
    or a property head fine-tuning another backbone in the same process neither inherits nor overwrites it
    (``fn = None`` removes it). Without ``model`` the hook applies to every backbone that has none of its own."""
    if model is not None:
        if fn is None:
            model.__dict__.pop("_tail_ready_hook", None)
        else:
            model.__dict__["_tail_ready_hook"] = (fn, int(layer_index))
        return
    _TAIL_HOOK[0], _TAIL_HOOK[1] = fn, int(layer_index)


def _tail_hook_of(model):
    own = model.__dict__.get("_tail_ready_hook")
    return own if own is not None else (_TAIL_HOOK[0], _TAIL_HOOK[1])


def _ck(rc, what):
    if rc != 0:
        _lib.check(rc, what)


def _dp(drop):
    return None if drop is None else C.byref(drop)


def _gemm(A, W, bias, out, M, n_pad, n_real, K, ldc, epi, aux_in=None, aux_out=None, drop=None, site=0):
    _ck(_lib.lib().bd_gemm_bf16_drop(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(aux_in),
                                     _lib.ptr(aux_out), M, n_pad, n_real, K, ldc, epi, _dp(drop), site, _s()),
        "bd_gemm_bf16")


def _ln(x, g, b, out_bf16, stats, M, d, eps):
    _ck(_lib.lib().bd_layernorm(_lib.ptr(x), _lib.ptr(g), _lib.ptr(b), _lib.ptr(out_bf16), None, _lib.ptr(stats),
                                M, d, eps, _s()), "bd_layernorm")


def _ln_bwd(dy, x, stats, g, dres, dx, dg, db, M, d, dx_bf16=None, drop=None, site=0, drop_dx=False, parts=None):
    """dx (fp32) and optionally its bf16 copy (dropped at ``site`` when ``drop`` is given) in one pass;
    ``drop_dx``: no bf16 copy, the dropout applies to dx itself."""
    use_drop = drop if (dx_bf16 is not None or drop_dx) else None
    if parts is not None:      # gamma / beta sums into the next slot of the copies; parts.finish() adds them to dg / db
        _ck(_lib.lib().bd_layernorm_bwd_parts(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(stats), _lib.ptr(g), _lib.ptr(dres),
                                              _lib.ptr(dx), _lib.ptr(parts.slot(dg, db)), parts.n_parts, M, d,
                                              _lib.ptr(dx_bf16), _dp(use_drop), site, _s()), "bd_layernorm_bwd_parts")
        return
    _ck(_lib.lib().bd_layernorm_bwd_cast(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(stats), _lib.ptr(g), _lib.ptr(dres),
                                         _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), M, d, _lib.ptr(dx_bf16),
                                         _dp(use_drop), site, _s()), "bd_layernorm_bwd")


class _LnPartsJob(C.Structure):
    _fields_ = [("parts", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p)]


class _LnParts:
    """Copies of the LayerNorm gamma / beta gradient rows that one backward pass's LayerNorm launches add to
    (``n_parts`` copies per launch: an address then takes n_parts times fewer atomics, and those retire one after
    another), and the launch that adds them to the parameters' gradients. ``slot`` hands out the copies in call
    order; ``finish`` covers the slots handed out since the last ``finish`` and leaves them zeroed."""
    n_parts = 8

    def __init__(self, n_slots, d, device):
        self.buf = torch.zeros(n_slots, self.n_parts, 2, d, device=device)
        self.d = d
        self.used, self.done, self.dst = 0, 0, []
        self._tables = {}
        self.dirty = False

    def begin(self):
        if self.dirty:                 # a backward pass that did not reach its last finish()
            self.buf.zero_()
        self.used, self.done, self.dst = 0, 0, []
        self.dirty = True

    def slot(self, dgamma, dbeta):
        self.dst.append((dgamma, dbeta))
        self.used += 1
        return self.buf[self.used - 1]

    def finish(self, last=False):
        if self.used > self.done:
            key = (self.done,) + tuple(t.data_ptr() for pair in self.dst[self.done:self.used] for t in pair)
            table = self._tables.get(key)
            if table is None:
                jobs = [_LnPartsJob(self.buf[k].data_ptr(), self.dst[k][0].data_ptr(), self.dst[k][1].data_ptr())
                        for k in range(self.done, self.used)]
                arr = (_LnPartsJob * len(jobs))(*jobs)
                if len(self._tables) > 8:
                    self._tables.clear()
                table = self._tables[key] = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.buf.device)
            _ck(_lib.lib().bd_layernorm_parts_finish(_lib.ptr(table), self.used - self.done, self.n_parts, self.d, _s()),
                "bd_layernorm_parts_finish")
            self.done = self.used
        if last:
            self.dirty = False


def _wgrad(dY, X, dW, M, N, K, db=None):
    """dW += dY^T X and (optionally) db += column sums of dY, one launch. ``dW`` / ``db`` may be lists of
    up to three destinations for equal row segments of dY (the stacked q | k | v gradient)."""
    dws = dW if isinstance(dW, (list, tuple)) else [dW]
    dbs = db if isinstance(db, (list, tuple)) else ([db] if db is not None else None)
    n = len(dws)
    pw = (C.c_void_p * n)(*[t.data_ptr() for t in dws])
    pb = None if dbs is None else (C.c_void_p * n)(*[t.data_ptr() for t in dbs])
    _ck(_lib.lib().bd_gemm_wgrad_segments(_lib.ptr(dY), dY.stride(0), _lib.ptr(X), X.stride(0), pw, pb, n, N // n, M, K,
                                          _s()), "bd_gemm_wgrad_segments")


def _wgrad_group(jobs):
    """The weight-gradient products of one hand-over ((dY, X, dW, M, N, K, db) each, as for ``_wgrad``) in one launch:
    every output tile owned by one workgroup (bd_gemm_wgrad_group)."""
    arr = (_lib.WgradJob * len(jobs))()
    M = jobs[0][3]
    for j, (dY, X, dW, M_, N, K, db) in zip(arr, jobs):
        if M_ != M:
            raise ValueError("weight-gradient jobs of one launch share their token range")
        dws = dW if isinstance(dW, (list, tuple)) else [dW]
        dbs = db if isinstance(db, (list, tuple)) else ([db] if db is not None else [])
        j.dY, j.X = dY.data_ptr(), X.data_ptr()
        for k, t in enumerate(dws):
            j.dW[k] = t.data_ptr()
        for k, t in enumerate(dbs):
            j.db[k] = t.data_ptr()
        j.ldy, j.ldx, j.n_seg, j.seg_rows, j.K = dY.stride(0), X.stride(0), len(dws), N // len(dws), K
    _ck(_lib.lib().bd_gemm_wgrad_group(arr, len(jobs), M, -1, _s()), "bd_gemm_wgrad_group")


def _colsum(dY, out, M, N):
    _ck(_lib.lib().bd_colsum_bf16(_lib.ptr(dY), dY.stride(0), _lib.ptr(out), M, N, _s()), "bd_colsum_bf16")


def _to_bf16(x, out, drop=None, site=0):
    if drop is None:
        _ck(_lib.lib().bd_cast_f32_to_bf16(_lib.ptr(x), _lib.ptr(out), x.numel(), _s()), "bd_cast_f32_to_bf16")
    else:
        _ck(_lib.lib().bd_dropout_f32_to_bf16(_lib.ptr(x), _lib.ptr(out), x.shape[0], x.shape[1], _dp(drop), site,
                                              _s()), "bd_dropout_f32_to_bf16")


class _PinnedRing:
    """Small host -> device uploads that never block the host nor reuse a buffer a copy may still read."""

    def __init__(self, slots=4):
        self.slots, self.pos, self.ring = slots, 0, []

    def upload_i32(self, arr, dev):
        import numpy as np
        n = int(arr.size)
        if len(self.ring) < self.slots:
            self.ring.append([torch.empty(max(n, 4096), dtype=torch.int32).pin_memory(), None])
            slot = self.ring[-1]
        else:
            slot = self.ring[self.pos % self.slots]
            self.pos += 1
            if slot[1] is not None:
                slot[1].synchronize()
            if slot[0].numel() < n:
                slot[0] = torch.empty(n, dtype=torch.int32).pin_memory()
        slot[0][:n].copy_(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)))
        out = slot[0][:n].to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        slot[1] = ev
        return out


_RING = _PinnedRing(slots=16)


class RowPacking:
    """Packed-row layout of one [B, L] batch: the rows behind each sequence's last attended position are left out
    (they receive no attention as keys, carry no loss and no gradient). ``lengths`` (host integers, one per
    sequence, = last attended position + 1) fixes the layout without a device round trip."""

    def __init__(self, lengths, B, L, device, pad_to=128):
        import numpy as np
        ln = np.asarray(lengths, dtype=np.int64).reshape(-1)
        if ln.size != B or (ln < 0).any() or (ln > L).any():
            raise ValueError("RowPacking: lengths must hold one value in [0, L] per sequence")
        cu = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(ln, out=cu[1:])
        self.B, self.L = int(B), int(L)
        self.n_real = int(cu[-1])
        self.n_rows = max(pad_to, (self.n_real + pad_to - 1) // pad_to * pad_to)
        self.max_len = int(ln.max()) if B else 0
        self.cu = _RING.upload_i32(cu, device)
        self.rowmap = torch.empty(self.n_rows, dtype=torch.int32, device=device)
        _ck(_lib.lib().bd_build_rowmap(_lib.ptr(self.cu), self.B, self.L, self.n_rows, _lib.ptr(self.rowmap),
                                       _lib.stream_handle()), "bd_build_rowmap")
        self._gather = None

    @staticmethod
    def lengths_from_mask(attention_mask):
        """Last attended position + 1 per sequence (one device -> host copy)."""
        am = attention_mask != 0
        L = am.shape[1]
        last = L - am.flip(1).to(torch.int8).argmax(dim=1)
        return torch.where(am.any(dim=1), last, torch.zeros_like(last)).cpu().numpy()

    def gather_index(self):
        """int64 index of each packed row's logical token (fillers point at token 0) and the filler mask."""
        if self._gather is None:
            rm = self.rowmap.to(torch.long)
            self._gather = (rm.clamp(min=0), rm < 0)
        return self._gather


class _TrainWeights:
    """bf16 copies (and transposes for the data-gradient GEMMs) of the current parameters in persistent buffers
    that ONE launch (bd_multi_cast_bf16 over a job table of all matrices) refreshes after each optimizer step.

    Head sizes the attention kernels do not tile (e.g. 48) run zero-padded to 32 / 64 / 128 (``dhp``): the copies of
    the q / k / v projections have their rows, the copy of the output projection its columns, re-blocked head by head
    by the same job table (the padding entries are zeroed once and never written); the q rows absorb
    sqrt(dhp / dh) because the kernels scale by 1 / sqrt(dhp). Their weight gradients are accumulated in padded
    fp32 buffers (``pad_grads``) and added to the parameters' gradients by one bd_multi_unpad_add_f32 launch."""

    def __init__(self, model):
        d, V, H = model.hidden_size, model.vocab_size, model.num_heads
        dh = d // H
        if dh > 128:
            raise NotImplementedError("bidiff attention kernels support head_dim <= 128")
        self.dh = dh
        self.dhp = dhp = dh if dh in (32, 64, 128) else (32 if dh < 32 else 64 if dh < 64 else 128)
        self.aw = aw = H * dhp
        if aw % 128:
            raise NotImplementedError(f"bidiff: num_heads * padded head_dim = {aw} must be a multiple of 128")
        self.qfix = (dhp / dh) ** 0.5
        self.vpad = (V + 127) // 128 * 128
        self.ptr_key = tuple(p.data_ptr() for p in model._params())
        self.padded = dhp != dh
        self._build(model)

    def _build(self, model):
        d, F, V, aw = model.hidden_size, model.ffn_hidden_size, model.vocab_size, self.aw
        dh, dhp, padded = self.dh, self.dhp, self.padded
        dev = model.output_proj.weight.device
        bf = torch.bfloat16
        jobs = []

        def job(src, dst, ld, mode, row_off=0, col_off=0, pad_axis=0, scale=1.0):
            rows, cols = (src.shape if src.dim() == 2 else (1, src.shape[0]))
            esz = 4 if mode == 2 else 2
            if not padded:
                pad_axis = 0
            jobs.append(_lib.CastJob(src.data_ptr(), dst.data_ptr() + esz * (row_off * ld + col_off), rows, cols, ld, mode,
                                     pad_axis, dh, dhp, float(scale)))

        new = torch.zeros if padded else torch.empty          # padding entries stay zero
        self.layers = []
        n_l = len(model.layers)
        # padded fp32 accumulators of the attention projections' weight gradients, one flat buffer zeroed per backward
        self.pad_grads = torch.zeros(n_l * (3 * aw * d + 3 * aw + d * aw), device=dev) if padded else None
        self.pad_views, unpad = [], []

        def unjob(src, dst_param, ld, pad_axis, scale):
            rows, cols = (dst_param.shape if dst_param.dim() == 2 else (1, dst_param.shape[0]))
            unpad.append((src, dst_param, rows, cols, ld, pad_axis, float(scale)))

        off = 0
        for blk in model.layers:
            a = blk.attn
            w = {"wqkv": new(3 * aw, d, dtype=bf, device=dev), "wqkvT": new(d, 3 * aw, dtype=bf, device=dev),
                 "bqkv": new(3 * aw, device=dev),
                 "wo": new(d, aw, dtype=bf, device=dev), "woT": new(aw, d, dtype=bf, device=dev),
                 "w1": torch.empty(F, d, dtype=bf, device=dev), "w1T": torch.empty(d, F, dtype=bf, device=dev),
                 "w2": torch.empty(d, F, dtype=bf, device=dev), "w2T": torch.empty(F, d, dtype=bf, device=dev)}
            for i, pr in enumerate((a.q_proj, a.k_proj, a.v_proj)):
                f = self.qfix if i == 0 else 1.0
                job(pr.weight, w["wqkv"], d, 0, row_off=i * aw, pad_axis=1, scale=f)
                job(pr.weight, w["wqkvT"], 3 * aw, 1, col_off=i * aw, pad_axis=1, scale=f)
                job(pr.bias, w["bqkv"], 3 * aw, 2, col_off=i * aw, pad_axis=2, scale=f)
            job(a.out_proj.weight, w["wo"], aw, 0, pad_axis=2)
            job(a.out_proj.weight, w["woT"], d, 1, pad_axis=2)
            job(blk.ffn.fc1.weight, w["w1"], d, 0)
            job(blk.ffn.fc1.weight, w["w1T"], F, 1)
            job(blk.ffn.fc2.weight, w["w2"], F, 0)
            job(blk.ffn.fc2.weight, w["w2T"], d, 1)
            self.layers.append(w)
            if padded:
                g = self.pad_grads
                gw = [g[off + i * aw * d: off + (i + 1) * aw * d].view(aw, d) for i in range(3)]
                off += 3 * aw * d
                gb = [g[off + i * aw: off + (i + 1) * aw] for i in range(3)]
                off += 3 * aw
                go = g[off: off + d * aw].view(d, aw)
                off += d * aw
                self.pad_views.append({"w": gw, "b": gb, "o": go})
                for i, pr in enumerate((a.q_proj, a.k_proj, a.v_proj)):
                    f = self.qfix if i == 0 else 1.0
                    unjob(gw[i], pr.weight, d, 1, f)
                    unjob(gb[i], pr.bias, aw, 2, f)
                unjob(go, a.out_proj.weight, aw, 2, 1.0)
        self.unpad = unpad
        self.wout = torch.zeros(self.vpad, d, dtype=bf, device=dev)          # rows past V stay zero
        self.woutT = torch.zeros(d, self.vpad, dtype=bf, device=dev)
        job(model.output_proj.weight, self.wout, d, 0)
        job(model.output_proj.weight, self.woutT, self.vpad, 1)
        arr = (_lib.CastJob * len(jobs))(*jobs)
        self.jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.n_jobs = len(jobs)
        self.max_rows = max(j.rows for j in jobs)
        self.max_cols = max(j.cols for j in jobs)
        self._unpad_key, self._unpad_jobs = None, None
        self._ln_parts = None
        self.refresh(model)

    def refresh(self, model):
        _ck(_lib.lib().bd_multi_cast_bf16(_lib.ptr(self.jobs), self.n_jobs, self.max_rows, self.max_cols, _s()),
            "bd_multi_cast_bf16")

    def ln_parts(self, d, device):
        """The backward pass's LayerNorm partial-sum copies (two LayerNorms per layer and the final one)."""
        if self._ln_parts is None:
            self._ln_parts = _LnParts(2 * len(self.layers) + 1, d, device)
        return self._ln_parts

    def unpad_gradients(self, gbuf, first_layer=0, end_layer=None):
        """pad_grads -> the parameters' gradient buffers (``gbuf(param)``) for the layers [first_layer, end_layer), one
        launch. The job table is rebuilt only when a destination moved (gradients bound to the optimizer's flat
        buffer keep their addresses)."""
        dsts = [gbuf(p) for _, p, *_ in self.unpad]
        key = tuple(t.data_ptr() for t in dsts)
        if key != self._unpad_key:
            jobs = [_lib.CastJob(src.data_ptr(), dst.data_ptr(), rows, cols, ld, 0, pad_axis, self.dh, self.dhp, scale)
                    for (src, _, rows, cols, ld, pad_axis, scale), dst in zip(self.unpad, dsts)]
            arr = (_lib.CastJob * len(jobs))(*jobs)
            self._unpad_jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.pad_grads.device)
            self._unpad_key = key
            self._unpad_dims = (len(jobs), max(j.rows for j in jobs), max(j.cols for j in jobs))
        n, mr, mc = self._unpad_dims
        per_layer = n // len(self.layers)              # the table is in layer order, the same jobs for every layer
        end_layer = len(self.layers) if end_layer is None else end_layer
        if end_layer <= first_layer:
            return
        first = C.c_void_p(self._unpad_jobs.data_ptr() + first_layer * per_layer * C.sizeof(_lib.CastJob))
        _ck(_lib.lib().bd_multi_unpad_add_f32(first, (end_layer - first_layer) * per_layer, mr, mc, _s()),
            "bd_multi_unpad_add_f32")


def _train_weights(model):
    key = model._pack_key()
    tw = getattr(model, "_train_w", None)
    if tw is not None and tw[0] == key:
        return tw[1]
    if tw is not None and tw[1].ptr_key == tuple(p.data_ptr() for p in model._params()):
        tw[1].refresh(model)          # same storage, new values: one launch
        model._train_w = (key, tw[1])
        return tw[1]
    tw = (key, _TrainWeights(model))
    model._train_w = tw
    return tw[1]


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, ids, timesteps, keymask, mode, pk, *params):
        """mode 0: logits [B, L, V] (the diffusion training step); mode 1: the final-norm hidden states
        [B, L, d] in fp32 (what get_hidden_states / get_pooled_output return to a property head that
        fine-tunes the last blocks, reference property_head.py:102-112)."""
        _enter()
        lib = _lib.lib()
        B, L = ids.shape
        M = B * L if pk is None else pk.n_rows       # token rows every kernel below works on
        d, F, V, H = model.hidden_size, model.ffn_hidden_size, model.vocab_size, model.num_heads
        dev = ids.device
        bf = torch.bfloat16
        eps = float(model.final_norm.eps)
        tw = _train_weights(model)
        aw = tw.aw
        drop = model._next_dropout()          # None in eval mode or with dropout = 0
        x = torch.empty(M, d, device=dev)
        if pk is None:
            _ck(lib.bd_embed_drop(_lib.ptr(ids), _lib.ptr(timesteps), _lib.ptr(model.token_embedding.weight),
                                  _lib.ptr(model.position_embedding.weight), _lib.ptr(model.timestep_embedding.weight),
                                  _lib.ptr(x), B, L, d, V, model.max_position_embeddings, model.num_diffusion_steps + 1,
                                  _dp(drop), 0, _s()), "bd_embed")
        else:
            _ck(lib.bd_embed_rows(_lib.ptr(ids), _lib.ptr(timesteps), _lib.ptr(pk.rowmap),
                                  _lib.ptr(model.token_embedding.weight), _lib.ptr(model.position_embedding.weight),
                                  _lib.ptr(model.timestep_embedding.weight), _lib.ptr(x), M, L, d, V,
                                  model.max_position_embeddings, model.num_diffusion_steps + 1, _dp(drop), 0, _s()),
                "bd_embed_rows")
        # attention outputs of all layers in one allocation: the filler rows of the packed layout belong to no
        # sequence, the attention kernels never write them, and the weight-gradient GEMMs must not meet NaN bits there
        n_l = len(model.layers)
        att_all = torch.empty(n_l, M, aw, dtype=bf, device=dev)
        if pk is not None and pk.n_rows > pk.n_real and n_l:
            att_all[:, pk.n_real:].zero_()
        saved = []
        for li, (blk, w) in enumerate(zip(model.layers, tw.layers)):
            s = {"x_in": x}
            s["st1"] = torch.empty(M, 2, device=dev)
            s["h1"] = torch.empty(M, d, dtype=bf, device=dev)
            _ln(x, blk.norm1.weight, blk.norm1.bias, s["h1"], s["st1"], M, d, eps)
            s["qkv"] = torch.empty(M, 3 * aw, dtype=bf, device=dev)
            _gemm(s["h1"], w["wqkv"], w["bqkv"], s["qkv"], M, 3 * aw, 3 * aw, d, 3 * aw, 0)
            s["att"] = att_all[li]
            s["lse"] = torch.empty(B, H, L, device=dev)
            if pk is None:
                _ck(lib.bd_attention_fwd_drop(_lib.ptr(s["qkv"]), _lib.ptr(keymask), _lib.ptr(s["att"]), _lib.ptr(s["lse"]),
                                              B, L, H, aw // H, _dp(drop), 4 * li + 1, _s()), "bd_attention_fwd")
            else:
                _ck(lib.bd_attention_fwd_varlen(_lib.ptr(s["qkv"]), _lib.ptr(keymask), _lib.ptr(s["att"]),
                                                _lib.ptr(s["lse"]), _lib.ptr(pk.cu), B, L, H, aw // H, _dp(drop),
                                                4 * li + 1, _s()), "bd_attention_fwd_varlen")
            x_mid = torch.empty(M, d, device=dev)
            _gemm(s["att"], w["wo"], blk.attn.out_proj.bias, x_mid, M, d, d, aw, d, 2, aux_in=x)
            s["x_mid"] = x_mid
            s["st2"] = torch.empty(M, 2, device=dev)
            s["h2"] = torch.empty(M, d, dtype=bf, device=dev)
            _ln(x_mid, blk.norm2.weight, blk.norm2.bias, s["h2"], s["st2"], M, d, eps)
            s["u"] = torch.empty(M, F, dtype=bf, device=dev)
            s["a"] = torch.empty(M, F, dtype=bf, device=dev)
            # (s["u"] receives gelu'(pre-activation) times the dropout factor: the backward pass's multiplier)
            _gemm(s["h2"], w["w1"], blk.ffn.fc1.bias, s["a"], M, F, F, d, F, 6, aux_out=s["u"], drop=drop,
                  site=4 * li + 2)
            x_out = torch.empty(M, d, device=dev)
            _gemm(s["a"], w["w2"], blk.ffn.fc2.bias, x_out, M, d, d, F, d, 2, aux_in=x_mid, drop=drop,
                  site=4 * li + 3)
            saved.append(s)
            x = x_out
        stf = torch.empty(M, 2, device=dev)
        ctx.model, ctx.saved, ctx.tw, ctx.drop, ctx.mode, ctx.pk = model, saved, tw, drop, int(mode), pk
        ctx.ids, ctx.timesteps, ctx.keymask = ids, timesteps, keymask
        if mode == 1:
            hid = torch.empty(B, L, d, device=dev) if pk is None else torch.empty(M, d, device=dev)
            _ck(lib.bd_layernorm(_lib.ptr(x), _lib.ptr(model.final_norm.weight), _lib.ptr(model.final_norm.bias), None,
                                 _lib.ptr(hid), _lib.ptr(stf), M, d, eps, _s()), "bd_layernorm")
            ctx.x_last, ctx.stf, ctx.hf = x, stf, None
            return hid
        hf = torch.empty(M, d, dtype=bf, device=dev)
        _ln(x, model.final_norm.weight, model.final_norm.bias, hf, stf, M, d, eps)
        logits = torch.empty(B, L, V, device=dev) if pk is None else torch.empty(M, V, device=dev)
        _gemm(hf, tw.wout, None, logits, M, tw.vpad, V, d, V, 3)
        ctx.x_last, ctx.stf, ctx.hf = x, stf, hf
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        _enter()
        lib = _lib.lib()
        model, tw, drop, pk = ctx.model, ctx.tw, ctx.drop, ctx.pk
        ids, timesteps, keymask = ctx.ids, ctx.timesteps, ctx.keymask
        B, L = ids.shape
        M = B * L if pk is None else pk.n_rows
        d, F, V, H = model.hidden_size, model.ffn_hidden_size, model.vocab_size, model.num_heads
        dev = ids.device
        bf = torch.bfloat16
        # Parameter gradients. The kernels ACCUMULATE (atomic adds), so when every parameter already
        # has a dense fp32 .grad (FlatAdamW binds them to one flat buffer) they add straight into it
        # and autograd gets None for the parameters: no per-parameter zero fill, no AccumulateGrad
        # add. Otherwise fresh zero buffers are returned to autograd the usual way.
        params = list(model._params())
        direct = all(p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous()
                     and p.grad.device == dev for p in params if p.requires_grad)
        grads = {}

        def gbuf(p):
            if direct and p.requires_grad:
                return p.grad
            g = torch.zeros_like(p, dtype=torch.float32)
            grads[p] = g
            return g

        def zeros(*shape):
            return torch.zeros(*shape, device=dev)

        dh_buf = torch.empty(M, d, dtype=bf, device=dev)
        head_job = None
        if ctx.mode == 1:
            # the incoming gradient is that of the final-norm output itself
            _to_bf16(dlogits.reshape(M, d).float().contiguous(), dh_buf)
        else:
            # vocabulary head (rows padded to a multiple of 128: own buffer, then one add / slice)
            dl = torch.zeros(M, tw.vpad, dtype=bf, device=dev)
            dl[:, :V] = dlogits.reshape(M, V)
            dwout = zeros(tw.vpad, d)
            head_job = (dl, ctx.hf, dwout, M, tw.vpad, d, None)      # launched with the weight-gradient GEMMs below
            if not (direct and model.output_proj.weight.requires_grad):
                grads[model.output_proj.weight] = dwout[:V]
            _gemm(dl, tw.woutT, None, dh_buf, M, d, d, tw.vpad, d, 0)
        dx = torch.empty(M, d, device=dev)
        n_layers = len(model.layers)
        # the chain stops below the lowest block with a trainable parameter (a property head fine-tuning the last
        # blocks leaves the rest frozen); the embeddings need it all the way down
        need_embed = any(p.requires_grad for e in (model.token_embedding, model.position_embedding,
                                                   model.timestep_embedding) for p in e.parameters())
        stop_layer = 0 if need_embed else next(
            (i for i, blk in enumerate(model.layers) if any(p.requires_grad for p in blk.parameters())), n_layers)
        aw, dh, dhp, padded = tw.aw, tw.dh, tw.dhp, tw.dhp != tw.dh

        # The weight-gradient GEMMs feed nothing further down the chain, so they run on a second stream beside
        # the data-gradient chain (their atomic drains and tails overlap with its kernels). Stream-to-stream
        # waits are the expensive host calls here (about 27 us each), so a layer's four weight-gradient launches
        # are handed over together at the end of the layer: one wait of the second stream on the main one. The
        # gradient buffers they read rotate (two sets, three for the feed-forward branch gradient that a layer
        # writes for the next one), and a layer begins by waiting for the hand-over of two layers before: one wait
        # of the main stream. The last layer hands over launch by launch so that no tail is left behind it.
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        if padded:
            tw.pad_grads.zero_()
        main_h = _STREAM[0]
        side_h = C.c_void_p(side.cuda_stream) if side is not None else None
        nb = 2 if side is not None else 1
        dyb = [torch.empty(M, d, dtype=bf, device=dev) for _ in range(3 if side is not None else 1)]   # feed-forward branch
        dy2b = [torch.empty(M, d, dtype=bf, device=dev) for _ in range(nb)]                            # attention branch
        dub = [torch.empty(M, F, dtype=bf, device=dev) for _ in range(nb)]
        dqkvb = [torch.empty(M, 3 * aw, dtype=bf, device=dev) for _ in range(nb)]
        if pk is not None and pk.n_rows > pk.n_real:
            for t_ in dqkvb:                  # filler rows: never written by the attention backward, read by the wgrad
                t_[pk.n_real:].zero_()
        pending, handed = [], {}
        unpadded_from = [n_layers]            # layers from here up were un-padded early, for the tail hook

        def hand_over(li):
            if not pending:
                return
            ready = torch.cuda.Event()
            ready.record(main)
            side.wait_event(ready)
            _STREAM[0] = side_h              # the launch helpers take their stream from here
            try:
                _wgrad_group([tuple(job[:6]) + ((job[6] if len(job) > 6 else None),) for job in pending])
            finally:
                _STREAM[0] = main_h
            pending.clear()
            if head_add:          # the vocabulary head's padded gradient joins the parameter's behind its GEMM
                with torch.cuda.stream(side):
                    head_add.pop()()
            done = torch.cuda.Event()
            done.record(side)
            handed[li] = done

        hook_fn, hook_layer = _tail_hook_of(model)

        def tail_hook(li):
            """Every gradient of the layers >= li is enqueued (main stream: this layer's kernels; second stream: the
            weight-gradient GEMMs handed over so far): un-pad the upper layers' attention gradients, then call the hook
            on a stream whose order covers both."""
            if side is not None:
                ready = torch.cuda.Event()
                ready.record(main)
                side.wait_event(ready)
                _STREAM[0] = side_h
            try:
                if padded:
                    tw.unpad_gradients(gbuf, first_layer=li)
                    unpadded_from[0] = li
                lnp.finish()             # the LayerNorm gamma / beta sums of the layers >= li
            finally:
                _STREAM[0] = main_h
            with torch.cuda.stream(side if side is not None else main):
                hook_fn()

        def wgrad_side(li, dY, X, dW, N, K, db):
            if side is None:
                return _wgrad(dY, X, dW, M, N, K, db=db)
            pending.append((dY, X, dW, M, N, K, db))
            if li == stop_layer:
                hand_over(li)

        head_add = []
        if head_job is not None:
            direct_head = direct and model.output_proj.weight.requires_grad
            if side is None:
                _wgrad(*head_job[:6])
                if direct_head:
                    model.output_proj.weight.grad.add_(head_job[2][:V])
            else:
                pending.append(head_job)
                if direct_head:
                    head_add.append(lambda: model.output_proj.weight.grad.add_(head_job[2][:V]))

        # every LayerNorm backward also emits the bf16 gradient the next branch reads (with the fc2-output
        # dropout of that branch folded in), so dx is not read again just to be cast
        top = (n_layers - 1) % len(dyb) if n_layers else 0
        lnp = tw.ln_parts(d, dev)
        lnp.begin()
        _ln_bwd(dh_buf, ctx.x_last, ctx.stf, model.final_norm.weight, None, dx, gbuf(model.final_norm.weight),
                gbuf(model.final_norm.bias), M, d, dx_bf16=dyb[top] if n_layers else None, drop=drop,
                site=max(0, 4 * (n_layers - 1) + 3), parts=lnp)

        datt = torch.empty(M, aw, dtype=bf, device=dev)
        delta = torch.empty(B, H, L, device=dev)
        embed_drop_done = False
        for li, blk, w, s in zip(reversed(range(len(model.layers))), reversed(model.layers), reversed(tw.layers),
                                 reversed(ctx.saved)):
            if li < stop_layer:
                break
            a = blk.attn
            ev = handed.pop(li + 2, None)
            if ev is not None:
                main.wait_event(ev)      # this layer's buffer set (and the buffer it writes for the next) are free
            dy, dy_next = dyb[li % len(dyb)], dyb[(li - 1) % len(dyb)]
            dy2, du, dqkv = dy2b[li % nb], dub[li % nb], dqkvb[li % nb]
            # x_out = x_mid + drop(fc2(drop(gelu(fc1(LN2(x_mid))))))
            wgrad_side(li, dy, s["a"], gbuf(blk.ffn.fc2.weight), d, F, gbuf(blk.ffn.fc2.bias))
            _gemm(dy, w["w2T"], None, du, M, F, F, d, F, 7, aux_in=s["u"])
            wgrad_side(li, du, s["h2"], gbuf(blk.ffn.fc1.weight), F, d, gbuf(blk.ffn.fc1.bias))
            _gemm(du, w["w1T"], None, dh_buf, M, d, d, F, d, 0)
            _ln_bwd(dh_buf, s["x_mid"], s["st2"], blk.norm2.weight, dx, dx, gbuf(blk.norm2.weight), gbuf(blk.norm2.bias), M, d,
                    dx_bf16=dy2, parts=lnp)
            # x_mid = x_in + out_proj(attention(qkv(LN1(x_in))))
            # (zero-padded heads: the padded copies' gradients go to tw.pad_grads and are added to the parameters' at the end)
            wgrad_side(li, dy2, s["att"], tw.pad_views[li]["o"] if padded else gbuf(a.out_proj.weight), d, aw,
                       gbuf(a.out_proj.bias))
            _gemm(dy2, w["woT"], None, datt, M, aw, aw, d, aw, 0)
            if pk is None:
                _ck(lib.bd_attention_bwd_drop(_lib.ptr(s["qkv"]), _lib.ptr(keymask), _lib.ptr(s["att"]), _lib.ptr(datt),
                                              _lib.ptr(s["lse"]), _lib.ptr(delta), _lib.ptr(dqkv), B, L, H, aw // H,
                                              _dp(drop), 4 * li + 1, _s()), "bd_attention_bwd")
            else:
                _ck(lib.bd_attention_bwd_varlen(_lib.ptr(s["qkv"]), _lib.ptr(keymask), _lib.ptr(s["att"]), _lib.ptr(datt),
                                                _lib.ptr(s["lse"]), _lib.ptr(delta), _lib.ptr(dqkv), _lib.ptr(pk.cu), B, L,
                                                H, aw // H, _dp(drop), 4 * li + 1, _s()), "bd_attention_bwd_varlen")
            projs = (a.q_proj, a.k_proj, a.v_proj)
            if not padded:      # the three row segments of the stacked gradient, one launch
                wgrad_side(li, dqkv, s["h1"], [gbuf(pr.weight) for pr in projs], 3 * aw, d, [gbuf(pr.bias) for pr in projs])
            else:   # rows of the padded projections (the q rows were scaled by qfix in the forward: undone when un-padding)
                wgrad_side(li, dqkv, s["h1"], tw.pad_views[li]["w"], 3 * aw, d, tw.pad_views[li]["b"])
            _gemm(dqkv, w["wqkvT"], None, dh_buf, M, d, d, 3 * aw, d, 0)
            # (the lowest layer's dx is the gradient of the embedding dropout's output: dropped in the same pass)
            embed_drop_done = li == 0 and need_embed and drop is not None and d % 8 == 0
            _ln_bwd(dh_buf, s["x_in"], s["st1"], blk.norm1.weight, dx, dx, gbuf(blk.norm1.weight), gbuf(blk.norm1.bias), M, d,
                    dx_bf16=dy_next if li > stop_layer else None, drop=drop, site=max(0, 4 * (li - 1) + 3),
                    drop_dx=embed_drop_done, parts=lnp)
            if side is not None:
                hand_over(li)
            if direct and hook_fn is not None and li == hook_layer:
                tail_hook(li)
        if need_embed and drop is not None and not embed_drop_done:
            _ck(lib.bd_dropout_f32(_lib.ptr(dx), _lib.ptr(dx), M, d, _dp(drop), 0, _s()), "bd_dropout_f32")
        if need_embed and pk is None:
            _ck(lib.bd_embed_bwd(_lib.ptr(dx), _lib.ptr(ids), _lib.ptr(timesteps), _lib.ptr(gbuf(model.token_embedding.weight)),
                                 _lib.ptr(gbuf(model.position_embedding.weight)),
                                 _lib.ptr(gbuf(model.timestep_embedding.weight)), B, L, d, int(model.pad_token_id), V,
                                 model.max_position_embeddings, model.num_diffusion_steps + 1, _s()),
                "bd_embed_bwd")
        elif need_embed:
            _ck(lib.bd_embed_bwd_varlen(_lib.ptr(dx), _lib.ptr(ids), _lib.ptr(timesteps),
                                        _lib.ptr(gbuf(model.token_embedding.weight)),
                                        _lib.ptr(gbuf(model.position_embedding.weight)),
                                        _lib.ptr(gbuf(model.timestep_embedding.weight)), _lib.ptr(pk.cu), B, L, d,
                                        int(model.pad_token_id), V, model.max_position_embeddings,
                                        model.num_diffusion_steps + 1, _s()), "bd_embed_bwd_varlen")
        if side is not None:
            hand_over(-1)                # whatever no layer handed over (a model without trainable blocks)
            main.wait_stream(side)       # every gradient is complete before autograd / the optimizer go on
        if padded:
            tw.unpad_gradients(gbuf, end_layer=unpadded_from[0])
        lnp.finish(last=True)
        out = [None, None, None, None, None, None]
        for p in params:
            g = grads.get(p)
            out.append(g if (g is not None and p.requires_grad) else None)
        ctx.saved = None
        return tuple(out)


def backbone_forward_train(model, input_ids, timesteps, attention_mask, hidden_states=False, packing=None):
    """Training forward: logits, or (``hidden_states=True``) the fp32 final-norm hidden states. With a
    ``RowPacking`` the result holds the packed rows ([n_rows, V] or [n_rows, d]) instead of [B, L, .]."""
    _lib.require_device(input_ids, "DiffusionBackbone.forward")
    if model.hidden_size % 128 or model.ffn_hidden_size % 128:
        raise ValueError("bidiff training path needs hidden and ffn sizes that are multiples of 128")
    timesteps = model._check_inputs(input_ids, timesteps)
    ids = input_ids.to(torch.long).contiguous()
    keymask = model._keymask_bytes(attention_mask, ids)
    for p in model._params():
        if p.dtype != torch.float32 or not p.is_contiguous():
            raise RuntimeError("bidiff training path expects contiguous fp32 parameters")
    if packing is not None and (packing.B, packing.L) != tuple(ids.shape):
        raise ValueError("RowPacking was built for another batch shape")
    return _BackboneFn.apply(model, ids, timesteps, keymask, 1 if hidden_states else 0, packing, *model._params())


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, indicator, attention_mask):
        _enter()
        V = logits.shape[-1]
        lg = logits.contiguous().view(-1, V)
        M = lg.shape[0]
        dev = lg.device
        tg = targets.to(torch.long).contiguous().view(-1)
        ind = indicator.to(torch.uint8).contiguous().view(-1)
        am = None if attention_mask is None else attention_mask.to(torch.long).contiguous().view(-1)
        n_parts = 16         # the workgroups' sums go to pairs in cache lines of their own; the finish launch adds them up
        sums = torch.zeros(32 * (1 + n_parts), device=dev)
        lse = torch.empty(M, device=dev)
        _ck(_lib.lib().bd_masked_ce_fwd_parts(_lib.ptr(lg), _lib.ptr(tg), _lib.ptr(ind), _lib.ptr(am), _lib.ptr(lse),
                                              _lib.ptr(sums), n_parts, M, V, _s()), "bd_masked_ce_fwd")
        ctx.save_for_backward(lg, tg, ind, lse, sums)
        ctx.am = am
        ctx.shape = logits.shape
        loss = torch.empty((), device=dev)
        _ck(_lib.lib().bd_masked_ce_finish(_lib.ptr(sums), _lib.ptr(loss), n_parts, _s()), "bd_masked_ce_finish")
        return loss

    @staticmethod
    def backward(ctx, gout):
        _enter()
        lg, tg, ind, lse, sums = ctx.saved_tensors
        M, V = lg.shape
        g = gout.reshape(1).float().contiguous()
        dl = torch.empty_like(lg)
        _ck(_lib.lib().bd_masked_ce_bwd(_lib.ptr(lg), _lib.ptr(tg), _lib.ptr(ind), _lib.ptr(ctx.am), _lib.ptr(lse),
                                        _lib.ptr(sums), _lib.ptr(g), _lib.ptr(dl), M, V, _s()), "bd_masked_ce_bwd")
        return dl.view(ctx.shape), None, None, None


def masked_cross_entropy(logits, targets, mask_indicator, attention_mask=None):
    _lib.require_device(logits, "masked cross-entropy")
    if logits.dtype != torch.float32:
        logits = logits.float()
    return _MaskedCE.apply(logits, targets, mask_indicator, attention_mask)


def packed_training_loss(diffusion, noisy_ids, timesteps, attention_mask, targets, mask_indicator, lengths=None):
    """Denoiser + masked CE on the packed rows of the batch. Returns (loss, logits): the loss and its gradients
    equal the padded computation's (padded rows contribute to neither); ``logits()`` builds the [B, L, V] logits, a
    detached copy with zeros at the rows that were left out (a training loop never looks at them, and writing
    B * L * V floats per step costs as much as two of the step's GEMMs)."""
    backbone = diffusion.backbone
    B, L = noisy_ids.shape
    if lengths is None:
        lengths = RowPacking.lengths_from_mask(attention_mask)
    pk = RowPacking(lengths, B, L, noisy_ids.device)
    logits_p = backbone_forward_train(backbone, noisy_ids, timesteps, attention_mask, packing=pk)
    # targets and loss flags of the packed rows (attention mask folded into the flag, fillers flagged off): one launch
    tg = torch.empty(pk.n_rows, dtype=torch.long, device=noisy_ids.device)
    ind = torch.empty(pk.n_rows, dtype=torch.uint8, device=noisy_ids.device)
    ind_src = mask_indicator.contiguous()
    ind_src = ind_src.view(torch.uint8) if ind_src.dtype == torch.bool else (ind_src != 0).to(torch.uint8)
    tg_src = targets.to(torch.long).contiguous()          # (converted copies stay referenced until the launch is queued)
    am_src = attention_mask.to(torch.long).contiguous()
    _ck(_lib.lib().bd_pack_targets(_lib.ptr(tg_src), _lib.ptr(ind_src), _lib.ptr(am_src), _lib.ptr(pk.rowmap), pk.n_rows,
                                   _lib.ptr(tg), _lib.ptr(ind), _s()), "bd_pack_targets")
    loss = masked_cross_entropy(logits_p, tg, ind, None)

    def full_logits():
        """[B, L, V] copy of the packed logits, zeros at the rows that were left out (built when somebody asks)."""
        with torch.no_grad():
            idx, _ = pk.gather_index()
            full = torch.zeros(B * L, logits_p.shape[-1], device=logits_p.device, dtype=logits_p.dtype)
            full.index_copy_(0, idx[:pk.n_real], logits_p.detach()[:pk.n_real])
        return full.view(B, L, -1)
    return loss, full_logits
