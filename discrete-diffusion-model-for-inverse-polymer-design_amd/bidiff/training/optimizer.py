"""Fused AdamW over one flat fp32 parameter buffer (bd_adamw_step) with the global-norm
gradient clip folded in (bd_sumsq_f32), and the whole-model gradient bucket used for the
data-parallel all-reduce.

Replaces, for the default AdamW branch, ``build_step1_backbone_optimizer``
(shared/optimizer_setup.py:280-317) + ``clip_grad_norm_`` + ``optimizer.step()``
(trainer_backbone.py:441-455). It is a ``torch.optim.Optimizer`` so the reference's LR
schedulers and ``state_dict`` format (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``)
keep working; the per-parameter tensors are views into the flat buffers.

``skip_nonfinite`` (set by the trainer when AMP is on) gives ``step`` the behaviour of
``GradScaler.unscale_`` + ``clip_grad_norm_`` + ``GradScaler.step`` (trainer_backbone.py:441-453): a step whose
gradients hold an inf or NaN changes nothing and does not count for the bias corrections. The decision stays on
the device (the reference's ``scaler.step`` synchronises the host every step to read ``found_inf``); the counts of
applied and skipped steps come back only when ``state_dict`` / ``skipped_steps`` ask for them.
"""
import math

import torch

from .. import _lib


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FlatAdamW: no trainable parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW: parameters must live on a ROCm device (no CPU path)")
        n = sum(p.numel() for p in params)
        self.numel = n
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._step = 0
        self.skip_nonfinite = False
        self._guard = torch.zeros(2, dtype=torch.int32, device=dev)     # applied, skipped (skip_nonfinite steps)
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + k].view(p.shape)
                p.grad = self.flat_grad[off:off + k].view(p.shape)
                self.state[p] = {"step": torch.tensor(0.0), "exp_avg": self.flat_m[off:off + k].view(p.shape),
                                 "exp_avg_sq": self.flat_v[off:off + k].view(p.shape)}
                off += k
        self._params = params
        self._offsets = []
        off = 0
        for p in params:
            self._offsets.append(off)
            off += p.numel()

    def zero_grad(self, set_to_none: bool = False):
        self.flat_grad.zero_()
        base = self.flat_grad.data_ptr()
        for p, off in zip(self._params, self._offsets):       # keep .grad bound to the flat buffer
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)

    def sync_grads(self):
        """Make the flat gradient buffer hold every parameter's gradient before it is reduced or consumed.
        After ``model.zero_grad()`` (torch's default sets ``.grad`` to None) or ``p.grad = None`` the backward
        pass hands gradients to autograd, which creates fresh ``.grad`` tensors beside the flat buffer: those are
        copied into their slots and re-bound. A parameter without any gradient raises (the fused update always
        covers the whole buffer)."""
        base = self.flat_grad.data_ptr()
        for p, off in zip(self._params, self._offsets):
            g = p.grad
            if g is not None and g.data_ptr() == base + 4 * off:
                continue
            if g is None:
                raise RuntimeError("flat optimizer: a parameter has no gradient (was .grad set to None after the "
                                   "backward pass?); use optimizer.zero_grad() between steps")
            k = p.numel()
            slot = self.flat_grad[off:off + k].view(p.shape)
            slot.copy_(g)
            p.grad = slot

    def grad_norm(self):
        """L2 norm of all gradients (device tensor, no host sync)."""
        self._sumsq.zero_()
        _lib.check(_lib.lib().bd_sumsq_f32(_lib.ptr(self.flat_grad), self.numel, _lib.ptr(self._sumsq),
                                           _lib.stream_handle()), "bd_sumsq_f32")
        return self._sumsq.sqrt()

    @torch.no_grad()
    def step(self, closure=None, max_norm: float = 0.0):
        """One AdamW update; ``max_norm > 0`` first clips the global gradient norm to it."""
        self.sync_grads()
        g = self.param_groups[0]
        self._step += 1
        b1, b2 = g["betas"]
        if self.skip_nonfinite:
            return self._guarded_step(g, b1, b2, max_norm)
        bc1 = 1.0 - b1 ** self._step
        bc2_sqrt = math.sqrt(1.0 - b2 ** self._step)
        ss = None
        if max_norm and max_norm > 0:
            self._sumsq.zero_()
            _lib.check(_lib.lib().bd_sumsq_f32(_lib.ptr(self.flat_grad), self.numel, _lib.ptr(self._sumsq),
                                               _lib.stream_handle()), "bd_sumsq_f32")
            ss = self._sumsq
        rc = _lib.lib().bd_adamw_step(_lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), _lib.ptr(self.flat_m),
                                      _lib.ptr(self.flat_v), self.numel, float(g["lr"]), float(b1), float(b2),
                                      float(g["eps"]), float(g["weight_decay"]), float(bc1), float(bc2_sqrt),
                                      _lib.ptr(ss), float(max_norm or 0.0), _lib.stream_handle())
        _lib.check(rc, "bd_adamw_step")
        step_t = torch.tensor(float(self._step))
        for p in self._params:
            self.state[p]["step"] = step_t
        _lib.bump_weights_generation()   # parameters changed behind torch's back: refresh bf16 copies
        return None

    def _guarded_step(self, g, b1, b2, max_norm):
        lib, s = _lib.lib(), _lib.stream_handle()
        self._sumsq.zero_()
        _lib.check(lib.bd_sumsq_f32(_lib.ptr(self.flat_grad), self.numel, _lib.ptr(self._sumsq), s), "bd_sumsq_f32")
        _lib.check(lib.bd_adamw_step_guarded(_lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), _lib.ptr(self.flat_m),
                                             _lib.ptr(self.flat_v), self.numel, float(g["lr"]), float(b1), float(b2),
                                             float(g["eps"]), float(g["weight_decay"]), _lib.ptr(self._guard),
                                             _lib.ptr(self._sumsq), float(max_norm or 0.0), s), "bd_adamw_step_guarded")
        _lib.check(lib.bd_step_guard_update(_lib.ptr(self._guard), _lib.ptr(self._sumsq), s), "bd_step_guard_update")
        self._steps_stale = True
        _lib.bump_weights_generation()
        return None

    def _refresh_steps(self):
        """Per-parameter ``step`` entries from the device's count of applied steps (one host read)."""
        if self.skip_nonfinite and getattr(self, "_steps_stale", False):
            self._step = int(self._guard[0].item())
            step_t = torch.tensor(float(self._step))
            for p in self._params:
                self.state[p]["step"] = step_t
            self._steps_stale = False

    def skipped_steps(self) -> int:
        """Steps left out because their gradients were not finite (synchronises)."""
        return int(self._guard[1].item())

    def state_dict(self):
        self._refresh_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        sd = state_dict["state"]
        for i, p in enumerate(self._params):
            st = sd.get(i)
            if st is None:
                continue
            self.state[p]["exp_avg"].copy_(st["exp_avg"])
            self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
            self._step = int(float(st["step"]))
            self.state[p]["step"] = torch.tensor(float(self._step))
        self._guard[0] = self._step
        for g_new, g_old in zip(self.param_groups, state_dict["param_groups"]):
            for k in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
                if k in g_old:
                    g_new[k] = g_old[k]


class FlatMuonAdamW(torch.optim.Optimizer):
    """Mixed Muon / AdamW optimizer over one flat fp32 buffer: drop-in for ``MuonAdamW(param_groups)``
    (shared/optimizer_setup.py:70-169). ``params`` is the reference's list of group dicts with
    ``kind`` in {"adamw", "muon"}. Flat layout: all AdamW parameters first (one fused bd_adamw_step
    over that range), then the Muon matrices grouped by shape, each shape class back to back so that
    one bd_muon_step (momentum, Newton-Schulz on batched bf16 MFMA GEMMs, update) handles the class.
    The flat gradient buffer is the data-parallel all-reduce bucket, as in FlatAdamW; ``state_dict``
    keeps the reference's per-parameter keys (``step`` / ``exp_avg`` / ``exp_avg_sq`` or
    ``momentum_buffer``), the tensors being views into the flat buffers."""

    def __init__(self, params):
        defaults = dict(lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0, kind="adamw", momentum=0.95,
                        nesterov=True, backend_steps=5)
        super().__init__(params, defaults)
        adamw, muon = [], []
        for g in self.param_groups:
            kind = g.get("kind", "adamw")
            if kind not in {"adamw", "muon"}:
                raise ValueError(f"Unsupported optimizer group kind: {kind}")
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                if kind == "muon":
                    if p.ndim != 2:
                        raise RuntimeError(f"Muon path only supports 2D tensors, got shape {tuple(p.shape)}")
                    if p.shape[0] % 128 or p.shape[1] % 128:
                        raise NotImplementedError(f"bidiff Muon kernels need matrix sizes that are multiples of 128, "
                                                  f"got {tuple(p.shape)}")
                    muon.append((p, g))
                else:
                    adamw.append((p, g))
        if not adamw and not muon:
            raise ValueError("FlatMuonAdamW: no trainable parameters")
        if len({id(g) for _, g in adamw}) > 1 or len({id(g) for _, g in muon}) > 1:
            raise NotImplementedError("FlatMuonAdamW: one adamw group and one muon group, as the reference builds them")
        self._adamw_group = adamw[0][1] if adamw else None
        self._muon_group = muon[0][1] if muon else None
        classes = {}
        for p, _ in muon:
            classes.setdefault(tuple(p.shape), []).append(p)
        order = [p for p, _ in adamw] + [p for ps in classes.values() for p in ps]
        dev = order[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatMuonAdamW: parameters must live on a ROCm device (no CPU path)")
        self.n_adamw = sum(p.numel() for p, _ in adamw)
        self.numel = n = sum(p.numel() for p in order)
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)      # exp_avg | momentum buffers
        self.flat_v = torch.zeros(max(self.n_adamw, 1), dtype=torch.float32, device=dev)
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._step = 0
        self.skip_nonfinite = False
        self._guard = torch.zeros(2, dtype=torch.int32, device=dev)     # applied, skipped (skip_nonfinite steps)
        self._params, self._offsets = order, []
        off = 0
        with torch.no_grad():
            for i, p in enumerate(order):
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + k].view(p.shape)
                p.grad = self.flat_grad[off:off + k].view(p.shape)
                if off < self.n_adamw:
                    self.state[p] = {"step": 0, "exp_avg": self.flat_m[off:off + k].view(p.shape),
                                     "exp_avg_sq": self.flat_v[off:off + k].view(p.shape)}
                else:
                    self.state[p] = {"momentum_buffer": self.flat_m[off:off + k].view(p.shape)}
                self._offsets.append(off)
                off += k
        self._classes = []          # (flat offset, count, rows, cols)
        off = self.n_adamw
        ws = 0
        for (rows, cols), ps in classes.items():
            self._classes.append((off, len(ps), rows, cols))
            ws = max(ws, int(_lib.lib().bd_muon_workspace_bytes(len(ps), rows, cols)))
            off += len(ps) * rows * cols
        self._ws = torch.empty(max(ws, 1), dtype=torch.uint8, device=dev)

    def zero_grad(self, set_to_none: bool = False):
        self.flat_grad.zero_()
        for p, off in zip(self._params, self._offsets):       # keep .grad bound to the flat buffer
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)

    def sync_grads(self):
        """Make the flat gradient buffer hold every parameter's gradient before it is reduced or consumed.
        After ``model.zero_grad()`` (torch's default sets ``.grad`` to None) or ``p.grad = None`` the backward
        pass hands gradients to autograd, which creates fresh ``.grad`` tensors beside the flat buffer: those are
        copied into their slots and re-bound. A parameter without any gradient raises (the fused update always
        covers the whole buffer)."""
        base = self.flat_grad.data_ptr()
        for p, off in zip(self._params, self._offsets):
            g = p.grad
            if g is not None and g.data_ptr() == base + 4 * off:
                continue
            if g is None:
                raise RuntimeError("flat optimizer: a parameter has no gradient (was .grad set to None after the "
                                   "backward pass?); use optimizer.zero_grad() between steps")
            k = p.numel()
            slot = self.flat_grad[off:off + k].view(p.shape)
            slot.copy_(g)
            p.grad = slot

    def grad_norm(self):
        self._sumsq.zero_()
        _lib.check(_lib.lib().bd_sumsq_f32(_lib.ptr(self.flat_grad), self.numel, _lib.ptr(self._sumsq),
                                           _lib.stream_handle()), "bd_sumsq_f32")
        return self._sumsq.sqrt()

    @torch.no_grad()
    def step(self, closure=None, max_norm: float = 0.0):
        """One update of both groups; ``max_norm > 0`` first clips the global gradient norm (all
        parameters of both groups) to it."""
        self.sync_grads()
        lib, s = _lib.lib(), _lib.stream_handle()
        self._step += 1
        guarded = self.skip_nonfinite
        ss = None
        if guarded or (max_norm and max_norm > 0):
            self._sumsq.zero_()
            _lib.check(lib.bd_sumsq_f32(_lib.ptr(self.flat_grad), self.numel, _lib.ptr(self._sumsq), s), "bd_sumsq_f32")
            ss = self._sumsq
        if self.n_adamw:
            g = self._adamw_group
            b1, b2 = g.get("betas", (0.9, 0.95))
            if guarded:
                _lib.check(lib.bd_adamw_step_guarded(_lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), _lib.ptr(self.flat_m),
                                                     _lib.ptr(self.flat_v), self.n_adamw, float(g["lr"]), float(b1), float(b2),
                                                     float(g.get("eps", 1e-8)), float(g.get("weight_decay", 0.0)),
                                                     _lib.ptr(self._guard), _lib.ptr(ss), float(max_norm or 0.0), s),
                           "bd_adamw_step_guarded")
            else:
                bc1 = 1.0 - b1 ** self._step
                bc2_sqrt = math.sqrt(1.0 - b2 ** self._step)
                _lib.check(lib.bd_adamw_step(_lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), _lib.ptr(self.flat_m),
                                             _lib.ptr(self.flat_v), self.n_adamw, float(g["lr"]), float(b1), float(b2),
                                             float(g.get("eps", 1e-8)), float(g.get("weight_decay", 0.0)), float(bc1),
                                             float(bc2_sqrt), _lib.ptr(ss), float(max_norm or 0.0), s), "bd_adamw_step")
        if self._classes:
            g = self._muon_group
            muon_step = lib.bd_muon_step_guarded if guarded else lib.bd_muon_step
            for off, nb, rows, cols in self._classes:
                _lib.check(muon_step(_lib.ptr(self.flat_param[off:]), _lib.ptr(self.flat_grad[off:]),
                                            _lib.ptr(self.flat_m[off:]), _lib.ptr(self._ws), self._ws.numel(), nb, rows, cols,
                                            float(g["lr"]), float(g.get("weight_decay", 0.0)),
                                            float(g.get("momentum", 0.95)), int(bool(g.get("nesterov", True))),
                                            int(g.get("backend_steps", 5)), _lib.ptr(ss), float(max_norm or 0.0), s),
                           "bd_muon_step")
        if guarded:
            _lib.check(lib.bd_step_guard_update(_lib.ptr(self._guard), _lib.ptr(self._sumsq), s), "bd_step_guard_update")
            self._steps_stale = True
        else:
            for p in self._params:
                if "step" in self.state[p]:
                    self.state[p]["step"] = self._step
        _lib.bump_weights_generation()
        return None

    def _refresh_steps(self):
        if self.skip_nonfinite and getattr(self, "_steps_stale", False):
            self._step = int(self._guard[0].item())
            for p in self._params:
                if "step" in self.state[p]:
                    self.state[p]["step"] = self._step
            self._steps_stale = False

    def skipped_steps(self) -> int:
        """Steps left out because their gradients were not finite (synchronises)."""
        return int(self._guard[1].item())

    def state_dict(self):
        self._refresh_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        sd = state_dict["state"]
        idx = 0
        for g in self.param_groups:
            for p in g["params"]:
                st = sd.get(idx)
                idx += 1
                if st is None or p not in self.state:
                    continue
                for k in ("exp_avg", "exp_avg_sq", "momentum_buffer"):
                    if k in st and k in self.state[p]:
                        self.state[p][k].copy_(st[k])
                if "step" in st:
                    self._step = int(float(st["step"]))
                    self.state[p]["step"] = self._step
        self._guard[0] = self._step
        for g_new, g_old in zip(self.param_groups, state_dict["param_groups"]):
            for k in ("lr", "betas", "eps", "weight_decay", "initial_lr", "momentum", "nesterov", "backend_steps"):
                if k in g_old:
                    g_new[k] = g_old[k]
