"""Fused AdamW over one flat fp32 parameter buffer (bd_adamw_step) with the global-norm
gradient clip folded in (bd_sumsq_f32), and the whole-model gradient bucket used for the
data-parallel all-reduce.

Replaces, for the default AdamW branch, ``build_step1_backbone_optimizer``
(shared/optimizer_setup.py:280-317) + ``clip_grad_norm_`` + ``optimizer.step()``
(trainer_backbone.py:441-455). It is a ``torch.optim.Optimizer`` so the reference's LR
schedulers and ``state_dict`` format (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``)
keep working; the per-parameter tensors are views into the flat buffers.
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

    def zero_grad(self, set_to_none: bool = False):
        self.flat_grad.zero_()
        off = 0
        for p in self._params:       # keep .grad bound to the flat buffer
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + k].view(p.shape)
            off += k

    def grad_norm(self):
        """L2 norm of all gradients (device tensor, no host sync)."""
        self._sumsq.zero_()
        _lib.check(_lib.lib().bd_sumsq_f32(_lib.ptr(self.flat_grad), self.numel, _lib.ptr(self._sumsq),
                                           _lib.stream_handle()), "bd_sumsq_f32")
        return self._sumsq.sqrt()

    @torch.no_grad()
    def step(self, closure=None, max_norm: float = 0.0):
        """One AdamW update; ``max_norm > 0`` first clips the global gradient norm to it."""
        g = self.param_groups[0]
        self._step += 1
        b1, b2 = g["betas"]
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
        for p in self._params:
            self.state[p]["step"] = torch.tensor(float(self._step))
        _lib.bump_weights_generation()   # parameters changed behind torch's back: refresh bf16 copies
        return None

    def load_state_dict(self, state_dict):
        sd = state_dict["state"]
        for i, p in enumerate(self._params):
            st = sd.get(i)
            if st is None:
                continue
            self.state[p]["exp_avg"].copy_(st["exp_avg"])
            self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
            self._step = int(float(st["step"]))
        for g_new, g_old in zip(self.param_groups, state_dict["param_groups"]):
            for k in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
                if k in g_old:
                    g_new[k] = g_old[k]
