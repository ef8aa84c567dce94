"""Discrete masking diffusion: noise schedule, q(x_t|x_0) corruption, masked CE loss.

Drop-in for the reference's ``src/model/diffusion.py`` (``DiscreteMaskingDiffusion``, :10-273):
same constructor, buffer name (``mask_schedule``), method names and the same keys in the
dict returned by ``forward``. The corruption runs in ``bd_corrupt`` and the loss in the
masked cross-entropy kernels of ``libbidiff_hip.so``.
"""
import ctypes as C
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .. import _lib


class _Outputs(dict):
    """The forward pass's result dict with entries that are built on first access (``deferred``: key -> function).
    Every way of looking at the dict sees them: indexing, ``get``, ``in``, iteration, ``keys`` / ``items`` / ``values``."""

    def __init__(self, eager, deferred):
        super().__init__(eager)
        self._deferred = dict(deferred)

    def _build_all(self):
        for key in list(self._deferred):
            self[key] = self._deferred.pop(key)()

    def __missing__(self, key):
        if key in self._deferred:
            self[key] = self._deferred.pop(key)()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._deferred

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __iter__(self):
        self._build_all()
        return dict.__iter__(self)

    def __len__(self):
        return dict.__len__(self) + len(self._deferred)

    def keys(self):
        self._build_all()
        return dict.keys(self)

    def items(self):
        self._build_all()
        return dict.items(self)

    def values(self):
        self._build_all()
        return dict.values(self)

    def __repr__(self):
        self._build_all()
        return dict.__repr__(self)

    def __eq__(self, other):
        self._build_all()
        return dict.__eq__(self, other)

    def __ne__(self, other):
        self._build_all()
        return dict.__ne__(self, other)

    __hash__ = None


class DiscreteMaskingDiffusion(nn.Module):
    def __init__(
        self,
        backbone: nn.Module,
        num_steps: int = 100,
        beta_min: float = 0.05,
        beta_max: float = 0.95,
        mask_token_id: int = 1,
        pad_token_id: int = 0,
        bos_token_id: Optional[int] = None,
        eos_token_id: Optional[int] = None,
        force_clean_t0: bool = False,
    ):
        super().__init__()
        self.backbone = backbone
        self.num_steps = num_steps
        self.beta_min = beta_min
        self.beta_max = beta_max
        self.mask_token_id = mask_token_id
        self.pad_token_id = pad_token_id
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.force_clean_t0 = force_clean_t0
        # "torch": uniforms from torch.rand (follows torch.manual_seed like the reference);
        # "philox": in-kernel counter-based stream seeded from torch's generator per call.
        self.corruption_rng = "torch"
        # training forward: leave out the rows behind each sequence's last attended position (no loss, no gradient
        # and no influence on other rows comes from them; the returned logits are zero there). False = every row.
        self.skip_padded_rows = True
        self.register_buffer("mask_schedule", self._compute_mask_schedule())

    def _compute_mask_schedule(self) -> torch.Tensor:
        """Linear schedule beta_min..beta_max over t = 0..T, evaluated in fp32 in this operand
        order (a float64 evaluation differs in the last bit for a quarter of the entries);
        entry 0 forced to 0 when ``force_clean_t0`` (reference diffusion.py:60-71)."""
        t = torch.arange(self.num_steps + 1, dtype=torch.float32)
        sched = self.beta_min + (self.beta_max - self.beta_min) * t / self.num_steps
        if self.force_clean_t0:
            sched[0] = 0.0
        return sched

    def get_mask_prob(self, t: torch.Tensor) -> torch.Tensor:
        return self.mask_schedule[t]

    # ------------------------------------------------------------------ q(x_t | x_0)
    def forward_process(self, input_ids: torch.Tensor, t: torch.Tensor,
                        attention_mask: Optional[torch.Tensor] = None,
                        uniforms: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Mask each token independently with probability mask_schedule[t[b]], never PAD
        (attention_mask == 1, or ids != pad without a mask), BOS or EOS
        (reference diffusion.py:84-127). ``uniforms`` [B,L] fp32 overrides the random draw."""
        _lib.require_device(input_ids, "forward_process")
        ids = input_ids.to(torch.long).contiguous()
        B, L = ids.shape
        dev = ids.device
        tt = t.to(device=dev, dtype=torch.long).contiguous()
        am = None if attention_mask is None else attention_mask.to(device=dev, dtype=torch.long).contiguous()
        seed = 0
        if uniforms is None and self.corruption_rng == "torch":
            uniforms = torch.rand(B, L, device=dev)
        elif uniforms is None:
            seed = int(torch.randint(0, 2**62, (1,)).item())
        if uniforms is not None:
            uniforms = uniforms.to(device=dev, dtype=torch.float32).contiguous()
        noisy = torch.empty_like(ids)
        ind = torch.empty(B, L, dtype=torch.bool, device=dev)
        sched = self.mask_schedule
        if sched.device != dev or sched.dtype != torch.float32:
            sched = sched.to(device=dev, dtype=torch.float32)
        rc = _lib.lib().bd_corrupt(
            _lib.ptr(ids), _lib.ptr(am), _lib.ptr(tt), _lib.ptr(sched), sched.numel(), _lib.ptr(uniforms),
            _lib.ptr(noisy), _lib.ptr(ind), B, L, int(self.mask_token_id), int(self.pad_token_id),
            -1 if self.bos_token_id is None else int(self.bos_token_id),
            -1 if self.eos_token_id is None else int(self.eos_token_id),
            seed, 0, _lib.stream_handle())
        _lib.check(rc, "bd_corrupt")
        return noisy, ind

    # ------------------------------------------------------------------ training forward
    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                timesteps: Optional[torch.Tensor] = None, lengths=None) -> Dict[str, torch.Tensor]:
        """t ~ U{1..T} (drawn before the corruption uniforms, like the reference) ->
        corruption -> denoiser -> masked CE (reference diffusion.py:129-167).

        ``lengths`` (optional, host integers: last attended position + 1 per sequence) spares the training path
        the device -> host copy it otherwise makes to lay out the packed rows (``skip_padded_rows``)."""
        B = input_ids.shape[0]
        dev = input_ids.device
        if timesteps is None:
            timesteps = torch.randint(1, self.num_steps + 1, (B,), device=dev)
        noisy_ids, mask_indicator = self.forward_process(input_ids, timesteps, attention_mask)
        bb = self.backbone
        if (self.skip_padded_rows and attention_mask is not None and torch.is_grad_enabled()
                and getattr(bb, "compute_dtype", "bf16") == "bf16" and any(p.requires_grad for p in bb.parameters())):
            from ..training.autograd import packed_training_loss
            loss, logits = packed_training_loss(self, noisy_ids, bb._check_inputs(noisy_ids, timesteps), attention_mask,
                                                input_ids, mask_indicator, lengths)
            return _Outputs({"loss": loss, "noisy_ids": noisy_ids, "mask_indicator": mask_indicator,
                             "timesteps": timesteps}, {"logits": logits})
        logits = self.backbone(noisy_ids, timesteps, attention_mask)
        loss = self._compute_loss(logits, input_ids, mask_indicator, attention_mask)
        return {
            "loss": loss,
            "logits": logits,
            "noisy_ids": noisy_ids,
            "mask_indicator": mask_indicator,
            "timesteps": timesteps,
        }

    def _compute_loss(self, logits, targets, mask_indicator, attention_mask=None) -> torch.Tensor:
        """Mean cross-entropy over positions that were masked and are not padding; an exact
        zero (still attached to the graph) when there is none (reference diffusion.py:169-209)."""
        from ..training.autograd import masked_cross_entropy
        return masked_cross_entropy(logits, targets, mask_indicator, attention_mask)

    def sample_step(self, noisy_ids, t: int, attention_mask=None, temperature: float = 1.0):
        """Unconstrained one-step resampling of every MASK position (reference :211-250)."""
        from ..sampling.core import plain_sample_step
        return plain_sample_step(self, noisy_ids, t, attention_mask, temperature)

    def get_loss_at_timestep(self, input_ids, timestep: int, attention_mask=None) -> torch.Tensor:
        B = input_ids.shape[0]
        ts = torch.full((B,), timestep, device=input_ids.device, dtype=torch.long)
        return self.forward(input_ids, attention_mask, ts)["loss"]
