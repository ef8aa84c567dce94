"""BackboneTrainer: the reference's training loop around the native training step.

Drop-in for ``src/training/trainer_backbone.py`` (``BackboneTrainer``, :61-715): same
constructor, ``train()`` / ``load_checkpoint()``, history dict, checkpoint dict layout
(``epoch, global_step, model_state_dict, optimizer_state_dict, scheduler_state_dict, val_loss,
best_val_loss, config``) and CSV files. Differences that follow from the native step:
  * the denoiser computes in bf16 on the MFMA path, which is the reference's step under ``use_amp: true`` (its
    configs' setting); ``use_amp: false`` (fp32 products, trainer_backbone.py:434) is refused with
    NotImplementedError rather than trained in narrower arithmetic. ``use_amp`` also keeps what ``GradScaler``
    contributes to the reference's step (trainer_backbone.py:129,439-453): a step whose gradients hold an inf or NaN is skipped
    and does not count for the bias corrections (``optimizer.skip_nonfinite``; decided on the device, without the
    host synchronisation ``scaler.step`` makes). The scale factor itself is dropped: multiplying the loss by a power
    of two and dividing the gradients by it again is exact in fp32;
  * AdamW + global-norm clipping are one fused kernel pair over a flat parameter buffer;
  * data parallelism = one process per GPU; the flat gradient buffer is all-reduced (mean) in
    a single RCCL call on optimizer steps only (accumulation micro-steps skip it, which is what
    ``no_sync`` does in the reference, trainer_backbone.py:427-431).
"""
import csv
import math
from pathlib import Path
from typing import Dict, Optional

import torch
import torch.distributed as dist
from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, SequentialLR

from .optimizer import FlatAdamW  # noqa: F401
from .optimizer_setup import build_step1_backbone_optimizer
import os

from . import autograd as _autograd
from ..parallel import GradientExchange, allreduce_mean_, tail_split_offset


def _f(v, name):
    try:
        return float(v)
    except (TypeError, ValueError) as e:
        raise ValueError(f"training_backbone.{name} must be a number, got {v!r}") from e


class BackboneTrainer:
    def __init__(self, model, train_dataloader, val_dataloader, config: Dict, device: str = "cuda",
                 output_dir: str = "results", step_dir: str = None, distributed: bool = False, rank: int = 0,
                 world_size: int = 1, local_rank: Optional[int] = None):
        self.model = model.to(device)
        self.train_dataloader = train_dataloader
        self.val_dataloader = val_dataloader
        self.config = config
        self.device = device
        self.distributed = bool(distributed) and dist.is_available() and dist.is_initialized()
        self.rank, self.world_size, self.local_rank = rank, world_size, local_rank
        self.is_main_process = (not self.distributed) or rank == 0
        self.output_dir = Path(output_dir)
        self.step_dir = Path(step_dir) if step_dir else self.output_dir
        self.checkpoint_dir = self.output_dir / "checkpoints"
        self.metrics_dir = self.step_dir / "metrics"
        if self.is_main_process:
            self.checkpoint_dir.mkdir(parents=True, exist_ok=True)
            self.metrics_dir.mkdir(parents=True, exist_ok=True)

        opt = config.get("optimization", {})
        self.use_amp = bool(opt.get("use_amp", False))
        if not self.use_amp:
            # the reference runs the step under autocast(enabled=use_amp) (trainer_backbone.py:434): without it every
            # product of the forward and backward pass is fp32. The training kernels here compute in bf16 (fp32
            # accumulation, fp32 master weights and optimizer); fp32 exists for inference only
            # (DiffusionBackbone.compute_dtype = "fp32"). Training in narrower arithmetic than asked for is refused.
            raise NotImplementedError(
                "BackboneTrainer: optimization.use_amp = false (or absent) asks for an fp32 training step; the bidiff "
                "training kernels compute in bf16 only. Set optimization.use_amp: true (what the reference's "
                "configs/config.yaml does) to train with bf16 products and fp32 accumulation.")
        self.grad_accum_steps = int(opt.get("gradient_accumulation_steps", 1))
        tc = config["training_backbone"]
        self.learning_rate = _f(tc["learning_rate"], "learning_rate")
        self.weight_decay = _f(tc["weight_decay"], "weight_decay")
        self.warmup_steps = int(_f(tc["warmup_steps"], "warmup_steps"))
        self.max_steps = int(_f(tc["max_steps"], "max_steps"))
        self.gradient_clip_norm = _f(tc["gradient_clip_norm"], "gradient_clip_norm")
        self.eval_every = int(_f(tc["eval_every"], "eval_every"))
        self.save_every = int(_f(tc["save_every"], "save_every"))
        self.num_epochs = int(_f(tc.get("num_epochs", 50), "num_epochs"))
        ck = config.get("checkpointing", {})
        self.save_best_only = ck.get("save_best_only", True)
        self.save_last = ck.get("save_last", False)
        self.save_periodic = ck.get("save_periodic", False)
        if self.distributed:   # same start on every rank, as DDP's construction broadcast does
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                dist.broadcast(t.data, src=0)
        # AdamW or Muon + AdamW groups, driven by config["optimization"]["optimizer"] as in the reference
        # (trainer_backbone.py:164-178)
        self.optimizer, info = build_step1_backbone_optimizer(self.model, opt, self.learning_rate, self.weight_decay)
        self.optimizer.skip_nonfinite = self.use_amp
        if self.is_main_process and info.get("type") == "muon_adamw":
            print("Using Muon+AdamW optimizer groups: "
                  f"muon_tensors={info.get('num_muon_tensors')}, adamw_tensors={info.get('num_adamw_tensors')}, "
                  f"adamw_lr={info.get('adamw_lr'):.3e}, muon_lr={info.get('muon_lr'):.3e}")
        warm = LinearLR(self.optimizer, start_factor=0.01, end_factor=1.0, total_iters=max(1, self.warmup_steps))
        cos = CosineAnnealingLR(self.optimizer, T_max=max(1, self.max_steps - self.warmup_steps), eta_min=1e-6)
        self.scheduler = SequentialLR(self.optimizer, schedulers=[warm, cos], milestones=[self.warmup_steps])

        # data parallel: the gradient exchange in two buckets, the upper half of the network's gradients reduced
        # while the lower half's backward still runs (BD_OVERLAP_ALLREDUCE=0: one blocking all-reduce per step)
        self._exchange = None
        if self.distributed and self.world_size > 1:
            bb = getattr(self.model, "backbone", None)
            split_layer = len(bb.layers) // 2 if bb is not None else 0
            off = tail_split_offset(self.optimizer, bb, split_layer) if bb is not None else None
            if off is None or os.environ.get("BD_OVERLAP_ALLREDUCE", "1") == "0":
                off = self.optimizer.flat_grad.numel()           # nothing to start early
            self._exchange = GradientExchange(self.optimizer.flat_grad, off, self.world_size)
            _autograd.set_tail_ready_hook(self._exchange.tail_ready, split_layer, model=bb)
        self.global_step = 0
        self.best_val_loss = float("inf")
        self.train_losses, self.val_losses, self.val_steps, self.learning_rates = [], [], [], []
        self.epoch_train_losses, self.epoch_val_losses = [], []

    # ------------------------------------------------------------------ helpers
    def _get_model_state(self):
        return self.model.state_dict()

    def _reduce_mean(self, value: float) -> float:
        if not self.distributed:
            return value
        t = torch.tensor(value, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return (t / self.world_size).item()

    def _barrier(self):
        if self.distributed:
            dist.barrier()

    # ------------------------------------------------------------------ the hot step
    def train_step_device(self, input_ids, attention_mask, should_step: bool = True, lengths=None):
        """One micro-batch on device tensors; returns the (unscaled) loss as a device scalar
        without synchronising the host (reference _train_step, trainer_backbone.py:414-457).
        ``lengths``: optional host integers (tokens per sequence) for the packed-row layout."""
        out = self.model(input_ids, attention_mask, lengths=lengths) if lengths is not None \
            else self.model(input_ids, attention_mask)
        loss = out["loss"] / self.grad_accum_steps
        if should_step and self._exchange is not None:
            self._exchange.arm()
        loss.backward()
        if should_step:
            if self._exchange is not None:
                if self._exchange._tail is None:
                    self.optimizer.sync_grads()      # gradients autograd kept beside the flat buffer
                self._exchange.finish()
            elif self.distributed:
                self.optimizer.sync_grads()
                allreduce_mean_(self.optimizer.flat_grad, self.world_size)
            self.optimizer.step(max_norm=self.gradient_clip_norm)
            self.optimizer.zero_grad()
            self.scheduler.step()
        return out["loss"].detach()

    def _train_step(self, batch, should_step: bool) -> float:
        ids = batch["input_ids"].to(self.device)
        am = batch["attention_mask"].to(self.device)
        from ..data.loader import ROW_LENGTHS_KEY
        return float(self.train_step_device(ids, am, should_step, lengths=batch.get(ROW_LENGTHS_KEY)).item())

    def _validate(self) -> float:
        self.model.eval()
        total, n = 0.0, 0
        with torch.no_grad():
            for batch in self.val_dataloader:
                out = self.model(batch["input_ids"].to(self.device), batch["attention_mask"].to(self.device))
                total += out["loss"].item()
                n += 1
        return self._reduce_mean(total / max(n, 1))

    def _train_epoch(self, epoch: int) -> float:
        self.model.train()
        total, n = 0.0, 0
        if self.distributed:
            for s in (getattr(self.train_dataloader, "sampler", None), getattr(self.train_dataloader, "batch_sampler", None)):
                if hasattr(s, "set_epoch"):
                    s.set_epoch(epoch)
        nb = len(self.train_dataloader)
        for i, batch in enumerate(self.train_dataloader):
            if self.global_step >= self.max_steps:
                break
            should_step = ((i + 1) % self.grad_accum_steps == 0) or (i + 1 == nb)
            loss = self._train_step(batch, should_step)
            total += loss
            n += 1
            if self.is_main_process:
                self.train_losses.append(loss)
                self.learning_rates.append(self.optimizer.param_groups[0]["lr"])
            if should_step:
                self.global_step += 1
                if self.global_step % self.eval_every == 0:
                    val = self._validate()
                    self.model.train()
                    if self.is_main_process:
                        self.val_losses.append(val)
                        self.val_steps.append(self.global_step)
                        self._save_checkpoint(val, epoch)
                    self._barrier()
                if (not self.save_best_only) and self.save_periodic and self.global_step % self.save_every == 0:
                    if self.is_main_process:
                        torch.save(self._checkpoint(float("nan"), epoch), self.checkpoint_dir / f"backbone_step_{self.global_step}.pt")
                    self._barrier()
        self._barrier()
        return self._reduce_mean(total / max(n, 1))

    def train(self) -> Dict:
        val_loss, epoch = float("nan"), 0
        for epoch in range(self.num_epochs):
            tr = self._train_epoch(epoch)
            val_loss = self._validate()
            self.epoch_train_losses.append(tr)
            self.epoch_val_losses.append(val_loss)
            self._save_checkpoint(val_loss, epoch)
            self._barrier()
            if self.is_main_process:
                print(f"Epoch {epoch + 1}/{self.num_epochs} - Train Loss: {tr:.4f} - Val Loss: {val_loss:.4f} - "
                      f"LR: {self.optimizer.param_groups[0]['lr']:.2e}")
            if self.global_step >= self.max_steps:
                break
        self._save_checkpoint(val_loss, epoch, final=True)
        self._save_history()
        return {"train_losses": self.train_losses, "val_losses": self.val_losses,
                "learning_rates": self.learning_rates, "epoch_train_losses": self.epoch_train_losses,
                "epoch_val_losses": self.epoch_val_losses, "best_val_loss": self.best_val_loss,
                "fp8_phase2_eval": None}

    # ------------------------------------------------------------------ checkpoints / history
    def _checkpoint(self, val_loss, epoch):
        return {"epoch": epoch, "global_step": self.global_step, "model_state_dict": self._get_model_state(),
                "optimizer_state_dict": self.optimizer.state_dict(), "scheduler_state_dict": self.scheduler.state_dict(),
                "val_loss": val_loss, "best_val_loss": self.best_val_loss, "config": self.config}

    def _save_checkpoint(self, val_loss: float, epoch: int, final: bool = False):
        if not self.is_main_process:
            return
        if not math.isnan(val_loss) and val_loss < self.best_val_loss:
            self.best_val_loss = val_loss
            torch.save(self._checkpoint(val_loss, epoch), self.checkpoint_dir / "backbone_best.pt")
        if final and not self.save_best_only and self.save_last:      # reference trainer_backbone.py:631
            torch.save(self._checkpoint(val_loss, epoch), self.checkpoint_dir / "backbone_last.pt")

    def _save_history(self):
        if not self.is_main_process:
            return

        def dump(name, header, rows):
            with open(self.metrics_dir / name, "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(header)
                w.writerows(rows)
        # same columns and rounding as the reference (trainer_backbone.py:654-692): losses and bits per byte
        # (loss / ln 2) to 4 decimals, learning rate to 8
        bpb = 1.0 / math.log(2.0)
        r4 = lambda v: round(v, 4)  # noqa: E731
        dump("backbone_loss_curve.csv", ["step", "train_loss", "train_bpb", "learning_rate"],
             [(i, r4(l), r4(l * bpb), round(lr, 8)) for i, (l, lr) in enumerate(zip(self.train_losses, self.learning_rates))])
        if self.val_losses and self.val_steps:
            dump("backbone_val_loss.csv", ["step", "val_loss", "val_bpb"],
                 [(st, r4(v), r4(v * bpb)) for st, v in zip(self.val_steps, self.val_losses)])
        if self.epoch_train_losses and self.epoch_val_losses:
            dump("backbone_epoch_loss_curve.csv", ["epoch", "train_loss", "val_loss", "train_bpb", "val_bpb"],
                 [(i + 1, r4(a), r4(b), r4(a * bpb), r4(b * bpb))
                  for i, (a, b) in enumerate(zip(self.epoch_train_losses, self.epoch_val_losses))])

    def load_checkpoint(self, checkpoint_path: str):
        ck = torch.load(checkpoint_path, map_location=self.device, weights_only=True)   # tensors, numbers, the config dict
        self.model.load_state_dict(ck["model_state_dict"])
        self.optimizer.load_state_dict(ck["optimizer_state_dict"])
        self.scheduler.load_state_dict(ck["scheduler_state_dict"])
        self.global_step = ck["global_step"]
        self.best_val_loss = ck.get("best_val_loss", float("inf"))    # periodic checkpoints carry no such key
