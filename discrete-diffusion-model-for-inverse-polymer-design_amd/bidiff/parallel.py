"""Multi-GPU helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm).

Sampling shards by sequence with no data-path collective: every rank takes a contiguous range
of the global sample indices (the index keys the random streams, so the result does not depend
on the number of ranks). Training exchanges exactly one thing per optimizer step: the flat
gradient bucket (mean over ranks).
"""
from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(num_samples: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[start, stop) of the contiguous block of global sample indices owned by ``rank``."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    base, rem = divmod(num_samples, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def allreduce_mean_(flat: torch.Tensor, world_size: int) -> torch.Tensor:
    """In-place mean over ranks of the whole-model gradient bucket (a single collective)."""
    if world_size > 1 and dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / world_size)
    return flat


class GradientExchange:
    """The training step's one exchange, cut into two buckets so that the first overlaps with the rest of the
    backward pass (what DDP's bucketed all-reduce does in the reference, trainer_backbone.py:216-219).

    The flat fp32 gradient buffer is laid out in parameter order, and backward produces gradients in reverse order:
    the tail ``flat[split:]`` (upper layers, final norm, vocabulary head) is complete when the backward pass has
    left layer ``split_layer``. ``tail_ready()`` is called at that point (by the backward pass, on the stream whose
    work completes the tail) and starts an asynchronous all-reduce of the tail; ``finish()`` reduces the head,
    waits for both and divides by the world size. Every rank issues the SAME collectives in the same order whatever
    its backward pass did: with a split inside the buffer always the tail's all-reduce, then the head's (if
    ``tail_ready`` was not called before ``finish`` - accumulated gradients handed to autograd, a backward pass
    without the hook - ``finish`` issues the tail's itself); without a split one all-reduce of the whole buffer.
    Which of the two it is depends on ``split`` alone, a fact of the model's layout that all ranks share, so ranks
    cannot meet in collectives of different sizes. Skipping everything on accumulation micro-steps is the
    reference's ``no_sync`` (:427-431). Not yet run with more than one RCCL rank on hardware
    (``BD_OVERLAP_ALLREDUCE=0`` keeps the single blocking all-reduce)."""

    def __init__(self, flat: torch.Tensor, split: int, world_size: int):
        if not (0 <= split <= flat.numel()):
            raise ValueError("GradientExchange: split outside the buffer")
        self.flat, self.split, self.world = flat, int(split), int(world_size)
        self.two_buckets = 0 < self.split < flat.numel()
        self._tail = None
        self.armed = False
        self.early_tails = 0          # how many steps started their tail from the backward pass (diagnostics)

    def _on(self):
        return self.world > 1 and dist.is_available() and dist.is_initialized()

    def arm(self):
        """Before the backward pass of an optimizer step: the next ``tail_ready`` starts the tail's all-reduce."""
        self.armed = True
        self._tail = None

    def tail_ready(self):
        if not self.armed or self._tail is not None or not self._on() or not self.two_buckets:
            return
        self._tail = dist.all_reduce(self.flat[self.split:], op=dist.ReduceOp.SUM, async_op=True)
        self.early_tails += 1

    def finish(self):
        self.armed = False
        if not self._on():
            return self.flat
        if not self.two_buckets:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        else:
            if self._tail is None:     # the backward pass did not reach the hook: same two collectives, same order
                self._tail = dist.all_reduce(self.flat[self.split:], op=dist.ReduceOp.SUM, async_op=True)
            dist.all_reduce(self.flat[:self.split], op=dist.ReduceOp.SUM)
            self._tail.wait()          # the current stream waits for the tail's collective
            self._tail = None
        self.flat.mul_(1.0 / self.world)
        return self.flat


def tail_split_offset(optimizer, backbone, split_layer: int):
    """Offset in ``optimizer.flat_grad`` from which on the buffer holds exactly the parameters of
    ``backbone.layers[split_layer:]``, the final norm and the vocabulary head; None if the optimizer's layout
    does not end with that block (e.g. the Muon layout, which groups matrices by shape)."""
    params, offsets = getattr(optimizer, "_params", None), getattr(optimizer, "_offsets", None)
    if not params or not offsets or not (0 < split_layer < len(backbone.layers)):
        return None
    tail = [p for blk in list(backbone.layers)[split_layer:] for p in blk.parameters()]
    tail += list(backbone.final_norm.parameters()) + list(backbone.output_proj.parameters())
    tail = [p for p in tail if p.requires_grad]
    if not tail or len(tail) >= len(params):
        return None
    got = params[len(params) - len(tail):]
    if any(a is not b for a, b in zip(got, tail)):
        return None
    return int(offsets[len(params) - len(tail)])


def sample_sharded(sampler, num_samples: int, seq_length: int, batch_size: int, rank: int, world_size: int,
                   lengths=None):
    """This rank's share of ``sample_batch(num_samples, ...)``; returns (ids list, strings, (start, stop))."""
    start, stop = shard_range(num_samples, rank, world_size)
    my_lengths = None if lengths is None else lengths[start:stop]
    kw = {"lengths": my_lengths} if my_lengths is not None else {}
    ids, strings = sampler.sample_batch(stop - start, seq_length, batch_size=batch_size, show_progress=False,
                                        seq_offset=start, **kw)
    return ids, strings, (start, stop)
