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


def sample_sharded(sampler, num_samples: int, seq_length: int, batch_size: int, rank: int, world_size: int,
                   lengths=None):
    """This rank's share of ``sample_batch(num_samples, ...)``; returns (ids list, strings, (start, stop))."""
    start, stop = shard_range(num_samples, rank, world_size)
    my_lengths = None if lengths is None else lengths[start:stop]
    kw = {"lengths": my_lengths} if my_lengths is not None else {}
    ids, strings = sampler.sample_batch(stop - start, seq_length, batch_size=batch_size, show_progress=False,
                                        seq_offset=start, **kw)
    return ids, strings, (start, stop)
