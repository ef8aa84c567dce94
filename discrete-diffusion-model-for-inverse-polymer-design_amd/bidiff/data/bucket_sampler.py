"""Length-bucketed batch sampler: drop-in for the reference's ``LengthBucketBatchSampler``
(src/data/samplers.py:10-120). Host logic; it yields the same index lists for the same
(lengths, batch_size, seed, epoch, rank) because it draws from the same generator
(``random.Random(seed + epoch)``) in the same order: shuffle all indices, shard them over the
ranks with wrap-around padding, sort each bucket of ``batch_size * bucket_size_multiplier``
indices by length (longest first, stable), cut it into batches."""
import math
import random
from typing import Iterator, List, Sequence


class LengthBucketBatchSampler:
    def __init__(self, lengths: Sequence[int], batch_size: int, drop_last: bool = False, shuffle: bool = True,
                 seed: int = 42, bucket_size_multiplier: int = 50, num_replicas: int = 1, rank: int = 0):
        if batch_size <= 0:
            raise ValueError(f"batch_size must be > 0, got {batch_size}")
        if bucket_size_multiplier <= 0:
            raise ValueError(f"bucket_size_multiplier must be > 0, got {bucket_size_multiplier}")
        if num_replicas <= 0:
            raise ValueError(f"num_replicas must be > 0, got {num_replicas}")
        if rank < 0 or rank >= num_replicas:
            raise ValueError(f"rank must be in [0, {num_replicas - 1}], got {rank}")
        self.lengths = [int(v) for v in lengths]
        self.batch_size = int(batch_size)
        self.drop_last = bool(drop_last)
        self.shuffle = bool(shuffle)
        self.seed = int(seed)
        self.bucket_size_multiplier = int(bucket_size_multiplier)
        self.num_replicas = int(num_replicas)
        self.rank = int(rank)
        self.epoch = 0
        self.total_size = len(self.lengths)
        self.samples_per_rank = int(math.ceil(self.total_size / self.num_replicas))

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def _rank_share(self, order: List[int]) -> List[int]:
        if self.num_replicas == 1:
            return order
        need = self.samples_per_rank * self.num_replicas
        order = order + order[: need - len(order)] if len(order) < need else order[:need]
        return order[self.rank:need:self.num_replicas]

    def __iter__(self) -> Iterator[List[int]]:
        order = list(range(self.total_size))
        if self.shuffle:
            random.Random(self.seed + self.epoch).shuffle(order)
        mine = self._rank_share(order)
        bucket = max(self.batch_size, self.batch_size * self.bucket_size_multiplier)
        lengths = self.lengths
        for b0 in range(0, len(mine), bucket):
            chunk = sorted(mine[b0:b0 + bucket], key=lambda i: lengths[i], reverse=True)
            for s0 in range(0, len(chunk), self.batch_size):
                batch = chunk[s0:s0 + self.batch_size]
                if len(batch) == self.batch_size or (batch and not self.drop_last):
                    yield batch

    def __len__(self) -> int:
        if self.drop_last:
            return self.samples_per_rank // self.batch_size
        return int(math.ceil(self.samples_per_rank / self.batch_size))
