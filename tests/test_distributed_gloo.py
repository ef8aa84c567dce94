"""world_size-2 gloo tests (CPU) of the multi-process paths: the contiguous sample sharding used
by batch-sharded sampling and the single gradient-bucket all-reduce of the training step."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H  # noqa: F401  (sys.path)
from bidiff.parallel import allreduce_mean_, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 256, 4096, 4097):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(10007, generator=g)
        mine = flat.clone()
        allreduce_mean_(flat, world)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        want = torch.stack(gathered).mean(0)
        ok = torch.allclose(flat, want, atol=1e-7)
        start, stop = shard_range(1001, rank, world)
        sizes = [torch.zeros(1, dtype=torch.long) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([stop - start]))
        ok = ok and int(sum(s.item() for s in sizes)) == 1001
        out.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_bucket_allreduce_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
