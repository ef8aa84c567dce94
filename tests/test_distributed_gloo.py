"""world_size-2 gloo tests (CPU) of the multi-process paths: the contiguous sample sharding used
by batch-sharded sampling and the single gradient-bucket all-reduce of the training step."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H  # noqa: F401  (sys.path)
from bidiff.parallel import GradientExchange, allreduce_mean_, sample_sharded, shard_range, tail_split_offset


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 256, 4096, 4097):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


class _FakeSampler:
    """Stands in for a ConstrainedSampler on a machine without a GPU: same sample_batch contract (independent
    batches, sequences numbered globally by seq_offset), each "sequence" a pure function of its global index and
    length, as the Philox-keyed device sampler's are."""

    def sample_batch(self, num_samples, seq_length, batch_size=256, show_progress=True, lengths=None, seq_offset=0):
        all_ids, strings = [], []
        idx = 0
        while idx < num_samples:
            cur = min(batch_size, num_samples - idx)
            rows = []
            for j in range(cur):
                gidx = seq_offset + idx + j
                ln = seq_length if lengths is None else lengths[idx + j]
                rs = np.random.RandomState(1000 + gidx)
                rows.append(rs.randint(5, 39, size=seq_length) * (np.arange(seq_length) < ln))
                strings.append(f"{gidx}:{ln}:{int(rows[-1].sum())}")
            all_ids.append(torch.from_numpy(np.stack(rows)))
            idx += cur
        return all_ids, strings


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
        # ---- the training step's two-bucket exchange on a fake flat gradient buffer: the tail is reduced
        # while "the rest of the backward pass" still writes the head; accumulation micro-steps exchange nothing
        n, split = 5003, 3001
        g2 = torch.Generator().manual_seed(7 + rank)
        flat = torch.zeros(n)
        ex = GradientExchange(flat, split, world)
        micro = torch.randn(n, generator=g2)
        flat += micro                        # accumulation micro-step: no arm(), no exchange
        ex.tail_ready()
        ok = ok and ex._tail is None and torch.equal(flat, micro)
        ex.arm()
        step = torch.randn(n, generator=g2)
        flat[split:] += step[split:]         # upper layers' gradients are complete ...
        ex.tail_ready()                      # ... their all-reduce starts
        ok = ok and ex._tail is not None
        flat[:split] += step[:split]         # lower layers finish meanwhile
        ex.finish()
        mine2 = micro + step
        gathered = [torch.zeros(n) for _ in range(world)]
        dist.all_gather(gathered, mine2)
        ok = ok and torch.allclose(flat, torch.stack(gathered).mean(0), atol=1e-6)
        # a backward pass that never reports the tail (gradients handed to autograd): one whole-buffer reduce
        flat.copy_(mine2)
        ex.arm()
        ex.finish()
        ok = ok and torch.allclose(flat, torch.stack(gathered).mean(0), atol=1e-6)
        # ---- batch-sharded sampling end to end: each rank takes its contiguous share of the global sample
        # indices (with the per-sample lengths), the shares concatenate to the single-process result
        fake = _FakeSampler()
        lengths = [5 + (i * 7) % 20 for i in range(37)]
        ids, strings, (start, stop) = sample_sharded(fake, 37, 32, 8, rank, world, lengths=lengths)
        parts = [None] * world
        dist.all_gather_object(parts, (start, stop, strings, [t.tolist() for t in ids]))
        if rank == 0:
            whole = _FakeSampler()
            w_ids, w_strings = whole.sample_batch(37, 32, batch_size=8, show_progress=False, lengths=lengths, seq_offset=0)
            parts.sort()
            ok = ok and parts[0][0] == 0 and parts[-1][1] == 37 and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            ok = ok and [s_ for p_ in parts for s_ in p_[2]] == w_strings
            flat_rows = [row for p_ in parts for t in p_[3] for row in t]
            ok = ok and flat_rows == [row for t in w_ids for row in t.tolist()]
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


def test_tail_split_offset_follows_the_flat_layout():
    """The offset from which the flat gradient buffer holds layers >= split, the final norm and the head."""
    class P:      # minimal stand-ins: the function only walks parameters() and compares identities
        def __init__(self, ps):
            self._ps = ps

        def parameters(self):
            return iter(self._ps)

    mk = lambda n: torch.nn.Parameter(torch.zeros(n))  # noqa: E731
    emb = [mk(5), mk(3)]
    layers = [P([mk(4), mk(2)]) for _ in range(4)]
    fn, head = P([mk(2), mk(2)]), P([mk(7)])

    class BB:
        pass
    bb = BB()
    bb.layers, bb.final_norm, bb.output_proj = layers, fn, head

    class Opt:
        pass
    opt = Opt()
    opt._params = emb + [p for l in layers for p in l.parameters()] + list(fn.parameters()) + list(head.parameters())
    opt._offsets = list(np.cumsum([0] + [p.numel() for p in opt._params[:-1]]))
    assert tail_split_offset(opt, bb, 2) == 8 + 2 * 6
    assert tail_split_offset(opt, bb, 0) is None and tail_split_offset(opt, bb, 4) is None
    opt._params = opt._params[::-1]      # a layout that does not end with the block (e.g. grouped by shape)
    assert tail_split_offset(opt, bb, 2) is None
