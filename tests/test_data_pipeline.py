"""Training input pipeline: the bucket sampler and the batch assembly against vectors captured from
the reference's LengthBucketBatchSampler / PolymerDataset / (dynamic_)collate_fn
(tests/golden/gen_data_fixtures.py). CPU tests cover the host logic and the oracle; the GPU tests
check the device-side assembly bit for bit."""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import replay_inputs as RI
from oracle import data_ref as DR
from bidiff.data import DeviceBatchLoader, LengthBucketBatchSampler, TokenStore

BLOB = np.load(os.path.join(H.GOLDEN, "data.npz"))
N_DYN = int(BLOB["n_dyn_batches"])


def golden_rows():
    """Unpadded id rows of the fixture's data set, recovered from the reference's own batches."""
    rows = {}
    for b in range(N_DYN):
        idx, ids, mask = BLOB[f"dyn/{b}/indices"], BLOB[f"dyn/{b}/input_ids"], BLOB[f"dyn/{b}/attention_mask"]
        for i, r in enumerate(idx):
            rows[int(r)] = ids[i, :int(mask[i].sum())].tolist()
    assert sorted(rows) == list(range(len(BLOB["lengths"])))
    return [rows[i] for i in range(len(rows))]


class RecordedTokenizer:
    """Stands in for the reference tokenizer (which does not travel): encode() returns the ids the
    reference produced for that string."""
    pad_token_id, vocab_size = 0, 38

    def __init__(self):
        self.max_length = RI.POLYMER_MAX_LENGTH
        self.table = dict(zip(RI.polymer_strings(), golden_rows()))

    def encode(self, s, add_special_tokens=True, padding=True, return_attention_mask=True):
        assert add_special_tokens and not padding
        return {"input_ids": list(self.table[s])}


@pytest.mark.parametrize("ci", range(len(RI.SAMPLER_CASES)))
def test_bucket_sampler_yields_the_reference_batches(ci):
    n, bs, drop, shuf, seed, mult, reps, rank, epoch = RI.SAMPLER_CASES[ci]
    smp = LengthBucketBatchSampler(RI.sampler_lengths(n).tolist(), bs, drop_last=drop, shuffle=shuf, seed=seed,
                                   bucket_size_multiplier=mult, num_replicas=reps, rank=rank)
    smp.set_epoch(epoch)
    batches = list(smp)
    assert [len(b) for b in batches] == BLOB[f"sampler/{ci}/sizes"].tolist()
    assert [i for b in batches for i in b] == BLOB[f"sampler/{ci}/flat"].tolist()
    assert len(smp) == int(BLOB[f"sampler/{ci}/len"])
    lengths = RI.sampler_lengths(n)
    for b in batches:          # inside a batch the lengths never increase
        assert all(lengths[b[i]] >= lengths[b[i + 1]] for i in range(len(b) - 1))


def test_bucket_sampler_argument_checks():
    for kw in (dict(batch_size=0), dict(batch_size=4, bucket_size_multiplier=0), dict(batch_size=4, num_replicas=0),
               dict(batch_size=4, num_replicas=2, rank=2), dict(batch_size=4, rank=-1)):
        with pytest.raises(ValueError):
            LengthBucketBatchSampler([5, 6, 7], **kw)


def test_oracle_collate_matches_reference():
    rows = golden_rows()
    assert [len(r) for r in rows] == BLOB["lengths"].tolist()
    for b in range(N_DYN):
        idx = BLOB[f"dyn/{b}/indices"]
        ids, mask = DR.dynamic_collate([rows[i] for i in idx], 0)
        assert np.array_equal(ids, BLOB[f"dyn/{b}/input_ids"]) and np.array_equal(mask, BLOB[f"dyn/{b}/attention_mask"])
        if b < 2:
            ids, mask = DR.fixed_collate([rows[i] for i in idx], 0, RI.POLYMER_MAX_LENGTH)
            assert np.array_equal(ids, BLOB[f"fix/{b}/input_ids"]) and np.array_equal(mask, BLOB[f"fix/{b}/attention_mask"])


def test_token_store_build_save_load_and_checks(tmp_path):
    tok = RecordedTokenizer()
    store = TokenStore.from_strings(RI.polymer_strings(), tok)
    assert len(store) == len(BLOB["lengths"]) and store.lengths.tolist() == BLOB["lengths"].tolist()
    assert store.tokens.dtype == np.uint16 and store.max_length == RI.POLYMER_MAX_LENGTH
    rows = golden_rows()
    assert all(store.row(i).tolist() == rows[i] for i in (0, 5, len(rows) - 1))
    path = str(tmp_path / "store.npz")
    store.save(path)
    again = TokenStore.load(path)
    assert np.array_equal(again.tokens, store.tokens) and np.array_equal(again.offsets, store.offsets)
    assert (again.pad_token_id, again.max_length) == (store.pad_token_id, store.max_length)
    with pytest.raises(ValueError):
        TokenStore(np.zeros(4, np.uint16), np.asarray([0, 2, 2, 4]), 0, 8)       # empty row
    with pytest.raises(ValueError):
        TokenStore(np.zeros(4, np.uint16), np.asarray([0, 4]), 0, 3)             # longer than max_length
    with pytest.raises(ValueError):
        TokenStore(np.zeros(4, np.uint16), np.asarray([1, 4]), 0, 8)
    with pytest.raises(RuntimeError):       # no CPU assembly path
        DeviceBatchLoader(store, [[0, 1]], "cpu")


@pytest.mark.gpu
def test_device_loader_batches_equal_reference(gpu_device):
    store = TokenStore.from_strings(RI.polymer_strings(), RecordedTokenizer())
    smp = LengthBucketBatchSampler(store.lengths.tolist(), 16, seed=11, bucket_size_multiplier=2)
    loader = DeviceBatchLoader(store, smp, gpu_device)
    assert len(loader) == N_DYN
    n = 0
    for b, batch in enumerate(loader):
        assert batch["input_ids"].dtype == torch.long and batch["input_ids"].is_cuda
        assert np.array_equal(batch["input_ids"].cpu().numpy(), BLOB[f"dyn/{b}/input_ids"]), b
        assert np.array_equal(batch["attention_mask"].cpu().numpy(), BLOB[f"dyn/{b}/attention_mask"]), b
        n += 1
    assert n == N_DYN
    loader.check()
    fixed = DeviceBatchLoader(store, smp, gpu_device, pad_to_max_length=True)
    for b in range(2):
        batch = fixed.assemble(BLOB[f"dyn/{b}/indices"])
        assert np.array_equal(batch["input_ids"].cpu().numpy(), BLOB[f"fix/{b}/input_ids"])
        assert np.array_equal(batch["attention_mask"].cpu().numpy(), BLOB[f"fix/{b}/attention_mask"])
    # a second epoch reshuffles; ragged last batch; single-row batch
    loader.set_epoch(1)
    first = next(iter(loader))["input_ids"].cpu().numpy()
    assert first.shape != BLOB["dyn/0/input_ids"].shape or not np.array_equal(first, BLOB["dyn/0/input_ids"])
    one = loader.assemble([7])
    assert one["input_ids"].shape == (1, int(store.lengths[7])) and int(one["attention_mask"].sum()) == int(store.lengths[7])
    with pytest.raises(IndexError):
        loader.assemble([0, len(store)])
    with pytest.raises(ValueError):
        loader.assemble([])


@pytest.mark.gpu
def test_device_loader_host_running_ahead_of_the_stream(gpu_device):
    """Many batches assembled back to back behind a long-running kernel, no synchronisation in between:
    every batch must still hold its own rows (the pinned index buffers rotate and are rewritten only after
    the upload that read them has finished)."""
    dev = gpu_device
    rs = np.random.RandomState(3)
    lens = rs.randint(2, 60, size=3000)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tokens = rs.randint(5, 500, size=int(offsets[-1])).astype(np.uint16)
    store = TokenStore(tokens, offsets, 0, 60)
    loader = DeviceBatchLoader(store, None, dev, pad_to_max_length=True)
    a = torch.randn(6144, 6144, device=dev)
    torch.cuda.synchronize()
    for _ in range(12):           # a few hundred ms of queued work the uploads have to wait behind
        a = (a @ a).clamp_(-1, 1)
    batches, want = [], []
    for i in range(3 * loader.ring_slots + 2):
        idx = rs.randint(0, 3000, size=64)
        batches.append(loader.assemble(idx))
        want.append(idx)
    torch.cuda.synchronize()
    for batch, idx in zip(batches, want):
        ids, mask = DR.dynamic_collate([store.row(i) for i in idx], 0)
        got = batch["input_ids"].cpu().numpy()
        assert np.array_equal(got[:, :ids.shape[1]], ids) and (got[:, ids.shape[1]:] == 0).all()
    loader.check()


@pytest.mark.gpu
def test_collate_kernel_flags_bad_rows_and_large_batches(gpu_device):
    from bidiff import _lib
    dev = gpu_device
    rs = np.random.RandomState(0)
    lens = rs.randint(1, 200, size=5000)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tokens = rs.randint(5, 60000, size=int(offsets[-1])).astype(np.uint16)
    store = TokenStore(tokens, offsets, 3, 200)
    loader = DeviceBatchLoader(store, None, dev)
    idx = rs.randint(0, 5000, size=4096)
    batch = loader.assemble(idx)
    ids, mask = DR.dynamic_collate([store.row(i) for i in idx], 3)
    assert np.array_equal(batch["input_ids"].cpu().numpy(), ids) and np.array_equal(batch["attention_mask"].cpu().numpy(), mask)
    loader.check()
    # the raw entry point reports an out-of-range row and an over-long row through the flag
    tok, off = store.to_device(dev)
    bad = torch.tensor([0, 5000, 2], device=dev)
    out_i = torch.empty(3, 4, dtype=torch.long, device=dev)
    out_m = torch.empty_like(out_i)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().bd_collate_batch(_lib.ptr(tok), _lib.ptr(off), _lib.ptr(bad), 5000, 3, 4, 3, _lib.ptr(out_i),
                                           _lib.ptr(out_m), _lib.ptr(err), _lib.stream_handle()), "collate")
    flag = int(err.item())
    assert flag & 1 and (flag & 2) == (2 if max(lens[0], lens[2]) > 4 else 0)
    assert out_m[1].sum().item() == 0 and (out_i[1] == 3).all()


def test_bucket_sampler_ranks_partition_the_epoch():
    """Every rank gets the same number of samples, together they cover the data set (the wrap-around
    padding repeats at most world - 1 samples), and a new epoch reshuffles."""
    n, world = 1003, 4
    lengths = RI.sampler_lengths(n).tolist()
    per_rank = []
    for rank in range(world):
        smp = LengthBucketBatchSampler(lengths, 16, seed=5, bucket_size_multiplier=3, num_replicas=world, rank=rank)
        smp.set_epoch(2)
        per_rank.append([i for b in smp for i in b])
    assert len({len(x) for x in per_rank}) == 1 and len(per_rank[0]) == -(-n // world)
    allidx = [i for x in per_rank for i in x]
    assert set(allidx) == set(range(n)) and len(allidx) - n == world * len(per_rank[0]) - n < world
    smp = LengthBucketBatchSampler(lengths, 16, seed=5, bucket_size_multiplier=3, num_replicas=world, rank=0)
    smp.set_epoch(3)
    assert [i for b in smp for i in b] != per_rank[0]
