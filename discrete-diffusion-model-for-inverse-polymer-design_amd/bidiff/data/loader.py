"""Training batches assembled on the device: stands where the reference has
``DataLoader(PolymerDataset, batch_sampler=LengthBucketBatchSampler, collate_fn=dynamic_collate_fn)``
(src/data/dataset.py:200-240, scripts/step1_train_backbone.py) and yields the same dicts
(``input_ids`` / ``attention_mask`` int64 [B, L]), already on the GPU. Per batch the host sends the
B row indices (pinned, asynchronous); ``bd_collate_batch`` gathers and pads from the resident store.
``pad_to_max_length=True`` gives the fixed-length batches of the default ``collate_fn``;
False pads to the longest row of the batch (``dynamic_collate_fn``)."""
import numpy as np
import torch

from .. import _lib


ROW_LENGTHS_KEY = "bidiff_row_lengths"


class DeviceBatchLoader:
    def __init__(self, store, batch_sampler, device, pad_to_max_length: bool = False):
        self.store, self.batch_sampler = store, batch_sampler
        self.device = torch.device(device)
        self.pad_to_max_length = bool(pad_to_max_length)
        self._tok, self._off = store.to_device(self.device)
        self._err = torch.zeros(1, dtype=torch.int32, device=self.device)
        # ring of pinned index buffers, each with the event of its last upload: a buffer is rewritten only
        # after the copy that read it has finished, so the host may run any number of batches ahead
        self._ring = []
        self._ring_pos = 0
        self.ring_slots = 4

    def __len__(self):
        return len(self.batch_sampler)

    def set_epoch(self, epoch: int):
        if hasattr(self.batch_sampler, "set_epoch"):
            self.batch_sampler.set_epoch(epoch)

    def assemble(self, indices):
        """One batch from store rows ``indices`` (any sequence of ints)."""
        idx = np.asarray(indices, dtype=np.int64)
        B = int(idx.size)
        if B == 0:
            raise ValueError("DeviceBatchLoader: empty batch")
        if idx.min() < 0 or idx.max() >= len(self.store):
            raise IndexError("DeviceBatchLoader: sample index out of range")
        L = self.store.max_length if self.pad_to_max_length else int(self.store.lengths[idx].max())
        didx = self._upload_indices(idx, B)
        ids = torch.empty(B, L, dtype=torch.long, device=self.device)
        mask = torch.empty(B, L, dtype=torch.long, device=self.device)
        rc = _lib.lib().bd_collate_batch(_lib.ptr(self._tok), _lib.ptr(self._off), _lib.ptr(didx), len(self.store), B, L,
                                         self.store.pad_token_id, _lib.ptr(ids), _lib.ptr(mask), _lib.ptr(self._err),
                                         _lib.stream_handle())
        _lib.check(rc, "bd_collate_batch")
        # ROW_LENGTHS_KEY (host integers: last attended position + 1 of every row, exactly what the mask says) lets the
        # training step lay out its packed rows without a device round trip. The key is this package's own: a
        # "lengths" entry of some other data set may mean something else and is never read by the trainer.
        return {"input_ids": ids, "attention_mask": mask, ROW_LENGTHS_KEY: np.minimum(self.store.lengths[idx], L)}

    def _upload_indices(self, idx, B):
        if len(self._ring) < self.ring_slots:
            self._ring.append([torch.empty(max(B, 1024), dtype=torch.long).pin_memory(), None])
            slot = self._ring[-1]
        else:
            slot = self._ring[self._ring_pos % self.ring_slots]
            self._ring_pos += 1
            if slot[1] is not None:
                slot[1].synchronize()          # the upload that last read this buffer is done
            if slot[0].numel() < B:
                slot[0] = torch.empty(B, dtype=torch.long).pin_memory()
        slot[0][:B].copy_(torch.from_numpy(idx))
        didx = slot[0][:B].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        slot[1] = ev
        return didx

    def __iter__(self):
        for batch in self.batch_sampler:
            yield self.assemble(batch)

    def check(self):
        """Raises if any batch so far hit an out-of-range row or an over-long sequence (one host sync)."""
        flag = int(self._err.item())
        if flag:
            raise RuntimeError(f"bd_collate_batch reported error bits {flag:#x} (1: index out of range, 2: row longer than L)")
