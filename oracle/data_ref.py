"""TEST INFRASTRUCTURE - CPU restatement of the reference's batch assembly.

``dynamic_collate`` follows ``dynamic_collate_fn`` (src/data/dataset.py:216-240): pad every item to
the longest of the batch with ``pad_token_id`` / mask 0; ``fixed_collate`` is ``collate_fn`` (:200-213)
over items the dataset padded to ``max_length`` (dataset.py:72-84, tokenizer.py:193-199).
Items are the unpadded id lists ``tokenizer.encode(..., padding=False)`` gives."""
import numpy as np


def dynamic_collate(rows, pad_token_id):
    L = max(len(r) for r in rows)
    return fixed_collate(rows, pad_token_id, L)


def fixed_collate(rows, pad_token_id, max_length):
    ids = np.full((len(rows), max_length), pad_token_id, dtype=np.int64)
    mask = np.zeros((len(rows), max_length), dtype=np.int64)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = np.asarray(r, dtype=np.int64)
        mask[i, :len(r)] = 1
    return ids, mask
