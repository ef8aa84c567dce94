"""TEST INFRASTRUCTURE - the reference's reverse-sampling loop restated on the CPU:
fp32 PyTorch denoiser (oracle/backbone_ref.py) + the C step/repair oracle. bench.py times it on
the host cores as the CPU baseline ("port"); tests use it as an end-to-end checker.
Follows Bi_Diffusion_SMILES/src/sampling/sampler.py:634-726 (loop) and :796-808 (initial state).
"""
import time

import numpy as np
import torch

from . import backbone_ref as BR
from . import sampler_ref as ora


def unmask_table(T, L):
    tab = np.ones((T + 1, L + 1), dtype=np.int32)
    for t in range(1, T + 1):
        p = 1.0 / t
        for n in range(L + 1):
            tab[t, n] = max(1, int(n * p))
    return tab


def sample_smiles(sd, cfg, tb, B, L, T, temperature=1.0, use_constraints=True, seed=0, seq_offset=0,
                  prefix="backbone."):
    """Full reverse sampling of B sequences on the CPU. Returns (ids, seconds in the denoiser,
    seconds in the step/repair code)."""
    ids = np.full((B, L), tb.mask_id, dtype=np.int64)
    ids[:, 0] = tb.bos_id
    ids[:, -1] = tb.eos_id
    attn = np.ones((B, L), dtype=np.int64)
    fixed = (ids != tb.mask_id).astype(np.uint8)
    ktab = unmask_table(T, L)
    t_net = t_step = 0.0
    last = None
    ta = torch.from_numpy(attn)
    for t in range(T, 0, -1):
        t0 = time.perf_counter()
        with torch.no_grad():
            lg = BR.logits(sd, cfg, torch.from_numpy(ids), torch.full((B,), t, dtype=torch.long), ta, prefix).numpy()
        t1 = time.perf_counter()
        o = ora.sample_step(tb, ids, lg, ktab[t], t, temperature=temperature, use_constraints=use_constraints,
                            fixed=fixed, seed=seed, seq_offset=seq_offset)
        ids, last = o["ids"], o["logits"]
        t2 = time.perf_counter()
        t_net += t1 - t0
        t_step += t2 - t1
    if use_constraints:
        t0 = time.perf_counter()
        ids = ora.sample_fixup(tb, ids, last)["ids"]
        t_step += time.perf_counter() - t0
    return ids, t_net, t_step
