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


def train_steps_bpe(sd, cfg, ids, attn, T, steps, warmup=1, lr=3e-4, weight_decay=0.01, max_norm=1.0, seed=0):
    """The reference's training step restated on the CPU in fp32: corruption (diffusion.py:84-127), denoiser
    forward, masked CE (:169-209), backward, clip_grad_norm_ and AdamW (trainer_backbone.py:414-457), dropout as
    the identity. Returns (seconds per step over ``steps`` timed steps, last loss)."""
    params = {k: torch.nn.Parameter(v.clone()) for k, v in sd.items() if v.dtype.is_floating_point and k != "mask_schedule"}
    opt = torch.optim.AdamW(params.values(), lr=lr, weight_decay=weight_decay)
    sched = BR.mask_schedule(T)
    g = torch.Generator().manual_seed(seed)
    B, L = ids.shape
    el, loss = 0.0, None
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        t = torch.randint(1, T + 1, (B,), generator=g)
        u = torch.rand(B, L, generator=g)
        ind = (u < sched[t][:, None]) & (attn == 1) & (ids != 2) & (ids != 3)
        noisy = torch.where(ind, torch.ones_like(ids), ids)
        lg = BR.logits(params, cfg, noisy, t, attn, prefix="backbone.")
        loss = BR.masked_ce(lg, ids, ind, attn)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params.values(), max_norm)
        opt.step()
        if i >= warmup:
            el += time.perf_counter() - t0
    return el / max(steps, 1), float(loss.item())
