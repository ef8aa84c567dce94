"""bd_sample_step alone in hidden mode at a given vocabulary, with and without the chosen-row workspace."""
import os, sys, ctypes as C, numpy as np, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, 'discrete-diffusion-model-for-inverse-polymer-design_amd')); sys.path.insert(0, _ROOT)
from bidiff import _lib
from bidiff.sampling import core
from tests import helpers as H
variant = sys.argv[1] if len(sys.argv) > 1 else "bpe"
V_force = int(sys.argv[2]) if len(sys.argv) > 2 else 803
B, L, d, T = 4096, 128, 384, 50
dev = torch.device('cuda:0')
smp, tok = H.make_sampler(variant, num_steps=T)
tb = H.oracle_tables(smp)
dt = H.hip_tables(tb, dev)
print("fixture vocabulary", tb.V)
rs = np.random.RandomState(0)
hidden = torch.randn(B, L, d, device=dev).bfloat16()
wout = (torch.randn(max(tb.V, 1024), d, device=dev) / d ** 0.5).bfloat16()
ktab = core.unmask_count_table(T, L)
lib = _lib.lib()
for t, frac in ((40, 0.8), (10, 0.2), (2, 0.04)):
    ids0 = torch.from_numpy(np.where(rs.rand(B, L) < frac, tb.mask_id, 5).astype(np.int64)).to(dev)
    for cap in (0, 2, 4, 8):
        times = []
        for rep in range(6):
            ids = ids0.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            import tests.helpers
            a_ids = ids
            torch.cuda.synchronize(); e0.record()
            H.hip_step(tb, dt, a_ids, ktab[t], t, hidden=hidden, wout=wout, seed=3, row_cap=cap, use_constraints=True)
            e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        print(f"V={tb.V} t={t} masked={frac} k={ktab[t][int(frac * L)]} row_cap={cap}: {min(times) * 1e3:.0f} us (incl. workspace fill)")
