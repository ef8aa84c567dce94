"""Masked cross-entropy forward: time against the share of live rows and the row length."""
import sys, os, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, 'discrete-diffusion-model-for-inverse-polymer-design_amd')); sys.path.insert(0, _ROOT)
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib(); P = _lib.ptr; s = _lib.stream_handle
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters
for M, V in ((32768, 803), (10240, 803), (32768, 128)):
    lg = torch.randn(M, V, device=dev); tg = torch.randint(0, V, (M,), device=dev)
    lse = torch.empty(M, device=dev); sums = torch.zeros(32 * 17, device=dev)
    for frac in (0.0, 0.16, 0.5, 1.0):
        ind = (torch.rand(M, device=dev) < frac).to(torch.uint8)
        t = timeit(lambda: lib.bd_masked_ce_fwd_parts(P(lg), P(tg), P(ind), None, P(lse), P(sums), 16, M, V, s()))
        print(f"M={M} V={V} live={frac}: {t:.1f} us", flush=True)
