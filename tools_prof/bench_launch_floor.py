"""Back-to-back launches of a kernel that does next to nothing (LayerNorm of one row through the library, and a
one-element torch add): the per-launch floor of a stream on this machine, as HIP events around 200 launches see it."""
import os, sys, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, 'discrete-diffusion-model-for-inverse-polymer-design_amd')); sys.path.insert(0, _ROOT)
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib()
def timed(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (1, 1024, 10240):
    x = torch.randn(M, 384, device=dev); g = torch.ones(384, device=dev); b = torch.zeros(384, device=dev)
    ob = torch.empty(M, 384, dtype=torch.bfloat16, device=dev)
    s = _lib.stream_handle()
    print(f"bd_layernorm M={M}: {timed(lambda: lib.bd_layernorm(_lib.ptr(x), _lib.ptr(g), _lib.ptr(b), _lib.ptr(ob), None, None, M, 384, 1e-5, s)):.2f} us per launch")
y = torch.zeros(1, device=dev)
print(f"torch add_ of one element: {timed(lambda: y.add_(1.0)):.2f} us per launch")
