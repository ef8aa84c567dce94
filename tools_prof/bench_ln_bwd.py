"""LayerNorm backward at the training shape: time and achieved bandwidth."""
import sys, torch, ctypes as C
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = sys.argv[1]
dev = torch.device('cuda:0'); lib = _lib.lib(); s = _lib.stream_handle(); P = _lib.ptr
M, d = (int(sys.argv[2]) if len(sys.argv) > 2 else 32768), (int(sys.argv[3]) if len(sys.argv) > 3 else 384)
dy = torch.randn(M, d, device=dev).bfloat16(); x = torch.randn(M, d, device=dev)
stats = torch.stack([x.mean(1), 1 / x.std(1)], 1).contiguous(); g = torch.ones(d, device=dev)
dx = torch.randn(M, d, device=dev); dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
ob = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
drop = _lib.Dropout(0.1, 3, 1234)
for name, dr in (("no dropout", None), ("dropout", C.byref(drop))):
    call = lambda: lib.bd_layernorm_bwd_cast(P(dy), P(x), P(stats), P(g), P(dx), P(dx), P(dg), P(db), M, d, P(ob), dr, 3, s)
    for _ in range(3): assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'base'} M={M} d={d} {name}: {us:.1f} us, {M * d * (2 + 4 + 4 + 4 + 2) / us / 1e3:.0f} GB/s", flush=True)
