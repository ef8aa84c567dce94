"""LayerNorm backward at the training step's shape: gamma / beta sums into one copy against n_parts copies."""
import ctypes as C, os, sys, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, 'discrete-diffusion-model-for-inverse-polymer-design_amd')); sys.path.insert(0, _ROOT)
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib(); P = _lib.ptr; s = _lib.stream_handle
M, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 10240), 384
dy = torch.randn(M, d, device=dev).bfloat16(); x = torch.randn(M, d, device=dev)
st = torch.stack([x.mean(1), 1 / (x.var(1, unbiased=False) + 1e-5).sqrt()], 1).contiguous()
gm = torch.randn(d, device=dev); dx = torch.randn(M, d, device=dev); ob = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
drop = _lib.Dropout(0.1, 1, 2)
def timeit(name, fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {1e3 * e0.elapsed_time(e1) / iters:.1f} us", flush=True)
timeit("one copy", lambda: lib.bd_layernorm_bwd_cast(P(dy), P(x), P(st), P(gm), P(dx), P(dx), P(dg), P(db), M, d, P(ob), C.byref(drop), 3, s()))
ref_g, ref_b = None, None
for R in (2, 4, 8, 16, 32):
    parts = torch.zeros(R, 2, d, device=dev)
    timeit(f"{R} copies", lambda: lib.bd_layernorm_bwd_parts(P(dy), P(x), P(st), P(gm), P(dx), P(dx), P(parts), R, M, d, P(ob), C.byref(drop), 3, s()))
# same sums
dg.zero_(); db.zero_(); dx0 = torch.randn(M, d, device=dev)
d1 = dx0.clone(); lib.bd_layernorm_bwd_cast(P(dy), P(x), P(st), P(gm), P(d1), P(d1), P(dg), P(db), M, d, P(ob), C.byref(drop), 3, s())
parts = torch.zeros(8, 2, d, device=dev); d2 = dx0.clone(); g2 = torch.zeros(d, device=dev); b2 = torch.zeros(d, device=dev)
lib.bd_layernorm_bwd_parts(P(dy), P(x), P(st), P(gm), P(d2), P(d2), P(parts), 8, M, d, P(ob), C.byref(drop), 3, s())
class Job(C.Structure):
    _fields_ = [("parts", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p)]
arr = (Job * 1)(Job(parts.data_ptr(), g2.data_ptr(), b2.data_ptr()))
jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
assert lib.bd_layernorm_parts_finish(P(jobs), 1, 8, d, s()) == 0
torch.cuda.synchronize()
print("max diff", float((g2 - dg).abs().max()), float((b2 - db).abs().max()), "dx equal", bool(torch.equal(d1, d2)), "parts zero", float(parts.abs().max()))
