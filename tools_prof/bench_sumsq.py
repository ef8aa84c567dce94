import sys, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib(); P = _lib.ptr; s = _lib.stream_handle
for n in (10_800_000, 10_800_003, 1000, 5):
    x = torch.randn(n, device=dev); out = torch.zeros(1, device=dev)
    assert lib.bd_sumsq_f32(P(x), n, P(out), s()) == 0
    ref = float((x.double() ** 2).sum())
    print(n, float(out), ref, abs(float(out) - ref) / ref)
x = torch.randn(10_800_000, device=dev); out = torch.zeros(1, device=dev)
for _ in range(5): lib.bd_sumsq_f32(P(x), x.numel(), P(out), s())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): lib.bd_sumsq_f32(P(x), x.numel(), P(out), s())
e1.record(); torch.cuda.synchronize()
print("sumsq 43 MB:", 1e3 * e0.elapsed_time(e1) / 50, "us")
