"""AdamW and sum-of-squares kernels on flat buffers of the small and the xl model's size: time, bandwidth, and a
checksum of the updated buffers (the same whatever the access width: BD_ADAMW_VEC=0 against the default)."""
import sys, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib(); P = _lib.ptr; s = _lib.stream_handle
for n in (11_400_003, 412_000_000):
    g0 = torch.Generator(device='cuda').manual_seed(3)
    p = torch.randn(n, device=dev, generator=g0); g = torch.randn(n, device=dev, generator=g0) * 0.01
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev); ss = torch.zeros(1, device=dev)
    def step():
        return lib.bd_adamw_step(P(p), P(g), P(m), P(v), n, 3e-4, 0.9, 0.999, 1e-8, 0.01, 0.1, 0.0316, None, 0.0, s())
    for _ in range(3): assert step() == 0
    chk = (float(p.double().sum()), float(m.double().sum()), float(v.double().sum()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): step()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 10
    print(f"adamw n={n}: {us:.1f} us, {28.0 * n / us / 1e6:.2f} TB/s, checksums {chk}", flush=True)
    e0.record()
    for _ in range(10): lib.bd_sumsq_f32(P(g), n, P(ss), s())
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 10
    print(f"sumsq n={n}: {us:.1f} us, {4.0 * n / us / 1e6:.2f} TB/s", flush=True)
    del p, g, m, v
    torch.cuda.empty_cache()

# the same AdamW launch right behind a burst of matrix products (as in a training step), timed by itself
import math
n = 412_000_000
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 0.01; m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
A = torch.randn(8192, 8192, device=dev).bfloat16(); W = (torch.randn(8192, 8192, device=dev) / 90).bfloat16()
out = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
def burst(k):
    for _ in range(k):
        lib.bd_gemm_bf16(P(A), P(W), None, P(out), 8192, 8192, 8192, 8192, 8192, 0, s())
for k in (0, 3, 20):
    ts = []
    for rep in range(4):
        burst(k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.bd_adamw_step(P(p), P(g), P(m), P(v), n, 3e-4, 0.9, 0.999, 1e-8, 0.01, 0.1, 0.0316, None, 0.0, s())
        e1.record(); torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1))
    print(f"adamw n={n} behind {k} products of 8192^3: {[round(t) for t in ts]} us", flush=True)
