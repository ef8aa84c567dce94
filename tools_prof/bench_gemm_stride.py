"""256 x 256-tile GEMM at row strides that are / are not powers of two (memory-channel spread of the operand rows)."""
import os, sys, math, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, 'discrete-diffusion-model-for-inverse-polymer-design_amd')); sys.path.insert(0, _ROOT)
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib()
def bench(M, N, K, iters=10):
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    s = _lib.stream_handle()
    call = lambda: lib.bd_gemm_bf16(_lib.ptr(A), _lib.ptr(W), None, _lib.ptr(out), M, N, N, K, N, 0, s)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): rc = call()
    e1.record(); torch.cuda.synchronize()
    assert rc == 0
    ms = e0.elapsed_time(e1) / iters
    print(f"M={M} N={N} K={K}: {ms*1e3:9.1f} us {2*M*N*K/ms/1e9:8.1f} TF/s", flush=True)
for K in (8192, 8256, 8320, 7936):
    bench(8192, 8192, K)
for K in (1024, 1088, 1152, 960):
    bench(65536, 4096, K)
