import os, sys, math, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
if os.environ.get('BD_LIB_PATH'): _lib.LIB_PATH = os.environ['BD_LIB_PATH']      # A/B against another build
dev = torch.device('cuda:0'); lib = _lib.lib()
def bench(M, N, K, epi, iters=20):
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.bfloat16)
    s = _lib.stream_handle()
    call = lambda: lib.bd_gemm_bf16(_lib.ptr(A), _lib.ptr(W), None, _lib.ptr(out), M, N, N, K, N, epi, s)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): rc = call()
    e1.record(); torch.cuda.synchronize()
    assert rc == 0
    ms = e0.elapsed_time(e1) / iters
    print(f"M={M} N={N} K={K} epi={epi}: {ms*1e3:9.1f} us {2*M*N*K/ms/1e9:8.1f} TF/s", flush=True)
for epi in (0, 2):
    bench(65536, 1024, 1024, epi); bench(65536, 1024, 4096, epi)
for epi in (0, 1):
    bench(65536, 3072, 1024, epi); bench(65536, 4096, 1024, epi)
