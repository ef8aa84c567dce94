import sys, math, time, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
dev = torch.device('cuda:0')
lib = _lib.lib()
def bench(M, N, K, epi, iters=20):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.bfloat16)
    s = _lib.stream_handle()
    for _ in range(3):
        lib.bd_gemm_bf16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), M, N, N, K, N, epi, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        rc = lib.bd_gemm_bf16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), M, N, N, K, N, epi, s)
    e1.record(); torch.cuda.synchronize()
    assert rc == 0, lib.bd_last_error()
    ms = e0.elapsed_time(e1) / iters
    print(f"M={M} N={N} K={K} epi={epi}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s", flush=True)
M = 4096 * 128
bench(M, 1152, 384, 0)
bench(M, 384, 384, 2)
bench(M, 1536, 384, 1)
bench(M, 384, 1536, 2)
bench(8192, 8192, 8192, 0, 5)
bench(4096, 4096, 4096, 0, 10)
