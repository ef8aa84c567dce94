import sys, math, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
dev = torch.device('cuda:0')
lib = _lib.lib()
def bench(M, N, K, ldy=None, iters=30):
    ldy = ldy or N
    dY = torch.randn(M, ldy, device=dev).bfloat16()
    X = torch.randn(M, K, device=dev).bfloat16()
    dW = torch.zeros(N, K, device=dev)
    s = _lib.stream_handle()
    call = lambda: lib.bd_gemm_wgrad_bf16(_lib.ptr(dY), ldy, _lib.ptr(X), K, _lib.ptr(dW), M, N, K, s)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): rc = call()
    e1.record(); torch.cuda.synchronize()
    assert rc == 0
    ms = e0.elapsed_time(e1) / iters
    print(f"wgrad M={M} N={N} K={K} ldy={ldy}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s", flush=True)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 128
bench(M, 384, 384)
bench(M, 384, 384, 1152)
bench(M, 1152, 384)
bench(M, 1536, 384)
bench(M, 384, 1536)
bench(M, 896, 384)
bench(M, 1536, 6144)
bench(M, 4608, 1536)
