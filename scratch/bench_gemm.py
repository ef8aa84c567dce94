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
import os
M = 4096 * 128
if not os.environ.get("ONLY_PANEL"):
    bench(M, 1152, 384, 0)
    bench(M, 384, 384, 2)
    bench(M, 1536, 384, 1)
    bench(M, 384, 1536, 2)
    bench(8192, 8192, 8192, 0, 5)

def benchp(M, N, K, epi, ln=False, iters=20):
    A = (torch.randn(M, K, device=dev) if ln else torch.randn(M, K, device=dev).bfloat16())
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.bfloat16)
    s = _lib.stream_handle()
    call = lambda: lib.bd_gemm_panel(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(out) if epi == 2 else None, M, N, K, N, epi, int(ln), 1e-5, s)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): rc = call()
    e1.record(); torch.cuda.synchronize()
    assert rc == 0, lib.bd_last_error()
    ms = e0.elapsed_time(e1) / iters
    byt = M*K*(4 if ln else 2) + M*N*(8 if epi == 2 else 2)
    print(f"PANEL M={M} N={N} K={K} epi={epi} ln={ln}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s  {byt/ms/1e9:6.2f} TB/s", flush=True)
print("--- panel kernel")
benchp(M, 1152, 384, 0, True)
benchp(M, 1152, 384, 0, False)
benchp(M, 384, 384, 2)
benchp(M, 1536, 384, 1, True)
benchp(M, 384, 1536, 2)

for dbg in (1, 2, 3, 4, 8, 5, 7, 15):
    os.environ["BD_PANEL_DBG"] = str(dbg)
    print("--- dbg", dbg)
    benchp(M, 1152, 384, 0, False)
    benchp(M, 384, 1536, 2)
