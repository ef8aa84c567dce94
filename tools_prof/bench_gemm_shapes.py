"""bd_gemm_bf16 (bias epilogue) at shapes given as M,N,K triples on the command line; with BD_GEMM128_RING / BD_GEMM256
in the environment this compares the dispatch rules of the three kernels."""
import sys, math, os, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, 'discrete-diffusion-model-for-inverse-polymer-design_amd')); sys.path.insert(0, _ROOT)
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib()
for spec in sys.argv[1:]:
    M, N, K = (int(x) for x in spec.split(","))
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev); out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    s = _lib.stream_handle()
    call = lambda: lib.bd_gemm_bf16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), M, N, N, K, N, 0, s)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): rc = call()
    e1.record(); torch.cuda.synchronize()
    assert rc == 0
    ms = e0.elapsed_time(e1) / 20
    print(f"M={M} N={N} K={K} tiles128={(M + 127) // 128 * (N // 128)}: {ms*1e3:8.1f} us {2*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
