"""Three launches of the eight-wave GEMM at the stress model's feed-forward shape (65536 x 4096 x 1024, GELU epilogue) and
three at its down projection (65536 x 1024 x 4096, fp32 residual epilogue), for rocprofv3 --pmc passes."""
import sys, math, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib()
for (M, N, K, epi) in ((65536, 4096, 1024, 1), (65536, 1024, 4096, 2)):
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.bfloat16)
    for _ in range(3):
        assert lib.bd_gemm_bf16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), M, N, N, K, N, epi, _lib.stream_handle()) == 0
    torch.cuda.synchronize()
