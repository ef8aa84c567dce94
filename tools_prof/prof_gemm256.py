import sys, math, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib()
M = N = K = 8192
A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    lib.bd_gemm_bf16(_lib.ptr(A), _lib.ptr(W), None, _lib.ptr(out), M, N, N, K, N, 0, _lib.stream_handle())
torch.cuda.synchronize()
