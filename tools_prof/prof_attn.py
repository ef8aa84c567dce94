import sys, math, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib(); s = _lib.stream_handle()
B, L, d, H = 4096, 128, 384, 3
x = torch.randn(B * L, d, device=dev)
wqkv = (torch.randn(3 * d, d, device=dev) / math.sqrt(d)).bfloat16(); wo = (torch.randn(d, d, device=dev) / math.sqrt(d)).bfloat16()
bq = torch.randn(3 * d, device=dev) * 0.1; bo = torch.randn(d, device=dev) * 0.1
pack = torch.empty(lib.bd_attn_pack_bytes(d), dtype=torch.uint8, device=dev)
lib.bd_attn_pack_weights(_lib.ptr(wqkv), _lib.ptr(wo), _lib.ptr(pack), d, H, s)
out = torch.empty_like(x)
for _ in range(3):
    lib.bd_attn_block_fused(_lib.ptr(x), _lib.ptr(out), _lib.ptr(pack), _lib.ptr(bq), _lib.ptr(bo), _lib.ptr(bo), None, B, L, d, H, 1e-5, s)
torch.cuda.synchronize()
