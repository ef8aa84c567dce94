"""Three launches of the stand-alone attention forward at the stress model's shape (B = 128, 8 heads of 128, L = 512),
for rocprofv3 --pmc passes."""
import os, sys, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, 'discrete-diffusion-model-for-inverse-polymer-design_amd')); sys.path.insert(0, _ROOT)
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib(); s = _lib.stream_handle()
B, H, L, dh = 128, 8, 512, 128
qkv = torch.randn(B * L, 3 * H * dh, device=dev).bfloat16()
out = torch.empty(B * L, H * dh, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    assert lib.bd_attention_fwd(_lib.ptr(qkv), None, _lib.ptr(out), None, B, L, H, dh, s) == 0
torch.cuda.synchronize()
