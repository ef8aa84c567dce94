"""Stand-alone attention forward at the shapes of the large models: time per launch and TFLOP/s."""
import os, sys, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, 'discrete-diffusion-model-for-inverse-polymer-design_amd')); sys.path.insert(0, _ROOT)
from bidiff import _lib
if os.environ.get('BD_LIB_PATH'): _lib.LIB_PATH = os.environ['BD_LIB_PATH']      # A/B against another build
dev = torch.device('cuda:0'); lib = _lib.lib(); s = _lib.stream_handle()
for (B, H, L, dh) in ((128, 8, 512, 128), (2048, 6, 128, 128), (2048, 9, 128, 128), (256, 8, 512, 64), (2048, 8, 128, 64)):
    qkv = torch.randn(B * L, 3 * H * dh, device=dev).bfloat16()
    out = torch.empty(B * L, H * dh, dtype=torch.bfloat16, device=dev)
    f = lambda: lib.bd_attention_fwd(_lib.ptr(qkv), None, _lib.ptr(out), None, B, L, H, dh, s)
    for _ in range(3): assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"B={B} H={H} L={L} dh={dh}: {ms*1e3:8.1f} us  {4.0*B*H*L*L*dh/ms/1e9:7.1f} TF/s", flush=True)
