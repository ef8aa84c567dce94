import sys, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib()
def bench(B, H, L, dh, iters=20):
    d = H * dh
    qkv = torch.randn(B * L, 3 * d, device=dev).bfloat16()
    out = torch.empty(B * L, d, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, H, L, device=dev)
    s = _lib.stream_handle()
    call = lambda: lib.bd_attention_fwd(_lib.ptr(qkv), None, _lib.ptr(out), _lib.ptr(lse), B, L, H, dh, s)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): rc = call()
    e1.record(); torch.cuda.synchronize()
    assert rc == 0
    ms = e0.elapsed_time(e1) / iters
    fl = 4.0 * B * H * L * L * dh
    print(f"attn_fwd B={B} H={H} L={L} dh={dh}: {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF/s", flush=True)
bench(128, 8, 512, 128)
bench(128, 16, 512, 64)
bench(256, 3, 128, 128)
bench(64, 8, 2048, 128)
