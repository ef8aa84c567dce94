"""Packed-row attention forward / backward at the training step's shape (3 heads of 128, lengths as bench.py draws
them), for several batch sizes: how the launches scale with the number of workgroups."""
import ctypes as C, os, sys, numpy as np, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, 'discrete-diffusion-model-for-inverse-polymer-design_amd')); sys.path.insert(0, _ROOT)
import bench
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib(); P = _lib.ptr; s = _lib.stream_handle
H, dh, L = 3, 128, 128
d = H * dh

def timeit(fn, iters=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters

for B in (32, 85, 170, 256, 512):
    ids, attn = bench.bpe_batch(B, L, 803, 1)
    lens = attn.sum(1)
    cu = np.zeros(B + 1, dtype=np.int32); cu[1:] = np.cumsum(lens)
    n_real = int(cu[-1]); n_rows = (n_real + 127) // 128 * 128
    cu_d = torch.from_numpy(cu).to(dev); km = torch.from_numpy(attn.astype(np.uint8)).to(dev)
    qkv = torch.randn(n_rows, 3 * d, device=dev).bfloat16(); out = torch.zeros(n_rows, d, device=dev, dtype=torch.bfloat16)
    dout = torch.randn(n_rows, d, device=dev).bfloat16(); dqkv = torch.zeros(n_rows, 3 * d, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, H, L, device=dev); delta = torch.empty(B, H, L, device=dev)
    drop = _lib.Dropout(0.1, 1, 2)
    f = timeit(lambda: lib.bd_attention_fwd_varlen(P(qkv), P(km), P(out), P(lse), P(cu_d), B, L, H, dh, C.byref(drop), 1, s()))
    b = timeit(lambda: lib.bd_attention_bwd_varlen(P(qkv), P(km), P(out), P(dout), P(lse), P(delta), P(dqkv), P(cu_d), B, L, H, dh, C.byref(drop), 1, s()))
    f0 = timeit(lambda: lib.bd_attention_fwd_varlen(P(qkv), P(km), P(out), P(lse), P(cu_d), B, L, H, dh, None, 1, s()))
    b0 = timeit(lambda: lib.bd_attention_bwd_varlen(P(qkv), P(km), P(out), P(dout), P(lse), P(delta), P(dqkv), P(cu_d), B, L, H, dh, None, 1, s()))
    print(f"B={B} rows={n_real} over64={(lens > 64).sum()}: fwd {f:.1f} us, bwd (both kernels) {b:.1f} us; without dropout {f0:.1f} / {b0:.1f}", flush=True)
