"""Achieved bandwidth of the HBM-bound kernels of the path: algorithmic bytes / HIP-event duration."""
import sys, json, torch, ctypes as C
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
lib = _lib.lib()
s = _lib.stream_handle
P = _lib.ptr
res = {}
def timeit(name, fn, bytes_, iters=20, note=""):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): rc = fn()
    e1.record(); torch.cuda.synchronize()
    assert rc == 0, lib.bd_last_error()
    us = e0.elapsed_time(e1) / iters * 1e3
    res[name] = {"algorithmic_bytes": bytes_, "avg_us": round(us, 2), "GBps": round(bytes_ / us / 1e3, 1),
                 "frac_of_8TBps": round(bytes_ / us / 1e3 / 8000, 4), "note": note}
    print(name, res[name], flush=True)

# q(x_t | x_0): Philox uniforms in-kernel: ids 8 + attn 8 + noisy 8 + indicator 1 = 25 B/token
for B in (4096, 262144):
    L, T = 128, 100
    ids = torch.randint(5, 39, (B, L), device=dev); attn = torch.ones(B, L, dtype=torch.long, device=dev)
    t = torch.randint(1, T + 1, (B,), device=dev); sched = torch.linspace(0.05, 0.95, T + 1, device=dev)
    noisy = torch.empty_like(ids); ind = torch.empty(B, L, dtype=torch.uint8, device=dev)
    timeit(f"corrupt_kernel B={B} L={L}", lambda: lib.bd_corrupt(P(ids), P(attn), P(t), P(sched), T + 1, None, P(noisy), P(ind), B, L,
                                                                 1, 0, 2, 3, 42, 0, s()), 25 * B * L + 8 * B)
    del ids, attn, noisy, ind
# masked CE at the training shape: rows with weight 0 are never read (about 30 % of the rows are non-pad, ~half of them masked)
M, V = 32768, 803
lg = torch.randn(M, V, device=dev); tg = torch.randint(0, V, (M,), device=dev)
indm = (torch.rand(M, device=dev) < 0.16).to(torch.uint8); lse = torch.empty(M, device=dev); sums = torch.zeros(2, device=dev)
nz = int(indm.sum().item())
timeit("ce_fwd_kernel M=32768 V=803 (16% rows live)", lambda: lib.bd_masked_ce_fwd(P(lg), P(tg), P(indm), None, P(lse), P(sums), M, V, s()),
       nz * V * 4 + M * (8 + 1 + 4), note="reads only live rows")
indf = torch.ones(M, dtype=torch.uint8, device=dev)
timeit("ce_fwd_kernel M=32768 V=803 (all rows live)", lambda: lib.bd_masked_ce_fwd(P(lg), P(tg), P(indf), None, P(lse), P(sums), M, V, s()),
       M * V * 4 + M * 13)
g1 = torch.ones(1, device=dev); dl = torch.empty_like(lg)
timeit("ce_bwd_kernel M=32768 V=803 (all rows live)", lambda: lib.bd_masked_ce_bwd(P(lg), P(tg), P(indf), None, P(lse), P(sums), P(g1), P(dl), M, V, s()),
       2 * M * V * 4 + M * 17)
# LayerNorm and embeddings at the sampling shape
M, d = 4096 * 128, 384
x = torch.randn(M, d, device=dev); gm = torch.ones(d, device=dev); bt = torch.zeros(d, device=dev)
ob = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
timeit("layernorm_kernel M=524288 d=384", lambda: lib.bd_layernorm(P(x), P(gm), P(bt), P(ob), None, None, M, d, 1e-5, s()), M * d * 6)
ids = torch.randint(0, 39, (4096, 128), device=dev); tt = torch.randint(1, 101, (4096,), device=dev)
te, pe, me = torch.randn(39, d, device=dev), torch.randn(256, d, device=dev), torch.randn(101, d, device=dev)
timeit("embed_kernel B=4096 L=128 d=384", lambda: lib.bd_embed(P(ids), P(tt), P(te), P(pe), P(me), P(x), 4096, 128, d, te.shape[0], pe.shape[0], me.shape[0], s()), M * d * 4 + M * 8,
       note="tables are cache resident; bytes = output + ids")
# optimizer over the flat buffers of the BPE small model
n = 11382144
p_, g_, m_, v_ = (torch.randn(n, device=dev) * 0.01 for _ in range(4)); v_.abs_()
ss = torch.zeros(1, device=dev)
timeit("sumsq_kernel n=11.4M", lambda: lib.bd_sumsq_f32(P(g_), n, P(ss), s()), n * 4)
timeit("adamw_kernel n=11.4M", lambda: lib.bd_adamw_step(P(p_), P(g_), P(m_), P(v_), n, 3e-4, 0.9, 0.999, 1e-8, 0.01, 0.1, 0.03, P(ss), 1.0, s()), n * 28)
json.dump(res, open("gpurun_out/bandwidth_kernels.json", "w"), indent=1)
