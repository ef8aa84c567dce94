import sys, math, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
dev = torch.device('cuda:0')
lib = _lib.lib()
s = _lib.stream_handle()

def run(M, d, F, iters=10, check=True):
    g = torch.Generator().manual_seed(M + d + F)
    x = (torch.randn(M, d, generator=g) * 1.5 + 0.3)
    gm = 1 + 0.2 * torch.randn(d, generator=g); bt = 0.3 * torch.randn(d, generator=g)
    W1 = torch.randn(F, d, generator=g) / math.sqrt(d); b1 = 0.3 * torch.randn(F, generator=g)
    W2 = torch.randn(d, F, generator=g) / math.sqrt(F); b2 = 0.3 * torch.randn(d, generator=g)
    w1f = (W1 * gm).bfloat16().to(dev); w2b = W2.bfloat16().to(dev)
    b1f = (b1 + W1 @ bt).to(dev); b2d = b2.to(dev)
    pack = torch.empty(lib.bd_ffn_pack_bytes(d, F), dtype=torch.uint8, device=dev)
    _lib.check(lib.bd_ffn_pack_weights(_lib.ptr(w1f), _lib.ptr(w2b), _lib.ptr(pack), d, F, s), "pack")
    xd = x.to(dev); out = torch.empty_like(xd)
    call = lambda: lib.bd_ffn_fused(_lib.ptr(xd), _lib.ptr(out), _lib.ptr(pack), _lib.ptr(b1f), _lib.ptr(b2d), M, d, F, 1e-5, s)
    _lib.check(call(), "ffn"); torch.cuda.synchronize()
    if check:
        n = min(M, 4096)
        xs = xd[:n].double()
        mu = xs.mean(-1, keepdim=True); var = xs.var(-1, unbiased=False, keepdim=True)
        xn = ((xs - mu) / torch.sqrt(var + 1e-5)).float().bfloat16().double()
        h = xn @ w1f.double().t() + b1f.double()
        h = torch.nn.functional.gelu(h).float().bfloat16().double()
        ref = xs + h @ w2b.double().t() + b2d.double()
        err = (out[:n].double() - ref).abs().max().item()
        print(f"  check M={M} d={d} F={F}: max abs err {err:.3e} (ref max {ref.abs().max().item():.2f})", flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"FFN M={M} d={d} F={F}: {ms*1e3:8.1f} us  {4*M*d*F/ms/1e9:8.1f} TF/s  {M*d*8/ms/1e9:6.2f} TB/s", flush=True)

run(300, 384, 1536, 1)
run(128, 128, 96, 1)
run(1000, 256, 1024, 1)
run(4096 * 128, 384, 1536)
import os
for dbg in ():
    os.environ["BD_FFN_DBG"] = str(dbg)
    print("dbg", dbg)
    run(4096 * 128, 384, 1536, check=False)
