import sys, math, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib(); s = _lib.stream_handle()

def run(B, L, d, H, iters=10, masked=False, check=True):
    g = torch.Generator().manual_seed(B + L + d + H)
    dh = d // H
    x = torch.randn(B * L, d, generator=g) * 1.5 + 0.3
    gm = 1 + 0.2 * torch.randn(d, generator=g); bt = 0.3 * torch.randn(d, generator=g)
    Wqkv = torch.randn(3 * d, d, generator=g) / math.sqrt(d); bqkv = 0.3 * torch.randn(3 * d, generator=g)
    Wo = torch.randn(d, d, generator=g) / math.sqrt(d); bo = 0.3 * torch.randn(d, generator=g)
    km = None
    if masked:
        km = torch.ones(B, L, dtype=torch.uint8)
        for b in range(B):
            n = int(torch.randint(1, L + 1, (1,), generator=g))
            km[b, n:] = 0
        if B > 2: km[2] = 0      # a fully masked sequence
    scale = math.log2(math.e) / math.sqrt(dh)
    wf = Wqkv * gm[None, :]; bf = bqkv + Wqkv @ bt
    wf[:d] *= scale; bf = bf.clone(); bf[:d] *= scale
    wqkv_b = wf.bfloat16().to(dev); wo_b = Wo.bfloat16().to(dev)
    pack = torch.empty(lib.bd_attn_pack_bytes(d), dtype=torch.uint8, device=dev)
    _lib.check(lib.bd_attn_pack_weights(_lib.ptr(wqkv_b), _lib.ptr(wo_b), _lib.ptr(pack), d, H, s), "pack")
    xd = x.to(dev); out = torch.empty_like(xd); bfd = bf[:d].contiguous().to(dev); bod = bo.to(dev); bofd = (bo + Wo @ bf[2 * d:]).to(dev)
    kmd = km.to(dev) if km is not None else None
    call = lambda: lib.bd_attn_block_fused(_lib.ptr(xd), _lib.ptr(out), _lib.ptr(pack), _lib.ptr(bfd), _lib.ptr(bod), _lib.ptr(bofd),
                                           _lib.ptr(kmd) if kmd is not None else None, B, L, d, H, 1e-5, s)
    _lib.check(call(), "attn"); torch.cuda.synchronize()
    if check:
        nb = min(B, 8)
        xs = x[:nb * L].view(nb, L, d)
        hN = torch.nn.functional.layer_norm(xs, (d,), gm, bt, 1e-5)
        qkv = hN @ Wqkv.t() + bqkv
        q, k, v = [t.view(nb, L, H, dh).transpose(1, 2) for t in qkv.split(d, dim=-1)]
        sc = q @ k.transpose(-1, -2) / math.sqrt(dh)
        if km is not None:
            sc = sc.masked_fill(km[:nb, None, None, :] == 0, float('-inf'))
        pr = torch.softmax(sc, dim=-1)
        pr = torch.nan_to_num(pr, nan=0.0)
        o = (pr @ v).transpose(1, 2).reshape(nb, L, d)
        ref = xs + o @ Wo.t() + bo
        got = out[:nb * L].cpu().view(nb, L, d)
        err = (got - ref).abs().max().item()
        print(f"  check B={B} L={L} d={d} H={H} masked={masked}: max abs err {err:.3e} (ref max {ref.abs().max().item():.2f}) finite={torch.isfinite(got).all().item()}", flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * L * (4 * d * d + 2 * L * d)
    print(f"ATTN B={B} L={L} d={d} H={H}: {ms*1e3:8.1f} us  {fl/ms/1e9:8.1f} TF/s", flush=True)

run(3, 128, 384, 3, 1)
run(4, 128, 384, 3, 1, masked=True)
run(5, 100, 384, 3, 1, masked=True)
run(2, 128, 256, 4, 1)
run(2, 77, 256, 2, 1, masked=True)
run(3, 128, 128, 2, 1)
run(3, 50, 128, 4, 1, masked=True)
run(4096, 128, 384, 3)
