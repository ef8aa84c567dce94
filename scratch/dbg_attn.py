import sys, math, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
from tests.test_gpu_kernels import ref_attention
dev = torch.device('cuda:0')
def run(qkv, mask, B, L, H, dh):
    out = torch.full((B*L, H*dh), float('nan'), dtype=torch.bfloat16, device=dev)
    rc = _lib.lib().bd_attention_fwd(_lib.ptr(qkv.to(dev)), _lib.ptr(mask.to(dev)) if mask is not None else None, _lib.ptr(out), None, B, L, H, dh, _lib.stream_handle())
    _lib.check(rc, 'attn'); torch.cuda.synchronize()
    return out.float().cpu()
for dh, H in [(32, 1), (64, 1), (128, 1), (32, 2)]:
    for L in [16, 24, 128]:
        B = 1; d = H*dh
        g = torch.Generator().manual_seed(1)
        qkv = torch.randn(B*L, 3*d, generator=g).bfloat16()
        # diag 1: uniform attention
        q2 = qkv.clone(); q2[:, :2*d] = 0
        e1 = (run(q2, None, B, L, H, dh) - ref_attention(q2, None, B, L, H, dh)).abs()
        # diag 2: V = key index
        q3 = qkv.clone(); q3[:, 2*d:] = torch.arange(L).float()[:, None].bfloat16()
        e2 = (run(q3, None, B, L, H, dh) - ref_attention(q3, None, B, L, H, dh)).abs()
        e3 = (run(qkv, None, B, L, H, dh) - ref_attention(qkv, None, B, L, H, dh)).abs()
        print(f"dh={dh} H={H} L={L}: uniformP err={e1.max():.4f}  V=keyidx err={e2.max():.4f}  full err={e3.max():.4f}")
        if e1.max() > 0.02:
            bad = (e1 > 0.02)
            print("   uniformP bad rows(q):", bad.any(1).nonzero().flatten().tolist()[:20], " bad cols(d):", bad.any(0).nonzero().flatten().tolist()[:40])
        if e3.max() > 0.02:
            bad = (e3 > 0.02)
            print("   full bad rows(q):", bad.any(1).nonzero().flatten().tolist()[:40], " bad cols(d):", bad.any(0).nonzero().flatten().tolist()[:40])
