"""A/B of the single-launch forward between library builds: python3 tools_prof/ab_mega.py [path/to/lib.so]"""
import sys, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
if len(sys.argv) > 1 and sys.argv[1] != "base":
    _lib.LIB_PATH = sys.argv[1]
    import ctypes
    _h = ctypes.CDLL(sys.argv[1])          # an older build lacks the newer entry points: bind what it has
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(_h, k)}
from bidiff.model.backbone import DiffusionBackbone
dev = torch.device('cuda:0')
torch.manual_seed(42)
bb = DiffusionBackbone(39, 384, 6, 3, 1536, 256, 100, 0.0, 0).to(dev).eval()
B, L = 4096, 128
ids = torch.randint(5, 39, (B, L), device=dev); t = torch.full((B,), 50, device=dev)
with torch.no_grad():
    for _ in range(3):
        out = bb.run_forward(ids, t, None, want_hidden=True, want_logits=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        out = bb.run_forward(ids, t, None, want_hidden=True, want_logits=False)
    e1.record(); torch.cuda.synchronize()
h = out["hidden"] if isinstance(out, dict) else out
h = h[0] if isinstance(h, (tuple, list)) else h
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'base'}: {e0.elapsed_time(e1) / 300:.3f} ms per forward, checksum {h.float().abs().mean().item():.6f}", flush=True)
# the same launch on all-zero weights: cycle count unchanged, switching power far lower -> shows how much the clock
# the chip holds under this kernel's load costs
with torch.no_grad():
    for p_ in bb.parameters(): p_.zero_()
    bb.invalidate_cache() if hasattr(bb, "invalidate_cache") else None
    for _ in range(3):
        out = bb.run_forward(ids, t, None, want_hidden=True, want_logits=False)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(300):
        out = bb.run_forward(ids, t, None, want_hidden=True, want_logits=False)
    e1.record(); torch.cuda.synchronize()
print(f"  zero weights: {e0.elapsed_time(e1) / 300:.3f} ms per forward", flush=True)
