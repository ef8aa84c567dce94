"""Single-launch forward on short sequences: two (L <= 64) or four (L <= 32) sequences per workgroup against one
(BD_MEGA_PACK=0). usage: python3 tools_prof/bench_short_seq.py"""
import os, sys, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff.model.backbone import DiffusionBackbone
dev = torch.device('cuda:0')
torch.manual_seed(42)
bb = DiffusionBackbone(39, 384, 6, 3, 1536, 256, 100, 0.0, 0).to(dev).eval()
B = 4096
for L in (128, 96, 64, 48, 32, 16):
    ids = torch.randint(5, 39, (B, L), device=dev); t = torch.full((B,), 50, device=dev)
    res = {}
    for pack in ("0", "1"):
        os.environ["BD_MEGA_PACK"] = pack
        with torch.no_grad():
            for _ in range(3):
                out = bb.run_forward(ids, t, None, want_hidden=True, want_logits=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                out = bb.run_forward(ids, t, None, want_hidden=True, want_logits=False)
            e1.record(); torch.cuda.synchronize()
        h = out["hidden"] if isinstance(out, dict) else out
        h = h[0] if isinstance(h, (tuple, list)) else h
        res[pack] = (e0.elapsed_time(e1) / 50, h.float().abs().mean().item())
    print(f"L={L:4d}: one sequence per workgroup {res['0'][0]:.3f} ms, shared {res['1'][0]:.3f} ms "
          f"({res['0'][0] / res['1'][0]:.2f}x), checksums {res['0'][1]:.6f} {res['1'][1]:.6f}", flush=True)
