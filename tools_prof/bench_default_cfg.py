"""Forward time of the reference's default backbone (d384, 8 layers, 8 heads of 48, ffn 1536) against the small preset."""
import sys, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff.model.backbone import DiffusionBackbone
dev = torch.device('cuda:0')
B, L = 4096, 128
for name, layers, heads in (("small preset 6L/3H", 6, 3), ("default 8L/8H", 8, 8), ("6 heads 8L/6H", 8, 6)):
    torch.manual_seed(42)
    bb = DiffusionBackbone(39, 384, layers, heads, 1536, 256, 100, 0.0, 0).to(dev).eval()
    ids = torch.randint(5, 39, (B, L), device=dev); t = torch.full((B,), 50, device=dev)
    with torch.no_grad():
        for _ in range(3):
            bb.run_forward(ids, t, None, want_hidden=True, want_logits=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            bb.run_forward(ids, t, None, want_hidden=True, want_logits=False)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"{name}: {ms:.2f} ms per forward, {ms / layers:.2f} ms per layer", flush=True)
