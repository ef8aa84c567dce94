import sys, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff.model.backbone import DiffusionBackbone
dev = torch.device('cuda:0')
torch.manual_seed(42)
bb = DiffusionBackbone(39, 384, 6, 3, 1536, 256, 100, 0.0, 0).to(dev).eval()
B, L = 4096, 128
ids = torch.randint(5, 39, (B, L), device=dev); t = torch.full((B,), 50, device=dev)
with torch.no_grad():
    for _ in range(3):
        bb.run_forward(ids, t, None, want_hidden=True, want_logits=False)
torch.cuda.synchronize()
