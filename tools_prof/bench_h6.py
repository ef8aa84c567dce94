import sys, time, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff.model.backbone import DiffusionBackbone
dev = torch.device('cuda:0')
for H in (3, 6):
    torch.manual_seed(0)
    bb = DiffusionBackbone(39, 384, 6, H, 1536, 256, 100, 0.1, 0).to(dev).eval()
    B, L = 4096, 128
    ids = torch.randint(5, 39, (B, L), device=dev); t = torch.full((B,), 50, device=dev); am = torch.ones(B, L, dtype=torch.long, device=dev)
    with torch.no_grad():
        for _ in range(2): bb.get_hidden_states(ids, t, am)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): bb.get_hidden_states(ids, t, am)
        torch.cuda.synchronize()
    print(f"H={H}: forward {(time.perf_counter()-t0)/5*1e3:.2f} ms", flush=True)
