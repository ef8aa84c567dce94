import sys, time, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from tests.golden import replay_inputs as RI
from bidiff.model.backbone import DiffusionBackbone
from bidiff.training.optimizer_setup import build_step1_backbone_optimizer
dev = torch.device('cuda:0')
torch.manual_seed(0)
bb = DiffusionBackbone(803, 384, 6, 3, 1536, 256, 50, 0.1, 0).to(dev)
for name, cfg in (("adamw", {}), ("muon_adamw", RI.MUON_OPT_CFG)):
    opt, info = build_step1_backbone_optimizer(bb, cfg, 3e-4, 0.01)
    opt.flat_grad.normal_(0, 0.01)
    for _ in range(3): opt.step(max_norm=1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): opt.step(max_norm=1.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name}: optimizer step {dt*1e3:.3f} ms  info={ {k: v for k, v in info.items() if 'num' in k} }", flush=True)
