import sys, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from tests.golden import replay_inputs as RI
from bidiff.model.backbone import DiffusionBackbone
from bidiff.training.optimizer_setup import build_step1_backbone_optimizer
dev = torch.device('cuda:0')
cfg = RI.BACKBONE_CFGS["tiny_h2"]
torch.manual_seed(7)
bb = DiffusionBackbone(**cfg).to(dev)
opt, _ = build_step1_backbone_optimizer(bb, RI.MUON_OPT_CFG, 3e-4, 0.01)
for step in range(3):
    grads = RI.optimizer_grads(bb, step)
    for n, p in bb.named_parameters(): p.grad.copy_(torch.from_numpy(grads[n]))
    opt.step(max_norm=1.0); opt.zero_grad()
sd = opt.state_dict()
torch.save({"opt": sd, "model": bb.state_dict()}, "/tmp/o.pt")
ck = torch.load("/tmp/o.pt", weights_only=True)
bb2 = DiffusionBackbone(**cfg).to(dev)
bb2.load_state_dict(ck["model"])
opt2, _ = build_step1_backbone_optimizer(bb2, RI.MUON_OPT_CFG, 3e-4, 0.01)
opt2.load_state_dict(ck["opt"])
print("step", opt._step, opt2._step, [ (g['lr'], g.get('betas')) for g in opt.param_groups], [(g['lr'], g.get('betas')) for g in opt2.param_groups])
print("m diff", (opt.flat_m - opt2.flat_m).abs().max().item(), "v diff", (opt.flat_v - opt2.flat_v).abs().max().item(), "p diff", (opt.flat_param-opt2.flat_param).abs().max().item())
grads = RI.optimizer_grads(bb, 7)
for m, o in ((bb, opt), (bb2, opt2)):
    for n, p in m.named_parameters(): p.grad.copy_(torch.from_numpy(grads[n]))
    print("gdiff setup", o.flat_grad.abs().sum().item())
    o.step(max_norm=1.0)
for (n, p), (_, q) in zip(bb.named_parameters(), bb2.named_parameters()):
    d = (p - q).abs().max().item()
    if d: print(n, d)
