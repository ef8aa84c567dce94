"""Training step of the reference's default `backbone:` block (d = 384, 8 layers, 8 heads of 48, run as 8 zero-padded
heads of 64) against the 3-head small size, same batch as the bench's training leg."""
import os, sys, time, numpy as np, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, 'discrete-diffusion-model-for-inverse-polymer-design_amd')); sys.path.insert(0, _ROOT)
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
from bidiff.training.optimizer import FlatAdamW
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
V, T, batch, L = 803, 50, 256, 128
rs = np.random.RandomState(1)
lens = np.clip(np.round(rs.normal(38.5, 15.0, size=batch)), 12, L).astype(np.int64)
ids = rs.randint(5, V, size=(batch, L)).astype(np.int64)
attn = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
ids[attn == 0] = 0
ids_d, attn_d = torch.from_numpy(ids).to(dev), torch.from_numpy(attn).to(dev)
for name, layers, heads in (("small (6 layers, 3 heads of 128)", 6, 3), ("default block (8 layers, 8 heads of 48)", 8, 8), ("8 layers, 6 heads of 64", 8, 6)):
    torch.manual_seed(42)
    bb = DiffusionBackbone(V, 384, layers, heads, 1536, 256, T, 0.1, 0)
    model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=1, pad_token_id=0, bos_token_id=2, eos_token_id=3).to(dev).train()
    model.corruption_rng = "philox"
    opt = FlatAdamW(model.parameters(), lr=3e-4, weight_decay=0.01)
    def step():
        out = model(ids_d, attn_d, lengths=lens.tolist()); out["loss"].backward(); opt.step(max_norm=1.0); opt.zero_grad()
        return out["loss"]
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): loss = step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {dt*1e3:.2f} ms per step (host enqueue {host/20*1e3:.2f} ms), {int(attn.sum())/dt/1e6:.2f} M non-pad tokens/s, loss {float(loss):.3f}", flush=True)
    del model, opt, bb
    torch.cuda.empty_cache()
