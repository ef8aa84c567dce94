"""Host-side profile (cProfile) of the training step: where the Python time per step goes."""
import sys, cProfile, pstats, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
import numpy as np
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
from bidiff.training.optimizer import FlatAdamW
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
V, T, batch, L = 803, 50, 256, 128
torch.manual_seed(42)
bb = DiffusionBackbone(V, 384, 6, 3, 1536, 256, T, 0.1, 0)
model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=1, pad_token_id=0, bos_token_id=2, eos_token_id=3).to(dev).train()
model.corruption_rng = "philox"
opt = FlatAdamW(model.parameters(), lr=3e-4, weight_decay=0.01)
rs = np.random.RandomState(1)
lens = np.clip(np.round(rs.normal(38.5, 15.0, size=batch)), 12, L).astype(np.int64)
ids = rs.randint(5, V, size=(batch, L)).astype(np.int64)
attn = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
ids[attn == 0] = 0
ids_d, attn_d = torch.from_numpy(ids).to(dev), torch.from_numpy(attn).to(dev)
lens_l = lens.tolist()
def step():
    out = model(ids_d, attn_d, lengths=lens_l); out["loss"].backward(); opt.step(max_norm=1.0); opt.zero_grad()
for _ in range(5): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(30): step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"30 steps: host enqueue {t_host / 30 * 1e3:.2f} ms per step, wall {(time.perf_counter() - t0) / 30 * 1e3:.2f} ms per step")
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for _ in range(30): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
