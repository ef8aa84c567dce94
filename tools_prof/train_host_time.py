"""Training step of the bench: how long the HOST needs to enqueue a step (loop return before the device has caught up)
beside the step time with the device drained, and the launches per step."""
import sys, time, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
import bench
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
from bidiff.training.optimizer import FlatAdamW
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
V, T, batch, L = 803, 50, 256, 128
cfg = dict(vocab_size=V, hidden_size=384, num_layers=6, num_heads=3, ffn_hidden_size=1536,
           max_position_embeddings=256, num_diffusion_steps=T, dropout=0.1, pad_token_id=0)
torch.manual_seed(42)
bb = DiffusionBackbone(**cfg)
model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=1, pad_token_id=0, bos_token_id=2, eos_token_id=3).to(dev).train()
model.corruption_rng = "philox"
opt = FlatAdamW(model.parameters(), lr=3e-4, weight_decay=0.01)
ids, attn = bench.bpe_batch(batch, L, V, 1)
ids_d, attn_d = torch.from_numpy(ids).to(dev), torch.from_numpy(attn).to(dev)
lens = attn.sum(1).tolist()
def step():
    out = model(ids_d, attn_d, lengths=lens)
    out["loss"].backward()
    opt.step(max_norm=1.0)
    opt.zero_grad()
for _ in range(10): step()
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n): step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {1e3*t_host/n:.3f} ms per step, step {1e3*t_all/n:.3f} ms (device drained), device behind the host at loop end: {1e3*(t_all-t_host):.2f} ms")
# host alone: the same loop with the device idle in between (sync after every step): enqueue time without back-pressure
ts = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
print(f"host enqueue with an idle device: median {1e3*sorted(ts)[len(ts)//2]:.3f} ms")
import cProfile, pstats, io
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(20): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
