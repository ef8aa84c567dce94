"""Training step (packed rows, dropout 0.1, AdamW + clip) at the reference's medium / large model sizes, SMILES_BPE vocabulary,
per-GPU batch 128 x 128: time per step, non-pad tokens per second, first and last loss of twelve steps."""
import sys, os, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "discrete-diffusion-model-for-inverse-polymer-design_amd")); sys.path.insert(0, ROOT)
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
from bidiff.training.optimizer import FlatAdamW
import bench
dev = torch.device("cuda:0")
for name, (d, n, h, f) in {"medium": (768, 12, 6, 3072), "large": (1152, 18, 9, 4608)}.items():
    torch.manual_seed(0)
    bb = DiffusionBackbone(803, d, n, h, f, 256, 50, 0.1, 0)
    model = DiscreteMaskingDiffusion(bb, num_steps=50, mask_token_id=1, pad_token_id=0, bos_token_id=2, eos_token_id=3).to(dev).train()
    model.corruption_rng = "philox"
    opt = FlatAdamW(model.parameters(), lr=3e-4, weight_decay=0.01)
    ids, attn = bench.bpe_batch(128, 128, 803, 1)
    ids_d, attn_d = torch.from_numpy(ids).to(dev), torch.from_numpy(attn).to(dev)
    lens = attn.sum(1).tolist()
    losses = []
    for i in range(12):
        if i == 4:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        out = model(ids_d, attn_d, lengths=lens); out["loss"].backward(); opt.step(max_norm=1.0); opt.zero_grad()
        losses.append(out["loss"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    ls = [float(x) for x in losses]
    print(name, f"{dt*1e3:.2f} ms/step", f"{int(attn.sum())/dt:.0f} tok/s", "loss", ls[0], ls[-1], flush=True)
    del model, opt, bb
    torch.cuda.empty_cache()
