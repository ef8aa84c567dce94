"""Training step (packed rows, dropout 0.1, AdamW + clip) at the reference's medium / large / xl model sizes, SMILES_BPE
vocabulary, per-GPU batch 128 x 128 (xl: 64 x 128, its config's batch): time per step, non-pad tokens per second, first and last loss of twelve steps."""
import sys, os, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "discrete-diffusion-model-for-inverse-polymer-design_amd")); sys.path.insert(0, ROOT)
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
from bidiff.training.optimizer import FlatAdamW
import bench
dev = torch.device("cuda:0")
for name, (d, n, h, f, Bt) in {"medium": (768, 12, 6, 3072, 128), "large": (1152, 18, 9, 4608, 128), "xl": (1536, 24, 12, 6144, 64)}.items():
    if os.environ.get("BD_TRAIN_SIZES_ONLY") not in (None, "", name):
        continue
    torch.manual_seed(0)
    bb = DiffusionBackbone(803, d, n, h, f, 256, 50, 0.1, 0)
    model = DiscreteMaskingDiffusion(bb, num_steps=50, mask_token_id=1, pad_token_id=0, bos_token_id=2, eos_token_id=3).to(dev).train()
    model.corruption_rng = "philox"
    opt = FlatAdamW(model.parameters(), lr=3e-4, weight_decay=0.01)
    ids, attn = bench.bpe_batch(Bt, 128, 803, 1)
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
    rows = (int(attn.sum()) + 127) // 128 * 128
    fl = 3.0 * ((2 * n * (4 * d * d + 2 * d * f) + 2 * d * 803) * rows + 4 * n * d * float(sum(x * x for x in lens)))
    print(name, f"batch {Bt} x 128, {rows} packed rows: {dt*1e3:.2f} ms/step", f"{int(attn.sum())/dt:.0f} tok/s",
          f"{fl/dt/1e12:.0f} TFLOP/s ({fl/dt/1e12/2500:.3f} of the bf16 peak)", "loss", round(ls[0], 3), round(ls[-1], 3), flush=True)
    if os.environ.get("BD_TRAIN_SIZES_TABLE") == "1":      # per-kernel table of three more steps (library events)
        from bidiff import _lib
        lib = _lib.lib()
        bench.prof_reset(lib); t0 = time.perf_counter()
        for i in range(3):
            out = model(ids_d, attn_d, lengths=lens); out["loss"].backward(); opt.step(max_norm=1.0); opt.zero_grad()
        torch.cuda.synchronize()
        table = bench.prof_table(lib, 1e3 * (time.perf_counter() - t0))
        for k, v in bench.named(table).items():
            print(f"    {k}: {v['launches'] / 3:.0f} launches per step, {v['avg_launch_ms']*1e3:.0f} us avg, share {v['share_of_timed_region']:.2f}, "
                  f"{v.get('tflops', v.get('gbps', 0)):.0f} {'TFLOP/s' if 'tflops' in v else 'GB/s'}", flush=True)
    del model, opt, bb
    torch.cuda.empty_cache()
