"""The SMILES_BPE training step of bench.py alone (for rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, 'discrete-diffusion-model-for-inverse-polymer-design_amd')); sys.path.insert(0, _ROOT)
import bench
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
r = bench.train_bench(dev, 0, 1, 40, 5, 0, 16)
print({k: r[k] for k in ("value", "ms_per_step", "rows_computed_per_step", "all_rows_variant")})
