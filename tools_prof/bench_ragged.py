"""Reverse sampling on per-sample lengths (the reference's step2 draws them from the training distribution and pads a
batch to its longest row): sequences dealt to the forward's waves by their own lengths against every padded row
computed. usage: python3 tools_prof/bench_ragged.py"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
from bidiff.sampling.sampler import ConstrainedSampler
from bidiff.utils import synthetic

dev = torch.device('cuda:0')
tok = synthetic.smiles_tokenizer()
T = 100
torch.manual_seed(42)
bb = DiffusionBackbone(tok.vocab_size, 384, 6, 3, 1536, 256, T, 0.1, tok.pad_token_id)
model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=tok.mask_token_id, pad_token_id=tok.pad_token_id,
                                 bos_token_id=tok.bos_token_id, eos_token_id=tok.eos_token_id).to(dev).eval()
smp = ConstrainedSampler(model, tok, num_steps=T, temperature=1.0, use_constraints=True, device=str(dev))
smp._core.seed = 7
rs = np.random.RandomState(0)
for B, mean in ((4096, 38), (4096, 70), (256, 38)):
    lengths = np.clip(rs.lognormal(np.log(mean), 0.45, size=B).astype(int), 8, 128).tolist()
    out = {}
    for ragged in (False, True):
        smp._core.ragged = ragged
        smp.sample_with_lengths(lengths, max_length=128, show_progress=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ids, strings = smp.sample_with_lengths(lengths, max_length=128, show_progress=False)
        torch.cuda.synchronize()
        out[ragged] = (time.perf_counter() - t0, ids)
    same = torch.equal(out[False][1], out[True][1])
    print(f"B={B} lengths mean {np.mean(lengths):.1f} max {max(lengths)}: all rows {out[False][0]:.3f} s "
          f"({B / out[False][0]:.0f} seq/s), dealt to waves {out[True][0]:.3f} s ({B / out[True][0]:.0f} seq/s), "
          f"{out[False][0] / out[True][0]:.2f}x, same ids: {same}", flush=True)
