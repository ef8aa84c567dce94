"""sample_variable_length / SELFIES sample_batch(lengths=...) as the reference's step2 scripts call them: groups of
different lengths sharing reverse loops against one loop per group. usage: python3 tools_prof/bench_variable_length.py"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
from bidiff.utils import synthetic

dev = torch.device('cuda:0')
T = 100


def make(variant):
    if variant == "smiles":
        from bidiff.sampling.sampler import ConstrainedSampler
        tok = synthetic.smiles_tokenizer()
    else:
        from bidiff.sampling.selfies_sampler import ConstrainedSampler
        tok = synthetic.selfies_tokenizer()
    torch.manual_seed(42)
    bb = DiffusionBackbone(tok.vocab_size, 384, 6, 3, 1536, 256, T, 0.1, tok.pad_token_id)
    model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=tok.mask_token_id, pad_token_id=tok.pad_token_id,
                                     bos_token_id=tok.bos_token_id, eos_token_id=tok.eos_token_id).to(dev).eval()
    smp = ConstrainedSampler(model, tok, num_steps=T, temperature=1.0, use_constraints=True, device=str(dev))
    smp._core.seed = 7
    return smp


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


smp = make("smiles")
res = {}
for merged in (False, True):
    smp.merge_variable_length = merged
    def fn():
        np.random.seed(3)
        return smp.sample_variable_length(1000, length_range=(20, 100), batch_size=256, samples_per_length=16,
                                          show_progress=False)
    res[merged] = timed(fn)
print(f"SMILES sample_variable_length(1000): loop per group {res[False][0]:.2f} s, shared loops {res[True][0]:.2f} s "
      f"({res[False][0] / res[True][0]:.1f}x), same strings: {res[False][1][1] == res[True][1][1]}", flush=True)
smp = make("selfies")
lengths = np.clip(np.random.RandomState(0).lognormal(np.log(38), 0.45, size=1000).astype(int), 8, 128).tolist()
for merged in (False, True):
    smp.merge_variable_length = merged
    res[merged] = timed(lambda: smp.sample_batch(1000, 128, batch_size=256, show_progress=False, lengths=lengths))
print(f"SELFIES sample_batch(1000, lengths around 38): loop per length {res[False][0]:.2f} s, shared loops "
      f"{res[True][0]:.2f} s ({res[False][0] / res[True][0]:.1f}x), same strings: {res[False][1][1] == res[True][1][1]}",
      flush=True)
