"""Sampling throughput of the four tokenisations with the small backbone (B=4096, L=128, T=100)."""
import sys, time, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from tests import helpers as H
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
dev = torch.device('cuda:0')
B, L, T = 4096, 128, 100
for variant in ("smiles", "bpe", "selfies", "group"):
    _, tok = H.make_sampler(variant)
    torch.manual_seed(42)
    bb = DiffusionBackbone(tok.vocab_size, 384, 6, 3, 1536, 256, T, 0.0, tok.pad_token_id)
    model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=tok.mask_token_id, pad_token_id=tok.pad_token_id,
                                     bos_token_id=tok.bos_token_id, eos_token_id=tok.eos_token_id).to(dev).eval()
    smp, _ = H.make_sampler(variant, model=model, num_steps=T, device=str(dev))
    smp._core.seed = 1
    smp.sample(B, L, show_progress=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(2): smp.sample(B, L, show_progress=False, seq_offset=(i + 1) * B)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 2
    print(f"{variant:8s} V={tok.vocab_size:4d}: {B / el:7.0f} sequences/s ({el * 1e3:.0f} ms per batch)", flush=True)
