"""C5 stress configuration (24 layers, d=1024, F=4096, L=512) sampling on one GPU: time per denoiser forward
and a short reverse loop."""
import sys, time, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
from bidiff.sampling.sampler import ConstrainedSampler
from bidiff.utils import synthetic
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
tok = synthetic.smiles_tokenizer()
H = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = int(sys.argv[3]) if len(sys.argv) > 3 else 10
L = 512
torch.manual_seed(42)
bb = DiffusionBackbone(tok.vocab_size, 1024, 24, H, 4096, 512, 100, 0.1, tok.pad_token_id)
import os
bb.defer_layernorm = os.environ.get('BD_DEFER') == '1'      # A/B of the deferred LayerNorm
model = DiscreteMaskingDiffusion(bb, num_steps=100, mask_token_id=tok.mask_token_id, pad_token_id=tok.pad_token_id,
                                 bos_token_id=tok.bos_token_id, eos_token_id=tok.eos_token_id).to(dev).eval()
ids = torch.randint(5, tok.vocab_size, (B, L), device=dev)
t = torch.full((B,), 50, device=dev, dtype=torch.long)
am = torch.ones(B, L, dtype=torch.long, device=dev)
with torch.no_grad():
    for _ in range(2): bb.get_hidden_states(ids, t, am)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): bb.get_hidden_states(ids, t, am)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
fl = B * L * 654.39e6
print(f"C5 H={H} B={B} L={L}: forward {dt*1e3:.1f} ms  {fl/dt/1e12:.1f} TFLOP/s  ({B/dt/100:.2f} seq/s at T=100)", flush=True)
smp = ConstrainedSampler(model, tok, num_steps=100, temperature=1.0, use_constraints=True, device=str(dev))

ids0, am0, fm0 = smp.initial_state(B, L)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = smp._core.run(ids0, am0, fm0, seq_offset=0)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"C5 full reverse loop T=100: {dt:.2f} s per batch of {B} -> {B/dt:.2f} seq/s, {B*L*654.39e6*100/dt/1e12:.1f} TFLOP/s", flush=True)
