"""The forward of a reverse step at t > 1 as the samplers run it (rows chosen before the forward, the last feed-forward
block and final LayerNorm for the chosen rows only): three steps' worth of choose + forward on a half-masked batch of
the benchmark shape, for rocprofv3 --pmc / --kernel-trace."""
import ctypes as C
import sys
import torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion
from bidiff.sampling.sampler import ConstrainedSampler
from bidiff.sampling import core
from bidiff.utils import synthetic

dev = torch.device('cuda:0')
tok = synthetic.smiles_tokenizer()
T, B, L = 100, 4096, 128
torch.manual_seed(42)
bb = DiffusionBackbone(tok.vocab_size, 384, 6, 3, 1536, 256, T, 0.0, tok.pad_token_id)
model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=tok.mask_token_id, pad_token_id=tok.pad_token_id,
                                 bos_token_id=tok.bos_token_id, eos_token_id=tok.eos_token_id).to(dev).eval()
smp = ConstrainedSampler(model, tok, num_steps=T, temperature=1.0, use_constraints=True, device=str(dev))
c = smp._core
ids, am, fm = smp.initial_state(B, L)
ids[:, 1:L - 1:2] = torch.randint(5, tok.vocab_size, (B, (L - 2 + 1) // 2), device=dev)     # every second position drawn
fixed = fm.to(torch.uint8).contiguous()
keymask = (am != 0).to(torch.uint8).contiguous()
a = _lib.StepArgs()
core.fill_step_args(a, c.tt, c._tables())
ktab = c._ktable(L)
t = 50
a.ids, a.fixed, a.ktable = ids.data_ptr(), fixed.data_ptr(), ktab[t].data_ptr()
a.B, a.L, a.d, a.variant, a.use_constraints, a.t = B, L, 384, c.tt.variant, 1, t
a.temperature, a.seed, a.seq_offset = 1.0, 7, 0
cap = 2
row_pos = torch.empty(B * cap, dtype=torch.int32, device=dev)
need_all = torch.empty(B, dtype=torch.uint8, device=dev)
hidden = torch.empty(B, L, 384, dtype=torch.bfloat16, device=dev)
ts = torch.full((B,), t, device=dev, dtype=torch.long)
lib = _lib.lib()
with torch.no_grad():
    for _ in range(3):
        _lib.check(lib.bd_sample_choose_rows(C.byref(a), row_pos.data_ptr(), need_all.data_ptr(), cap, _lib.stream_handle()), "choose")
        bb.run_forward_rows(ids, ts, keymask, row_pos, need_all, cap, hidden)
torch.cuda.synchronize()
print("need_all", int(need_all.sum().item()), "rows", int((row_pos >= 0).sum().item()))
