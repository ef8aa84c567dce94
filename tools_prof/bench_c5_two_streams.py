"""Stress configuration (24 layers, d=1024, F=4096, L=512): one forward of B sequences against two forwards of B/2 on
two streams at once (two model instances: each has its own workspace). With BD_GEMM8P_SLOTS=128 every GEMM launch of
the two-stream run takes half the CUs."""
import sys, time, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff.model.backbone import DiffusionBackbone
from bidiff.utils import synthetic
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
tok = synthetic.smiles_tokenizer()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mode = sys.argv[2] if len(sys.argv) > 2 else "one"
L = 512
def make():
    torch.manual_seed(42)
    return DiffusionBackbone(tok.vocab_size, 1024, 24, 8, 4096, 512, 100, 0.1, tok.pad_token_id).to(dev).eval()
ids = torch.randint(5, tok.vocab_size, (B, L), device=dev)
t = torch.full((B,), 50, device=dev, dtype=torch.long)
am = torch.ones(B, L, dtype=torch.long, device=dev)
n = 5
with torch.no_grad():
    if mode == "one":
        bb = make()
        for _ in range(2): bb.get_hidden_states(ids, t, am)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): bb.get_hidden_states(ids, t, am)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    else:
        bbs = [make(), make()]
        ss = [torch.cuda.Stream(), torch.cuda.Stream()]
        h = B // 2
        parts = [(ids[:h].contiguous(), t[:h].contiguous(), am[:h].contiguous()), (ids[h:].contiguous(), t[h:].contiguous(), am[h:].contiguous())]
        def both():
            for k in range(2):
                with torch.cuda.stream(ss[k]):
                    bbs[k].get_hidden_states(*parts[k])
        for _ in range(2): both()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): both()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
fl = B * L * 654.39e6
print(f"C5 B={B} mode={mode}: {dt*1e3:.1f} ms per {B} sequences  {fl/dt/1e12:.1f} TFLOP/s", flush=True)
