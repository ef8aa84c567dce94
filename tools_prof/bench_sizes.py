"""Denoiser forward of the reference's medium / large model sizes (configs/config.yaml model_sizes) at the sampling
shape B x 128: time per forward, whole-path TFLOP/s and the library's per-kernel table."""
import os, sys, time, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, 'discrete-diffusion-model-for-inverse-polymer-design_amd')); sys.path.insert(0, _ROOT)
import bench
from bidiff import _lib
from bidiff.model.backbone import DiffusionBackbone
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
lib = _lib.lib()
SIZES = {"medium": (768, 12, 6, 3072), "large": (1152, 18, 9, 4608), "xl": (1536, 24, 12, 6144)}
B, L, V = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 128, 39
for name, (d, n, h, f) in SIZES.items():
    torch.manual_seed(1)
    bb = DiffusionBackbone(V, d, n, h, f, 256, 100, 0.1, 0).to(dev).eval()
    ids = torch.randint(5, V, (B, L), device=dev); t = torch.full((B,), 50, device=dev); am = torch.ones(B, L, dtype=torch.long, device=dev)
    with torch.no_grad():
        for _ in range(2): bb.get_hidden_states(ids, t, am)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): bb.get_hidden_states(ids, t, am)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        bench.prof_reset(lib); t0 = time.perf_counter()
        for _ in range(2): bb.get_hidden_states(ids, t, am)
        torch.cuda.synchronize()
        table = bench.prof_table(lib, 1e3 * (time.perf_counter() - t0))
    fl = bench.fwd_flops_per_token(d, n, f, L, V) * B * L
    print(f"{name} d={d} layers={n} heads={h} B={B} L={L}: forward {dt*1e3:.1f} ms, {fl/dt/1e12:.0f} TFLOP/s whole path", flush=True)
    for k, v in bench.named(table).items():
        print(f"    {k}: {v['launches']} launches, {v['avg_launch_ms']*1e3:.0f} us avg, share {v['share_of_timed_region']:.2f}, "
              f"{v.get('tflops', v.get('gbps', 0)):.0f} {'TFLOP/s' if 'tflops' in v else 'GB/s'}", flush=True)
    del bb
    torch.cuda.empty_cache()
