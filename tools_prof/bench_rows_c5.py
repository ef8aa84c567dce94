"""Stress model (24 layers, d 1024, L 512): full forward against the chosen-row forward, per launch."""
import sys, time, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
import bench
from bidiff import _lib
from bidiff.model.backbone import DiffusionBackbone
dev = torch.device('cuda:0'); lib = _lib.lib()
B, L, cap = 128, 512, 8
torch.manual_seed(1)
bb = DiffusionBackbone(39, 1024, 24, 8, 4096, 512, 100, 0.0, 0).to(dev).eval()
print("mode", bb.rows_forward_mode(B, L))
ids = torch.randint(5, 39, (B, L), device=dev); ts = torch.full((B,), 50, device=dev, dtype=torch.long)
km = torch.ones(B, L, dtype=torch.uint8, device=dev)
hid = torch.empty(B, L, 1024, dtype=torch.bfloat16, device=dev)
rp = torch.randint(0, L, (B * cap,), device=dev, dtype=torch.int32); na = torch.zeros(B, dtype=torch.uint8, device=dev)
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    print("full  %.2f ms" % t(lambda: bb.run_forward(ids, ts, km, want_hidden=True, hidden_out=hid)))
    print("rows  %.2f ms" % t(lambda: bb.run_forward_rows(ids, ts, km, rp, na, cap, hid)))
    bench.prof_reset(lib); t0 = time.perf_counter()
    for _ in range(2): bb.run_forward_rows(ids, ts, km, rp, na, cap, hid)
    torch.cuda.synchronize()
    table = bench.prof_table(lib, 1e3 * (time.perf_counter() - t0))
    for k, v in bench.named(table).items():
        print(f"    {k}: {v['launches']} launches, {v['avg_launch_ms']*1e3:.0f} us avg, total {v['total_ms']:.2f} ms")
