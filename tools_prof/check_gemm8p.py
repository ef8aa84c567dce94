"""Exact-integer screen of the GEMM kernels (run with BD_GEMM8P=2 to force the eight-wave kernel on every shape):
small-integer operands make every fp32 sum exact, so any staging / fragment / synchronisation mistake is a hard
mismatch. Each shape runs several times on fresh data to catch races."""
import sys, torch
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
dev = torch.device('cuda:0'); lib = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(1)
bad = 0
shapes = [(256, 256, 128), (256, 256, 192), (200, 256, 128), (1000, 384, 256), (513, 640, 768), (4096, 1024, 1024),
          (777, 1152, 768), (2048, 3072, 1024), (3000, 1024, 4096), (65536, 1024, 1024), (8192, 4096, 1024),
          (255, 128, 4096), (257, 128, 64 * 7), (65536, 3072, 1024)]
for (M, N, K) in shapes:
    reps = 2 if M * N * K > 5e10 else 6
    for rep in range(reps):
        A = torch.randint(-3, 4, (M, K), device=dev, generator=g).float()
        W = torch.randint(-2, 3, (N, K), device=dev, generator=g).float()
        bias = torch.randint(-4, 5, (N,), device=dev, generator=g).float()
        ref = A @ W.t() + bias
        Ab, Wb = A.bfloat16(), W.bfloat16()
        for epi, dt in ((3, torch.float32), (0, torch.bfloat16)):
            out = torch.full((M, N), float('nan'), device=dev, dtype=dt)
            rc = lib.bd_gemm_bf16(_lib.ptr(Ab), _lib.ptr(Wb), _lib.ptr(bias), _lib.ptr(out), M, N, N, K,
                                  N, epi, _lib.stream_handle())
            assert rc == 0, lib.bd_last_error()
            torch.cuda.synchronize()
            want = ref.to(dt)
            ok = torch.equal(out, want)
            if not ok:
                bad += 1
                d = (out.float() - want.float()).abs()
                nz = torch.nonzero(d > 0)
                print(f"MISMATCH M={M} N={N} K={K} epi={epi} rep={rep}: {nz.shape[0]} elements, first {nz[:4].tolist()}, "
                      f"rows {sorted(set((nz[:, 0] % 256).tolist()))[:16]} cols {sorted(set((nz[:, 1] % 256).tolist()))[:16]}",
                      flush=True)
                break
        if bad:
            break
    print(f"M={M} N={N} K={K}: {'ok' if not bad else 'FAILED'}", flush=True)
    if bad:
        break
print("ALL OK" if not bad else "FAILED")
sys.exit(1 if bad else 0)
