"""Headline benchmark: generated sequences/sec of Bi_Diffusion_SMILES reverse sampling.

One "step" = one pass of the hot path over one batch: a full T-step reverse sampling of
`--batch` sequences of length `--seq-len` (denoiser forward + constrained unmasking step x T,
then the post-loop repairs), small SMILES backbone (d384 / 6 layers / 3 heads / ffn 1536,
V = 39, T = 100), bf16 MFMA compute, random-init weights, synthetic vocabulary.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0. With N > 1 every rank samples its own `--batch` sequences
(weak scaling, no data-path collective; the global sequence index keys the random streams).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "discrete-diffusion-model-for-inverse-polymer-design_amd")
for _p in (PKG, ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 peak of MI355X


def fwd_flops_per_token(d, n_layers, ffn, L, V):
    return 2 * n_layers * (4 * d * d + 2 * d * ffn) + 4 * n_layers * L * d + 2 * d * V


def cpu_baseline(sd, cfg, sampler, L, T, n_seq, threads):
    """Oracle port of the same workload on the host cores, bounded sample."""
    from oracle import cpu_baseline as CB
    from oracle import sampler_ref as ora
    torch.set_num_threads(threads)
    tb = ora.Tables.from_token_tables(sampler._core.tt)
    t0 = time.perf_counter()
    _, t_net, t_step = CB.sample_smiles(sd, cfg, tb, n_seq, L, T, seed=1)
    dt = time.perf_counter() - t0
    return {"value": n_seq / dt, "unit": "sequences/s", "cores": threads, "kind": "port",
            "sample": f"{n_seq} sequences x {T} steps, L={L}, fp32 PyTorch denoiser + C sampling oracle, "
                      f"{dt:.1f} s ({t_net:.1f} s denoiser, {t_step:.1f} s sampling loop)"}


def train_bench(dev, rank, world, steps, warmup, batch=256, L=128):
    """Second headline number: non-pad tokens/s of the SMILES_BPE training step (V = 803, T = 50,
    per-rank batch 256, L = 128 padded, dropout 0.1 on, AdamW lr 3e-4 wd 0.01, clip 1.0; one gradient-bucket
    all-reduce per step when world > 1)."""
    import numpy as np
    from bidiff.model.backbone import DiffusionBackbone
    from bidiff.model.diffusion import DiscreteMaskingDiffusion
    from bidiff.training.optimizer import FlatAdamW
    from bidiff.parallel import allreduce_mean_
    V, T = 803, 50
    torch.manual_seed(42)
    bb = DiffusionBackbone(V, 384, 6, 3, 1536, 256, T, 0.1, 0)
    model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=1, pad_token_id=0, bos_token_id=2,
                                     eos_token_id=3).to(dev).train()
    model.corruption_rng = "philox"
    staged = world > 1 and dist.get_backend() == "gloo"      # rehearsal on one device: exchange through the host
    if world > 1:
        for t_ in list(model.parameters()):
            if staged:
                c_ = t_.data.cpu()
                dist.broadcast(c_, src=0)
                t_.data.copy_(c_)
            else:
                dist.broadcast(t_.data, src=0)
    opt = FlatAdamW(model.parameters(), lr=3e-4, weight_decay=0.01)
    rs = np.random.RandomState(1 + rank)
    lens = np.clip(np.round(rs.normal(38.5, 15.0, size=batch)), 12, L).astype(np.int64)
    ids = rs.randint(5, V, size=(batch, L)).astype(np.int64)
    attn = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
    ids[attn == 0] = 0
    ids[:, 0] = 2
    ids[np.arange(batch), lens - 1] = 3
    ids_d, attn_d = torch.from_numpy(ids).to(dev), torch.from_numpy(attn).to(dev)
    tokens = int(attn.sum())

    def step():
        out = model(ids_d, attn_d)
        out["loss"].backward()
        if world > 1:
            if staged:
                c_ = opt.flat_grad.cpu()
                allreduce_mean_(c_, world)
                opt.flat_grad.copy_(c_)
            else:
                allreduce_mean_(opt.flat_grad, world)
        opt.step(max_norm=1.0)
        opt.zero_grad()
        return out["loss"]

    for _ in range(warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tok = torch.tensor([float(tokens)], device=dev, dtype=torch.float64)
    tm = torch.tensor([el], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tok, op=dist.ReduceOp.SUM)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    el = float(tm.item())
    return {"metric": "train non-pad tokens/sec (Bi_Diffusion_SMILES_BPE training step)",
            "value": float(tok.item()) * steps / el, "unit": "non-pad tokens/s",
            "padded_tokens_per_s": world * batch * L * steps / el, "ms_per_step": 1e3 * el / steps,
            "steps": steps, "warmup": warmup, "final_loss": float(loss.item()),
            "config": {"workload": f"SMILES_BPE small backbone (V=803, T=50), per-GPU batch {batch}, L={L} padded, "
                                   f"{tokens} non-pad tokens per rank-batch, dropout 0.1 at all four sites (in-kernel Philox), AdamW + clip 1.0",
                       "parallelism": f"dp{world}, one flat-gradient all-reduce per step"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--diffusion-steps", type=int, default=100)
    ap.add_argument("--cpu-seqs", type=int, default=64, help="sequences in the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-constraints", action="store_true")
    ap.add_argument("--train-steps", type=int, default=20, help="timed training steps (0 = skip the training leg)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 logic on a one-GPU box: BD_BENCH_REHEARSAL=1 puts every rank on device 0 and
    # exchanges over gloo (RCCL refuses two ranks on one device); never set by the driver
    rehearsal = os.environ.get("BD_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(120, exit=True)      # a stuck rehearsal names its line
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group(backend="gloo", init_method="env://")
        else:
            dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)

    from bidiff import _lib
    from bidiff.model.backbone import DiffusionBackbone
    from bidiff.model.diffusion import DiscreteMaskingDiffusion
    from bidiff.sampling.sampler import ConstrainedSampler
    from bidiff.utils import synthetic

    tok = synthetic.smiles_tokenizer()
    T, B, L = args.diffusion_steps, args.batch, args.seq_len
    cfg = dict(vocab_size=tok.vocab_size, hidden_size=384, num_layers=6, num_heads=3, ffn_hidden_size=1536,
               max_position_embeddings=max(256, L), num_diffusion_steps=T, dropout=0.1, pad_token_id=tok.pad_token_id)
    torch.manual_seed(42)
    bb = DiffusionBackbone(**cfg)
    model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=tok.mask_token_id, pad_token_id=tok.pad_token_id,
                                     bos_token_id=tok.bos_token_id, eos_token_id=tok.eos_token_id)
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model = model.to(dev).eval()
    smp = ConstrainedSampler(model, tok, num_steps=T, temperature=1.0, use_constraints=not args.no_constraints,
                             device=str(dev))
    smp._core.seed = 20260227

    def run_batch(i):
        ids0, am, fm = smp.initial_state(B, L)
        return smp._core.run(ids0, am, fm, seq_offset=(i * world + rank) * B)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        run_batch(i)
    lib = _lib.lib()
    lib.bd_prof_enable(1)
    lib.bd_prof_read(None, None, None, 1)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ids = run_batch(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    import ctypes as C
    kinds = {0: "gemm_bf16_kernel (tiled GEMM: vocabulary head and unfused projections)",
             1: "gemm_panel_kernel (row-panel projection with fused LayerNorm)",
             2: "ffn_fused_kernel (LayerNorm + fc1 + GELU + fc2 + residual)",
             3: "attn_block_kernel (LayerNorm + QKV + attention + out-proj + residual)",
             4: "backbone_mega_kernel (whole forward: embeddings, all layers' attention + feed-forward phases, final LayerNorm)"}
    prof = {}
    for kind in kinds:
        n_k, ms_k, fl_k = C.c_longlong(0), C.c_double(0.0), C.c_double(0.0)
        lib.bd_prof_read_kind(kind, C.byref(n_k), C.byref(ms_k), C.byref(fl_k), 1)
        if n_k.value:
            prof[kind] = (n_k.value, ms_k.value, fl_k.value)
    lib.bd_prof_enable(0)
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # strings are decoded on the host outside the timed region (host tokenizer work)
    assert (ids != tok.mask_token_id).all()
    train = None
    if args.train_steps > 0:
        del ids
        torch.cuda.empty_cache()
        try:
            train = train_bench(dev, rank, world, args.train_steps, 5)
        except Exception as exc:      # the sampling line (the headline metric) is still reported
            train = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        total_seqs = world * B * args.steps
        value = total_seqs / elapsed
        flops_seq = fwd_flops_per_token(384, 6, 1536, L, tok.vocab_size) * L * T
        # roofline of the kernel the timed region spends most of its time in (HIP events recorded by the
        # library on the launch stream around every launch of that kernel)
        dom = max(prof, key=lambda k: prof[k][1])
        n_dom, ms_dom, fl_dom = prof[dom]
        dom_tflops = fl_dom / (ms_dom * 1e-3) / 1e12
        step_ms = 1e3 * elapsed
        # HBM-side bytes per launch of that kernel: PMC counters cannot be read from inside this process, so
        # the figure comes from the committed rocprofv3 --pmc passes (profiles/r1_pmc_traffic.json)
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_pmc_traffic.json")) as fh:
                tj = json.load(fh)
            ent = tj["kernels"].get(kinds[dom].split(" ")[0])
            if ent and B * L == 4096 * 128:
                traffic, traffic_src = ent["traffic_bytes"], tj["source"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "generated sequences/sec (Bi_Diffusion_SMILES reverse sampling, 100 steps, L=128)",
            "value": value, "unit": "sequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init weights, synthetic 39-token vocabulary)",
            "config": {"workload": f"Bi_Diffusion_SMILES small backbone (d384, 6 layers, 3 heads, ffn 1536, V=39), "
                                   f"batch {B} per GPU, L={L}, T={T} reverse steps, constraints "
                                   f"{'off' if args.no_constraints else 'on'}, logits at chosen rows only (t>1), "
                                   f"full logits at t=1, on-device repairs",
                       "batch_per_gpu": B, "seq_len": L, "diffusion_steps": T, "sharding": f"{world} x independent batches"},
            "roofline": {"bound": "mfma", "kernel": kinds[dom],
                         "achieved": dom_tflops, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": dom_tflops / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "launches": n_dom, "avg_launch_ms": ms_dom / n_dom, "flops_per_launch": fl_dom / n_dom,
                         "share_of_timed_region": ms_dom / step_ms},
            "kernels": {kinds[k].split(" ")[0]: {"launches": v[0], "avg_launch_ms": v[1] / v[0],
                                                 "tflops": v[2] / (v[1] * 1e-3) / 1e12, "share_of_timed_region": v[1] / step_ms}
                        for k, v in prof.items()},
            "whole_path_tflops": value / world * flops_seq / 1e12,
        }
        if train is not None:
            out["train"] = train
        if args.cpu_seqs > 0 and world == 1:
            threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
            out["cpu_baseline"] = cpu_baseline(sd_cpu, cfg, smp, L, T, args.cpu_seqs, threads)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
