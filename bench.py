"""Headline benchmark: generated sequences/sec of Bi_Diffusion_SMILES reverse sampling.

One "step" = one pass of the hot path over one batch: a full T-step reverse sampling of
`--batch` sequences of length `--seq-len` (denoiser forward + constrained unmasking step x T,
then the post-loop repairs), small SMILES backbone (d384 / 6 layers / 3 heads / ffn 1536,
V = 39, T = 100), bf16 MFMA compute, random-init weights, synthetic vocabulary.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0. With N > 1 every rank samples its own `--batch` sequences
(weak scaling, no data-path collective; the global sequence index keys the random streams).
Besides the headline it carries: `train` (the second half of the metric: the SMILES_BPE training
step, with its own roofline, per-kernel table and CPU baseline), `configs` (the other sampling
configurations: SELFIES, Group SELFIES and the 24-layer d = 1024, L = 512 stress backbone),
`roofline` and `cpu_baseline` of the headline.
"""
import argparse
import ctypes as C
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
HBM_PEAK_GBPS = 8000.0

# kernel families of the library's event profiler (bd_prof_read_kind): name, unit of the work figure
KINDS = {0: ("gemm_bf16_kernel (tiled GEMM family, gemm8p_kernel on large shapes: vocabulary head, projections, data gradients)", "flop"),
         1: ("gemm_panel_kernel (row-panel projection with fused LayerNorm)", "flop"),
         2: ("ffn_fused_kernel (LayerNorm + fc1 + GELU + fc2 + residual)", "flop"),
         3: ("attn_block_kernel (LayerNorm + QKV + attention + out-proj + residual)", "flop"),
         4: ("backbone_mega_kernel (whole forward: embeddings, all layers' attention + feed-forward phases, final LayerNorm)", "flop"),
         5: ("wgrad_dma_kernel (weight-gradient GEMMs with bias sums)", "flop"),
         6: ("attn_fwd_kernel (stand-alone attention forward)", "flop"),
         7: ("attn_bwd_dq_kernel + attn_bwd_dkv_kernel (attention backward)", "flop"),
         8: ("layernorm_kernel / layernorm_bwd_kernel", "byte"),
         9: ("ce_fwd_kernel / ce_bwd_kernel (masked cross-entropy)", "byte"),
         10: ("embed / embed_bwd / corrupt / dropout / cast kernels", "byte"),
         11: ("sumsq / adamw / multi_cast kernels (clip, AdamW, bf16 weight refresh)", "byte"),
         12: ("sample_step_kernel / fixup_kernel (reverse step and repairs)", "byte")}


def fwd_flops_per_token(d, n_layers, ffn, L, V):
    return 2 * n_layers * (4 * d * d + 2 * d * ffn) + 4 * n_layers * L * d + 2 * d * V


def prof_reset(lib):
    lib.bd_prof_enable(1)
    lib.bd_prof_read(None, None, None, 1)


def prof_table(lib, wall_ms):
    """Per-family launches / average duration / rate since the last reset (HIP events recorded by the library on
    the launch stream around each entry point's launches)."""
    out = {}
    for kind, (name, unit) in KINDS.items():
        n_k, ms_k, w_k = C.c_longlong(0), C.c_double(0.0), C.c_double(0.0)
        lib.bd_prof_read_kind(kind, C.byref(n_k), C.byref(ms_k), C.byref(w_k), 1)
        if not n_k.value:
            continue
        ent = {"kernel": name, "launches": n_k.value, "avg_launch_ms": ms_k.value / n_k.value, "total_ms": ms_k.value,
               "share_of_timed_region": ms_k.value / wall_ms if wall_ms else None}
        rate = w_k.value / (ms_k.value * 1e-3) if ms_k.value > 0 else 0.0
        if unit == "flop":
            ent.update(bound="mfma", tflops=rate / 1e12, frac=rate / 1e12 / MFMA_BF16_PEAK_TFLOPS, work_per_launch=w_k.value / n_k.value)
        else:
            ent.update(bound="hbm", gbps=rate / 1e9, frac=rate / 1e9 / HBM_PEAK_GBPS, work_per_launch=w_k.value / n_k.value)
        out[kind] = ent
    lib.bd_prof_enable(0)
    return out


class BoardSampler:
    """Samples the board's own sensors (sysfs hwmon of the GPU's PCI device: average socket power, shader clock) every
    50 ms on a thread while a timed region runs, so that a bench line tells a box that runs hot or slow from a slower
    kernel. Everything is optional: unreadable files give nulls."""

    def __init__(self, dev):
        import glob
        self.power, self.sclk = [], []
        self._files = {}
        try:
            p = torch.cuda.get_device_properties(dev)
            bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            base = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
        except Exception:
            base = []
        if not base:
            base = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")[:1]
        for b in base[:1]:
            for key, names in (("power", ("power1_average", "power1_input")), ("sclk", ("freq1_input",))):
                for n in names:
                    if os.path.exists(os.path.join(b, n)):
                        self._files[key] = os.path.join(b, n)
                        break
            cap = os.path.join(b, "power1_cap")
            self.cap_w = self._read(cap, 1e-6) if os.path.exists(cap) else None
        self._stop = None

    @staticmethod
    def _read(path, scale):
        try:
            with open(path) as fh:
                return float(fh.read().strip()) * scale
        except (OSError, ValueError):
            return None

    def __enter__(self):
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                if "power" in self._files:
                    v = self._read(self._files["power"], 1e-6)
                    if v is not None:
                        self.power.append(v)
                if "sclk" in self._files:
                    v = self._read(self._files["sclk"], 1e-6)
                    if v is not None:
                        self.sclk.append(v)
                self._stop.wait(0.05)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thr.join(timeout=1.0)

    def summary(self):
        def stats(v):
            return None if not v else {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "samples": len(v)}
        return {"power_w": stats(self.power), "power_cap_w": getattr(self, "cap_w", None), "sclk_mhz": stats(self.sclk),
                "source": "sysfs hwmon of the GPU's PCI device, 50 ms samples over the timed region" if self._files else None}


def kernel_clock_mhz(lib, dev, launch, n_workgroups):
    """Shader clock the chip holds INSIDE the single-launch forward: every workgroup stamps the shader-clock counter
    and the 100 MHz real-time counter at its start and end (bd_clock_stamps_set); median over workgroups of
    delta(shader clock) / delta(real time) x 100 MHz. Run right after the timed region, on the same data."""
    stamps = torch.zeros(n_workgroups * 4, dtype=torch.int64, device=dev)
    if lib.bd_clock_stamps_set(C.c_void_p(stamps.data_ptr()), n_workgroups) != 0:
        return None
    try:
        launch()
        torch.cuda.synchronize()
    finally:
        lib.bd_clock_stamps_set(None, 0)
    s = stamps.view(n_workgroups, 4).cpu().double()
    ok = (s[:, 3] > s[:, 1]) & (s[:, 2] > s[:, 0])
    if int(ok.sum()) == 0:
        return None
    mhz = (s[ok, 2] - s[ok, 0]) / (s[ok, 3] - s[ok, 1]) * 100.0
    cyc = s[ok, 2] - s[ok, 0]
    return {"median": float(mhz.median()), "min": float(mhz.min()), "max": float(mhz.max()),
            "workgroups": int(ok.sum()), "median_workgroup_cycles": float(cyc.median()),
            "how": "s_memtime / s_memrealtime stamped by every workgroup of one more single-launch forward right after the timed region"}


def zero_operand_control(lib, dev, cfg, bb, ids, ts, keymask, hidden, seconds=0.7):
    """The single-launch forward on the benchmark batch with the model's weights and with ALL-ZERO weights: the same
    instruction stream and flop count, but almost nothing switches in the matrix cores, so the board stays far below
    its power cap and holds its full clock. The gap between the two launch times is what the chip's power management
    takes on real operands; the zero-weight figure is what the kernel's structure sustains at the nominal clock."""
    from bidiff.model.backbone import DiffusionBackbone
    zb = DiffusionBackbone(**cfg).to(dev).eval()
    with torch.no_grad():
        for p_ in zb.parameters():
            p_.zero_()
    B, L = ids.shape
    flops = B * L * fwd_flops_per_token(cfg["hidden_size"], cfg["num_layers"], cfg["ffn_hidden_size"], L, 0)
    out = {}
    for name, model in (("model_weights", bb), ("zero_weights", zb)):
        f = lambda: model.run_forward(ids, ts, keymask, want_hidden=True, hidden_out=hidden)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n, t0 = 0, time.perf_counter()
        with BoardSampler(dev) as bs:
            e0.record()
            while time.perf_counter() - t0 < seconds:
                for _ in range(10):
                    f()
                n += 10
                torch.cuda.synchronize()
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        clk = kernel_clock_mhz(lib, dev, f, B)
        out[name] = {"ms_per_launch": ms, "tflops": flops / ms / 1e9, "frac_of_peak": flops / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS,
                     "kernel_clock_mhz": clk and clk["median"], "power_w": (bs.summary().get("power_w") or {}).get("mean")}
    out["what"] = ("full single-launch forward (every row of every layer), same batch; zero weights = same instructions "
                   "and flops with idle matrix-core datapaths: the board holds its nominal clock")
    return out


def roofline_of(table, traffic=None, traffic_src=None):
    dom = max(table, key=lambda k: table[k]["total_ms"])
    e = table[dom]
    mfma = e["bound"] == "mfma"
    return {"bound": e["bound"], "kernel": e["kernel"], "achieved": e["tflops"] if mfma else e["gbps"],
            "peak": MFMA_BF16_PEAK_TFLOPS if mfma else HBM_PEAK_GBPS, "unit": "TFLOP/s" if mfma else "GB/s",
            "frac": e["frac"], "traffic": traffic, "traffic_source": traffic_src, "launches": e["launches"],
            "avg_launch_ms": e["avg_launch_ms"], ("flops_per_launch" if mfma else "bytes_per_launch"): e["work_per_launch"],
            "share_of_timed_region": e["share_of_timed_region"]}


def named(table):
    return {v["kernel"].split(" ")[0]: {k: x for k, x in v.items() if k != "kernel"} for v in table.values()}


def cpu_baseline(sd, cfg, sampler, L, T, n_seq, threads):
    """Oracle port of the same workload on the host cores: BASELINE configuration 1 when n_seq = 256."""
    from oracle import cpu_baseline as CB
    from oracle import sampler_ref as ora
    torch.set_num_threads(threads)
    tb = ora.Tables.from_token_tables(sampler._core.tt)
    t0 = time.perf_counter()
    _, t_net, t_step = CB.sample_smiles(sd, cfg, tb, n_seq, L, T, seed=1)
    dt = time.perf_counter() - t0
    return {"value": n_seq / dt, "unit": "sequences/s", "cores": threads, "kind": "port",
            "sample": f"{n_seq} sequences in one batch x {T} steps, L={L}, constraints on, fp32 PyTorch denoiser + C sampling "
                      f"oracle, {dt:.1f} s ({t_net:.1f} s denoiser, {t_step:.1f} s sampling loop)"}


def bpe_batch(batch, L, V, seed):
    import numpy as np
    rs = np.random.RandomState(seed)
    lens = np.clip(np.round(rs.normal(38.5, 15.0, size=batch)), 12, L).astype(np.int64)
    ids = rs.randint(5, V, size=(batch, L)).astype(np.int64)
    attn = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
    ids[attn == 0] = 0
    ids[:, 0] = 2
    ids[np.arange(batch), lens - 1] = 3
    return ids, attn


def train_bench(dev, rank, world, steps, warmup, cpu_steps, threads, batch=256, L=128):
    """Second headline number: non-pad tokens/s of the SMILES_BPE training step (V = 803, T = 50,
    per-rank batch 256, L = 128 padded, dropout 0.1 on, AdamW lr 3e-4 wd 0.01, clip 1.0; one gradient-bucket
    all-reduce per step when world > 1)."""
    from bidiff import _lib
    from bidiff.model.backbone import DiffusionBackbone
    from bidiff.model.diffusion import DiscreteMaskingDiffusion
    from bidiff.training.optimizer import FlatAdamW
    from bidiff.parallel import GradientExchange, allreduce_mean_, tail_split_offset
    from bidiff.training import autograd as AG
    V, T = 803, 50
    cfg = dict(vocab_size=V, hidden_size=384, num_layers=6, num_heads=3, ffn_hidden_size=1536,
               max_position_embeddings=256, num_diffusion_steps=T, dropout=0.1, pad_token_id=0)
    torch.manual_seed(42)
    bb = DiffusionBackbone(**cfg)
    model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=1, pad_token_id=0, bos_token_id=2, eos_token_id=3)
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model = model.to(dev).train()
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
    ids, attn = bpe_batch(batch, L, V, 1 + rank)
    ids_d, attn_d = torch.from_numpy(ids).to(dev), torch.from_numpy(attn).to(dev)
    tokens = int(attn.sum())
    lens = attn.sum(1).tolist()       # host-side lengths, as a data loader knows them: no device round trip

    # the exchange in two buckets: the upper three layers' gradients (plus final norm and vocabulary head) are
    # all-reduced while the lower three layers' backward runs (BD_OVERLAP_ALLREDUCE=0: one blocking all-reduce)
    exchange = None
    if world > 1 and not staged:
        off = tail_split_offset(opt, bb, 3)
        if off is None or os.environ.get("BD_OVERLAP_ALLREDUCE", "1") == "0":
            off = opt.flat_grad.numel()
        exchange = GradientExchange(opt.flat_grad, off, world)
        AG.set_tail_ready_hook(exchange.tail_ready, 3, model=bb)

    def step():
        out = model(ids_d, attn_d, lengths=lens)
        if exchange is not None:
            exchange.arm()
        out["loss"].backward()
        if world > 1:
            if staged:
                c_ = opt.flat_grad.cpu()
                allreduce_mean_(c_, world)
                opt.flat_grad.copy_(c_)
            else:
                exchange.finish()
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
    final_loss = float(loss.item())
    ms_step = 1e3 * el / steps
    # per-kernel table: a few more steps with the library's events on (they cost a few microseconds per launch, so
    # these steps are not the timed ones)
    lib = _lib.lib()
    n_prof = max(2, min(5, steps))
    torch.cuda.synchronize()
    prof_reset(lib)
    t0 = time.perf_counter()
    for _ in range(n_prof):
        step()
    torch.cuda.synchronize()
    table = prof_table(lib, 1e3 * (time.perf_counter() - t0))
    # the same step with every padded row computed (skip_padded_rows off), for comparison
    model.skip_padded_rows = False
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_all = max(3, steps // 4)
    for _ in range(n_all):
        step()
    torch.cuda.synchronize()
    ms_all_rows = 1e3 * (time.perf_counter() - t0) / n_all
    model.skip_padded_rows = True
    # flops of the rows that are computed: projections, feed-forward and head per packed row (rounded up to 128
    # rows), attention per sequence over its own length; forward + 2x backward
    rows = (tokens + 127) // 128 * 128
    per_row = 2 * 6 * (4 * 384 * 384 + 2 * 384 * 1536) + 2 * 384 * V
    flops_step = 3.0 * (per_row * rows + 4 * 6 * 384 * float(sum(n * n for n in lens)))
    flops_all_rows = 3.0 * fwd_flops_per_token(384, 6, 1536, L, V) * batch * L
    tfl = flops_step / (ms_step * 1e-3) / 1e12
    res = {"metric": "train non-pad tokens/sec (Bi_Diffusion_SMILES_BPE training step)",
           "value": float(tok.item()) * steps / el, "unit": "non-pad tokens/s",
           "ms_per_step": ms_step, "rows_computed_per_step": rows, "rows_padded_layout": batch * L,
           "all_rows_variant": {"ms_per_step": ms_all_rows, "value": tokens / (ms_all_rows * 1e-3),
                                "tflops": flops_all_rows / (ms_all_rows * 1e-3) / 1e12,
                                "note": "skip_padded_rows = False: all B*L rows computed, as the reference does"},
           "steps": steps, "warmup": warmup, "final_loss": final_loss, "dtype": "bf16",
           "config": {"workload": f"SMILES_BPE small backbone (V=803, T=50), per-GPU batch {batch}, L={L} padded, "
                                  f"{tokens} non-pad tokens per rank-batch, dropout 0.1 at all four sites (in-kernel Philox), AdamW + clip 1.0; "
                                  f"rows behind each sequence's last token are left out of the kernels (no loss, no gradient, "
                                  f"no influence on other rows; same loss and gradients)",
                      "parallelism": f"dp{world}, flat gradient buffer all-reduced per step in two buckets, the first "
                                     f"overlapped with the lower layers' backward"},
           # whole step against the MFMA roofline: 3 x forward flops of the computed rows / step time
           "roofline": {"bound": "mfma", "kernel": "whole training step (forward + backward + clip + AdamW)",
                        "achieved": tfl, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / MFMA_BF16_PEAK_TFLOPS,
                        "flops_per_step": flops_step, "traffic": None,
                        # the same step measured in the work of the padded layout the reference computes (every one of
                        # the B * L rows): what the step retires of THAT workload per second; the packed path does a
                        # third of these flops, so this is a throughput equivalent, not matrix-core utilisation
                        "padded_layout_equivalent": {"flops_per_step": flops_all_rows,
                                                     "achieved": flops_all_rows / (ms_step * 1e-3) / 1e12,
                                                     "frac": flops_all_rows / (ms_step * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS},
                        "dominant_kernel": roofline_of(table) if table else None},
           "kernels": named(table), "profiled_steps": n_prof}
    if cpu_steps > 0 and world == 1 and rank == 0:
        from oracle import cpu_baseline as CB
        torch.set_num_threads(threads)
        sec, closs = CB.train_steps_bpe(sd_cpu, cfg, torch.from_numpy(ids), torch.from_numpy(attn), T, cpu_steps)
        res["cpu_baseline"] = {"value": tokens / sec, "unit": "non-pad tokens/s", "cores": threads, "kind": "port",
                               "sample": f"{cpu_steps} timed steps (1 warm-up) of the same batch ({batch} x {L}, {tokens} non-pad tokens): "
                                         f"fp32 PyTorch restatement of corruption + denoiser + masked CE, autograd, "
                                         f"clip_grad_norm_ + torch AdamW, dropout off; {sec:.2f} s per step"}
    return res


def variant_bench(dev, variant, B, L, T, world, rank):
    """One more sampling configuration on the small backbone: SELFIES (V = 43), Group SELFIES (V = 233) or
    SMILES_BPE (V = 803, where the chosen rows' vocabulary logits are a visible part of a reverse step)."""
    from bidiff import _lib
    from bidiff.model.backbone import DiffusionBackbone
    from bidiff.model.diffusion import DiscreteMaskingDiffusion
    from bidiff.utils import synthetic
    if variant == "selfies":
        from bidiff.sampling.selfies_sampler import ConstrainedSampler
        tok = synthetic.selfies_tokenizer()
    elif variant == "bpe":
        from bidiff.sampling.sampler import ConstrainedSampler
        tok = synthetic.bpe_tokenizer()
    else:
        from bidiff.sampling.group_selfies_sampler import ConstrainedSampler
        tok = synthetic.group_selfies_tokenizer()
    torch.manual_seed(42)
    bb = DiffusionBackbone(tok.vocab_size, 384, 6, 3, 1536, max(256, L), T, 0.1, tok.pad_token_id)
    model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=tok.mask_token_id, pad_token_id=tok.pad_token_id,
                                     bos_token_id=tok.bos_token_id, eos_token_id=tok.eos_token_id).to(dev).eval()
    smp = ConstrainedSampler(model, tok, num_steps=T, temperature=1.0, use_constraints=True, device=str(dev))
    smp._core.seed = 7
    lib = _lib.lib()

    def run(i):
        if variant == "bpe":
            ids, am, fm = smp.initial_state(B, L)
            return smp._core.run(ids, am, fm, seq_offset=(i * world + rank) * B)
        ids = torch.full((B, L), tok.mask_token_id, dtype=torch.long, device=dev)
        ids[:, 0] = tok.bos_token_id
        ids[:, -1] = tok.eos_token_id
        has_eos = torch.ones(B, dtype=torch.uint8, device=dev) if variant == "selfies" else None
        return smp._core.run(ids, torch.ones_like(ids), None, seq_offset=(i * world + rank) * B, has_eos=has_eos)
    run(0)
    torch.cuda.synchronize()
    prof_reset(lib)
    t0 = time.perf_counter()
    out = run(1)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    table = prof_table(lib, 1e3 * el)
    assert (out != tok.mask_token_id).all()
    fl = fwd_flops_per_token(384, 6, 1536, L, tok.vocab_size) * L * T * B
    rep = {"selfies": "SELFIES", "group": "Group_SELFIES", "bpe": "SMILES_BPE"}[variant]
    return {"metric": f"generated sequences/sec (Bi_Diffusion_{rep} reverse sampling)",
            "value": B / el, "unit": "sequences/s (per GPU)", "ms_per_batch": 1e3 * el,
            "config": {"workload": f"small backbone, V={tok.vocab_size}, batch {B}, L={L}, T={T}, constraints on"},
            "whole_path_tflops_reference_work": fl / el / 1e12, "roofline": roofline_of(table), "kernels": named(table)}


def ragged_bench(dev, B, T, mean_len=38):
    """SMILES reverse sampling on per-sample lengths (log-normal around the training distribution's mean, clipped to
    8..128, batch padded to 128) as the reference's step2 script draws them: sequences dealt to the forward's waves."""
    import numpy as np
    from bidiff import _lib
    from bidiff.model.backbone import DiffusionBackbone
    from bidiff.model.diffusion import DiscreteMaskingDiffusion
    from bidiff.sampling.sampler import ConstrainedSampler
    from bidiff.utils import synthetic
    tok = synthetic.smiles_tokenizer()
    torch.manual_seed(42)
    bb = DiffusionBackbone(tok.vocab_size, 384, 6, 3, 1536, 256, T, 0.1, tok.pad_token_id)
    model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=tok.mask_token_id, pad_token_id=tok.pad_token_id,
                                     bos_token_id=tok.bos_token_id, eos_token_id=tok.eos_token_id).to(dev).eval()
    smp = ConstrainedSampler(model, tok, num_steps=T, temperature=1.0, use_constraints=True, device=str(dev))
    smp._core.seed = 7
    lengths = np.clip(np.random.RandomState(0).lognormal(np.log(mean_len), 0.45, size=B).astype(int), 8, 128).tolist()
    lib = _lib.lib()
    smp.sample_with_lengths(lengths, max_length=128, show_progress=False)
    torch.cuda.synchronize()
    prof_reset(lib)
    t0 = time.perf_counter()
    ids, _ = smp.sample_with_lengths(lengths, max_length=128, show_progress=False)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    table = prof_table(lib, 1e3 * el)
    assert (ids != tok.mask_token_id).all()
    return {"metric": "generated sequences/sec (Bi_Diffusion_SMILES reverse sampling, per-sample lengths)",
            "value": B / el, "unit": "sequences/s (per GPU)", "ms_per_batch": 1e3 * el,
            "config": {"workload": f"small backbone, batch {B} padded to {max(lengths)}, lengths mean "
                                   f"{float(np.mean(lengths)):.1f}, T={T}, constraints on, decode included"},
            "kernels": named(table)}


def c5_bench(dev, B, L, T):
    """The stress configuration: 24 layers, d = 1024, ffn 4096, 8 heads of 128, L = 512."""
    from bidiff import _lib
    from bidiff.model.backbone import DiffusionBackbone
    from bidiff.model.diffusion import DiscreteMaskingDiffusion
    from bidiff.sampling.sampler import ConstrainedSampler
    from bidiff.utils import synthetic
    tok = synthetic.smiles_tokenizer()
    torch.manual_seed(42)
    bb = DiffusionBackbone(tok.vocab_size, 1024, 24, 8, 4096, L, T, 0.1, tok.pad_token_id)
    model = DiscreteMaskingDiffusion(bb, num_steps=T, mask_token_id=tok.mask_token_id, pad_token_id=tok.pad_token_id,
                                     bos_token_id=tok.bos_token_id, eos_token_id=tok.eos_token_id).to(dev).eval()
    smp = ConstrainedSampler(model, tok, num_steps=T, temperature=1.0, use_constraints=True, device=str(dev))
    smp._core.seed = 11
    lib = _lib.lib()
    ids0, am0, fm0 = smp.initial_state(B, L)
    t_ = torch.full((B,), T, device=dev, dtype=torch.long)
    with torch.no_grad():
        bb.get_hidden_states(ids0, t_, am0)                 # warm-up forward (weight packing)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = smp._core.run(ids0, am0, fm0, seq_offset=0)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert (out != tok.mask_token_id).all()
    # per-kernel table from three more forwards with the library's events on (about 170 launches per forward:
    # the events are kept out of the timed loop)
    prof_reset(lib)
    t1 = time.perf_counter()
    with torch.no_grad():
        for _ in range(3):
            bb.get_hidden_states(out, t_, am0)
    torch.cuda.synchronize()
    table = prof_table(lib, 1e3 * (time.perf_counter() - t1))
    fl = fwd_flops_per_token(1024, 24, 4096, L, tok.vocab_size) * L * T * B
    return {"metric": "generated sequences/sec (scaled-up backbone, 24 layers, d=1024, L=512)",
            "value": B / el, "unit": "sequences/s (per GPU)", "ms_per_batch": 1e3 * el, "ms_per_forward": 1e3 * el / T,
            "config": {"workload": f"24 layers, d=1024, ffn 4096, 8 heads, V={tok.vocab_size}, batch {B}, L={L}, T={T}, constraints on"},
            "whole_path_tflops": fl / el / 1e12, "whole_path_frac": fl / el / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            "roofline": roofline_of(table), "kernels": named(table)}


def sizes_bench(dev, B=2048, L=128):
    """Full denoiser forward of the reference's other model sizes (Bi_Diffusion_SMILES/configs/config.yaml model_sizes:
    medium, large, xl) at the sampling shape B x 128: time per forward and whole-path TFLOP/s."""
    from bidiff.model.backbone import DiffusionBackbone
    from bidiff.utils import synthetic
    tok = synthetic.smiles_tokenizer()
    res = {}
    for name, (d, n, h, f) in (("medium", (768, 12, 6, 3072)), ("large", (1152, 18, 9, 4608)), ("xl", (1536, 24, 12, 6144))):
        torch.manual_seed(1)
        bb = DiffusionBackbone(tok.vocab_size, d, n, h, f, 256, 100, 0.1, tok.pad_token_id).to(dev).eval()
        ids = torch.randint(5, tok.vocab_size, (B, L), device=dev)
        t_ = torch.full((B,), 50, device=dev, dtype=torch.long)
        am = torch.ones(B, L, dtype=torch.long, device=dev)
        with torch.no_grad():
            for _ in range(2):
                bb.get_hidden_states(ids, t_, am)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                bb.get_hidden_states(ids, t_, am)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        tfl = fwd_flops_per_token(d, n, f, L, tok.vocab_size) * B * L / dt / 1e12
        res[name] = {"hidden": d, "layers": n, "heads": h, "ffn": f, "batch": B, "seq_len": L, "ms_per_forward": 1e3 * dt,
                     "whole_path_tflops": tfl, "whole_path_frac": tfl / MFMA_BF16_PEAK_TFLOPS}
        del bb
        torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--diffusion-steps", type=int, default=100)
    ap.add_argument("--cpu-seqs", type=int, default=256,
                    help="sequences in the CPU baseline (256 = BASELINE configuration 1; 0 = skip)")
    ap.add_argument("--cpu-train-steps", type=int, default=2, help="timed CPU training steps (0 = skip)")
    ap.add_argument("--no-constraints", action="store_true")
    ap.add_argument("--train-steps", type=int, default=50, help="timed training steps after 10 warm-up steps (0 = skip the training leg)")
    ap.add_argument("--no-configs", action="store_true", help="skip the SELFIES / Group SELFIES / stress-backbone records")
    ap.add_argument("--c5-batch", type=int, default=128)
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
    prof_reset(lib)
    board = BoardSampler(dev)
    barrier()
    with board:
        t0 = time.perf_counter()
        for i in range(args.steps):
            ids = run_batch(args.warmup + i)
        barrier()
        elapsed = time.perf_counter() - t0
    table = prof_table(lib, 1e3 * elapsed)
    # the clock the chip holds inside the dominant kernel, measured while it is still warm from the timed region
    clock, control = None, None
    if rank == 0:
        try:
            ids_c, am_c, _ = smp.initial_state(B, L)
            ts_c = torch.full((B,), T // 2, device=dev, dtype=torch.long)
            km_c = (am_c != 0).to(torch.uint8)
            hid_c = torch.empty(B, L, 384, dtype=torch.bfloat16, device=dev)
            clock = kernel_clock_mhz(lib, dev, lambda: bb.run_forward(ids_c, ts_c, km_c, want_hidden=True, hidden_out=hid_c), B)
        except Exception as exc:
            clock = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1:
            try:
                control = zero_operand_control(lib, dev, cfg, bb, ids_c, ts_c, km_c, hid_c)
            except Exception as exc:
                control = {"error": f"{type(exc).__name__}: {exc}"}
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # strings are decoded on the host outside the timed region (host tokenizer work)
    assert (ids != tok.mask_token_id).all()
    del ids
    torch.cuda.empty_cache()
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    train = None
    if args.train_steps > 0:
        try:
            train = train_bench(dev, rank, world, args.train_steps, 10, args.cpu_train_steps, threads)
        except Exception as exc:      # the sampling line (the headline metric) is still reported
            train = {"error": f"{type(exc).__name__}: {exc}"}
    configs = None
    if not args.no_configs:
        configs = {}
        for name, fn in (("selfies", lambda: variant_bench(dev, "selfies", B, L, T, world, rank)),
                         ("group_selfies", lambda: variant_bench(dev, "group", B, L, T, world, rank)),
                         ("smiles_bpe", lambda: variant_bench(dev, "bpe", B, L, 50, world, rank)),
                         # sequences of at most 64 tokens: two per workgroup of the single-launch forward
                         ("selfies_L64", lambda: variant_bench(dev, "selfies", B, 64, T, world, rank)),
                         ("smiles_lengths_mean38", lambda: ragged_bench(dev, B, T)),
                         ("c5_24L_d1024_L512", lambda: c5_bench(dev, args.c5_batch, 512, T)),
                         ("model_sizes_forward", lambda: sizes_bench(dev))):
            torch.cuda.empty_cache()
            try:
                configs[name] = fn()
            except Exception as exc:
                configs[name] = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        total_seqs = world * B * args.steps
        value = total_seqs / elapsed
        flops_seq = fwd_flops_per_token(384, 6, 1536, L, tok.vocab_size) * L * T
        # HBM-side bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process,
        # so the figure comes from the committed rocprofv3 --pmc passes (profiles/*_pmc_traffic.json)
        traffic, traffic_src = None, None
        dom = max(table, key=lambda k: table[k]["total_ms"])
        for fname in ("r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as fh:
                    tj = json.load(fh)
                ent = tj["kernels"].get(KINDS[dom][0].split(" ")[0])
                if ent and B * L == 4096 * 128:
                    traffic, traffic_src = ent["traffic_bytes"], tj["source"]
                    break
            except (OSError, ValueError, KeyError):
                pass
        out = {
            "metric": "generated sequences/sec (Bi_Diffusion_SMILES reverse sampling, 100 steps, L=128)",
            "value": value, "unit": "sequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init weights, synthetic 39-token vocabulary)",
            "config": {"workload": f"Bi_Diffusion_SMILES small backbone (d384, 6 layers, 3 heads, ffn 1536, V=39), "
                                   f"batch {B} per GPU, L={L}, T={T} reverse steps, constraints "
                                   f"{'off' if args.no_constraints else 'on'}, rows chosen before each forward (t>1): the last "
                                   f"feed-forward block, final LayerNorm and vocabulary logits run for the chosen rows only; "
                                   f"full logits at t=1, on-device repairs. Timed call: ReverseSampler.run (what sample_batch "
                                   f"runs per batch; host decode of the ids to strings excluded), wall clock around "
                                   f"{args.steps} batches between device synchronisations; kernel durations from HIP events",
                       "batch_per_gpu": B, "seq_len": L, "diffusion_steps": T, "sharding": f"{world} x independent batches"},
            "roofline": roofline_of(table, traffic, traffic_src),
            # (the contract's objects and a short summary first, the long per-kernel tables behind them)
            "cpu_baseline": None,
            "summary": None,
            "kernels": named(table),
            # reference-forward flops per generated sequence / time: a throughput equivalent in the work the
            # reference's loop does (it computes every row of every layer at every step), not matrix-core utilisation
            "whole_path_tflops_reference_work": value / world * flops_seq / 1e12,
            "board": dict(board.summary(), kernel_clock_mhz=clock, zero_operand_control=control),
        }
        if train is not None:
            out["train"] = train
        if configs is not None:
            out["configs"] = configs
        if args.cpu_seqs > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(sd_cpu, cfg, smp, L, T, args.cpu_seqs, threads)
        else:
            del out["cpu_baseline"]
        summ = {"sequences_per_s": value, "dominant_kernel_frac": out["roofline"].get("frac")}
        if isinstance(control, dict) and "zero_weights" in control:
            summ["zero_operand_control_frac"] = control["zero_weights"].get("frac_of_peak")
        if isinstance(train, dict) and "ms_per_step" in train:
            summ["train_ms_per_step"] = train["ms_per_step"]
            summ["train_tokens_per_s"] = train["value"]
            summ["train_frac"] = train["roofline"]["frac"]
        if isinstance(configs, dict):
            c5 = configs.get("c5_24L_d1024_L512")
            if isinstance(c5, dict) and "whole_path_frac" in c5:
                summ["c5_sequences_per_s"] = c5["value"]
                summ["c5_whole_path_frac"] = c5["whole_path_frac"]
            ms = configs.get("model_sizes_forward")
            if isinstance(ms, dict) and "xl" in ms:
                summ["model_sizes_forward_frac"] = {k: v["whole_path_frac"] for k, v in ms.items() if isinstance(v, dict) and "whole_path_frac" in v}
        out["summary"] = summ
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
