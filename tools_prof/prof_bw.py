"""Achieved bandwidth of the HBM-bound kernels of the path: algorithmic bytes / HIP-event duration."""
import sys, json, torch, ctypes as C
sys.path.insert(0, 'discrete-diffusion-model-for-inverse-polymer-design_amd'); sys.path.insert(0, '.')
from bidiff import _lib
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
lib = _lib.lib()
s = _lib.stream_handle
P = _lib.ptr
res = {}
def timeit(name, fn, bytes_, iters=20, note=""):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): rc = fn()
    e1.record(); torch.cuda.synchronize()
    assert rc == 0, lib.bd_last_error()
    us = e0.elapsed_time(e1) / iters * 1e3
    res[name] = {"algorithmic_bytes": bytes_, "avg_us": round(us, 2), "GBps": round(bytes_ / us / 1e3, 1),
                 "frac_of_8TBps": round(bytes_ / us / 1e3 / 8000, 4), "note": note}
    print(name, res[name], flush=True)

# q(x_t | x_0): Philox uniforms in-kernel: ids 8 + attn 8 + noisy 8 + indicator 1 = 25 B/token
for B in (4096, 262144):
    L, T = 128, 100
    ids = torch.randint(5, 39, (B, L), device=dev); attn = torch.ones(B, L, dtype=torch.long, device=dev)
    t = torch.randint(1, T + 1, (B,), device=dev); sched = torch.linspace(0.05, 0.95, T + 1, device=dev)
    noisy = torch.empty_like(ids); ind = torch.empty(B, L, dtype=torch.uint8, device=dev)
    timeit(f"corrupt_kernel B={B} L={L}", lambda: lib.bd_corrupt(P(ids), P(attn), P(t), P(sched), T + 1, None, P(noisy), P(ind), B, L,
                                                                 1, 0, 2, 3, 42, 0, s()), 25 * B * L + 8 * B)
    del ids, attn, noisy, ind
# masked CE at the training shape: rows with weight 0 are never read (about 30 % of the rows are non-pad, ~half of them masked)
M, V = 32768, 803
lg = torch.randn(M, V, device=dev); tg = torch.randint(0, V, (M,), device=dev)
indm = (torch.rand(M, device=dev) < 0.16).to(torch.uint8); lse = torch.empty(M, device=dev); sums = torch.zeros(32 * 17, device=dev)
nz = int(indm.sum().item())
timeit("ce_fwd_kernel M=32768 V=803 (16% rows live)", lambda: lib.bd_masked_ce_fwd_parts(P(lg), P(tg), P(indm), None, P(lse), P(sums), 16, M, V, s()),
       nz * V * 4 + M * (8 + 1 + 4), note="reads only live rows")
indf = torch.ones(M, dtype=torch.uint8, device=dev)
timeit("ce_fwd_kernel M=32768 V=803 (all rows live)", lambda: lib.bd_masked_ce_fwd_parts(P(lg), P(tg), P(indf), None, P(lse), P(sums), 16, M, V, s()),
       M * V * 4 + M * 13, note="sums into 16 pairs in cache lines of their own")
lib.bd_masked_ce_finish(P(sums), P(torch.empty(1, device=dev)), 16, s())
g1 = torch.ones(1, device=dev); dl = torch.empty_like(lg)
timeit("ce_bwd_kernel M=32768 V=803 (all rows live)", lambda: lib.bd_masked_ce_bwd(P(lg), P(tg), P(indf), None, P(lse), P(sums), P(g1), P(dl), M, V, s()),
       2 * M * V * 4 + M * 17)
# LayerNorm and embeddings at the sampling shape
M, d = 4096 * 128, 384
x = torch.randn(M, d, device=dev); gm = torch.ones(d, device=dev); bt = torch.zeros(d, device=dev)
ob = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
timeit("layernorm_kernel M=524288 d=384", lambda: lib.bd_layernorm(P(x), P(gm), P(bt), P(ob), None, None, M, d, 1e-5, s()), M * d * 6)
ids = torch.randint(0, 39, (4096, 128), device=dev); tt = torch.randint(1, 101, (4096,), device=dev)
te, pe, me = torch.randn(39, d, device=dev), torch.randn(256, d, device=dev), torch.randn(101, d, device=dev)
timeit("embed_kernel B=4096 L=128 d=384", lambda: lib.bd_embed(P(ids), P(tt), P(te), P(pe), P(me), P(x), 4096, 128, d, te.shape[0], pe.shape[0], me.shape[0], s()), M * d * 4 + M * 8,
       note="tables are cache resident; bytes = output + ids")
# optimizer over the flat buffers of the BPE small model
n = 11382144
p_, g_, m_, v_ = (torch.randn(n, device=dev) * 0.01 for _ in range(4)); v_.abs_()
ss = torch.zeros(1, device=dev)
timeit("sumsq_kernel n=11.4M", lambda: lib.bd_sumsq_f32(P(g_), n, P(ss), s()), n * 4)
timeit("adamw_kernel n=11.4M", lambda: lib.bd_adamw_step(P(p_), P(g_), P(m_), P(v_), n, 3e-4, 0.9, 0.999, 1e-8, 0.01, 0.1, 0.03, P(ss), 1.0, s()), n * 28)
# LayerNorm backward at the packed and at the padded training shape (dx in place, bf16 copy with dropout)
drop = _lib.Dropout(0.1, 1, 2)
for M in (10240, 32768):
    d = 384
    xx = torch.randn(M, d, device=dev); st = torch.stack([xx.mean(1), xx.var(1, unbiased=False).add(1e-5).rsqrt()], 1).contiguous()
    dy = torch.randn(M, d, device=dev).bfloat16(); dx = torch.randn(M, d, device=dev); obb = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
    dgm, dbt = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    parts = torch.zeros(8, 2, d, device=dev)
    timeit(f"layernorm_bwd_kernel M={M} d=384 (dx in place, dropped bf16 copy, gamma / beta sums into 8 copies)",
           lambda: lib.bd_layernorm_bwd_parts(P(dy), P(xx), P(st), P(gm), P(dx), P(dx), P(parts), 8, M, d, P(obb), C.byref(drop), 3, s()),
           M * d * (2 + 4 + 4 + 4 + 2))
# id -> text decode of a sampled batch
from bidiff.sampling.decode import DeviceDecoder
from bidiff.utils import synthetic
tok = synthetic.smiles_tokenizer()
dec = DeviceDecoder(tok.id_to_token, tok.SPECIAL_TOKENS, tok.vocab_size, dev)
B, L = 4096, 128
ids = torch.randint(5, tok.vocab_size, (B, L), device=dev)
stride = L * dec.max_token_bytes
outb = torch.empty(B, stride, dtype=torch.uint8, device=dev); lens = torch.empty(B, dtype=torch.int32, device=dev)
lib.bd_decode_ids(P(ids), B, L, P(dec._text), P(dec._offs), dec.V, P(outb), P(lens), stride, s()); torch.cuda.synchronize()
timeit("decode_kernel B=4096 L=128 (V=39)", lambda: lib.bd_decode_ids(P(ids), B, L, P(dec._text), P(dec._offs), dec.V, P(outb), P(lens), stride, s()),
       B * L * 8 + int(lens.sum().item()) + 4 * B, note="ids in + text bytes out")
# one reverse step at V = 803 with the chosen-row workspace (selection + row product + draws), hidden mode
from tests import helpers as H
from bidiff.sampling import core
import numpy as np
smp, tk = H.make_sampler("bpe", num_steps=50)
tb = H.oracle_tables(smp); dt = H.hip_tables(tb, dev)
d = 384
hidden = torch.randn(B, L, d, device=dev).bfloat16(); wout = (torch.randn(1024, d, device=dev) / d ** 0.5).bfloat16()
ktab = core.unmask_count_table(50, L)
t_ = 40
ids0 = torch.from_numpy(np.where(np.random.RandomState(0).rand(B, L) < 0.8, tb.mask_id, 5).astype(np.int64)).to(dev)
kt = torch.as_tensor(np.ascontiguousarray(ktab[t_], dtype=np.int32)).to(dev)
for cap in (0, 4):
    a = _lib.StepArgs()
    work = ids0.clone()
    a.ids = work.data_ptr(); a.hidden, a.wout, a.logits, a.d = hidden.data_ptr(), wout.data_ptr(), None, d
    a.ktable = kt.data_ptr()
    a.tok_flags, a.contrib, a.ring_slot = dt["tok_flags"].data_ptr(), dt["contrib"].data_ptr(), dt["ring_slot"].data_ptr()
    a.counted_ids, a.bond_ids, a.non_special = dt["counted_ids"].data_ptr(), dt["bond_ids"].data_ptr(), dt["non_special"].data_ptr()
    a.n_counted, a.n_bond, a.n_ns, a.R = len(tb.counted_list), len(tb.bond_ids_list), len(tb.non_special_list), len(tb.ring_ids_list)
    a.B, a.L, a.V = B, L, tb.V
    a.mask_id, a.pad_id, a.bos_id, a.eos_id = tb.mask_id, tb.pad_id, tb.bos_id, tb.eos_id
    a.open_id, a.close_id, a.boost_id, a.count_id = tb.open_id, tb.close_id, tb.boost_id, tb.count_id
    a.variant, a.use_constraints, a.t, a.temperature, a.seed, a.seq_offset = tb.variant, 1, t_, 1.0, 3, 0
    if cap:
        rl = torch.empty(B * cap, (tb.V + 31) // 32 * 32, device=dev); rp = torch.empty(B * cap, dtype=torch.int32, device=dev)
        a.row_logits, a.row_pos, a.row_cap = rl.data_ptr(), rp.data_ptr(), cap
    def one():
        work.copy_(ids0)
        return lib.bd_sample_step(C.byref(a), s())
    k = int(ktab[t_][int(0.8 * L)])
    timeit(f"bd_sample_step V=803 B=4096 L=128 t=40 ({'row workspace, row_cap 4' if cap else 'rows one by one'}; incl. an ids copy)", one,
           B * (17 * L + k * (2 * d + 8)), note=f"about {k} draws per sequence; algorithmic bytes as for the V = 39 step kernel (ids, fixed mask, chosen hidden rows)")
json.dump(res, open("gpurun_out/bandwidth_kernels.json", "w"), indent=1)
