"""fp32 compute mode of the denoiser (DiffusionBackbone.compute_dtype = "fp32"): logits within 1e-5 (relative to
the largest logit) of the golden vectors captured from the reference's fp32 forward and of the fp32 CPU oracle,
on the f32-input MFMA GEMM and the fp32 attention kernel, called through the C-ABI."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import replay_inputs as RI
from oracle import backbone_ref as BR
from bidiff import _lib
from bidiff.model.backbone import DiffusionBackbone

pytestmark = pytest.mark.gpu
BLOB = np.load(os.path.join(H.GOLDEN, "model.npz"))
FP32_TOL = 1e-5   # relative error bound stated for fp32 logits


def rel_err(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def tiny_model(name):
    cfg = RI.BACKBONE_CFGS[name]
    torch.manual_seed(7)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 11)
    return cfg, bb


@pytest.mark.parametrize("name", list(RI.BACKBONE_CFGS))
def test_fp32_logits_match_reference_golden(gpu_device, name):
    cfg, bb = tiny_model(name)
    bb = bb.to(gpu_device)
    bb.compute_dtype = "fp32"
    L = 24 if name != "tiny_h4" else 150
    ids, attn, t = RI.backbone_inputs(cfg, 21, B=5, L=L)
    ti, ta, tt = (torch.from_numpy(x).to(gpu_device) for x in (ids, attn, t))
    with torch.no_grad():
        got = bb(ti, tt, ta).cpu().numpy()
        got_nm = bb(ti, tt, None).cpu().numpy()
        pooled = bb.get_pooled_output(ti[:3], tt[:3], ta[:3], "mean").cpu().numpy()
    assert got.dtype == np.float32 and got.shape == (5, L, cfg["vocab_size"]) and np.isfinite(got).all()
    assert rel_err(got, BLOB[f"{name}/logits_masked"]) < FP32_TOL
    assert rel_err(got_nm, BLOB[f"{name}/logits_nomask"]) < FP32_TOL
    assert rel_err(pooled, BLOB[f"{name}/pooled_mean"]) < FP32_TOL


@pytest.mark.parametrize("V", [39, 803])
def test_fp32_benchmark_architecture_matches_reference_golden(gpu_device, V, capsys):
    """The benchmark architecture (d 384, 6 layers, 3 heads of 128, L = 128) in fp32 mode against the reference's own
    logits at that size: 1e-5 of the largest logit; per-element relative error above a quarter of the largest 4e-5."""
    blob = np.load(os.path.join(H.GOLDEN, "model_small.npz"))
    cfg, ids, attn, t, stride = RI.small_arch_case(V)
    torch.manual_seed(42)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 5)
    bb = bb.to(gpu_device)
    bb.compute_dtype = "fp32"
    with torch.no_grad():
        got = bb(torch.from_numpy(ids).to(gpu_device), torch.from_numpy(t).to(gpu_device),
                 torch.from_numpy(attn).to(gpu_device)).cpu().numpy()[:, ::stride]
    e_max, e_rel = H.logits_errors(got, blob[f"v{V}/logits"])
    with capsys.disabled():
        print(f"\n[V={V} fp32] logits max-norm error {e_max:.2e}, per-element relative error (|ref| >= max/4) {e_rel:.2e}")
    assert e_max < FP32_TOL and e_rel < 4 * FP32_TOL


@pytest.mark.parametrize("hidden,heads,layers,ffn,B,L", [
    (384, 3, 6, 1536, 6, 128),     # the benchmark architecture
    (384, 6, 2, 1536, 3, 100),     # heads of 64, ragged tile edges (M = 300)
    (768, 8, 2, 3072, 2, 77),      # heads of 96
    (384, 8, 2, 1536, 3, 90),      # the reference's default block: heads of 48 (8-byte accesses per lane)
    (256, 16, 1, 1024, 2, 40),     # heads of 16
    (1024, 8, 2, 4096, 2, 512),    # stress shape: d = 1024, L = 512
])
def test_fp32_logits_match_oracle(gpu_device, hidden, heads, layers, ffn, B, L):
    cfg = dict(RI.SMALL_CFG, hidden_size=hidden, num_heads=heads, num_layers=layers, ffn_hidden_size=ffn,
               max_position_embeddings=max(256, L), dropout=0.0)
    torch.manual_seed(1)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 9)
    sd = {k: v.clone() for k, v in bb.state_dict().items()}
    ids, attn, t = RI.backbone_inputs(cfg, 4, B=B, L=L)
    with torch.no_grad():
        ref = BR.logits(sd, cfg, torch.from_numpy(ids), torch.from_numpy(t), torch.from_numpy(attn)).numpy()
        ref_h = BR.hidden_states(sd, cfg, torch.from_numpy(ids), torch.from_numpy(t), torch.from_numpy(attn), layer_idx=0).numpy()
        bbg = bb.to(gpu_device)
        bbg.compute_dtype = "fp32"
        args = [torch.from_numpy(x).to(gpu_device) for x in (ids, t, attn)]
        got = bbg(*args).cpu().numpy()
        got_h = bbg.get_hidden_states(*args, layer_idx=0).cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_err(got, ref) < FP32_TOL
    valid = attn.astype(bool)
    assert rel_err(got_h[valid], ref_h[valid]) < FP32_TOL


def test_fp32_fully_padded_sequence_is_finite(gpu_device):
    """Every key of a sequence padded: attention rows come out as zeros, logits stay finite (no NaN)."""
    cfg, bb = tiny_model("tiny_h2")
    bb = bb.to(gpu_device)
    bb.compute_dtype = "fp32"
    ids, attn, t = RI.backbone_inputs(cfg, 5, B=3, L=16)
    attn[1] = 0
    with torch.no_grad():
        got = bb(*(torch.from_numpy(x).to(gpu_device) for x in (ids, t, attn))).cpu().numpy()
    assert np.isfinite(got).all()


def test_gemm_f32_epilogues_and_edges(gpu_device):
    """bd_gemm_f32 against torch fp64 on ragged shapes: plain + bias, erf-GELU, residual in place."""
    lib = _lib.lib()
    g = torch.Generator().manual_seed(3)
    for M, N, K in ((1, 39, 128), (300, 803, 384), (257, 130, 1536)):
        A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
        R = torch.randn(M, N, generator=g)
        Ad, Wd, bd, Rd = (x.to(gpu_device) for x in (A, W, b, R))
        ref = A.double() @ W.double().t() + b.double()
        for epi in (0, 1, 2):
            out = Rd.clone() if epi == 2 else torch.empty(M, N, device=gpu_device)
            rc = lib.bd_gemm_f32(_lib.ptr(Ad), K, _lib.ptr(Wd), K, _lib.ptr(bd), _lib.ptr(out) if epi == 2 else None,
                                 _lib.ptr(out), N, M, N, K, epi, _lib.stream_handle())
            assert rc == 0, lib.bd_last_error()
            want = ref if epi == 0 else (torch.nn.functional.gelu(ref) if epi == 1 else ref + R.double())
            assert rel_err(out.cpu().double().numpy(), want.numpy()) < 2e-6, (M, N, K, epi)
    # argument checks fail loudly
    assert lib.bd_gemm_f32(_lib.ptr(Ad), K, _lib.ptr(Wd), K, None, None, _lib.ptr(out), N, M, N, 10, 0, _lib.stream_handle()) != 0


def test_sampler_runs_in_fp32_mode(gpu_device):
    """The sampler takes full fp32 logits every step in fp32 mode; same seed, same sequences; the output keeps
    the constraints (no MASK left, two stars)."""
    from tests.test_gpu_sampler_e2e import build_model
    T, L = 8, 32
    _, tok = H.make_sampler("smiles")
    model = build_model(tok, gpu_device, T)
    model.backbone.compute_dtype = "fp32"
    outs = []
    for _ in range(2):
        smp, _ = H.make_sampler("smiles", model=model, num_steps=T, device=str(gpu_device))
        smp._core.seed = 77
        smp._core.record = []
        outs.append(smp.sample(16, L, show_progress=False)[0].cpu())
        assert all(r["mode"] == "logits" for r in smp._core.record)
    assert torch.equal(outs[0], outs[1])
    assert not (outs[0] == tok.mask_token_id).any()
    assert ((outs[0] == tok.get_star_token_id()).sum(dim=1) == 2).all()
