"""GPU parity of the product modules (DiffusionBackbone / DiscreteMaskingDiffusion on the HIP
library) against the golden vectors captured from the reference and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import replay_inputs as RI
from oracle import backbone_ref as BR
from oracle import sampler_ref as ora
from bidiff.model.backbone import DiffusionBackbone
from bidiff.model.diffusion import DiscreteMaskingDiffusion

pytestmark = pytest.mark.gpu
BLOB = np.load(os.path.join(H.GOLDEN, "model.npz"))
BF16_TOL = 2e-2   # relative error bound stated for bf16 logits


def tiny_model(name):
    cfg = RI.BACKBONE_CFGS[name]
    torch.manual_seed(7)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 11)
    return cfg, bb


def rel_err(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("name", list(RI.BACKBONE_CFGS))
def test_backbone_logits_match_reference_golden(gpu_device, name):
    cfg, bb = tiny_model(name)
    bb = bb.to(gpu_device)
    L = 24 if name != "tiny_h4" else 150
    ids, attn, t = RI.backbone_inputs(cfg, 21, B=5, L=L)
    ti, ta, tt = (torch.from_numpy(x).to(gpu_device) for x in (ids, attn, t))
    with torch.no_grad():
        got = bb(ti, tt, ta).float().cpu().numpy()
        got_nm = bb(ti, tt, None).float().cpu().numpy()
        got_2d = bb(ti, tt.unsqueeze(-1), ta).float().cpu().numpy()
    assert got.shape == (5, L, cfg["vocab_size"]) and np.isfinite(got).all()
    assert rel_err(got, BLOB[f"{name}/logits_masked"]) < BF16_TOL
    assert rel_err(got_nm, BLOB[f"{name}/logits_nomask"]) < BF16_TOL
    assert rel_err(got_2d, BLOB[f"{name}/logits_t2d"]) < BF16_TOL
    with torch.no_grad():
        pooled = bb.get_pooled_output(ti[:3], tt[:3], ta[:3], "mean").cpu().numpy()
        pmax = bb.get_pooled_output(ti[:3], tt[:3], ta[:3], "max").cpu().numpy()
        pcls = bb.get_pooled_output(ti[:3], tt[:3], ta[:3], "cls").cpu().numpy()
    assert rel_err(pooled, BLOB[f"{name}/pooled_mean"]) < BF16_TOL
    assert rel_err(pmax, BLOB[f"{name}/pooled_max"]) < BF16_TOL
    assert rel_err(pcls, BLOB[f"{name}/pooled_cls"]) < BF16_TOL


def test_small_config_logits_match_oracle(gpu_device):
    """The benchmark architecture (d384 / 6 layers / 3 heads of 128) and the 6-head variant."""
    for heads in (3, 6):
        cfg = dict(RI.SMALL_CFG, num_heads=heads, dropout=0.0)
        torch.manual_seed(42)
        bb = DiffusionBackbone(**cfg).eval()
        RI.perturb_state_dict(bb.state_dict(), 5)
        sd = {k: v.clone() for k, v in bb.state_dict().items()}
        ids, attn, t = RI.backbone_inputs(cfg, 3, B=6, L=128)
        with torch.no_grad():
            ref = BR.logits(sd, cfg, torch.from_numpy(ids), torch.from_numpy(t), torch.from_numpy(attn)).numpy()
            bbg = bb.to(gpu_device)
            got = bbg(torch.from_numpy(ids).to(gpu_device), torch.from_numpy(t).to(gpu_device),
                      torch.from_numpy(attn).to(gpu_device)).cpu().numpy()
        assert rel_err(got, ref) < BF16_TOL


@pytest.mark.parametrize("V", [39, 803])
def test_benchmark_architecture_matches_reference_golden(gpu_device, V, capsys):
    """d 384, 6 layers, 3 heads of 128, L = 128 at V = 39 / 803 against vectors the reference produced at that size
    (model_small.npz): logits in bf16 within 2e-2 of the largest logit, per-element relative error of the logits
    above a quarter of the largest within 8e-2 (what the max-norm bound implies there; reported, smaller in
    practice); the training forward's corruption bit for bit and its loss."""
    blob = np.load(os.path.join(H.GOLDEN, "model_small.npz"))
    cfg, ids, attn, t, stride = RI.small_arch_case(V)
    torch.manual_seed(42)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 5)
    dm = DiscreteMaskingDiffusion(bb, num_steps=100, mask_token_id=1, pad_token_id=0, bos_token_id=2,
                                  eos_token_id=3).to(gpu_device).eval()
    dev = gpu_device
    with torch.no_grad():
        got = dm.backbone(torch.from_numpy(ids).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(attn).to(dev))
    got = got.float().cpu().numpy()[:, ::stride]
    e_max, e_rel = H.logits_errors(got, blob[f"v{V}/logits"])
    with capsys.disabled():
        print(f"\n[V={V} bf16] logits max-norm error {e_max:.2e}, per-element relative error (|ref| >= max/4) {e_rel:.2e}")
    assert e_max < BF16_TOL and e_rel < 4 * BF16_TOL
    # the training forward with the reference's draws
    ids2 = blob[f"v{V}/fwd_ids"]
    torch.manual_seed(78)
    tt = torch.randint(1, 101, (2,))
    u = torch.rand(2, 128)
    with torch.no_grad():
        noisy, ind = dm.forward_process(torch.from_numpy(ids2).to(dev), tt.to(dev), torch.from_numpy(attn).to(dev),
                                        uniforms=u.to(dev))
        lg = dm.backbone(noisy, tt.to(dev), torch.from_numpy(attn).to(dev))
        loss = dm._compute_loss(lg, torch.from_numpy(ids2).to(dev), ind, torch.from_numpy(attn).to(dev))
    np.testing.assert_array_equal(noisy.cpu().numpy(), blob[f"v{V}/fwd_noisy_ids"])
    np.testing.assert_array_equal(ind.cpu().numpy(), blob[f"v{V}/fwd_mask_indicator"])
    e_max, e_rel = H.logits_errors(lg.float().cpu().numpy()[:, ::stride], blob[f"v{V}/fwd_logits"])
    assert e_max < BF16_TOL and e_rel < 4 * BF16_TOL
    assert abs(float(loss.item()) - float(blob[f"v{V}/fwd_loss"])) < BF16_TOL * float(blob[f"v{V}/fwd_loss"])


@pytest.mark.parametrize("hidden,heads,layers,ffn,B,L", [
    (384, 8, 2, 1536, 4, 96),      # the reference's default model block: 8 heads of 48 (padded to 64 inside)
    (256, 4, 2, 1024, 3, 128),     # the reference's "tiny" size: fused kernels with head_dim 64
    (1024, 16, 2, 4096, 2, 512),   # scaled-up stress shape (d = 1024, L = 512): unfused kernels
    (384, 3, 2, 1536, 3, 200),     # small backbone beyond 128 tokens: fused feed-forward, unfused attention
    (768, 6, 2, 3072, 40, 128),    # the reference's "medium" size (config.yaml model_sizes), 5120 rows
    (768, 6, 3, 3072, 88, 128),    # medium, 11264 rows: enough tiles for the eight-wave GEMM
    (1152, 9, 2, 4608, 60, 128),   # its "large" size: widths that are odd multiples of 128, 7680 rows (eight-wave GEMM)
    (1536, 12, 2, 6144, 48, 128),  # its "xl" size, 6144 rows (eight-wave GEMM on every projection)
])
def test_other_architectures_match_oracle(gpu_device, hidden, heads, layers, ffn, B, L):
    cfg = dict(RI.SMALL_CFG, hidden_size=hidden, num_heads=heads, num_layers=layers, ffn_hidden_size=ffn,
               max_position_embeddings=max(256, L), dropout=0.0)
    torch.manual_seed(1)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 9)
    sd = {k: v.clone() for k, v in bb.state_dict().items()}
    ids, attn, t = RI.backbone_inputs(cfg, 4, B=B, L=L)
    with torch.no_grad():
        ref = BR.logits(sd, cfg, torch.from_numpy(ids), torch.from_numpy(t), torch.from_numpy(attn)).numpy()
        bbg = bb.to(gpu_device)
        got = bbg(torch.from_numpy(ids).to(gpu_device), torch.from_numpy(t).to(gpu_device),
                  torch.from_numpy(attn).to(gpu_device)).cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_err(got, ref) < BF16_TOL
    if (hidden, heads, L) == (384, 8, 96):     # the padded heads also take the single-launch forward
        from bidiff import _lib
        import ctypes as C
        assert _lib.lib().bd_backbone_mega_supported(C.byref(bbg.packed().desc), L) == 1


@pytest.mark.parametrize("hidden,heads,layers,ffn,B,L", [(768, 6, 3, 3072, 88, 128), (1152, 9, 3, 4608, 60, 128),
                                                        (1024, 8, 3, 4096, 24, 512)])
def test_deferred_layernorm_forward_against_layernorm_kernels(gpu_device, hidden, heads, layers, ffn, B, L):
    """defer_layernorm = True runs LayerNorm deferred across the residual projections of large models; the default keeps
    a LayerNorm kernel in front of every projection. Same logits within the bf16 noise of either (they are not
    bit-equal: the rows are rounded before, not after, the normalisation), ragged key masks included."""
    from bidiff import _lib
    cfg = dict(RI.SMALL_CFG, hidden_size=hidden, num_heads=heads, num_layers=layers, ffn_hidden_size=ffn,
               max_position_embeddings=max(256, L), dropout=0.0)
    torch.manual_seed(3)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 9)
    bb = bb.to(gpu_device)
    M = B * L
    lib = _lib.lib()
    assert lib.bd_gemm_deferred_ln_supported(M, hidden, hidden) == 1 and lib.bd_gemm_deferred_ln_supported(M, hidden, ffn) == 1
    ids, attn, t = RI.backbone_inputs(cfg, 4, B=B, L=L)
    ti, ta, tt = (torch.from_numpy(x).to(gpu_device) for x in (ids, attn, t))
    with torch.no_grad():
        b = bb(ti, tt, ta).float()
        bb.defer_layernorm = True
        a = bb(ti, tt, ta).float()
        bb.defer_layernorm = False
        b2 = bb(ti, tt, ta).float()
        bb.defer_layernorm = True
        a2 = bb(ti, tt, ta).float()
    assert torch.isfinite(a).all() and torch.equal(a, a2) and torch.equal(b, b2)
    assert not torch.equal(a, b)                       # the deferred path did run
    assert float((a - b).abs().max() / b.abs().max()) < BF16_TOL / 2


def test_stress_backbone_at_full_depth_matches_oracle(gpu_device):
    """The stress configuration as bench.py runs it: 24 layers, d = 1024, ffn 4096, 8 heads of 128, L = 512, bf16
    (LayerNorm + tiled GEMM + stand-alone attention kernels), two sequences (one ragged) against the fp32 oracle."""
    cfg = dict(RI.SMALL_CFG, hidden_size=1024, num_heads=8, num_layers=24, ffn_hidden_size=4096,
               max_position_embeddings=512, dropout=0.0)
    torch.manual_seed(1)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 9)
    sd = {k: v.clone() for k, v in bb.state_dict().items()}
    ids, attn, t = (x[[0, 5]].copy() for x in RI.backbone_inputs(cfg, 4, B=6, L=512))
    attn[1, 300:] = 0
    ids[1, 300:] = 0
    ids[1, 299] = 3
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = BR.logits(sd, cfg, torch.from_numpy(ids), torch.from_numpy(t), torch.from_numpy(attn)).numpy()
        bbg = bb.to(gpu_device)
        got = bbg(torch.from_numpy(ids).to(gpu_device), torch.from_numpy(t).to(gpu_device),
                  torch.from_numpy(attn).to(gpu_device)).cpu().numpy()
    e_max, e_rel = H.logits_errors(got, ref)
    print(f"\n[24 layers, d=1024, 8x128, L=512 bf16] logits max-norm error {e_max:.2e}, per-element (|ref| >= max/4) {e_rel:.2e}")
    assert np.isfinite(got).all()
    assert e_max < BF16_TOL


def test_intermediate_hidden_states_match_oracle(gpu_device):
    """get_hidden_states(layer_idx): residual stream after block i, no final norm (-1: final norm)."""
    cfg = dict(RI.SMALL_CFG, num_layers=3, dropout=0.0)
    torch.manual_seed(2)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 13)
    sd = {k: v.clone() for k, v in bb.state_dict().items()}
    ids, attn, t = (torch.from_numpy(x) for x in RI.backbone_inputs(cfg, 8, B=5, L=64))
    bbg = bb.to(gpu_device)
    with torch.no_grad():
        for idx in (0, 1, 2, -1):
            ref = BR.hidden_states(sd, cfg, ids, t, attn, layer_idx=idx).numpy()
            got = bbg.get_hidden_states(ids.to(gpu_device), t.to(gpu_device), attn.to(gpu_device), layer_idx=idx)
            assert got.dtype == torch.float32 and tuple(got.shape) == ref.shape
            valid = attn.bool().numpy()           # padded rows are not specified
            assert rel_err(got.cpu().numpy()[valid], ref[valid]) < BF16_TOL, idx
        with pytest.raises(IndexError):
            bbg.get_hidden_states(ids.to(gpu_device), t.to(gpu_device), attn.to(gpu_device), layer_idx=3)


def test_single_launch_forward_equals_per_layer_kernels(gpu_device):
    """The one-kernel forward (rows in registers across all layers) and the per-layer fused kernels run
    the same phases on the same operands: the final hidden states agree to bf16 rounding."""
    from bidiff import _lib
    import ctypes as C
    cfg = dict(RI.SMALL_CFG, num_layers=3, dropout=0.0)
    torch.manual_seed(2)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 13)
    ids, attn, t = (torch.from_numpy(x).to(gpu_device) for x in RI.backbone_inputs(cfg, 8, B=7, L=100))
    bbg = bb.to(gpu_device)
    with torch.no_grad():
        assert _lib.lib().bd_backbone_mega_supported(C.byref(bbg.packed().desc), 100) == 1
        one = bbg.get_hidden_states(ids, t, attn)                         # single launch + final norm inside
        resid = bbg.get_hidden_states(ids, t, attn, layer_idx=2)          # per-layer kernels, residual stream
        per_layer = torch.nn.functional.layer_norm(resid, (cfg["hidden_size"],), bbg.final_norm.weight,
                                                   bbg.final_norm.bias, bbg.final_norm.eps)
    valid = attn.bool()
    a, b = one[valid], per_layer[valid]
    assert torch.isfinite(a).all()
    assert (a - b).abs().max().item() <= 2.0 ** -7 * b.abs().max().item()    # one bf16 ulp of the largest value
    # not supported: longer sequences fall back to the per-layer path
    assert _lib.lib().bd_backbone_mega_supported(C.byref(bbg.packed().desc), 129) == 0


@pytest.mark.parametrize("hidden,heads,ffn", [(384, 3, 1536), (384, 8, 1536), (256, 4, 1024), (128, 4, 512)])
def test_short_sequences_sharing_a_workgroup_change_nothing(gpu_device, hidden, heads, ffn, monkeypatch):
    """Up to 64 (32) tokens, two (four) sequences share a workgroup of the single-launch forward, side by side in the
    key tiles and hidden from each other by per-wave mask tables. Every attended row must come out with the bits of
    the one-sequence-per-workgroup launch (BD_MEGA_PACK=0), for batch sizes that leave the last workgroup
    partly empty, ragged key masks included; one case is also held against the oracle."""
    from bidiff import _lib
    import ctypes as C
    cfg = dict(RI.SMALL_CFG, hidden_size=hidden, num_heads=heads, num_layers=2, ffn_hidden_size=ffn, dropout=0.0)
    torch.manual_seed(3)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 21)
    sd = {k: v.clone() for k, v in bb.state_dict().items()}
    bbg = bb.to(gpu_device)
    for L in (17, 24, 32, 33, 64, 65):
        for B in (1, 5, 8):
            ids, attn, t = (torch.from_numpy(x) for x in RI.backbone_inputs(cfg, 100 + L + B, B=B, L=L))
            assert _lib.lib().bd_backbone_mega_supported(C.byref(bbg.packed().desc), L) == 1
            dev = [x.to(gpu_device) for x in (ids, t, attn)]
            with torch.no_grad():
                monkeypatch.setenv("BD_MEGA_PACK", "0")
                alone = bbg.get_hidden_states(*dev).cpu()
                monkeypatch.delenv("BD_MEGA_PACK")
                shared = bbg.get_hidden_states(*dev).cpu()
                nomask = bbg.get_hidden_states(dev[0], dev[1], None).cpu()
                monkeypatch.setenv("BD_MEGA_PACK", "0")
                nomask_alone = bbg.get_hidden_states(dev[0], dev[1], None).cpu()
                monkeypatch.delenv("BD_MEGA_PACK")
            valid = attn.bool()
            assert torch.isfinite(shared[valid]).all()
            assert torch.equal(shared[valid], alone[valid]), (L, B)
            assert torch.equal(nomask, nomask_alone), (L, B)
            if (L, B) == (33, 5):
                ref = BR.hidden_states(sd, cfg, ids, t, attn, layer_idx=-1).numpy()
                assert rel_err(shared.numpy()[valid.numpy()], ref[valid.numpy()]) < BF16_TOL


@pytest.mark.parametrize("hidden,heads,ffn", [(384, 3, 1536), (384, 8, 1536), (128, 4, 512)])
def test_ragged_batches_deal_sequences_to_waves(gpu_device, hidden, heads, ffn):
    """A batch padded to its longest row, sequences dealt to the 32-row waves of the single-launch forward by their
    own lengths (bd_backbone_forward_ragged): every sequence sits in exactly one run of adjacent waves, and every
    row inside its waves carries the bits of the ordinary launch; rows beyond them keep what the buffer held."""
    cfg = dict(RI.SMALL_CFG, hidden_size=hidden, num_heads=heads, num_layers=2, ffn_hidden_size=ffn, dropout=0.0)
    torch.manual_seed(4)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 22)
    bbg = bb.to(gpu_device)
    rs = np.random.RandomState(5)
    for L, B in ((100, 37), (128, 64), (64, 21), (40, 9)):
        lens = rs.randint(0, L + 1, size=B)
        lens[0], lens[1] = L, 1
        attn = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
        attn[2, : lens[2] // 2] = 0                      # a mask that is not a prefix: its last visible row counts
        ids = rs.randint(5, cfg["vocab_size"], size=(B, L)).astype(np.int64) * attn
        t = rs.randint(1, cfg["num_diffusion_steps"] + 1, size=B).astype(np.int64)
        ids_d, t_d, attn_d = (torch.from_numpy(x).to(gpu_device) for x in (ids, t, attn))
        km = (attn_d != 0).to(torch.uint8)
        table = bbg.ragged_wave_table(lens.tolist(), L, gpu_device)
        assert table is not None and table.dtype == torch.int32 and table.shape[1] == 4
        tb = table.cpu().numpy()
        assert tb.shape[0] < (B if L > 64 else -(-B // 2))
        waves = np.zeros(B, dtype=np.int64)
        for row in tb:
            runs = [s for i, s in enumerate(row) if s >= 0 and (i == 0 or row[i - 1] != s)]
            assert len(runs) == len(set(runs))           # the waves of a sequence are adjacent
            for s in row[row >= 0]:
                waves[s] += 1
        assert (waves == np.maximum(1, -(-lens // 32))).all()     # every sequence once, with the waves its length needs
        with torch.no_grad():
            want, want_logits = bbg.run_forward(ids_d, t_d, km, want_hidden=True, want_logits=True)
            got = torch.full_like(want, 7.0)
            _, got_logits = bbg.run_forward(ids_d, t_d, km, want_hidden=True, want_logits=True, hidden_out=got,
                                            wave_table=table)
            only_logits = bbg.run_forward(ids_d, t_d, km, want_hidden=False, want_logits=True, wave_table=table)[1]
        inside = torch.from_numpy(np.arange(L)[None] < (32 * waves)[:, None]).to(gpu_device)
        assert torch.equal(got[inside], want[inside]), (L, B)
        assert (got[~inside] == 7.0).all()
        assert torch.equal(got_logits[inside], want_logits[inside])
        assert torch.equal(only_logits[inside], want_logits[inside]) and (only_logits[~inside] == 0).all()
    # a batch of full-length sequences gains nothing: no table
    assert bbg.ragged_wave_table([128] * 16, 128, gpu_device) is None
    # the C entry point refuses a missing table, an empty one and lengths the single launch does not take
    from bidiff import _lib
    import ctypes as C
    lib, pk = _lib.lib(), bbg.packed()
    hid = torch.empty(B, L, hidden, dtype=torch.bfloat16, device=gpu_device)
    args = lambda tab, nb, L_: (C.byref(pk.desc), _lib.ptr(ids_d), _lib.ptr(t_d), _lib.ptr(km), tab, nb, B, L_,
                                _lib.ptr(hid), _lib.stream_handle())
    assert lib.bd_backbone_mega_ragged(*args(None, 3, L)) != 0
    assert b"wave table" in lib.bd_last_error()
    assert lib.bd_backbone_mega_ragged(*args(_lib.ptr(table), 0, L)) != 0
    assert lib.bd_backbone_mega_ragged(*args(_lib.ptr(table), int(table.shape[0]), 200)) != 0
    assert b"unsupported" in lib.bd_last_error()
    assert lib.bd_backbone_mega_ragged_supported(C.byref(pk.desc), 200) == 0
    # table entries that name no sequence of the batch are treated as empty waves
    bad = table.clone()
    bad[0] = torch.tensor([-7, B, B + 5, 2 ** 30], dtype=torch.int32)
    torch.cuda.synchronize()
    assert lib.bd_backbone_mega_ragged(*args(_lib.ptr(bad), int(bad.shape[0]), L)) == 0
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="wave_table"):
        bbg.run_forward(ids_d, t_d, km, wave_table=table.to(torch.int64))


def test_cpu_tensors_fail_loudly():
    cfg, bb = tiny_model("tiny_h2")
    ids, attn, t = RI.backbone_inputs(cfg, 21)
    with pytest.raises(RuntimeError):
        bb(torch.from_numpy(ids), torch.from_numpy(t), torch.from_numpy(attn))


@pytest.mark.parametrize("tag,bos,eos,use_attn", [("a", 2, 3, True), ("b", 2, 3, False), ("c", None, None, True)])
def test_forward_process_bit_exact_vs_reference(gpu_device, tag, bos, eos, use_attn):
    ids, attn, t = BLOB["fp/ids"], BLOB["fp/attn"], BLOB["fp/t"]
    m = DiscreteMaskingDiffusion(torch.nn.Identity(), num_steps=50, mask_token_id=1, pad_token_id=0,
                                 bos_token_id=bos, eos_token_id=eos).to(gpu_device)
    noisy, ind = m.forward_process(torch.from_numpy(ids).to(gpu_device), torch.from_numpy(t).to(gpu_device),
                                   torch.from_numpy(attn).to(gpu_device) if use_attn else None,
                                   uniforms=torch.from_numpy(BLOB[f"fp/{tag}_u"]).to(gpu_device))
    assert noisy.dtype == torch.long and ind.dtype == torch.bool
    np.testing.assert_array_equal(noisy.cpu().numpy(), BLOB[f"fp/{tag}_noisy"])
    np.testing.assert_array_equal(ind.cpu().numpy(), BLOB[f"fp/{tag}_ind"])


def test_forward_process_philox_matches_oracle_at_scale(gpu_device):
    """Counter-based stream: bit-exact against the oracle on a ragged batch, plus the size-
    independent properties at a large batch (never touches PAD/BOS/EOS, rate follows the schedule)."""
    rs = np.random.RandomState(0)
    B, L, T = 64, 128, 50
    ids = rs.randint(5, 803, size=(B, L)).astype(np.int64)
    lens = rs.randint(2, L + 1, size=B)
    attn = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
    ids[attn == 0] = 0
    ids[:, 0] = 2
    ids[np.arange(B), lens - 1] = 3
    t = rs.randint(1, T + 1, size=B).astype(np.int64)
    m = DiscreteMaskingDiffusion(torch.nn.Identity(), num_steps=T, mask_token_id=1, pad_token_id=0,
                                 bos_token_id=2, eos_token_id=3).to(gpu_device)
    m.corruption_rng = "philox"
    torch.manual_seed(5)
    seed = int(torch.randint(0, 2**62, (1,)).item())
    torch.manual_seed(5)
    noisy, ind = m.forward_process(torch.from_numpy(ids).to(gpu_device), torch.from_numpy(t).to(gpu_device),
                                   torch.from_numpy(attn).to(gpu_device))
    o_noisy, o_ind = ora.corrupt(ids, attn, t, m.mask_schedule.cpu().numpy(), None, 1, 0, 2, 3, seed=seed)
    np.testing.assert_array_equal(noisy.cpu().numpy(), o_noisy)
    np.testing.assert_array_equal(ind.cpu().numpy(), o_ind)
    # large batch properties
    B2 = 4096
    ids2 = torch.randint(5, 803, (B2, L), device=gpu_device)
    ids2[:, 0] = 2
    ids2[:, -1] = 3
    t2 = torch.full((B2,), 25, device=gpu_device)
    n2, i2 = m.forward_process(ids2, t2, torch.ones_like(ids2))
    assert not i2[:, 0].any() and not i2[:, -1].any()
    assert torch.equal(n2[~i2], ids2[~i2]) and (n2[i2] == 1).all()
    rate = i2[:, 1:-1].float().mean().item()
    assert abs(rate - float(m.mask_schedule[25])) < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("d,heads,ffn", [(128, 2, 256), (384, 3, 1536), (256, 8, 512)])
def test_block_level_modules_are_callable_and_match_the_fp32_restatement(gpu_device, d, heads, ffn):
    """MultiHeadAttention / FeedForward / TransformerBlock forward on their own (reference backbone.py:32-78,
    95-101, 120-129): bf16 operands, so 2e-2 of the fp32 restatement; a fully padded sequence gets exactly the
    output bias from the attention; stacking the blocks reproduces the backbone's hidden states."""
    import torch.nn.functional as F
    dev = gpu_device
    torch.manual_seed(5)
    bb = DiffusionBackbone(39, d, 2, heads, ffn, 64, 10, 0.1, 0)
    RI.perturb_state_dict(bb.state_dict(), 9)
    sd = {k: v.clone() for k, v in bb.state_dict().items()}
    bb = bb.to(dev).eval()
    B, L = 5, 37
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, L, d, generator=g)
    mask = torch.ones(B, L, dtype=torch.long)
    mask[1, 20:] = 0
    mask[3, :] = 0                                             # fully padded
    blk = bb.layers[0]
    rel = lambda a, b: ((a.cpu() - b).abs().max() / b.abs().max()).item()
    with torch.no_grad():
        att = blk.attn(x.to(dev), mask.to(dev))
        att_ref = BR.attention(x, sd, "layers.0.attn", heads, mask)
        assert att.shape == (B, L, d) and att.dtype == torch.float32
        assert rel(att, att_ref) < 2e-2
        assert torch.allclose(att[3].cpu(), sd["layers.0.attn.out_proj.bias"].expand(L, d), atol=1e-6)
        assert rel(blk.attn(x.to(dev)), BR.attention(x, sd, "layers.0.attn", heads, None)) < 2e-2
        ff = blk.ffn(x.to(dev))
        ff_ref = F.linear(F.gelu(F.linear(x, sd["layers.0.ffn.fc1.weight"], sd["layers.0.ffn.fc1.bias"])),
                          sd["layers.0.ffn.fc2.weight"], sd["layers.0.ffn.fc2.bias"])
        assert rel(ff, ff_ref) < 2e-2
        y = blk(x.to(dev), mask.to(dev))
        h = F.layer_norm(x, (d,), sd["layers.0.norm1.weight"], sd["layers.0.norm1.bias"], 1e-5)
        y_ref = x + BR.attention(h, sd, "layers.0.attn", heads, mask)
        h = F.layer_norm(y_ref, (d,), sd["layers.0.norm2.weight"], sd["layers.0.norm2.bias"], 1e-5)
        y_ref = y_ref + F.linear(F.gelu(F.linear(h, sd["layers.0.ffn.fc1.weight"], sd["layers.0.ffn.fc1.bias"])),
                                 sd["layers.0.ffn.fc2.weight"], sd["layers.0.ffn.fc2.bias"])
        assert rel(y, y_ref) < 2e-2
        assert torch.equal(x, x.clone()) and y.data_ptr() != x.data_ptr()
        # the blocks one after another on the embedded input = the backbone's hidden states after the last block
        ids = torch.randint(5, 39, (B, L), generator=g)
        t = torch.randint(1, 11, (B,), generator=g)
        pos = torch.arange(L)
        emb = sd["token_embedding.weight"][ids] + sd["position_embedding.weight"][pos][None] + sd["timestep_embedding.weight"][t][:, None]
        z = emb.to(dev)
        for layer in bb.layers:
            z = layer(z, mask.to(dev))
        want = bb.get_hidden_states(ids.to(dev), t.to(dev), mask.to(dev), layer_idx=len(bb.layers) - 1)
        live = mask.bool()
        assert ((z.cpu() - want.cpu()).abs()[live].max() / want.cpu().abs()[live].max()).item() < 2e-2
    # a parameter update is picked up (the bf16 copies are rebuilt)
    with torch.no_grad():
        before = blk.ffn(x.to(dev)).clone()
        blk.ffn.fc2.bias.add_(1.0)
        after = blk.ffn(x.to(dev))
        assert torch.allclose(after, before + 1.0, atol=1e-5)
    # inference only: under autograd with trainable parameters the stand-alone forward refuses
    with pytest.raises(NotImplementedError, match="inference-only"):
        blk(x.to(dev), mask.to(dev))
    with pytest.raises(RuntimeError, match="ROCm device"):
        with torch.no_grad():
            blk(x, mask)


@pytest.mark.parametrize("case", ["fc1_beyond_f16", "fc2_below_f16_normal", "in_range"])
def test_feed_forward_weights_outside_f16_keep_the_bf16_kernels(gpu_device, case):
    """The fused feed-forward phase works in f16 behind the first product. Weights it cannot serve (first-product
    outputs that may pass +-65504; second-layer weights below the smallest normal f16) are recognised when the
    weights are packed, that model runs the bf16 kernels, and the logits stay within the bf16 bound of the oracle."""
    from bidiff import _lib
    import ctypes as C
    cfg = dict(RI.SMALL_CFG, num_layers=2, dropout=0.0)
    torch.manual_seed(4)
    bb = DiffusionBackbone(**cfg).eval()
    RI.perturb_state_dict(bb.state_dict(), 6)
    with torch.no_grad():
        if case == "fc1_beyond_f16":
            bb.layers[1].ffn.fc1.weight.mul_(30000.0)      # pre-activations up to about 1e5
            bb.layers[1].ffn.fc2.weight.mul_(1.0 / 30000.0)
        elif case == "fc2_below_f16_normal":
            bb.layers[0].ffn.fc2.weight.mul_(2e-4)         # every entry below 6.1e-5
    sd = {k: v.clone() for k, v in bb.state_dict().items()}
    ids, attn, t = RI.backbone_inputs(cfg, 12, B=6, L=128)
    with torch.no_grad():
        ref = BR.logits(sd, cfg, torch.from_numpy(ids), torch.from_numpy(t), torch.from_numpy(attn)).numpy()
        bbg = bb.to(gpu_device)
        got = bbg(torch.from_numpy(ids).to(gpu_device), torch.from_numpy(t).to(gpu_device),
                  torch.from_numpy(attn).to(gpu_device)).cpu().numpy()
    pk = bbg.packed()
    fused = [bool(pk.layers[i].ffn_pack) for i in range(2)]
    assert fused == {"fc1_beyond_f16": [True, False], "fc2_below_f16_normal": [False, True], "in_range": [True, True]}[case]
    assert _lib.lib().bd_backbone_mega_supported(C.byref(pk.desc), 128) == int(case == "in_range")
    assert np.isfinite(got).all()
    assert rel_err(got, ref) < BF16_TOL
