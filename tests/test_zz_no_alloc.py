"""bd_attention_bwd_static: the attention backward whose scratch (the per-query row sums the two kernels
exchange) is a file-scope device array, for callers that cannot allocate. Called through the C ABI and compared,
bit for bit, with bd_attention_bwd given a caller buffer: two calls in a row, the second one smaller, then a
size beyond the built-in scratch."""
import pytest
import torch

from bidiff import _lib

pytestmark = pytest.mark.gpu


def _case(dev, B, L, H, dh, seed):
    g = torch.Generator().manual_seed(seed)
    d = H * dh
    qkv = (torch.randn(B * L, 3 * d, generator=g) * 0.7).to(dev).bfloat16()
    dout = torch.randn(B * L, d, generator=g).to(dev).bfloat16()
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None] < lens[:, None]).to(torch.uint8).to(dev)
    out = torch.empty(B * L, d, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, H, L, device=dev)
    s = _lib.stream_handle()
    _lib.check(_lib.lib().bd_attention_fwd(_lib.ptr(qkv), _lib.ptr(mask), _lib.ptr(out), _lib.ptr(lse), B, L, H, dh, s), "fwd")
    return qkv, mask, out, dout, lse


def _both(dev, B, L, H, dh, seed):
    lib, s = _lib.lib(), _lib.stream_handle()
    qkv, mask, out, dout, lse = _case(dev, B, L, H, dh, seed)
    d = H * dh
    want = torch.full((B * L, 3 * d), float("nan"), dtype=torch.bfloat16, device=dev)
    delta = torch.empty(B, H, L, device=dev)
    _lib.check(lib.bd_attention_bwd(_lib.ptr(qkv), _lib.ptr(mask), _lib.ptr(out), _lib.ptr(dout), _lib.ptr(lse),
                                    _lib.ptr(delta), _lib.ptr(want), B, L, H, dh, s), "bd_attention_bwd")
    got = torch.full((B * L, 3 * d), float("nan"), dtype=torch.bfloat16, device=dev)
    free_before = torch.cuda.mem_get_info()[0]
    _lib.check(lib.bd_attention_bwd_static(_lib.ptr(qkv), _lib.ptr(mask), _lib.ptr(out), _lib.ptr(dout), _lib.ptr(lse),
                                           _lib.ptr(got), B, L, H, dh, s), "bd_attention_bwd_static")
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before, "the call must not allocate device memory"
    assert torch.isfinite(want.float()).all()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


def test_static_scratch_attention_backward_matches_the_caller_buffer_variant(gpu_device):
    dev = gpu_device
    _both(dev, 64, 128, 3, 128, 1)      # the training shape's head layout
    _both(dev, 5, 40, 2, 64, 2)         # a smaller call right after: the array still holds the first call's sums
    _both(dev, 3, 300, 4, 32, 3)        # several key tiles
    lib = _lib.lib()
    cap = int(lib.bd_attention_bwd_static_capacity())
    assert cap == 1 << 22
    x = torch.zeros(8, dtype=torch.bfloat16, device=dev)
    rc = lib.bd_attention_bwd_static(_lib.ptr(x), None, _lib.ptr(x), _lib.ptr(x), _lib.ptr(x), _lib.ptr(x),
                                     65535, 128, 1, 64, _lib.stream_handle())    # 8.4 M entries > 4.2 M
    assert rc != 0 and b"exceeds the built-in scratch" in lib.bd_last_error()
