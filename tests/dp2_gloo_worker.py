"""Child process of tests/test_zx_dp2_gloo_one_gpu.py: one of TWO data-parallel ranks that share cuda:0 and talk over
gloo (RCCL refuses two ranks on one device). Each rank runs the product's training step on its half of a fixed batch
with the two-bucket gradient exchange started from the backward pass's hook; rank 0 also computes both halves'
gradients by itself, without any exchange, and checks that the exchanged buffer is their mean (what DDP leaves in
.grad, reference trainer_backbone.py:216-219). The transport differs from RCCL's; the order of collectives, the
hook, the streams that feed it and the averaging are the product's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "discrete-diffusion-model-for-inverse-polymer-design_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(rank, port, overlap):
    from bidiff.model.backbone import DiffusionBackbone
    from bidiff.model.diffusion import DiscreteMaskingDiffusion
    from bidiff.parallel import GradientExchange, tail_split_offset
    from bidiff.training import autograd as AG
    from bidiff.training.optimizer import FlatAdamW

    world = 2
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="gloo", init_method="env://")
    try:
        cfg = dict(vocab_size=39, hidden_size=128, num_layers=4, num_heads=2, ffn_hidden_size=256,
                   max_position_embeddings=64, num_diffusion_steps=20, dropout=0.0, pad_token_id=0)
        torch.manual_seed(42 + rank)           # ranks start from DIFFERENT weights: the broadcast must make them equal
        bb = DiffusionBackbone(**cfg)
        model = DiscreteMaskingDiffusion(bb, num_steps=20, mask_token_id=1, pad_token_id=0, bos_token_id=2,
                                         eos_token_id=3).to(dev).train()
        model.corruption_rng = "philox"
        for p_ in model.parameters():
            c_ = p_.data.cpu()
            dist.broadcast(c_, src=0)
            p_.data.copy_(c_)
        opt = FlatAdamW(model.parameters(), lr=3e-4, weight_decay=0.01)
        g = torch.Generator().manual_seed(3)
        B, L = 32, 48
        lens = torch.randint(6, L + 1, (B,), generator=g)
        attn = (torch.arange(L)[None] < lens[:, None]).long()
        ids = torch.randint(5, 39, (B, L), generator=g) * attn
        ids[:, 0] = 2
        ids[torch.arange(B), lens - 1] = 3
        half = B // world

        def part(r):
            sl = slice(r * half, (r + 1) * half)
            return ids[sl].to(dev), attn[sl].to(dev), lens[sl].tolist()

        off = tail_split_offset(opt, bb, 2)
        assert off is not None and 0 < off < opt.flat_grad.numel()
        if not overlap:
            off = opt.flat_grad.numel()         # BD_OVERLAP_ALLREDUCE=0: one blocking all-reduce
        ex = GradientExchange(opt.flat_grad, off, world)
        assert ex.two_buckets == bool(overlap)
        AG.set_tail_ready_hook(ex.tail_ready, 2, model=bb)

        def step(r, exchange):
            opt.zero_grad()
            torch.manual_seed(11)
            i_, a_, l_ = part(r)
            res = model(i_, a_, lengths=l_)
            if exchange:
                ex.arm()
            res["loss"].backward()
            if exchange:
                ex.finish()
            torch.cuda.synchronize()
            return opt.flat_grad.clone()

        steps = 3
        mine = [step(rank, True) for _ in range(steps)]
        assert ex.early_tails == (steps if overlap else 0), f"tail started early {ex.early_tails} times in {steps} steps"
        # every rank holds the same buffer after the exchange
        other = [t.cpu() for t in mine]
        for t in other:
            ref = t.clone()
            dist.broadcast(ref, src=0)
            assert torch.equal(ref, t) or (ref - t).abs().max().item() == 0.0, "ranks hold different gradients after the exchange"
        if rank == 0:
            AG.set_tail_ready_hook(None, model=bb)
            want = 0.5 * (step(0, False) + step(1, False))
            for t in mine:
                rel = ((t - want).norm() / want.norm()).item()
                assert torch.isfinite(t).all() and rel < 1e-4, f"exchanged gradients are not the ranks' mean: rel {rel}"
            head, tail = want[:off], want[off:]
            assert head.abs().sum().item() > 0 and (not overlap or tail.abs().sum().item() > 0)
        dist.barrier()
        print(f"dp2 gloo rank {rank} ok")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
