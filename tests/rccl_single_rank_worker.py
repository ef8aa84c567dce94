"""Child process of tests/test_zy_rccl_single_rank.py: the training step's gradient exchange and the bench's
collectives over the RCCL backend with ONE rank on cuda:0.

A one-GPU box cannot host two RCCL ranks, so the transport between GPUs is not what this covers. It covers what
the library's side of a data-parallel step can get wrong on real hardware: the asynchronous all-reduce of the
flat buffer's tail is started by a hook INSIDE the backward pass, on the stream that completes those gradients,
while the lower layers' kernels keep writing the head of the same buffer on other streams; finish() reduces the
head and makes the optimizer's stream wait for both collectives. With one rank every sum is the identity, so the
exchanged gradients must equal the ones of a step without any exchange (to rounding: they are sums of float
atomics, so two runs of the same step differ in the last bits)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "discrete-diffusion-model-for-inverse-polymer-design_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(port):
    from bidiff.model.backbone import DiffusionBackbone
    from bidiff.model.diffusion import DiscreteMaskingDiffusion
    from bidiff.parallel import GradientExchange, allreduce_mean_, tail_split_offset
    from bidiff.training import autograd as AG
    from bidiff.training.optimizer import FlatAdamW

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
    try:
        assert dist.get_backend() == "nccl"

        class OneRankExchange(GradientExchange):
            def _on(self):                   # the product class is inert at world size 1; here the calls are the point
                return True

        cfg = dict(vocab_size=803, hidden_size=384, num_layers=6, num_heads=3, ffn_hidden_size=1536,
                   max_position_embeddings=256, num_diffusion_steps=50, dropout=0.0, pad_token_id=0)
        torch.manual_seed(42)
        bb = DiffusionBackbone(**cfg)
        model = DiscreteMaskingDiffusion(bb, num_steps=50, mask_token_id=1, pad_token_id=0, bos_token_id=2,
                                         eos_token_id=3).to(dev).train()
        model.corruption_rng = "philox"
        for p_ in model.parameters():          # what rank 0 does for the other ranks before the first step
            dist.broadcast(p_.data, src=0)
        opt = FlatAdamW(model.parameters(), lr=3e-4, weight_decay=0.01)
        g = torch.Generator().manual_seed(3)
        B, L = 64, 128
        lens = torch.randint(12, L + 1, (B,), generator=g)
        attn = (torch.arange(L)[None] < lens[:, None]).long()
        ids = torch.randint(5, 803, (B, L), generator=g) * attn
        ids[:, 0] = 2
        ids[torch.arange(B), lens - 1] = 3
        ids_d, attn_d = ids.to(dev), attn.to(dev)
        off = tail_split_offset(opt, bb, 3)
        assert off is not None and 0 < off < opt.flat_grad.numel()

        def grads(exchange):
            out = []
            for _ in range(3):                 # several steps back to back: the collectives of one step must not
                opt.zero_grad()                # reach into the next one's gradients
                torch.manual_seed(11)          # same timesteps and corruption each time
                res = model(ids_d, attn_d, lengths=lens.tolist())
                if exchange is not None:
                    exchange.arm()
                res["loss"].backward()
                if exchange is not None:
                    exchange.finish()
                out.append(opt.flat_grad.clone())
            torch.cuda.synchronize()
            return out

        plain = grads(None)
        ex = OneRankExchange(opt.flat_grad, off, 1)
        fired = []

        def hook():
            fired.append(1)
            ex.tail_ready()

        AG.set_tail_ready_hook(hook, 3)
        try:
            with_ex = grads(ex)
        finally:
            AG.set_tail_ready_hook(None)
        assert len(fired) == 3, f"tail hook fired {len(fired)} times in 3 steps"
        for a, b in zip(plain, with_ex):
            assert torch.isfinite(b).all()
            # gradients are sums of float atomics, so two runs of the SAME step agree to rounding only
            rel = ((a - b).norm() / a.norm()).item()
            assert rel < 1e-4, f"exchanged gradients differ from the plain step's: rel {rel}"
        # the whole-buffer path (no tail reported) and the plain helper
        ex.arm()
        before = opt.flat_grad.clone()
        ex.finish()
        assert torch.equal(before, opt.flat_grad)
        allreduce_mean_(opt.flat_grad, 1)
        assert torch.equal(before, opt.flat_grad)
        # the bench's own collectives: MAX of a float64 time, SUM of a float64 token count, barrier
        tm = torch.tensor([1.25], device=dev, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        tk = torch.tensor([10153.0], device=dev, dtype=torch.float64)
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        dist.barrier()
        torch.cuda.synchronize()
        assert tm.item() == 1.25 and tk.item() == 10153.0
        opt.step(max_norm=1.0)
        torch.cuda.synchronize()
        print("rccl single-rank exchange ok")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]))
