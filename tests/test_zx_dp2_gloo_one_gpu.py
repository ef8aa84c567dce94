"""Two data-parallel ranks of the training step on ONE GPU over gloo (child processes): the gradient exchange through
the backward pass's hook with real two-rank sums. See the worker's docstring."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [1, 0])
def test_two_ranks_exchange_gradients_through_the_backward_hook(gpu_device, overlap):
    port = 29700 + (os.getpid() % 2000) + 7 * overlap
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "dp2_gloo_worker.py"), str(r), str(port), str(overlap)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=150))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{so[-2000:]}\n{se[-4000:]}"
        assert f"dp2 gloo rank {r} ok" in so
