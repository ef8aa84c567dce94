"""The data-parallel training step's collectives on the RCCL backend ("nccl" on ROCm), one rank on cuda:0, in a
child process (a process group and RCCL's threads stay out of the test process). See the worker's docstring for
what one rank can and cannot show."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_gradient_exchange_over_rccl_single_rank(gpu_device):
    port = 29600 + (os.getpid() % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-u", os.path.join(HERE, "rccl_single_rank_worker.py"), str(port)],
                         capture_output=True, text=True, timeout=110, env=env)
    assert res.returncode == 0, f"worker failed:\n{res.stdout[-2000:]}\n{res.stderr[-4000:]}"
    assert "rccl single-rank exchange ok" in res.stdout
