"""The RCCL code paths of zkp_hip/dist.py on real hardware with the one GPU a test box has: a one-rank `nccl` process group,
with the four-step transform forced through it (asynchronous all_to_all_single, waits on the compute stream, four column chunks)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_four_step_and_msm_exchange_through_a_one_rank_rccl_group():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_worker.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "OK rccl one-rank" in p.stdout
