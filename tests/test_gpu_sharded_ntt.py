"""The multi-GPU Fr transform INSIDE the library (zkp_ntt_fr_sharded_dev / zkp_ntt_fr_sharded, BASELINE configs[4]): 2 / 4 / 8
device slots of one process -- sharing GPU 0 on a 1-GPU box -- against the single-device transform, bit for bit, for every layout
pair and direction; 2^26 over 8 slots is configs[4]'s own size."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sharded_ntt_worker.py")] + args, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "OK sharded ntt" in p.stdout
    return p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("slots,logs", [(1, "4,12,17"), (2, "6,12,16,20"), (4, "8,13,18,21"), (8, "10,12,16,19,22")])
def test_sharded_ntt_every_layout_pair_vs_single_device(slots, logs):
    _run([str(slots), logs], 900)


@pytest.mark.gpu
def test_sharded_ntt_2_26_over_8_slots():
    """BASELINE configs[4]'s transform size through the C ABI: 2^24 and 2^26 over 8 slots, each layout pair once."""
    _run(["8", "24,26", "big"], 1200)
