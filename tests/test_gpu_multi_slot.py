"""Several device slots in one process (zkp_init_devices; on a 1-GPU box they share the device): sharded SRS with host and with
resident scalars, plain and expanded, prefixes that leave chunks idle, and one independent worker thread per slot."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [3])
def test_sharded_msm_and_per_slot_workers(slots):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multi_slot_worker.py"), str(slots)], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "OK multi-slot" in p.stdout


@pytest.mark.gpu
def test_sharded_host_scalar_msm_with_concurrent_uploaders():
    """Two slots, chunks of 2^19 scalars and more over expanded bases: zkp_msm_g1 walks every chunk in ranges on its own caller thread,
    each with its slot's uploader thread (api.hip: Uploader) -- concurrently; the sum is the known answer every time."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multi_slot_worker.py"), "2", "big"], capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "OK multi-slot" in p.stdout
