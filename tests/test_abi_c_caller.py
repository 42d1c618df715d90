"""The C ABI is usable from plain C: tests/abi/c_smoke.c compiles against include/zkp_hip.h and links libzkp_hip.so with
gcc (CPU check); on a GPU it replays the reference's KZG test (kzg/src/commitment.rs:36-53) end to end."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "zkp-implementation_amd")


def _build(tmp_path):
    import sys
    sys.path.insert(0, PKG)
    import build as zbuild
    zbuild.build()
    exe = os.path.join(str(tmp_path), "c_smoke")
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi", "c_smoke.c"),
           "-o", exe, "-L", PKG, "-lzkp_hip", "-Wl,-rpath," + PKG]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_caller_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_c_caller_runs_reference_kzg_test(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "c_smoke ok" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [2, 3, 8])
def test_c_caller_multi_device_slots(tmp_path, slots):
    """`c_smoke --devices N`: N device slots behind the C ABI (sharing GPU 0 on a 1-GPU box): bases sharded at creation,
    zkp_msm_g1 / zkp_kzg_commit / zkp_kzg_open over the shards == the single-slot results, plain and expanded."""
    exe = _build(tmp_path)
    out = subprocess.run([exe, "--devices", str(slots)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert f"{slots} device slots" in out.stdout
