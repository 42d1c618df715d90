#!/usr/bin/env python3
"""Build libzkp_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.  In-tree output so the
library travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libzkp_hip.so")
SOURCES = ["api.hip"]
DEPS = ["api.hip", "ff.cuh", "fq28.cuh", "fr29.cuh", "g1.cuh", "g1_28.cuh", "msm.cuh", "ntt.cuh", "plonk.cuh",
        "plonk_host.inc", "fri.cuh", "fri_host.inc", "transcript_host.hpp", "pairing_host.hpp", "verify_host.inc", "host_ff.hpp", "kzg_host.hpp",
        os.path.join("..", "..", "include", "zkp_hip.h")]


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(os.path.join(SRC, d)) and os.path.getmtime(os.path.join(SRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result", "-pthread"] + [os.path.join(SRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
