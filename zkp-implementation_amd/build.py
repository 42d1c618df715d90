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
DEPS = ["api.hip", "ff.hpp", "fq28.hpp", "fq28_inv.hpp", "fr29.hpp", "fr29_mul2_asm.inc", "fq28_mul_asm.inc", "fq28_mul2x_asm.inc", "fq28_sqr_asm.inc", "fq28_mul2_asm.inc", "g1.hpp", "g1_28.hpp", "msm.hpp", "ntt.hpp", "plonk.hpp",
        "plonk_host.inc", "ntt_sharded.inc", "fri.hpp", "fri_host.inc", "transcript_host.hpp", "pairing_host.hpp", "verify_host.inc", "host_ff.hpp", "kzg_host.hpp", "host_threads.hpp",
        os.path.join("..", "..", "include", "zkp_hip.h")]


STAMP = OUT + ".srchash"  # hash of the sources the library was built from (mtimes do not survive the copy to the GPU box)


def source_hash():
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        path = os.path.join(SRC, d)
        if os.path.exists(path):
            h.update(d.encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()


def stale():
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    return open(STAMP).read().strip() != source_hash()


def build(force=False, verbose=False, defines=(), out=None):
    """defines / out: an experimental variant next to the product library (A/B runs pick it with ZKP_HIP_LIB)."""
    variant = out is not None
    if not variant and not force and not stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result", "-pthread"] + [f"-D{d}" for d in defines] + [os.path.join(SRC, s) for s in SOURCES] + ["-o", out or OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if not variant:
        with open(STAMP, "w") as f:
            f.write(source_hash())
    return out or OUT


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a[2:] for a in sys.argv[1:] if a.startswith("-o")]
    print(build(force="--force" in sys.argv, verbose=True, defines=defs, out=outs[0] if outs else None))
