#!/usr/bin/env python3
"""A/B helper for the Fr NTT: forward and inverse times at 2^LOG for LOG in argv (default 16..26) plus fifteen batched 2^18
transforms (PLONK round 3's shape), with whatever knobs the environment sets.  Compare inside ONE gpurun call:
  python tools/ab_ntt.py; ZKP_NTT_TW_MATRIX_MAX_LOG=0 python tools/ab_ntt.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import zkp_hip as zkp  # noqa: E402

logs = [int(a) for a in sys.argv[1:]] or list(range(16, 27))
zkp.init()
dev = torch.device("cuda", 0)


def best_of(fn, reps):
    fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3


tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("ZKP_NTT")) or "default"
row = []
for ln in logs:
    n = 1 << ln
    x = bench.rand_fr_tensor(torch, n, 3000 + ln, dev)
    ref = x.clone()
    zkp.ntt_fr_dev(x, ln)
    zkp.ntt_fr_dev(x, ln, inverse=True)
    ok = bool(torch.equal(x, ref))
    reps = 50 if ln <= 20 else 20 if ln <= 24 else 8
    f = best_of(lambda: zkp.ntt_fr_dev(x, ln), reps)
    i = best_of(lambda: zkp.ntt_fr_dev(x, ln, inverse=True), reps)
    row.append(f"2^{ln}: {f:.3f}/{i:.3f}{'' if ok else ' ROUNDTRIP-DIFFERS'}")
    del x, ref
xb = bench.rand_fr_tensor(torch, 15 << 18, 77, dev)
b = best_of(lambda: zkp.ntt_fr_dev(xb, 18, batch=15), 20)
print(f"[{tag}] fwd/inv ms  " + "  ".join(row) + f"  15x2^18: {b:.3f}", flush=True)
