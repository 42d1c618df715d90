#!/usr/bin/env python3
"""A/B helper: time the expanded-base MSM at n = 2^LOG (default 20) with the library ZKP_HIP_LIB points at.
Box-to-box spread is a few percent, so compare builds inside ONE gpurun call:  python tools/ab_msm.py; ZKP_HIP_LIB=... python tools/ab_msm.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import zkp_hip as zkp  # noqa: E402

ln = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
zkp.init()
dev = torch.device("cuda", 0)
n = 1 << ln
ks = bench.rand_fr_tensor(torch, n, 1000 + ln, dev)
sc = bench.rand_fr_tensor(torch, n, 2000 + ln, dev)
pts = torch.zeros(n * 12, dtype=torch.int64, device=dev)
zkp.g1_fixed_base_mul_dev(ks, n, pts)
torch.cuda.synchronize()
bases = zkp.G1Bases.from_device(pts, n)
bases.precompute(int(os.environ.get("ZKP_AB_C", "0")))  # 0 = automatic width
out = zkp.msm_g1_dev(bases, sc, n)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = zkp.msm_g1_dev(bases, sc, n)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / reps)
zkp.profile_reset()
zkp.profile_enable(True)
for _ in range(reps):
    zkp.msm_g1_dev(bases, sc, n)
torch.cuda.synchronize()
zkp.profile_enable(False)
ph = {k: round(zkp.profile_read(k)[0] / reps, 3) for k in ("msm_digits", "msm_sort", "msm_accumulate", "msm_bucket_reduce", "msm_tail_host")}
cyc, ref, _ = zkp.profile_clock_read("msm_accumulate")
ph["accumulate_mhz"] = round(100.0 * cyc / ref, 1) if ref else None   # the clock the kernel held (in-kernel stamps): compare builds in cycles
ph["accumulate_kcycles"] = round(ph["msm_accumulate"] * ph["accumulate_mhz"], 1) if ref else None
print(f"{os.environ.get('ZKP_HIP_LIB', 'default')} c={os.environ.get('ZKP_AB_C', 'auto')} range={os.environ.get('ZKP_MSM_RANGE_LOG', 'auto')} fuse={os.environ.get('ZKP_PYR_FUSE', '-')}: n=2^{ln} {best * 1e3:.3f} ms  result={out[0].tobytes().hex()[:16]} {ph}", flush=True)
