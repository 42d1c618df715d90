#!/usr/bin/env python3
"""Throughput of a stream of independent 2^LOG-term MSMs with ONE or TWO in flight on one GPU: two device slots on the same GPU
(zkp_init_devices({0, 0})), one host thread per slot, each with its own copy of the expanded SRS and its own workspaces and stream, so
that the sort and the latency-bound end of the bucket reduction of one MSM run under the accumulation of the other.
python3 tools/two_in_flight.py [LOG] [REPS]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import zkp_hip as zkp  # noqa: E402

ln = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
zkp.init_devices([0, 0])
dev = torch.device("cuda", 0)
n = 1 << ln
ks = bench.rand_fr_tensor(torch, n, 1000 + ln, dev)
pts = torch.zeros(n * 12, dtype=torch.int64, device=dev)
zkp.g1_fixed_base_mul_dev(ks, n, pts)
torch.cuda.synchronize()
state = {}


def setup(slot):
    zkp.set_device(slot)
    b = zkp.G1Bases.from_device(pts, n)
    b.precompute(0)
    sc = bench.rand_fr_tensor(torch, n, 2000 + ln + slot, dev)
    out = zkp.msm_g1_dev(b, sc, n)
    state[slot] = (b, sc, out)


def loop(slot, count, res):
    zkp.set_device(slot)
    b, sc, ref = state[slot]
    for _ in range(count):
        out = zkp.msm_g1_dev(b, sc, n)
    res[slot] = bool(np.array_equal(out[0], ref[0]))


for s in (0, 1):
    t = threading.Thread(target=setup, args=(s,))
    t.start()
    t.join()
for trial in range(3):
    res = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(0, reps, res)
    one = (time.perf_counter() - t0) / reps
    ts = [threading.Thread(target=loop, args=(s, reps, res)) for s in (0, 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    two = (time.perf_counter() - t0) / (2 * reps)
    print(f"n = 2^{ln}: one in flight {one * 1e3:.3f} ms per MSM ({n / one:.3e} scalar-muls/s); two in flight {two * 1e3:.3f} ms per MSM "
          f"({n / two:.3e}/s, {one / two:.3f}x); results exact: {all(res.values())}", flush=True)
