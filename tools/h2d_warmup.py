"""Per-call times of the host-scalar MSM (zkp_msm_g1, 2^20 terms): the same host buffer every call against a fresh buffer every call --
what warms up over the first calls?   python3 tools/h2d_warmup.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import numpy as np, torch
import bench, zkp_hip as zkp
zkp.init()
dev = torch.device("cuda", 0)
n = 1 << 20
ks = bench.rand_fr_tensor(torch, n, 1020, dev); sc = bench.rand_fr_tensor(torch, n, 2020, dev)
pts = torch.zeros(n * 12, dtype=torch.int64, device=dev)
zkp.g1_fixed_base_mul_dev(ks, n, pts); torch.cuda.synchronize()
bases = zkp.G1Bases.from_device(pts, n); bases.precompute(0)
h = sc.cpu().numpy().view(np.uint64).reshape(n, 4).copy()
ref = zkp.msm_g1_dev(bases, sc, n)
for _ in range(5): zkp.msm_g1_dev(bases, sc, n)
t0 = time.perf_counter()
for _ in range(20): zkp.msm_g1_dev(bases, sc, n)
print(f"resident: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
def series(fresh, label):
    ts = []
    for i in range(24):
        buf = h.copy() if fresh else h
        t0 = time.perf_counter()
        out = zkp.msm_g1(bases, buf)
        ts.append((time.perf_counter() - t0) * 1e3)
        assert (out[0] == ref[0]).all()
    print(label, " ".join(f"{t:.2f}" for t in ts))
series(False, "same buffer :")
pin = torch.empty((n, 4), dtype=torch.int64).pin_memory()
pin.copy_(torch.from_numpy(h.view(np.int64)))
hp = pin.numpy().view(np.uint64)
_h = h
h = hp
series(False, "pinned buffer (hipHostMalloc), same every call:")
def series_pinned_rewritten():
    ts = []
    for i in range(12):
        pin.copy_(torch.from_numpy(_h.view(np.int64)))   # the caller recomputes its scalars into the pinned buffer
        t0 = time.perf_counter(); out = zkp.msm_g1(bases, hp); ts.append((time.perf_counter() - t0) * 1e3)
    print("pinned buffer, rewritten before every call:", " ".join(f"{t:.2f}" for t in ts))
series_pinned_rewritten()
h = _h
series(True, "fresh buffer:")
time.sleep(0.5)
series(False, "same, after 0.5 s idle:")
