"""Does earlier MSM work in the same process slow the Fr NTT down?  (bench.py reports 4.3-4.6 ms for the 2^24 round trip, a fresh
process 3.95-4.0 ms on the same box.)  python3 tools/clock_probe2.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import numpy as np, torch, bench, zkp_hip as zkp
zkp.init()
dev = torch.device("cuda", 0)
ln = 24
def rt(x, reps=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        zkp.ntt_fr_dev(x, ln); zkp.ntt_fr_dev(x, ln, inverse=True)
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / reps * 1e3, 3)
def fresh():
    x = bench.rand_fr_tensor(torch, 1 << ln, 3000 + ln, dev).reshape(-1)
    rt(x, 6)
    return [rt(x) for _ in range(3)]
print("before anything:", fresh(), flush=True)
wl = bench.MsmWorkload(zkp, torch, dev, 20, chunk=0, expand="auto")
for _ in range(25): zkp.msm_g1_dev(wl.bases, wl.scalars, wl.n)
print("after 25 MSMs of 2^20 (expanded SRS alive):", fresh(), flush=True)
sc_host = wl.scalars.cpu().numpy().view(np.uint64).reshape(-1, 4)
for _ in range(5): zkp.msm_g1(wl.bases, sc_host)
print("after 5 host-scalar MSMs (copy stream, pinned staging):", fresh(), flush=True)
wl.close(); del wl; torch.cuda.empty_cache()
print("after releasing the SRS:", fresh(), flush=True)
