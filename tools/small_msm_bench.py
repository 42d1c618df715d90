"""Small-MSM timing: batches of 1..3 vectors of n terms over an expanded SRS.  python3 tools/small_msm_bench.py [LOG_N] [WINDOW]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import torch
import bench, zkp_hip as zkp
zkp.init()
dev = torch.device("cuda", 0)
ln = int(sys.argv[1]) if len(sys.argv) > 1 else 16
wb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = (1 << ln) + 3
ks = bench.rand_fr_tensor(torch, n, 1, dev)
pts = torch.zeros(n * 12, dtype=torch.int64, device=dev)
zkp.g1_fixed_base_mul_dev(ks, n, pts); torch.cuda.synchronize()
bases = zkp.G1Bases.from_device(pts, n)
if wb >= 0:
    bases.precompute(wb)
vecs = [bench.rand_fr_tensor(torch, n, 10 + i, dev) for i in range(3)]
for count in (1, 2, 3):
    for _ in range(3): zkp.msm_g1_batch_dev(bases, vecs[:count], n)
    zkp.profile_reset(); zkp.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(10): zkp.msm_g1_batch_dev(bases, vecs[:count], n)
    dt = (time.perf_counter() - t0) / 10
    zkp.profile_enable(False)
    ph = {k: round(zkp.profile_read(k)[0] / 10, 3) for k in ("msm_sort", "msm_accumulate", "msm_bucket_reduce", "msm_tail_host")}
    print(f"n=2^{ln}+3 batch {count}: {dt*1e3:.3f} ms {ph}", flush=True)
