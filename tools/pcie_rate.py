import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import numpy as np, torch
import bench, zkp_hip as zkp
zkp.init()
dev = torch.device("cuda", 0)
n = 1 << 20
ks = bench.rand_fr_tensor(torch, n, 1, dev); sc = bench.rand_fr_tensor(torch, n, 2, dev)
pts = torch.zeros(n * 12, dtype=torch.int64, device=dev)
zkp.g1_fixed_base_mul_dev(ks, n, pts); torch.cuda.synchronize()
bases = zkp.G1Bases.from_device(pts, n)
if len(sys.argv) > 1 and int(sys.argv[1]):
    bases.precompute(int(sys.argv[1]))
h = sc.cpu().numpy().view(np.uint64).reshape(n, 4)
zkp.msm_g1(bases, h)
t0 = time.perf_counter()
for _ in range(5): zkp.msm_g1(bases, h)
t_host = (time.perf_counter() - t0) / 5
zkp.msm_g1_dev(bases, sc, n)
t0 = time.perf_counter()
for _ in range(5): zkp.msm_g1_dev(bases, sc, n)
t_dev = (time.perf_counter() - t0) / 5
print(f"2^20 MSM: device-resident scalars {t_dev*1e3:.2f} ms; host scalars through zkp_msm_g1 (pageable, PCIe-inclusive) {t_host*1e3:.2f} ms")
# raw pageable host -> device rate of the same 32 MB, and the phases of the host-scalar path
buf = torch.empty(n * 4, dtype=torch.int64, device=dev)
src = torch.from_numpy(h.view(np.int64).reshape(-1))
buf.copy_(src); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): buf.copy_(src)
torch.cuda.synchronize()
t_copy = (time.perf_counter() - t0) / 5
print(f"plain pageable H2D of the {32 * n / 2**20:.0f} MiB of scalars: {t_copy*1e3:.2f} ms = {32 * n / t_copy / 1e9:.1f} GB/s")
zkp.profile_reset(); zkp.profile_enable(True)
for _ in range(5): zkp.msm_g1(bases, h)
zkp.profile_enable(False)
print("phases per MSM (host scalars, ZKP_MSM_FEED_RANGES=%s):" % os.environ.get("ZKP_MSM_FEED_RANGES", "4"),
      {k: round(zkp.profile_read(k)[0] / 5, 3) for k in ("msm_digits", "msm_sort", "msm_accumulate", "msm_bucket_reduce", "msm_tail_host")})
