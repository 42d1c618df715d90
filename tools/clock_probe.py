import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import torch, bench, zkp_hip as zkp
zkp.init()
dev = torch.device("cuda", 0)
ln = 24
x = bench.rand_fr_tensor(torch, 1 << ln, 3000 + ln, dev)
def rt(reps=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        zkp.ntt_fr_dev(x, ln); zkp.ntt_fr_dev(x, ln, inverse=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
rt(3)
print("cold-ish round trip ms:", [round(rt(5), 3) for _ in range(4)])
# burn: 4 seconds of 2^22 MSMs
n = 1 << 22
ks = bench.rand_fr_tensor(torch, n, 1, dev); sc = bench.rand_fr_tensor(torch, n, 2, dev)
pts = torch.zeros(n * 12, dtype=torch.int64, device=dev); zkp.g1_fixed_base_mul_dev(ks, n, pts); torch.cuda.synchronize()
b = zkp.G1Bases.from_device(pts, n); b.precompute(0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 5: zkp.msm_g1_dev(b, sc, n)
print("after a 5 s MSM burn:", [round(rt(5), 3) for _ in range(4)])
time.sleep(3)
print("after 3 s idle:", [round(rt(5), 3) for _ in range(4)])
zkp.profile_reset(); zkp.profile_enable(True)
print("with the per-pass event markers on:", [round(rt(5), 3) for _ in range(3)])
zkp.profile_enable(False); zkp.profile_reset()
print("markers off again:", [round(rt(5), 3) for _ in range(3)])
