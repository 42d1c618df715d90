import sys, time, os
sys.path.insert(0, os.path.join(os.getcwd(), "zkp-implementation_amd"))
import torch, numpy as np
import zkp_hip as zkp
zkp.init()
for ln in (20, 24, 26):
    n = 1 << ln
    g = torch.Generator(device="cuda"); g.manual_seed(ln)
    t = torch.randint(0, 2**62, (n,), dtype=torch.int64, device="cuda", generator=g)
    for _ in range(2):
        zkp.ntt_goldilocks_dev(t, ln); zkp.ntt_goldilocks_dev(t, ln, inverse=True)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        zkp.ntt_goldilocks_dev(t, ln)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"GL NTT 2^{ln}: {dt*1e3:.3f} ms  {n/dt/1e9:.2f} Gelem/s  algorithmic {16*n/dt/1e9:.0f} GB/s ({16*n/dt/8e12*100:.1f}% of 8 TB/s)")
