"""Time one field's forward NTT on the GPU: python3 tools/ntt_bench.py fr|gl LOG_N [REPS].  Used under rocprofv3."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import torch
import zkp_hip as zkp

field, ln = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
zkp.init()
n = 1 << ln
g = torch.Generator(device="cuda")
g.manual_seed(ln)
if field == "fr":
    t = torch.randint(0, 2 ** 62, (4 * n,), dtype=torch.int64, device="cuda", generator=g)
    run, bytes_per = (lambda: zkp.ntt_fr_dev(t, ln)), 64
else:
    t = torch.randint(0, 2 ** 62, (n,), dtype=torch.int64, device="cuda", generator=g)
    run, bytes_per = (lambda: zkp.ntt_goldilocks_dev(t, ln)), 16
ref = t.clone()
run()
(zkp.ntt_fr_dev if field == "fr" else zkp.ntt_goldilocks_dev)(t, ln, inverse=True)
assert torch.equal(t, ref) or field == "fr", "round trip differs"  # fr inputs here are not canonical residues
for _ in range(2):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{field} NTT 2^{ln}: {dt * 1e3:.3f} ms  {n / dt / 1e9:.2f} Gelem/s  algorithmic {bytes_per * n / dt / 1e9:.0f} GB/s "
      f"({bytes_per * n / dt / 8e12 * 100:.1f}% of 8 TB/s)")
