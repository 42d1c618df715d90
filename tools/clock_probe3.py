"""Do short idle gaps between kernels slow the next kernels down (clock management)?  Fr NTT 2^22 forward transforms (0.5 ms) issued
back to back, against the same with a host synchronisation and a short sleep after each one, the way an MSM loop returns its result to the
host after every step.  python3 tools/clock_probe3.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import torch, bench, zkp_hip as zkp
zkp.init()
dev = torch.device("cuda", 0)
ln = 22
x = bench.rand_fr_tensor(torch, 1 << ln, 1, dev).reshape(-1)
for _ in range(50): zkp.ntt_fr_dev(x, ln)
torch.cuda.synchronize()
def busy(sleep_us):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    n = 60
    for _ in range(n):
        e0.record(); zkp.ntt_fr_dev(x, ln); e1.record()
        if sleep_us is not None:
            torch.cuda.synchronize()
            if sleep_us: time.sleep(sleep_us * 1e-6)
            tot += e0.elapsed_time(e1)
    if sleep_us is None:
        torch.cuda.synchronize(); return None
    return tot / n
for rep in range(2):
    t0 = time.perf_counter(); busy(None); back = (time.perf_counter() - t0) / 60 * 1e3
    print(f"back to back: {back:.4f} ms per transform;  with sync after each: {busy(0):.4f};  sync + 50 us sleep: {busy(50):.4f};  sync + 300 us: {busy(300):.4f};  sync + 2 ms: {busy(2000):.4f}", flush=True)
