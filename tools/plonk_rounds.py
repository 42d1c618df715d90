import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import numpy as np, torch
import bench, zkp_hip as zkp
zkp.init()
log_n = 16; n = 1 << log_n
rnd = np.random.default_rng(1)
polys = {k: bench.fr_mont([int(x) for x in rnd.integers(1, 2**62, n)]) for k in zkp.CIRCUIT_POLYS}
f = lambda v: bench.fr_mont([v])[0]
srs = zkp.Srs.new_from_secret(f(12345), n)
vals = [int(x) for x in rnd.integers(1, 2**62, 14)]
for rep in range(3):
    pr = zkp.PlonkProver(srs.bases, log_n, polys, f(2), f(3))
    torch.cuda.synchronize(); ts = [time.perf_counter()]
    pr.round1(bench.fr_mont(vals[:6])); ts.append(time.perf_counter())
    pr.round2(f(vals[9]), f(vals[10]), bench.fr_mont(vals[6:9])); ts.append(time.perf_counter())
    try:
        pr.round3(f(vals[11]))
    except zkp.ZkpError as e:
        pass
    ts.append(time.perf_counter())
    print("round ms:", [round((b - a) * 1e3, 3) for a, b in zip(ts, ts[1:])])
    pr.close()
