"""FRI generate_proof timing: python3 tools/fri_bench.py [LOG_D_COEFFS] [BLOWUP] [QUERIES]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd"))
import numpy as np
import torch
import zkp_hip as zkp

ld = int(sys.argv[1]) if len(sys.argv) > 1 else 20
blow = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 32
zkp.init()
rnd = np.random.default_rng(1)
coeffs = rnd.integers(1, 2 ** 63, 1 << ld, dtype=np.uint64)
for rep in range(3):
    zkp.profile_reset()
    zkp.profile_enable(True)
    t0 = time.perf_counter()
    proof = zkp.fri_prove(coeffs, blow, nq)
    dt = time.perf_counter() - t0
    zkp.profile_enable(False)
    ph = {k: zkp.profile_read(k) for k in ("fri_merkle", "ntt_gl_pass")}
    print(f"fri_prove 2^{ld} coeffs x{blow}: {dt * 1e3:.2f} ms, proof {proof.size * 8 / 1024:.1f} KiB, phases {ph}")
t0 = time.perf_counter()
ok = zkp.fri_verify(proof)
print("verify", ok, f"{(time.perf_counter() - t0) * 1e3:.2f} ms")
n = 1 << (ld + 1)
d_leaves = torch.randint(0, 2 ** 62, (n,), dtype=torch.int64, device="cuda")
d_nodes = torch.zeros(zkp.fri_merkle_node_count(n), dtype=torch.int64, device="cuda")
for _ in range(2):
    zkp.fri_merkle_tree_dev(d_leaves, n, d_nodes)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    zkp.fri_merkle_tree_dev(d_leaves, n, d_nodes)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"merkle tree 2^{ld + 1} leaves: {dt * 1e3:.3f} ms = {(2 * n - 1) / dt / 1e9:.2f} G hashes/s")
