"""GPU soak (python3 tests/soak/fuzz_fri.py SEED ITERATIONS): FRI generate_proof over random degrees, blowups and query counts; bit-exact
against the oracle's prover for small degrees, accepted by the oracle's verifier and the library's own for all."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import zkp_hip as zkp
import oracle as orc
zkp.init()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    d = rnd.choice([rnd.randint(1, 40), rnd.randint(40, 3000), rnd.randint(3000, 300000)])
    blow = rnd.choice([1, 2, 2, 4, 8])
    nq = rnd.randint(0, 40)
    c = orc.rand_gl(7000 + it, d)
    if rnd.random() < 0.2: c[rnd.randrange(d)] = 0
    proof = zkp.fri_prove(c, blow, nq)
    ok = orc.fri_verify(proof) == 0 and bool(zkp.fri_verify(proof))
    if d * blow <= 4096:
        ok = ok and np.array_equal(proof, orc.fri_prove(c, blow, nq))
    if not ok:
        bad += 1
        print("MISMATCH", it, d, blow, nq, flush=True)
    if it % 10 == 0: print("it", it, d, blow, nq, ok, flush=True)
print("done, mismatches:", bad)
sys.exit(1 if bad else 0)
