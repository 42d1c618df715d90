"""GPU soak (python3 tests/soak/fuzz_plonk.py SEED ITERATIONS): PLONK generate_proof on synthetic circuits of 2^2 .. 2^15 gates (the
circuit family of tests/test_gpu_plonk.py), each proof checked by the pairing verifier (plonk/src/verifier.rs:19-157) and a
tampered copy rejected."""
import copy, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("", "zkp-implementation_amd", "oracle", "tests", os.path.join("tests", "model")):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import zkp_hip as zkp
import oracle as orc
import bigmodel as M
import pairing_model as PMod
from test_gpu_plonk import synthetic_circuit
from test_pairing_cpu import g2_from_ints
zkp.init()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
f = lambda v: orc.fr_from_ints([v])[0]
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    log_n = rnd.randint(2, 15)
    n = 1 << log_n
    polys, k1, k2 = synthetic_circuit(orc, zkp, log_n, 0xC000 + it)
    secret = M.rand_fr_list(0x700 + it, 1)[0]
    srs = zkp.Srs.new_from_secret(f(secret), n)
    if rnd.random() < 0.5 and n >= 64:
        srs.bases.precompute(0)
    g2s = g2_from_ints(PMod.g2_mul(PMod.G2, secret))
    pr = zkp.PlonkProver(srs.bases, log_n, polys, f(k1), f(k2))
    proof = pr.prove(orc.fr_from_ints(M.rand_fr_list(0x900 + it, 9)))
    ok = pr.verify(g2s, proof) == 1
    tam = copy.deepcopy(proof)
    tam["commits"]["t_mid"] = proof["commits"]["t_lo"]
    ok = ok and pr.verify(g2s, tam) != 1
    pr.close()
    if not ok:
        bad += 1
        print("MISMATCH", it, log_n, flush=True)
    print("it", it, "log_n", log_n, ok, flush=True)
print("done, mismatches:", bad)
sys.exit(1 if bad else 0)
