"""GPU soak (python3 tests/soak/fuzz_ntt.py SEED ITERATIONS): Fr / Goldilocks NTT sizes 2^1..2^22, forward, inverse, coset; against the oracle up to 2^14, round trips above."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zkp-implementation_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import zkp_hip as zkp
import oracle as orc
zkp.init()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 80):
    ln = rnd.randint(1, 22)
    n = 1 << ln
    use_coset = rnd.random() < 0.4
    if rnd.random() < 0.6:
        x = orc.rand_fr(100 + it, n)
        cs = orc.rand_fr(900 + it, 1)[0] if use_coset else None
        y = zkp.ntt_fr(x, False, cs)
        ok = np.array_equal(zkp.ntt_fr(y, True, cs), x)
        if ln <= 14:
            ok = ok and np.array_equal(y, orc.ntt_fr(x, False, cs)) and np.array_equal(zkp.ntt_fr(x, True, cs), orc.ntt_fr(x, True, cs))
        f = "fr"
    else:
        x = orc.rand_gl(100 + it, n)
        cs = orc.rand_gl(900 + it, 1) if use_coset else None
        y = zkp.ntt_goldilocks(x, False, cs)
        ok = np.array_equal(zkp.ntt_goldilocks(y, True, cs), x)
        if ln <= 14:
            ok = ok and np.array_equal(y, orc.ntt_gl(x, False, cs))
        f = "gl"
    if not ok:
        bad += 1
        print("MISMATCH", it, f, ln, use_coset, flush=True)
    if it % 20 == 0: print("it", it, f, ln, use_coset, ok, flush=True)
print("done, mismatches:", bad)
sys.exit(1 if bad else 0)
