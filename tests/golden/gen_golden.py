#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the independent big-int model.

Run from the repo root:  python tests/golden/gen_golden.py
The Rust reference cannot be executed in this environment (no cargo/rustc; arkworks is not
vendored), so these vectors come from `tests/model/bigmodel.py`, which is pinned by the
reference's own known-answer tests (see tests/test_model_kat.py and SURVEY.md §8c).
All values are canonical integers as lowercase hex strings (no Montgomery form).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "model"))
import bigmodel as M  # noqa: E402


def hx(v):
    return format(v, "x")


def pt(p):
    return None if p is M.INF else [hx(p[0]), hx(p[1])]


def main():
    out = {}

    # --- reference KATs (kzg/src/commitment.rs:36-53; fri/src/prover.rs:181-205;
    #     plonk/src/slice_polynomial.rs:80-111)
    srs2 = M.srs(2, 10)
    c17 = M.msm_naive([1, 2, 3], srs2)
    opening, y = M.kzg_open([1, 2, 3], 1, srs2)
    slices, deg = M.slice_poly(list(range(1, 13)))
    out["reference_kat"] = {
        "commit_1_2_3_secret2": pt(c17),
        "seventeen_G": pt(M.g1_mul(M.G1, 17)),
        "eval_at_1": hx(y),
        "open_at_1": pt(opening),
        "fri_fold_1234_r1": [hx(v) for v in M.fri_fold([1, 2, 3, 4], 1)],
        "fri_layer1_coset": hx(M.GL_GENERATOR ** 2 % M.GL),
        "slice12": {"slices": [[hx(v) for v in s] for s in slices], "degree": deg,
                    "compact_zeta5": [hx(v) for v in M.slice_compact(slices, deg, 5)]},
    }

    # --- scalar multiplications of G
    ks = [1, 2, 3, 17, M.R - 1, M.R - 2, (1 << 254) + 12345, 0x1234567890ABCDEF << 100, 0]
    ks += M.rand_fr_list(0xC0FFEE, 4)
    out["g1_mul"] = [{"k": hx(k), "out": pt(M.g1_mul(M.G1, k))} for k in ks]

    # --- MSM over a real SRS (secret known => expected == [p(s)]G as in commitment.rs:46-51)
    msm = []
    for secret, n, seed in [(2, 8, 1), (0xDEADBEEF12345, 64, 2), (M.rand_fr_list(77, 1)[0], 200, 3)]:
        points = M.srs(secret, n - 3)
        scalars = M.rand_fr_list(seed, n)
        if n == 64:  # adversarial digits: zero, one, r-1, repeated
            scalars[0], scalars[1], scalars[2], scalars[3], scalars[4] = 0, 1, M.R - 1, scalars[5], 2
        naive = M.msm_naive(scalars, points)
        trap = M.g1_mul(M.G1, M.poly_eval(scalars, secret, M.R))
        assert naive == trap
        msm.append({"secret": hx(secret), "n": n, "seed": seed, "scalars": [hx(s) for s in scalars],
                    "points": [pt(p) for p in points], "out": pt(naive)})
    # bases containing infinity and a cancelling pair
    pts = [M.G1, M.INF, M.g1_neg(M.G1), M.g1_mul(M.G1, 5)]
    sc = [9, 1234, 9, 3]
    msm.append({"secret": None, "n": 4, "seed": None, "scalars": [hx(s) for s in sc],
                "points": [pt(p) for p in pts], "out": pt(M.msm_naive(sc, pts))})
    out["msm"] = msm

    # --- Fr NTT (ark-poly radix-2 domain semantics)
    ntt = []
    for log_n, seed in [(0, 9), (1, 10), (2, 11), (3, 12), (4, 13), (5, 14), (10, 15)]:
        n = 1 << log_n
        a = M.rand_fr_list(seed, n)
        fwd = M.ntt(a, M.R)
        if n <= 32:
            assert fwd == M.dft_naive(a, M.R, M.root_of_unity(log_n, M.R))
        assert M.ntt(fwd, M.R, inverse=True) == a
        ent = {"log_n": log_n, "seed": seed, "omega": hx(M.root_of_unity(log_n, M.R)),
               "in": [hx(v) for v in a], "ntt": [hx(v) for v in fwd],
               "intt": [hx(v) for v in M.ntt(a, M.R, inverse=True)],
               "coset7_ntt": [hx(v) for v in M.coset_ntt(a, 7, M.R)],
               "coset7_intt": [hx(v) for v in M.coset_intt(a, 7, M.R)]}
        ntt.append(ent)
    out["ntt_fr"] = ntt

    # --- Goldilocks NTT + FRI layer evaluation (fri/src/fri_layer.rs:40-46)
    gl = []
    for log_n, seed in [(2, 21), (3, 22), (6, 23), (10, 24)]:
        n = 1 << log_n
        a = M.rand_gl_list(seed, n)
        fwd = M.ntt(a, M.GL)
        if n <= 64:
            assert fwd == M.dft_naive(a, M.GL, M.root_of_unity(log_n, M.GL))
        gl.append({"log_n": log_n, "seed": seed, "in": [hx(v) for v in a], "ntt": [hx(v) for v in fwd],
                   "intt": [hx(v) for v in M.ntt(a, M.GL, inverse=True)],
                   "coset7_ntt": [hx(v) for v in M.coset_ntt(a, 7, M.GL)]})
    out["ntt_goldilocks"] = gl

    fri = []
    for coeffs, coset, d in [([1, 2, 3, 4], 7, 4), ([1, 2, 3, 4], 7, 8), ([3, 7], 49, 2),
                             (M.rand_gl_list(31, 6), 7, 16), (M.rand_gl_list(32, 64), 7, 128)]:
        ev = M.fri_layer_eval(coeffs, coset, d)
        padded = coeffs + [0] * (d - len(coeffs))
        assert ev == M.coset_ntt(padded, coset, M.GL)
        fri.append({"coeffs": [hx(v) for v in coeffs], "coset": hx(coset), "domain": d,
                    "evals": [hx(v) for v in ev],
                    "fold_r5": [hx(v) for v in M.fri_fold(coeffs, 5)]})
    out["fri_layer"] = fri
    assert fri[0]["evals"] == [hx(v) for v in
                               [1534, 18064501051041513327, 18446744069414583083, 382243018373070702]]

    # --- polynomial product / vanishing-polynomial helpers (ark-poly semantics used by plonk/src/prover.rs)
    pm = []
    for la, lb, seed in [(1, 1, 41), (3, 5, 42), (8, 8, 43), (33, 17, 44)]:
        a, b = M.rand_fr_list(seed, la), M.rand_fr_list(seed + 100, lb)
        prod = M.poly_mul(a, b, M.R)
        n = 4
        mv = M.mul_by_vanishing(a, n, M.R)
        q, rem = M.divide_by_vanishing(mv, n, M.R)
        assert q == M.poly_trim(a) and rem == []
        pm.append({"a": [hx(v) for v in a], "b": [hx(v) for v in b], "prod": [hx(v) for v in prod],
                   "a_times_zh4": [hx(v) for v in mv]})
    out["poly"] = pm

    # --- KZG open on a seeded polynomial
    pts = M.srs(0xABCDEF, 13)
    coeffs = M.rand_fr_list(51, 16)
    z = M.rand_fr_list(52, 1)[0]
    w, y = M.kzg_open(coeffs, z, pts)
    out["kzg_open"] = {"secret": hx(0xABCDEF), "coeffs": [hx(v) for v in coeffs], "z": hx(z),
                       "points": [pt(p) for p in pts], "eval": hx(y), "opening": pt(w),
                       "commit": pt(M.msm_naive(coeffs, pts))}

    # --- FRI commitment path (hashlib model; third-party behaviour as documented in oracle/fri_oracle.c)
    GL = M.GL
    leaves = [1, 2, 3, 4]  # fri/src/merkle_tree.rs:141-151
    rnd_leaves = M.rand_gl_list(61, 13)
    stdrng = M.StdRng(0x0123456789ABCDEF)
    tr = M.FriTranscript()
    tr.digest(928459)  # fri/src/fiat_shamir/transcript.rs:160-172
    proofs = []
    for coeffs, blowup, nq in (([1, 2, 3, 4], 2, 2), ([1, 2, 3, 4, 5, 6], 2, 2), (M.rand_gl_list(62, 21), 4, 3)):
        pr = M.fri_prove(coeffs, blowup, nq)
        proofs.append({"coeffs": [hx(v) for v in coeffs], "blowup": blowup, "queries": nq,
                       "flat_canonical": [hx(v) for v in M.fri_flatten(pr, lambda v: v)]})
    out["fri_commit"] = {
        "hash": [{"in": hx(v), "out": hx(M.gl_hash_slice([v]))} for v in (0, 1, 10, GL - 1, 12345678901234567890 % GL)],
        "hash_pair": {"in": [hx(12), hx(3)], "out": hx(M.gl_hash_slice([12, 3]))},
        "merkle": [{"leaves": [hx(v) for v in ls], "nodes": [hx(v) for lvl in M.merkle_levels(ls) for v in lvl]}
                   for ls in (leaves, rnd_leaves)],
        "stdrng_seed": hx(0x0123456789ABCDEF), "stdrng_u64": [hx(stdrng.next_u64()) for _ in range(40)],
        "transcript_after_928459_challenge": hx(tr.challenge()),
        "proofs": proofs,
    }

    path = os.path.join(HERE, "vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
