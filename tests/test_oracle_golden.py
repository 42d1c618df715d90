"""The C oracle (oracle/zkp_oracle.c) against the golden vectors and the reference KATs. CPU only."""
import numpy as np
import pytest

import bigmodel as M
from conftest import hx, pt_from_hex


def test_field_roundtrip_and_ops(orc):
    vals = [0, 1, 2, M.R - 1, M.R - 2] + M.rand_fr_list(5, 20)
    a = orc.fr_from_ints(vals)
    assert orc.fr_to_ints(a) == vals
    # Montgomery form really is x*R mod r (arkworks in-memory form)
    assert orc.limbs_to_ints(a) == [M.fr_to_mont(v) for v in vals]
    b = orc.fr_from_ints(list(reversed(vals)))
    assert orc.fr_to_ints(orc.fr_mul(a, b)) == [x * y % M.R for x, y in zip(vals, reversed(vals))]
    assert orc.fr_to_ints(orc.fr_add(a, b)) == [(x + y) % M.R for x, y in zip(vals, reversed(vals))]
    assert orc.fr_to_ints(orc.fr_sub(a, b)) == [(x - y) % M.R for x, y in zip(vals, reversed(vals))]
    nz = [v for v in vals if v]
    assert orc.fr_to_ints(orc.fr_inv(orc.fr_from_ints(nz))) == [pow(v, -1, M.R) for v in nz]
    qv = [0, 1, M.P - 1, M.GX, M.GY]
    q = orc.fq_from_ints(qv)
    assert orc.limbs_to_ints(q) == [M.fq_to_mont(v) for v in qv]
    assert orc.fq_to_ints(orc.fq_mul(q, q)) == [v * v % M.P for v in qv]
    gv = [0, 1, M.GL - 1, 2 ** 48, 12345678901234567]
    g = orc.gl_from_ints(gv)
    assert [int(x) for x in g] == [M.gl_to_mont(v) for v in gv]
    assert orc.gl_to_ints(g) == gv


def test_rand_streams_match_model(orc):
    assert orc.fr_to_ints(orc.rand_fr(0x5EED, 50)) == M.rand_fr_list(0x5EED, 50)
    assert orc.gl_to_ints(orc.rand_gl(0x5EED, 50)) == M.rand_gl_list(0x5EED, 50)


def test_reference_kat_commit_17G(orc, golden):
    # kzg/src/commitment.rs:36-51
    two = orc.fr_from_ints([2])[0]
    srs = orc.srs(two, 13)
    coeffs = orc.fr_from_ints([1, 2, 3])
    out, inf = orc.msm_naive(srs, None, coeffs)
    assert not inf
    assert orc.points_to_ints(out)[0] == pt_from_hex(golden["reference_kat"]["seventeen_G"])
    out2, inf2 = orc.msm_pippenger(srs, None, coeffs)
    assert not inf2 and np.array_equal(out, out2)


def test_g1_mul_golden(orc, golden):
    g = orc.g1_generator()
    assert orc.points_to_ints(g)[0] == M.G1 and orc.g1_on_curve(g)
    for ent in golden["g1_mul"]:
        k = orc.fr_from_ints([hx(ent["k"])])[0]
        out, inf = orc.g1_mul(g, 0, k)
        exp = pt_from_hex(ent["out"])
        if exp is None:
            assert inf
        else:
            assert not inf and orc.points_to_ints(out)[0] == exp
    ks = orc.fr_from_ints([hx(e["k"]) for e in golden["g1_mul"]])
    xy, infs = orc.g1_fixed_base_mul(ks)
    assert orc.points_to_ints(xy, infs) == [pt_from_hex(e["out"]) for e in golden["g1_mul"]]


@pytest.mark.parametrize("algo", ["naive", "pippenger"])
def test_msm_golden(orc, golden, algo):
    for ent in golden["msm"]:
        xy, inf = orc.points_from_ints([pt_from_hex(p) for p in ent["points"]])
        sc = orc.fr_from_ints([hx(s) for s in ent["scalars"]])
        fn = orc.msm_naive if algo == "naive" else orc.msm_pippenger
        out, oinf = fn(xy, inf, sc)
        exp = pt_from_hex(ent["out"])
        assert (None if oinf else orc.points_to_ints(out)[0]) == exp


def test_msm_edge_cases(orc):
    g = orc.g1_generator().reshape(1, 12)
    out, inf = orc.msm_naive(g[:0], None, np.zeros((0, 4), dtype=np.uint64))
    assert inf  # empty => identity (scheme.rs:94)
    out, inf = orc.msm_naive(g, None, np.zeros((1, 4), dtype=np.uint64))
    assert inf
    # zip truncation: more scalars than points (scheme.rs:90-91)
    sc = orc.fr_from_ints([3, 5])
    out, inf = orc.msm_naive(g, None, sc)
    assert orc.points_to_ints(out)[0] == M.g1_mul(M.G1, 3)


def test_ntt_fr_golden(orc, golden):
    seven = orc.fr_from_ints([7])[0]
    for ent in golden["ntt_fr"]:
        a = orc.fr_from_ints([hx(v) for v in ent["in"]])
        assert orc.fr_to_ints(orc.fr_root_of_unity(ent["log_n"]).reshape(1, 4)) == [hx(ent["omega"])]
        assert orc.fr_to_ints(orc.ntt_fr(a)) == [hx(v) for v in ent["ntt"]]
        assert orc.fr_to_ints(orc.ntt_fr(a, inverse=True)) == [hx(v) for v in ent["intt"]]
        assert orc.fr_to_ints(orc.ntt_fr(a, coset=seven)) == [hx(v) for v in ent["coset7_ntt"]]
        assert orc.fr_to_ints(orc.ntt_fr(a, inverse=True, coset=seven)) == [hx(v) for v in ent["coset7_intt"]]


def test_ntt_goldilocks_and_fri_golden(orc, golden):
    seven = orc.gl_from_ints([7])
    for ent in golden["ntt_goldilocks"]:
        a = orc.gl_from_ints([hx(v) for v in ent["in"]])
        assert orc.gl_to_ints(orc.ntt_gl(a)) == [hx(v) for v in ent["ntt"]]
        assert orc.gl_to_ints(orc.ntt_gl(a, inverse=True)) == [hx(v) for v in ent["intt"]]
        assert orc.gl_to_ints(orc.ntt_gl(a, coset=seven)) == [hx(v) for v in ent["coset7_ntt"]]
    for ent in golden["fri_layer"]:
        c = orc.gl_from_ints([hx(v) for v in ent["coeffs"]])
        cs = orc.gl_from_ints([hx(ent["coset"])])[0]
        log_d = ent["domain"].bit_length() - 1
        assert orc.gl_to_ints(orc.fri_layer_eval(c, cs, log_d)) == [hx(v) for v in ent["evals"]]
        r5 = orc.gl_from_ints([5])[0]
        got = orc.gl_to_ints(orc.fri_fold(c, r5))
        assert M.poly_trim(got) == [hx(v) for v in ent["fold_r5"]]
    # reference KAT fri/src/prover.rs:181-192
    one = orc.gl_from_ints([1])[0]
    assert orc.gl_to_ints(orc.fri_fold(orc.gl_from_ints([1, 2, 3, 4]), one)) == [3, 7]


def test_poly_golden(orc, golden):
    for ent in golden["poly"]:
        a = orc.fr_from_ints([hx(v) for v in ent["a"]])
        b = orc.fr_from_ints([hx(v) for v in ent["b"]])
        assert M.poly_trim(orc.fr_to_ints(orc.poly_mul_fr(a, b))) == [hx(v) for v in ent["prod"]]
        mv = orc.fr_from_ints([hx(v) for v in ent["a_times_zh4"]])
        q, rem = orc.divide_by_vanishing_fr(mv, 4)
        assert M.poly_trim(orc.fr_to_ints(q)) == M.poly_trim([hx(v) for v in ent["a"]])
        assert all(v == 0 for v in orc.fr_to_ints(rem))


def test_kzg_open_golden(orc, golden):
    ent = golden["kzg_open"]
    xy, inf = orc.points_from_ints([pt_from_hex(p) for p in ent["points"]])
    c = orc.fr_from_ints([hx(v) for v in ent["coeffs"]])
    z = orc.fr_from_ints([hx(ent["z"])])[0]
    y = orc.poly_eval_fr(c, z)
    assert orc.fr_to_ints(y.reshape(1, 4)) == [hx(ent["eval"])]
    q = orc.poly_div_linear_fr(c, z)
    w, winf = orc.msm_naive(xy, inf, q)
    assert orc.points_to_ints(w)[0] == pt_from_hex(ent["opening"])
    cm, _ = orc.msm_naive(xy, inf, c)
    assert orc.points_to_ints(cm)[0] == pt_from_hex(ent["commit"])


def test_multithreaded_context_baseline_matches_the_oracle(orc):
    """bench.py's cpu_baseline.context (OpenMP bucket method / radix-2 stages on all cores) computes what the single-threaded oracle
    computes: same affine point, same transform, for thread counts that do and do not divide the work."""
    import numpy as np
    n = 3000
    ks = orc.rand_fr(0xC0DE, n)
    pts, inf = orc.g1_fixed_base_mul(ks)
    sc = orc.rand_fr(0xC0DF, n)
    sc[0] = 0
    exp = orc.msm_pippenger(pts, inf, sc)
    for threads in (1, 3, 8):
        got = orc.msm_pippenger_mt(pts, inf, sc, threads)
        assert got[1] == exp[1] and np.array_equal(got[0], exp[0])
    assert orc.msm_pippenger_mt(pts[:0], None, sc[:0], 4)[1] == 1
    g = orc.rand_fr(99, 1)[0]
    for log_n in (0, 1, 5, 12):
        a = orc.rand_fr(0xC0E0 + log_n, 1 << log_n)
        for threads in (1, 3, 8):
            assert np.array_equal(orc.ntt_fr_mt(a, threads), orc.ntt_fr(a))
            assert np.array_equal(orc.ntt_fr_mt(a, threads, inverse=True), orc.ntt_fr(a, inverse=True))
            assert np.array_equal(orc.ntt_fr_mt(a, threads, coset=g), orc.ntt_fr(a, coset=g))
            assert np.array_equal(orc.ntt_fr_mt(a, threads, inverse=True, coset=g), orc.ntt_fr(a, inverse=True, coset=g))
    assert orc.max_threads() >= 1
