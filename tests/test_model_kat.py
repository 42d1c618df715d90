"""The independent big-int model reproduces the reference's own known-answer tests."""
import bigmodel as M
from conftest import hx, pt_from_hex


def test_kzg_commit_kat_17G():
    # kzg/src/commitment.rs:36-51: secret 2, p = 1 + 2X + 3X^2 => commitment == [p(2)]G == 17*G
    srs = M.srs(2, 10)
    assert len(srs) == 13  # circuit_size + 3, kzg/src/srs.rs:51
    c = M.msm_naive([1, 2, 3], srs)
    assert M.poly_eval([1, 2, 3], 1, M.R) == 6  # commitment.rs:44
    assert c == M.g1_mul(M.G1, 17)
    # coordinates from SURVEY.md Appendix A
    assert c == (0x1098f178f84fc753a76bb63709e9be91eec3ff5f7f3a5f4836f34fe8a1a6d6c5578d8fd820573cef3a01e2bfef3eaf3a,
                 0x0ea923110b733b531006075f796cc9368f2477fe26020f465468efbb380ce1f8eebaf5c770f31d320f9bd378dc758436)
    assert M.g1_on_curve(c)


def test_kzg_open_trapdoor_identity():
    # the pairing check of kzg/src/scheme.rs:155-171 collapses, with known s, to W == [(p(s)-y)/(s-z)]G
    s, z = 2, 1
    srs = M.srs(s, 10)
    w, y = M.kzg_open([1, 2, 3], z, srs)
    q = (M.poly_eval([1, 2, 3], s, M.R) - y) * pow(s - z, -1, M.R) % M.R
    assert w == M.g1_mul(M.G1, q)


def test_msm_edge_cases():
    assert M.msm_naive([], M.srs(2, 0)) is M.INF  # empty => identity, scheme.rs:94
    assert M.msm_naive([0, 0], M.srs(2, 0)) is M.INF
    assert M.g1_mul(M.G1, M.R) is M.INF
    assert M.g1_add(M.G1, M.g1_neg(M.G1)) is M.INF


def test_fri_kats():
    # fri/src/prover.rs:181-192
    assert M.fri_fold([1, 2, 3, 4], 1) == [3, 7]
    # fri/src/prover.rs:195-205: layer-1 coset = GENERATOR^2 = 49
    assert M.GL_GENERATOR ** 2 % M.GL == 49
    # SURVEY.md §8c golden: layer-0 evaluations of 1+2X+3X^2+4X^3 on 7*<omega_4>, omega_4 = 2^48
    assert M.root_of_unity(2, M.GL) == 2 ** 48
    assert M.fri_layer_eval([1, 2, 3, 4], 7, 4) == [1534, 18064501051041513327, 18446744069414583083,
                                                    382243018373070702]


def test_fri_fold_matches_evaluation_domain_fold():
    # fri/src/verifier.rs:93-96: q'(x^2) = (q(x)+q(-x))/2 + r (q(x)-q(-x))/(2x)
    coeffs, r, x = [5, 9, 11, 2, 8, 1], 12345, 987654321
    f = M.fri_fold(coeffs, r)
    qx, qmx = M.poly_eval(coeffs, x, M.GL), M.poly_eval(coeffs, (-x) % M.GL, M.GL)
    inv2, inv2x = pow(2, -1, M.GL), pow(2 * x, -1, M.GL)
    assert M.poly_eval(f, x * x % M.GL, M.GL) == ((qx + qmx) * inv2 + r * (qx - qmx) * inv2x) % M.GL


def test_slice_polynomial_kat():
    # plonk/src/slice_polynomial.rs:80-111: 12 coefficients => 3 slices of 4, degree 3
    coeffs = list(range(1, 13))
    slices, deg = M.slice_poly(coeffs)
    assert [len(s) for s in slices] == [4, 4, 4] and deg == 3
    zeta = 5
    compact = M.slice_compact(slices, deg, zeta)
    assert M.poly_eval(compact, zeta, M.R) == M.poly_eval(coeffs, zeta, M.R)


def test_constants_appendix_a():
    assert M.FR_ROOT == 10238227357739495823651030575849232062558860180284477541189508159991286009131
    assert M.root_of_unity(2) == 3465144826073652318776269530687742778270252468765361963008
    assert M.root_of_unity(24) == 18596002123094854211120822350746157678791770803088570110573239418060655130524
    assert M.GL_ROOT == 1753635133440165772
    assert M.FR_MONT_R == 0x1824b159acc5056f998c4fefecbc4ff55884b7fa0003480200000001fffffffe
    assert M.FQ_MONT_R == 0x15f65ec3fa80e4935c071a97a256ec6d77ce5853705257455f48985753c758baebf4000bc40c0002760900000002fffd
    assert M.g1_mul(M.G1, M.R) is M.INF and M.g1_on_curve(M.G1)


def test_golden_file_is_current(golden):
    k = golden["reference_kat"]
    assert pt_from_hex(k["commit_1_2_3_secret2"]) == M.g1_mul(M.G1, 17)
    assert [hx(v) for v in k["fri_fold_1234_r1"]] == [3, 7]
    assert hx(k["fri_layer1_coset"]) == 49
