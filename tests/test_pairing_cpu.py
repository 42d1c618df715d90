"""Host-side pairing and KZG verifiers (no GPU needed) against the independent big-int pairing model
(tests/model/pairing_model.py) and the reference's own verification tests (kzg/src/commitment.rs:36-119)."""
import numpy as np
import pytest

import bigmodel as M
import pairing_model as PM

P, R = M.P, M.R
RQ_INV = pow(2 ** 384, -1, P)


@pytest.fixture(scope="module")
def zkp():
    import zkp_hip
    zkp_hip.lib()
    return zkp_hip


def fq_canon(limbs):
    return sum(int(l) << (64 * i) for i, l in enumerate(limbs)) * RQ_INV % P


def g2_to_ints(xy):
    c = [fq_canon(xy[6 * i:6 * i + 6]) for i in range(4)]
    return ((c[0], c[1]), (c[2], c[3]))


def g2_from_ints(pt):
    vals = [pt[0][0], pt[0][1], pt[1][0], pt[1][1]]
    out = np.zeros(24, dtype=np.uint64)
    for i, v in enumerate(vals):
        m = v * 2 ** 384 % P
        for k in range(6):
            out[6 * i + k] = (m >> (64 * k)) & (2 ** 64 - 1)
    return out


def test_g2_generator_and_mul_match_model(zkp, orc):
    g = zkp.g2_generator()
    assert g2_to_ints(g) == PM.G2 and PM.g2_on_curve(PM.G2)
    for k in (1, 2, 3, 0xDEADBEEF, R - 1):
        out, inf = zkp.g2_mul(g, orc.fr_from_ints([k])[0])
        assert not inf and g2_to_ints(out) == PM.g2_mul(PM.G2, k)
    out, inf = zkp.g2_mul(g, orc.fr_from_ints([0])[0])
    assert inf
    bad = g.copy()
    bad[0] ^= np.uint64(1)
    with pytest.raises(zkp.ZkpError):
        zkp.g2_mul(bad, orc.fr_from_ints([5])[0])


def test_pairing_matches_model_bit_for_bit(zkp, orc):
    g1xy, _ = orc.points_from_ints([M.G1])
    got = zkp.pairing(g1xy[0], zkp.g2_generator())
    want = PM.f12_flat(PM.pairing(M.G1, PM.G2))
    assert [fq_canon(row) for row in got] == want
    a, b = 0x1234567, 0xFEDCBA987
    pa, _ = orc.points_from_ints([M.g1_mul(M.G1, a)])
    qb = g2_from_ints(PM.g2_mul(PM.G2, b))
    got2 = [fq_canon(row) for row in zkp.pairing(pa[0], qb)]
    assert got2 == PM.f12_flat(PM.f12_pow(PM.pairing(M.G1, PM.G2), a * b % R))  # bilinear
    one = [1] + [0] * 11
    assert [fq_canon(r) for r in zkp.pairing(g1xy[0], zkp.g2_generator(), p_is_inf=1)] == one
    assert [fq_canon(r) for r in zkp.pairing(g1xy[0], zkp.g2_generator(), q_is_inf=1)] == one


def test_kzg_verify_reference_cases(zkp, orc):
    # kzg/src/commitment.rs:36-53: SRS from secret 2, p = 1 + 2X + 3X^2, opening at z = 1 verifies; wrong value is rejected
    s = 2
    g2s = g2_from_ints(PM.g2_mul(PM.G2, s))
    coeffs, z = [1, 2, 3], 1
    pts = M.srs(s, 13)
    w, y = M.kzg_open(coeffs, z, pts)
    commit = M.msm_naive(coeffs, pts)
    cxy, _ = orc.points_from_ints([commit])
    wxy, winf = orc.points_from_ints([w])
    f = lambda v: orc.fr_from_ints([v])[0]
    assert y == 6 and zkp.kzg_verify(g2s, (cxy[0], 0), (wxy[0], int(winf[0])), f(y), f(z))
    assert not zkp.kzg_verify(g2s, (cxy[0], 0), (wxy[0], int(winf[0])), f(y + 1), f(z))
    assert not zkp.kzg_verify(g2s, (cxy[0], 0), (wxy[0], int(winf[0])), f(y), f(z + 1))
    other = g2_from_ints(PM.g2_mul(PM.G2, 3))
    assert not zkp.kzg_verify(other, (cxy[0], 0), (wxy[0], int(winf[0])), f(y), f(z))


def test_kzg_batch_verify(zkp, orc):
    # kzg/src/commitment.rs:95-119: several (commitment, point, opening) triples under random weights
    s = 0xABCDEF
    g2s = g2_from_ints(PM.g2_mul(PM.G2, s))
    pts = M.srs(s, 11)
    cms, ws, zs, ys = [], [], [], []
    for k in range(3):
        coeffs = M.rand_fr_list(900 + k, 8)
        z = M.rand_fr_list(950 + k, 1)[0]
        w, y = M.kzg_open(coeffs, z, pts)
        cms.append(M.msm_naive(coeffs, pts)); ws.append(w); zs.append(z); ys.append(y)
    cxy, _ = orc.points_from_ints(cms)
    wxy, _ = orc.points_from_ints(ws)
    rp = orc.fr_from_ints([3, 2 ** 100 + 7, 2 ** 127 - 1])
    assert zkp.kzg_batch_verify(g2s, cxy, orc.fr_from_ints(zs), wxy, orc.fr_from_ints(ys), rp)
    ys[1] = (ys[1] + 1) % R
    assert not zkp.kzg_batch_verify(g2s, cxy, orc.fr_from_ints(zs), wxy, orc.fr_from_ints(ys), rp)


def test_verifiers_reject_invalid_g1_inputs(zkp, orc):
    """The C ABI is the deserialisation boundary: a G1 input that is off the curve, has non-canonical limbs or lies outside the
    prime-order subgroup must be refused with ZKP_E_ARG before any pairing runs (arkworks validates a `G1Affine` when it is
    built; kzg/src/scheme.rs:143-160 only ever sees valid points)."""
    s = 2
    g2s = g2_from_ints(PM.g2_mul(PM.G2, s))
    pts = M.srs(s, 13)
    w, y = M.kzg_open([1, 2, 3], 1, pts)
    commit = M.msm_naive([1, 2, 3], pts)
    cxy, _ = orc.points_from_ints([commit])
    wxy, _ = orc.points_from_ints([w])
    f = lambda v: orc.fr_from_ints([v])[0]
    assert zkp.kzg_verify(g2s, (cxy[0], 0), (wxy[0], 0), f(y), f(1))
    # (a) off the curve: y + 1
    bad, _ = orc.points_from_ints([(commit[0], (commit[1] + 1) % M.P)])
    # (b) non-canonical: the same residue class, limbs x + p
    noncanon = cxy[0].copy()
    xm = sum(int(v) << (64 * i) for i, v in enumerate(cxy[0][:6])) + M.P
    assert xm < 1 << 384
    noncanon[:6] = [(xm >> (64 * i)) & (2 ** 64 - 1) for i in range(6)]
    # (c) on the curve, outside the r-torsion (the cofactor of G1 is ~2^126: a random curve point is outside)
    x = 1
    while True:
        rhs = (x * x * x + 4) % M.P
        yy = pow(rhs, (M.P + 1) // 4, M.P)
        if yy * yy % M.P == rhs:
            break
        x += 1
    out_sub, _ = orc.points_from_ints([(x, yy)])
    assert orc.g1_on_curve(out_sub[0])
    for name, pt in (("off-curve", bad[0]), ("non-canonical", noncanon), ("wrong subgroup", out_sub[0])):
        for args in (((pt, 0), (wxy[0], 0)), ((cxy[0], 0), (pt, 0))):
            with pytest.raises(zkp.ZkpError) as ei:
                zkp.kzg_verify(g2s, args[0], args[1], f(y), f(1))
            assert ei.value.code == zkp.ZKP_E_ARG, name
        with pytest.raises(zkp.ZkpError):
            zkp.kzg_batch_verify(g2s, np.stack([pt]), orc.fr_from_ints([1]), wxy, orc.fr_from_ints([y]), orc.fr_from_ints([3]))
    # the identity is a valid input (flag, no coordinates to check)
    assert not zkp.kzg_verify(g2s, (cxy[0], 1), (wxy[0], 0), f(y), f(1))


def _f2_sqrt(a):
    """Square root in Fq2 = Fq[u]/(u^2 + 1) (p = 3 mod 4), or None."""
    a0, a1 = a
    sq = lambda v: (lambda r: r if r * r % P == v % P else None)(pow(v, (P + 1) // 4, P))
    if a1 == 0:
        r = sq(a0)
        return (r, 0) if r is not None else (lambda t: None if t is None else (0, t))(sq(-a0 % P))
    s = sq((a0 * a0 + a1 * a1) % P)
    if s is None:
        return None
    inv2 = pow(2, -1, P)
    for t in ((a0 + s) * inv2 % P, (a0 - s) * inv2 % P):
        y0 = sq(t)
        if y0:
            return (y0, a1 * pow(2 * y0, -1, P) % P)
    return None


def test_verifiers_reject_invalid_g2_inputs(zkp, orc):
    """[s]_2 and every other caller-supplied G2 point: canonical limbs, on the twist y^2 = x^3 + 4(1 + u) AND in the prime-order
    subgroup -- the twist's cofactor is ~2^381, a random curve point is outside (the verifiers checked on_curve() only)."""
    s = 2
    g2s = g2_from_ints(PM.g2_mul(PM.G2, s))
    pts = M.srs(s, 13)
    w, y = M.kzg_open([1, 2, 3], 1, pts)
    cxy, _ = orc.points_from_ints([M.msm_naive([1, 2, 3], pts)])
    wxy, _ = orc.points_from_ints([w])
    f = lambda v: orc.fr_from_ints([v])[0]
    assert zkp.kzg_verify(g2s, (cxy[0], 0), (wxy[0], 0), f(y), f(1))
    # on the twist, outside the r-torsion
    x = (1, 1)
    while True:
        rhs = PM.f2_add(PM.f2_mul(PM.f2_mul(x, x), x), (4, 4))
        yy = _f2_sqrt(rhs)
        if yy is not None and PM.f2_mul(yy, yy) == rhs:
            break
        x = (x[0] + 1, x[1])
    q = (x, yy)
    rq = None
    for bit in bin(R)[2:]:                     # [r]Q without the model's reduction of the scalar mod r
        rq = PM.g2_double(rq)
        if bit == "1":
            rq = PM.g2_add(rq, q)
    assert PM.g2_on_curve(q) and rq is not None
    outside = g2_from_ints(q)
    off = g2s.copy()
    off[0] ^= np.uint64(1)
    noncanon = g2s.copy()
    xm = sum(int(v) << (64 * i) for i, v in enumerate(g2s[:6])) + P
    assert xm < 1 << 384
    noncanon[:6] = [(xm >> (64 * i)) & (2 ** 64 - 1) for i in range(6)]
    g1xy, _ = orc.points_from_ints([M.G1])
    for name, bad in (("wrong subgroup", outside), ("off-curve", off), ("non-canonical", noncanon)):
        for call in (lambda: zkp.kzg_verify(bad, (cxy[0], 0), (wxy[0], 0), f(y), f(1)),
                     lambda: zkp.kzg_batch_verify(bad, cxy, orc.fr_from_ints([1]), wxy, orc.fr_from_ints([y]), orc.fr_from_ints([3])),
                     lambda: zkp.pairing(g1xy[0], bad),
                     lambda: zkp.g2_mul(bad, f(5))):
            with pytest.raises(zkp.ZkpError) as ei:
                call()
            assert ei.value.code == zkp.ZKP_E_ARG, name
        assert "G2" in str(ei.value) or "[s]_2" in str(ei.value)


def test_kzg_aggregate_commitments_reference_case(zkp, orc):
    """kzg/src/commitment.rs:78-89: aggregate([c1, c2], challenge) == c1 + c2 * challenge (and the general sum_i ch^i C_i of
    scheme.rs:187-202, with an identity among the inputs)."""
    pts = M.srs(0x5EC, 8)
    c1 = M.msm_naive([1, 2, 3, 4, 5], pts)
    c2 = M.msm_naive([1, 2, 3, 4, 8], pts)
    c3 = M.msm_naive([7, 0, 0, 9], pts)
    ch = M.rand_fr_list(0xA66, 1)[0] >> 127  # Fr::from(u128)
    cxy, _ = orc.points_from_ints([c1, c2])
    out, inf = zkp.kzg_aggregate_commitments(cxy, orc.fr_from_ints([ch])[0])
    exp = M.g1_add(c1, M.g1_mul(c2, ch))
    assert not inf and orc.points_to_ints(out.reshape(1, 12))[0] == exp
    cxy3, _ = orc.points_from_ints([c1, c2, c3, c3])
    out, inf = zkp.kzg_aggregate_commitments(cxy3, orc.fr_from_ints([ch])[0], is_inf=[0, 1, 0, 0])
    exp = M.g1_add(M.g1_add(c1, M.g1_mul(c3, ch * ch % R)), M.g1_mul(c3, pow(ch, 3, R)))
    assert not inf and orc.points_to_ints(out.reshape(1, 12))[0] == exp
    out, inf = zkp.kzg_aggregate_commitments(np.zeros((0, 12), dtype=np.uint64), orc.fr_from_ints([ch])[0])
    assert inf                                                                   # G1Projective::zero()
    bad = cxy.copy()
    bad[1][6] ^= np.uint64(1)
    with pytest.raises(zkp.ZkpError):
        zkp.kzg_aggregate_commitments(bad, orc.fr_from_ints([ch])[0])
