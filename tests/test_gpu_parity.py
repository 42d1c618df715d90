"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors and the CPU oracle, bit-exact.

Run on an MI355X with `pytest -m gpu`.  Nothing here reads /root/reference.
"""
import numpy as np
import pytest

import bigmodel as M
from conftest import hx, pt_from_hex

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zkp():
    import torch
    assert torch.cuda.is_available(), "no GPU"
    import zkp_hip
    zkp_hip.init()
    return zkp_hip


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def host(t, cols=None):
    a = t.cpu().numpy().view(np.uint64)
    return a.reshape(-1, cols) if cols else a


# ----------------------------------------------------------------------------- NTT over Fr
def test_ntt_fr_golden(zkp, orc, golden):
    seven = orc.fr_from_ints([7])[0]
    for ent in golden["ntt_fr"]:
        a = orc.fr_from_ints([hx(v) for v in ent["in"]])
        assert orc.fr_to_ints(zkp.ntt_fr(a)) == [hx(v) for v in ent["ntt"]]
        assert orc.fr_to_ints(zkp.ntt_fr(a, inverse=True)) == [hx(v) for v in ent["intt"]]
        assert orc.fr_to_ints(zkp.ntt_fr(a, coset=seven)) == [hx(v) for v in ent["coset7_ntt"]]
        assert orc.fr_to_ints(zkp.ntt_fr(a, inverse=True, coset=seven)) == [hx(v) for v in ent["coset7_intt"]]


@pytest.mark.parametrize("log_n", [6, 9, 11, 12, 13, 15, 16, 17, 18, 20])
def test_ntt_fr_vs_oracle(zkp, orc, log_n):
    # 11 = largest single-pass size, 12..16 two passes, 17+ three passes
    a = orc.rand_fr(0x01770000 + log_n, 1 << log_n)
    g = orc.rand_fr(99, 1)[0]
    assert np.array_equal(zkp.ntt_fr(a), orc.ntt_fr(a))
    assert np.array_equal(zkp.ntt_fr(a, inverse=True), orc.ntt_fr(a, inverse=True))
    assert np.array_equal(zkp.ntt_fr(a, coset=g), orc.ntt_fr(a, coset=g))
    assert np.array_equal(zkp.ntt_fr(a, inverse=True, coset=g), orc.ntt_fr(a, inverse=True, coset=g))


def test_ntt_fr_edge_values(zkp, orc):
    # all-zero, all r-1, single spike
    n = 1 << 12
    z = np.zeros((n, 4), dtype=np.uint64)
    assert np.array_equal(zkp.ntt_fr(z), z)
    m1 = np.tile(orc.fr_from_ints([M.R - 1]), (n, 1))
    assert np.array_equal(zkp.ntt_fr(m1), orc.ntt_fr(m1))
    spike = z.copy()
    spike[1] = orc.fr_from_ints([1])[0]
    assert np.array_equal(zkp.ntt_fr(spike), orc.ntt_fr(spike))


def test_ntt_fr_batch_dev_roundtrip(zkp, orc):
    log_n, batch = 14, 3
    a = orc.rand_fr(5, batch << log_n)
    t = dev(a)
    zkp.ntt_fr_dev(t, log_n, batch)
    fwd = host(t, 4)
    for b in range(batch):
        sl = slice(b << log_n, (b + 1) << log_n)
        assert np.array_equal(fwd[sl], orc.ntt_fr(a[sl]))
    zkp.ntt_fr_dev(t, log_n, batch, inverse=True)
    assert np.array_equal(host(t, 4), a)


def test_ntt_fr_full_size_roundtrip_and_linearity(zkp, orc):
    """BASELINE config 3 size (2^24): NTT -> iNTT is the identity and the transform is linear."""
    import torch
    log_n = 24
    n = 1 << log_n
    a = orc.rand_fr(0x01770018, n)
    t = dev(a)
    zkp.ntt_fr_dev(t, log_n)
    fwd = t.clone()
    zkp.ntt_fr_dev(t, log_n, inverse=True)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.from_numpy(a.view(np.int64)))
    # every output against the oracle's ark-poly style radix-2 transform (plonk/src/prover.rs:374-375,396-426 semantics), ~5 s of CPU
    fh = host(fwd, 4)
    assert np.array_equal(fh, orc.ntt_fr(a))
    # and four of them against the definition X[k] = sum_j a_j w^{jk} (Horner on the oracle): independent of either transform
    w = orc.fr_root_of_unity(log_n)
    for k in (0, 1, 12345, n - 1):
        wk = orc.fr_from_ints([pow(orc.fr_to_ints(w.reshape(1, 4))[0], k, M.R)])[0]
        assert np.array_equal(orc.poly_eval_fr(a, wk), fh[k])


def test_ntt_fr_wide_radix_passes_vs_oracle(zkp, orc):
    """The radix-2^9 two-column passes are only reached by 2^25..2^27 single transforms and by batches of 2^17 / 2^18 transforms with
    >= 2^19 elements per launch (api.hip: get_plan, allow_wide): one 2^25 transform (forward, and inverse on a coset) and a batch of
    two 2^18 transforms, every output against the oracle."""
    import torch
    log_n = 25
    a = orc.rand_fr(0x01770019, 1 << log_n)
    t = dev(a)
    zkp.ntt_fr_dev(t, log_n)
    assert np.array_equal(host(t, 4), orc.ntt_fr(a))
    g = orc.rand_fr(98, 1)[0]
    t = dev(a)
    zkp.ntt_fr_dev(t, log_n, inverse=True, coset=g)
    assert np.array_equal(host(t, 4), orc.ntt_fr(a, inverse=True, coset=g))
    del t
    torch.cuda.empty_cache()
    log_b, batch = 18, 2
    b = orc.rand_fr(0x01770012, batch << log_b)
    for inverse in (False, True):
        t = dev(b)
        zkp.ntt_fr_dev(t, log_b, batch, inverse=inverse)
        got = host(t, 4)
        for i in range(batch):
            sl = slice(i << log_b, (i + 1) << log_b)
            assert np.array_equal(got[sl], orc.ntt_fr(b[sl], inverse=inverse))


# ----------------------------------------------------------------------------- Goldilocks / FRI
def test_ntt_goldilocks_golden_and_fri(zkp, orc, golden):
    seven = orc.gl_from_ints([7])
    for ent in golden["ntt_goldilocks"]:
        a = orc.gl_from_ints([hx(v) for v in ent["in"]])
        assert orc.gl_to_ints(zkp.ntt_goldilocks(a)) == [hx(v) for v in ent["ntt"]]
        assert orc.gl_to_ints(zkp.ntt_goldilocks(a, inverse=True)) == [hx(v) for v in ent["intt"]]
        assert orc.gl_to_ints(zkp.ntt_goldilocks(a, coset=seven)) == [hx(v) for v in ent["coset7_ntt"]]
    for ent in golden["fri_layer"]:
        c = orc.gl_from_ints([hx(v) for v in ent["coeffs"]])
        cs = orc.gl_from_ints([hx(ent["coset"])])[0]
        log_d = ent["domain"].bit_length() - 1
        assert orc.gl_to_ints(zkp.fri_layer_eval(c, cs, log_d)) == [hx(v) for v in ent["evals"]]
        r5 = orc.gl_from_ints([5])[0]
        assert M.poly_trim(orc.gl_to_ints(zkp.fri_fold(c, r5))) == [hx(v) for v in ent["fold_r5"]]
    # reference KAT fri/src/prover.rs:181-192
    one = orc.gl_from_ints([1])[0]
    assert orc.gl_to_ints(zkp.fri_fold(orc.gl_from_ints([1, 2, 3, 4]), one)) == [3, 7]


@pytest.mark.parametrize("log_n", [5, 12, 13, 14, 16, 17, 20, 22, 24])
def test_ntt_goldilocks_vs_oracle(zkp, orc, log_n):
    a = orc.rand_gl(0x600D0000 + log_n, 1 << log_n)
    g = orc.rand_gl(7, 1)
    assert np.array_equal(zkp.ntt_goldilocks(a), orc.ntt_gl(a))
    assert np.array_equal(zkp.ntt_goldilocks(a, inverse=True), orc.ntt_gl(a, inverse=True))
    assert np.array_equal(zkp.ntt_goldilocks(a, coset=g), orc.ntt_gl(a, coset=g))
    assert np.array_equal(zkp.ntt_goldilocks(a, inverse=True, coset=g), orc.ntt_gl(a, inverse=True, coset=g))


def test_fri_layer_vs_horner_oracle(zkp, orc):
    # the reference's O(D*d) Horner loop (fri_layer.rs:40-46) against the coset NTT
    c = orc.rand_gl(3, 300)
    cs = orc.gl_from_ints([7])[0]
    assert np.array_equal(zkp.fri_layer_eval(c, cs, 10), orc.fri_layer_eval(c, cs, 10))


# ----------------------------------------------------------------------------- MSM
def test_msm_reference_kat_17G(zkp, orc, golden):
    # kzg/src/commitment.rs:36-51 through the KzgScheme mirror: SRS from secret 2 (GPU fixed-base), p = 1+2X+3X^2
    two = orc.fr_from_ints([2])[0]
    srs = zkp.Srs.new_from_secret(two, 10)
    assert srs.g1_points().shape[0] == 13
    assert np.array_equal(srs.g1_points(), orc.srs(two, 13))
    scheme = zkp.KzgScheme(srs)
    (xy, inf) = scheme.commit(orc.fr_from_ints([1, 2, 3]))
    assert not inf and orc.points_to_ints(xy)[0] == pt_from_hex(golden["reference_kat"]["seventeen_G"])
    (w, winf), ev = scheme.open(orc.fr_from_ints([1, 2, 3]), orc.fr_from_ints([1])[0])
    assert orc.fr_to_ints(ev.reshape(1, 4)) == [6]
    assert orc.points_to_ints(w)[0] == pt_from_hex(golden["reference_kat"]["open_at_1"])


def test_msm_golden(zkp, orc, golden):
    for ent in golden["msm"]:
        xy, inf = orc.points_from_ints([pt_from_hex(p) for p in ent["points"]])
        sc = orc.fr_from_ints([hx(s) for s in ent["scalars"]])
        bases = zkp.G1Bases.from_host(xy, inf)
        out, oinf = zkp.msm_g1(bases, sc)
        assert (None if oinf else orc.points_to_ints(out)[0]) == pt_from_hex(ent["out"])


def test_msm_edge_cases(zkp, orc):
    g = orc.g1_generator().reshape(1, 12)
    bases = zkp.G1Bases.from_host(np.tile(g, (4, 1)))
    out, inf = zkp.msm_g1(bases, np.zeros((0, 4), dtype=np.uint64))
    assert inf  # empty => identity (scheme.rs:94)
    out, inf = zkp.msm_g1(bases, np.zeros((3, 4), dtype=np.uint64))
    assert inf  # all-zero scalars
    with pytest.raises(zkp.ZkpError) as ei:  # more scalars than bases: the assert at scheme.rs:86
        zkp.msm_g1(bases, orc.fr_from_ints([1, 2, 3, 4, 5]))
    assert ei.value.code == zkp.ZKP_E_SIZE
    # repeated base, scalars r-1, 1, 1, 0  => (r-1+2) G = G
    out, inf = zkp.msm_g1(bases, orc.fr_from_ints([M.R - 1, 1, 1, 0]))
    assert not inf and orc.points_to_ints(out)[0] == M.G1
    # k*G + (r-k)*G = identity
    out, inf = zkp.msm_g1(bases, orc.fr_from_ints([12345, M.R - 12345]))
    assert inf


@pytest.mark.parametrize("expand", [0, 16])
def test_msm_starved_bucket_reduction_reports_device_error_not_a_wrong_sum(zkp, orc, expand):
    """The last levels of the bucket reduction run in one launch whose workgroups meet at a spinning device-scope barrier; a
    workgroup that never became resident (device shared with another job) must end in ZKP_E_DEVICE, never in a hang or a wrong
    point.  ZKP_TEST_TAIL_STARVE makes the barrier wait for one arrival more than there are workgroups, with a short time-out: the
    MSM_TAIL_TIMEOUT flag travels home with the results and the entry fails; the next MSM (counters re-zeroed) is exact again."""
    import os
    n = 3000
    ks = orc.rand_fr(0x57A0, n)
    sc = orc.rand_fr(0x57A1, n)
    import torch
    t_pts = torch.zeros(n * 12, dtype=torch.int64, device="cuda")
    zkp.g1_fixed_base_mul_dev(dev(ks), n, t_pts)
    bases = zkp.G1Bases.from_device(t_pts, n)
    if expand:
        bases.precompute(expand)
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc, ks))
    os.environ["ZKP_TEST_TAIL_STARVE"] = "1"
    try:
        with pytest.raises(zkp.ZkpError) as ei:
            zkp.msm_g1(bases, sc)
        assert ei.value.code == zkp.ZKP_E_DEVICE and "did not all become resident" in str(ei.value)
    finally:
        del os.environ["ZKP_TEST_TAIL_STARVE"]
    out, inf = zkp.msm_g1(bases, sc)
    assert inf == einf and np.array_equal(out, exp)


@pytest.mark.parametrize("expand", [0, 16])
def test_msm_equal_and_opposite_bucket_sums_meet_in_the_reduction(zkp, orc, expand):
    """The log-depth bucket reduction adds NEIGHBOURING buckets: with repeated base points and digits 1, 2 (and 2, 4 one level
    up) its adds see two equal points (doubling path) or two opposite points (cancellation) -- n = 64 so that the 16-bit-window
    geometry with its full-size reduction launches is used, not the small-problem kernels."""
    n = 64
    g = orc.g1_generator().reshape(1, 12)
    pts = np.tile(g, (n, 1))
    neg_g, _ = orc.points_from_ints([(M.G1[0], (-M.G1[1]) % M.P)])
    pts[3] = neg_g[0]
    pts[7] = neg_g[0]
    ints = [0] * n
    ints[0], ints[1] = 1, 2                      # window 0: buckets 1 and 2 both hold G        -> G + G
    ints[2], ints[3] = 1 << 16, 2 << 16          # window 1: buckets 1 and 2 hold G and -G        -> G + (-G)
    ints[4], ints[5] = 2 << 32, 4 << 32          # window 2: buckets 2 and 4 (odd-sum array) hold G, G
    ints[6], ints[7] = 2 << 48, 4 << 48          # window 3: buckets 2 and 4 hold G and -G
    ints[8] = M.R - 1                            # and a full-width scalar across every window
    sc = orc.fr_from_ints(ints)
    bases = zkp.G1Bases.from_host(pts)
    if expand:
        bases.precompute(expand)
    out, inf = zkp.msm_g1(bases, sc)
    exp, einf = orc.msm_pippenger(pts, None, sc)
    assert inf == einf and np.array_equal(out, exp)
    k = (1 + 2 + (1 << 16) - (2 << 16) + (2 << 32) + (4 << 32) + (2 << 48) - (4 << 48) + M.R - 1) % M.R
    assert orc.points_to_ints(out)[0] == M.g1_mul(M.G1, k)


@pytest.mark.parametrize("n,seed", [(1, 1), (2, 2), (33, 3), (1000, 4), (5000, 5), (1 << 14, 6)])
def test_msm_vs_oracle(zkp, orc, n, seed):
    ks = orc.rand_fr(0xBA5E0000 + seed, n)
    pts, pinf = orc.g1_fixed_base_mul(ks)
    sc = orc.rand_fr(0x5EED0000 + seed, n)
    if n >= 33:  # adversarial digits
        sc[0] = 0
        sc[1] = orc.fr_from_ints([1])[0]
        sc[2] = orc.fr_from_ints([M.R - 1])[0]
        sc[3] = sc[4]
        pts[6] = pts[5]  # repeated point with different scalars
        sc[7:20] = orc.fr_from_ints([3])[0]  # many equal small scalars -> one crowded bucket
    bases = zkp.G1Bases.from_host(pts)
    out, inf = zkp.msm_g1(bases, sc)
    exp, einf = orc.msm_pippenger(pts, None, sc)
    assert inf == einf and np.array_equal(out, exp)
    if n <= 1000:
        exp2, einf2 = orc.msm_naive(pts, None, sc)  # the reference-faithful path
        assert einf2 == einf and np.array_equal(exp, exp2)


def test_fixed_base_mul_vs_oracle(zkp, orc):
    n = 3000
    ks = orc.rand_fr(11, n)
    ks[0] = 0
    ks[1] = orc.fr_from_ints([1])[0]
    ks[2] = orc.fr_from_ints([M.R - 1])[0]
    import torch
    t_out = torch.zeros(n * 12, dtype=torch.int64, device="cuda")
    t_inf = torch.zeros(n, dtype=torch.uint8, device="cuda")
    zkp.g1_fixed_base_mul_dev(dev(ks), n, t_out, t_inf)
    exp, einf = orc.g1_fixed_base_mul(ks)
    assert np.array_equal(t_inf.cpu().numpy(), einf)
    assert np.array_equal(host(t_out, 12), exp)


def test_msm_full_size_trapdoor_and_linearity(zkp, orc):
    """BASELINE config 2 size (2^20 points): bases P_i = k_i G with known k_i, so the exact answer is
    (sum s_i k_i) G -- the same trapdoor identity as kzg/src/commitment.rs:46-51 -- plus linearity."""
    import torch
    n = 1 << 20
    ks = orc.rand_fr(0xBA5E0014, n)
    sc = orc.rand_fr(0x5EED0014, n)
    t_pts = torch.zeros(n * 12, dtype=torch.int64, device="cuda")
    zkp.g1_fixed_base_mul_dev(dev(ks), n, t_pts)
    bases = zkp.G1Bases.from_device(t_pts, n)
    # spot-check generated points against the oracle
    hp = host(t_pts, 12)
    exp_pts, _ = orc.g1_fixed_base_mul(ks[:64])
    assert np.array_equal(hp[:64], exp_pts)
    out, inf = zkp.msm_g1_dev(bases, dev(sc), n)
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc, ks))
    assert inf == einf and np.array_equal(out, exp)
    # determinism: same inputs, same bits
    out2, _ = zkp.msm_g1_dev(bases, dev(sc), n)
    assert np.array_equal(out, out2)
    # linearity: msm(9 s) == 9 msm(s)   (kzg/src/commitment.rs:61-71)
    nine = np.tile(orc.fr_from_ints([9]), (n, 1))
    out9, _ = zkp.msm_g1_dev(bases, dev(orc.fr_mul(sc, nine)), n)
    exp9, _ = orc.g1_mul(out, 0, orc.fr_from_ints([9])[0])
    assert np.array_equal(out9, exp9)
    # partial sums over two halves combine to the whole (the multi-GPU exchange unit)
    h = n // 2
    b_lo = zkp.G1Bases.from_device(t_pts[: h * 12], h)
    b_hi = zkp.G1Bases.from_device(t_pts[h * 12:].contiguous(), h)
    p0 = zkp.msm_g1_partial_dev(b_lo, dev(sc[:h]), h)
    p1 = zkp.msm_g1_partial_dev(b_hi, dev(sc[h:]), h)
    comb, cinf = zkp.g1_xyzz_sum(np.stack([p0, p1]))
    assert not cinf and np.array_equal(comb, out)


@pytest.mark.parametrize("n", [63, 64, 65, (1 << 15) - 1, 1 << 15, (1 << 15) + 1, (1 << 16) + 3, (1 << 18) - 1, 1 << 18,
                               (1 << 18) + 1])
def test_msm_sizes_around_the_geometry_thresholds(zkp, orc, n):
    """The window width (16 / 18 / 20 bits), the four-lanes-per-bucket kernels and the bucket-set dispatch order switch on the
    number of terms: sizes one below, at and one above every switch, over the expanded SRS (automatic window) and over the
    plain bases, against the trapdoor answer (sum s_i k_i) G (kzg/src/commitment.rs:46-51)."""
    import torch
    ks = orc.rand_fr(0xBA5E0500 + n % 997, n)
    sc = orc.rand_fr(0x5EED0500 + n % 997, n)
    sc[0] = 0
    sc[n - 1] = orc.fr_from_ints([M.R - 1])[0]
    t_pts = torch.zeros(n * 12, dtype=torch.int64, device="cuda")
    zkp.g1_fixed_base_mul_dev(dev(ks), n, t_pts)
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc, ks))
    bases = zkp.G1Bases.from_device(t_pts, n)
    out, inf = zkp.msm_g1_dev(bases, dev(sc), n)
    assert inf == einf and np.array_equal(out, exp)
    bases.precompute(0)
    out, inf = zkp.msm_g1_dev(bases, dev(sc), n)
    assert inf == einf and np.array_equal(out, exp)
    out, inf = zkp.msm_g1(bases, sc)  # host-pointer entry (what KzgScheme::commit binds)
    assert inf == einf and np.array_equal(out, exp)


def test_msm_host_scalars_upload_ranges(zkp, orc, monkeypatch):
    """zkp_msm_g1 with host scalars over an expanded SRS (what the Rust seam `evaluate_in_s`, kzg/src/scheme.rs:84-96, hands over)
    uploads them in two or three ranges that add into the same buckets -- short ranges first, whose kernels cover the upload of
    the rest (issued by the library's uploader thread).  Every split (default, equal halves, a tiny and a huge first range, three equal
    ranges, one range, two short ranges in front) gives the known answer (sum s_i k_i) G; n is not a multiple of anything."""
    import torch
    n = (1 << 19) + 1025 + 7
    ks = orc.rand_fr(0xBA5E0700, n)
    sc = orc.rand_fr(0x5EED0700, n)
    sc[5] = 0
    t_pts = torch.zeros(n * 12, dtype=torch.int64, device="cuda")
    zkp.g1_fixed_base_mul_dev(dev(ks), n, t_pts)
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc, ks))
    bases = zkp.G1Bases.from_device(t_pts, n)
    bases.precompute(0)
    for env in ({}, {"ZKP_MSM_FEED_FIRST_PCT": "0"}, {"ZKP_MSM_FEED_FIRST_PCT": "1"}, {"ZKP_MSM_FEED_FIRST_PCT": "90"},
                {"ZKP_MSM_FEED_RANGES": "3"}, {"ZKP_MSM_FEED_RANGES": "1"},
                {"ZKP_MSM_FEED_FIRST_PCT": "10", "ZKP_MSM_FEED_SECOND_PCT": "30"},   # the schedule of 2^21 terms and more (three unequal ranges)
                {"ZKP_MSM_FEED_FIRST_PCT": "60", "ZKP_MSM_FEED_SECOND_PCT": "80"},   # a second range that does not fit: dropped
                {"ZKP_MSM_FEED_FIRST_PCT": "2", "ZKP_MSM_FEED_SECOND_PCT": "3", "ZKP_MSM_RANGE_LOG": "17"}):  # two short ranges, then four more
        for k in ("ZKP_MSM_FEED_FIRST_PCT", "ZKP_MSM_FEED_SECOND_PCT", "ZKP_MSM_FEED_RANGES", "ZKP_MSM_RANGE_LOG"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out, inf = zkp.msm_g1(bases, sc)
        assert inf == einf and np.array_equal(out, exp), env
        out, inf = zkp.msm_g1(bases, sc[: n - 333])  # fewer scalars than bases: zip truncation (scheme.rs:88)
        exp2, einf2 = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc[: n - 333], ks[: n - 333]))
        assert inf == einf2 and np.array_equal(out, exp2), env


def test_profiling_levels_record_what_they_say(zkp, orc):
    """zkp_profile_enable: 1 records every phase of an MSM (events on the launch stream), 2 the dominant kernel only -- what
    bench.py's timed region uses, since every recorded phase boundary is a bubble on the stream -- with its clock stamps; 0 nothing."""
    import torch
    n = 1 << 14
    ks = orc.rand_fr(0xBA5E0900, n)
    sc = orc.rand_fr(0x5EED0900, n)
    t_pts = torch.zeros(n * 12, dtype=torch.int64, device="cuda")
    zkp.g1_fixed_base_mul_dev(dev(ks), n, t_pts)
    bases = zkp.G1Bases.from_device(t_pts, n)
    bases.precompute(0)
    d_sc = dev(sc)
    ref = zkp.msm_g1_dev(bases, d_sc, n)
    names = ("msm_digits", "msm_sort", "msm_accumulate", "msm_bucket_reduce", "msm_tail_host")
    try:
        for level, want in ((2, {"msm_accumulate"}), (True, set(names)), (False, set())):
            zkp.profile_reset()
            zkp.profile_enable(level)
            out = zkp.msm_g1_dev(bases, d_sc, n)
            out = zkp.msm_g1_dev(bases, d_sc, n)
            torch.cuda.synchronize()
            zkp.profile_enable(False)
            assert out[1] == ref[1] and np.array_equal(out[0], ref[0])
            got = {k for k in names if zkp.profile_read(k)[1]}
            assert got == want, (level, got)
            if want:
                ms, cnt = zkp.profile_read("msm_accumulate")
                assert cnt == 2 and 0 < ms < 50
                cyc, ticks, waves = zkp.profile_clock_read("msm_accumulate")
                assert waves > 0 and 300 < 100.0 * cyc / ticks < 3000   # MHz
    finally:
        zkp.profile_enable(False)
        zkp.profile_reset()


@pytest.mark.parametrize("knobs", [{"ZKP_MSM_RANGE_LOG": "13"}, {"ZKP_MSM_RANGE_LOG": "16", "ZKP_MSM_FIRST_PCT": "6"},
                                   {"ZKP_MSM_RANGE_LOG": "14", "ZKP_MSM_FIRST_PCT": "20", "ZKP_MSM_NO_OVERLAP": "1"}])
def test_msm_profiled_over_several_scalar_ranges(zkp, orc, knobs):
    """Profiling on (phase events + the clock stamps in msm_accumulate's epilogue) while the scalars are walked in several ranges, the
    first one short, overlapped and serialised: the combination in which round 5 found a GPU fault (profiles/r05_l_profile_mode_fault.md;
    the static half of the regression is tests/test_build_isa.py).  Every MSM must return the one-range, unprofiled result."""
    import os
    import torch
    n = (1 << 17) + 5
    ks = orc.rand_fr(0xBA5E0901, n)
    sc = orc.rand_fr(0x5EED0901, n)
    t_pts = torch.zeros(n * 12, dtype=torch.int64, device="cuda")
    zkp.g1_fixed_base_mul_dev(dev(ks), n, t_pts)
    bases = zkp.G1Bases.from_device(t_pts, n)
    bases.precompute(0)
    d_sc = dev(sc)
    ref = zkp.msm_g1_dev(bases, d_sc, n)
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc, ks))
    assert ref[1] == einf and np.array_equal(ref[0], exp)
    os.environ.update(knobs)
    try:
        zkp.profile_reset()
        zkp.profile_enable(True)
        for _ in range(12):
            out = zkp.msm_g1_dev(bases, d_sc, n)
            assert out[1] == ref[1] and np.array_equal(out[0], ref[0])
        torch.cuda.synchronize()
        ms, cnt = zkp.profile_read("msm_accumulate")
        ranges = -(-n // (1 << int(knobs["ZKP_MSM_RANGE_LOG"])))
        assert cnt >= 12 * 2 and cnt <= 12 * (ranges + 1), (cnt, ranges)   # one accumulate per range and MSM
        cyc, ticks, waves = zkp.profile_clock_read("msm_accumulate")
        assert waves > 0 and 300 < 100.0 * cyc / ticks < 3000
    finally:
        zkp.profile_enable(False)
        zkp.profile_reset()
        for k in knobs:
            del os.environ[k]


# ----------------------------------------------------------------------------- polynomial product / KZG
def test_poly_mul_golden_and_oracle(zkp, orc, golden):
    for ent in golden["poly"]:
        a = orc.fr_from_ints([hx(v) for v in ent["a"]])
        b = orc.fr_from_ints([hx(v) for v in ent["b"]])
        assert M.poly_trim(orc.fr_to_ints(zkp.poly_mul_fr(a, b))) == [hx(v) for v in ent["prod"]]
    a, b = orc.rand_fr(1, 700), orc.rand_fr(2, 1500)
    assert np.array_equal(zkp.poly_mul_fr(a, b), orc.poly_mul_fr(a, b))
    assert zkp.poly_mul_fr(a, np.zeros((0, 4), dtype=np.uint64)).shape[0] == 0


def test_kzg_open_golden(zkp, orc, golden):
    ent = golden["kzg_open"]
    xy, inf = orc.points_from_ints([pt_from_hex(p) for p in ent["points"]])
    srs = zkp.Srs(xy)
    scheme = zkp.KzgScheme(srs)
    c = orc.fr_from_ints([hx(v) for v in ent["coeffs"]])
    z = orc.fr_from_ints([hx(ent["z"])])[0]
    (w, winf), ev = scheme.open(c, z)
    assert orc.fr_to_ints(ev.reshape(1, 4)) == [hx(ent["eval"])]
    assert not winf and orc.points_to_ints(w)[0] == pt_from_hex(ent["opening"])
    cm, cinf = scheme.commit(c)
    assert orc.points_to_ints(cm)[0] == pt_from_hex(ent["commit"])
    # commit_para: one scalar-mul of g1_points[0] (scheme.rs:78-82)
    k = orc.fr_from_ints([123456789])[0]
    cp, _ = scheme.commit_para(k)
    exp, _ = orc.g1_mul(xy[0], 0, k)
    assert np.array_equal(cp, exp)
    # trailing zero coefficients are trimmed like DensePolynomial::from_coefficients_vec
    c_pad = np.concatenate([c, np.zeros((5, 4), dtype=np.uint64)])
    assert np.array_equal(scheme.commit(c_pad)[0], cm)
    with pytest.raises(zkp.ZkpError):
        scheme.open(np.zeros((0, 4), dtype=np.uint64), z)


@pytest.mark.parametrize("n,mode", [(3000, "equal"), (20000, "small"), (1 << 16, "uniform"), (70000, "two")])
def test_msm_skewed_scalars_oversized_buckets(zkp, orc, n, mode):
    """Skewed digit distributions put thousands of points into one bucket (and any n whose window width does not divide
    the scalar width has a sparse top window): those runs are cut into pieces and recombined (msm_order / msm_combine)."""
    ks = orc.rand_fr(0xBA5E1000 + n, n)
    pts, _ = orc.g1_fixed_base_mul(ks)
    if mode == "equal":
        sc = np.tile(orc.rand_fr(1, 1), (n, 1))                     # every scalar identical: one bucket per window
    elif mode == "small":
        sc = orc.fr_from_ints([(i % 7) + 1 for i in range(n)])        # 7 distinct tiny scalars
    elif mode == "two":
        a, b = orc.fr_from_ints([M.R - 1, 1])
        sc = np.where((np.arange(n) % 2 == 0)[:, None], a, b)
    else:
        sc = orc.rand_fr(0x5EED1000 + n, n)
    bases = zkp.G1Bases.from_host(pts)
    out, inf = zkp.msm_g1(bases, sc)
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc, ks))  # trapdoor: (sum s_i k_i) G
    assert inf == einf and np.array_equal(out, exp)


@pytest.mark.parametrize("log_n,world,chunks", [(12, 2, 1), (20, 4, 2), (21, 8, 4)])
def test_four_step_ntt_multi_gpu_dataflow_on_one_gpu(zkp, orc, log_n, world, chunks):
    """BASELINE config 5's NTT path: the four-step decomposition with all-to-all exchanges, `world` logical ranks on this
    one GPU (threads + in-memory exchange instead of RCCL), real HIP kernels (axis-0 column transforms with the fused twiddle,
    row transforms reading the gathered layout), against the single-GPU transform; natural-order output and the
    k1-slab layout with its mirrored inverse."""
    import torch
    from zkp_hip import dist as zd
    n = 1 << log_n
    a = orc.rand_fr(0xD157 + log_n, n)
    t_full = dev(a).reshape(n, 4)
    exp = t_full.clone().reshape(-1)
    zkp.ntt_fr_dev(exp, log_n)
    exp = exp.reshape(n, 4)
    slab = n // world
    ops = zd.TorchOps(zkp)

    def per_rank(r, exchange):
        local = t_full[r * slab:(r + 1) * slab].clone()
        return zd.ntt_fr_distributed(local, log_n, False, ops=ops, rank=r, world=world, exchange=exchange, natural_output=True,
                                     chunks=chunks)

    outs = zd.LoopbackExchange(world).run(per_rank)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(outs), exp)

    def per_rank_inv(r, exchange):
        return zd.ntt_fr_distributed(outs[r].clone(), log_n, True, ops=ops, rank=r, world=world, exchange=exchange,
                                     natural_output=True, chunks=chunks)

    back = zd.LoopbackExchange(world).run(per_rank_inv)
    assert torch.equal(torch.cat(back), t_full)

    # k1-slab layout: rank g holds [k1 - g r1][k2] = X[k1 + N1 k2]; the mirrored inverse takes it straight back
    def per_rank_k1(r, exchange):
        return zd.ntt_fr_distributed(t_full[r * slab:(r + 1) * slab], log_n, False, ops=ops, rank=r, world=world,
                                     exchange=exchange, chunks=chunks)

    mids = zd.LoopbackExchange(world).run(per_rank_k1)
    l1 = zd.four_step_split(log_n, world)
    n1, n2 = 1 << l1, 1 << (log_n - l1)
    want = exp.reshape(n2, n1, 4).permute(1, 0, 2).contiguous().reshape(n, 4)   # [k1][k2] <- X[k1 + N1 k2]
    assert torch.equal(torch.cat(mids), want)

    def per_rank_back(r, exchange):
        return zd.ntt_fr_distributed(mids[r], log_n, True, ops=ops, rank=r, world=world, exchange=exchange, chunks=chunks,
                                     input_layout="k1slab")

    back2 = zd.LoopbackExchange(world).run(per_rank_back)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(back2), t_full)

    # ONE exchange per transform: the columns layout (rank g holds columns [g r2, (g + 1) r2) of every row of the N1 x N2 matrix,
    # chunk-major) enters at the column transforms and leaves the same k1-slab layout; the mirrored inverse returns to it
    C = zd.columns_chunks(log_n, world, chunks)
    shares = [zd.columns_shard(t_full, log_n, r, world, C) for r in range(world)]
    assert torch.equal(zd.columns_gather(shares, log_n, C), t_full)

    def per_rank_cols(r, exchange):
        return zd.ntt_fr_distributed(shares[r], log_n, False, ops=ops, rank=r, world=world, exchange=exchange, chunks=C,
                                     input_layout="columns")

    mids2 = zd.LoopbackExchange(world).run(per_rank_cols)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(mids2), want)

    def per_rank_cols_back(r, exchange):
        return zd.ntt_fr_distributed(mids2[r], log_n, True, ops=ops, rank=r, world=world, exchange=exchange, chunks=C,
                                     input_layout="k1slab", output_layout="columns")

    back3 = zd.LoopbackExchange(world).run(per_rank_cols_back)
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(back3[r], shares[r])


def test_msm_batch_matches_single(zkp, orc):
    """Several commitments over the same SRS in one pass (plonk/src/prover.rs:92,150,267-268 commit in groups of 3/3/2)."""
    n = 5000
    ks = orc.rand_fr(0xBA7C, n)
    pts, _ = orc.g1_fixed_base_mul(ks)
    bases = zkp.G1Bases.from_host(pts)
    vecs = [orc.rand_fr(0x5EED2000 + i, n) for i in range(4)]
    vecs[1][100:] = 0  # a shorter polynomial padded with zeros
    vecs[2][:] = 0     # the zero polynomial: identity in the middle of the batch (the results share one field inversion)
    got = zkp.msm_g1_batch_dev(bases, [dev(v) for v in vecs], n)
    assert got[2][1] == 1
    for v, (xy, inf) in zip(vecs, got):
        exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(v, ks))
        assert inf == einf and np.array_equal(xy, exp)
        one, oinf = zkp.msm_g1(bases, v)
        assert oinf == inf and np.array_equal(one, xy)


def test_kzg_open_long_polynomial_device_path(zkp, orc):
    """open() of a 2^15-coefficient polynomial: evaluation and the division by (X - z) run on the GPU (scheme.rs:108-120);
    checked against the oracle's Horner / synthetic division and the trapdoor identity W = [(p(s) - y)/(s - z)]G."""
    n = 1 << 15
    s_int = 0x1234567
    srs = zkp.Srs.new_from_secret(orc.fr_from_ints([s_int])[0], n - 3)
    c = orc.rand_fr(0xAB, n)
    c[-3:] = 0  # trailing zeros are trimmed first
    z = orc.rand_fr(0xCD, 1)[0]
    (w, winf), ev = zkp.KzgScheme(srs).open(c, z)
    assert np.array_equal(ev, orc.poly_eval_fr(c, z))
    q = orc.poly_div_linear_fr(c[:-3], z)
    pts = srs.g1_points()
    exp, einf = orc.msm_pippenger(pts[:q.shape[0]], None, q)
    assert winf == einf and np.array_equal(w, exp)
    # z = 0 edge: quotient is the shifted coefficient vector
    zero = np.zeros(4, dtype=np.uint64)
    (w0, _), ev0 = zkp.KzgScheme(srs).open(c, zero)
    assert np.array_equal(ev0, c[0])
    exp0, _ = orc.msm_pippenger(pts[:n - 4], None, c[1:n - 3])
    assert np.array_equal(w0, exp0)


@pytest.mark.parametrize("wb", [12, 16, 20, 22, 23, 24])
def test_msm_shared_buckets_with_expanded_bases(zkp, orc, wb):
    """zkp_g1_bases_precompute: all windows of a scalar share one bucket set through pre-multiplied copies of the bases.
    Same group element as the plain path, for full and partial lengths (down to 1 and 65 scalars in 2^(wb-1) buckets),
    batches and skewed scalars."""
    n = 12000 if wb == 20 else 6000
    ks = orc.rand_fr(0xE0 + wb, n)
    pts, _ = orc.g1_fixed_base_mul(ks)
    plain = zkp.G1Bases.from_host(pts)
    expanded = zkp.G1Bases.from_host(pts).precompute(wb)
    for m, seed in ((n, 1), (n - 1234, 2), (1, 3), (65, 4)):
        sc = orc.rand_fr(0x5EED3000 + seed, m)
        if m > 100:
            sc[0] = 0
            sc[1] = orc.fr_from_ints([M.R - 1])[0]
            sc[2:40] = orc.fr_from_ints([5])[0]
        a, ainf = zkp.msm_g1(expanded, sc)
        b, binf = zkp.msm_g1(plain, sc)
        exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc, ks[:m]))
        assert ainf == binf == einf and np.array_equal(a, b) and np.array_equal(a, exp)
    vecs = [orc.rand_fr(0x5EED3100 + i, n) for i in range(3)]
    got = zkp.msm_g1_batch_dev(expanded, [dev(v) for v in vecs], n)
    for v, (xy, inf) in zip(vecs, got):
        exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(v, ks))
        assert inf == einf and np.array_equal(xy, exp)
    same = np.tile(orc.rand_fr(9, 1), (n, 1))  # every digit equal: one bucket gets all n * W entries
    a, ainf = zkp.msm_g1(expanded, same)
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(same, ks))
    assert ainf == einf and np.array_equal(a, exp)


def test_precompute_reports_an_expansion_that_does_not_fit(zkp, orc, monkeypatch):
    """An SRS too large to expand (a 2^27 SRS on one device; here a byte budget stands in for the device size) is refused with
    ZKP_E_NOMEM and the plane arithmetic in the message; the handle stays usable unexpanded."""
    n = 3000
    ks = orc.rand_fr(0xE77, n)
    pts, _ = orc.g1_fixed_base_mul(ks)
    sc = orc.rand_fr(0xE78, n)
    bases = zkp.G1Bases.from_host(pts)
    monkeypatch.setenv("ZKP_SRS_EXPAND_MAX_BYTES", str(128 * 13 * n - 1))
    with pytest.raises(zkp.ZkpError) as ei:
        bases.precompute(20)
    assert ei.value.code == zkp.ZKP_E_NOMEM
    assert "13 planes x 3000 points x 128 B = %d bytes" % (128 * 13 * n) in str(ei.value)
    assert bases.info() == (0, 0)
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc, ks))
    out, inf = zkp.msm_g1(bases, sc)
    assert inf == einf and np.array_equal(out, exp)
    monkeypatch.setenv("ZKP_SRS_EXPAND_MAX_BYTES", str(128 * 13 * n))
    bases.precompute(20)
    assert bases.info() == (20, 13)
    out, inf = zkp.msm_g1(bases, sc)
    assert inf == einf and np.array_equal(out, exp)


@pytest.mark.gpu
def test_msm_2_24_expanded_two_ranges_trapdoor(zkp, orc):
    """2^24 + 5 terms over an SRS expanded at the AUTOMATIC width (precompute(0): 12 balanced slices of 21/22 bits over 2^21 buckets,
    scalar ranges of at most 2^24 -- the geometry bench.py times from 2^22 points on): the walk is split into two scalar ranges that add
    into the same buckets; exact answer from the trapdoor identity (sum s_i k_i) G with the ORACLE's inner product and scalar mul."""
    import torch
    n = (1 << 24) + 5
    ks = orc.rand_fr(0xBA5E0018, n)
    sc = orc.rand_fr(0x5EED0018, n)
    t_pts = torch.zeros(n * 12, dtype=torch.int64, device="cuda")
    zkp.g1_fixed_base_mul_dev(dev(ks), n, t_pts)
    bases = zkp.G1Bases.from_device(t_pts, n).precompute(0)
    assert bases.info() == (22, 12)
    del t_pts
    out, inf = zkp.msm_g1_dev(bases, dev(sc), n)
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc, ks))
    assert inf == einf and np.array_equal(out, exp)
    bases.close()
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_msm_2_23_forced_20_bit_two_ranges_trapdoor(zkp, orc):
    """The forced 20-bit geometry at a multi-range size (13 planes: ranges of at most 2^23 scalars, so 2^23 + 3 terms take two), which the
    automatic width no longer selects above 2^22 points."""
    import torch
    n = (1 << 23) + 3
    ks = orc.rand_fr(0xBA5E0017, n)
    sc = orc.rand_fr(0x5EED0017, n)
    t_pts = torch.zeros(n * 12, dtype=torch.int64, device="cuda")
    zkp.g1_fixed_base_mul_dev(dev(ks), n, t_pts)
    bases = zkp.G1Bases.from_device(t_pts, n).precompute(20)
    assert bases.info() == (20, 13)
    del t_pts
    out, inf = zkp.msm_g1_dev(bases, dev(sc), n)
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc, ks))
    assert inf == einf and np.array_equal(out, exp)
    bases.close()
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_msm_host_scalars_pipelined_upload(zkp, orc):
    """zkp_msm_g1 with host scalars over expanded bases uploads the scalars range by range on a second stream while the
    earlier ranges are being accumulated (uneven last range included); same result as the trapdoor identity."""
    import torch
    n = (1 << 19) + 12345
    ks = orc.rand_fr(0xBA5E0A11, n)
    sc = orc.rand_fr(0x5EED0A11, n)
    t_pts = torch.zeros(n * 12, dtype=torch.int64, device="cuda")
    zkp.g1_fixed_base_mul_dev(dev(ks), n, t_pts)
    bases = zkp.G1Bases.from_device(t_pts, n).precompute(20)
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc, ks))
    for _ in range(2):  # second call reuses the copy stream and event
        out, inf = zkp.msm_g1(bases, sc)
        assert inf == einf and np.array_equal(out, exp)
    out, inf = zkp.msm_g1(bases, sc[: (1 << 19) - 1])  # just below the pipelining threshold: single upload
    exp2, _ = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc[: (1 << 19) - 1], ks[: (1 << 19) - 1]))
    assert np.array_equal(out, exp2)


@pytest.mark.gpu
@pytest.mark.parametrize("wb", [16, 20])
def test_expanded_planes_are_the_shifted_points(zkp, orc, wb):
    """Plane s of the expanded bases is 2^(wb s) P: a scalar vector with the single entry 2^(wb s) selects exactly that
    stored point (first and last base, every plane)."""
    n = 12000 if wb == 20 else 6000
    ks = orc.rand_fr(0xE9 + wb, n)
    pts, _ = orc.g1_fixed_base_mul(ks)
    expanded = zkp.G1Bases.from_host(pts).precompute(wb)
    for s in range(-(-256 // wb)):
        k = (1 << (wb * s)) % M.R
        if k == 0:
            continue
        for idx in (0, n - 1):
            sc = np.zeros((n, 4), dtype=np.uint64)
            sc[idx] = orc.fr_from_ints([k])[0]
            got, ginf = zkp.msm_g1(expanded, sc)
            exp, einf = orc.g1_mul(pts[idx], 0, sc[idx])
            assert ginf == einf and np.array_equal(got, exp), (s, idx)


@pytest.mark.gpu
def test_msm_shared_buckets_in_several_ranges(zkp, orc, monkeypatch):
    """Above 2^23 scalars the shared-bucket walk is split into ranges that add into the same buckets; force that split at
    2^10 so that a small case covers it (uneven last range, infinity bases, skewed scalars that create pieces)."""
    monkeypatch.setenv("ZKP_MSM_RANGE_LOG", "10")
    n = 5000
    ks = orc.rand_fr(0xE7, n)
    ks[17] = 0  # base 17 is the point at infinity
    pts, inf = orc.g1_fixed_base_mul(ks)
    expanded = zkp.G1Bases.from_host(pts, inf).precompute(16)
    for m, seed in ((n, 1), (4097, 2), (1024, 3), (1025, 4)):
        sc = orc.rand_fr(0x5EED3200 + seed, m)
        sc[2:900] = orc.fr_from_ints([5])[0]
        a, ainf = zkp.msm_g1(expanded, sc)
        exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(sc, ks[:m]))
        assert ainf == einf and np.array_equal(a, exp)
    vecs = [orc.rand_fr(0x5EED3300 + i, n) for i in range(2)]
    got = zkp.msm_g1_batch_dev(expanded, [dev(v) for v in vecs], n)
    for v, (xy, i) in zip(vecs, got):
        exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_inner_product(v, ks))
        assert i == einf and np.array_equal(xy, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("split_log", [0, 1, 2])
def test_msm_bucket_runs_split_over_lanes(zkp, orc, monkeypatch, split_log):
    """Small MSMs over narrow windows: 2^split_log lanes (or quads) share a bucket's run and msm_fold_parts adds the parts up.
    Forced here on cases that meet every path next to it: the four-lanes-per-bucket kernel (few entries) and the lane-per-bucket
    kernel (a batch), an infinity base, empty parts (runs shorter than the split), identical scalars (one oversized bucket per
    slice, cut into pieces while its other parts stay empty), expanded and plain bases.  The trapdoor gives the exact answer."""
    monkeypatch.setenv("ZKP_MSM_SPLIT_LOG", str(split_log))
    n = 6000
    ks = orc.rand_fr(0x5B17, n)
    ks[5] = 0  # base 5 is the point at infinity
    pts, inf = orc.g1_fixed_base_mul(ks)
    g = orc.g1_generator()
    for wb in (0, 12, 16):
        bases = zkp.G1Bases.from_host(pts, inf)
        if wb:
            bases.precompute(wb)
        vecs = [orc.rand_fr(0x5EED5B00 + i, n) for i in range(3)]
        vecs[1][:] = orc.rand_fr(7, 1)[0]              # every scalar identical
        vecs[2][40:] = 0                               # 40 terms: most parts empty
        for m in (n, 333, 1):
            for v in vecs[:2]:
                xy, i = zkp.msm_g1(bases, v[:m])
                exp, einf = orc.g1_mul(g, 0, orc.fr_inner_product(v[:m], ks[:m]))
                assert i == einf and np.array_equal(xy, exp)
        got = zkp.msm_g1_batch_dev(bases, [dev(v) for v in vecs], n)
        for v, (xy, i) in zip(vecs, got):
            exp, einf = orc.g1_mul(g, 0, orc.fr_inner_product(v, ks))
            assert i == einf and np.array_equal(xy, exp)
    # the lane-per-bucket kernel with split runs: more than 2^20 entries in one pass
    n2 = (1 << 16) + 3
    ks2 = orc.rand_fr(0x5B18, n2)
    pts2, _ = orc.g1_fixed_base_mul(ks2)
    b2 = zkp.G1Bases.from_host(pts2).precompute(16)
    sc2 = orc.rand_fr(0x5EED5B20, n2)
    sc2[100:5000] = orc.fr_from_ints([3])[0]
    xy, i = zkp.msm_g1(b2, sc2)
    exp, einf = orc.g1_mul(g, 0, orc.fr_inner_product(sc2, ks2))
    assert i == einf and np.array_equal(xy, exp)


def test_concurrent_callers_are_serialised_and_exact(zkp, orc):
    """include/zkp_hip.h promises that handles may be shared across threads: four host threads (ctypes drops the GIL) hammer
    the MSM, the Fr NTT and a Goldilocks NTT at once; every result must equal the single-threaded one."""
    import threading
    n = 4096
    ks = orc.rand_fr(0xBA5E0777, n)
    pts, _ = orc.g1_fixed_base_mul(ks)
    bases = zkp.G1Bases.from_host(pts)
    bases.precompute(0)
    scs = [orc.rand_fr(0x5EED0777 + i, n) for i in range(4)]
    want_msm = [zkp.msm_g1(bases, s) for s in scs]
    data = orc.rand_fr(0x0177A, 1 << 12)
    want_ntt = zkp.ntt_fr(data)
    gl = orc.rand_gl(0x61, 1 << 12)
    want_gl = zkp.ntt_goldilocks(gl)
    errors = []

    def worker(i):
        try:
            for _ in range(5):
                out, inf = zkp.msm_g1(bases, scs[i])
                assert inf == want_msm[i][1] and np.array_equal(out, want_msm[i][0])
                assert np.array_equal(zkp.ntt_fr(data), want_ntt)
                assert np.array_equal(zkp.ntt_goldilocks(gl), want_gl)
        except Exception as e:  # noqa: BLE001 - reported below, from the main thread
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


# ----------------------------------------------------------------------------- BASELINE.json configs, verbatim
@pytest.mark.gpu
def test_config0_kzg_commit_2_10_over_an_srs(zkp, orc):
    """BASELINE configs[0] as written: kzg::commit on 2^10 random Fr coefficients over an SRS [s^i]G of circuit_size + 3 =
    2^10 + 3 points (kzg/src/srs.rs:51), against the reference-faithful single-thread path (oracle.msm_naive restates
    kzg/src/scheme.rs:88-94) and against the trapdoor value [p(s)]G (kzg/src/commitment.rs:46-51)."""
    n = 1 << 10
    s = orc.rand_fr(0x5EC0000A, 1)[0]
    coeffs = orc.rand_fr(0x5EED000A, n)
    srs_xy = zkp.srs_g1(s, n + 3)                                    # Srs::new_from_secret on the GPU
    exp_srs = orc.srs(s, n + 3)
    assert np.array_equal(srs_xy, exp_srs)
    exp, einf = orc.msm_naive(srs_xy[:n], None, coeffs)              # evaluate_in_s: zip truncates to the 2^10 coefficients
    p_s = orc.poly_eval_fr(coeffs, s)
    trap, tinf = orc.g1_mul(orc.g1_generator(), 0, np.asarray(p_s).reshape(4))
    assert not einf and einf == tinf and np.array_equal(exp, trap)
    bases = zkp.G1Bases.from_host(srs_xy)
    got, ginf = zkp.kzg_commit(bases, coeffs)                        # zkp_kzg_commit, plain SRS
    assert ginf == einf and np.array_equal(got, exp)
    scheme = zkp.KzgScheme(zkp.Srs(srs_xy))                          # KzgScheme::new expands the fixed SRS once
    got2, ginf2 = scheme.commit(coeffs)
    assert ginf2 == einf and np.array_equal(got2, exp)
    got3, ginf3 = zkp.msm_g1(bases, coeffs)                          # the raw seam, same numbers
    assert ginf3 == einf and np.array_equal(got3, exp)


@pytest.mark.gpu
def test_msm_2_26_single_gpu_plain_and_expanded_trapdoor(zkp, orc):
    """BASELINE configs[4]'s size on ONE GPU: 2^26 terms over the plain bases (16 bucket sets) and over the SRS expanded at the
    automatic width (precompute(0): 12 planes of 21/22-bit slices = 103 GB, four scalar ranges of 2^24 adding into one bucket set --
    what bench.py's msm_grid times), both against the trapdoor answer (sum s_i k_i) G from the oracle."""
    import torch
    import bench
    from zkp_hip import trapdoor
    n = 1 << 26
    ks = bench.rand_fr_tensor(torch, n, 0xBA5E0000 + 26 * 64, "cuda")
    sc = bench.rand_fr_tensor(torch, n, 0x5EED0000 + 26 * 64, "cuda")
    h_ks, h_sc = host(ks, 4), host(sc, 4)
    e_ref = orc.fr_inner_product(h_sc, h_ks)                         # oracle (C, one core)
    e_dev = trapdoor.fr_inner_product(sc, ks)                        # what bench.py's bit_exact_full uses
    assert orc.fr_to_ints(e_ref.reshape(1, 4)) == [e_dev]
    del h_ks, h_sc
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, e_ref)
    t_pts = torch.zeros(n * 12, dtype=torch.int64, device="cuda")
    zkp.g1_fixed_base_mul_dev(ks, n, t_pts)
    del ks
    bases = zkp.G1Bases.from_device(t_pts, n)
    del t_pts
    torch.cuda.empty_cache()
    out, inf = zkp.msm_g1_dev(bases, sc, n)
    assert inf == einf and np.array_equal(out, exp)
    bases.precompute(0)
    assert bases.info() == (22, 12)
    out, inf = zkp.msm_g1_dev(bases, sc, n)
    assert inf == einf and np.array_equal(out, exp)
    bases.close()
    del sc
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_four_step_ntt_2_26_with_8_logical_ranks(zkp, orc):
    """BASELINE configs[4]'s NTT at full size: 2^26 elements, 8 logical ranks on this one GPU (threads + in-memory exchange in
    place of RCCL, the real kernels), against the single-GPU transform, then the inverse back to the input."""
    import torch
    import bench
    from zkp_hip import dist as zd
    log_n, world = 26, 8
    n = 1 << log_n
    t_full = bench.rand_fr_tensor(torch, n, 0x0177001A, "cuda")
    exp = t_full.clone().reshape(-1)
    zkp.ntt_fr_dev(exp, log_n)
    exp = exp.reshape(n, 4)
    slab = n // world
    ops = zd.TorchOps(zkp)

    def per_rank(r, exchange):
        return zd.ntt_fr_distributed(t_full[r * slab:(r + 1) * slab], log_n, False, ops=ops, rank=r, world=world,
                                     exchange=exchange, natural_output=True)

    outs = zd.LoopbackExchange(world).run(per_rank)
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(outs[r], exp[r * slab:(r + 1) * slab])
    del exp

    del outs

    def per_rank_k1(r, exchange):
        return zd.ntt_fr_distributed(t_full[r * slab:(r + 1) * slab], log_n, False, ops=ops, rank=r, world=world,
                                     exchange=exchange, chunks=4)

    mids = zd.LoopbackExchange(world).run(per_rank_k1)

    def per_rank_back(r, exchange):
        return zd.ntt_fr_distributed(mids[r], log_n, True, ops=ops, rank=r, world=world, exchange=exchange, chunks=4,
                                     input_layout="k1slab")

    back = zd.LoopbackExchange(world).run(per_rank_back)
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(back[r], t_full[r * slab:(r + 1) * slab])
    del back

    # the one-exchange form at the same size: columns layout in -> the same k1-slab layout -> columns layout back
    shares = [zd.columns_shard(t_full, log_n, r, world, 4) for r in range(world)]

    def per_rank_cols(r, exchange):
        return zd.ntt_fr_distributed(shares[r], log_n, False, ops=ops, rank=r, world=world, exchange=exchange, chunks=4,
                                     input_layout="columns")

    mids2 = zd.LoopbackExchange(world).run(per_rank_cols)
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(mids2[r], mids[r])
    del mids

    def per_rank_cols_back(r, exchange):
        return zd.ntt_fr_distributed(mids2[r], log_n, True, ops=ops, rank=r, world=world, exchange=exchange, chunks=4,
                                     input_layout="k1slab", output_layout="columns")

    back2 = zd.LoopbackExchange(world).run(per_rank_cols_back)
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(back2[r], shares[r])


@pytest.mark.gpu
def test_dev_entries_on_two_streams_share_workspaces_safely(zkp, orc):
    """`*_dev` entries return without synchronising and all of them share the slot's scratch buffer and coset tables: two
    callers on different (non-blocking) streams must still get exact results -- the second stream waits on the device for
    what the first one enqueued (cross-stream reuse of ntt_scratch / coset cache)."""
    import torch
    log_n = 21
    n = 1 << log_n
    a, b = orc.rand_fr(0x57A, n), orc.rand_fr(0x57B, n)
    cosets = [orc.fr_from_ints([7 + k])[0] for k in range(10)]  # more cosets than cache ways: tables are evicted and rebuilt
    exp = []
    for k, c in enumerate(cosets):
        t = dev(a if k % 2 == 0 else b).reshape(-1)
        zkp.ntt_fr_dev(t, log_n, coset=c)
        exp.append(t.clone())
    torch.cuda.synchronize()
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = []
    for k, c in enumerate(cosets):
        with torch.cuda.stream(s[k % 2]):
            t = dev(a if k % 2 == 0 else b).reshape(-1)
            zkp.ntt_fr_dev(t, log_n, coset=c)   # launched on the current (side) stream, returns at once
            got.append(t)
    torch.cuda.synchronize()
    for k in range(len(cosets)):
        assert torch.equal(got[k], exp[k]), k
    # and back, alternating streams the other way round
    for k, c in enumerate(cosets):
        with torch.cuda.stream(s[(k + 1) % 2]):
            got[k].record_stream(s[(k + 1) % 2])
            s[(k + 1) % 2].wait_stream(s[k % 2])
            zkp.ntt_fr_dev(got[k], log_n, inverse=True, coset=c)
    torch.cuda.synchronize()
    for k in range(len(cosets)):
        assert torch.equal(got[k].cpu(), dev(a if k % 2 == 0 else b).reshape(-1).cpu()), k


@pytest.mark.parametrize("form", [0, 1])
def test_device_safegcd_inverse_on_edge_inputs(zkp, orc, form):
    """The device division-step inversion itself (csrc/fq28_inv.hpp, behind Srs::new_from_secret and the SRS expansion = the
    `into_affine` of kzg/src/scheme.rs:92-93), driven lane by lane through zkp_selftest_fq_inverse_dev on the inputs the Python model
    of tests/test_safegcd_model.py walks: 0, 1, p - 1, the non-canonical p, p + 5, 2p - 1, powers of two +- 1, p >> k, Fibonacci-ratio
    values (the longest division-step chains) and random residues -- a whole wave of slow inputs next to a wave of fast ones, so that
    the wave-uniform early exit is taken at different rounds.  Checked against big integers and against the oracle's Fq product."""
    import random
    import torch
    P = M.P
    rbits = 384 if form == 0 else 392
    rnd = random.Random(0x5AFE + form)
    xs = [0, 1, 2, 3, P - 1, P - 2, (P + 1) // 2, (P - 1) // 2, 1 << 380, (1 << 381) - 1, P, P + 1, P + 5, 2 * P - 1, 2 * P - 2]
    a, b = 1, 1
    while b < P:
        a, b = b, a + b
        xs.append(b % P)
    for k in range(1, 381, 7):
        xs += [(1 << k) - 1, (1 << k) + 1, P >> k, (P >> k) | 1, P - (1 << k)]
    phi = (P * 0x9E3779B97F4A7C15) >> 64
    xs += [phi, phi + 1, P - phi]
    xs += [rnd.randrange(1, P) for _ in range(700)] + [rnd.randrange(P, 2 * P) for _ in range(100)]
    xs += [1] * 64 + [rnd.randrange(1, 1 << 40) for _ in range(64)]       # waves that finish early
    n = len(xs)
    if form == 0:
        words = np.array([[(x >> (32 * w)) & 0xffffffff for w in range(12)] for x in xs], dtype=np.uint32)
    else:
        words = np.array([[(x >> (28 * w)) & 0xfffffff for w in range(14)] + [0, 0] for x in xs], dtype=np.uint32)
    d_in = torch.from_numpy(words.view(np.int32).reshape(-1)).cuda()
    d_out = torch.zeros_like(d_in)
    zkp.selftest_fq_inverse_dev(d_in, n, form, d_out)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy().view(np.uint32).reshape(n, -1)
    r2 = pow(2, 2 * rbits, P)
    got = []
    for i, x in enumerate(xs):
        if form == 0:
            v = sum(int(out[i, w]) << (32 * w) for w in range(12))
            assert v < P, (i, hex(x))                                      # canonical
        else:
            assert all(int(out[i, w]) < (1 << 28) for w in range(13)) and out[i, 14] == 0 and out[i, 15] == 0
            v = sum(int(out[i, w]) << (28 * w) for w in range(14))
            assert v < 2 * P, (i, hex(x))                                  # tight
        got.append(v)
        exp = pow(x, -1, P) * r2 % P if x % P else 0                       # (x / R)^-1 R = x^-1 R^2
        assert v % P == exp, (i, hex(x))
    if form == 0:  # and with the oracle's own Montgomery product: x * inverse(x) == R  (the residue of 1)
        lim = lambda vals: np.array([[(v >> (64 * w)) & 0xffffffffffffffff for w in range(6)] for v in vals], dtype=np.uint64)
        prod = orc.fq_mul(lim([x % P for x in xs]), lim(got))
        one = pow(2, 384, P)
        for i, x in enumerate(xs):
            pv = sum(int(prod[i, w]) << (64 * w) for w in range(6))
            assert pv == (one if x % P else 0), (i, hex(x))


@pytest.mark.parametrize("with_event", [False, True])
def test_sharded_dev_entry_orders_after_a_side_stream_producer_single_slot(zkp, orc, with_event):
    """zkp_msm_g1_sharded_dev on a single-slot handle: the scalars are produced on a non-blocking side stream behind a
    long spin and handed over WITHOUT any host synchronisation.  The entry must wait -- for the whole device (no event) or, with
    zkp_msm_g1_sharded_dev_after, for the producer's event on the device -- before its digits kernel reads them."""
    import torch
    n = (1 << 16) + 3
    pts, _ = orc.g1_fixed_base_mul(orc.rand_fr(0xE7E, 600))
    pts = np.tile(pts, (n // 600 + 1, 1))[:n]
    sc = orc.rand_fr(0xE7F, n)
    bases = zkp.G1Bases.from_host(pts)
    exp, einf = zkp.msm_g1(bases, sc)
    assert not einf
    src = dev(sc)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(side):
            torch.cuda._sleep(200_000_000)           # ~0.1 s of spinning in front of the producer
            resident = (src ^ 0).contiguous()        # fresh allocation, filled only when the spin is over
            ev = torch.cuda.Event()
            ev.record(side)
        got, ginf = zkp.msm_g1_sharded_dev(bases, [resident], n, events=[ev] if with_event else None)
        assert not ginf and np.array_equal(got, exp)
        del resident
    bases.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log_len,cols,inverse,tw", [(5, 8, False, 12), (8, 4, True, 11), (9, 16, False, 14), (13, 8, False, 17),
                                                     (13, 4, True, 0), (16, 4, True, 19)])
def test_ntt_axis0_kernel_vs_specification(zkp, orc, log_len, cols, inverse, tw):
    """zkp_ntt_fr_axis0_dev (one and two strided passes, natural-order rows, fused four-step twiddle, 1/len for the inverse)
    against the CPU statement of its specification (tests/oracle_ops.py: oracle NTT per column + big-int twiddles)."""
    import torch
    from oracle_ops import OracleOps
    L = 1 << log_len
    a = orc.rand_fr(0xA810 + log_len + cols, L * cols)
    col0 = 3 if tw else 0
    src = torch.from_numpy(a.view(np.int64).copy())
    want = torch.empty_like(src)
    OracleOps(orc).axis0(src, want, log_len, cols, inverse, tw, col0)
    d_in = dev(a).reshape(-1)
    d_out = torch.zeros_like(d_in)
    zkp.ntt_fr_axis0_dev(d_in, d_out, log_len, cols, inverse=inverse, tw_log_n=tw, tw_col0=col0)
    assert torch.equal(d_out.cpu(), want.reshape(-1))
    assert torch.equal(d_in.cpu(), src.reshape(-1))            # out of place: the input is untouched
    if log_len <= 8:                                           # single pass: in place is allowed
        zkp.ntt_fr_axis0_dev(d_in, d_in, log_len, cols, inverse=inverse, tw_log_n=tw, tw_col0=col0)
        assert torch.equal(d_in.cpu(), want.reshape(-1))


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,batch,inverse,mode", [(6, 4, False, "gather"), (10, 4, True, "scatter"), (13, 8, False, "gather"),
                                                      (13, 8, True, "scatter"), (12, 2, False, "plain")])
def test_ntt_layout_kernel_vs_specification(zkp, orc, log_n, batch, inverse, mode):
    """zkp_ntt_fr_layout_dev: gathered input ([g][q][row][c] blocks as the second all-to-all leaves them), scattered and
    twiddled output (the send blocks of the mirrored inverse), and the plain contiguous case, against tests/oracle_ops.py."""
    import torch
    from oracle_ops import OracleOps
    n = 1 << log_n
    G, C = 4, 2
    cw = n // (G * C)
    layout = (cw.bit_length() - 1, 1, G * batch * cw, batch * cw, cw)   # lo = c_lo, mid = q (2 chunks), hi = g
    a = orc.rand_fr(0x1A70 + log_n, n * batch)
    src = torch.from_numpy(a.view(np.int64).copy())
    want = torch.zeros_like(src)
    kw = {"gather": dict(in_layout=layout), "scatter": dict(out_layout=layout, tw_log_n=log_n + 4, tw_row0=5), "plain": {}}[mode]
    OracleOps(orc).layout(src, want, log_n, batch, inverse, **kw)
    d_in = dev(a).reshape(-1)
    d_out = torch.zeros_like(d_in)
    mk = lambda l: zkp.NttLayout(*l)
    zkp.ntt_fr_layout_dev(d_in, d_out, log_n, batch, inverse=inverse,
                          in_layout=mk(kw["in_layout"]) if "in_layout" in kw else None,
                          out_layout=mk(kw["out_layout"]) if "out_layout" in kw else None,
                          tw_log_n=kw.get("tw_log_n", 0), tw_row0=kw.get("tw_row0", 0))
    assert torch.equal(d_out.cpu(), want.reshape(-1))
    with pytest.raises(zkp.ZkpError):  # a gathered transform cannot run in place
        zkp.ntt_fr_layout_dev(d_in, d_in, log_n, batch, in_layout=mk(layout))
