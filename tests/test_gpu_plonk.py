"""GPU: the PLONK prover rounds (zkp_plonk_round1..5) against the big-int restatement of the reference prover on small
circuits (coefficient-exact polynomials, bit-exact commitments), and at BASELINE config 4 size (2^16 gates) through the
verifier's equations with the SRS trapdoor plus polynomial identities at random points."""
import numpy as np
import pytest

import bigmodel as M
import plonk_model as PM
from test_plonk_model import challenges, verifier_identity

pytestmark = pytest.mark.gpu
R = M.R


@pytest.fixture(scope="module")
def zkp():
    import torch
    assert torch.cuda.is_available()
    import zkp_hip
    zkp_hip.init()
    return zkp_hip


def run_gpu_prover(zkp, orc, cc, secret, blinders, ch):
    n = cc["n"]
    log_n = n.bit_length() - 1
    srs = zkp.Srs.new_from_secret(orc.fr_from_ints([secret])[0], n)  # n + 3 points (srs.rs:51)
    polys = {k: orc.fr_from_ints(cc[k]) if len(cc[k]) else np.zeros((0, 4), dtype=np.uint64) for k in zkp.CIRCUIT_POLYS}
    pr = zkp.PlonkProver(srs.bases, log_n, polys, orc.fr_from_ints([cc["k1"]])[0], orc.fr_from_ints([cc["k2"]])[0])
    f = lambda v: orc.fr_from_ints([v])[0]
    res = {}
    res["abc"] = pr.round1(orc.fr_from_ints(blinders[:6]))
    res["z"] = pr.round2(f(ch["beta"]), f(ch["gamma"]), orc.fr_from_ints(blinders[6:9]))
    res["t"], res["degree"] = pr.round3(f(ch["alpha"]))
    res["polys"] = {"t": pr.get_poly("t")}
    res["bars"] = pr.round4(f(ch["zeta"]))
    res["w"] = pr.round5(f(ch["v"]))
    for name in ("ax", "bx", "cx", "z", "r", "w_zeta", "w_zeta_omega", "tx_compact"):
        res["polys"][name] = pr.get_poly(name)
    return res, srs


def check_commit(orc, got, dlog):
    exp, einf = orc.g1_mul(orc.g1_generator(), 0, orc.fr_from_ints([dlog])[0])
    assert got[1] == einf and (einf or np.array_equal(got[0], exp))


SMALL_CIRCUITS = {"ref": PM.reference_test_circuit,       # plonk/src/verifier.rs:232-258 (verifier_accepted_test_01)
                  "ref02": PM.reference_test_circuit_02,  # :306-357: mul / add / constant gates, 7 padded to n = 8
                  "ref03": PM.reference_test_circuit_03,  # :361-383: n = 2, quotient on the 8n domain, t one coefficient short
                  "pi": PM.public_input_circuit}          # non-zero pi on a mul, an add and a constant gate (gate.rs:38-111)


@pytest.mark.parametrize("case,seed", [("ref", 1), ("ref", 2), ("chain5", 3), ("chain37", 4), ("ref02", 5), ("ref03", 6), ("pi", 7),
                                       ("pi", 8)])
def test_plonk_rounds_match_reference_model(zkp, orc, case, seed):
    if case in SMALL_CIRCUITS:
        cc = SMALL_CIRCUITS[case]().compile()
        if case in ("ref02", "pi"):
            assert any(cc["q_c"])
        if case == "pi":
            assert any(cc["pi"])
    else:
        m = int(case[5:])
        c = PM.Circuit()
        a = 3
        for i in range(m):
            b = 7 + 3 * i
            out = a * b % R if i % 2 == 0 else (a + b) % R
            a_pos = (2, i - 1) if i else (0, 0)
            c_pos = (0, i + 1) if i < m - 1 else (2, i)
            (c.add_multiplication_gate if i % 2 == 0 else c.add_addition_gate)((a_pos[0], a_pos[1], a), (1, i, b), (c_pos[0], c_pos[1], out))
            a = out
        cc = c.compile()
    blinders, ch = challenges(seed)
    secret = M.rand_fr_list(200 + seed, 1)[0]
    ref = PM.prove(cc, secret, blinders, ch)
    got, _ = run_gpu_prover(zkp, orc, cc, secret, blinders, ch)
    for name in ("ax", "bx", "cx", "z", "t", "r", "w_zeta", "w_zeta_omega", "tx_compact"):
        assert M.poly_trim(orc.fr_to_ints(got["polys"][name])) == ref["polys"][name], name
    assert orc.fr_to_ints(got["bars"]) == ref["bars"]
    assert got["degree"] == ref["degree"]
    d = ref["commit_dlog"]
    for g, k in zip(got["abc"], ("ax", "bx", "cx")):
        check_commit(orc, g, d[k])
    check_commit(orc, got["z"], d["z"])
    for g, k in zip(got["t"], ("t_lo", "t_mid", "t_hi")):
        check_commit(orc, g, d[k])
    for g, k in zip(got["w"], ("w_zeta", "w_zeta_omega")):
        check_commit(orc, g, d[k])


@pytest.mark.parametrize("what", ["gate_row", "copy_constraint"])
def test_plonk_unsatisfied_circuit_is_rejected(zkp, orc, what):
    if what == "gate_row":
        c = PM.Circuit()
        c.add_multiplication_gate((1, 0, 3), (0, 0, 3), (0, 3, 9))
        c.add_multiplication_gate((1, 1, 4), (0, 1, 4), (1, 3, 16))
        c.add_multiplication_gate((1, 2, 5), (0, 2, 5), (2, 3, 25))
        c.add_addition_gate((2, 0, 9), (2, 1, 16), (2, 2, 20))  # verifier.rs:296
        msg = "gate row"          # the reference panics with "No remainder" at prover.rs:404
    else:
        from test_plonk_model import broken_copy_constraint_circuit
        c = broken_copy_constraint_circuit()   # every row holds, the wiring does not: "No remainder" at prover.rs:431
        msg = "copy constraints"
    cc = c.compile()
    blinders, ch = challenges(5)
    with pytest.raises(zkp.ZkpError) as ei:
        run_gpu_prover(zkp, orc, cc, 4242, blinders, ch)
    assert "No remainder expected" in str(ei.value) and msg in str(ei.value), str(ei.value)


def synthetic_circuit(orc, zkp, log_n, seed):
    """2^log_n gates: a chain mul / add / mul / constant, the output of gate i wired to the left input of gate i+1, right
    inputs free, non-zero public inputs on gates of all three kinds (gate.rs:38-111: q_m ab + q_l a + q_r b + q_o c + q_c - pi = 0;
    a constant gate has q_l = 1, q_o = 0, q_c = -constant and passes its input on).  Built directly as the 12 evaluation vectors
    of Circuit::compile (circuit.rs:166-245), then interpolated on the GPU (the 12 iFFTs of circuit.rs:173-176, 230-232)."""
    n = 1 << log_n
    rb = M.rand_fr_list(seed, n)
    a_v, b_v, c_v = [0] * n, rb, [0] * n
    q_m, q_l, q_r, q_o, q_c, pi_v = ([0] * n for _ in range(6))
    a = 5
    for i in range(n):
        kind = i % 4
        pi = 7 * i + 1 if i % 8 in (0, 1, 3) else 0
        a_v[i] = a
        if kind == 3:      # constant gate: a - constant - pi = 0
            q_l[i], q_c[i], c_v[i] = 1, (pi - a) % R, a
        elif kind == 1:    # addition gate: a + b - c - pi = 0
            q_l[i], q_r[i], q_o[i], c_v[i] = 1, 1, R - 1, (a + rb[i] - pi) % R
        else:              # multiplication gate: ab - c - pi = 0
            q_m[i], q_o[i], c_v[i] = 1, R - 1, (a * rb[i] - pi) % R
        pi_v[i] = (-pi) % R
        a = c_v[i]
    w = M.root_of_unity(log_n)
    roots = [1] * n
    for i in range(1, n):
        roots[i] = roots[i - 1] * w % R
    k1, k2 = 2, 3
    s1 = [(roots[i - 1] * k2) % R if i else roots[0] for i in range(n)]       # a_i <- c_{i-1}
    s2 = [roots[i] * k1 % R for i in range(n)]                               # b_i free
    s3 = [roots[i + 1] if i < n - 1 else roots[i] * k2 % R for i in range(n)]  # c_i <- a_{i+1}
    cols = {"f_a": a_v, "f_b": b_v, "f_c": c_v, "q_m": q_m, "q_l": q_l, "q_r": q_r, "q_o": q_o, "q_c": q_c, "pi": pi_v,
            "s_sigma_1": s1, "s_sigma_2": s2, "s_sigma_3": s3}
    polys = {k: zkp.ntt_fr(orc.fr_from_ints(v), inverse=True) for k, v in cols.items()}
    return polys, k1, k2


@pytest.mark.parametrize("expand", [0, -1, 18])
def test_plonk_full_size_2_16(zkp, orc, expand):
    """BASELINE config 4: 2^16-gate synthetic circuit, MSM + NTT combined, KZG opens; checked with the verifier's
    equations in the exponent (known SRS secret) and with polynomial identities at a random point.  expand = -1: over the
    SRS expanded with the library's automatic width (16 slices of 16 bits, bucket runs split over lanes), the configuration
    bench.py times; 18: 15 slices of 17/18 bits (round 1's geometry, no split); 0: the plain SRS."""
    log_n = 16
    n = 1 << log_n
    polys, k1, k2 = synthetic_circuit(orc, zkp, log_n, 0xC16C)
    secret = M.rand_fr_list(0x5EC, 1)[0]
    blinders, ch = challenges(0x16)
    f = lambda v: orc.fr_from_ints([v])[0]
    srs = zkp.Srs.new_from_secret(f(secret), n)
    if expand:
        srs.bases.precompute(max(expand, 0))
    pr = zkp.PlonkProver(srs.bases, log_n, polys, f(k1), f(k2))
    abc = pr.round1(orc.fr_from_ints(blinders[:6]))
    zc = pr.round2(f(ch["beta"]), f(ch["gamma"]), orc.fr_from_ints(blinders[6:9]))
    tc, degree = pr.round3(f(ch["alpha"]))
    t = pr.get_poly("t")
    bars = pr.round4(f(ch["zeta"]))
    wc = pr.round5(f(ch["v"]))
    assert t.shape[0] == 3 * n + 6 and degree == n + 1  # slice_polynomial.rs:22-43 on 3n+6 coefficients
    got = {name: pr.get_poly(name) for name in ("ax", "bx", "cx", "z", "r", "w_zeta", "w_zeta_omega")}
    ev = lambda arr, x: orc.fr_to_ints(orc.poly_eval_fr(arr, f(x)).reshape(1, 4))[0]
    # (1) every commitment is [p(s)]G
    for c, name in zip(abc, ("ax", "bx", "cx")):
        check_commit(orc, c, ev(got[name], secret))
    check_commit(orc, zc, ev(got["z"], secret))
    chunk = degree + 1
    dl = {}
    for i, (c, name) in enumerate(zip(tc, ("t_lo", "t_mid", "t_hi"))):
        dl[name] = ev(t[i * chunk:(i + 1) * chunk], secret)
        check_commit(orc, c, dl[name])
    for c, name in zip(wc, ("w_zeta", "w_zeta_omega")):
        check_commit(orc, c, ev(got[name], secret))
    # (2) quotient identity t(x) Z_H(x) = gate + alpha (perm) + alpha^2 (z - 1) L1 at a random x (prover.rs:381-444)
    x = M.rand_fr_list(0xABC, 1)[0]
    w = M.root_of_unity(log_n)
    beta, gamma, alpha, zeta, v = ch["beta"], ch["gamma"], ch["alpha"], ch["zeta"], ch["v"]
    e = {k: ev(p, x) for k, p in polys.items()}
    a, b, c, z = ev(got["ax"], x), ev(got["bx"], x), ev(got["cx"], x), ev(got["z"], x)
    zw = ev(got["z"], x * w % R)
    zh = (pow(x, n, R) - 1) % R
    l1 = zh * pow(n * (x - 1), -1, R) % R
    rhs = (a * b * e["q_m"] + a * e["q_l"] + b * e["q_r"] + c * e["q_o"] + e["pi"] + e["q_c"]
           + alpha * ((a + beta * x + gamma) * (b + beta * k1 * x + gamma) * (c + beta * k2 * x + gamma) * z
                      - (a + beta * e["s_sigma_1"] + gamma) * (b + beta * e["s_sigma_2"] + gamma) * (c + beta * e["s_sigma_3"] + gamma) * zw)
           + alpha * alpha * (z - 1) * l1) % R
    assert ev(t, x) * zh % R == rhs
    # (3) z(1) = 1 and the bars are the evaluations (prover.rs:164-178)
    assert ev(got["z"], 1) == 1
    ib = orc.fr_to_ints(bars)
    assert ib == [ev(got["ax"], zeta), ev(got["bx"], zeta), ev(got["cx"], zeta), ev(polys["s_sigma_1"], zeta),
                  ev(polys["s_sigma_2"], zeta), ev(got["z"], zeta * w % R)]
    # (4) the verifier's batched opening equation (verifier.rs:66-145) in the exponent
    cc = {k: orc.fr_to_ints(p) for k, p in polys.items()}
    cc.update(n=n, k1=k1, k2=k2)
    out = {"bars": ib, "degree": degree, "secret": secret,
           "commit_dlog": {"ax": ev(got["ax"], secret), "bx": ev(got["bx"], secret), "cx": ev(got["cx"], secret),
                           "z": ev(got["z"], secret), "w_zeta": ev(got["w_zeta"], secret),
                           "w_zeta_omega": ev(got["w_zeta_omega"], secret), **dl}}
    assert verifier_identity(cc, out, ch, u=M.rand_fr_list(0xD, 1)[0])


def _py_transcript_feed(data, pt):
    """plonk/src/challenge.rs:36-45 with ark-bls12-381's serialize_uncompressed (x || y, big-endian; bit 6 = infinity)."""
    import hashlib
    raw = (bytes([0x40]) + bytes(95)) if pt is None else pt[0].to_bytes(48, "big") + pt[1].to_bytes(48, "big")
    return hashlib.sha256(data + raw).digest()


def _py_challenges(data, n):
    rng = M.StdRng(int.from_bytes(data[:8], "little"))
    rinv = pow(2 ** 256, -1, R)
    return [rng.rand_field(R, 4) * rinv % R for _ in range(n)]  # sampled limbs are the Montgomery value


@pytest.mark.parametrize("seed", [1, 2])
def test_plonk_prove_end_to_end_with_reference_transcript(zkp, orc, seed):
    """zkp_plonk_prove = generate_proof (prover.rs:61-293): the challenges are re-derived here from the returned
    commitments with an independent transcript (hashlib + the Python StdRng model), and the big-int prover model run with
    those challenges must reproduce every commitment, evaluation and u."""
    cc = PM.reference_test_circuit().compile()
    blinders, _ = challenges(seed)
    secret = M.rand_fr_list(300 + seed, 1)[0]
    n = cc["n"]
    srs = zkp.Srs.new_from_secret(orc.fr_from_ints([secret])[0], n)
    polys = {k: orc.fr_from_ints(cc[k]) if len(cc[k]) else np.zeros((0, 4), dtype=np.uint64) for k in zkp.CIRCUIT_POLYS}
    pr = zkp.PlonkProver(srs.bases, n.bit_length() - 1, polys, orc.fr_from_ints([cc["k1"]])[0], orc.fr_from_ints([cc["k2"]])[0])
    proof = pr.prove(orc.fr_from_ints(blinders))

    def pt(name):
        xy, inf = proof["commits"][name]
        return None if inf else orc.points_to_ints(xy.reshape(1, 12))[0]

    data = b""
    for k in ("a", "b", "c"):
        data = _py_transcript_feed(data, pt(k))
    beta, gamma = _py_challenges(data, 2)
    data = _py_transcript_feed(data, pt("z"))
    alpha, = _py_challenges(data, 1)
    for k in ("t_lo", "t_mid", "t_hi"):
        data = _py_transcript_feed(data, pt(k))
    zeta, = _py_challenges(data, 1)
    bars = orc.fr_to_ints(proof["bars"])
    for b in bars:
        data = _py_transcript_feed(data, M.g1_mul(M.G1, b) if b else None)  # commit_para(bar) = bar * g1_points[0]
    v, = _py_challenges(data, 1)
    for k in ("w_ev_x", "w_ev_wx"):
        data = _py_transcript_feed(data, pt(k))
    u, = _py_challenges(data, 1)
    assert orc.fr_to_ints(proof["u"].reshape(1, 4)) == [u]

    ref = PM.prove(cc, secret, blinders, {"beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta, "v": v})
    assert bars == ref["bars"] and proof["degree"] == ref["degree"]
    d = ref["commit_dlog"]
    for name, key in (("a", "ax"), ("b", "bx"), ("c", "cx"), ("z", "z"), ("t_lo", "t_lo"), ("t_mid", "t_mid"), ("t_hi", "t_hi"),
                      ("w_ev_x", "w_zeta"), ("w_ev_wx", "w_zeta_omega")):
        check_commit(orc, proof["commits"][name], d[key])
    # a second proof from the same prover object (transcript state must not leak between calls)
    again = pr.prove(orc.fr_from_ints(blinders))
    assert np.array_equal(again["u"], proof["u"])
    pr.close()


def _make_prover(zkp, orc, cc, srs):
    n = cc["n"]
    polys = {k: orc.fr_from_ints(cc[k]) if len(cc[k]) else np.zeros((0, 4), dtype=np.uint64) for k in zkp.CIRCUIT_POLYS}
    return zkp.PlonkProver(srs.bases, n.bit_length() - 1, polys, orc.fr_from_ints([cc["k1"]])[0], orc.fr_from_ints([cc["k2"]])[0])


@pytest.mark.parametrize("case", ["ref", "ref02", "ref03", "pi"])
def test_plonk_prove_then_verify_with_pairings(zkp, orc, case):
    """generate_proof -> verify (plonk/src/verifier.rs:19-157) on the reference's own accepted circuits (tests at :232-258,
    :306-357 with a constant gate, :361-383 with n = 2) and on one with public inputs: accepted with the SRS's [s]_2, rejected
    after tampering with the proof, and rejected by a verifier whose circuit differs in q_c or in the public input."""
    import copy
    import pairing_model as PairM
    from test_pairing_cpu import g2_from_ints
    cc = SMALL_CIRCUITS[case]().compile()
    blinders, _ = challenges(7)
    secret = M.rand_fr_list(407, 1)[0]
    n = cc["n"]
    srs = zkp.Srs.new_from_secret(orc.fr_from_ints([secret])[0], n)
    g2s = g2_from_ints(PairM.g2_mul(PairM.G2, secret))
    pr = _make_prover(zkp, orc, cc, srs)
    proof = pr.prove(orc.fr_from_ints(blinders))
    assert pr.verify(g2s, proof) == 1
    assert pr.verify(g2_from_ints(PairM.g2_mul(PairM.G2, secret + 1)), proof) == 0  # another SRS: "Pairing failed, rejected"
    bad = copy.deepcopy(proof)
    bad["degree"] += 1                                                              # not part of the transcript
    assert pr.verify(g2s, bad) == 0
    bad = copy.deepcopy(proof)
    bad["bars"][0][0] ^= np.uint64(1)                                               # changes v, hence u
    assert pr.verify(g2s, bad) == -1
    bad = copy.deepcopy(proof)
    bad["commits"]["z"] = proof["commits"]["a"]
    assert pr.verify(g2s, bad) == -1
    for key in ("q_c", "pi"):
        # the transcript does not absorb the circuit (challenge.rs:22-77), so u still matches: the pairing check must fail --
        # through [q_c]_1 in d_line1 (verifier.rs:66-70) resp. pi(zeta) in r_0 (:47-58)
        cc2 = dict(cc)
        cc2[key] = M.poly_trim([((cc[key][0] if cc[key] else 0) + 1) % R] + list(cc[key][1:]))
        other = _make_prover(zkp, orc, cc2, srs)
        assert other.verify(g2s, proof) == 0, key
        with pytest.raises(zkp.ZkpError):   # and its own proof does not exist: "No remainder expected" (prover.rs:404)
            other.prove(orc.fr_from_ints(blinders))
        other.close()
    pr.close()
