"""CPU: the big-int restatement of the reference's PLONK prover (tests/model/plonk_model.py) reproduces the behaviour the
reference's own tests pin: the x^2+y^2=z^2 circuit proves (plonk/src/verifier.rs:232-258), the tampered one panics with
"No remainder" (verifier.rs:270-305), and the proof satisfies the verifier's equations with the SRS trapdoor."""
import pytest

import bigmodel as M
import plonk_model as PM

R = M.R


def challenges(seed):
    vals = M.rand_fr_list(seed, 14)
    return vals[:9], dict(beta=vals[9], gamma=vals[10], alpha=vals[11], zeta=vals[12], v=vals[13])


def verifier_identity(cc, out, ch, u):
    """plonk/src/verifier.rs:66-145 with commitments replaced by their discrete logs (known SRS secret)."""
    n, k1, k2 = cc["n"], cc["k1"], cc["k2"]
    beta, gamma, alpha, zeta, v = ch["beta"], ch["gamma"], ch["alpha"], ch["zeta"], ch["v"]
    ba, bb, bc, bs1, bs2, bzw = out["bars"]
    d = out["commit_dlog"]
    s = out["secret"]
    w = M.root_of_unity(n.bit_length() - 1)
    zh = (pow(zeta, n, R) - 1) % R
    l1 = zh * pow(n * (zeta - 1), -1, R) % R
    pi_e = M.poly_eval(cc["pi"], zeta, R)
    ev = lambda name: M.poly_eval(cc[name], s, R)
    r0 = (pi_e - l1 * alpha * alpha - alpha * (ba + beta * bs1 + gamma) * (bb + beta * bs2 + gamma) * (bc + gamma) * bzw) % R
    D = (ba * bb * ev("q_m") + ba * ev("q_l") + bb * ev("q_r") + bc * ev("q_o") + ev("q_c")
         + d["z"] * ((ba + beta * zeta + gamma) * (bb + beta * k1 * zeta + gamma) * (bc + beta * k2 * zeta + gamma) * alpha
                     + l1 * alpha * alpha + u)
         - (ba + beta * bs1 + gamma) * (bb + beta * bs2 + gamma) * alpha * beta * bzw * ev("s_sigma_3")
         - zh * (d["t_lo"] + pow(zeta, out["degree"] + 1, R) * d["t_mid"] + pow(zeta, 2 * (out["degree"] + 1), R) * d["t_hi"])) % R
    F = (D + v * d["ax"] + v ** 2 * d["bx"] + v ** 3 * d["cx"] + v ** 4 * ev("s_sigma_1") + v ** 5 * ev("s_sigma_2")) % R
    E = (-r0 + v * ba + v ** 2 * bb + v ** 3 * bc + v ** 4 * bs1 + v ** 5 * bs2 + u * bzw) % R
    lhs = (d["w_zeta"] + u * d["w_zeta_omega"]) * s % R
    rhs = (zeta * d["w_zeta"] + u * zeta * w * d["w_zeta_omega"] + F - E) % R
    return lhs == rhs


def test_reference_circuit_compiles_and_satisfies_gates():
    cc = PM.reference_test_circuit().compile()
    assert cc["n"] == 4 and cc["k1"] == 2 and cc["k2"] == 3  # circuit.rs:238-245
    w = M.root_of_unity(2)
    for i in range(4):
        x = pow(w, i, R)
        e = lambda k: M.poly_eval(cc[k], x, R)
        assert (e("f_a") * e("f_b") * e("q_m") + e("f_a") * e("q_l") + e("f_b") * e("q_r") + e("f_c") * e("q_o") + e("pi") + e("q_c")) % R == 0


@pytest.mark.parametrize("seed", [1, 2])
def test_reference_circuit_proves_and_verifies(seed):
    cc = PM.reference_test_circuit().compile()
    blinders, ch = challenges(seed)
    secret = M.rand_fr_list(100 + seed, 1)[0]
    out = PM.prove(cc, secret, blinders, ch)
    out["secret"] = secret
    n = cc["n"]
    assert len(out["polys"]["ax"]) == n + 2 and len(out["polys"]["z"]) == n + 3 and len(out["polys"]["t"]) == 3 * n + 6
    assert out["degree"] == n + 1
    assert verifier_identity(cc, out, ch, u=M.rand_fr_list(7, 1)[0])


def test_tampered_circuit_has_remainder():
    c = PM.Circuit()
    c.add_multiplication_gate((1, 0, 3), (0, 0, 3), (0, 3, 9))
    c.add_multiplication_gate((1, 1, 4), (0, 1, 4), (1, 3, 16))
    c.add_multiplication_gate((1, 2, 5), (0, 2, 5), (2, 3, 25))
    c.add_addition_gate((2, 0, 9), (2, 1, 16), (2, 2, 20))  # verifier.rs:296: 25 -> 20
    cc = c.compile()
    blinders, ch = challenges(3)
    with pytest.raises(AssertionError):
        PM.prove(cc, 12345, blinders, ch)


def broken_copy_constraint_circuit():
    """Every gate row holds on its own (3*3 = 9, 4*4 = 16, 5*5 = 25, 10 + 16 = 26), but a_3 = 10 is wired to c_0 = 9 and c_3 = 26 to
    c_2 = 25: the permutation argument fails -- the second "No remainder expected" of the reference (prover.rs:431)."""
    c = PM.Circuit()
    c.add_multiplication_gate((1, 0, 3), (0, 0, 3), (0, 3, 9))
    c.add_multiplication_gate((1, 1, 4), (0, 1, 4), (1, 3, 16))
    c.add_multiplication_gate((1, 2, 5), (0, 2, 5), (2, 3, 25))
    c.add_addition_gate((2, 0, 10), (2, 1, 16), (2, 2, 26))
    return c


def test_broken_copy_constraint_has_remainder():
    cc = broken_copy_constraint_circuit().compile()
    w = M.root_of_unity(2)
    for i in range(4):  # the gate equation holds on every row: line 1 of the quotient divides
        e = {k: PM.peval(cc[k], pow(w, i, R)) for k in ("f_a", "f_b", "f_c", "q_m", "q_l", "q_r", "q_o", "q_c", "pi")}
        assert (e["f_a"] * e["f_b"] * e["q_m"] + e["f_a"] * e["q_l"] + e["f_b"] * e["q_r"] + e["f_c"] * e["q_o"] + e["pi"] + e["q_c"]) % R == 0
    blinders, ch = challenges(4)
    with pytest.raises(AssertionError):
        PM.prove(cc, 12345, blinders, ch)


def test_larger_circuit_with_padding():
    # 5 gates -> padded to 8; chain of mul/add gates with the output wired to the next gate's left input
    c = PM.Circuit()
    a, vals = 3, []
    for i in range(5):
        b = 7 + i
        out = a * b % R if i % 2 == 0 else (a + b) % R
        a_pos = (2, i - 1) if i else (0, 0)
        c_pos = (0, i + 1) if i < 4 else (2, i)
        (c.add_multiplication_gate if i % 2 == 0 else c.add_addition_gate)((a_pos[0], a_pos[1], a), (1, i, b), (c_pos[0], c_pos[1], out))
        a = out
    cc = c.compile()
    assert cc["n"] == 8
    blinders, ch = challenges(4)
    out = PM.prove(cc, 777, blinders, ch)
    out["secret"] = 777
    assert verifier_identity(cc, out, ch, u=99)


@pytest.mark.parametrize("name,n", [("reference_test_circuit_02", 8), ("reference_test_circuit_03", 2), ("public_input_circuit", 8)])
def test_more_reference_circuits_prove_and_verify(name, n):
    """plonk/src/verifier.rs:306-357 (constant gate, 7 gates padded to 8), :361-383 (n = 2), and non-zero public inputs on the
    three gate types (gate.rs:38-111): gate equations hold on the domain, the model proves and the verifier's equation holds."""
    cc = getattr(PM, name)().compile()
    assert cc["n"] == n
    w = M.root_of_unity(n.bit_length() - 1)
    for i in range(n):
        x = pow(w, i, R)
        e = lambda k: M.poly_eval(cc[k], x, R)
        assert (e("f_a") * e("f_b") * e("q_m") + e("f_a") * e("q_l") + e("f_b") * e("q_r") + e("f_c") * e("q_o") + e("pi") + e("q_c")) % R == 0
    if name != "reference_test_circuit_03":
        assert any(cc["q_c"])
    if name == "public_input_circuit":
        assert any(cc["pi"])
    blinders, ch = challenges(11)
    secret = M.rand_fr_list(111, 1)[0]
    out = PM.prove(cc, secret, blinders, ch)
    out["secret"] = secret
    # n = 2: w^(n+2) = 1, so the leading terms b1 b3 b5 b7 of the two permutation products cancel and t is one coefficient short
    assert len(out["polys"]["t"]) == (3 * n + 6 if n > 2 else 11) and out["degree"] == n + 1
    assert verifier_identity(cc, out, ch, u=M.rand_fr_list(8, 1)[0])
    # a tampered q_c / pi no longer divides (prover.rs:404) -- or, for the verifier, breaks the identity
    bad = dict(cc)
    bad["q_c"] = M.poly_trim([(cc["q_c"][0] if cc["q_c"] else 0) + 1] + list(cc["q_c"][1:]))
    with pytest.raises(AssertionError):
        PM.prove(bad, secret, blinders, ch)
    assert not verifier_identity(bad, out, ch, u=5)
    bad = dict(cc)
    bad["pi"] = M.poly_trim([(cc["pi"][0] if cc["pi"] else 0) + 1] + list(cc["pi"][1:]))
    assert not verifier_identity(bad, out, ch, u=5)
