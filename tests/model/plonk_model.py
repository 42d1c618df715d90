"""Line-by-line big-int restatement of the reference's PLONK prover polynomial algebra (TEST INFRASTRUCTURE ONLY).

Follows plonk/src/circuit.rs (compile, cal_permutation), plonk/src/prover.rs (generate_proof, compute_acc,
compute_quotient_polynomial, compute_linearisation_polynomial) and plonk/src/slice_polynomial.rs with the reference's own
algorithms (O(n^2) Horner accumulator, coefficient-form products, division by the vanishing polynomial), so it is only
usable for small circuits.  Blinders and challenges are parameters (the reference draws them from an entropy-seeded RNG
and a SHA-256 transcript).  Commitments use the trapdoor identity commit(p) = [p(s)]G, which the reference's own test
asserts (kzg/src/commitment.rs:46-51).
"""
import bigmodel as M

R = M.R


def padd(a, b):
    n = max(len(a), len(b))
    return M.poly_trim([((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % R for i in range(n)])


def psub(a, b):
    n = max(len(a), len(b))
    return M.poly_trim([((a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0)) % R for i in range(n)])


def pscale(a, s):
    return M.poly_trim([x * s % R for x in a])


def pmul(a, b):
    return M.poly_mul(a, b, R)


def peval(a, x):
    return M.poly_eval(a, x, R)


def interpolate(evals, n):
    """Evaluations::from_vec_and_domain(v, domain).interpolate(): zero-pad to the domain, iFFT, trim."""
    v = list(evals) + [0] * (n - len(evals))
    return M.poly_trim(M.ntt(v, R, inverse=True))


class Circuit:
    """plonk/src/circuit.rs: gates with (wire column, index, value) triples."""

    def __init__(self):
        self.gates = []  # (a_pos, b_pos, c_pos, q_l, q_r, q_o, q_m, q_c, pi) ; pos None = dummy
        self.vals = [[], [], []]

    def _add(self, a, b, c, sel, pi):
        self.gates.append(((a[0], a[1]), (b[0], b[1]), (c[0], c[1])) + sel + ((-pi) % R,))
        self.vals[0].append(a[2] % R)
        self.vals[1].append(b[2] % R)
        self.vals[2].append(c[2] % R)

    def add_addition_gate(self, a, b, c, pi=0):       # gate.rs:38-55
        self._add(a, b, c, (1, 1, R - 1, 0, 0), pi)

    def add_multiplication_gate(self, a, b, c, pi=0):  # gate.rs:57-74
        self._add(a, b, c, (0, 0, R - 1, 1, 0), pi)

    def add_constant_gate(self, a, b, c, pi=0, constant=None):  # gate.rs:76-94; circuit.rs:76-79: the constant is a's value
        self._add(a, b, c, (1, 0, 0, 0, (-(a[2] if constant is None else constant)) % R), pi)

    def compile(self):
        """circuit.rs:166-245 -> dict of coefficient lists + n, k1, k2."""
        ln = len(self.gates)
        n = 1 << ((ln - 1).bit_length()) if ln > 1 else 1  # pad_circuit: (len-1).ilog2() + 1
        if ln == 1:
            n = 1
        w = M.root_of_unity(n.bit_length() - 1)
        roots = [pow(w, i, R) for i in range(n)]
        k1 = (roots[0] + 1) % R
        k2 = (k1 + 1) % R
        cos = [roots, [r * k1 % R for r in roots], [r * k2 % R for r in roots]]
        # get_assignment skips dummy gates: vectors are the real gates only, interpolate zero-pads
        cols = {"f_a": self.vals[0], "f_b": self.vals[1], "f_c": self.vals[2],
                "q_l": [g[3] for g in self.gates], "q_r": [g[4] for g in self.gates], "q_o": [g[5] for g in self.gates],
                "q_m": [g[6] for g in self.gates], "q_c": [g[7] for g in self.gates], "pi": [g[8] for g in self.gates]}
        out = {k: interpolate(v, n) for k, v in cols.items()}
        sig = [list(cos[0]), list(cos[1]), list(cos[2])]
        for i, g in enumerate(self.gates):
            for col in range(3):
                pc, pi_ = g[col]
                sig[col][i] = cos[pc][pi_]
        out["s_sigma_1"], out["s_sigma_2"], out["s_sigma_3"] = (interpolate(s, n) for s in sig)
        out.update(n=n, k1=k1, k2=k2)
        return out


def mul_by_vanishing(c, n):
    return M.mul_by_vanishing(c, n, R)


def divide_by_vanishing(c, n):
    q, rem = M.divide_by_vanishing(c, n, R)
    assert rem == [], "No remainder expected"  # prover.rs:404,431,441
    return q


def compute_acc(cc, beta, gamma):
    """prover.rs:302-377, O(n^2) Horner as in the reference."""
    n, k1, k2 = cc["n"], cc["k1"], cc["k2"]
    w = M.root_of_unity(n.bit_length() - 1)
    acc_e, pre = [1], 1
    for i in range(1, n):
        x = pow(w, i - 1, R)
        a, b, c = peval(cc["f_a"], x), peval(cc["f_b"], x), peval(cc["f_c"], x)
        num = (a + beta * x + gamma) * (b + beta * k1 * x + gamma) * (c + beta * k2 * x + gamma) % R
        den = ((a + beta * peval(cc["s_sigma_1"], x) + gamma) * (b + beta * peval(cc["s_sigma_2"], x) + gamma) *
               (c + beta * peval(cc["s_sigma_3"], x) + gamma)) % R
        pre = pre * num * pow(den, -1, R) % R
        acc_e.append(pre)
    shifted = acc_e[1:] + acc_e[:1]
    return interpolate(acc_e, n), interpolate(shifted, n)


def prove(cc, secret, blinders, ch):
    """generate_proof (prover.rs:61-293).  blinders = [b1..b9]; ch = dict beta gamma alpha zeta v.
    Returns every polynomial, the bar values and the commitments' discrete logs p(s)."""
    n, k1, k2 = cc["n"], cc["k1"], cc["k2"]
    b1, b2, b3, b4, b5, b6, b7, b8, b9 = blinders
    w = M.root_of_unity(n.bit_length() - 1)
    out = {}
    ax = padd(cc["f_a"], mul_by_vanishing([b2, b1], n))
    bx = padd(cc["f_b"], mul_by_vanishing([b4, b3], n))
    cx = padd(cc["f_c"], mul_by_vanishing([b6, b5], n))
    beta, gamma, alpha, zeta, v = ch["beta"], ch["gamma"], ch["alpha"], ch["zeta"], ch["v"]
    acc, acc_w = compute_acc(cc, beta, gamma)
    z = padd(mul_by_vanishing([b9, b8, b7], n), acc)
    zw = padd(mul_by_vanishing([b9, b8 * w % R, b7 * w * w % R], n), acc_w)
    assert peval(z, w * beta % R) == peval(zw, beta)  # prover.rs:127
    # --- quotient (prover.rs:381-444)
    line1 = padd(padd(padd(padd(padd(pmul(pmul(ax, bx), cc["q_m"]), pmul(ax, cc["q_l"])), pmul(bx, cc["q_r"])),
                           pmul(cx, cc["q_o"])), cc["pi"]), cc["q_c"])
    q1 = divide_by_vanishing(line1, n)
    line2 = pscale(pmul(pmul(pmul(padd(ax, [gamma, beta]), padd(bx, [gamma, beta * k1 % R])), padd(cx, [gamma, beta * k2 % R])), z), alpha)
    line3 = pscale(pmul(pmul(pmul(padd(padd(ax, pscale(cc["s_sigma_1"], beta)), [gamma]),
                                  padd(padd(bx, pscale(cc["s_sigma_2"], beta)), [gamma])),
                             padd(padd(cx, pscale(cc["s_sigma_3"], beta)), [gamma])), zw), alpha)
    q23 = divide_by_vanishing(psub(line2, line3), n)
    l1 = interpolate([1] + [0] * (n - 1), n)
    line4 = pscale(pmul(psub(z, [1]), l1), alpha * alpha % R)
    q4 = divide_by_vanishing(line4, n)
    t = padd(padd(q1, q23), q4)
    slices, degree = M.slice_poly(t)
    # --- round 4
    bar = {"a": peval(ax, zeta), "b": peval(bx, zeta), "c": peval(cx, zeta), "s1": peval(cc["s_sigma_1"], zeta),
           "s2": peval(cc["s_sigma_2"], zeta), "zw": peval(z, zeta * w % R)}
    pi_e = peval(cc["pi"], zeta)
    txc = M.slice_compact(slices, degree, zeta)
    # --- linearisation (prover.rs:469-568)
    ln1 = padd(padd(padd(padd(pscale(cc["q_m"], bar["a"] * bar["b"] % R), pscale(cc["q_l"], bar["a"])), pscale(cc["q_r"], bar["b"])),
                    pscale(cc["q_o"], bar["c"])), cc["q_c"])
    ln1 = padd(ln1, [pi_e])
    c2 = (bar["a"] + beta * zeta + gamma) * (bar["b"] + beta * k1 * zeta + gamma) * (bar["c"] + beta * k2 * zeta + gamma) * alpha % R
    ln2 = pscale(z, c2)
    c3 = (bar["a"] + beta * bar["s1"] + gamma) * (bar["b"] + beta * bar["s2"] + gamma) * bar["zw"] * alpha % R
    ln3 = pscale(padd(pscale(cc["s_sigma_3"], beta), [(bar["c"] + gamma) % R]), c3)
    ln4 = pscale(psub(z, [1]), peval(l1, zeta) * alpha * alpha % R)
    ln5 = pscale(txc, (pow(zeta, n, R) - 1) % R)
    r = psub(padd(psub(padd(ln1, ln2), ln3), ln4), ln5)
    bar_r = peval(r, zeta)
    wev = padd(padd(padd(padd(padd(psub(r, [bar_r]), pscale(psub(ax, [bar["a"]]), v)), pscale(psub(bx, [bar["b"]]), v * v % R)),
                         pscale(psub(cx, [bar["c"]]), pow(v, 3, R))), pscale(psub(cc["s_sigma_1"], [bar["s1"]]), pow(v, 4, R))),
               pscale(psub(cc["s_sigma_2"], [bar["s2"]]), pow(v, 5, R)))
    assert peval(wev, zeta) == 0  # "w_ev_x was computed incorrectly" check, prover.rs:232-241
    w_zeta = M.poly_div_linear(wev, zeta, R)
    wev2 = psub(z, [bar["zw"]])
    assert peval(wev2, zeta * w % R) == 0
    w_zeta_omega = M.poly_div_linear(wev2, zeta * w % R, R)
    polys = {"ax": ax, "bx": bx, "cx": cx, "z": z, "t": t, "r": r, "w_zeta": w_zeta, "w_zeta_omega": w_zeta_omega,
             "tx_compact": txc, "t_lo": slices[0], "t_mid": slices[1], "t_hi": slices[2]}
    out["polys"] = polys
    out["bars"] = [bar["a"], bar["b"], bar["c"], bar["s1"], bar["s2"], bar["zw"]]
    out["degree"] = degree
    out["commit_dlog"] = {k: peval(p, secret) for k, p in polys.items()}
    return out


def reference_test_circuit():
    """plonk/src/verifier.rs:232-258: x^2 + y^2 = z^2 with (3, 4, 5)."""
    c = Circuit()
    c.add_multiplication_gate((1, 0, 3), (0, 0, 3), (0, 3, 9))
    c.add_multiplication_gate((1, 1, 4), (0, 1, 4), (1, 3, 16))
    c.add_multiplication_gate((1, 2, 5), (0, 2, 5), (2, 3, 25))
    c.add_addition_gate((2, 0, 9), (2, 1, 16), (2, 2, 25))
    return c


def reference_test_circuit_02():
    """plonk/src/verifier.rs:306-357: xy + 3x^2 + xyz = 11 -- five mul/add gates, a constant gate, padded 7 -> 8."""
    c = Circuit()
    c.add_multiplication_gate((0, 1, 1), (1, 0, 2), (0, 3, 2))
    c.add_multiplication_gate((1, 1, 1), (0, 0, 1), (0, 2, 1))
    c.add_multiplication_gate((2, 1, 1), (2, 6, 3), (1, 3, 3))
    c.add_addition_gate((0, 4, 2), (2, 2, 3), (0, 5, 5))
    c.add_multiplication_gate((2, 0, 2), (1, 4, 3), (1, 5, 6))
    c.add_addition_gate((2, 3, 5), (2, 4, 6), (2, 5, 11))
    c.add_constant_gate((0, 6, 3), (1, 6, 0), (1, 2, 3))
    return c


def reference_test_circuit_03():
    """plonk/src/verifier.rs:361-383: xyz = 6 -- two gates, n = 2 (the 8n quotient domain at its smallest)."""
    c = Circuit()
    c.add_multiplication_gate((0, 0, 1), (1, 0, 2), (0, 1, 2))
    c.add_multiplication_gate((2, 0, 2), (1, 1, 3), (2, 1, 6))
    return c


def public_input_circuit():
    """Non-zero public inputs on a multiplication, an addition and a constant gate (gate.rs:38-111 stores `pi` negated:
    q_m ab + q_l a + q_r b + q_o c + q_c - pi = 0), chained by copy constraints; 5 gates -> n = 8.
      g0 mul: 3 * 4 - 7 - 5 = 0      g1 add: 7 + 10 - 15 - 2 = 0      g2 const: 15 - 11 - 4 = 0 (constant 11 != a: not reachable
      through Circuit::add_constant_gate, which takes the constant from a's value, but a legal Gate::new_constant_gate)
      g3 mul: 15 * 2 - 30 - 0 = 0    g4 const as the reference builds it: 30 - 30 = 0, pi = 0"""
    c = Circuit()
    c.add_multiplication_gate((0, 0, 3), (1, 0, 4), (0, 1, 7), pi=5)
    c.add_addition_gate((2, 0, 7), (1, 1, 10), (0, 2, 15), pi=2)
    c.add_constant_gate((0, 3, 15), (1, 2, 0), (2, 2, 0), pi=4, constant=11)
    c.add_multiplication_gate((2, 1, 15), (1, 3, 2), (0, 4, 30))
    c.add_constant_gate((2, 3, 30), (1, 4, 0), (2, 4, 0))
    return c
