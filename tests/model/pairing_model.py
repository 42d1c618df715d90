"""Independent big-int model of the BLS12-381 optimal ate pairing (TEST INFRASTRUCTURE ONLY), as the reference uses it
through ark-ec's `Bls12_381::pairing` for verification only (kzg/src/scheme.rs:143-160,215-245, plonk/src/verifier.rs:130-157).

Tower: Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3 - (1+u)), Fq12 = Fq6[w]/(w^2 - v).  G2 on the M-twist y^2 = x^3 + 4(1+u).
The pairing value is the CANONICAL reduced ate pairing  f_{|x|,Q}(P)^(+-)((p^12 - 1)/r)  with the plain (unoptimised) final
exponent, lines scaled by powers of w (killed by the final exponentiation).  The reference only ever compares two pairing
values, so any fixed non-degenerate bilinear normalisation gives the same accept/reject decisions.
"""
import bigmodel as M

P = M.P
R = M.R
X_ABS = 0xD201000000010000  # |x|, x = -X_ABS is the BLS12-381 curve parameter
FINAL_EXP = (P ** 12 - 1) // R

G2_X = (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E)
G2_Y = (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)
G2 = (G2_X, G2_Y)


# ---- Fq2
def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_scale(a, k): return (a[0] * k % P, a[1] * k % P)
def f2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return (a[0] * n % P, (-a[1]) * n % P)
def f2_mul_xi(a): return ((a[0] - a[1]) % P, (a[0] + a[1]) % P)  # * (1 + u)
F2_ZERO, F2_ONE = (0, 0), (1, 0)


# ---- Fq6 = triples of Fq2
def f6_add(a, b): return tuple(f2_add(x, y) for x, y in zip(a, b))
def f6_sub(a, b): return tuple(f2_sub(x, y) for x, y in zip(a, b))
def f6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    c0 = f2_add(f2_mul(a0, b0), f2_mul_xi(f2_add(f2_mul(a1, b2), f2_mul(a2, b1))))
    c1 = f2_add(f2_add(f2_mul(a0, b1), f2_mul(a1, b0)), f2_mul_xi(f2_mul(a2, b2)))
    c2 = f2_add(f2_add(f2_mul(a0, b2), f2_mul(a1, b1)), f2_mul(a2, b0))
    return (c0, c1, c2)
def f6_mul_v(a): return (f2_mul_xi(a[2]), a[0], a[1])  # * v
F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)


# ---- Fq12 = pairs of Fq6
def f12_mul(a, b):
    a0, a1 = a
    b0, b1 = b
    return (f6_add(f6_mul(a0, b0), f6_mul_v(f6_mul(a1, b1))), f6_add(f6_mul(a0, b1), f6_mul(a1, b0)))
def f12_conj(a): return (a[0], tuple(f2_neg(x) for x in a[1]))
def f12_pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_mul(r, r)
        if bit == "1":
            r = f12_mul(r, a)
    return r
F12_ONE = (F6_ONE, F6_ZERO)


# ---- G2 (affine, None = infinity)
def g2_double(t):
    if t is None or t[1] == F2_ZERO:
        return None
    lam = f2_mul(f2_scale(f2_mul(t[0], t[0]), 3), f2_inv(f2_scale(t[1], 2)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), t[0]), t[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(t[0], x3)), t[1]))
def g2_add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        return g2_double(a) if a[1] == b[1] else None
    lam = f2_mul(f2_sub(b[1], a[1]), f2_inv(f2_sub(b[0], a[0])))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), a[0]), b[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(a[0], x3)), a[1]))
def g2_neg(a): return None if a is None else (a[0], f2_neg(a[1]))
def g2_mul(a, k):
    r = None
    for bit in bin(k % R)[2:] if k % R else "":
        r = g2_double(r)
        if bit == "1":
            r = g2_add(r, a)
    return r
def g2_on_curve(a):
    return a is None or f2_mul(a[1], a[1]) == f2_add(f2_mul(f2_mul(a[0], a[0]), a[0]), (4, 4))


def _line(lam, t, p):
    """Line through T with slope lam (both on the twist) at P in G1, scaled by w^3:
    (lam xT - yT) + (-lam xP) v + (yP) v w   ->   Fq12 = ((A, B, 0), (0, C, 0))."""
    a = f2_sub(f2_mul(lam, t[0]), t[1])
    b = f2_scale(lam, (-p[0]) % P)
    return ((a, b, F2_ZERO), (F2_ZERO, (p[1], 0), F2_ZERO))


def miller_loop(p, q):
    """f_{|x|,Q}(P), conjugated because x < 0.  p: G1 affine (ints) or None, q: G2 affine or None."""
    if p is None or q is None:
        return F12_ONE
    f, t = F12_ONE, q
    for bit in bin(X_ABS)[3:]:
        lam = f2_mul(f2_scale(f2_mul(t[0], t[0]), 3), f2_inv(f2_scale(t[1], 2)))
        f = f12_mul(f12_mul(f, f), _line(lam, t, p))
        t = g2_double(t)
        if bit == "1":
            lam = f2_mul(f2_sub(q[1], t[1]), f2_inv(f2_sub(q[0], t[0])))
            f = f12_mul(f, _line(lam, t, p))
            t = g2_add(t, q)
    return f12_conj(f)


def pairing(p, q):
    return f12_pow(miller_loop(p, q), FINAL_EXP)


def f12_flat(a):
    """12 Fq coefficients in memory order: c0.c0.c0, c0.c0.c1, c0.c1.c0, ... (arkworks Fp12 = [Fp6; 2], Fp6 = [Fp2; 3])."""
    return [c for half in a for f2 in half for c in f2]
