"""Big-int / fixed-width model of zkp-implementation_amd/csrc/fq28_inv.hpp (Bernstein-Yang division steps, 30-bit signed limbs,
37 rounds of 30 steps): the same arithmetic with Python integers, every 32- and 64-bit quantity range-checked, so that the CPU tests
can run the algorithm on edge cases and on inputs chosen to need many steps.  Test infrastructure, not product code."""
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
M30 = (1 << 30) - 1
NL = 13
ROUNDS = 37
MOD = [(P >> (30 * i)) & M30 for i in range(NL)]
MOD_INV30 = pow(P, -1, 1 << 30)


def _s32(x):
    x &= 0xffffffff
    return x - (1 << 32) if x >> 31 else x


def _check64(x):
    assert -(1 << 63) <= x < (1 << 63), "64-bit accumulator overflow"
    return x


def to_limbs(x):
    """non-negative x < 2^390 -> 13 limbs of 30 bits"""
    return [(x >> (30 * i)) & M30 for i in range(NL)]


def value(l):
    return sum(v << (30 * i) for i, v in enumerate(l))


def divsteps_30(eta, f0, g0):
    """uint32 arithmetic of the device loop; returns (eta, u, v, q, r) as signed 32-bit values"""
    u, v, q, r, f, g = 1, 0, 0, 1, f0 & 0xffffffff, g0 & 0xffffffff
    for _ in range(30):
        c1 = 0xffffffff if eta < 0 else 0
        c2 = (0 - (g & 1)) & 0xffffffff
        x, y, z = ((f ^ c1) - c1) & 0xffffffff, ((u ^ c1) - c1) & 0xffffffff, ((v ^ c1) - c1) & 0xffffffff
        g = (g + (x & c2)) & 0xffffffff
        q = (q + (y & c2)) & 0xffffffff
        r = (r + (z & c2)) & 0xffffffff
        sw = c1 & c2
        eta = _s32(((eta & 0xffffffff) ^ sw) - (sw + 1))
        f = (f + (g & sw)) & 0xffffffff
        u = (u + (q & sw)) & 0xffffffff
        v = (v + (r & sw)) & 0xffffffff
        g >>= 1
        u = (u << 1) & 0xffffffff
        v = (v << 1) & 0xffffffff
    return eta, _s32(u), _s32(v), _s32(q), _s32(r)


def update_fg(f, g, u, v, q, r):
    cf = _check64(u * f[0] + v * g[0])
    cg = _check64(q * f[0] + r * g[0])
    assert cf & M30 == 0 and cg & M30 == 0
    cf >>= 30
    cg >>= 30
    for i in range(1, NL):
        cf = _check64(cf + u * f[i] + v * g[i])
        cg = _check64(cg + q * f[i] + r * g[i])
        f[i - 1], g[i - 1] = cf & M30, cg & M30
        cf >>= 30
        cg >>= 30
    f[NL - 1], g[NL - 1] = _s32(cf), _s32(cg)
    assert f[NL - 1] == cf and g[NL - 1] == cg, "top limb does not fit 32 bits"


def update_de(d, e, u, v, q, r):
    sd, se = (-1 if d[NL - 1] < 0 else 0), (-1 if e[NL - 1] < 0 else 0)
    md, me = (u & sd) + (v & se), (q & sd) + (r & se)
    cd = _check64(u * d[0] + v * e[0])
    ce = _check64(q * d[0] + r * e[0])
    md -= (MOD_INV30 * (cd & 0xffffffff) + md) & M30
    me -= (MOD_INV30 * (ce & 0xffffffff) + me) & M30
    cd = _check64(cd + MOD[0] * md)
    ce = _check64(ce + MOD[0] * me)
    assert cd & M30 == 0 and ce & M30 == 0
    cd >>= 30
    ce >>= 30
    for i in range(1, NL):
        cd = _check64(cd + u * d[i] + v * e[i] + MOD[i] * md)
        ce = _check64(ce + q * d[i] + r * e[i] + MOD[i] * me)
        d[i - 1], e[i - 1] = cd & M30, ce & M30
        cd >>= 30
        ce >>= 30
    d[NL - 1], e[NL - 1] = _s32(cd), _s32(ce)
    assert d[NL - 1] == cd and e[NL - 1] == ce
    assert -2 * P < value(d) < P and -2 * P < value(e) < P, "d, e leave (-2p, p)"


def normalize(r, negate):
    v = value(r)
    if v < 0:
        v += P
    if negate:
        v = -v
    if v < 0:
        v += P
    assert 0 <= v < P
    return v


def modinv(x, early_exit=True):
    """-> (x^-1 mod p, rounds used); x < 2p (0 and p -> 0).  With early_exit the loop stops like a wave whose lanes have all reached
    g = 0."""
    assert 0 <= x < 2 * P
    gl = to_limbs(x)                      # s30_reduce_once: limb-wise g - p with borrows, kept when non-negative
    t, c = [0] * NL, 0
    for i in range(NL):
        t[i] = gl[i] - MOD[i] + c
        if i < NL - 1:
            c = t[i] >> 30
            t[i] &= M30
    assert -(1 << 31) <= t[NL - 1] < (1 << 31)
    if t[NL - 1] >= 0:
        gl = t
    assert value(gl) == x % P if x < 2 * P else True
    f, g = list(MOD), gl
    d, e = [0] * NL, [1] + [0] * (NL - 1)
    eta, used = -1, ROUNDS
    for it in range(ROUNDS):
        eta, u, v, q, r = divsteps_30(eta, f[0], g[0])
        assert abs(u) + abs(v) <= 1 << 30 and abs(q) + abs(r) <= 1 << 30
        update_de(d, e, u, v, q, r)
        update_fg(f, g, u, v, q, r)
        if value(g) == 0 and used == ROUNDS:
            used = it + 1
            if early_exit:
                break
    assert value(g) == 0, "g != 0 after the fixed number of rounds"
    fv = value(f)
    assert abs(fv) in (1, P), "f must end at +-gcd"
    return normalize(d, fv < 0), used
