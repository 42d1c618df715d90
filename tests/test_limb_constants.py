"""The hard-coded limb constants of the device field headers, re-derived from the big-int model (CPU only)."""
import os
import re

import bigmodel as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zkp-implementation_amd", "csrc")


def arrays(path, struct):
    src = open(os.path.join(CSRC, path)).read()
    body = src[src.index("struct " + struct):]
    body = body[:body.index("\n};")]
    out = {}
    for name, vals in re.findall(r"(\w+)\[\d+\]\s*=\s*\{([^}]*)\}", body):
        out[name] = [int(v.strip().rstrip("u"), 16) for v in vals.split(",")]
    for name, val in re.findall(r"uint32_t (\w+) = (0x[0-9a-f]+|\d+)u?;", body):
        out[name] = int(val, 0)
    return out


def limbs(x, bits, n):
    m = (1 << bits) - 1
    return [(x >> (bits * i)) & m if i < n - 1 else x >> (bits * i) for i in range(n)]


def value(l, bits):
    return sum(x << (bits * i) for i, x in enumerate(l))


def test_saturated_params():
    for struct, p, n in (("FrParams", M.R, 8), ("FqParams", M.P, 12)):
        c = arrays("ff.hpp", struct)
        assert c["MOD"] == limbs(p, 32, n)
        assert c["ONE"] == limbs((1 << (32 * n)) % p, 32, n)
        assert c["R2"] == limbs(pow(1 << (32 * n), 2, p), 32, n)
        assert c["INV"] == (-pow(p, -1, 1 << 32)) % (1 << 32)


def test_fq28_constants():
    c = arrays("fq28.hpp", "Fq28C")
    assert c["MOD"] == limbs(M.P, 28, 14)
    assert c["INV"] == (-pow(M.P, -1, 1 << 28)) % (1 << 28)
    assert c["ONE"] == limbs((1 << 392) % M.P, 28, 14)
    tight_top = (2 * M.P) >> 364
    for name, k, j in (("KP4_29", 4, 29), ("KP8_29", 8, 29), ("KP16_29", 16, 29), ("KP8_30", 8, 30)):
        l = c[name]
        assert value(l, 28) == k * M.P                      # it IS k*p
        assert all(x < (1 << 32) for x in l)
        assert all(x >= (1 << j) - 4 for x in l[:13])       # dominates every limb < 2^j - 4
    # top limbs dominate the documented subtrahend bounds (g1_28.hpp): tight < 2p, Y < 6p, X < 14p, 2Q < 4p
    assert c["KP4_29"][13] >= tight_top
    assert c["KP8_29"][13] >= ((6 * M.P) >> 364) + 1
    assert c["KP16_29"][13] >= ((14 * M.P) >> 364) + 1
    assert c["KP8_30"][13] >= ((4 * M.P) >> 364) + 1
    # value-bound bookkeeping: 2^392 / p > 2520
    assert (1 << 392) // M.P >= 2520
    # column-sum bound of the product: 14 * (2^30)^2 + 14 * 2^56 + carry < 2^64
    assert 14 * (1 << 60) + 14 * (1 << 56) + (1 << 37) < (1 << 64)


def test_fr29_constants():
    c = arrays("fr29.hpp", "Fr29C")
    assert c["MOD"] == limbs(M.R, 29, 9)
    assert (-pow(M.R, -1, 1 << 29)) % (1 << 29) == (1 << 29) - 1 and c["MOD"][0] == 1   # what the product relies on
    assert c["ONE"] == limbs((1 << 261) % M.R, 29, 9)
    l = c["KP4"]
    assert value(l, 29) == 4 * M.R and all((1 << 29) - 1 <= x < (1 << 32) for x in l[:8])
    assert l[8] >= (2 * M.R) >> 232
    l8 = c["KP8"]
    assert value(l8, 29) == 8 * M.R and all((1 << 29) - 1 <= x < (1 << 31) for x in l8[:8])
    assert l8[8] >= ((4 * M.R) >> 232) + 1          # dominates a carry-propagated subtrahend below 4r (ntt.hpp unit_butterfly)
    assert 12 + 9 * 4 <= 70                         # value growth of the largest tile (2^11) after a product-free stage 1
    assert (1 << 261) // M.R >= 70
    assert 9 * (1 << 60) + 9 * (1 << 58) + (1 << 36) < (1 << 64)
    # quotient estimate: q = ((v >> 249) * QEST) >> 16 never overshoots and leaves < 2r
    import random
    rnd = random.Random(1)
    for _ in range(20000):
        v = rnd.randrange(0, 1 << 261)
        q = ((v >> 249) * c["QEST"]) >> 16
        assert q * M.R <= v < (q + 2) * M.R
    for v in (0, M.R - 1, M.R, 2 * M.R - 1, 46 * M.R, (1 << 261) - 1):
        q = ((v >> 249) * c["QEST"]) >> 16
        assert q * M.R <= v < (q + 2) * M.R


def test_fq30_safegcd_constants():
    """fq28_inv.hpp: the modulus in 30-bit limbs, its inverse mod 2^30, the Montgomery corrections R^3 for both limb forms, and the
    division-step bound the fixed round count rests on."""
    src = open(os.path.join(CSRC, "fq28_inv.hpp")).read()
    body = src[src.index("struct Fq30C"):]
    body = body[:body.index("\n};")]
    arr = {name: [int(v.strip().rstrip("u"), 16) for v in vals.split(",")] for name, vals in re.findall(r"(\w+)\[\d+\]\s*=\s*\{([^}]*)\}", body)}
    assert arr["MOD"] == limbs(M.P, 30, 13)
    inv30 = int(re.search(r"MOD_INV30 = (0x[0-9a-f]+)u", body).group(1), 16)
    assert inv30 == pow(M.P, -1, 1 << 30)
    assert arr["R3"] == limbs(pow(2, 3 * 392, M.P), 28, 14)
    assert arr["R3_384"] == limbs(pow(2, 3 * 384, M.P), 32, 12)
    # Bernstein-Yang Theorem 11.2 (delta = 1): f = p < 2^381 odd, 0 <= g < 2p: f^2 + 4 g^2 < 17 * 2^762 <= 5 * 2^(2d) for d = 381.9
    import math
    d = 0.5 * (762 + math.log2(17 / 5))
    rounds = int(re.search(r"for \(int it = 0; it < (\d+); it\+\+\)", src).group(1))
    assert (49 * d + 57) / 17 <= 30 * rounds
