// fq28_inv.hpp -- modular inverse of an Fq28 element without an exponentiation: Bernstein-Yang "safegcd" division steps
// (https://gcd.cr.yp.to/papers.html#safegcd) on signed 30-bit limbs, the layout of libsecp256k1's modinv32 restated for a
// 381-bit modulus.  Every lane runs the same instruction stream (conditional moves, no data-dependent branch), so 64 lanes
// invert 64 different elements in lockstep; the only branch is wave-uniform (all lanes finished early).
//
// Cost: <= 37 rounds of (30 division steps on the low words + a 2x2 matrix applied to f, g, d, e); measured at three waves per
// SIMD (bench_micro/batch_affine.hip, profiles/r03_b_batched_affine.md): one inversion = 67 Fq28 products, the Fermat power 592.
//
// Bound: with f = p odd and 0 <= g < p < 2^381, gcd(f, g) is reached after at most floor((49 * 381 + 57) / 17) = 1101 division
// steps (Bernstein-Yang, Theorem 11.2, delta = 1 variant) <= 37 x 30.
#pragma once
#include "fq28.hpp"

namespace zkp {

constexpr int NL30 = 13;
constexpr int32_t M30 = 0x3fffffff;

struct Fq30C {  // generated like the constants of fq28.hpp (tests/test_limb_constants.py checks them against the modulus)
    // p in 13 limbs of 30 bits
    static constexpr int32_t MOD[13] = {0x3fffaaab, 0x27fbffff, 0x153ffffb, 0x2affffac, 0x30f6241e, 0x034a83da, 0x112bf673,
                                        0x12e13ce1, 0x2cd76477, 0x1ed90d2e, 0x29a4b1ba, 0x3a8e5ff9, 0x001a0111};
    static constexpr uint32_t MOD_INV30 = 0x00030003u;  // p^-1 mod 2^30
    // 2^(3 * 392) mod p as 28-bit limbs: mont_mul(x^-1, R^3) = (a R)^-1 R^3 R^-1 = a^-1 R for the Montgomery residue x = a R
    static constexpr uint32_t R3[14] = {0x1f7b890u, 0x294cc4du, 0x9f3af22u, 0xb5ba56cu, 0xcb5c0ccu, 0xc0d975cu, 0xc89a8c5u,
                                        0x6c968b4u, 0x22672eau, 0x91de8c9u, 0x35652a6u, 0x84977c8u, 0x424bbb9u, 0x00141abu};
    // 2^(3 * 384) mod p as 32-bit limbs: the same correction for the saturated Montgomery form of ff.hpp (radix 2^384)
    static constexpr uint32_t R3_384[12] = {0xd94ca1e0u, 0xed48ac6bu, 0x03a7adf8u, 0x315f831eu, 0x615e29ddu, 0x9a53352au,
                                            0x921e1761u, 0x34c04e5eu, 0x65724728u, 0x2512d435u, 0x91755d4du, 0x0aa63460u};
};

struct S30 {
    int32_t v[NL30];  // value = sum v[i] 2^(30 i); limbs 0..11 in [0, 2^30), the top limb carries the sign
};

// 30 division steps on the low words of f and g (delta = 1 variant, eta = -delta): returns the new eta and the transition
// matrix t = [u v; q r] with  t [f; g] = 2^30 [f'; g'].
ZKP_DEV int32_t divsteps_30(int32_t eta, uint32_t f0, uint32_t g0, int32_t& tu, int32_t& tv, int32_t& tq, int32_t& tr) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll
    for (int i = 0; i < 30; i++) {
        const uint32_t c1 = (uint32_t)(eta >> 31);      // delta > 0
        const uint32_t c2 = 0u - (g & 1u);              // g odd
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // (f, u, v) negated when delta > 0
        g += x & c2;
        q += y & c2;
        r += z & c2;
        const uint32_t sw = c1 & c2;                    // swap case: (delta, f, g) <- (1 - delta, g, (g - f) / 2)
        eta = (int32_t)(((uint32_t)eta ^ sw) - (sw + 1u));  // swap: eta <- -eta - 1 (delta <- 1 - delta); else eta - 1 (delta + 1)
        f += g & sw;
        u += q & sw;
        v += r & sw;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    tu = (int32_t)u; tv = (int32_t)v; tq = (int32_t)q; tr = (int32_t)r;
    return eta;
}

// (f, g) <- t (f, g) / 2^30 (exact)
ZKP_DEV void update_fg_30(S30& f, S30& g, int32_t u, int32_t v, int32_t q, int32_t r) {
    int64_t cf = (int64_t)u * f.v[0] + (int64_t)v * g.v[0];
    int64_t cg = (int64_t)q * f.v[0] + (int64_t)r * g.v[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < NL30; i++) {
        const int32_t fi = f.v[i], gi = g.v[i];
        cf += (int64_t)u * fi + (int64_t)v * gi;
        cg += (int64_t)q * fi + (int64_t)r * gi;
        f.v[i - 1] = (int32_t)cf & M30;
        g.v[i - 1] = (int32_t)cg & M30;
        cf >>= 30;
        cg >>= 30;
    }
    f.v[NL30 - 1] = (int32_t)cf;
    g.v[NL30 - 1] = (int32_t)cg;
}

// (d, e) <- t (d, e) / 2^30 mod p, both kept in (-2p, p)
ZKP_DEV void update_de_30(S30& d, S30& e, int32_t u, int32_t v, int32_t q, int32_t r) {
    const int32_t sd = d.v[NL30 - 1] >> 31, se = e.v[NL30 - 1] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0];
    int64_t ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
    md -= (int32_t)((Fq30C::MOD_INV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((Fq30C::MOD_INV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)Fq30C::MOD[0] * md;
    ce += (int64_t)Fq30C::MOD[0] * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < NL30; i++) {
        const int32_t di = d.v[i], ei = e.v[i];
        cd += (int64_t)u * di + (int64_t)v * ei + (int64_t)Fq30C::MOD[i] * md;
        ce += (int64_t)q * di + (int64_t)r * ei + (int64_t)Fq30C::MOD[i] * me;
        d.v[i - 1] = (int32_t)cd & M30;
        e.v[i - 1] = (int32_t)ce & M30;
        cd >>= 30;
        ce >>= 30;
    }
    d.v[NL30 - 1] = (int32_t)cd;
    e.v[NL30 - 1] = (int32_t)ce;
}

// r in (-2p, p) -> [0, p), negated first when `negate` (f ended at -1)
ZKP_DEV void normalize_30(S30& r, int32_t negate) {
    int32_t cond_add = r.v[NL30 - 1] >> 31;
#pragma unroll
    for (int i = 0; i < NL30; i++) r.v[i] += Fq30C::MOD[i] & cond_add;
    const int32_t cn = negate ? -1 : 0;
#pragma unroll
    for (int i = 0; i < NL30; i++) r.v[i] = (r.v[i] ^ cn) - cn;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL30 - 1; i++) {
        r.v[i] += c;
        c = r.v[i] >> 30;
        r.v[i] &= M30;
    }
    r.v[NL30 - 1] += c;
    cond_add = r.v[NL30 - 1] >> 31;
#pragma unroll
    for (int i = 0; i < NL30; i++) r.v[i] += Fq30C::MOD[i] & cond_add;
    c = 0;
#pragma unroll
    for (int i = 0; i < NL30 - 1; i++) {
        r.v[i] += c;
        c = r.v[i] >> 30;
        r.v[i] &= M30;
    }
    r.v[NL30 - 1] += c;
}

// non-negative integer below 2^390 given as 14 limbs with value(a) = sum l[i] 2^(28 i) (limbs may exceed 28 bits) -> 13 x 30 bit
ZKP_DEV S30 s30_from_fq28(const Fq28& a) {
    const Fq28 n = normalise(a);  // limbs 0..12 < 2^28, top limb small (value < 2^390)
    S30 r;
#pragma unroll
    for (int i = 0; i < NL30; i++) {
        const int bit = 30 * i, lo = bit / 28, sh = bit % 28;  // bits [bit, bit + 30) live in limbs lo, lo + 1 (and lo + 2 when sh > 26)
        uint64_t v = (uint64_t)n.l[lo] >> sh;
        if (lo + 1 < NL28) v |= (uint64_t)n.l[lo + 1] << (28 - sh);
        if (lo + 2 < NL28) v |= (uint64_t)n.l[lo + 2] << (56 - sh);
        r.v[i] = (int32_t)((uint32_t)v & (uint32_t)M30);
    }
    return r;
}
ZKP_DEV Fq28 fq28_from_s30(const S30& a) {  // a in [0, p)
    Fq28 r;
#pragma unroll
    for (int i = 0; i < NL28; i++) {
        const int bit = 28 * i, lo = bit / 30, sh = bit % 30;
        uint64_t v = (uint64_t)(uint32_t)a.v[lo] >> sh;
        if (lo + 1 < NL30) v |= (uint64_t)(uint32_t)a.v[lo + 1] << (30 - sh);
        r.l[i] = (uint32_t)v & MASK28;
    }
    return r;
}

// g in [0, 2p) -> [0, p): one conditional subtraction.  The division steps themselves accept any g below 2p, but for g = p they
// end with f = p and d = 1 instead of the documented 0; reduced first, every multiple of p takes the "0 -> 0" path.
ZKP_DEV void s30_reduce_once(S30& g) {
    S30 t;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL30; i++) {
        t.v[i] = g.v[i] - Fq30C::MOD[i] + c;
        if (i < NL30 - 1) {
            c = t.v[i] >> 30;  // arithmetic: -1 on a borrow
            t.v[i] &= M30;
        }
    }
    const int32_t keep = t.v[NL30 - 1] >> 31;  // all ones: g < p
#pragma unroll
    for (int i = 0; i < NL30; i++) g.v[i] = (g.v[i] & keep) | (t.v[i] & ~keep);
}

// g^-1 mod p as an integer in [0, p) for 0 <= g < 2p (0 and p -> 0)
ZKP_DEV S30 s30_modinv(S30 g) {
    s30_reduce_once(g);
    S30 f, d, e;
#pragma unroll
    for (int i = 0; i < NL30; i++) {
        f.v[i] = Fq30C::MOD[i];
        d.v[i] = 0;
        e.v[i] = i == 0 ? 1 : 0;
    }
    int32_t eta = -1;
#pragma unroll 1
    for (int it = 0; it < 37; it++) {
        int32_t u, v, q, r;
        eta = divsteps_30(eta, (uint32_t)f.v[0], (uint32_t)g.v[0], u, v, q, r);
        update_de_30(d, e, u, v, q, r);
        update_fg_30(f, g, u, v, q, r);
        int32_t nz = 0;
#pragma unroll
        for (int i = 0; i < NL30; i++) nz |= g.v[i];
        if (!__any(nz != 0)) break;  // wave-uniform: every lane has reached g = 0
    }
    // f = +-gcd = +-1 (or +-p for g = 0 mod p, where d = 0): d = +-1/g
    normalize_30(d, f.v[NL30 - 1] >> 31);
    return d;
}

// 1 / a for a Montgomery residue of fq28.hpp (radix 2^392; any value below 2p -- tight -- with the limb bounds of a product operand);
// 0 -> 0.  Result tight.
ZKP_DEV Fq28 fq28_inverse_gcd(const Fq28& a) {
    const S30 d = s30_modinv(s30_from_fq28(a));
    Fq28 r3;
#pragma unroll
    for (int i = 0; i < NL28; i++) r3.l[i] = Fq30C::R3[i];
    return fq28_from_s30(d) * r3;
}

// the same for the saturated form of ff.hpp (12 x 32-bit limbs, radix 2^384; input below 2p, canonical output)
ZKP_DEV Fq fq_inverse_gcd(const Fq& a) {
    S30 g;
#pragma unroll
    for (int i = 0; i < NL30; i++) {
        const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
        uint64_t v = (uint64_t)a.l[w];
        if (w + 1 < 12) v |= (uint64_t)a.l[w + 1] << 32;
        g.v[i] = (int32_t)((uint32_t)(v >> sh) & (uint32_t)M30);
    }
    const S30 d = s30_modinv(g);
    Fq y, r3;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 32 * w, lo = bit / 30, sh = bit % 30;
        uint64_t v = (uint64_t)(uint32_t)d.v[lo] >> sh;
        if (lo + 1 < NL30) v |= (uint64_t)(uint32_t)d.v[lo + 1] << (30 - sh);
        if (lo + 2 < NL30) v |= (uint64_t)(uint32_t)d.v[lo + 2] << (60 - sh);
        y.l[w] = (uint32_t)v;
        r3.l[w] = Fq30C::R3_384[w];
    }
    return y * r3;  // (a R)^-1 R^3 R^-1 = a^-1 R
}

}  // namespace zkp
