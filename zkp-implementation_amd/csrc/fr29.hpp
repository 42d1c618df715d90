// fr29.hpp -- BLS12-381 scalar field in UNSATURATED form for the NTT butterflies on gfx950: 9 limbs of 29 bits,
// Montgomery radix 2^261.  Same rationale as fq28.hpp: one v_mad_u64_u32 per partial product, no carry instructions
// (a product is 162 mads + ~40 cheap ops instead of ~580 instructions on saturated 8 x 32-bit limbs), limb-wise adds.
//
// Conventions
//   Data elements keep the VALUE they have in memory (the arkworks residue a * 2^256 mod r, a plain integer < r);
//   only the limb slicing changes.  Twiddles are stored as w * 2^261 mod r, so mul(x, tw) = x * w exactly in the
//   data's own domain.
//   mul(a, b): limbs of a < 2^31, limbs of b < 2^29 (column sums < 2^64), value(a) * value(b) <= 70 r^2
//   (2^261 / r = 70.66); result TIGHT: limbs < 2^29, value < 2r.
//   DIT butterfly (u, v, w) -> (u + v w, u - v w + 4r): values grow ADDITIVELY (+2r / +4r per stage), so a whole
//   radix-2^8 pass needs no reduction (< 2r + 8 * 4r = 34r < 70r); limbs are re-normalised every two stages.
#pragma once
#include "ff.hpp"

namespace zkp {

constexpr int NL29 = 9;
constexpr uint32_t MASK29 = (1u << 29) - 1;

struct Fr29C {  // constants generated from tests/model/bigmodel.py (checked by tests/test_limb_constants.py)
    static constexpr uint32_t MOD[9] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu,
                                        0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
    // -r^-1 mod 2^29 = 2^29 - 1 (r = 1 mod 2^32): m = (-acc) mod 2^29, and MOD[0] = 1
    static constexpr uint32_t ONE[9] = {0x1fffffbau, 0x0000022fu, 0x1cb61180u, 0x0a4e5c00u, 0x0ee8b1a2u,
                                        0x16e6aedfu, 0x1907f8bbu, 0x0853ddf7u, 0x004d043fu};  // 2^261 mod r
    // 4r with limb i raised by 2^29 (borrowed from limb i+1): dominates any TIGHT subtrahend
    static constexpr uint32_t KP4[9] = {0x20000004u, 0x3fffffdfu, 0x3e5bfefeu, 0x2d2017feu, 0x360154eeu,
                                        0x30101342u, 0x3483339cu, 0x2994cebdu, 0x01cfb69cu};
    // 8r likewise: dominates a carry-propagated subtrahend below 4r (limbs 0..7 < 2^29, top limb <= 4r >> 232)
    static constexpr uint32_t KP8[9] = {0x20000008u, 0x3fffffbfu, 0x3cb7fdfeu, 0x3a402ffeu, 0x2c02a9ddu,
                                        0x20202686u, 0x2906673au, 0x33299d7cu, 0x039f6d39u};
    static constexpr uint32_t QEST = 1130;  // floor(2^16 / (r / 2^249)): quotient estimate never overshoots
};

struct Fr29 {
    uint32_t l[NL29];
};

// Montgomery product (radix 2^261), product scanning with two accumulation chains per column
ZKP_DEV Fr29 operator*(const Fr29& a, const Fr29& b) {
    uint64_t acc = 0;
    uint32_t m[NL29];
    Fr29 r;
#pragma unroll
    for (int k = 0; k < NL29; k++) {
        uint64_t acc2 = 0;
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc2 += (uint64_t)m[i] * Fr29C::MOD[k - i];
        acc += acc2;
        m[k] = (0u - (uint32_t)acc) & MASK29;  // * (-r^-1) = * (2^29 - 1)
        acc += m[k];                           // * MOD[0] = 1
        acc >>= 29;
    }
#pragma unroll
    for (int k = NL29; k < 2 * NL29 - 1; k++) {
        uint64_t acc2 = 0;
#pragma unroll
        for (int i = k - NL29 + 1; i < NL29; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc2 += (uint64_t)m[i] * Fr29C::MOD[k - i];
        }
        acc += acc2;
        r.l[k - NL29] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r.l[NL29 - 1] = (uint32_t)acc;
    return r;
}
ZKP_DEV Fr29 sqr(const Fr29& a) { return a * a; }

ZKP_DEV Fr29 operator+(const Fr29& a, const Fr29& b) {  // lazy: limb and value bounds add
    Fr29 r;
#pragma unroll
    for (int i = 0; i < NL29; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// a - b + 4r for a TIGHT b: never negative in any limb
ZKP_DEV Fr29 sub_tight(const Fr29& a, const Fr29& b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < NL29; i++) r.l[i] = a.l[i] + (Fr29C::KP4[i] - b.l[i]);
    return r;
}
// a - b + 8r for a carry-propagated b < 4r
ZKP_DEV Fr29 sub_wide8(const Fr29& a, const Fr29& b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < NL29; i++) r.l[i] = a.l[i] + (Fr29C::KP8[i] - b.l[i]);
    return r;
}
// carry-propagate: limbs 0..7 < 2^29, the top limb absorbs the excess (value unchanged)
ZKP_DEV Fr29 normalise(const Fr29& a) {
    Fr29 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL29 - 1; i++) {
        uint32_t t = a.l[i] + c;
        r.l[i] = t & MASK29;
        c = t >> 29;
    }
    r.l[NL29 - 1] = a.l[NL29 - 1] + c;
    return r;
}

// saturated (8 x 32-bit) integer < 2^256  ->  9 x 29-bit limbs (same value)
ZKP_DEV Fr29 fr29_from_sat(const Fr& s) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < NL29; i++) {
        const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        uint64_t v = (uint64_t)s.l[w];
        if (w + 1 < 8) v |= (uint64_t)s.l[w + 1] << 32;
        r.l[i] = (uint32_t)(v >> sh) & MASK29;
    }
    return r;
}

// any lazily reduced element (value < 2^261) -> the canonical saturated residue (< r), i.e. the memory form
ZKP_DEV Fr fr29_to_canonical(const Fr29& x) {
    const Fr29 n = normalise(x);
    // quotient estimate from the top 12 bits: q <= floor(v / r), v - q r < 1.1 r
    const uint32_t q = ((n.l[NL29 - 1] >> 17) * Fr29C::QEST) >> 16;
    uint32_t y[NL29];
    uint64_t carry = 0;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL29; i++) {
        const uint64_t p = (uint64_t)q * Fr29C::MOD[i] + carry;
        const uint32_t lo = i < NL29 - 1 ? ((uint32_t)p & MASK29) : (uint32_t)p;
        carry = p >> 29;
        const int32_t d = (int32_t)n.l[i] - (int32_t)lo + borrow;
        if (i < NL29 - 1) {
            y[i] = (uint32_t)d & MASK29;
            borrow = d >> 29;
        } else {
            y[i] = (uint32_t)d;
        }
    }
    // one conditional subtraction of r
    uint32_t z[NL29];
    borrow = 0;
#pragma unroll
    for (int i = 0; i < NL29; i++) {
        const int32_t d = (int32_t)y[i] - (int32_t)Fr29C::MOD[i] + borrow;
        if (i < NL29 - 1) {
            z[i] = (uint32_t)d & MASK29;
            borrow = d >> 29;
        } else {
            z[i] = (uint32_t)d;
            borrow = d >> 31;  // negative top limb <=> y < r
        }
    }
    const bool keep = borrow != 0;
    uint32_t c[NL29];
#pragma unroll
    for (int i = 0; i < NL29; i++) c[i] = keep ? y[i] : z[i];
    Fr out;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int bit = 32 * w, i0 = bit / 29, o = bit - 29 * i0;
        uint64_t v = (uint64_t)c[i0] >> o;
        if (i0 + 1 < NL29) v |= (uint64_t)c[i0 + 1] << (29 - o);
        if (i0 + 2 < NL29 && 58 - o < 32) v |= (uint64_t)c[i0 + 2] << (58 - o);
        out.l[w] = (uint32_t)v;
    }
    return out;
}

// a TIGHT element (value < 2r < 2^256) -> saturated 8 x 32-bit words WITHOUT reduction: the cheap hand-off format between
// the passes of one transform (the next pass accepts any value < 2r; only the final pass canonicalises)
ZKP_DEV Fr fr29_pack_tight(const Fr29& c) {
    Fr out;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int bit = 32 * w, i0 = bit / 29, o = bit - 29 * i0;
        uint64_t v = (uint64_t)c.l[i0] >> o;
        if (i0 + 1 < NL29) v |= (uint64_t)c.l[i0 + 1] << (29 - o);
        if (i0 + 2 < NL29 && 58 - o < 32) v |= (uint64_t)c.l[i0 + 2] << (58 - o);
        out.l[w] = (uint32_t)v;
    }
    return out;
}

// twiddle-table conversion: saturated Montgomery residue (w * 2^256) -> w * 2^261 mod r in 29-bit limbs
ZKP_DEV Fr29 fr29_twiddle_from_mont(Fr s) {
#pragma unroll
    for (int k = 0; k < 5; k++) s = dbl(s);
    return fr29_from_sat(s);
}

}  // namespace zkp
