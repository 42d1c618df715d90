// host_ff.hpp -- host-side (CPU) field and G1 arithmetic used by the product's HOST code only:
// plan set-up (roots of unity, n^-1, coset inverses), the O(W*c) serial tail of an MSM (window
// combine + one inversion to affine) and the KzgScheme mirror's O(n) polynomial bookkeeping.
// It is not a fallback path: there is no host implementation of an MSM or NTT in this library.
//
// Values are arkworks-style Montgomery residues, little-endian u64 limbs (same bytes as the device's
// u32 limbs).
#pragma once
#include <cstdint>
#include <cstring>

namespace zkp {
namespace host {

typedef unsigned __int128 u128;

template <int N>
struct Mont {
    uint64_t p[N], one[N], r2[N], inv;
    explicit Mont(const uint64_t (&mod)[N]) {
        for (int i = 0; i < N; i++) p[i] = mod[i];
        uint64_t x = 1;
        for (int i = 0; i < 6; i++) x *= 2 - p[0] * x;
        inv = 0 - x;
        uint64_t t[N] = {1};
        for (int round = 0; round < 2; round++) {
            for (int i = 0; i < 64 * N; i++) {
                uint64_t c = add(t, t, t);
                if (c || ge(t, p)) sub(t, t, p);
            }
            std::memcpy(round == 0 ? one : r2, t, sizeof t);
        }
    }
    static bool ge(const uint64_t* a, const uint64_t* b) {
        for (int i = N - 1; i >= 0; i--) {
            if (a[i] != b[i]) return a[i] > b[i];
        }
        return true;
    }
    static uint64_t add(uint64_t* o, const uint64_t* a, const uint64_t* b) {
        uint64_t c = 0;
        for (int i = 0; i < N; i++) {
            u128 t = (u128)a[i] + b[i] + c;
            o[i] = (uint64_t)t;
            c = (uint64_t)(t >> 64);
        }
        return c;
    }
    static uint64_t sub(uint64_t* o, const uint64_t* a, const uint64_t* b) {
        uint64_t br = 0;
        for (int i = 0; i < N; i++) {
            u128 t = (u128)a[i] - b[i] - br;
            o[i] = (uint64_t)t;
            br = (uint64_t)(t >> 64) & 1;
        }
        return br;
    }
};

// Field element bound to a static Mont<N> instance supplied by Tag::ctx().
template <int N, class Tag>
struct El {
    uint64_t l[N];
    static const Mont<N>& M() { return Tag::ctx(); }
    static El zero() { El r; std::memset(r.l, 0, sizeof r.l); return r; }
    static El one() { El r; std::memcpy(r.l, M().one, sizeof r.l); return r; }
    static El from_u64(uint64_t v) { El r = zero(); r.l[0] = v; return r.to_mont(); }
    static El load(const uint64_t* p) { El r; std::memcpy(r.l, p, sizeof r.l); return r; }
    void store(uint64_t* p) const { std::memcpy(p, l, sizeof l); }
    bool is_zero() const { uint64_t x = 0; for (int i = 0; i < N; i++) x |= l[i]; return x == 0; }
    bool operator==(const El& o) const { return std::memcmp(l, o.l, sizeof l) == 0; }
    bool operator!=(const El& o) const { return !(*this == o); }
    El operator+(const El& o) const {
        El r;
        uint64_t c = Mont<N>::add(r.l, l, o.l);
        if (c || Mont<N>::ge(r.l, M().p)) Mont<N>::sub(r.l, r.l, M().p);
        return r;
    }
    El operator-(const El& o) const {
        El r;
        if (Mont<N>::sub(r.l, l, o.l)) Mont<N>::add(r.l, r.l, M().p);
        return r;
    }
    El neg() const { return is_zero() ? *this : zero() - *this; }
    El dbl() const { return *this + *this; }
    El operator*(const El& o) const {
        if (Tag::SPARE_BIT) return mul_nocarry(o);
        return mul_cios(o);
    }
    // CIOS without the extra carry word: valid when the modulus leaves the top bit of its top limb clear (Fq, Fr)
    El mul_nocarry(const El& o) const {
        const Mont<N>& m = M();
        uint64_t t[N] = {0};
#pragma GCC unroll 8
        for (int i = 0; i < N; i++) {
            u128 s = (u128)l[0] * o.l[i] + t[0];
            uint64_t A = (uint64_t)(s >> 64);
            const uint64_t q = (uint64_t)s * m.inv;
            u128 r = (u128)q * m.p[0] + (uint64_t)s;
            uint64_t C = (uint64_t)(r >> 64);
#pragma GCC unroll 8
            for (int j = 1; j < N; j++) {
                s = (u128)l[j] * o.l[i] + t[j] + A;
                A = (uint64_t)(s >> 64);
                r = (u128)q * m.p[j] + (uint64_t)s + C;
                C = (uint64_t)(r >> 64);
                t[j - 1] = (uint64_t)r;
            }
            t[N - 1] = C + A;
        }
        El r;
        if (Mont<N>::ge(t, m.p)) Mont<N>::sub(t, t, m.p);
        std::memcpy(r.l, t, sizeof r.l);
        return r;
    }
    El mul_cios(const El& o) const {
        const Mont<N>& m = M();
        uint64_t t[N + 2] = {0};
        for (int i = 0; i < N; i++) {
            uint64_t c = 0;
            for (int j = 0; j < N; j++) {
                u128 s = (u128)l[j] * o.l[i] + t[j] + c;
                t[j] = (uint64_t)s;
                c = (uint64_t)(s >> 64);
            }
            u128 s = (u128)t[N] + c;
            t[N] = (uint64_t)s;
            t[N + 1] = (uint64_t)(s >> 64);
            uint64_t q = t[0] * m.inv;
            s = (u128)q * m.p[0] + t[0];
            c = (uint64_t)(s >> 64);
            for (int j = 1; j < N; j++) {
                s = (u128)q * m.p[j] + t[j] + c;
                t[j - 1] = (uint64_t)s;
                c = (uint64_t)(s >> 64);
            }
            s = (u128)t[N] + c;
            t[N - 1] = (uint64_t)s;
            t[N] = t[N + 1] + (uint64_t)(s >> 64);
        }
        El r;
        if (t[N] || Mont<N>::ge(t, m.p)) Mont<N>::sub(t, t, m.p);
        std::memcpy(r.l, t, sizeof r.l);
        return r;
    }
    El sqr() const { return *this * *this; }
    El to_mont() const { El r2; std::memcpy(r2.l, M().r2, sizeof r2.l); return *this * r2; }
    El from_mont() const { El o = zero(); o.l[0] = 1; return *this * o; }
    El pow(const uint64_t* e, int en) const {
        El acc = one(), base = *this;
        for (int i = 0; i < en; i++)
            for (int b = 0; b < 64; b++) {
                if ((e[i] >> b) & 1) acc = acc * base;
                base = base.sqr();
            }
        return acc;
    }
    El pow_u64(uint64_t e) const { return pow(&e, 1); }
    El inverse_fermat() const {  // inverse of zero is zero
        uint64_t e[N], two[N] = {2};
        Mont<N>::sub(e, M().p, two);
        return pow(e, N);
    }
    // Inverse of zero is zero.  Binary extended Euclid on the residue itself (odd modulus with a spare top bit: x + p never
    // overflows the N limbs): u x1 = v x2 = the input (mod p) throughout, u and v shrink by at least one bit every two steps.  The
    // integer inverse of the Montgomery residue a R is a^-1 R^-1, so two Montgomery products by R^2 bring it back to a^-1 R.
    // Fq 12.7 us and Fr 5.5 us against 35 / 11 us for the Fermat power (tests/host/ff_inverse.cpp on the build container's CPU) -- a
    // commitment needs one per batch of results, a PLONK proof about ten.
    El inverse() const {
        if (!Tag::SPARE_BIT) return inverse_fermat();
        if (is_zero()) return *this;
        const Mont<N>& m = M();
        uint64_t u[N], v[N], x1[N] = {1}, x2[N] = {0};
        std::memcpy(u, l, sizeof u);
        std::memcpy(v, m.p, sizeof v);
        // load() does not reduce and raw ABI limbs reach this function: a non-canonical residue (p, 2p, ... still fits the limbs) is
        // reduced first -- with u a multiple of p the loop below would reach u == 0 and strip() would never return
        while (Mont<N>::ge(u, m.p)) Mont<N>::sub(u, u, m.p);
        {
            uint64_t any = 0;
            for (int i = 0; i < N; i++) any |= u[i];
            if (!any) return zero();
        }
        auto is_one = [](const uint64_t* a) {
            uint64_t x = a[0] ^ 1;
            for (int i = 1; i < N; i++) x |= a[i];
            return x == 0;
        };
        // a >>= k and x = x / 2^k mod p in one pass each (0 < k < 64): x + q p with q = x * (-p^-1) mod 2^k is divisible by 2^k
        auto shr = [](uint64_t* a, unsigned k) {
            for (int i = 0; i < N - 1; i++) a[i] = (a[i] >> k) | (a[i + 1] << (64 - k));
            a[N - 1] >>= k;
        };
        auto div2k = [&](uint64_t* x, unsigned k) {
            const uint64_t q = (x[0] * m.inv) & ((1ull << k) - 1);
            u128 c = 0;
            uint64_t t[N + 1];
            for (int i = 0; i < N; i++) {
                c += (u128)q * m.p[i] + x[i];
                t[i] = (uint64_t)c;
                c >>= 64;
            }
            t[N] = (uint64_t)c;
            for (int i = 0; i < N; i++) x[i] = (t[i] >> k) | (t[i + 1] << (64 - k));
        };
        auto strip = [&](uint64_t* a, uint64_t* x) {  // make a odd (a != 0)
            while (!(a[0] & 1)) {
                const unsigned k = a[0] ? (unsigned)__builtin_ctzll(a[0]) : 63u;  // (a zero low limb: 63 bits now, the rest next time round)
                shr(a, k);
                div2k(x, k);
            }
        };
        while (!is_one(u) && !is_one(v)) {
            strip(u, x1);
            strip(v, x2);
            if (Mont<N>::ge(u, v)) {
                Mont<N>::sub(u, u, v);
                if (Mont<N>::sub(x1, x1, x2)) Mont<N>::add(x1, x1, m.p);
            } else {
                Mont<N>::sub(v, v, u);
                if (Mont<N>::sub(x2, x2, x1)) Mont<N>::add(x2, x2, m.p);
            }
        }
        El y = load(is_one(u) ? x1 : x2), r2;
        std::memcpy(r2.l, m.r2, sizeof r2.l);
        return (y * r2) * r2;
    }
};

struct FrTag {
    static constexpr bool SPARE_BIT = true;
    static const Mont<4>& ctx() {
        static const uint64_t mod[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                                        0x73eda753299d7d48ULL};
        static const Mont<4> m(mod);
        return m;
    }
};
struct FqTag {
    static constexpr bool SPARE_BIT = true;
    static const Mont<6>& ctx() {
        static const uint64_t mod[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                        0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
        static const Mont<6> m(mod);
        return m;
    }
};
struct GlTag {
    static constexpr bool SPARE_BIT = false;  // p = 2^64 - 2^32 + 1 fills its limb
    static const Mont<1>& ctx() {
        static const uint64_t mod[1] = {0xffffffff00000001ULL};
        static const Mont<1> m(mod);
        return m;
    }
};
typedef El<4, FrTag> HFr;
typedef El<6, FqTag> HFq;
typedef El<1, GlTag> HGl;  // Montgomery form, as the reference stores Goldilocks elements

// 2^32-th primitive roots of unity: generator^((p-1)/2^32), generator 7 for both fields
// (ark-bls12-381 FrConfig; fri/src/fields/goldilocks.rs:6)
inline HFr fr_root_2_32() {
    const Mont<4>& m = FrTag::ctx();
    uint64_t e[4];
    for (int i = 0; i < 4; i++) e[i] = (i < 3 ? (m.p[i] >> 32) | (m.p[i + 1] << 32) : m.p[i] >> 32);  // (r-1)>>32 == r>>32
    return HFr::from_u64(7).pow(e, 4);
}
inline HFr fr_root_of_unity(unsigned log_n) {
    static const HFr root = fr_root_2_32();
    HFr w = root;
    for (unsigned i = log_n; i < 32; i++) w = w.sqr();
    return w;
}
inline HGl gl_root_of_unity(unsigned log_n) {
    static const HGl root = HGl::from_u64(7).pow_u64(0xffffffffULL);  // (p-1)/2^32 = 2^32 - 1
    HGl w = root;
    for (unsigned i = log_n; i < 32; i++) w = w.sqr();
    return w;
}

// --------------------------------------------------------------------------------------------
// G1 in XYZZ on the host (serial tail of the MSM, KzgScheme helpers).  Same formulas as g1.hpp.
// --------------------------------------------------------------------------------------------
struct HXyzz {
    HFq x, y, zz, zzz;
    static HXyzz infinity() { return HXyzz{HFq::zero(), HFq::zero(), HFq::zero(), HFq::zero()}; }
    static HXyzz from_affine(const uint64_t xy[12], bool inf) {
        if (inf) return infinity();
        return HXyzz{HFq::load(xy), HFq::load(xy + 6), HFq::one(), HFq::one()};
    }
    static HXyzz load(const uint64_t* p) { return HXyzz{HFq::load(p), HFq::load(p + 6), HFq::load(p + 12), HFq::load(p + 18)}; }
    void store(uint64_t* p) const { x.store(p); y.store(p + 6); zz.store(p + 12); zzz.store(p + 18); }
    bool is_inf() const { return zz.is_zero(); }
    HXyzz dbl() const {
        if (is_inf()) return *this;
        HFq u = y.dbl(), v = u.sqr(), w = u * v, s = x * v, xx = x.sqr(), m = xx.dbl() + xx;
        HXyzz r;
        r.x = m.sqr() - s.dbl();
        r.y = m * (s - r.x) - w * y;
        r.zz = v * zz;
        r.zzz = w * zzz;
        return r;
    }
    HXyzz add(const HXyzz& b) const {
        if (b.is_inf()) return *this;
        if (is_inf()) return b;
        HFq u1 = x * b.zz, u2 = b.x * zz, s1 = y * b.zzz, s2 = b.y * zzz;
        HFq p = u2 - u1, r = s2 - s1;
        if (p.is_zero()) return r.is_zero() ? dbl() : infinity();
        HFq pp = p.sqr(), ppp = p * pp, q = u1 * pp;
        HXyzz o;
        o.x = r.sqr() - ppp - q.dbl();
        o.y = r * (q - o.x) - s1 * ppp;
        o.zz = zz * b.zz * pp;
        o.zzz = zzz * b.zzz * ppp;
        return o;
    }
    // this + (x2, y2), the second operand affine and finite (madd-2008-s: 10 products instead of 14)
    HXyzz madd(const HFq& x2, const HFq& y2) const {
        if (is_inf()) return HXyzz{x2, y2, HFq::one(), HFq::one()};
        HFq u2 = x2 * zz, s2 = y2 * zzz;
        HFq p = u2 - x, r = s2 - y;
        if (p.is_zero()) return r.is_zero() ? HXyzz{x2, y2, HFq::one(), HFq::one()}.dbl() : infinity();
        HFq pp = p.sqr(), ppp = p * pp, q = x * pp;
        HXyzz o;
        o.x = r.sqr() - ppp - q.dbl();
        o.y = r * (q - o.x) - y * ppp;
        o.zz = zz * pp;
        o.zzz = zzz * ppp;
        return o;
    }
    HXyzz negate() const { HXyzz r = *this; r.y = r.y.neg(); return r; }
    // canonical affine coordinates (Montgomery limbs), the form the reference compares
    void to_affine(uint64_t out_xy[12], uint8_t* out_inf) const {
        if (is_inf()) {
            std::memset(out_xy, 0, 96);
            *out_inf = 1;
            return;
        }
        HFq zi3 = zzz.inverse();          // 1/ZZZ
        HFq zi2 = zi3 * zz;               // ZZ/ZZZ = 1/Z ; (1/Z)^2 = 1/ZZ
        zi2 = zi2.sqr();
        (x * zi2).store(out_xy);
        (y * zi3).store(out_xy + 6);
        *out_inf = 0;
    }
    // k * P for a canonical scalar k (4 limbs), MSB-first double-and-add
    HXyzz mul(const uint64_t k[4]) const {
        HXyzz acc = infinity();
        for (int i = 255; i >= 0; i--) {
            acc = acc.dbl();
            if ((k[i >> 6] >> (i & 63)) & 1) acc = acc.add(*this);
        }
        return acc;
    }
};

// Device-internal base-field element (fq28.hpp: 14 limbs of 28 bits, lazily reduced, Montgomery radix 2^392)
// -> host HFq (canonical, Montgomery radix 2^384).
inline HFq fq_from_limbs28(const uint32_t* l) {
    uint64_t w[7] = {0, 0, 0, 0, 0, 0, 0};  // up to 2^392 * small
    for (int i = 0; i < 14; i++) {           // limbs may exceed 28 bits only in the top position
        const int bit = 28 * i, k = bit >> 6, sh = bit & 63;
        u128 v = (u128)l[i] << sh;
        u128 s = (u128)w[k] + (uint64_t)v;
        w[k] = (uint64_t)s;
        u128 c = (s >> 64) + (v >> 64);
        for (int j = k + 1; j < 7 && c; j++) {
            u128 t = (u128)w[j] + c;
            w[j] = (uint64_t)t;
            c = t >> 64;
        }
    }
    const Mont<6>& m = FqTag::ctx();
    uint64_t p7[7];
    for (int i = 0; i < 6; i++) p7[i] = m.p[i];
    p7[6] = 0;
    for (;;) {  // value < 16p: a handful of subtractions
        bool ge = true;
        for (int i = 6; i >= 0; i--) {
            if (w[i] != p7[i]) { ge = w[i] > p7[i]; break; }
        }
        if (!ge) break;
        uint64_t br = 0;
        for (int i = 0; i < 7; i++) {
            u128 t = (u128)w[i] - p7[i] - br;
            w[i] = (uint64_t)t;
            br = (uint64_t)(t >> 64) & 1;
        }
    }
    HFq x = HFq::load(w);                         // = a * 2^392 mod p (as a plain integer)
    static const HFq c = HFq::from_u64(256).inverse();  // Montgomery form of 2^-8
    return x * c;                                  // = a * 2^384 mod p: the radix-2^384 Montgomery residue
}
inline HXyzz xyzz_from_internal(const uint32_t* p) {  // 4 x 16 words
    return HXyzz{fq_from_limbs28(p), fq_from_limbs28(p + 16), fq_from_limbs28(p + 32), fq_from_limbs28(p + 48)};
}

}  // namespace host
}  // namespace zkp
