// pairing_host.hpp -- BLS12-381 optimal ate pairing on the host, for the verifiers only (kzg/src/scheme.rs:143-160,
// 215-245; plonk/src/verifier.rs:130-157 call ark-ec's Bls12_381::pairing and compare two values).  Verification is never a
// throughput path: two pairings per proof, so everything here is the plainest correct form --
// affine Miller loop over the M-twist, dense Fq12 products, and the final exponentiation as ONE square-and-multiply with
// the full exponent (p^12 - 1) / r (no Frobenius constants, no cyclotomic tricks: ~20 ms per pairing).
//
// Tower: Fq2 = Fq[u]/(u^2 + 1), Fq6 = Fq2[v]/(v^3 - (1 + u)), Fq12 = Fq6[w]/(w^2 - v); G2 on y^2 = x^3 + 4 (1 + u).
// Untwist (x', y') -> (x'/w^2, y'/w^3).  The line through T with twist-slope L, evaluated at P and scaled by w^3 (killed by
// the final exponentiation), is  (L xT - yT) + (-L xP) v + yP v w.  Pinned bit for bit by tests/model/pairing_model.py
// (bilinear, non-degenerate, order r).  The reference only compares pairing values, so the normalisation is immaterial.
#pragma once
#include "host_ff.hpp"

namespace zkp {
namespace host {

struct Fq2 {
    HFq c0, c1;
    static Fq2 zero() { return Fq2{HFq::zero(), HFq::zero()}; }
    static Fq2 one() { return Fq2{HFq::one(), HFq::zero()}; }
    static Fq2 load(const uint64_t* p) { return Fq2{HFq::load(p), HFq::load(p + 6)}; }
    void store(uint64_t* p) const { c0.store(p); c1.store(p + 6); }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
    Fq2 operator+(const Fq2& o) const { return Fq2{c0 + o.c0, c1 + o.c1}; }
    Fq2 operator-(const Fq2& o) const { return Fq2{c0 - o.c0, c1 - o.c1}; }
    Fq2 neg() const { return Fq2{c0.neg(), c1.neg()}; }
    Fq2 operator*(const Fq2& o) const { return Fq2{c0 * o.c0 - c1 * o.c1, c0 * o.c1 + c1 * o.c0}; }
    Fq2 scale(const HFq& k) const { return Fq2{c0 * k, c1 * k}; }
    Fq2 mul_xi() const { return Fq2{c0 - c1, c0 + c1}; }  // * (1 + u)
    Fq2 inverse() const {
        const HFq n = (c0 * c0 + c1 * c1).inverse();
        return Fq2{c0 * n, c1.neg() * n};
    }
};

struct Fq6 {
    Fq2 c0, c1, c2;
    static Fq6 zero() { return Fq6{Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
    static Fq6 one() { return Fq6{Fq2::one(), Fq2::zero(), Fq2::zero()}; }
    bool operator==(const Fq6& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
    Fq6 operator+(const Fq6& o) const { return Fq6{c0 + o.c0, c1 + o.c1, c2 + o.c2}; }
    Fq6 operator*(const Fq6& o) const {
        return Fq6{c0 * o.c0 + (c1 * o.c2 + c2 * o.c1).mul_xi(), c0 * o.c1 + c1 * o.c0 + (c2 * o.c2).mul_xi(),
                   c0 * o.c2 + c1 * o.c1 + c2 * o.c0};
    }
    Fq6 mul_v() const { return Fq6{c2.mul_xi(), c0, c1}; }
};

struct Fq12 {
    Fq6 c0, c1;
    static Fq12 one() { return Fq12{Fq6::one(), Fq6::zero()}; }
    bool operator==(const Fq12& o) const { return c0 == o.c0 && c1 == o.c1; }
    Fq12 operator*(const Fq12& o) const { return Fq12{c0 * o.c0 + (c1 * o.c1).mul_v(), c0 * o.c1 + c1 * o.c0}; }
    Fq12 conj() const { return Fq12{c0, Fq6{c1.c0.neg(), c1.c1.neg(), c1.c2.neg()}}; }
    // 12 x 6 limbs in arkworks memory order (c0.c0.c0, c0.c0.c1, c0.c1.c0, ...), Montgomery form
    void store(uint64_t* p) const {
        const Fq2* f[6] = {&c0.c0, &c0.c1, &c0.c2, &c1.c0, &c1.c1, &c1.c2};
        for (int i = 0; i < 6; i++) f[i]->store(p + 12 * i);
    }
};

// (p^12 - 1) / r, little-endian 64-bit limbs (generated from tests/model/bigmodel.py constants)
static const uint64_t FINAL_EXP[68] = {
    0xc0bcb9b55df57510ull, 0x25f98630e68bfb24ull, 0x4406fbc8fbd5f489ull, 0x8e2f8491d12191a0ull,
    0x3e9d71650a6f8069ull, 0x226c2f011d4cab80ull, 0x67f67c4717489119ull, 0xaf3f881bd88592d7ull,
    0x1a67e49eeed2161dull, 0xe5b78c7869aeb218ull, 0xf6539314043f7bbcull, 0x73f62537f2701aaeull,
    0xaff1c910e9622d2aull, 0x6283313492caa9d4ull, 0x2e2f3ec2bea83d19ull, 0xa4c7e79fb02faa73ull,
    0x6c49637fd7961be1ull, 0x08e88adce8817745ull, 0x35de3f7a36399917ull, 0x9c1d9f7c31759c36ull,
    0xfa9e13c24ea820b0ull, 0x3fc56947a403577dull, 0xa4c1b6dcfc5cceb7ull, 0x1bbd81367066bca6ull,
    0x0418a3ef0bc62775ull, 0x49bf9b71a9f9e010ull, 0x511291097db60b17ull, 0x498345c6e5308f1cull,
    0x6d8823b19dadd7c2ull, 0x92004cedd556952cull, 0x4c6bec3ec03ef195ull, 0x0a1fad20044ce6adull,
    0xc55d3109cd15948dull, 0x334f46c02c3f0bd0ull, 0x3b5a62eb34c05739ull, 0x724538411d1676a5ull,
    0x127a1b5ad0463434ull, 0x61a474c5c85b0129ull, 0x8dfc8e2886ef965eull, 0x96532fef459f1243ull,
    0x40ee7169cdc10412ull, 0x9c40a68eb74bb22aull, 0x25118790f4684d0bull, 0x596bc293c8d4c01full,
    0x1064837f27611212ull, 0x077ffb10bf24dde4ull, 0xc49f570bcd2b01f3ull, 0x1a0c5bf24c374693ull,
    0x350da5359bc73ab6ull, 0xd2670d93e4d7acddull, 0xd39099b86e1ab656ull, 0x19328148978e2b0dull,
    0xb113f414386b0e88ull, 0x07a0dce2630d9aa4ull, 0xa927e7bb93753318ull, 0xe347aa68ad49466full,
    0x1c0ad0d6106feaf4ull, 0xc872ee83ff3a0f0full, 0x074e43b9a660835cull, 0xc0aadff5e9cfee9aull,
    0x30698e8cc7deada9ull, 0xd1073776ab353f2cull, 0x17848517badc3a43ull, 0x7363baa13f8d14a9ull,
    0xd4977b3f7d4507d0ull, 0x496a1c0a89ee0193ull, 0xdcc825b7e1bda9c0ull, 0x0000000002ee1db5ull};

inline Fq12 final_exponentiation(const Fq12& f) {
    Fq12 r = Fq12::one();
    bool started = false;
    for (int i = 68 - 1; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            if (started) r = r * r;
            if ((FINAL_EXP[i] >> b) & 1) {
                r = started ? r * f : f;
                started = true;
            }
        }
    return r;
}

// G2 affine on the twist; infinity as a flag (memory form of ark-ec's Affine<g2::Config>: x.c0 x.c1 y.c0 y.c1, 24 limbs)
struct G2Aff {
    Fq2 x, y;
    bool inf;
    static G2Aff infinity() { return G2Aff{Fq2::zero(), Fq2::zero(), true}; }
    static G2Aff load(const uint64_t* p, bool inf) { return inf ? infinity() : G2Aff{Fq2::load(p), Fq2::load(p + 12), false}; }
    void store(uint64_t* p, uint8_t* out_inf) const {
        if (inf) {
            std::memset(p, 0, 8 * 24);
            *out_inf = 1;
            return;
        }
        x.store(p);
        y.store(p + 12);
        *out_inf = 0;
    }
    static G2Aff generator() {
        static const uint64_t g[24] = {  // canonical coordinates (kzg/src/srs.rs:65 G2Point::generator()), not Montgomery
            0xd48056c8c121bdb8ull, 0x0bac0326a805bbefull, 0xb4510b647ae3d177ull, 0xc6e47ad4fa403b02ull, 0x260805272dc51051ull, 0x024aa2b2f08f0a91ull,
            0xe5ac7d055d042b7eull, 0x334cf11213945d57ull, 0xb5da61bbdc7f5049ull, 0x596bd0d09920b61aull, 0x7dacd3a088274f65ull, 0x13e02b6052719f60ull,
            0xe193548608b82801ull, 0x923ac9cc3baca289ull, 0x6d429a695160d12cull, 0xadfd9baa8cbdd3a7ull, 0x8cc9cdc6da2e351aull, 0x0ce5d527727d6e11ull,
            0xaaa9075ff05f79beull, 0x3f370d275cec1da1ull, 0x267492ab572e99abull, 0xcb3e287e85a763afull, 0x32acd2b02bc28b99ull, 0x0606c4a02ea734ccull};
        G2Aff r;
        r.inf = false;
        r.x = Fq2{HFq::load(g).to_mont(), HFq::load(g + 6).to_mont()};
        r.y = Fq2{HFq::load(g + 12).to_mont(), HFq::load(g + 18).to_mont()};
        return r;
    }
    G2Aff neg() const { return inf ? *this : G2Aff{x, y.neg(), false}; }
    bool on_curve() const {
        if (inf) return true;
        const HFq four = HFq::from_u64(4);
        return y * y == x * x * x + Fq2{four, four};
    }
    G2Aff dbl() const {
        if (inf || y.is_zero()) return infinity();
        const HFq two = HFq::from_u64(2), three = HFq::from_u64(3);
        const Fq2 lam = (x * x).scale(three) * y.scale(two).inverse();
        const Fq2 x3 = lam * lam - x - x;
        return G2Aff{x3, lam * (x - x3) - y, false};
    }
    G2Aff add(const G2Aff& o) const {
        if (inf) return o;
        if (o.inf) return *this;
        if (x == o.x) return y == o.y ? dbl() : infinity();
        const Fq2 lam = (o.y - y) * (o.x - x).inverse();
        const Fq2 x3 = lam * lam - x - o.x;
        return G2Aff{x3, lam * (x - x3) - y, false};
    }
    G2Aff mul(const uint64_t k[4]) const {  // canonical scalar, MSB first
        G2Aff r = infinity();
        for (int i = 255; i >= 0; i--) {
            r = r.dbl();
            if ((k[i >> 6] >> (i & 63)) & 1) r = r.add(*this);
        }
        return r;
    }
};

inline Fq12 line_eval(const Fq2& lam, const G2Aff& t, const HFq& px, const HFq& py) {
    const Fq2 a = lam * t.x - t.y, b = lam.scale(px.neg());
    return Fq12{Fq6{a, b, Fq2::zero()}, Fq6{Fq2::zero(), Fq2{py, HFq::zero()}, Fq2::zero()}};
}

// f_{|x|,Q}(P) conjugated (x < 0); P affine G1 coordinates in Montgomery form
inline Fq12 miller_loop(const HFq& px, const HFq& py, bool p_inf, const G2Aff& q) {
    if (p_inf || q.inf) return Fq12::one();
    const uint64_t X_ABS = 0xd201000000010000ull;
    const HFq two = HFq::from_u64(2), three = HFq::from_u64(3);
    Fq12 f = Fq12::one();
    G2Aff t = q;
    for (int b = 62; b >= 0; b--) {  // bit 63 is the leading one
        Fq2 lam = (t.x * t.x).scale(three) * t.y.scale(two).inverse();
        f = f * f * line_eval(lam, t, px, py);
        t = t.dbl();
        if ((X_ABS >> b) & 1) {
            lam = (q.y - t.y) * (q.x - t.x).inverse();
            f = f * line_eval(lam, t, px, py);
            t = t.add(q);
        }
    }
    return f.conj();
}

inline Fq12 pairing(const uint64_t p_xy[12], bool p_inf, const G2Aff& q) {
    const HFq px = p_inf ? HFq::zero() : HFq::load(p_xy), py = p_inf ? HFq::zero() : HFq::load(p_xy + 6);
    return final_exponentiation(miller_loop(px, py, p_inf, q));
}

}  // namespace host
}  // namespace zkp
