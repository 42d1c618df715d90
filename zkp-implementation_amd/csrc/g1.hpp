// g1.hpp -- BLS12-381 G1 (y^2 = x^3 + 4) device arithmetic in extended Jacobian ("XYZZ") coordinates.
//
// The reference adds points through ark-ec's Jacobian `Projective` and normalises after every step
// (kzg/src/scheme.rs:92-93).  Bucket accumulation wants the cheapest mixed addition instead, so the
// device uses XYZZ (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): mixed add 8M+2S, general add 12M+2S,
// doubling 6M+3S (EFD: madd-2008-s, add-2008-s, dbl-2008-s-1, mdbl-2008-s-1).  The final result is
// normalised to the canonical affine pair, which is what makes bit-exact parity well defined.
#pragma once
#include "ff.hpp"

namespace zkp {

struct G1Affine {  // 96 bytes on the wire: x || y, Montgomery Fq, little-endian (include/zkp_hip.h)
    Fq x, y;
    static ZKP_DEV G1Affine load(const void* p) {
        G1Affine a;
        a.x = Fq::load(p);
        a.y = Fq::load(reinterpret_cast<const char*>(p) + 48);
        return a;
    }
    ZKP_DEV void store(void* p) const {
        x.store(p);
        y.store(reinterpret_cast<char*>(p) + 48);
    }
};

struct G1Xyzz {  // 192 bytes; infinity <=> ZZ == 0
    Fq x, y, zz, zzz;
    static ZKP_DEV G1Xyzz infinity() {
        G1Xyzz r;
        r.x = Fq::zero(); r.y = Fq::zero(); r.zz = Fq::zero(); r.zzz = Fq::zero();
        return r;
    }
    ZKP_DEV bool is_inf() const { return zz.is_zero(); }
    static ZKP_DEV G1Xyzz from_affine(const G1Affine& a) {
        G1Xyzz r;
        r.x = a.x; r.y = a.y; r.zz = Fq::one(); r.zzz = Fq::one();
        return r;
    }
    static ZKP_DEV G1Xyzz load(const void* p) {
        const char* c = reinterpret_cast<const char*>(p);
        G1Xyzz r;
        r.x = Fq::load(c); r.y = Fq::load(c + 48); r.zz = Fq::load(c + 96); r.zzz = Fq::load(c + 144);
        return r;
    }
    ZKP_DEV void store(void* p) const {
        char* c = reinterpret_cast<char*>(p);
        x.store(c); y.store(c + 48); zz.store(c + 96); zzz.store(c + 144);
    }
};

// 2*(x1,y1) from affine (mdbl-2008-s-1, a = 0)
ZKP_DEV G1Xyzz g1_double_affine(const G1Affine& p) {
    G1Xyzz r;
    Fq u = dbl(p.y);
    Fq v = sqr(u);
    Fq w = u * v;
    Fq s = p.x * v;
    Fq xx = sqr(p.x);
    Fq m = dbl(xx) + xx;
    r.x = sqr(m) - dbl(s);
    r.y = m * (s - r.x) - w * p.y;
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2*P in XYZZ (dbl-2008-s-1, a = 0)
ZKP_DEV G1Xyzz g1_double(const G1Xyzz& p) {
    if (p.is_inf()) return p;
    G1Xyzz r;
    Fq u = dbl(p.y);
    Fq v = sqr(u);
    Fq w = u * v;
    Fq s = p.x * v;
    Fq xx = sqr(p.x);
    Fq m = dbl(xx) + xx;
    r.x = sqr(m) - dbl(s);
    r.y = m * (s - r.x) - w * p.y;
    r.zz = v * p.zz;
    r.zzz = w * p.zzz;
    return r;
}

// acc += q for an affine q that is NOT the point at infinity (madd-2008-s).  All exceptional cases are
// handled: acc at infinity, q == acc (doubling) and q == -acc (result infinity).
ZKP_DEV void g1_madd(G1Xyzz& acc, const G1Affine& q) {
    if (acc.is_inf()) {
        acc = G1Xyzz::from_affine(q);
        return;
    }
    Fq u2 = q.x * acc.zz;
    Fq s2 = q.y * acc.zzz;
    Fq p = u2 - acc.x;
    Fq r = s2 - acc.y;
    if (p.is_zero()) {
        if (r.is_zero()) acc = g1_double_affine(q);
        else acc = G1Xyzz::infinity();
        return;
    }
    Fq pp = sqr(p);
    Fq ppp = p * pp;
    Fq qq = acc.x * pp;
    Fq x3 = sqr(r) - ppp - dbl(qq);
    acc.y = r * (qq - x3) - acc.y * ppp;
    acc.x = x3;
    acc.zz = acc.zz * pp;
    acc.zzz = acc.zzz * ppp;
}

// a += b, both XYZZ (add-2008-s), exceptional cases handled
ZKP_DEV void g1_add(G1Xyzz& a, const G1Xyzz& b) {
    if (b.is_inf()) return;
    if (a.is_inf()) {
        a = b;
        return;
    }
    Fq u1 = a.x * b.zz;
    Fq u2 = b.x * a.zz;
    Fq s1 = a.y * b.zzz;
    Fq s2 = b.y * a.zzz;
    Fq p = u2 - u1;
    Fq r = s2 - s1;
    if (p.is_zero()) {
        if (r.is_zero()) a = g1_double(a);
        else a = G1Xyzz::infinity();
        return;
    }
    Fq pp = sqr(p);
    Fq ppp = p * pp;
    Fq qq = u1 * pp;
    Fq x3 = sqr(r) - ppp - dbl(qq);
    a.y = r * (qq - x3) - s1 * ppp;
    a.x = x3;
    a.zz = a.zz * b.zz * pp;
    a.zzz = a.zzz * b.zzz * ppp;
}

}  // namespace zkp
