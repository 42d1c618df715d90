// g1_28.cuh -- G1 bucket arithmetic on the unsaturated base field (fq28.cuh): XYZZ coordinates, same formulas as
// g1.cuh (EFD madd-2008-s / add-2008-s / dbl-2008-s-1), with the lazy-reduction bookkeeping spelled out.
//
// Stored-point invariants (what a bucket / partial sum satisfies between operations):
//   X  : limbs 0..12 < 2^28, value < 14p          Y : limbs 0..12 < 2^28, value < 6p
//   ZZ, ZZZ : tight (products)                    infinity <=> ZZ is the all-zero limb vector
// Affine base points (internal form, 128 B): x, y canonical (< p), 28-bit limbs, Montgomery radix 2^392.
#pragma once
#include "fq28.cuh"

namespace zkp {

struct A28 {  // affine point, 2 x 64 B in memory
    Fq28 x, y;
    static ZKP_DEV A28 load(const uint4* p) {
        A28 a;
        a.x = Fq28::load(p);
        a.y = Fq28::load(p + 4);
        return a;
    }
    ZKP_DEV void store(uint4* p) const {
        x.store(p);
        y.store(p + 4);
    }
};

struct X28 {  // extended Jacobian point, 4 x 64 B in memory
    Fq28 x, y, zz, zzz;
    static ZKP_DEV X28 infinity() {
        X28 r;
        r.x = Fq28::zero(); r.y = Fq28::zero(); r.zz = Fq28::zero(); r.zzz = Fq28::zero();
        return r;
    }
    ZKP_DEV bool is_inf() const { return zz.all_zero(); }
    static ZKP_DEV X28 from_affine(const A28& a) {
        X28 r;
        r.x = a.x; r.y = a.y; r.zz = Fq28::one(); r.zzz = Fq28::one();
        return r;
    }
    static ZKP_DEV X28 load(const uint4* p) {
        X28 r;
        r.x = Fq28::load(p); r.y = Fq28::load(p + 4); r.zz = Fq28::load(p + 8); r.zzz = Fq28::load(p + 12);
        return r;
    }
    ZKP_DEV void store(uint4* p) const {
        x.store(p); y.store(p + 4); zz.store(p + 8); zzz.store(p + 12);
    }
};

// shared tail of the three formulas: given U1 (x of the left operand in the common denominator), S1 likewise,
// P = U2 - U1, R = S2 - S1 (both loose), PP = P^2 (tight) and the two denominators' products, produce X3, Y3.
//   X3 = R^2 - PPP - 2Q           value < 2p + 4p + 8p = 14p   (normalised)
//   Y3 = R (Q - X3) - S1 PPP      value < 2p + 4p = 6p          (normalised)
ZKP_DEV void xyzz_finish(Fq28& x3, Fq28& y3, const Fq28& r, const Fq28& pp, const Fq28& ppp, const Fq28& u1,
                         const Fq28& s1) {
    Fq28 q = u1 * pp;                                   // tight
    Fq28 rr = sqr(r);                                   // tight (r < 10p: 100 / 2520)
    x3 = normalise(sub8w(sub4(rr, ppp), q + q));        // limbs < 2^32 before, see fq28.cuh
    Fq28 t = sub16(q, x3);                              // < 18p, limbs < 2^30
    y3 = normalise(sub4(r * t, s1 * ppp));              // 10 * 18 / 2520 < 1
}

// 2 * (x, y) for an affine point (mdbl-2008-s-1, a = 0)
ZKP_DEV X28 g1_28_double_affine(const A28& p) {
    X28 o;
    Fq28 u = p.y + p.y;                  // < 2p (y canonical) or <= 8p (negated y); limbs < 2^31 -> normalise
    u = normalise(u);
    Fq28 v = sqr(u);                     // 64 / 2520 -> tight
    Fq28 w = u * v;
    Fq28 s = p.x * v;
    Fq28 xx = sqr(p.x);
    Fq28 m = xx + xx + xx;               // < 6p, limbs < 2^30
    o.x = normalise(sub8w(sqr(m), s + s));          // < 2p + 8p
    Fq28 t = sub16(s, o.x);                         // < 18p
    o.y = normalise(sub4(m * t, w * p.y));          // 6 * 18, 2 * 4 <= 2520; result < 6p
    o.zz = v;
    o.zzz = w;
    return o;
}

// 2 * P in XYZZ (dbl-2008-s-1, a = 0); P finite
ZKP_DEV X28 g1_28_double(const X28& p) {
    X28 o;
    Fq28 u = p.y + p.y;                  // < 12p, limbs < 2^29
    Fq28 v = sqr(u);                     // 144 / 2520
    Fq28 w = u * v;
    Fq28 s = p.x * v;                    // 14 * 2 / 2520
    Fq28 xx = sqr(p.x);                  // 196 / 2520
    Fq28 m = xx + xx + xx;
    o.x = normalise(sub8w(sqr(m), s + s));
    Fq28 t = sub16(s, o.x);
    o.y = normalise(sub4(m * t, w * p.y));
    o.zz = v * p.zz;
    o.zzz = w * p.zzz;
    return o;
}

// acc += q, q affine and finite (madd-2008-s).  Exceptional cases (acc infinite, q == acc, q == -acc) handled.
// q.y may be a negated coordinate (neg4: <= 4p, limbs < 2^30).
ZKP_DEV void g1_28_madd(X28& acc, const A28& q) {
    if (acc.is_inf()) {
        acc.x = q.x;
        acc.y = normalise(q.y);
        acc.zz = Fq28::one();
        acc.zzz = Fq28::one();
        return;
    }
    Fq28 u2 = q.x * acc.zz;              // tight
    Fq28 s2 = q.y * acc.zzz;             // 4 * 2 / 2520 -> tight
    Fq28 p = sub16(u2, acc.x);           // < 18p
    Fq28 r = sub8(s2, acc.y);            // < 10p
    Fq28 pp = sqr(p);                    // 324 / 2520 -> tight
    if (tight_is_zero_mod_p(pp)) {       // P == 0: same x
        if (tight_is_zero_mod_p(sqr(r))) acc = g1_28_double_affine(q);
        else acc = X28::infinity();
        return;
    }
    Fq28 ppp = p * pp;
    Fq28 x3, y3;
    xyzz_finish(x3, y3, r, pp, ppp, acc.x, acc.y);
    acc.x = x3;
    acc.y = y3;
    acc.zz = acc.zz * pp;
    acc.zzz = acc.zzz * ppp;
}

// a += b, both XYZZ (add-2008-s), exceptional cases handled
ZKP_DEV void g1_28_add(X28& a, const X28& b) {
    if (b.is_inf()) return;
    if (a.is_inf()) {
        a = b;
        return;
    }
    Fq28 u1 = a.x * b.zz;                // 14 * 2 / 2520 -> tight
    Fq28 u2 = b.x * a.zz;
    Fq28 s1 = a.y * b.zzz;
    Fq28 s2 = b.y * a.zzz;
    Fq28 p = sub4(u2, u1);               // < 6p
    Fq28 r = sub4(s2, s1);               // < 6p
    Fq28 pp = sqr(p);
    if (tight_is_zero_mod_p(pp)) {
        if (tight_is_zero_mod_p(sqr(r))) a = g1_28_double(a);
        else a = X28::infinity();
        return;
    }
    Fq28 ppp = p * pp;
    Fq28 x3, y3;
    xyzz_finish(x3, y3, r, pp, ppp, u1, s1);
    a.x = x3;
    a.y = y3;
    a.zz = a.zz * b.zz * pp;
    a.zzz = a.zzz * b.zzz * ppp;
}

}  // namespace zkp
