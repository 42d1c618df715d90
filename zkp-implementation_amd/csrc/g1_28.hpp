// g1_28.hpp -- G1 bucket arithmetic on the unsaturated base field (fq28.hpp): XYZZ coordinates, same formulas as
// g1.hpp (EFD madd-2008-s / add-2008-s / dbl-2008-s-1), with the lazy-reduction bookkeeping spelled out.
//
// Stored-point invariants (what a bucket / partial sum satisfies between operations):
//   X  : limbs 0..12 < 2^28, value < 14p          Y : limbs 0..12 < 2^28, value < 6p
//   ZZ, ZZZ : tight (products)                    infinity <=> ZZ is the all-zero limb vector
// Affine base points (internal form, 128 B): x, y canonical (< p), 28-bit limbs, Montgomery radix 2^392.
#pragma once
#include "fq28.hpp"

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
    // plane-major arrays (Fq28::load_s): chunk q of the point at p[q * stride]
    static ZKP_DEV X28 load_s(const uint4* p, uint64_t stride) {
        X28 r;
        r.x = Fq28::load_s(p, stride); r.y = Fq28::load_s(p + 4 * stride, stride);
        r.zz = Fq28::load_s(p + 8 * stride, stride); r.zzz = Fq28::load_s(p + 12 * stride, stride);
        return r;
    }
    ZKP_DEV void store_s(uint4* p, uint64_t stride) const {
        x.store_s(p, stride); y.store_s(p + 4 * stride, stride);
        zz.store_s(p + 8 * stride, stride); zzz.store_s(p + 12 * stride, stride);
    }
};

// shared tail of the three formulas: given U1 (x of the left operand in the common denominator), S1 likewise,
// P = U2 - U1, R = S2 - S1 (both loose), PP = P^2 (tight) and the two denominators' products, produce X3, Y3.
//   X3 = R^2 - PPP - 2Q           value < 2p + 4p + 8p = 14p   (normalised)
//   Y3 = R (Q - X3) - S1 PPP      value < 2p + 4p = 6p          (normalised)
// CHAIN: the plain products as strict multiply-add chains in asm (fq28_mul_chain / fq28_mul_chain2): msm_accumulate only
template <bool CHAIN>
ZKP_DEV Fq28 pmul(const Fq28& a, const Fq28& b) { return CHAIN ? fq28_mul_chain(a, b) : a * b; }
template <bool CHAIN>
ZKP_DEV void pmul2(const Fq28& a0, const Fq28& b0, const Fq28& a1, const Fq28& b1, Fq28& r0, Fq28& r1) {
    if (CHAIN) {
        fq28_mul_chain2(a0, b0, a1, b1, r0, r1);
    } else {
        r0 = a0 * b0;
        r1 = a1 * b1;
    }
}
template <bool CHAIN>
ZKP_DEV Fq28 psqr(const Fq28& a) { return CHAIN ? fq28_sqr_chain(a) : sqr(a); }
template <bool CHAIN = false>
ZKP_DEV void xyzz_finish(Fq28& x3, Fq28& y3, const Fq28& r, const Fq28& pp, const Fq28& ppp, const Fq28& u1,
                         const Fq28& s1) {
    Fq28 q = pmul<CHAIN>(u1, pp);                       // tight
    const Fq28 rn = normalise(r);                       // for the squaring and for the two-product reduction below
    Fq28 rr = psqr<CHAIN>(rn);                          // tight (r < 12p: 144 / 2520)
    x3 = normalise(sub8w(sub4(rr, ppp), q + q));        // limbs < 2^32 before, see fq28.hpp
    Fq28 t = sub16(q, x3);                              // < 18p, limbs < 2^30
    // Y3 = R t + (8p - S1) PPP with one reduction (fq28_mul2): limbs 2^28 x 2^30 and 2^30 x 2^28, (18 * 18 + 8 * 2) p^2 <= 2520 p^2;
    // the result is tight, which is inside the "< 6p, limbs < 2^28" contract of a stored Y
    y3 = CHAIN ? fq28_mul2_chain(rn, t, sub8(Fq28::zero(), s1), ppp) : fq28_mul2(rn, t, sub8(Fq28::zero(), s1), ppp);
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
template <bool CHAIN = false>
ZKP_DEV void g1_28_madd(X28& acc, const A28& q) {
    if (acc.is_inf()) {
        acc.x = q.x;
        acc.y = normalise(q.y);
        acc.zz = Fq28::one();
        acc.zzz = Fq28::one();
        return;
    }
    Fq28 u2, s2;
    pmul2<CHAIN>(q.x, acc.zz, q.y, acc.zzz, u2, s2);  // tight; 4 * 2 / 2520 -> tight
    Fq28 p = sub16(u2, acc.x);           // < 18p
    Fq28 r = sub8(s2, acc.y);            // < 10p
    Fq28 pp = psqr<CHAIN>(p);            // 324 / 2520 -> tight
    if (tight_is_zero_mod_p(pp)) {       // P == 0: same x
        if (tight_is_zero_mod_p(sqr(r))) acc = g1_28_double_affine(q);
        else acc = X28::infinity();
        return;
    }
    Fq28 ppp = pmul<CHAIN>(p, pp);
    Fq28 x3, y3;
    xyzz_finish<CHAIN>(x3, y3, r, pp, ppp, acc.x, acc.y);
    acc.x = x3;
    acc.y = y3;
    Fq28 zz3, zzz3;
    pmul2<CHAIN>(acc.zz, pp, acc.zzz, ppp, zz3, zzz3);
    acc.zz = zz3;
    acc.zzz = zzz3;
}

// acc += q where acc is still the AFFINE point the first insertion into an empty bucket left (ZZ = ZZZ = 1 implied, x canonical,
// y normalised and at most 4p) and q is affine with a canonical or normalised y: mmadd-2008-s.  U2 = X2, S2 = Y2, ZZ3 = PP,
// ZZZ3 = PPP -- six products instead of ten.  Every lane of a wave makes its second insertion in the same loop iteration, so the
// special case costs no divergence (msm_accumulate_run).  Returns false and leaves acc.x, acc.y alone when the two points share
// their x (the caller falls back to g1_28_madd, which knows how to double).  Bounds: P < 17p, R < 12p (144 / 2520), the rest as g1_28_madd.
template <bool CHAIN = false>
ZKP_DEV bool g1_28_mmadd(X28& acc, const A28& q) {
    Fq28 p = sub16(q.x, acc.x);
    Fq28 pp = psqr<CHAIN>(p);
    if (tight_is_zero_mod_p(pp)) return false;
    Fq28 r = sub8(q.y, acc.y);
    Fq28 ppp = pmul<CHAIN>(p, pp);
    Fq28 x3, y3;
    xyzz_finish<CHAIN>(x3, y3, r, pp, ppp, acc.x, acc.y);
    acc.x = x3;
    acc.y = y3;
    acc.zz = pp;
    acc.zzz = ppp;
    return true;
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

// ---- streaming add: dst = A + B with all three points in plane-major memory (dst aliases neither source) ---------------
// The same add-2008-s as g1_28_add, ordered so that few field elements are live at once: each coordinate is loaded right
// before its only use and each result stored as soon as it exists.  g1_28_add on two register-resident points needs ~200
// VGPRs (two waves per SIMD) and runs as load phase / arithmetic phase / store phase, which the waves of a level all enter
// together; this order fits three waves per SIMD and spreads the memory operations over the arithmetic.  The compiler
// barriers keep the loads from being hoisted back to the top.
#define ZKP_MEM_FENCE() asm volatile("" ::: "memory")
// P = 0: the operands have the same x.  Equal points double (dbl-2008-s-1, same bounds as g1_28_double), opposite points
// cancel.  Rare; written in the same style so that it does not set the caller's register budget.
ZKP_DEV void g1_28_same_x_stream(const uint4* __restrict__ pa, const uint4* __restrict__ pb, uint4* __restrict__ dst, uint64_t st) {
    bool opposite;
    {
        const Fq28 s1 = Fq28::load_s(pa + 4 * st, st) * Fq28::load_s(pb + 12 * st, st);
        const Fq28 s2 = Fq28::load_s(pb + 4 * st, st) * Fq28::load_s(pa + 12 * st, st);
        opposite = !tight_is_zero_mod_p(sqr(sub4(s2, s1)));
    }
    if (opposite) {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int q = 0; q < 16; q++) dst[q * st] = z;
        return;
    }
    ZKP_MEM_FENCE();
    Fq28 v, w;
    {
        const Fq28 y = Fq28::load_s(pa + 4 * st, st);
        const Fq28 u = y + y;                            // 2Y < 12p, limbs < 2^29
        v = sqr(u);
        w = u * v;
    }
    (v * Fq28::load_s(pa + 8 * st, st)).store_s(dst + 8 * st, st);
    (w * Fq28::load_s(pa + 12 * st, st)).store_s(dst + 12 * st, st);
    ZKP_MEM_FENCE();
    Fq28 s, mm;
    {
        const Fq28 x = Fq28::load_s(pa, st);
        s = x * v;
        const Fq28 xx = sqr(x);
        mm = xx + xx + xx;
    }
    const Fq28 x3 = normalise(sub8w(sqr(mm), s + s));
    x3.store_s(dst, st);
    const Fq28 t = sub16(s, x3);
    normalise(sub4(mm * t, w * Fq28::load_s(pa + 4 * st, st))).store_s(dst + 4 * st, st);
}
// (An out-of-line product for this add -- one copy of the multiplier instead of fourteen, against instruction-cache misses -- was
// built and measured in round 4: no gain, profiles/r04_c; removed.)
// (the body takes plain pointers: the in-place form below passes dst == pa.  Every chunk of A is loaded before the chunk of dst at the
// same place is stored -- each stored coordinate is computed from the loaded one -- so the sum may replace its first operand.)
template <bool CHAIN>
ZKP_DEV void g1_28_add_stream_body(const uint4* pa, const uint4* pb, uint4* dst, uint64_t st) {
    Fq28 u1, p, pp, zz3;
    {
        const Fq28 zz1 = Fq28::load_s(pa + 8 * st, st), zz2 = Fq28::load_s(pb + 8 * st, st);
        const bool inf1 = zz1.all_zero();
        if (inf1 || zz2.all_zero()) {  // an infinite operand: the sum is the other one
            const uint4* src = inf1 ? pb : pa;
#pragma unroll
            for (int q = 0; q < 16; q++) dst[q * st] = src[q * st];
            return;
        }
        Fq28 u2;
        pmul2<CHAIN>(Fq28::load_s(pa, st), zz2, Fq28::load_s(pb, st), zz1, u1, u2);  // 14 * 2 / 2520 -> tight
        p = sub4(u2, u1);                                // < 6p
        pp = psqr<CHAIN>(p);
        zz3 = pmul<CHAIN>(zz1, zz2);
    }
    if (tight_is_zero_mod_p(pp)) {
        g1_28_same_x_stream(pa, pb, dst, st);
        return;
    }
    pmul<CHAIN>(zz3, pp).store_s(dst + 8 * st, st);
    ZKP_MEM_FENCE();
    const Fq28 ppp = pmul<CHAIN>(p, pp);
    Fq28 s1, r;
    {
        const Fq28 zzz1 = Fq28::load_s(pa + 12 * st, st), zzz2 = Fq28::load_s(pb + 12 * st, st);
        Fq28 s2;
        pmul2<CHAIN>(Fq28::load_s(pa + 4 * st, st), zzz2, Fq28::load_s(pb + 4 * st, st), zzz1, s1, s2);
        r = sub4(s2, s1);                                // < 6p
        pmul<CHAIN>(pmul<CHAIN>(zzz1, zzz2), ppp).store_s(dst + 12 * st, st);
    }
    ZKP_MEM_FENCE();
    Fq28 x3, y3;
    xyzz_finish<CHAIN>(x3, y3, r, pp, ppp, u1, s1);
    x3.store_s(dst, st);
    y3.store_s(dst + 4 * st, st);
}
template <bool CHAIN = false>
ZKP_DEV void g1_28_add_stream(const uint4* __restrict__ pa, const uint4* __restrict__ pb, uint4* __restrict__ dst, uint64_t st) {
    g1_28_add_stream_body<CHAIN>(pa, pb, dst, st);
}
template <bool CHAIN = false>
ZKP_DEV void g1_28_add_stream_inplace(uint4* pa, const uint4* __restrict__ pb, uint64_t st) {  // A += B
    g1_28_add_stream_body<CHAIN>(pa, pb, pa, st);
}

// ---- cooperative add: FOUR adjacent lanes produce dst = A + B (all XYZZ, 16 chunks `stride` uint4 apart, in memory) -----
// The 14 products of add-2008-s fall into four rounds of (at most) four independent products, so a quad finishes an add in
// 4 product times instead of 14: used where the bucket reduction is latency-bound (few adds per level, msm.hpp).  Every
// lane runs the same instruction stream; its role j = lane & 3 only selects operands:
//   round 1   U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1        then d = (neighbour's product) - (own) = +-P | +-R
//   round 2   PP = d^2    | ZZ1 ZZ2     | RR = d^2     | ZZZ1 ZZZ2
//   round 3   PPP = P PP  | ZZ3 = . PP  | Q = U1 PP    | (idle)
//   round 4   V = S1 PPP  | (idle)      | T = R (Q-X3) | ZZZ3 = . PPP          X3 = RR - PPP - 2Q, Y3 = T - V on lane 2
// Values cross lanes with DPP quad_perm moves (quad_bcast / quad_xor1).  Infinity operands and P = 0 (equal or opposite
// points) are detected on the way and handed to lane 0's scalar g1_28_add.  Same bounds as g1_28_add / xyzz_finish.
// All four lanes of the quad must be active.
// DPP quad_perm moves (one full-rate VALU instruction per word, no trip through the LDS crossbar as with ds_bpermute): control
// byte = sel0 | sel1 << 2 | sel2 << 4 | sel3 << 6, lane i of every quad reads lane sel_i of the same quad.
template <int CTRL>
ZKP_DEV Fq28 quad_perm(const Fq28& v) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < NL28; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[i], CTRL, 0xf, 0xf, true);
    return r;
}
ZKP_DEV Fq28 quad_bcast(const Fq28& v, int src) {  // src is a literal at every call site: the switch folds away
    switch (src & 3) {
        case 0: return quad_perm<0x00>(v);
        case 1: return quad_perm<0x55>(v);
        case 2: return quad_perm<0xaa>(v);
        default: return quad_perm<0xff>(v);
    }
}
ZKP_DEV Fq28 quad_xor1(const Fq28& v) { return quad_perm<0xb1>(v); }  // [1, 0, 3, 2]
ZKP_DEV int quad_bcast0(int v) { return __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true); }
ZKP_DEV Fq28 fq28_select(bool c, const Fq28& a, const Fq28& b) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < NL28; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}
// dst may be srcA (msm_fold_parts): every load precedes the first store of its lane, and a quad touches its own entries only.
ZKP_DEV void g1_28_add_quad(const uint4* srcA, const uint4* __restrict__ srcB, uint4* dst, uint64_t stride, int j) {
    const bool odd = (j & 1) != 0, hi = (j & 2) != 0;
    const uint4* mine = odd ? srcB : srcA;    // operand whose X / Y this lane multiplies
    const uint4* other = odd ? srcA : srcB;   // operand whose ZZ / ZZZ it multiplies by
    const int zf = hi ? 12 : 8;               // uint4 offset of ZZZ / ZZ inside a point
    const Fq28 xy = Fq28::load_s(mine + (hi ? 4 : 0) * stride, stride);
    const Fq28 zo = Fq28::load_s(other + zf * stride, stride);
    const Fq28 zm = Fq28::load_s(mine + zf * stride, stride);
    // infinity <=> ZZ == 0: lanes 0/1 hold ZZ of B/A in zo and of A/B in zm
    const int inf_mine = zm.all_zero() ? 1 : 0, inf_other = zo.all_zero() ? 1 : 0;
    const int inf_a = quad_bcast0(inf_mine), inf_b = quad_bcast0(inf_other);
    const Fq28 m1 = xy * zo;                                   // U1 | U2 | S1 | S2
    const Fq28 d = sub4(quad_xor1(m1), m1);                    // P | -P | R | -R   (< 6p)
    const Fq28 m2 = fq28_select(odd, zo, d) * fq28_select(odd, zm, d);   // PP | ZZ1 ZZ2 | RR | ZZZ1 ZZZ2
    const Fq28 pp = quad_bcast(m2, 0);
    const int p_zero = quad_bcast0(tight_is_zero_mod_p(m2) ? 1 : 0);
    if (inf_a | inf_b | p_zero) {  // uniform over the quad
        if (j == 0) {
            X28 a = X28::load_s(srcA, stride);
            const X28 b = X28::load_s(srcB, stride);
            g1_28_add(a, b);
            a.store_s(dst, stride);
        }
        return;
    }
    const Fq28 u1 = quad_bcast(m1, 0);
    const Fq28 m3 = fq28_select(hi, u1, fq28_select(odd, m2, d)) * pp;   // PPP | ZZ3 | Q | (ZZZ12 PP, unused) ; lane 3: hi -> u1 pp
    const Fq28 ppp = quad_bcast(m3, 0);
    const Fq28 s1 = quad_bcast(m1, 2);
    // lane 2: X3 and T; other lanes compute the same expressions on don't-care values
    const Fq28 x3 = normalise(sub8w(sub4(m2, ppp), m3 + m3));  // RR - PPP - 2Q on lane 2
    const Fq28 t = sub16(m3, x3);                              // Q - X3
    const Fq28 lhs4 = fq28_select(hi, fq28_select(odd, m2, d), s1);   // lane 2: R, lane 3: ZZZ12, lanes 0/1: S1
    const Fq28 rhs4 = fq28_select(hi && !odd, t, ppp);
    const Fq28 m4 = lhs4 * rhs4;                                // V | (unused) | T | ZZZ3
    const Fq28 v = quad_bcast(m4, 0);
    if (j == 2) {
        x3.store_s(dst, stride);
        normalise(sub4(m4, v)).store_s(dst + 4 * stride, stride);                  // Y3 = T - V
    } else if (j == 1) {
        m3.store_s(dst + 8 * stride, stride);                                      // ZZ3
    } else if (j == 3) {
        m4.store_s(dst + 12 * stride, stride);                                     // ZZZ3
    }
}

}  // namespace zkp
