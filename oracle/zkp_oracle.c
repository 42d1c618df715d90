/*
 * zkp_oracle.c -- CPU restatement of the reference's MSM / NTT hot path (plain C, gcc, unsigned __int128).
 * TEST INFRASTRUCTURE ONLY: see zkp_oracle.h for who may call it and for the parity status.
 *
 * Each function cites the reference file:line (under /root/reference) or the arkworks 0.4.x behaviour
 * (third-party, not vendored: ark-ff/ark-ec/ark-poly 0.4.2, ark-bls12-381 0.4.0) that it restates.
 */
#include "zkp_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------------
 * Generic N-limb Montgomery arithmetic (ark-ff MontBackend: values held as Montgomery residues).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int n;
    u64 p[6];
    u64 one[6]; /* R mod p */
    u64 r2[6];  /* R^2 mod p */
    u64 inv;    /* -p^-1 mod 2^64 */
} mont_t;

static mont_t FR, FQ, GLF;

static inline int ge_n(const u64 *a, const u64 *b, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline u64 add_n(u64 *o, const u64 *a, const u64 *b, int n) {
    u64 c = 0;
    for (int i = 0; i < n; i++) {
        u128 t = (u128)a[i] + b[i] + c;
        o[i] = (u64)t;
        c = (u64)(t >> 64);
    }
    return c;
}
static inline u64 sub_n(u64 *o, const u64 *a, const u64 *b, int n) {
    u64 br = 0;
    for (int i = 0; i < n; i++) {
        u128 t = (u128)a[i] - b[i] - br;
        o[i] = (u64)t;
        br = (u64)(t >> 64) & 1;
    }
    return br;
}
static inline int is_zero_n(const u64 *a, int n) {
    u64 x = 0;
    for (int i = 0; i < n; i++) x |= a[i];
    return x == 0;
}
static inline int eq_n(const u64 *a, const u64 *b, int n) { return memcmp(a, b, 8 * (size_t)n) == 0; }

#define DEF_FIELD(PFX, N, M)                                                                          \
    static inline void PFX##_add(u64 *o, const u64 *a, const u64 *b) {                                 \
        u64 t[N];                                                                                      \
        u64 c = add_n(t, a, b, N);                                                                     \
        if (c || ge_n(t, (M).p, N)) sub_n(t, t, (M).p, N);                                             \
        memcpy(o, t, 8 * N);                                                                           \
    }                                                                                                  \
    static inline void PFX##_sub(u64 *o, const u64 *a, const u64 *b) {                                 \
        u64 t[N];                                                                                      \
        if (sub_n(t, a, b, N)) add_n(t, t, (M).p, N);                                                  \
        memcpy(o, t, 8 * N);                                                                           \
    }                                                                                                  \
    static inline void PFX##_neg(u64 *o, const u64 *a) {                                               \
        if (is_zero_n(a, N)) { memset(o, 0, 8 * N); } else { u64 t[N]; sub_n(t, (M).p, a, N); memcpy(o, t, 8 * N); } \
    }                                                                                                  \
    static inline void PFX##_dbl(u64 *o, const u64 *a) { PFX##_add(o, a, a); }                         \
    /* CIOS Montgomery product a*b*R^-1 mod p */                                                       \
    static inline void PFX##_mul(u64 *o, const u64 *a, const u64 *b) {                                 \
        u64 t[N + 2];                                                                                  \
        memset(t, 0, sizeof t);                                                                        \
        for (int i = 0; i < N; i++) {                                                                  \
            u64 c = 0;                                                                                 \
            for (int j = 0; j < N; j++) {                                                              \
                u128 s = (u128)a[j] * b[i] + t[j] + c;                                                 \
                t[j] = (u64)s;                                                                         \
                c = (u64)(s >> 64);                                                                    \
            }                                                                                          \
            u128 s = (u128)t[N] + c;                                                                   \
            t[N] = (u64)s;                                                                             \
            t[N + 1] = (u64)(s >> 64);                                                                 \
            u64 m = t[0] * (M).inv;                                                                    \
            s = (u128)m * (M).p[0] + t[0];                                                             \
            c = (u64)(s >> 64);                                                                        \
            for (int j = 1; j < N; j++) {                                                              \
                s = (u128)m * (M).p[j] + t[j] + c;                                                     \
                t[j - 1] = (u64)s;                                                                     \
                c = (u64)(s >> 64);                                                                    \
            }                                                                                          \
            s = (u128)t[N] + c;                                                                        \
            t[N - 1] = (u64)s;                                                                         \
            t[N] = t[N + 1] + (u64)(s >> 64);                                                          \
        }                                                                                              \
        if (t[N] || ge_n(t, (M).p, N)) sub_n(t, t, (M).p, N);                                          \
        memcpy(o, t, 8 * N);                                                                           \
    }                                                                                                  \
    static inline void PFX##_sqr(u64 *o, const u64 *a) { PFX##_mul(o, a, a); }                         \
    static inline void PFX##_to_mont(u64 *o, const u64 *a) { PFX##_mul(o, a, (M).r2); }                \
    static inline void PFX##_from_mont(u64 *o, const u64 *a) {                                         \
        u64 one[N];                                                                                    \
        memset(one, 0, sizeof one);                                                                    \
        one[0] = 1;                                                                                    \
        PFX##_mul(o, a, one);                                                                          \
    }                                                                                                  \
    /* a^e, e given as canonical limbs */                                                              \
    static void PFX##_pow(u64 *o, const u64 *a, const u64 *e, int en) {                                \
        u64 acc[N], base[N];                                                                           \
        memcpy(acc, (M).one, 8 * N);                                                                   \
        memcpy(base, a, 8 * N);                                                                        \
        for (int i = 0; i < en; i++)                                                                   \
            for (int b = 0; b < 64; b++) {                                                             \
                if ((e[i] >> b) & 1) PFX##_mul(acc, acc, base);                                        \
                PFX##_sqr(base, base);                                                                 \
            }                                                                                          \
        memcpy(o, acc, 8 * N);                                                                         \
    }                                                                                                  \
    /* Fermat inverse a^(p-2); inverse of 0 is 0 (callers check) */                                    \
    static void PFX##_inv(u64 *o, const u64 *a) {                                                      \
        u64 e[N], two[N];                                                                              \
        memset(two, 0, sizeof two);                                                                    \
        two[0] = 2;                                                                                    \
        sub_n(e, (M).p, two, N);                                                                       \
        PFX##_pow(o, a, e, N);                                                                         \
    }

DEF_FIELD(fr, 4, FR)
DEF_FIELD(fq, 6, FQ)
DEF_FIELD(gl, 1, GLF)

static void mont_setup(mont_t *m, int n, const u64 *p) {
    m->n = n;
    memset(m->p, 0, sizeof m->p);
    memcpy(m->p, p, 8 * (size_t)n);
    u64 inv = 1; /* Newton: inv = p^-1 mod 2^64 */
    for (int i = 0; i < 6; i++) inv *= 2 - p[0] * inv;
    m->inv = (u64)0 - inv;
    /* one = 2^(64n) mod p by doubling 1 */
    u64 t[6] = {1, 0, 0, 0, 0, 0};
    for (int round = 0; round < 2; round++) {
        for (int i = 0; i < 64 * n; i++) {
            u64 c = add_n(t, t, t, n);
            if (c || ge_n(t, m->p, n)) sub_n(t, t, m->p, n);
        }
        if (round == 0) memcpy(m->one, t, sizeof t);
        else memcpy(m->r2, t, sizeof t);
    }
}

static u64 FR_ROOT32[4]; /* 7^((r-1)/2^32), Montgomery */
static u64 GL_ROOT32[1];
static u64 G1_GEN[12];   /* Montgomery */
static u64 FQ_B[6];      /* curve b = 4, Montgomery */

__attribute__((constructor)) static void oracle_init(void) {
    static const u64 p_fq[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
    static const u64 p_fr[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                                0x73eda753299d7d48ULL};
    static const u64 p_gl[1] = {0xffffffff00000001ULL};
    mont_setup(&FQ, 6, p_fq);
    mont_setup(&FR, 4, p_fr);
    mont_setup(&GLF, 1, p_gl);
    /* multiplicative generator 7, two-adicity 32 (ark-bls12-381 FrConfig; fri/src/fields/goldilocks.rs:5-6) */
    u64 seven[4] = {7, 0, 0, 0}, e[4], m7[4];
    fr_to_mont(m7, seven);
    u64 one4[4] = {1, 0, 0, 0};
    sub_n(e, FR.p, one4, 4); /* r-1 */
    for (int i = 0; i < 4; i++) e[i] = (i < 3 ? (e[i] >> 32) | (e[i + 1] << 32) : e[i] >> 32);
    fr_pow(FR_ROOT32, m7, e, 4);
    u64 g7[1] = {7}, ge[1] = {(p_gl[0] - 1) >> 32}, gm7[1];
    gl_to_mont(gm7, g7);
    gl_pow(GL_ROOT32, gm7, ge, 1);
    static const u64 gx[6] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                              0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
    static const u64 gy[6] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                              0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
    fq_to_mont(G1_GEN, gx);
    fq_to_mont(G1_GEN + 6, gy);
    u64 four[6] = {4, 0, 0, 0, 0, 0};
    fq_to_mont(FQ_B, four);
}

/* ------------------------------------------------------------------------------------------------
 * batch field helpers
 * ------------------------------------------------------------------------------------------------ */
#define BATCH1(NAME, FN, N) \
    void NAME(const u64 *in, u64 *out, size_t n) { for (size_t i = 0; i < n; i++) FN(out + i * N, in + i * N); }
#define BATCH2(NAME, FN, N) \
    void NAME(const u64 *a, const u64 *b, u64 *out, size_t n) { for (size_t i = 0; i < n; i++) FN(out + i * N, a + i * N, b + i * N); }
BATCH1(oracle_fr_to_mont, fr_to_mont, 4)
BATCH1(oracle_fr_from_mont, fr_from_mont, 4)
BATCH1(oracle_fq_to_mont, fq_to_mont, 6)
BATCH1(oracle_fq_from_mont, fq_from_mont, 6)
BATCH1(oracle_gl_to_mont, gl_to_mont, 1)
BATCH1(oracle_gl_from_mont, gl_from_mont, 1)
BATCH1(oracle_fr_inv, fr_inv, 4)
BATCH2(oracle_fr_mul, fr_mul, 4)
BATCH2(oracle_fr_add, fr_add, 4)
BATCH2(oracle_fr_sub, fr_sub, 4)
BATCH2(oracle_fq_mul, fq_mul, 6)

void oracle_fr_inner_product(const u64 *a, const u64 *b, size_t n, u64 out[4]) {
    u64 acc[4] = {0, 0, 0, 0}, t[4];
    for (size_t i = 0; i < n; i++) {
        fr_mul(t, a + 4 * i, b + 4 * i);
        fr_add(acc, acc, t);
    }
    memcpy(out, acc, 32);
}

static inline u64 splitmix64(u64 *st) {
    u64 z = (*st += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void oracle_rand_fr(u64 seed, size_t n, u64 *out) {
    u64 st = seed;
    for (size_t i = 0; i < n;) {
        u64 v[4];
        for (int k = 0; k < 4; k++) v[k] = splitmix64(&st);
        v[3] &= 0x7FFFFFFFFFFFFFFFULL;
        if (ge_n(v, FR.p, 4)) continue;
        fr_to_mont(out + 4 * i, v);
        i++;
    }
}
void oracle_rand_gl(u64 seed, size_t n, u64 *out) {
    u64 st = seed;
    for (size_t i = 0; i < n;) {
        u64 v = splitmix64(&st);
        if (v >= GLF.p[0]) continue;
        gl_to_mont(out + i, &v);
        i++;
    }
}

/* ------------------------------------------------------------------------------------------------
 * G1: y^2 = x^3 + 4 over Fq.  Jacobian coordinates like ark-ec's short_weierstrass::Projective.
 * ------------------------------------------------------------------------------------------------ */
typedef struct { u64 x[6], y[6], z[6]; } jac_t; /* infinity <=> z == 0 */

static inline void jac_set_inf(jac_t *p) {
    memcpy(p->x, FQ.one, 48);
    memcpy(p->y, FQ.one, 48);
    memset(p->z, 0, 48);
}
static inline int jac_is_inf(const jac_t *p) { return is_zero_n(p->z, 6); }
static inline void jac_from_affine(jac_t *p, const u64 *xy, int inf) {
    if (inf) { jac_set_inf(p); return; }
    memcpy(p->x, xy, 48);
    memcpy(p->y, xy + 6, 48);
    memcpy(p->z, FQ.one, 48);
}
/* dbl-2009-l (a = 0) */
static void jac_double(jac_t *o, const jac_t *p) {
    if (jac_is_inf(p)) { *o = *p; return; }
    u64 a[6], b[6], c[6], d[6], e[6], f[6], t[6], z3[6];
    fq_sqr(a, p->x);
    fq_sqr(b, p->y);
    fq_sqr(c, b);
    fq_add(t, p->x, b);
    fq_sqr(t, t);
    fq_sub(t, t, a);
    fq_sub(t, t, c);
    fq_dbl(d, t);
    fq_dbl(e, a);
    fq_add(e, e, a);
    fq_sqr(f, e);
    fq_mul(z3, p->y, p->z);
    fq_dbl(z3, z3);
    fq_dbl(t, d);
    fq_sub(o->x, f, t);
    fq_sub(t, d, o->x);
    fq_mul(t, e, t);
    fq_dbl(c, c);
    fq_dbl(c, c);
    fq_dbl(c, c);
    fq_sub(o->y, t, c);
    memcpy(o->z, z3, 48);
}
/* madd-2007-bl: Jacobian + affine (x2,y2), all special cases handled */
static void jac_add_affine(jac_t *o, const jac_t *p, const u64 *xy, int inf) {
    if (inf) { *o = *p; return; }
    if (jac_is_inf(p)) { jac_from_affine(o, xy, 0); return; }
    u64 z1z1[6], u2[6], s2[6], h[6], hh[6], i4[6], j[6], r[6], v[6], t[6];
    fq_sqr(z1z1, p->z);
    fq_mul(u2, xy, z1z1);
    fq_mul(s2, xy + 6, p->z);
    fq_mul(s2, s2, z1z1);
    fq_sub(h, u2, p->x);
    fq_sub(r, s2, p->y);
    if (is_zero_n(h, 6)) {
        if (is_zero_n(r, 6)) { jac_double(o, p); return; }
        jac_set_inf(o);
        return;
    }
    fq_dbl(r, r);
    fq_sqr(hh, h);
    fq_dbl(i4, hh);
    fq_dbl(i4, i4);
    fq_mul(j, h, i4);
    fq_mul(v, p->x, i4);
    u64 x3[6], y3[6], z3[6];
    fq_sqr(x3, r);
    fq_sub(x3, x3, j);
    fq_sub(x3, x3, v);
    fq_sub(x3, x3, v);
    fq_sub(t, v, x3);
    fq_mul(y3, r, t);
    fq_mul(t, p->y, j);
    fq_dbl(t, t);
    fq_sub(y3, y3, t);
    fq_add(z3, p->z, h);
    fq_sqr(z3, z3);
    fq_sub(z3, z3, z1z1);
    fq_sub(z3, z3, hh);
    memcpy(o->x, x3, 48);
    memcpy(o->y, y3, 48);
    memcpy(o->z, z3, 48);
}
/* add-2007-bl: Jacobian + Jacobian */
static void jac_add(jac_t *o, const jac_t *p, const jac_t *q) {
    if (jac_is_inf(p)) { *o = *q; return; }
    if (jac_is_inf(q)) { *o = *p; return; }
    u64 z1z1[6], z2z2[6], u1[6], u2[6], s1[6], s2[6], h[6], i4[6], j[6], r[6], v[6], t[6];
    fq_sqr(z1z1, p->z);
    fq_sqr(z2z2, q->z);
    fq_mul(u1, p->x, z2z2);
    fq_mul(u2, q->x, z1z1);
    fq_mul(s1, p->y, q->z);
    fq_mul(s1, s1, z2z2);
    fq_mul(s2, q->y, p->z);
    fq_mul(s2, s2, z1z1);
    fq_sub(h, u2, u1);
    fq_sub(r, s2, s1);
    if (is_zero_n(h, 6)) {
        if (is_zero_n(r, 6)) { jac_double(o, p); return; }
        jac_set_inf(o);
        return;
    }
    fq_dbl(r, r);
    fq_dbl(i4, h);
    fq_sqr(i4, i4);
    fq_mul(j, h, i4);
    fq_mul(v, u1, i4);
    u64 x3[6], y3[6], z3[6];
    fq_sqr(x3, r);
    fq_sub(x3, x3, j);
    fq_sub(x3, x3, v);
    fq_sub(x3, x3, v);
    fq_sub(t, v, x3);
    fq_mul(y3, r, t);
    fq_mul(t, s1, j);
    fq_dbl(t, t);
    fq_sub(y3, y3, t);
    fq_add(z3, p->z, q->z);
    fq_sqr(z3, z3);
    fq_sub(z3, z3, z1z1);
    fq_sub(z3, z3, z2z2);
    fq_mul(z3, z3, h);
    memcpy(o->x, x3, 48);
    memcpy(o->y, y3, 48);
    memcpy(o->z, z3, 48);
}
/* into_affine: one Fq inversion (ark-ec CurveGroup::into_affine) */
static void jac_to_affine(const jac_t *p, u64 *xy, uint8_t *inf) {
    if (jac_is_inf(p)) { memset(xy, 0, 96); *inf = 1; return; }
    u64 zi[6], zi2[6], zi3[6];
    fq_inv(zi, p->z);
    fq_sqr(zi2, zi);
    fq_mul(zi3, zi2, zi);
    fq_mul(xy, p->x, zi2);
    fq_mul(xy + 6, p->y, zi3);
    *inf = 0;
}
/* MSB-first double-and-add over the canonical scalar (ark-ec `Affine * Fr` -> mul_bigint) */
static void jac_mul_affine(jac_t *o, const u64 *xy, int inf, const u64 *k_canon) {
    jac_t acc;
    jac_set_inf(&acc);
    int started = 0;
    for (int i = 255; i >= 0; i--) {
        int bit = (k_canon[i >> 6] >> (i & 63)) & 1;
        if (started) jac_double(&acc, &acc);
        if (bit) { jac_add_affine(&acc, &acc, xy, inf); started = 1; }
    }
    *o = acc;
}

void oracle_g1_generator(u64 out_xy[12]) { memcpy(out_xy, G1_GEN, 96); }

int oracle_g1_on_curve(const u64 xy[12], uint8_t inf) {
    if (inf) return 1;
    u64 l[6], r[6];
    fq_sqr(l, xy + 6);
    fq_sqr(r, xy);
    fq_mul(r, r, xy);
    fq_add(r, r, FQ_B);
    return eq_n(l, r, 6);
}

void oracle_g1_mul(const u64 base_xy[12], uint8_t base_inf, const u64 scalar[4], u64 out_xy[12], uint8_t *out_inf) {
    u64 k[4];
    fr_from_mont(k, scalar);
    jac_t r;
    jac_mul_affine(&r, base_xy, base_inf, k);
    jac_to_affine(&r, out_xy, out_inf);
}

void oracle_g1_add(const u64 a_xy[12], uint8_t a_inf, const u64 b_xy[12], uint8_t b_inf, u64 out_xy[12], uint8_t *out_inf) {
    jac_t a;
    jac_from_affine(&a, a_xy, a_inf);
    jac_add_affine(&a, &a, b_xy, b_inf);
    jac_to_affine(&a, out_xy, out_inf);
}

/* kzg/src/srs.rs:48-63 */
void oracle_srs(const u64 secret[4], size_t n, u64 *out_xy) {
    u64 cur[4];
    memcpy(cur, FR.one, 32);
    for (size_t i = 0; i < n; i++) {
        uint8_t inf;
        oracle_g1_mul(G1_GEN, 0, cur, out_xy + 12 * i, &inf);
        fr_mul(cur, cur, secret);
    }
}

/* batch-normalise Jacobian points with one inversion (Montgomery's trick); zero z -> infinity */
static void jac_batch_to_affine(const jac_t *p, size_t n, u64 *out_xy, uint8_t *out_inf) {
    u64 *pref = (u64 *)malloc(48 * (n + 1));
    u64 acc[6];
    memcpy(acc, FQ.one, 48);
    for (size_t i = 0; i < n; i++) {
        memcpy(pref + 6 * i, acc, 48);
        if (!jac_is_inf(&p[i])) fq_mul(acc, acc, p[i].z);
    }
    u64 inv[6];
    fq_inv(inv, acc);
    for (size_t i = n; i-- > 0;) {
        if (jac_is_inf(&p[i])) {
            memset(out_xy + 12 * i, 0, 96);
            if (out_inf) out_inf[i] = 1;
            continue;
        }
        u64 zi[6], zi2[6], zi3[6];
        fq_mul(zi, inv, pref + 6 * i);
        fq_mul(inv, inv, p[i].z);
        fq_sqr(zi2, zi);
        fq_mul(zi3, zi2, zi);
        fq_mul(out_xy + 12 * i, p[i].x, zi2);
        fq_mul(out_xy + 12 * i + 6, p[i].y, zi3);
        if (out_inf) out_inf[i] = 0;
    }
    free(pref);
}

void oracle_g1_fixed_base_mul(const u64 *scalars, size_t n, u64 *out_xy, uint8_t *out_inf) {
    /* table[w][d] = d * 2^(8w) * G, d in 1..255, affine */
    enum { W = 32, D = 255 };
    jac_t *tj = (jac_t *)malloc(sizeof(jac_t) * W * D);
    jac_t base;
    jac_from_affine(&base, G1_GEN, 0);
    for (int w = 0; w < W; w++) {
        tj[w * D] = base;
        for (int d = 1; d < D; d++) jac_add(&tj[w * D + d], &tj[w * D + d - 1], &base);
        jac_add(&base, &tj[w * D + D - 1], &base); /* 256 * base */
    }
    u64 *tab = (u64 *)malloc(96 * W * D);
    jac_batch_to_affine(tj, W * D, tab, NULL);
    free(tj);
    jac_t *res = (jac_t *)malloc(sizeof(jac_t) * n);
    for (size_t i = 0; i < n; i++) {
        u64 k[4];
        fr_from_mont(k, scalars + 4 * i);
        jac_t acc;
        jac_set_inf(&acc);
        for (int w = 0; w < W; w++) {
            unsigned d = (unsigned)(k[w >> 3] >> ((w & 7) * 8)) & 0xFF;
            if (d) jac_add_affine(&acc, &acc, tab + 12 * (size_t)(w * D + d - 1), 0);
        }
        res[i] = acc;
    }
    jac_batch_to_affine(res, n, out_xy, out_inf);
    free(res);
    free(tab);
}

/* kzg/src/scheme.rs:88-94 */
void oracle_msm_naive(const u64 *points_xy, const uint8_t *points_inf, const u64 *scalars, size_t n,
                      u64 out_xy[12], uint8_t *out_inf) {
    u64 acc_xy[12];
    uint8_t acc_inf = 1;
    memset(acc_xy, 0, sizeof acc_xy);
    for (size_t i = 0; i < n; i++) {
        u64 t_xy[12];
        uint8_t t_inf;
        /* .map(|(cof, s)| s.mul(cof).into_affine()) */
        oracle_g1_mul(points_xy + 12 * i, points_inf ? points_inf[i] : 0, scalars + 4 * i, t_xy, &t_inf);
        if (i == 0) {
            memcpy(acc_xy, t_xy, 96);
            acc_inf = t_inf;
        } else {
            /* .reduce(|acc, e| acc.add(e).into_affine()) */
            oracle_g1_add(acc_xy, acc_inf, t_xy, t_inf, acc_xy, &acc_inf);
        }
    }
    /* .unwrap_or(G1Point::zero()) */
    memcpy(out_xy, acc_xy, 96);
    *out_inf = acc_inf;
}

void oracle_msm_pippenger(const u64 *points_xy, const uint8_t *points_inf, const u64 *scalars, size_t n,
                          u64 out_xy[12], uint8_t *out_inf) {
    unsigned c = 3;
    while (c < 16 && ((size_t)1 << (c + 4)) < n) c++;
    unsigned nwin = (255 + c - 1) / c;
    size_t nb = ((size_t)1 << c) - 1;
    u64 *canon = (u64 *)malloc(32 * (n ? n : 1));
    for (size_t i = 0; i < n; i++) fr_from_mont(canon + 4 * i, scalars + 4 * i);
    jac_t *buckets = (jac_t *)malloc(sizeof(jac_t) * nb);
    jac_t total;
    jac_set_inf(&total);
    for (int w = (int)nwin - 1; w >= 0; w--) {
        for (unsigned k = 0; k < c; k++) jac_double(&total, &total);
        for (size_t b = 0; b < nb; b++) jac_set_inf(&buckets[b]);
        unsigned lo = (unsigned)w * c;
        for (size_t i = 0; i < n; i++) {
            const u64 *k = canon + 4 * i;
            unsigned limb = lo >> 6, sh = lo & 63;
            u64 d = k[limb] >> sh;
            if (sh + c > 64 && limb < 3) d |= k[limb + 1] << (64 - sh);
            d &= nb;
            if (d) jac_add_affine(&buckets[d - 1], &buckets[d - 1], points_xy + 12 * i, points_inf ? points_inf[i] : 0);
        }
        jac_t run, sum;
        jac_set_inf(&run);
        jac_set_inf(&sum);
        for (size_t b = nb; b-- > 0;) {
            jac_add(&run, &run, &buckets[b]);
            jac_add(&sum, &sum, &run);
        }
        jac_add(&total, &total, &sum);
    }
    jac_to_affine(&total, out_xy, out_inf);
    free(buckets);
    free(canon);
}

/* ------------------------------------------------------------------------------------------------
 * NTT: serial in-order radix-2 (ark-poly 0.4 Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}
 * semantics; reference call sites plonk/src/prover.rs:374-375,463, plonk/src/circuit.rs:175,230-232).
 * ------------------------------------------------------------------------------------------------ */
#define DEF_NTT(PFX, N, M, ROOT32)                                                                    \
    static void PFX##_root(unsigned log_n, u64 *out) {                                                 \
        memcpy(out, ROOT32, 8 * N);                                                                    \
        for (unsigned i = log_n; i < 32; i++) PFX##_sqr(out, out);                                     \
    }                                                                                                  \
    static void PFX##_ntt(u64 *a, unsigned log_n, int inverse, const u64 *coset) {                     \
        size_t n = (size_t)1 << log_n;                                                                 \
        u64 w[N];                                                                                      \
        PFX##_root(log_n, w);                                                                          \
        if (inverse) PFX##_inv(w, w);                                                                  \
        if (coset && !inverse) { /* coset_fft: scale coefficient j by g^j */                           \
            u64 t[N];                                                                                  \
            memcpy(t, (M).one, 8 * N);                                                                 \
            for (size_t i = 0; i < n; i++) { PFX##_mul(a + i * N, a + i * N, t); PFX##_mul(t, t, coset); } \
        }                                                                                              \
        for (size_t i = 1, j = 0; i < n; i++) { /* bit reversal */                                     \
            size_t bit = n >> 1;                                                                       \
            for (; j & bit; bit >>= 1) j ^= bit;                                                       \
            j |= bit;                                                                                  \
            if (i < j) { u64 t[N]; memcpy(t, a + i * N, 8 * N); memcpy(a + i * N, a + j * N, 8 * N); memcpy(a + j * N, t, 8 * N); } \
        }                                                                                              \
        u64 *tw = (u64 *)malloc(8 * N * (n / 2 ? n / 2 : 1));                                          \
        memcpy(tw, (M).one, 8 * N);                                                                    \
        for (size_t i = 1; i < n / 2; i++) PFX##_mul(tw + i * N, tw + (i - 1) * N, w);                 \
        for (size_t len = 2; len <= n; len <<= 1) {                                                    \
            size_t half = len >> 1, step = n / len;                                                    \
            for (size_t s = 0; s < n; s += len)                                                        \
                for (size_t k = 0; k < half; k++) {                                                    \
                    u64 u[N], v[N];                                                                    \
                    memcpy(u, a + (s + k) * N, 8 * N);                                                 \
                    PFX##_mul(v, a + (s + k + half) * N, tw + k * step * N);                           \
                    PFX##_add(a + (s + k) * N, u, v);                                                  \
                    PFX##_sub(a + (s + k + half) * N, u, v);                                           \
                }                                                                                      \
        }                                                                                              \
        free(tw);                                                                                      \
        if (inverse) {                                                                                 \
            u64 ninv[N], nn[N];                                                                        \
            memset(nn, 0, sizeof nn);                                                                  \
            nn[0] = (u64)n;                                                                            \
            PFX##_to_mont(nn, nn);                                                                     \
            PFX##_inv(ninv, nn);                                                                       \
            if (coset) { /* coset_ifft: scale output j by g^-j */                                      \
                u64 gi[N], t[N];                                                                       \
                PFX##_inv(gi, coset);                                                                  \
                memcpy(t, ninv, 8 * N);                                                                \
                for (size_t i = 0; i < n; i++) { PFX##_mul(a + i * N, a + i * N, t); PFX##_mul(t, t, gi); } \
            } else                                                                                     \
                for (size_t i = 0; i < n; i++) PFX##_mul(a + i * N, a + i * N, ninv);                  \
        }                                                                                              \
    }

DEF_NTT(fr, 4, FR, FR_ROOT32)
DEF_NTT(gl, 1, GLF, GL_ROOT32)

void oracle_ntt_fr(u64 *data, unsigned log_n, int inverse, const u64 *coset) { fr_ntt(data, log_n, inverse, coset); }
void oracle_ntt_gl(u64 *data, unsigned log_n, int inverse, const u64 *coset) { gl_ntt(data, log_n, inverse, coset); }
void oracle_fr_root_of_unity(unsigned log_n, u64 out[4]) { fr_root(log_n, out); }
void oracle_gl_root_of_unity(unsigned log_n, u64 out[1]) { gl_root(log_n, out); }

/* ------------------------------------------------------------------------------------------------
 * polynomial helpers (ark-poly DensePolynomial semantics)
 * ------------------------------------------------------------------------------------------------ */
void oracle_poly_mul_fr(const u64 *a, size_t la, const u64 *b, size_t lb, u64 *out) {
    if (!la || !lb) return;
    memset(out, 0, 32 * (la + lb - 1));
    for (size_t i = 0; i < la; i++)
        for (size_t j = 0; j < lb; j++) {
            u64 t[4];
            fr_mul(t, a + 4 * i, b + 4 * j);
            fr_add(out + 4 * (i + j), out + 4 * (i + j), t);
        }
}
/* DensePolynomial::divide_by_vanishing_poly coefficient recurrence (plonk/src/prover.rs:446-455) */
int oracle_divide_by_vanishing_fr(const u64 *c, size_t len, size_t n, u64 *quot, u64 *rem) {
    if (len < n + 1) {
        memset(rem, 0, 32 * n);
        memcpy(rem, c, 32 * len);
        return 0;
    }
    u64 *w = (u64 *)malloc(32 * len);
    memcpy(w, c, 32 * len);
    for (size_t i = len; i-- > n;) {
        memcpy(quot + 4 * (i - n), w + 4 * i, 32);
        fr_add(w + 4 * (i - n), w + 4 * (i - n), w + 4 * i);
    }
    memcpy(rem, w, 32 * n);
    free(w);
    return 0;
}
void oracle_poly_eval_fr(const u64 *c, size_t len, const u64 z[4], u64 out[4]) {
    u64 acc[4] = {0, 0, 0, 0};
    for (size_t i = len; i-- > 0;) {
        fr_mul(acc, acc, z);
        fr_add(acc, acc, c + 4 * i);
    }
    memcpy(out, acc, 32);
}
void oracle_poly_div_linear_fr(const u64 *c, size_t len, const u64 z[4], u64 *quot) {
    u64 acc[4] = {0, 0, 0, 0};
    for (size_t i = len; i-- > 1;) {
        fr_mul(acc, acc, z);
        fr_add(acc, acc, c + 4 * i);
        memcpy(quot + 4 * (i - 1), acc, 32);
    }
}

/* ------------------------------------------------------------------------------------------------
 * FRI over Goldilocks
 * ------------------------------------------------------------------------------------------------ */
/* fri/src/fri_layer.rs:40-46 */
void oracle_fri_layer_eval(const u64 *coeffs, size_t d, u64 coset, unsigned log_D, u64 *out) {
    u64 w, root = GLF.one[0];
    gl_root(log_D, &w);
    for (size_t i = 0; i < ((size_t)1 << log_D); i++) {
        u64 x, acc = 0;
        gl_mul(&x, &root, &coset);
        for (size_t k = d; k-- > 0;) {
            gl_mul(&acc, &acc, &x);
            gl_add(&acc, &acc, coeffs + k);
        }
        out[i] = acc;
        gl_mul(&root, &root, &w);
    }
}
/* fri/src/prover.rs:34-42 */
void oracle_fri_fold(const u64 *coeffs, size_t d, u64 r, u64 *out) {
    for (size_t j = 0; j < (d + 1) / 2; j++) {
        u64 v = coeffs[2 * j];
        if (2 * j + 1 < d) {
            u64 t;
            gl_mul(&t, &r, coeffs + 2 * j + 1);
            gl_add(&v, &v, &t);
        }
        out[j] = v;
    }
}

/* ------------------------------------------------------------------------------------------------
 * CPU-best CONTEXT baseline: the bucket method and the radix-2 transform above on
 * ALL host cores with OpenMP.  Not the reference's algorithm (its evaluate_in_s is n double-and-add multiplications on one
 * thread) -- this is "what a CPU could do", reported next to it by bench.py's cpu_baseline.context; the single-threaded
 * functions above stay the parity oracle.  Results are checked against them in tests/test_oracle_golden.py.
 * ------------------------------------------------------------------------------------------------ */
#ifdef _OPENMP
#include <omp.h>
#endif
int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* Work items = (window, point chunk), each with a private bucket array; the window width balances insertions against the
 * running-sum reduction of every item. */
void oracle_msm_pippenger_mt(const u64 *points_xy, const uint8_t *points_inf, const u64 *scalars, size_t n, int threads,
                             u64 out_xy[12], uint8_t *out_inf) {
    if (threads < 1) threads = 1;
    unsigned best_c = 3;
    size_t best_chunks = 1;
    double best_cost = 1e300;
    for (unsigned c = 3; c <= 16; c++) {
        unsigned nwin = (255 + c - 1) / c;
        size_t chunks = (size_t)threads / nwin;  /* at most one item per thread: a second round would double the time */
        if (chunks < 1) chunks = 1;
        if (chunks > n / 64 + 1) chunks = n / 64 + 1;
        size_t items = (size_t)nwin * chunks, rounds = (items + (size_t)threads - 1) / (size_t)threads;
        /* a private bucket array beyond ~1 MB (c > 12 at 144 B per bucket) misses the core's cache on every insertion */
        double per_add = c > 12 ? 2.0 : 1.0;
        double cost = (double)rounds * (per_add * (double)n / (double)chunks + 2.0 * (double)((size_t)1 << c));
        if (cost < best_cost) { best_cost = cost; best_c = c; best_chunks = chunks; }
    }
    const unsigned c = best_c, nwin = (255 + c - 1) / c;
    const size_t chunks = best_chunks, nb = ((size_t)1 << c) - 1, items = (size_t)nwin * chunks;
    u64 *canon = (u64 *)malloc(32 * (n ? n : 1));
    jac_t *part = (jac_t *)malloc(sizeof(jac_t) * items);
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
    for (size_t i = 0; i < n; i++) fr_from_mont(canon + 4 * i, scalars + 4 * i);
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        jac_t *buckets = (jac_t *)malloc(sizeof(jac_t) * nb);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (size_t it = 0; it < items; it++) {
            const unsigned w = (unsigned)(it / chunks);
            const size_t q = it % chunks, i0 = n * q / chunks, i1 = n * (q + 1) / chunks;
            for (size_t b = 0; b < nb; b++) jac_set_inf(&buckets[b]);
            const unsigned lo = w * c, limb = lo >> 6, sh = lo & 63;
            for (size_t i = i0; i < i1; i++) {
                const u64 *k = canon + 4 * i;
                u64 d = k[limb] >> sh;
                if (sh + c > 64 && limb < 3) d |= k[limb + 1] << (64 - sh);
                d &= nb;
                if (d) jac_add_affine(&buckets[d - 1], &buckets[d - 1], points_xy + 12 * i, points_inf ? points_inf[i] : 0);
            }
            jac_t run, sum;
            jac_set_inf(&run);
            jac_set_inf(&sum);
            for (size_t b = nb; b-- > 0;) {
                jac_add(&run, &run, &buckets[b]);
                jac_add(&sum, &sum, &run);
            }
            part[it] = sum;
        }
        free(buckets);
    }
    jac_t total;
    jac_set_inf(&total);
    for (int w = (int)nwin - 1; w >= 0; w--) {
        for (unsigned k = 0; k < c; k++) jac_double(&total, &total);
        for (size_t q = 0; q < chunks; q++) jac_add(&total, &total, &part[(size_t)w * chunks + q]);
    }
    jac_to_affine(&total, out_xy, out_inf);
    free(part);
    free(canon);
}

/* out[i] = base^(first + i) * c0 for i < count, split over the threads (each starts from its own power) */
static void fr_pow_table_mt(u64 *out, size_t count, const u64 *base, const u64 *c0, int threads) {
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
#ifdef _OPENMP
        const size_t t = (size_t)omp_get_thread_num(), nt = (size_t)omp_get_num_threads();
#else
        const size_t t = 0, nt = 1;
#endif
        const size_t i0 = count * t / nt, i1 = count * (t + 1) / nt;
        if (i0 < i1) {
            u64 p[4], b[4];
            memcpy(p, c0, 32);
            memcpy(b, base, 32);
            for (size_t e = i0; e; e >>= 1) { /* p = c0 * base^i0 */
                if (e & 1) fr_mul(p, p, b);
                fr_sqr(b, b);
            }
            for (size_t i = i0; i < i1; i++) {
                memcpy(out + 4 * i, p, 32);
                fr_mul(p, p, base);
            }
        }
    }
}
/* the transform of oracle_ntt_fr (same in-order radix-2 stages, same outputs) with every loop over the threads */
void oracle_ntt_fr_mt(u64 *a, unsigned log_n, int inverse, const u64 *coset, int threads) {
    if (threads < 1) threads = 1;
    const size_t n = (size_t)1 << log_n;
    u64 w[4];
    fr_root(log_n, w);
    if (inverse) fr_inv(w, w);
    u64 *tw = (u64 *)malloc(32 * (n > 1 ? n : 2));
    if (coset && !inverse) {
        fr_pow_table_mt(tw, n, coset, FR.one, threads);
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
        for (size_t i = 0; i < n; i++) fr_mul(a + 4 * i, a + 4 * i, tw + 4 * i);
    }
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
    for (size_t i = 0; i < n; i++) {
        size_t j = 0;
        for (unsigned b = 0; b < log_n; b++) j |= ((i >> b) & 1) << (log_n - 1 - b);
        if (i < j) { u64 t[4]; memcpy(t, a + 4 * i, 32); memcpy(a + 4 * i, a + 4 * j, 32); memcpy(a + 4 * j, t, 32); }
    }
    fr_pow_table_mt(tw, n / 2 ? n / 2 : 1, w, FR.one, threads);
    /* the first stages block by block (2^12 elements = 128 KiB stay in the core's cache for 12 stages), the rest stage by stage */
    const unsigned log_blk = log_n < 12 ? log_n : 12;
    const size_t blk = (size_t)1 << log_blk;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
    for (size_t b0 = 0; b0 < n; b0 += blk)
        for (size_t len = 2; len <= blk; len <<= 1) {
            const size_t half = len >> 1, step = n / len;
            for (size_t s = b0; s < b0 + blk; s += len)
                for (size_t k = 0; k < half; k++) {
                    u64 u[4], v[4];
                    memcpy(u, a + 4 * (s + k), 32);
                    fr_mul(v, a + 4 * (s + k + half), tw + 4 * k * step);
                    fr_add(a + 4 * (s + k), u, v);
                    fr_sub(a + 4 * (s + k + half), u, v);
                }
        }
    for (size_t len = blk << 1; len <= n; len <<= 1) {
        const size_t half = len >> 1, step = n / len;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
        for (size_t j = 0; j < n / 2; j++) { /* butterfly j: block j / half, offset j % half */
            const size_t s = (j / half) * len, k = j % half;
            u64 u[4], v[4];
            memcpy(u, a + 4 * (s + k), 32);
            fr_mul(v, a + 4 * (s + k + half), tw + 4 * k * step);
            fr_add(a + 4 * (s + k), u, v);
            fr_sub(a + 4 * (s + k + half), u, v);
        }
    }
    if (inverse) {
        u64 ninv[4], nn[4] = {(u64)n, 0, 0, 0}, gi[4];
        fr_to_mont(nn, nn);
        fr_inv(ninv, nn);
        if (coset) {
            fr_inv(gi, coset);
            fr_pow_table_mt(tw, n, gi, ninv, threads);
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
            for (size_t i = 0; i < n; i++) fr_mul(a + 4 * i, a + 4 * i, tw + 4 * i);
        } else {
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
            for (size_t i = 0; i < n; i++) fr_mul(a + 4 * i, a + 4 * i, ninv);
        }
    }
    free(tw);
}
