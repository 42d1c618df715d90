/*
 * fri_oracle.c -- CPU restatement of the reference's FRI commitment path around the Goldilocks NTT:
 * SHA-256 Merkle trees over decimal strings, the Fiat-Shamir transcript, generate_proof and verify.
 * TEST INFRASTRUCTURE ONLY (see zkp_oracle.h).  Plain C, canonical integers inside, arkworks memory form
 * (Montgomery residues, R = 2^64) at the interface.
 *
 * Follows   fri/src/hasher.rs:14-36, fri/src/merkle_tree.rs:42-129, fri/src/fiat_shamir/transcript.rs:30-139,
 *           fri/src/fri_layer.rs:36-56, fri/src/prover.rs:34-168, fri/src/verifier.rs:10-127,
 *           plonk/src/challenge.rs:36-77 (the BLS12-381 Fr flavour of the same challenge generator).
 * Third-party behaviour restated from the published sources, NOT verifiable offline (parity unpinned for these):
 *   ark-ff 0.4.2   Display for Fp = decimal of the canonical integer with leading zeros trimmed, so that ZERO PRINTS
 *                  AS THE EMPTY STRING (fields/models/fp/mod.rs, `trim_start_matches('0')`); set the environment
 *                  variable ZKP_FRI_ZERO_AS_0=1 to print "0" instead;
 *                  UniformRand for Fp = fill N u64 limbs, mask the top limb, reject >= p, value used AS the Montgomery
 *                  residue; from_le_bytes_mod_order = little-endian integer mod p;
 *   rand 0.8.5     StdRng = ChaCha12Rng; rand_core 0.6 seed_from_u64 = PCG32 expansion of the u64 to a 32-byte key;
 *   rand_chacha 0.3 64-bit block counter in words 12-13, stream id 0 in words 14-15, words consumed in order,
 *                  next_u64 = low word first;
 *   sha2 0.10      SHA-256 (FIPS 180-4; pinned by the NIST "abc" vector in tests).
 */
#include "zkp_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;
typedef unsigned __int128 u128;

#define GLP 0xffffffff00000001ull

/* ---------------------------------------------------------------- SHA-256 (FIPS 180-4) */
static const u32 K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
typedef struct {
    u32 h[8];
    u8 buf[64];
    u64 len;
} sha_t;
static u32 rotr(u32 x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha_block(u32 h[8], const u8 *p) {
    u32 w[64];
    for (int i = 0; i < 16; i++) w[i] = (u32)p[4 * i] << 24 | (u32)p[4 * i + 1] << 16 | (u32)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        u32 s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        u32 s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], k = h[7];
    for (int i = 0; i < 64; i++) {
        u32 t1 = k + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        u32 t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        k = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += k;
}
static void sha_init(sha_t *s) {
    static const u32 iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(s->h, iv, sizeof iv);
    s->len = 0;
}
static void sha_update(sha_t *s, const void *data, size_t n) {
    const u8 *p = (const u8 *)data;
    while (n) {
        size_t off = s->len % 64, take = 64 - off < n ? 64 - off : n;
        memcpy(s->buf + off, p, take);
        s->len += take;
        p += take;
        n -= take;
        if (s->len % 64 == 0) sha_block(s->h, s->buf);
    }
}
static void sha_final(sha_t *s, u8 out[32]) {
    u64 bits = s->len * 8;
    u8 pad = 0x80;
    sha_update(s, &pad, 1);
    pad = 0;
    while (s->len % 64 != 56) sha_update(s, &pad, 1);
    u8 lb[8];
    for (int i = 0; i < 8; i++) lb[i] = (u8)(bits >> (56 - 8 * i));
    sha_update(s, lb, 8);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (u8)(s->h[i] >> 24);
        out[4 * i + 1] = (u8)(s->h[i] >> 16);
        out[4 * i + 2] = (u8)(s->h[i] >> 8);
        out[4 * i + 3] = (u8)s->h[i];
    }
}
void oracle_sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    sha_t s;
    sha_init(&s);
    sha_update(&s, msg, len);
    sha_final(&s, out);
}

/* ---------------------------------------------------------------- Goldilocks, canonical integers */
static u64 gmul(u64 a, u64 b) { return (u64)((u128)a * b % GLP); }
static u64 gadd(u64 a, u64 b) { return (u64)(((u128)a + b) % GLP); }
static u64 gsub(u64 a, u64 b) { return (u64)(((u128)a + GLP - b) % GLP); }
static u64 gpow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = gmul(r, a);
        a = gmul(a, a);
        e >>= 1;
    }
    return r;
}
static u64 ginv(u64 a) { return gpow(a, GLP - 2); }
static u64 to_mont(u64 a) { return (u64)(((u128)a << 64) % GLP); }
static u64 from_mont(u64 a) { return gmul(a, 0xfffffffe00000001ull); } /* 2^-64 = 2^128 = -2^32 (2^96 = -1) */
static u64 groot(unsigned log_n) { /* ark-ff FftField: TWO_ADIC_ROOT_OF_UNITY = 7^((p-1)/2^32), squared down */
    u64 w = gpow(7, (GLP - 1) >> 32);
    for (unsigned i = log_n; i < 32; i++) w = gmul(w, w);
    return w;
}

/* ark-ff 0.4.2 Display for Fp: decimal, leading zeros trimmed (zero -> "") */
static int zero_as_0(void) {
    const char *e = getenv("ZKP_FRI_ZERO_AS_0");
    return e && e[0] == '1';
}
static size_t gl_display(u64 canon, char *out) {
    char tmp[24];
    size_t n = 0;
    while (canon) {
        tmp[n++] = (char)('0' + canon % 10);
        canon /= 10;
    }
    if (n == 0 && zero_as_0()) tmp[n++] = '0';
    for (size_t i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
    return n;
}
/* F::from_le_bytes_mod_order(&digest): the 32 bytes as a little-endian integer, mod p */
static u64 digest_to_gl(const u8 h[32]) {
    u64 acc = 0;
    for (int i = 31; i >= 0; i--) acc = gadd(gmul(acc, 256), h[i]);
    return acc;
}
/* hasher.rs:14-19 / 30-35 */
static u64 hash_slice_c(const u64 *canon, size_t n) {
    sha_t s;
    sha_init(&s);
    char buf[24];
    for (size_t i = 0; i < n; i++) {
        size_t l = gl_display(canon[i], buf);
        sha_update(&s, buf, l);
    }
    u8 h[32];
    sha_final(&s, h);
    return digest_to_gl(h);
}
void oracle_gl_hash(const uint64_t *in_mont, size_t n, uint64_t *out_mont) {
    for (size_t i = 0; i < n; i++) {
        u64 c = from_mont(in_mont[i]);
        out_mont[i] = to_mont(hash_slice_c(&c, 1));
    }
}
void oracle_gl_hash_slice(const uint64_t *in_mont, size_t n, uint64_t out_mont[1]) {
    u64 *c = (u64 *)malloc(8 * (n ? n : 1));
    for (size_t i = 0; i < n; i++) c[i] = from_mont(in_mont[i]);
    out_mont[0] = to_mont(hash_slice_c(c, n));
    free(c);
}

/* merkle_tree.rs:42-63.  nodes = level 0 (n hashes of the leaves), level 1 (ceil(n/2)), ... up to depth =
 * log2(next_pow2(n)) further levels, concatenated; canonical values.  Returns the number of nodes. */
static size_t merkle_depth(size_t n) {
    size_t p = 1, d = 0;
    while (p < n) {
        p <<= 1;
        d++;
    }
    return d;
}
static size_t merkle_build(const u64 *leaves, size_t n, u64 *nodes) {
    size_t depth = merkle_depth(n), off = 0, len = n;
    for (size_t i = 0; i < n; i++) nodes[i] = hash_slice_c(leaves + i, 1);
    for (size_t l = 0; l < depth; l++) {
        size_t nl = (len + 1) / 2;
        for (size_t j = 0; j < nl; j++) nodes[off + len + j] = hash_slice_c(nodes + off + 2 * j, 2 * j + 1 < len ? 2 : 1);
        off += len;
        len = nl;
    }
    return off + len;
}
size_t oracle_merkle_node_count(size_t n) {
    size_t depth = merkle_depth(n), total = n, len = n;
    for (size_t l = 0; l < depth; l++) {
        len = (len + 1) / 2;
        total += len;
    }
    return total;
}
void oracle_merkle_tree(const uint64_t *leaves_mont, size_t n, uint64_t *nodes_mont) {
    u64 *c = (u64 *)malloc(8 * (n ? n : 1));
    for (size_t i = 0; i < n; i++) c[i] = from_mont(leaves_mont[i]);
    size_t total = merkle_build(c, n, nodes_mont);
    for (size_t i = 0; i < total; i++) nodes_mont[i] = to_mont(nodes_mont[i]);
    free(c);
}

/* ---------------------------------------------------------------- StdRng::seed_from_u64 + ChaCha12 */
typedef struct {
    u32 key[8];
    u64 counter;
    u32 buf[16];
    int idx;
} chacha_t;
static u32 rotl(u32 x, int n) { return (x << n) | (x >> (32 - n)); }
#define QR(a, b, c, d) \
    a += b; d ^= a; d = rotl(d, 16); c += d; b ^= c; b = rotl(b, 12); a += b; d ^= a; d = rotl(d, 8); c += d; b ^= c; b = rotl(b, 7);
static void chacha_block(const u32 key[8], u64 counter, u64 stream, int rounds, u32 out[16]) {
    u32 s[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                 (u32)counter, (u32)(counter >> 32), (u32)stream, (u32)(stream >> 32)};
    u32 x[16];
    memcpy(x, s, sizeof x);
    for (int i = 0; i < rounds; i += 2) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
void oracle_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]) {
    chacha_block(key, counter, stream, rounds, out);
}
static void rng_seed_from_u64(chacha_t *r, u64 state) { /* rand_core 0.6 SeedableRng::seed_from_u64 */
    for (int i = 0; i < 8; i++) {
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        u32 xs = (u32)(((state >> 18) ^ state) >> 27);
        u32 rot = (u32)(state >> 59);
        r->key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    r->counter = 0;
    r->idx = 16;
}
static u32 rng_u32(chacha_t *r) {
    if (r->idx == 16) {
        chacha_block(r->key, r->counter++, 0, 12, r->buf);
        r->idx = 0;
    }
    return r->buf[r->idx++];
}
static u64 rng_u64(chacha_t *r) { /* BlockRng::next_u64: low word first */
    u64 lo = rng_u32(r);
    u64 hi = rng_u32(r);
    return lo | hi << 32;
}
void oracle_stdrng_u64(uint64_t seed, size_t n, uint64_t *out) {
    chacha_t r;
    rng_seed_from_u64(&r, seed);
    for (size_t i = 0; i < n; i++) out[i] = rng_u64(&r);
}
/* Fp::rand for Goldilocks: one u64, no bits to shave, reject >= p; the value IS the Montgomery residue */
static u64 gl_rand_mont(chacha_t *r) {
    for (;;) {
        u64 v = rng_u64(r);
        if (v < GLP) return v;
    }
}
/* Fr::rand for BLS12-381 (plonk/src/challenge.rs:69-77): four u64, top limb masked to 63 bits, reject >= r */
void oracle_fr_rand_from_seed(uint64_t seed, size_t n, uint64_t *out_mont) {
    static const u64 RMOD[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    chacha_t r;
    rng_seed_from_u64(&r, seed);
    for (size_t i = 0; i < n; i++) {
        for (;;) {
            u64 t[4];
            for (int k = 0; k < 4; k++) t[k] = rng_u64(&r);
            t[3] &= ~0ull >> 1;
            int geq = 1;
            for (int k = 3; k >= 0; k--) {
                if (t[k] != RMOD[k]) {
                    geq = t[k] > RMOD[k];
                    break;
                }
            }
            if (!geq) {
                memcpy(out_mont + 4 * i, t, 32);
                break;
            }
        }
    }
}

/* ---------------------------------------------------------------- transcript.rs:30-139 */
typedef struct {
    u8 data[32];
    int has_data;
    u64 index;
} transcript_t;
static void tr_digest(transcript_t *t, u64 canon) { /* transcript.rs:64-72 */
    sha_t s;
    sha_init(&s);
    if (t->has_data) sha_update(&s, t->data, 32);
    u8 le[8];
    for (int i = 0; i < 8; i++) le[i] = (u8)(t->index >> (8 * i));
    sha_update(&s, le, 8);
    char buf[24];
    size_t l = gl_display(canon, buf);
    sha_update(&s, buf, l);
    sha_final(&s, t->data);
    t->has_data = 1;
    t->index++;
}
static void tr_new(transcript_t *t) { /* Transcript::new(F::ZERO), transcript.rs:30-40 */
    t->has_data = 0;
    t->index = 0;
    tr_digest(t, 0);
}
static void tr_rng(const transcript_t *t, chacha_t *r) { /* transcript.rs:74-84 */
    u64 seed = 0;
    for (int i = 0; i < 8; i++) seed |= (u64)t->data[i] << (8 * i);
    rng_seed_from_u64(r, seed);
}
/* The challenges both sides derive (verifier.rs:13-29): r_l after digesting root l, then the query list after
 * digesting the constant.  roots / const_val in memory form; r_out in memory form; q_out = canonical value as usize. */
void oracle_fri_challenges(const uint64_t *roots_mont, size_t layers, uint64_t const_mont, size_t nq, uint64_t *r_out_mont,
                           uint64_t *q_out) {
    transcript_t t;
    chacha_t rng;
    tr_new(&t);
    for (size_t l = 0; l < layers; l++) {
        tr_digest(&t, from_mont(roots_mont[l]));
        tr_rng(&t, &rng);
        r_out_mont[l] = gl_rand_mont(&rng);
    }
    tr_digest(&t, from_mont(const_mont));
    tr_rng(&t, &rng);
    for (size_t i = 0; i < nq; i++) q_out[i] = from_mont(gl_rand_mont(&rng));
}

/* ---------------------------------------------------------------- prover.rs:34-168 */
static size_t ilog2(size_t x) {
    size_t l = 0;
    while (x >>= 1) l++;
    return l;
}
size_t oracle_fri_proof_words(size_t domain_size, size_t nq) {
    size_t L = ilog2(domain_size);
    return 4 + L + 1 + nq * (3 * L + L * (L + 1));
}
/* Flat proof: [domain_size, layers, nq, coset] roots[layers] const_val, then per query, per layer:
 * index, eval, sym_eval, path[depth_l], sym_path[depth_l] with depth_l = log2(domain_size >> l).  Field elements in
 * memory form.  Returns the number of words, 0 when the reference would panic (zero polynomial). */
size_t oracle_fri_prove(const uint64_t *coeffs_mont, size_t d, size_t blowup, size_t nq, uint64_t *out) {
    while (d && coeffs_mont[d - 1] == 0) d--; /* DensePolynomial::from_coefficients_vec trims */
    size_t D = 1;
    while (D < d * blowup) D <<= 1; /* prover.rs:146 */
    size_t L = ilog2(D);
    u64 *poly = (u64 *)malloc(8 * (d ? d : 1));
    for (size_t i = 0; i < d; i++) poly[i] = from_mont(coeffs_mont[i]);
    u64 **evals = (u64 **)calloc(L ? L : 1, sizeof(u64 *));
    u64 **trees = (u64 **)calloc(L ? L : 1, sizeof(u64 *));
    u64 coset = 7; /* F::GENERATOR, fri/src/fields/goldilocks.rs:6 */
    transcript_t t;
    chacha_t rng;
    tr_new(&t);
    size_t dom = D, len = d;
    out[0] = D; out[1] = L; out[2] = nq; out[3] = to_mont(7);
    for (size_t l = 0; l < L; l++) { /* folding_phase, prover.rs:56-70 */
        evals[l] = (u64 *)malloc(8 * dom);
        trees[l] = (u64 *)malloc(8 * oracle_merkle_node_count(dom));
        u64 w = groot((unsigned)ilog2(dom)), root = 1;
        for (size_t i = 0; i < dom; i++) { /* fri_layer.rs:40-46: Horner at coset * w^i */
            u64 x = gmul(root, coset), acc = 0;
            for (size_t k = len; k-- > 0;) acc = gadd(gmul(acc, x), poly[k]);
            evals[l][i] = acc;
            root = gmul(root, w);
        }
        size_t total = merkle_build(evals[l], dom, trees[l]);
        u64 mroot = trees[l][total - 1];
        out[4 + l] = to_mont(mroot);
        tr_digest(&t, mroot);
        tr_rng(&t, &rng);
        u64 r = from_mont(gl_rand_mont(&rng)); /* the random limbs ARE the Montgomery residue */
        size_t nl = (len + 1) / 2;             /* fold_polynomial, prover.rs:34-42 */
        for (size_t j = 0; j < nl; j++) {
            u64 v = poly[2 * j];
            if (2 * j + 1 < len) v = gadd(v, gmul(r, poly[2 * j + 1]));
            poly[j] = v;
        }
        len = nl;
        while (len && poly[len - 1] == 0) len--;
        coset = gmul(coset, coset);
        dom /= 2;
    }
    size_t words = 0;
    if (len == 1) { /* assert_eq!(poly.len(), 1), prover.rs:72 */
        u64 cst = poly[0];
        out[4 + L] = to_mont(cst);
        tr_digest(&t, cst);
        tr_rng(&t, &rng);
        u64 *p = out + 4 + L + 1;
        for (size_t q = 0; q < nq && L; q++) { /* query_phase, prover.rs:84-134 */
            size_t ch = (size_t)(from_mont(gl_rand_mont(&rng)) % D);
            size_t ds = D;
            for (size_t l = 0; l < L; l++, ds /= 2) {
                size_t idx = ch % ds, sym = (idx + ds / 2) % ds, depth = ilog2(ds);
                *p++ = idx;
                *p++ = to_mont(evals[l][idx]);
                *p++ = to_mont(evals[l][sym]);
                for (int pass = 0; pass < 2; pass++) { /* MerkleTree::generate_proof, merkle_tree.rs:84-107 */
                    size_t cur = pass ? sym : idx, off = 0, ll = ds;
                    for (size_t i = 0; i < depth; i++) {
                        *p++ = to_mont(trees[l][off + (cur ^ 1)]);
                        off += ll;
                        ll = (ll + 1) / 2;
                        cur /= 2;
                    }
                }
            }
        }
        words = (size_t)(p - out);
    }
    for (size_t l = 0; l < L; l++) {
        free(evals[l]);
        free(trees[l]);
    }
    free(evals);
    free(trees);
    free(poly);
    return words;
}

/* verifier.rs:10-127 on the flat proof.  0 = accepted; 1 wrong index, 2 evaluation/path mismatch (n/a in the flat
 * form), 3 Merkle path, 4 folding, 5 malformed. */
int oracle_fri_verify(const uint64_t *proof, size_t words) {
    if (words < 4) return 5;
    size_t D = proof[0], L = proof[1], nq = proof[2];
    if (D == 0 || (D & (D - 1)) || ilog2(D) != L || words != oracle_fri_proof_words(D, nq)) return 5;
    u64 coset0 = from_mont(proof[3]);
    const u64 *roots = proof + 4;
    u64 cst = from_mont(proof[4 + L]);
    u64 *r = (u64 *)malloc(8 * (L ? L : 1));
    u64 *qs = (u64 *)malloc(8 * (nq ? nq : 1));
    oracle_fri_challenges(roots, L, proof[4 + L], nq, r, qs);
    const u64 *p = proof + 4 + L + 1;
    int rc = 0;
    u64 inv2 = ginv(2);
    for (size_t q = 0; q < nq && !rc && L; q++) {
        size_t ch = (size_t)(qs[q] % D), ds = D;
        u64 coset = coset0;
        for (size_t l = 0; l < L && !rc; l++, ds /= 2) {
            size_t idx = ch % ds, sym = (idx + ds / 2) % ds, depth = ilog2(ds);
            const u64 *rec = p;
            p += 3 + 2 * depth;
            if (rec[0] != idx) { rc = 1; break; }
            u64 ev = from_mont(rec[1]), sv = from_mont(rec[2]);
            for (int pass = 0; pass < 2 && !rc; pass++) { /* verify_merkle_proof, merkle_tree.rs:119-135 */
                size_t cur = pass ? sym : idx;
                u64 leaf = pass ? sv : ev;
                u64 h = hash_slice_c(&leaf, 1);
                const u64 *path = rec + 3 + pass * depth;
                for (size_t i = 0; i < depth; i++) {
                    u64 pair[2];
                    u64 nb = from_mont(path[i]);
                    if (cur % 2 == 0) { pair[0] = h; pair[1] = nb; } else { pair[0] = nb; pair[1] = h; }
                    h = hash_slice_c(pair, 2);
                    cur /= 2;
                }
                if (h != from_mont(roots[l])) rc = 3;
            }
            if (rc) break;
            /* verifier.rs:96-101: q_fold = (r + w) e / (2 w) - (r - w) s / (2 w),  w = omega^idx * coset */
            u64 w = gmul(gpow(groot((unsigned)depth), idx), coset);
            u64 rl = from_mont(r[l]);
            u64 i2w = gmul(inv2, ginv(w));
            u64 qf = gsub(gmul(gmul(gadd(rl, w), ev), i2w), gmul(gmul(gsub(rl, w), sv), i2w));
            if (l + 1 < L) {
                if (qf != from_mont(p[1])) rc = 4; /* next layer's evaluation for this query */
            } else if (qf != cst) {
                rc = 4;
            }
            coset = gmul(coset, coset);
        }
    }
    free(r);
    free(qs);
    return rc;
}
