/*
 * zkp_oracle.h -- CPU restatement of the reference's MSM / NTT hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (zkp-implementation_amd/) never links, imports or calls it.
 *
 * Parity status: the Rust reference cannot be built here (no cargo/rustc, arkworks 0.4.x not vendored),
 * so this restatement is pinned by (1) the reference's own known-answer tests (kzg/src/commitment.rs:36-53,
 * fri/src/prover.rs:181-205, plonk/src/slice_polynomial.rs:80-111) and (2) the golden vectors in
 * tests/golden/vectors.json produced by the independent big-int model tests/model/bigmodel.py.
 * NTT outputs, quotient polynomials and FRI layer values are NOT pinned by any reference test
 * ("parity unpinned by the reference" for those; pinned by the mathematical definition instead).
 *
 * Memory formats = arkworks 0.4 in-memory forms: Montgomery residues, little-endian u64 limbs.
 *   Fr: 4 limbs (R = 2^256)   Fq: 6 limbs (R = 2^384)   Goldilocks: 1 limb (R = 2^64)
 *   G1 affine: x||y = 12 limbs, infinity carried in a separate byte array (1 = infinity).
 */
#ifndef ZKP_ORACLE_H
#define ZKP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- field helpers (batch, in place allowed) ---- */
void oracle_fr_to_mont(const uint64_t *in, uint64_t *out, size_t n);
void oracle_fr_from_mont(const uint64_t *in, uint64_t *out, size_t n);
void oracle_fq_to_mont(const uint64_t *in, uint64_t *out, size_t n);
void oracle_fq_from_mont(const uint64_t *in, uint64_t *out, size_t n);
void oracle_gl_to_mont(const uint64_t *in, uint64_t *out, size_t n);
void oracle_gl_from_mont(const uint64_t *in, uint64_t *out, size_t n);
void oracle_fr_mul(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void oracle_fr_add(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void oracle_fr_sub(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void oracle_fr_inv(const uint64_t *a, uint64_t *out, size_t n);
void oracle_fq_mul(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
/* sum_i a_i * b_i (Montgomery in, Montgomery out) */
void oracle_fr_inner_product(const uint64_t *a, const uint64_t *b, size_t n, uint64_t out[4]);
/* deterministic inputs: SplitMix64 stream, rejection sampled, output in Montgomery form */
void oracle_rand_fr(uint64_t seed, size_t n, uint64_t *out);
void oracle_rand_gl(uint64_t seed, size_t n, uint64_t *out);

/* ---- G1 ---- */
/* base * scalar, MSB-first double-and-add then into_affine (kzg/src/scheme.rs:80, 92) */
void oracle_g1_mul(const uint64_t base_xy[12], uint8_t base_inf, const uint64_t scalar[4],
                   uint64_t out_xy[12], uint8_t *out_inf);
void oracle_g1_add(const uint64_t a_xy[12], uint8_t a_inf, const uint64_t b_xy[12], uint8_t b_inf,
                   uint64_t out_xy[12], uint8_t *out_inf);
void oracle_g1_generator(uint64_t out_xy[12]);
int oracle_g1_on_curve(const uint64_t xy[12], uint8_t inf);
/* Srs::new_from_secret (kzg/src/srs.rs:48-63): n points [s^i]G, i < n, by n sequential scalar-muls */
void oracle_srs(const uint64_t secret[4], size_t n, uint64_t *out_xy);
/* P_i = k_i * G via a fixed-base 8-bit window table (fast input generator; same output as g1_mul) */
void oracle_g1_fixed_base_mul(const uint64_t *scalars, size_t n, uint64_t *out_xy, uint8_t *out_inf);

/* evaluate_in_s exactly as kzg/src/scheme.rs:88-94: per-term scalar-mul -> affine, left fold of
 * affine adds each normalised; n = min(len(scalars), len(points)) is the caller's zip truncation. */
void oracle_msm_naive(const uint64_t *points_xy, const uint8_t *points_inf, const uint64_t *scalars,
                      size_t n, uint64_t out_xy[12], uint8_t *out_inf);
/* CPU Pippenger -- NOT the reference's algorithm; same group element, used as a fast checker at
 * sizes where the naive path takes minutes. */
void oracle_msm_pippenger(const uint64_t *points_xy, const uint8_t *points_inf, const uint64_t *scalars,
                          size_t n, uint64_t out_xy[12], uint8_t *out_inf);

/* ---- NTT (ark-poly 0.4 Radix2EvaluationDomain semantics: natural order in/out, ifft scales by 1/n,
 *      coset fft scales coefficient j by g^j first, coset ifft scales output j by g^-j) ---- */
void oracle_ntt_fr(uint64_t *data, unsigned log_n, int inverse, const uint64_t *coset /* nullable */);
/* CPU-best CONTEXT baseline: the same bucket method / radix-2 stages on `threads` cores with OpenMP.
 * Same outputs as the single-threaded functions; not the reference's algorithm and not the parity oracle. */
int oracle_max_threads(void);
void oracle_msm_pippenger_mt(const uint64_t *points_xy, const uint8_t *points_inf, const uint64_t *scalars, size_t n, int threads,
                             uint64_t out_xy[12], uint8_t *out_inf);
void oracle_ntt_fr_mt(uint64_t *data, unsigned log_n, int inverse, const uint64_t *coset /* nullable */, int threads);
void oracle_ntt_gl(uint64_t *data, unsigned log_n, int inverse, const uint64_t *coset /* nullable */);
void oracle_fr_root_of_unity(unsigned log_n, uint64_t out[4]);
void oracle_gl_root_of_unity(unsigned log_n, uint64_t out[1]);

/* ---- polynomial helpers with ark-poly semantics (plonk/src/prover.rs:396-455) ---- */
/* out has la+lb-1 entries (0 if either is empty); schoolbook */
void oracle_poly_mul_fr(const uint64_t *a, size_t la, const uint64_t *b, size_t lb, uint64_t *out);
/* quotient gets len-n entries, remainder n entries; returns 0 */
int oracle_divide_by_vanishing_fr(const uint64_t *c, size_t len, size_t n, uint64_t *quot, uint64_t *rem);
void oracle_poly_eval_fr(const uint64_t *c, size_t len, const uint64_t z[4], uint64_t out[4]);
/* (p(X)-p(z))/(X-z): quotient has len-1 entries (kzg/src/scheme.rs:110-118) */
void oracle_poly_div_linear_fr(const uint64_t *c, size_t len, const uint64_t z[4], uint64_t *quot);

/* ---- FRI (Goldilocks) ---- */
/* FriLayer::from_poly evaluation loop, fri/src/fri_layer.rs:40-46: D Horner evaluations */
void oracle_fri_layer_eval(const uint64_t *coeffs, size_t d, uint64_t coset, unsigned log_D, uint64_t *out);
/* fold_polynomial, fri/src/prover.rs:34-42; out has ceil(d/2) entries */
void oracle_fri_fold(const uint64_t *coeffs, size_t d, uint64_t r, uint64_t *out);


/* ---- FRI commitment path around the NTT (fri_oracle.c; third-party behaviour listed there is parity-unpinned) ---- */
void oracle_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
/* hash(&F) element-wise, fri/src/hasher.rs:14-19; hash_slice, hasher.rs:30-35 */
void oracle_gl_hash(const uint64_t *in_mont, size_t n, uint64_t *out_mont);
void oracle_gl_hash_slice(const uint64_t *in_mont, size_t n, uint64_t out_mont[1]);
/* MerkleTree::new, fri/src/merkle_tree.rs:42-63: all levels concatenated (level 0 = n leaf hashes, then ceil halves) */
size_t oracle_merkle_node_count(size_t n);
void oracle_merkle_tree(const uint64_t *leaves_mont, size_t n, uint64_t *nodes_mont);
/* rand_chacha block function (rounds = 12 for StdRng, 20 for the RFC 8439 vector) and StdRng::seed_from_u64 output */
void oracle_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]);
void oracle_stdrng_u64(uint64_t seed, size_t n, uint64_t *out);
void oracle_fr_rand_from_seed(uint64_t seed, size_t n, uint64_t *out_mont); /* plonk/src/challenge.rs:69-77 */
/* Transcript replay, fri/src/verifier.rs:13-29 */
void oracle_fri_challenges(const uint64_t *roots_mont, size_t layers, uint64_t const_mont, size_t nq, uint64_t *r_out_mont,
                           uint64_t *q_out);
/* generate_proof / verify, fri/src/prover.rs:141-168 and fri/src/verifier.rs:10-127, on a flat proof (layout in fri_oracle.c) */
size_t oracle_fri_proof_words(size_t domain_size, size_t nq);
size_t oracle_fri_prove(const uint64_t *coeffs_mont, size_t d, size_t blowup, size_t nq, uint64_t *out);
int oracle_fri_verify(const uint64_t *proof, size_t words);

#ifdef __cplusplus
}
#endif
#endif
