// kzg_host.hpp -- C++ mirror of the reference's KzgScheme (kzg/src/scheme.rs:22-142) over the C ABI.
//
// Same names, argument meaning and error behaviour as the Rust surface; the body of evaluate_in_s is the
// one place that differs: it is a single call into the HIP MSM instead of n scalar multiplications.
// The O(n) polynomial bookkeeping of `open` (Horner evaluation, synthetic division by X - z) stays on the
// host exactly where the reference has it (scheme.rs:110-118).
#pragma once
#include <vector>

#include "../../include/zkp_hip.h"
#include "host_ff.hpp"

namespace zkp {
namespace host {

struct G1Point {  // kzg/src/types.rs:6 -- affine, canonical
    uint64_t xy[12];
    uint8_t infinity;
};
struct KzgCommitment { G1Point p; };           // kzg/src/commitment.rs:5
struct KzgOpening { G1Point p; HFr eval; };    // kzg/src/opening.rs:12

// DensePolynomial::from_coefficients_vec trims trailing zero coefficients (ark-poly 0.4)
inline size_t trimmed_len(const uint64_t* coeffs, size_t len) {
    while (len && (coeffs[4 * (len - 1)] | coeffs[4 * (len - 1) + 1] | coeffs[4 * (len - 1) + 2] | coeffs[4 * (len - 1) + 3]) == 0) len--;
    return len;
}

class KzgScheme {
  public:
    explicit KzgScheme(const zkp_bases* srs_g1) : srs_(srs_g1) {}  // KzgScheme::new, scheme.rs:34

    // scheme.rs:49-52 / 63-67
    int commit(const uint64_t* coeffs, size_t len, KzgCommitment* out) const { return evaluate_in_s(coeffs, len, &out->p); }

    // scheme.rs:78-82: para * g1_points[0]
    int commit_para(const uint64_t para[4], const uint64_t g1_0[12], KzgCommitment* out) const {
        return zkp_g1_mul(g1_0, 0, para, out->p.xy, &out->p.infinity);
    }

    // scheme.rs:108-120 / 132-142
    int open(const uint64_t* coeffs, size_t len, const uint64_t z_limbs[4], KzgOpening* out) const {
        if (len == 0) return ZKP_E_ARG;  // `.expect("at least 1")`, scheme.rs:112
        const HFr z = HFr::load(z_limbs);
        // evaluation_at_z = polynomial.evaluate(&z); quotient of (p - p(z)) by (X - z): one Horner sweep gives both
        std::vector<uint64_t> q(4 * (len - 1) + 4);
        HFr acc = HFr::zero();
        for (size_t i = len; i-- > 1;) {
            acc = acc * z + HFr::load(coeffs + 4 * i);
            acc.store(&q[4 * (i - 1)]);
        }
        out->eval = acc * z + HFr::load(coeffs);
        return evaluate_in_s(q.data(), len - 1, &out->p);
    }

  private:
    // scheme.rs:84-96
    int evaluate_in_s(const uint64_t* coeffs, size_t len, G1Point* out) const {
        len = trimmed_len(coeffs, len);
        // assert!(g1_points.len() > polynomial.degree()) -- degree() of the zero polynomial is 0
        const size_t have = zkp_g1_bases_len(srs_);
        if (have == 0 || len > have) return ZKP_E_SIZE;
        return zkp_msm_g1(srs_, coeffs, len, out->xy, &out->infinity);
    }
    const zkp_bases* srs_;
};

}  // namespace host
}  // namespace zkp
