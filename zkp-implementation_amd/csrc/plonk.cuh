// plonk.cuh -- element-wise and scan kernels of the PLONK prover rounds (plonk/src/prover.rs) on device-resident
// coefficient / evaluation vectors over Fr.  The heavy lifting (NTTs, MSMs) is in ntt.cuh / msm.cuh; the kernels here
// replace the reference's coefficient-form polynomial algebra:
//   compute_acc                 prover.rs:302-377  O(9 n^2) Horner  -> evaluations + batch inverse + prefix product
//   compute_quotient_polynomial prover.rs:381-444  12 FFT products  -> one pointwise kernel on a 4n coset
//   compute_linearisation_..    prover.rs:469-568  scalar * poly    -> one linear-combination kernel
//   poly.evaluate(z)            prover.rs:164-178                   -> chunked Horner + tree reduction
//   (p - p(z)) / (X - z)        prover.rs:243-265                   -> weighted suffix sums
// All values are arkworks Montgomery residues (saturated 8 x 32-bit, ff.cuh).
#pragma once
#include "ff.cuh"

namespace zkp {

constexpr int PK_THREADS = 256;

// -------------------------------------------------------------------------------------------------------------
// scans: out[i] = op(in[0..i]) inclusive, or exclusive with identity; three launches (chunk totals, scan of totals
// by one workgroup, apply).  CHUNK elements per thread.
// -------------------------------------------------------------------------------------------------------------
struct OpMul {
    static ZKP_DEV Fr id() { return Fr::one(); }
    static ZKP_DEV Fr op(const Fr& a, const Fr& b) { return a * b; }
};
struct OpAdd {
    static ZKP_DEV Fr id() { return Fr::zero(); }
    static ZKP_DEV Fr op(const Fr& a, const Fr& b) { return a + b; }
};
constexpr int SCAN_CHUNK = 16;

// totals[t] = op over in[t*CHUNK .. (t+1)*CHUNK)
template <class Op, bool REVERSE>
__global__ void scan_totals_kernel(const Fr* __restrict__ in, uint64_t n, Fr* __restrict__ totals, uint64_t nthreads) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nthreads) return;
    Fr acc = Op::id();
    for (int k = 0; k < SCAN_CHUNK; k++) {
        const uint64_t i = t * SCAN_CHUNK + k;
        if (i < n) acc = Op::op(acc, in[REVERSE ? n - 1 - i : i]);
    }
    totals[t] = acc;
}
// exclusive scan of `m` totals in place, single workgroup of 1024 threads (m <= 1024 * 1024)
template <class Op>
__global__ __launch_bounds__(1024) void scan_mid_kernel(Fr* __restrict__ totals, uint64_t m) {
    __shared__ Fr part[1024];
    const uint32_t tid = threadIdx.x;
    const uint64_t per = (m + 1023) / 1024;
    const uint64_t b0 = tid * per < m ? tid * per : m, b1 = b0 + per < m ? b0 + per : m;
    Fr acc = Op::id();
    for (uint64_t i = b0; i < b1; i++) acc = Op::op(acc, totals[i]);
    part[tid] = acc;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        Fr v = tid >= off ? part[tid - off] : Op::id();
        __syncthreads();
        part[tid] = Op::op(v, part[tid]);
        __syncthreads();
    }
    Fr run = tid ? part[tid - 1] : Op::id();
    for (uint64_t i = b0; i < b1; i++) {
        const Fr v = totals[i];
        totals[i] = run;
        run = Op::op(run, v);
    }
}
// out[i] = prefix (EXCLUSIVE ? before : through) element i; REVERSE scans from the top index down (suffix scan)
template <class Op, bool REVERSE, bool EXCLUSIVE>
__global__ void scan_apply_kernel(const Fr* __restrict__ in, uint64_t n, const Fr* __restrict__ totals, Fr* __restrict__ out,
                                  uint64_t nthreads) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nthreads) return;
    Fr acc = totals[t];
    for (int k = 0; k < SCAN_CHUNK; k++) {
        const uint64_t i = t * SCAN_CHUNK + k;
        if (i >= n) break;
        const uint64_t idx = REVERSE ? n - 1 - i : i;
        const Fr v = in[idx];
        if (EXCLUSIVE) {
            out[idx] = acc;
            acc = Op::op(acc, v);
        } else {
            acc = Op::op(acc, v);
            out[idx] = acc;
        }
    }
}

// -------------------------------------------------------------------------------------------------------------
// a^-1 by Fermat (a^(r-2)); 0 -> 0
// -------------------------------------------------------------------------------------------------------------
ZKP_DEV Fr fr_inverse(const Fr& a) {
    Fr r = Fr::one(), b = a;
    // r - 2 = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfefffffffeffffffff
    const uint32_t ex[8] = {0xffffffffu, 0xfffffffeu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll 1
        for (int k = 0; k < 32; k++) {
            if ((ex[i] >> k) & 1) r = r * b;
            b = sqr(b);
        }
    }
    return r;
}

// -------------------------------------------------------------------------------------------------------------
// Round 2 (prover.rs:302-377): ratio[i] = num(w^i) / den(w^i) from the six evaluation vectors
//   num = (a + beta w^i + gamma)(b + beta k1 w^i + gamma)(c + beta k2 w^i + gamma)
//   den = (a + beta s1 + gamma)(b + beta s2 + gamma)(c + beta s3 + gamma)
// -------------------------------------------------------------------------------------------------------------
struct AccParams {
    const Fr* a; const Fr* b; const Fr* c; const Fr* s1; const Fr* s2; const Fr* s3;
    Fr beta, gamma, k1, k2, omega;
    uint64_t n;
};
// No inversion per point (a Fermat inverse is a 380-product dependency chain: 0.5 ms whatever the parallelism).  With
// N_i = prod_{j<i} num_j, S_i = prod_{j>=i} den_j and T = prod_j den_j:  prod_{j<i} num_j / den_j = N_i * S_i / T,
// i.e. two scans and ONE inversion (on the host).
__global__ __launch_bounds__(PK_THREADS) void plonk_acc_numden_kernel(AccParams p, Fr* __restrict__ num_out,
                                                                     Fr* __restrict__ den_out) {
    const uint64_t i = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x;
    if (i >= p.n) return;
    const Fr wi = pow_u64(p.omega, i);
    const Fr bw = p.beta * wi;
    const Fr a = p.a[i], b = p.b[i], c = p.c[i];
    num_out[i] = (a + bw + p.gamma) * (b + bw * p.k1 + p.gamma) * (c + bw * p.k2 + p.gamma);
    den_out[i] = (a + p.beta * p.s1[i] + p.gamma) * (b + p.beta * p.s2[i] + p.gamma) * (c + p.beta * p.s3[i] + p.gamma);
}
// acc[i] = nprefix[i] * dsuffix[i] * inv_total
__global__ __launch_bounds__(PK_THREADS) void plonk_acc_combine_kernel(const Fr* __restrict__ nprefix, const Fr* __restrict__ dsuffix,
                                                                      Fr inv_total, uint64_t n, Fr* __restrict__ acc) {
    const uint64_t i = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x;
    if (i < n) acc[i] = nprefix[i] * dsuffix[i] * inv_total;
}

// -------------------------------------------------------------------------------------------------------------
// Round 3 (prover.rs:381-444) on the coset g<w_D>, D = 4n (8n for n < 8).  ev = 16 evaluation vectors of length D:
//   0 ax 1 bx 2 cx 3 z 4 q_m 5 q_l 6 q_r 7 q_o 8 q_c 9 pi 10 s1 11 s2 12 s3 13 L1 14 X   (z(wX) = z shifted by D/n)
//   t = [ line1 + alpha (line2 - line3) + alpha^2 (z - 1) L1 ] / Z_H,   Z_H(g w_D^j) takes D/n distinct values
// -------------------------------------------------------------------------------------------------------------
struct QuotParams {
    const Fr* ev;       // 15 x D
    uint64_t D;
    uint32_t shift;     // D / n
    Fr beta, gamma, alpha, alpha2, k1, k2;
    Fr zh_inv[8];       // 1 / Z_H on the coset, index j mod (D/n)
};
__global__ __launch_bounds__(PK_THREADS) void plonk_quotient_kernel(QuotParams p, Fr* __restrict__ t_ev) {
    const uint64_t j = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x;
    if (j >= p.D) return;
    const uint64_t D = p.D;
    const Fr a = p.ev[0 * D + j], b = p.ev[1 * D + j], c = p.ev[2 * D + j], z = p.ev[3 * D + j];
    const Fr zw = p.ev[3 * D + ((j + p.shift) & (D - 1))];
    const Fr x = p.ev[14 * D + j];
    // line 1: gate constraint
    Fr l1 = a * b * p.ev[4 * D + j] + a * p.ev[5 * D + j] + b * p.ev[6 * D + j] + c * p.ev[7 * D + j] + p.ev[9 * D + j] +
            p.ev[8 * D + j];
    // line 2 / 3: permutation argument
    const Fr bx = p.beta * x;
    Fr l2 = (a + bx + p.gamma) * (b + bx * p.k1 + p.gamma) * (c + bx * p.k2 + p.gamma) * z;
    Fr l3 = (a + p.beta * p.ev[10 * D + j] + p.gamma) * (b + p.beta * p.ev[11 * D + j] + p.gamma) *
            (c + p.beta * p.ev[12 * D + j] + p.gamma) * zw;
    // line 4: z(1) = 1
    Fr l4 = (z - Fr::one()) * p.ev[13 * D + j];
    Fr num = l1 + (l2 - l3) * p.alpha + l4 * p.alpha2;
    // zh_inv is a small per-lane-indexed table: select with static indexing to keep it in registers/SGPRs
    const uint32_t k = (uint32_t)(j & (p.shift - 1));
    Fr zi = p.zh_inv[0];
#pragma unroll
    for (int q = 1; q < 8; q++)
        if (k == (uint32_t)q) zi = p.zh_inv[q];
    t_ev[j] = num * zi;
}

// -------------------------------------------------------------------------------------------------------------
// out[i] = sum_k s_k * p_k[i]   (i < n_out; p_k[i] = 0 for i >= len_k)
// -------------------------------------------------------------------------------------------------------------
constexpr int LINCOMB_MAX = 12;
struct LincombParams {
    const Fr* p[LINCOMB_MAX];
    uint64_t len[LINCOMB_MAX];
    Fr s[LINCOMB_MAX];
    int terms;
    uint64_t n_out;
};
__global__ __launch_bounds__(PK_THREADS) void fr_lincomb_kernel(LincombParams L, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x;
    if (i >= L.n_out) return;
    Fr acc = Fr::zero();
#pragma unroll
    for (int k = 0; k < LINCOMB_MAX; k++) {
        if (k < L.terms && i < L.len[k]) acc = acc + L.s[k] * L.p[k][i];
    }
    out[i] = acc;
}

// -------------------------------------------------------------------------------------------------------------
// Polynomial evaluation, several (polynomial, point) pairs per launch (blockIdx.y = request).  A workgroup covers
// 256 x EVAL_CHUNK coefficients: every thread runs Horner over its EVAL_CHUNK coefficients, then the workgroup folds the 256
// values pairwise with the host-supplied constants zp[k] = z^(EVAL_CHUNK 2^k):  partial[b] = sum_{i in block} c[i] z^(i - lo_b).
// The host finishes with a Horner over the blocks (z^(256 EVAL_CHUNK) = zp[8]).  No exponentiation on the device: the
// longest dependency chain is EVAL_CHUNK + 8 products (was 32 + a 16-bit power).
// -------------------------------------------------------------------------------------------------------------
constexpr int EVAL_CHUNK = 8;
constexpr int EVAL_MAX_REQ = 8;
struct EvalParams {
    const Fr* c[EVAL_MAX_REQ];
    uint64_t len[EVAL_MAX_REQ];
    uint32_t part_off[EVAL_MAX_REQ];  // first partial of request r
    Fr z[EVAL_MAX_REQ];
    Fr zp[EVAL_MAX_REQ][8];           // z^(EVAL_CHUNK 2^k), k < 8
};
__global__ __launch_bounds__(PK_THREADS) void fr_poly_eval_kernel(EvalParams p, Fr* __restrict__ partial) {
    static_assert(PK_THREADS == 256, "eight folding levels");
    __shared__ Fr red[PK_THREADS];
    const uint32_t r = blockIdx.y;
    const uint64_t n = p.len[r];
    const uint64_t lo = ((uint64_t)blockIdx.x * PK_THREADS + threadIdx.x) * EVAL_CHUNK;
    if ((uint64_t)blockIdx.x * PK_THREADS * EVAL_CHUNK >= n) return;  // whole workgroup past the end
    const Fr* c = p.c[r];
    const Fr z = p.z[r];
    Fr acc = Fr::zero();
    if (lo < n) {
        const uint64_t hi = lo + EVAL_CHUNK < n ? lo + EVAL_CHUNK : n;
        for (uint64_t i = hi; i-- > lo;) acc = acc * z + c[i];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
        const int span = 1 << k;
        if ((threadIdx.x & (2 * span - 1)) == 0) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + span] * p.zp[r][k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[p.part_off[r] + blockIdx.x] = red[0];
}

// out[i] = in[i] * base^(i + e0), EVAL_CHUNK consecutive elements per thread.  The thread's first power base^(lo + e0) is the
// product of the host-supplied constants bp[k] = base^(EVAL_CHUNK 2^k) over the set bits of lo / EVAL_CHUNK, times c0 = base^e0:
// at most log2(n / EVAL_CHUNK) + EVAL_CHUNK dependent products and no squarings (was a 16-bit power + 32 products).
// Used for the division by (X - z):  (p(X) - p(z)) / (X - z) has coefficients q_j = z^-(j+1) * sum_{i>j} c_i z^i.
struct ScalePowParams {
    Fr base, c0;
    Fr bp[30];  // covers n < 2^33
};
__global__ __launch_bounds__(PK_THREADS) void fr_scale_pow_kernel(const Fr* __restrict__ in, uint64_t n, ScalePowParams p,
                                                                 Fr* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x;
    const uint64_t lo = t * EVAL_CHUNK;
    if (lo >= n) return;
    const uint64_t hi = lo + EVAL_CHUNK < n ? lo + EVAL_CHUNK : n;
    Fr pw = p.c0;
#pragma unroll 1
    for (int k = 0; (t >> k) != 0; k++)
        if ((t >> k) & 1) pw = pw * p.bp[k];
    for (uint64_t i = lo; i < hi; i++) {
        out[i] = in[i] * pw;
        pw = pw * p.base;
    }
}

// highest index with a non-zero coefficient, +1 (atomicMax into *len, which the caller zeroes)
__global__ __launch_bounds__(PK_THREADS) void fr_trim_len_kernel(const Fr* __restrict__ c, uint64_t n, unsigned long long* len) {
    const uint64_t i = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x;
    const bool nz = i < n && !c[i].is_zero();
    const unsigned long long mask = __ballot(nz);  // one atomic per wave, from its highest non-zero lane (2^18 atomics on one
    if (mask == 0) return;                          // address took 39 us)
    const int top = 63 - __builtin_clzll(mask);
    if ((int)(threadIdx.x & 63) == top) atomicMax(len, (unsigned long long)(i + 1));
}

}  // namespace zkp
