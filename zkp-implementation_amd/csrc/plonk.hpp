// plonk.hpp -- element-wise and scan kernels of the PLONK prover rounds (plonk/src/prover.rs) on device-resident
// coefficient / evaluation vectors over Fr.  The heavy lifting (NTTs, MSMs) is in ntt.hpp / msm.hpp; the kernels here
// replace the reference's coefficient-form polynomial algebra:
//   compute_acc                 prover.rs:302-377  O(9 n^2) Horner  -> evaluations + batch inverse + prefix product
//   compute_quotient_polynomial prover.rs:381-444  12 FFT products  -> one pointwise kernel on a 4n coset
//   compute_linearisation_..    prover.rs:469-568  scalar * poly    -> one linear-combination kernel
//   poly.evaluate(z)            prover.rs:164-178                   -> chunked Horner + tree reduction
//   (p - p(z)) / (X - z)        prover.rs:243-265                   -> weighted suffix sums
// All values are arkworks Montgomery residues (saturated 8 x 32-bit, ff.hpp).
#pragma once
#include "ff.hpp"

namespace zkp {

constexpr int PK_THREADS = 256;

// -------------------------------------------------------------------------------------------------------------
// scans: out[i] = op(in[0..i]) inclusive, or exclusive with identity, optionally from the top index down (suffix scan); up to
// SCAN_MAX_BATCH independent scans of one operator per launch (blockIdx.y = job: the numerator prefix and the denominator suffix of
// round 2, the two divisions of round 5).  Three launches: (1) `chunk` elements per thread, then the 256 thread totals of a
// workgroup are scanned in place (wave shuffles + four wave totals through LDS): every thread leaves its exclusive prefix inside the
// workgroup, the workgroup its total; (2) one workgroup per job scans the workgroup totals; (3) apply.  At PLONK's 2^16 elements
// these kernels are chains of dependent field products over a nearly empty machine, so the host picks the chunk that keeps about
// 2^16 threads (4 elements per thread there: 5 + 8, ~8, 5 dependent products in the three launches; the first version -- 16 elements
// per thread, one workgroup of 1024 threads over 4096 totals -- had 16, 18, 16 and took 103 us per scan against ~35 now).
// -------------------------------------------------------------------------------------------------------------
struct OpMul {
    static ZKP_DEV Fr id() { return Fr::one(); }
    static ZKP_DEV Fr op(const Fr& a, const Fr& b) { return a * b; }
};
struct OpAdd {
    static ZKP_DEV Fr id() { return Fr::zero(); }
    static ZKP_DEV Fr op(const Fr& a, const Fr& b) { return a + b; }
};
constexpr int SCAN_MAX_BATCH = 2;
constexpr int SCAN_MAX_CHUNK = 16;
struct ScanJob {
    const Fr* in;
    Fr* out;
    uint64_t n;
    uint32_t rev, excl;
};
struct ScanParams {
    ScanJob job[SCAN_MAX_BATCH];
    uint32_t chunk;    // elements per thread
    uint32_t blocks;   // workgroups per job (of the longest job)
    Fr* tpre;          // [job][blocks * 256] exclusive prefix of every thread inside its workgroup
    Fr* btot;          // [job][blocks] workgroup totals, then (after scan_mid) their exclusive prefixes
};
ZKP_HD uint32_t scan_chunk_for(uint64_t n) {  // about 2^16 threads, 1..16 elements each
    uint64_t c = n >> 16;
    return (uint32_t)(c < 1 ? 1 : c > SCAN_MAX_CHUNK ? SCAN_MAX_CHUNK : c);
}
ZKP_HD uint64_t scan_blocks_for(uint64_t n) {
    const uint64_t threads = (n + scan_chunk_for(n) - 1) / scan_chunk_for(n);
    return (threads + 255) / 256;
}
// scratch (tpre + btot of a batch) that covers every scan of at most n elements: below 2^20 elements the chunk rule keeps the thread
// count under 2^17 (512 workgroups), above it a thread takes 16 elements
ZKP_HD uint64_t scan_scratch_elems(uint64_t n) {
    const uint64_t blocks = (n >> 12) > 512 ? (n >> 12) : 512;
    return SCAN_MAX_BATCH * (blocks + 2) * 257;
}

ZKP_DEV Fr fr_shfl_up(const Fr& x, int off) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = (uint32_t)__shfl_up((int)x.l[i], off, 64);
    return r;
}
ZKP_DEV Fr fr_shfl(const Fr& x, int lane) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = (uint32_t)__shfl((int)x.l[i], lane, 64);
    return r;
}
// exclusive scan over the 256 threads of a workgroup (every thread must call it); *total = op over all of them
template <class Op>
ZKP_DEV Fr block_exclusive_scan(const Fr& mine, Fr* total) {
    __shared__ Fr wave_tot[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    Fr x = mine;  // inclusive scan inside the wave
#pragma unroll 1
    for (int off = 1; off < 64; off <<= 1) {
        const Fr v = fr_shfl_up(x, off);
        if (lane >= off) x = Op::op(v, x);
    }
    if (lane == 63) wave_tot[wv] = x;
    Fr ex = fr_shfl_up(x, 1);
    if (lane == 0) ex = Op::id();
    __syncthreads();
    Fr before = Op::id();
    Fr all = wave_tot[0];
#pragma unroll
    for (int w = 1; w < 4; w++) {
        if (w == wv) before = all;
        all = Op::op(all, wave_tot[w]);
    }
    *total = all;
    __syncthreads();  // wave_tot may be reused by a later call
    return wv ? Op::op(before, ex) : ex;
}
template <class Op>
__global__ __launch_bounds__(256) void scan_block_kernel(ScanParams p) {
    const ScanJob jb = p.job[blockIdx.y];
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if ((uint64_t)blockIdx.x * 256 * p.chunk >= jb.n) return;  // whole workgroup past the end of this job
    Fr acc = Op::id();
    for (uint32_t k = 0; k < p.chunk; k++) {
        const uint64_t i = t * p.chunk + k;
        if (i < jb.n) acc = Op::op(acc, jb.in[jb.rev ? jb.n - 1 - i : i]);
    }
    Fr total;
    const Fr ex = block_exclusive_scan<Op>(acc, &total);
    p.tpre[((uint64_t)blockIdx.y * p.blocks + blockIdx.x) * 256 + threadIdx.x] = ex;
    if (threadIdx.x == 0) p.btot[(uint64_t)blockIdx.y * p.blocks + blockIdx.x] = total;
}
// exclusive scan of the workgroup totals of one job, in place; one workgroup of 256 threads per job
template <class Op>
__global__ __launch_bounds__(256) void scan_mid_kernel(ScanParams p) {
    const ScanJob jb = p.job[blockIdx.y];
    const uint64_t threads = (jb.n + p.chunk - 1) / p.chunk, m = (threads + 255) / 256;
    Fr* tot = p.btot + (uint64_t)blockIdx.y * p.blocks;
    const uint64_t per = (m + 255) / 256;
    const uint64_t b0 = threadIdx.x * per < m ? threadIdx.x * per : m, b1 = b0 + per < m ? b0 + per : m;
    Fr acc = Op::id();
    for (uint64_t i = b0; i < b1; i++) acc = Op::op(acc, tot[i]);
    Fr total;
    Fr run = block_exclusive_scan<Op>(acc, &total);
    for (uint64_t i = b0; i < b1; i++) {
        const Fr v = tot[i];
        tot[i] = run;
        run = Op::op(run, v);
    }
}
template <class Op>
__global__ __launch_bounds__(256) void scan_apply_kernel(ScanParams p) {
    const ScanJob jb = p.job[blockIdx.y];
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t * p.chunk >= jb.n) return;
    Fr acc = Op::op(p.btot[(uint64_t)blockIdx.y * p.blocks + blockIdx.x],
                    p.tpre[((uint64_t)blockIdx.y * p.blocks + blockIdx.x) * 256 + threadIdx.x]);
    for (uint32_t k = 0; k < p.chunk; k++) {
        const uint64_t i = t * p.chunk + k;
        if (i >= jb.n) break;
        const uint64_t idx = jb.rev ? jb.n - 1 - i : i;
        const Fr v = jb.in[idx];
        if (jb.excl) {
            jb.out[idx] = acc;
            acc = Op::op(acc, v);
        } else {
            acc = Op::op(acc, v);
            jb.out[idx] = acc;
        }
    }
}

// Blinded working polynomials of rounds 1 and 2 in one launch (blockIdx.y = polynomial): dst[0..cap) = src[0..n) | zeros, then
// + blind(X) (X^n - 1): dst[i] -= c[i], dst[n + i] += c[i], i < len   (mul_by_vanishing_poly, prover.rs:83-89,109)
struct BlindParams {
    Fr* dst[3];
    const Fr* src[3];
    Fr c[3][3];
    uint32_t len;
    uint64_t n, cap;
};
__global__ __launch_bounds__(PK_THREADS) void plonk_blind_kernel(BlindParams p) {
    const uint64_t i = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x;
    if (i >= p.cap) return;
    const uint32_t w = blockIdx.y;
    Fr v = i < p.n ? p.src[w][i] : Fr::zero();
    if (i < p.len) v = v - p.c[w][i];
    if (i >= p.n && i - p.n < p.len) v = v + p.c[w][i - p.n];
    p.dst[w][i] = v;
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
    Fr beta, gamma, k1, k2;
    Fr wp[32];  // omega^(2^k): omega^i is the product over the set bits of i (no squarings on the device)
    uint64_t n;
};
// No inversion per point (a Fermat inverse is a 380-product dependency chain: 0.5 ms whatever the parallelism).  With
// N_i = prod_{j<i} num_j, S_i = prod_{j>=i} den_j and T = prod_j den_j:  prod_{j<i} num_j / den_j = N_i * S_i / T,
// i.e. two scans and ONE inversion (on the host).
__global__ __launch_bounds__(PK_THREADS) void plonk_acc_numden_kernel(AccParams p, Fr* __restrict__ num_out,
                                                                     Fr* __restrict__ den_out) {
    const uint64_t i = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x;
    if (i >= p.n) return;
    Fr wi = Fr::one();
#pragma unroll 1
    for (int k = 0; (i >> k) != 0; k++)
        if ((i >> k) & 1) wi = wi * p.wp[k];
    const Fr bw = p.beta * wi;
    const Fr a = p.a[i], b = p.b[i], c = p.c[i];
    num_out[i] = (a + bw + p.gamma) * (b + bw * p.k1 + p.gamma) * (c + bw * p.k2 + p.gamma);
    den_out[i] = (a + p.beta * p.s1[i] + p.gamma) * (b + p.beta * p.s2[i] + p.gamma) * (c + p.beta * p.s3[i] + p.gamma);
}
// Gate equations on the domain (the divisibility of line 1 of compute_quotient_polynomial, prover.rs:396-404: the blinded
// a, b, c agree with f_a, f_b, f_c on H, so "No remainder expected" there <=> every row satisfies its gate):
//   q_m a b + q_l a + q_r b + q_o c + pi + q_c == 0 at w^i, i < n.   de = 12 x n domain evaluations in circuit order
//   (q_m q_l q_r q_o q_c pi f_a f_b f_c s1 s2 s3); *violations counts the rows that fail (one atomic per failing wave).
__global__ __launch_bounds__(PK_THREADS) void plonk_gate_check_kernel(const Fr* __restrict__ de, uint64_t n,
                                                                     unsigned long long* __restrict__ violations) {
    const uint64_t i = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x;
    bool bad = false;
    if (i < n) {
        const Fr a = de[6 * n + i], b = de[7 * n + i], c = de[8 * n + i];
        const Fr g = a * b * de[0 * n + i] + a * de[1 * n + i] + b * de[2 * n + i] + c * de[3 * n + i] + de[5 * n + i] + de[4 * n + i];
        bad = !g.is_zero();
    }
    const unsigned long long m = __ballot(bad);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(violations, (unsigned long long)__builtin_popcountll(m));
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
    const Fr* ev;       // 15 x D: slots 4..14 (q_m q_l q_r q_o q_c pi s1 s2 s3 L1 X on the coset)
    const Fr* w;        // 4 x D: a, b, c, z on the coset (a buffer of their own: computed early, plonk_host.inc)
    uint64_t D;
    uint32_t shift;     // D / n
    Fr beta, gamma, alpha, alpha2, k1, k2;
    Fr zh_inv[8];       // 1 / Z_H on the coset, index j mod (D/n)
};
__global__ __launch_bounds__(PK_THREADS) void plonk_quotient_kernel(QuotParams p, Fr* __restrict__ t_ev) {
    const uint64_t j = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x;
    if (j >= p.D) return;
    const uint64_t D = p.D;
    const Fr a = p.w[0 * D + j], b = p.w[1 * D + j], c = p.w[2 * D + j], z = p.w[3 * D + j];
    const Fr zw = p.w[3 * D + ((j + p.shift) & (D - 1))];
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

// out[i] = in[i] * base^(i + e0), `chunk` consecutive elements per thread, one or two jobs per launch (blockIdx.y).  The thread's
// first power base^(lo + e0) is the product of the host-supplied constants bp[k] = base^(chunk 2^k) over the set bits of lo / chunk,
// times c0 = base^e0: at most log2(n / chunk) + 2 chunk dependent products and no squarings (was a 16-bit power + 32 products).
// Used for the division by (X - z):  (p(X) - p(z)) / (X - z) has coefficients q_j = z^-(j+1) * sum_{i>j} c_i z^i.
struct ScalePowParams {
    Fr base, c0;
    Fr bp[30];  // base^(chunk 2^k): covers n < 2^30 at chunk 1
};
struct ScalePowJob {
    const Fr* in;
    Fr* out;
    uint64_t n;
};
struct ScalePowBatch {
    ScalePowJob job[2];
    ScalePowParams sp[2];
    uint32_t chunk;  // consecutive elements per thread (the host keeps about 2^16 threads: a thread is a chain of
};                   // log2(threads) + 2 chunk dependent products)
__global__ __launch_bounds__(PK_THREADS) void fr_scale_pow_kernel(ScalePowBatch b) {
    const ScalePowJob jb = b.job[blockIdx.y];
    const ScalePowParams& p = b.sp[blockIdx.y];
    const uint64_t t = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x;
    const uint64_t lo = t * b.chunk;
    if (lo >= jb.n) return;
    const uint64_t hi = lo + b.chunk < jb.n ? lo + b.chunk : jb.n;
    Fr pw = p.c0;
#pragma unroll 1
    for (int k = 0; (t >> k) != 0; k++)
        if ((t >> k) & 1) pw = pw * p.bp[k];
    for (uint64_t i = lo; i < hi; i++) {
        jb.out[i] = jb.in[i] * pw;
        pw = pw * p.base;
    }
}

// highest index with a non-zero coefficient, +1 (atomicMax into *len, which the caller zeroes): one atomic per workgroup, and only
// from workgroups that can still raise the value (2^18 atomics on one address took 39 us, one per wave 37 us)
__global__ __launch_bounds__(PK_THREADS) void fr_trim_len_kernel(const Fr* __restrict__ c, uint64_t n, unsigned long long* len) {
    __shared__ unsigned long long top[PK_THREADS / 64];
    const uint64_t i = (uint64_t)blockIdx.x * PK_THREADS + threadIdx.x;
    const bool nz = i < n && !c[i].is_zero();
    const unsigned long long mask = __ballot(nz);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) top[wv] = mask ? (i + (unsigned long long)(63 - __builtin_clzll(mask)) + 1) : 0ull;  // i = first index of the wave
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = 0;
#pragma unroll
        for (int w = 0; w < PK_THREADS / 64; w++) m = top[w] > m ? top[w] : m;
        if (m > *reinterpret_cast<volatile unsigned long long*>(len)) atomicMax(len, m);
    }
}

// out[0..3] = *a, out[4..7] = *b, out[8..11] = *c: three field elements into (pinned, device-accessible) host memory in one launch
__global__ void fr_gather3_kernel(const Fr* __restrict__ a, const Fr* __restrict__ b, const Fr* __restrict__ c, uint64_t* __restrict__ out) {
    const uint32_t t = threadIdx.x;
    if (t >= 12) return;
    const Fr* src = t < 4 ? a : t < 8 ? b : c;
    out[t] = reinterpret_cast<const uint64_t*>(src)[t & 3];
}

// One thread, last in its stream: tells the polling host (plonk_host.inc: wait_stream) that everything enqueued before it is done
__global__ void stream_signal_kernel(unsigned long long* __restrict__ flag, unsigned long long seq) {
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace zkp
