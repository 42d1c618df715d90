// ntt.cuh -- LDS-tiled multi-pass NTT kernels for gfx950, generic over the field (Fr / Goldilocks).
//
// Semantics = ark-poly 0.4 Radix2EvaluationDomain as the reference uses it (natural order in and out;
// plonk/src/prover.rs:374-375,396-426,463; plonk/src/circuit.rs:175,230-232; fri/src/fri_layer.rs:40-46).
//
// Decomposition (Bailey-style, P <= 4 passes): log_n = r_0 + ... + r_{P-1}, input index
// n = (d_0, ..., d_{P-1}) most-significant digit first.  Pass p transforms digit d_p into the frequency
// digit k_p in place and multiplies by the inter-pass twiddle omega_{M_p}^{k_p * inner_index}
// (M_p = size of the remaining sub-problem).  The last pass also performs the digit reversal so the output
// index is k = k_0 + R_0 k_1 + R_0 R_1 k_2 + ...  (natural order).
//
// Each workgroup owns one tile of R x T elements in LDS (T adjacent columns so that every global access is a
// run of T*sizeof(F) = 256 contiguous bytes), loads the radix-R twiddles into LDS once, and runs the log2(R)
// radix-2 DIF stages K at a time in registers between LDS exchanges.
#pragma once
#include "ff.cuh"

namespace zkp {

enum { SCALE_NONE = 0, SCALE_CONST = 1, SCALE_POW = 2 };

// value(e) = lo[e & (2^h - 1)] * hi[e >> h]  -- two-level table of powers of one base
template <class F>
struct PowTab {
    const F* lo;
    const F* hi;
    uint32_t h;
};

template <class F>
struct ScaleSpec {
    int mode;     // SCALE_*
    F c;          // SCALE_CONST factor
    PowTab<F> t;  // SCALE_POW tables (index = natural element index)
};

template <class F>
ZKP_DEV F powtab_get(const PowTab<F>& t, uint64_t e) {
    F a = t.lo[e & ((1ull << t.h) - 1)];
    uint64_t hi = e >> t.h;
    if (hi) a = a * t.hi[hi];
    return a;
}
template <class F>
ZKP_DEV F apply_scale(const F& x, const ScaleSpec<F>& s, uint64_t idx) {
    if (s.mode == SCALE_CONST) return x * s.c;
    if (s.mode == SCALE_POW) return x * powtab_get(s.t, idx);
    return x;
}

template <class F> struct NttTraits;
template <> struct NttTraits<Fr> {
    static constexpr int LOG_T = 3;        // 8 x 32 B = 256 B runs
    static constexpr int MAX_TILE_LOG = 11;  // 2048 elements = 64 KiB
    static constexpr int K = 2;            // stages per register round
};
template <> struct NttTraits<Gl> {
    static constexpr int LOG_T = 5;        // 32 x 8 B = 256 B runs
    static constexpr int MAX_TILE_LOG = 13;  // 8192 elements = 64 KiB
    static constexpr int K = 3;
};
constexpr int NTT_THREADS = 256;
constexpr int NTT_MAX_PASS_LOG = 8;

ZKP_DEV uint32_t bitrev(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// K radix-2 DIF stages (s_hi .. s_hi-K+1) on an R x T tile: rows are the transform index, `stride` elements apart.
template <class F, int K>
ZKP_DEV void ntt_round(F* tile, const F* tw, int log_r, int s_hi, int t_log, int stride, int tid) {
    const int s_lo = s_hi - K + 1;
    const int items = ((1 << log_r) >> K) << t_log;
    for (int item = tid; item < items; item += NTT_THREADS) {
        const int t = item & ((1 << t_log) - 1);
        const int g = item >> t_log;
        const int low = g & ((1 << s_lo) - 1);
        const int base = ((g >> s_lo) << (s_hi + 1)) | low;
        F x[1 << K];
#pragma unroll
        for (int i = 0; i < (1 << K); i++) x[i] = tile[(base + (i << s_lo)) * stride + t];
#pragma unroll
        for (int q = K - 1; q >= 0; q--) {
            const int s = s_lo + q;
#pragma unroll
            for (int i = 0; i < (1 << K); i++) {
                if (i & (1 << q)) continue;
                const int row = base + (i << s_lo);
                F u = x[i], v = x[i | (1 << q)];
                x[i] = u + v;
                F d = u - v;
                if (s != 0) d = d * tw[(row & ((1 << s) - 1)) << (log_r - 1 - s)];  // omega_R^0 = 1 on the last stage
                x[i | (1 << q)] = d;
            }
        }
#pragma unroll
        for (int i = 0; i < (1 << K); i++) tile[(base + (i << s_lo)) * stride + t] = x[i];
    }
    __syncthreads();
}

// all log_r stages; leaves X[k] in row bitrev(k)
template <class F>
ZKP_DEV void ntt_tile(F* tile, const F* tw, int log_r, int t_log, int stride, int tid) {
    constexpr int K = NttTraits<F>::K;
    int s_hi = log_r - 1;
    while (s_hi >= K - 1) {
        ntt_round<F, K>(tile, tw, log_r, s_hi, t_log, stride, tid);
        s_hi -= K;
    }
    if (K >= 3 && s_hi == 1) { ntt_round<F, 2>(tile, tw, log_r, s_hi, t_log, stride, tid); s_hi -= 2; }
    if (s_hi == 0) ntt_round<F, 1>(tile, tw, log_r, s_hi, t_log, stride, tid);
}

template <class F>
struct NttStridedParams {
    const F* in;
    F* out;
    const F* tw;         // omega_R^j, j < R/2
    uint64_t n;          // transform size (batch stride)
    uint64_t inner;      // contiguous inner extent (elements), multiple of T
    uint32_t log_r;
    uint32_t tw_stride_log;  // inter-pass exponent = k * i << tw_stride_log (in units of omega_N)
    PowTab<F> inter;     // powers of omega_N
    ScaleSpec<F> pre;    // applied at load (first pass only), index = natural input index
};

// Non-final pass: view [outer][R][inner], tile = all R x T adjacent inner columns; in place.
template <class F>
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_strided(NttStridedParams<F> p) {
    extern __shared__ uint4 zkp_smem[];
    constexpr int LOG_T = NttTraits<F>::LOG_T;
    constexpr int T = 1 << LOG_T;
    const int tid = threadIdx.x;
    const int R = 1 << p.log_r;
    F* tile = reinterpret_cast<F*>(zkp_smem);
    F* tw = tile + (size_t)R * T;
    const uint64_t tiles_per_outer = p.inner >> LOG_T;
    const uint64_t o = blockIdx.x / tiles_per_outer;
    const uint64_t i0 = (blockIdx.x % tiles_per_outer) << LOG_T;
    const F* in = p.in + (uint64_t)blockIdx.y * p.n;
    F* out = p.out + (uint64_t)blockIdx.y * p.n;

    for (int j = tid; j < R / 2; j += NTT_THREADS) tw[j] = p.tw[j];
    for (int e = tid; e < R * T; e += NTT_THREADS) {
        const int j = e >> LOG_T, t = e & (T - 1);
        const uint64_t idx = (o * R + j) * p.inner + i0 + t;
        F x = in[idx];
        if (p.pre.mode != SCALE_NONE) x = apply_scale(x, p.pre, idx);
        tile[e] = x;
    }
    __syncthreads();
    ntt_tile<F>(tile, tw, p.log_r, LOG_T, T, tid);
    for (int e = tid; e < R * T; e += NTT_THREADS) {
        const int k = e >> LOG_T, t = e & (T - 1);
        F x = tile[(bitrev(k, p.log_r) << LOG_T) + t];
        const uint64_t ex = ((uint64_t)k * (i0 + t)) << p.tw_stride_log;
        if (ex) x = x * powtab_get(p.inter, ex);
        out[(o * R + k) * p.inner + i0 + t] = x;
    }
}

template <class F>
struct NttLastParams {
    const F* in;
    F* out;
    const F* tw;
    uint64_t n;
    uint32_t log_r;    // radix of this (last) pass
    uint32_t log_r0;   // radix of pass 0 (0 when P == 1)
    uint32_t log_m;    // log2 of the product of the middle radices
    uint32_t log_r1;   // radix of pass 1 when P == 4 (digit reversal of the middle index), else log_m
    uint32_t t_log;    // log2 of adjacent k_0 values per tile
    ScaleSpec<F> pre;  // applied at load when this is also the first pass (P == 1)
    ScaleSpec<F> post; // applied at store, index = natural output index
};

// Final pass: view [R0][M][R] -> out[k0 + R0*(rev(m) + M*k)].  Tile = 2^t_log adjacent k0 at one m.
template <class F>
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_last(NttLastParams<F> p) {
    extern __shared__ uint4 zkp_smem[];
    const int tid = threadIdx.x;
    const int R = 1 << p.log_r;
    const int T = 1 << p.t_log;
    const int stride = T > 1 ? T + 1 : 1;  // +1 element of padding: the transposing LDS writes stay conflict-light
    F* tile = reinterpret_cast<F*>(zkp_smem);
    F* tw = tile + (size_t)R * stride;
    const uint64_t m = blockIdx.x & ((1ull << p.log_m) - 1);
    const uint64_t k0b = (blockIdx.x >> p.log_m) << p.t_log;
    const F* in = p.in + (uint64_t)blockIdx.y * p.n;
    F* out = p.out + (uint64_t)blockIdx.y * p.n;

    for (int j = tid; j < R / 2; j += NTT_THREADS) tw[j] = p.tw[j];
    for (int e = tid; e < R * T; e += NTT_THREADS) {
        const int a = e >> p.log_r, j = e & (R - 1);
        const uint64_t idx = ((((k0b + a) << p.log_m) + m) << p.log_r) + j;
        F x = in[idx];
        if (p.pre.mode != SCALE_NONE) x = apply_scale(x, p.pre, idx);
        tile[j * stride + a] = x;
    }
    __syncthreads();
    ntt_tile<F>(tile, tw, p.log_r, p.t_log, stride, tid);
    // middle digits: m = (k_1, k_2) MS-first -> k_1 + R_1 k_2
    const uint32_t log_r2 = p.log_m - p.log_r1;
    const uint64_t mrev = (m >> log_r2) | ((m & ((1ull << log_r2) - 1)) << p.log_r1);
    for (int e = tid; e < R * T; e += NTT_THREADS) {
        const int k = e >> p.t_log, a = e & (T - 1);
        F x = tile[bitrev(k, p.log_r) * stride + a];
        const uint64_t idx = (k0b + a) + ((mrev + ((uint64_t)k << p.log_m)) << p.log_r0);
        if (p.post.mode != SCALE_NONE) x = apply_scale(x, p.post, idx);
        out[idx] = x;
    }
}

// out[e] = c * base^(e << shift), e < count
template <class F>
__global__ void pow_table_kernel(F base, F c, uint32_t shift, uint32_t count, F* out) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    F b = base;
    for (uint32_t i = 0; i < shift; i++) b = sqr(b);
    F r = c;
    uint32_t k = e;
    while (k) {
        if (k & 1) r = r * b;
        b = sqr(b);
        k >>= 1;
    }
    out[e] = r;
}

// c[i] = a[i] * b[i] (pointwise product between the forward and inverse transforms of a polynomial product)
template <class F>
__global__ void pointwise_mul_kernel(const F* a, const F* b, F* c, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = a[i] * b[i];
}

}  // namespace zkp
