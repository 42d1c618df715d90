// ntt.hpp -- LDS-tiled multi-pass NTT kernels for gfx950, generic over the field (Fr / Goldilocks).
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
// radix-2 DIT stages K at a time in registers between LDS exchanges (rows are loaded bit-reversed, so the tile
// ends in natural order).  DIT is chosen because its values grow additively under lazy reduction (fr29.hpp).
//
// Field policy NttOps<F>: F is the element type in global memory; E the in-register / in-LDS working type;
// W the twiddle type.  Fr works on unsaturated 29-bit limbs (E = W = Fr29); Goldilocks on plain u64.
#pragma once
#include "ff.hpp"
#include "fr29.hpp"

namespace zkp {

enum { SCALE_NONE = 0, SCALE_CONST = 1, SCALE_POW = 2, SCALE_POW_ROW = 3 };

template <class F> struct NttOps;
#ifndef ZKP_GL_LOG_T
#define ZKP_GL_LOG_T 4
#endif
#ifndef ZKP_GL_THREADS
#define ZKP_GL_THREADS 512
#endif
#ifndef ZKP_GL_MAX_PASS_LOG
#define ZKP_GL_MAX_PASS_LOG 9
#endif
template <> struct NttOps<Fr> {
    typedef Fr29 E;
    typedef Fr29 W;
    static constexpr int MAX_PASS_LOG = 8;   // radix of one pass of a multi-pass transform (2^9: 72 KiB tiles, one workgroup
                                             // per CU -- 2^26 in three passes measured 12.8 ms against 10.9 ms in four)
    static constexpr int THREADS = 256;      // workgroup size of the pass kernels
    static constexpr int LOG_T = 2;          // 4 x 32 B = 128 B runs (one cache line); 1024-element tiles = 36 KiB of
                                             // LDS, so 4 workgroups (4 waves/SIMD) fit a CU: the kernel is issue-bound
    static constexpr int MAX_TILE_LOG = 11;  // single-pass limit: 2048 elements x 36 B = 72 KiB of LDS
    // A radix-2^9 pass with tiles of TWO columns (64-byte runs, the same 36 KiB of LDS and one element-quad per thread as a radix-2^8
    // pass with four columns) where it saves a whole pass: 2^17 and 2^18 in two passes, 2^25 .. 2^27 in three.
    // ... and a radix-2^10 pass with single-column tiles (32-byte runs) likewise: 2^19 and 2^20 in two passes, 2^28 .. 2^30 in three.
    static constexpr int WIDE_PASS_LOG = 10;
    static ZKP_HD int log_t_of(int log_r) { return log_r <= MAX_PASS_LOG ? LOG_T : log_r == 9 ? 1 : 0; }
    // largest transform a wide radix is used for: 2^9 always, 2^10 (32-byte runs) only while the data is cache-resident
    static ZKP_HD int wide_max_log_n(int log_r) { return log_r <= 9 ? 64 : 20; }
    static constexpr int K = 2;              // stages per register round: 1024-element tiles / 4 = one item per thread
                                             // (K = 3 with a mid-round normalise leaves half the threads idle: measured 20 % slower)
    static constexpr bool MIDFIX = false;    // a third lazy stage would need re-normalised limbs (fr29.hpp)
    static constexpr int PAD = 0;            // 36-byte elements already spread over the LDS banks
    static constexpr bool LAST_LOAD_LDS_ORDER = true;  // see ntt_pass_last
    static ZKP_DEV E load(const Fr& x) { return fr29_from_sat(x); }
    static ZKP_DEV Fr store(const E& x) { return fr29_to_canonical(x); }
    static ZKP_DEV Fr store_tight(const E& x) { return fr29_pack_tight(x); }  // x is a product: limbs < 2^29, value < 2r
    static ZKP_DEV E mul(const E& a, const W& w) { return a * w; }
    // two independent products at once: the interleaved multiply-add chains of fr29.hpp (206 instead of 235 instructions each)
    static ZKP_DEV void mul2(const E& a0, const W& w0, const E& a1, const W& w1, E& r0, E& r1) { fr29_mul2(a0, w0, a1, w1, r0, r1); }
    static ZKP_DEV W wmul(const W& a, const W& b) { return a * b; }
    static ZKP_DEV E add(const E& u, const E& t) { return u + t; }
    static ZKP_DEV E sub(const E& u, const E& t) { return sub_tight(u, t); }
    static ZKP_DEV E fix(const E& x) { return normalise(x); }
    // (u, v) -> (u + v, u - v) with no product, for the stage-1 butterflies of a tile's first round whose twiddle is 1: u and v are
    // stage-0 sums (< 4r, limbs < 2^30); v is carry-propagated so that the 8r constant dominates it.  Results < 8r and < 12r: the
    // later stages add at most 4r each, 12r + 9 * 4r = 48r < 70r for the largest tile (2^11).
    static constexpr int UNIT_Q_MAX = 1;  // (profiles/r02_m: 2-4.5 % of a transform)
    static ZKP_DEV void unit_butterfly(E& u, E& v) {
        const E t = normalise(v);
        v = sub_wide8(u, t);
        u = u + t;
    }
    static ZKP_DEV W to_tw(const Fr& mont) { return fr29_twiddle_from_mont(mont); }
    // Pass 0 of a multi-pass transform reads its inter-pass twiddles omega_N^(k_0 i) from a matrix shaped like the data ([k_0][i],
    // one coalesced 32-byte load per element) instead of forming each one as the product of a low and a high table entry: one field
    // product less per element (13.5 -> 12.5 at 2^24, 10 -> 9 at 2^18) for 32 B per element of extra reads in an issue-bound kernel.
    static constexpr bool PASS0_MATRIX = true;
    static ZKP_DEV Fr tw_pack(const W& w) { return fr29_to_canonical(w); }
    static ZKP_DEV W tw_unpack(const Fr& x) { return fr29_from_sat(x); }
};
template <> struct NttOps<Gl> {
    typedef Gl E;
    typedef Gl W;
    // 16 x 8 B = 128 B runs; radix <= 2^9 so that 2^26 takes three passes (64 KiB tiles); 512 threads keep enough loads
    // in flight per tile.  Measured 2^20 / 2^24 / 2^26: (T 32, radix 2^8, 256 threads) 0.068 / 0.567 / 2.04 ms,
    // (16, 2^8, 512) 0.045 / 0.417 / 2.18, (16, 2^9, 512) 0.045 / 0.430 / 1.82, (16, 2^9, 1024) 0.048 / 0.451 / 1.75.
    static constexpr int LOG_T = ZKP_GL_LOG_T;
    static constexpr int MAX_PASS_LOG = ZKP_GL_MAX_PASS_LOG;
    static constexpr int THREADS = ZKP_GL_THREADS;
    static constexpr int MAX_TILE_LOG = 13;  // 8192 elements = 64 KiB
    // no wider radices for Goldilocks: radix 2^10 / 2^11 with 8 / 4-column tiles (two passes instead of three for 2^19 .. 2^22)
    // measured slower at every size (2^19 0.033 -> 0.060 ms, 2^22 0.112 -> 0.129 ms: the padded last-pass tile grows to 73 / 82 KiB
    // and the runs shrink to 64 / 32 bytes; profiles/r02_l_ntt_wide_pass.md)
    static constexpr int WIDE_PASS_LOG = MAX_PASS_LOG;
    static ZKP_HD int log_t_of(int) { return LOG_T; }
    static ZKP_HD int wide_max_log_n(int) { return 0; }
    static constexpr int K = 3;
    static constexpr bool MIDFIX = false;
    static constexpr int PAD = 1;            // +1 element per row keeps the transposing LDS writes conflict-light
    static constexpr bool LAST_LOAD_LDS_ORDER = false;
    static ZKP_DEV E load(const Gl& x) { return x; }
    static ZKP_DEV Gl store(const E& x) { return x; }
    static ZKP_DEV Gl store_tight(const E& x) { return x; }
    static ZKP_DEV E mul(const E& a, const W& w) { return a * w; }
    static ZKP_DEV void mul2(const E& a0, const W& w0, const E& a1, const W& w1, E& r0, E& r1) { r0 = a0 * w0; r1 = a1 * w1; }
    static ZKP_DEV W wmul(const W& a, const W& b) { return a * b; }
    static ZKP_DEV E add(const E& u, const E& t) { return u + t; }
    static ZKP_DEV E sub(const E& u, const E& t) { return u - t; }
    static ZKP_DEV E fix(const E& x) { return x; }
    static constexpr int UNIT_Q_MAX = K - 1;  // exact arithmetic: any stage of the first round
    static ZKP_DEV void unit_butterfly(E& u, E& v) {
        const E t = v;
        v = u - t;
        u = u + t;
    }
    static ZKP_DEV W to_tw(const Gl& canon) { return canon; }
    static constexpr bool PASS0_MATRIX = false;  // memory-bound: a product is cheaper than 8 more bytes per element
    static ZKP_DEV Gl tw_pack(const W& w) { return w; }
    static ZKP_DEV W tw_unpack(const Gl& x) { return x; }
};

// value(e) = lo[e & (2^h - 1)] * hi[e >> h]  -- two-level table of powers of one base
template <class F>
struct PowTab {
    const typename NttOps<F>::W* lo;
    const typename NttOps<F>::W* hi;
    uint32_t h;
};
template <class F>
struct ScaleSpec {
    int mode;                   // SCALE_*
    typename NttOps<F>::W c;    // SCALE_CONST factor
    PowTab<F> t;                // SCALE_POW tables (index = natural element index)
    uint64_t row0;              // SCALE_POW_ROW: element idx of transform b is multiplied by t[(row0 + b) * idx] -- the twiddle
                                // between the row and the column transforms of the four-step decomposition
};

template <class F>
ZKP_DEV typename NttOps<F>::W powtab_get(const PowTab<F>& t, uint64_t e) {
    typename NttOps<F>::W a = t.lo[e & ((1ull << t.h) - 1)];
    const uint64_t hi = e >> t.h;
    if (hi) a = NttOps<F>::wmul(a, t.hi[hi]);
    return a;
}
template <class F>
ZKP_DEV typename NttOps<F>::E apply_scale(const typename NttOps<F>::E& x, const ScaleSpec<F>& s, uint64_t idx, uint64_t batch = 0) {
    if (s.mode == SCALE_CONST) return NttOps<F>::mul(x, s.c);
    if (s.mode == SCALE_POW) return NttOps<F>::mul(x, powtab_get(s.t, idx));
    if (s.mode == SCALE_POW_ROW) return NttOps<F>::mul(x, powtab_get(s.t, (s.row0 + batch) * idx));
    return x;
}


ZKP_DEV uint32_t bitrev(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// K radix-2 DIT stages (s_lo .. s_lo+K-1) on an R x T tile; rows are `stride` elements apart.
// FIRST: s_lo == 0 is known at compile time.  Then the twiddle index of a butterfly depends on the register index alone
// (row mod 2^s = i mod 2^s), so besides all of stage 0 the butterflies of stage q whose low element has i mod 2^q == 0 multiply by
// omega^0 = 1 for EVERY lane: they are done without the product (O::unit_butterfly) -- for K = 2 one of the two stage-1 butterflies
// of each thread, a quarter of a field product per element and pass.
template <class F, int K, bool FIRST>
ZKP_DEV void ntt_round(typename NttOps<F>::E* tile, const typename NttOps<F>::W* tw, int log_r, int s_lo, int t_log,
                       int stride, int tid) {
    typedef NttOps<F> O;
    typedef typename O::E E;
    const int items = ((1 << log_r) >> K) << t_log;
    for (int item = tid; item < items; item += NttOps<F>::THREADS) {
        const int t = item & ((1 << t_log) - 1);
        const int g = item >> t_log;
        const int base = ((g >> s_lo) << (s_lo + K)) | (g & ((1 << s_lo) - 1));
        E x[1 << K];
#pragma unroll
        for (int i = 0; i < (1 << K); i++) x[i] = tile[(base + (i << s_lo)) * stride + t];
#pragma unroll
        for (int q = 0; q < K; q++) {
            const int s = s_lo + q;
            if (O::MIDFIX && q == 2) {
#pragma unroll
                for (int i = 0; i < (1 << K); i++) x[i] = O::fix(x[i]);
            }
            // the twiddle products of this stage first, two butterflies at a time (O::mul2: the lane's butterflies of one stage are
            // independent of each other), then the additions
            E tv[1 << K];
            if (s != 0) {  // omega_R^0 = 1 on stage 0
                int held = -1;
#pragma unroll
                for (int i = 0; i < (1 << K); i++) {
                    if (i & (1 << q)) continue;
                    if (FIRST && q >= 1 && q <= O::UNIT_Q_MAX && (i & ((1 << q) - 1)) == 0) continue;  // unit butterfly: no product
                    if (held < 0) {
                        held = i;
                        continue;
                    }
                    const int row0 = base + (held << s_lo), row1 = base + (i << s_lo);
                    O::mul2(x[held | (1 << q)], tw[(row0 & ((1 << s) - 1)) << (log_r - 1 - s)], x[i | (1 << q)],
                            tw[(row1 & ((1 << s) - 1)) << (log_r - 1 - s)], tv[held], tv[i]);
                    held = -1;
                }
                if (held >= 0) {
                    const int row = base + (held << s_lo);
                    tv[held] = O::mul(x[held | (1 << q)], tw[(row & ((1 << s) - 1)) << (log_r - 1 - s)]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < (1 << K); i++)
                    if (!(i & (1 << q))) tv[i] = x[i | (1 << q)];
            }
#pragma unroll
            for (int i = 0; i < (1 << K); i++) {
                if (i & (1 << q)) continue;
                if (FIRST && q >= 1 && q <= O::UNIT_Q_MAX && (i & ((1 << q) - 1)) == 0) {
                    O::unit_butterfly(x[i], x[i | (1 << q)]);
                    continue;
                }
                const E u = x[i];
                x[i] = O::add(u, tv[i]);
                x[i | (1 << q)] = O::sub(u, tv[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < (1 << K); i++) tile[(base + (i << s_lo)) * stride + t] = O::fix(x[i]);
    }
    __syncthreads();
}

// all log_r stages on a tile whose rows were loaded in bit-reversed order; leaves X[k] in row k
template <class F>
ZKP_DEV void ntt_tile(typename NttOps<F>::E* tile, const typename NttOps<F>::W* tw, int log_r, int t_log, int stride,
                      int tid) {
    constexpr int K = NttOps<F>::K;
    int s_lo = 0;
    if (K <= log_r) {
        ntt_round<F, K, true>(tile, tw, log_r, 0, t_log, stride, tid);
        s_lo = K;
    }
    while (s_lo + K <= log_r) {
        ntt_round<F, K, false>(tile, tw, log_r, s_lo, t_log, stride, tid);
        s_lo += K;
    }
    if (K >= 3 && log_r - s_lo == 2) { ntt_round<F, 2, false>(tile, tw, log_r, s_lo, t_log, stride, tid); s_lo += 2; }
    if (log_r - s_lo == 1) ntt_round<F, 1, false>(tile, tw, log_r, s_lo, t_log, stride, tid);
}

// Gathered input layout (first pass of a transform only): logical element e of transform b lives at physical element
//   b * batch_stride + (e mod 2^lo_bits) + ((e >> lo_bits) mod 2^mid_bits) * mid_stride + (e >> (lo_bits + mid_bits)) * hi_stride.
// This is how the row transforms of the multi-GPU four-step NTT read what the all-to-all delivered -- [source rank][my row]
// [that rank's columns] blocks, possibly in several column chunks -- without a transpose pass (zkp_hip/dist.py).
struct NttRemap {
    uint32_t on;  // 0: contiguous transforms, element e of transform b at b * n + e
    uint32_t lo_bits, mid_bits;
    uint64_t mid_stride, hi_stride, batch_stride;
};
ZKP_DEV uint64_t ntt_phys(const NttRemap& r, uint64_t b, uint64_t n, uint64_t e) {
    if (!r.on) return b * n + e;
    const uint64_t lo = e & ((1ull << r.lo_bits) - 1), rest = e >> r.lo_bits;
    return b * r.batch_stride + lo + (rest & ((1ull << r.mid_bits) - 1)) * r.mid_stride + (rest >> r.mid_bits) * r.hi_stride;
}

template <class F>
struct NttStridedParams {
    const F* in;
    F* out;
    const typename NttOps<F>::W* tw;  // omega_R^j, j < R/2
    uint64_t n;          // transform size (batch stride)
    uint64_t inner;      // contiguous inner extent (elements), multiple of T
    uint32_t log_r;
    uint32_t tw_stride_log;  // inter-pass exponent = k * i << tw_stride_log (in units of omega_N)
    PowTab<F> inter;     // powers of omega_N
    ScaleSpec<F> pre;    // applied at load (first pass only), index = natural input index
    NttRemap remap;      // gathered input (first pass only)
    // Transforms along axis 0 of a row-major matrix [L][B] (B = 2^col_bits columns, the "batch" is the contiguous direction):
    // the inner index is (remaining row digits, column), so the inter-pass exponent uses inner >> col_bits.
    uint32_t col_bits;
    // last pass of an axis-0 transform: rows leave in NATURAL order (row = outer index + outer_count * k instead of
    // outer index * R + k) and every element is multiplied by base^((col0 + column) * row) from `inter` -- the twiddle of the
    // four-step decomposition between the column and the row transforms, fused into the store.
    uint32_t axis0_last;
    uint32_t tw_on;      // 0: the table is a constant (1, or 1/len for the inverse): always read entry 0
    uint64_t outer_count;
    uint64_t col0;
    const F* tw_matrix;  // pass 0 only (one outer block): the inter-pass twiddle of output element e at tw_matrix[e], or null
    ClkRec* clk;         // profiling: clock stamps of one workgroup in sixteen (zkp_profile_clock_read "ntt_fr_pass" / "ntt_gl_pass"), or null
};

// Non-final pass: view [outer][R][inner], tile = all R x T adjacent inner columns; in place.
template <class F, int LOG_T>
__global__ __launch_bounds__(NttOps<F>::THREADS) void ntt_pass_strided(NttStridedParams<F> p) {
    extern __shared__ uint4 zkp_smem[];
    typedef NttOps<F> O;
    typedef typename O::E E;
    typedef typename O::W W;
    constexpr int T = 1 << LOG_T;
    const int tid = threadIdx.x;
    const int R = 1 << p.log_r;
    E* tile = reinterpret_cast<E*>(zkp_smem);
    W* tw = reinterpret_cast<W*>(tile + (size_t)R * T);
    const uint64_t tiles_per_outer = p.inner >> LOG_T;
    const uint64_t o = blockIdx.x / tiles_per_outer;
    const uint64_t i0 = (blockIdx.x % tiles_per_outer) << LOG_T;
    F* out = p.out + (uint64_t)blockIdx.y * p.n;
    ClkRec* const clk = (blockIdx.x & 15) == 0 ? p.clk : nullptr;  // a tile lives for tens of microseconds: one in sixteen is plenty
    uint64_t clk_t0 = 0, clk_r0 = 0;
    clk_begin(clk, clk_t0, clk_r0);

    for (int j = tid; j < R / 2; j += O::THREADS) tw[j] = p.tw[j];
    for (int e = tid; e < R * T; e += O::THREADS) {
        // walk the tile in LDS order (conflict-free stores); the global rows are a whole cache line apart either way
        const int j = (int)bitrev(e >> LOG_T, p.log_r), t = e & (T - 1);
        const uint64_t idx = (o * R + j) * p.inner + i0 + t;
        E x = O::load(p.in[ntt_phys(p.remap, blockIdx.y, p.n, idx)]);
        if (p.pre.mode != SCALE_NONE) x = apply_scale<F>(x, p.pre, idx);
        tile[e] = x;
    }
    __syncthreads();
    ntt_tile<F>(tile, tw, p.log_r, LOG_T, T, tid);
    // always multiply (exponent 0 hits the table's Montgomery one): the product is tight, so the hand-off to the next pass needs no
    // reduction at all.  Two elements per step: their products are independent (O::mul2).
    auto factor = [&](int e, uint64_t& at) -> W {
        const int k = e >> LOG_T, t = e & (T - 1);
        if (p.axis0_last) {
            const uint64_t row = o + p.outer_count * (uint64_t)k;  // natural order along axis 0
            at = row * p.inner + i0 + t;
            return powtab_get<F>(p.inter, p.tw_on ? row * (p.col0 + i0 + t) : 0);  // inner == number of columns here; no twiddle: entry 0 = the constant
        }
        at = (o * R + k) * p.inner + i0 + t;
        if (p.tw_matrix) return O::tw_unpack(p.tw_matrix[at]);
        return powtab_get<F>(p.inter, ((uint64_t)k * ((i0 + t) >> p.col_bits)) << p.tw_stride_log);
    };
    int e = tid;
    for (; e + O::THREADS < R * T; e += 2 * O::THREADS) {
        uint64_t at0, at1;
        const W w0 = factor(e, at0), w1 = factor(e + O::THREADS, at1);
        E x0, x1;
        O::mul2(tile[e], w0, tile[e + O::THREADS], w1, x0, x1);
        if (p.axis0_last) {  // leaves the library's hands: canonical
            out[at0] = O::store(x0);
            out[at1] = O::store(x1);
        } else {
            out[at0] = O::store_tight(x0);
            out[at1] = O::store_tight(x1);
        }
    }
    if (e < R * T) {
        uint64_t at;
        const W w = factor(e, at);
        const E x = O::mul(tile[e], w);
        out[at] = p.axis0_last ? O::store(x) : O::store_tight(x);
    }
    clk_end(clk, clk_t0, clk_r0);
}

template <class F>
struct NttLastParams {
    const F* in;
    F* out;
    const typename NttOps<F>::W* tw;
    uint64_t n;
    uint32_t log_r;    // radix of this (last) pass
    uint32_t log_r0;   // radix of pass 0 (0 when P == 1)
    uint32_t log_m;    // log2 of the product of the middle radices
    uint32_t log_r1;   // radix of pass 1 when P == 4 (digit reversal of the middle index), else log_m
    uint32_t t_log;    // log2 of adjacent k_0 values per tile
    ScaleSpec<F> pre;  // applied at load when this is also the first pass (P == 1)
    ScaleSpec<F> post; // applied at store, index = natural output index
    NttRemap remap;    // gathered input when this is also the first pass (P == 1)
    NttRemap out_remap; // scattered output (same mapping, applied to the natural output index)
    ClkRec* clk;       // as NttStridedParams::clk
};

// Final pass: view [R0][M][R] -> out[k0 + R0*(rev(m) + M*k)].  Tile = 2^t_log adjacent k0 at one m.
template <class F>
__global__ __launch_bounds__(NttOps<F>::THREADS) void ntt_pass_last(NttLastParams<F> p) {
    extern __shared__ uint4 zkp_smem[];
    typedef NttOps<F> O;
    typedef typename O::E E;
    typedef typename O::W W;
    const int tid = threadIdx.x;
    const int R = 1 << p.log_r;
    const int T = 1 << p.t_log;
    const int stride = T > 1 ? T + O::PAD : 1;
    E* tile = reinterpret_cast<E*>(zkp_smem);
    W* tw = reinterpret_cast<W*>(tile + (size_t)R * stride);
    const uint64_t m = blockIdx.x & ((1ull << p.log_m) - 1);
    const uint64_t k0b = (blockIdx.x >> p.log_m) << p.t_log;
    ClkRec* const clk = (blockIdx.x & 15) == 0 ? p.clk : nullptr;
    uint64_t clk_t0 = 0, clk_r0 = 0;
    clk_begin(clk, clk_t0, clk_r0);
    for (int j = tid; j < R / 2; j += O::THREADS) tw[j] = p.tw[j];
    for (int e = tid; e < R * T; e += O::THREADS) {
        // Fr walks the tile in LDS order, as the strided pass does: with consecutive lanes on consecutive input elements the
        // bit-reversed rows of a wave fall 16 rows apart, i.e. on the same LDS banks (16 % of this pass's wave cycles were bank
        // conflicts, profiles/r01_h_ntt_fr_counters.txt); the scattered 32-byte reads this way round stay inside the tile's
        // own few KB of input and are served by the caches: 2^24 2.379 -> 2.312 ms, 2^26 11.58 -> 11.44 ms
        // (profiles/r02_b_ntt_last_pass_load_order.md).  Goldilocks keeps the input order (8-byte elements, padded rows:
        // 0.410 ms against 0.417 ms the other way round).
        int row, a, j;
        if (O::LAST_LOAD_LDS_ORDER) {
            row = e >> p.t_log; a = e & (T - 1);
            j = (int)bitrev((uint32_t)row, p.log_r);
        } else {  // input order: coalesced reads, the row padding keeps the bit-reversed LDS writes apart
            a = e >> p.log_r; j = e & (R - 1);
            row = (int)bitrev((uint32_t)j, p.log_r);
        }
        const uint64_t idx = ((((k0b + a) << p.log_m) + m) << p.log_r) + j;
        E x = O::load(p.in[ntt_phys(p.remap, blockIdx.y, p.n, idx)]);
        if (p.pre.mode != SCALE_NONE) x = apply_scale<F>(x, p.pre, idx);
        tile[row * stride + a] = x;
    }
    __syncthreads();
    ntt_tile<F>(tile, tw, p.log_r, p.t_log, stride, tid);
    // middle digits: m = (k_1, k_2) MS-first -> k_1 + R_1 k_2
    const uint32_t log_r2 = p.log_m - p.log_r1;
    const uint64_t mrev = (m >> log_r2) | ((m & ((1ull << log_r2) - 1)) << p.log_r1);
    for (int e = tid; e < R * T; e += O::THREADS) {
        const int k = e >> p.t_log, a = e & (T - 1);
        E x = tile[k * stride + a];
        const uint64_t idx = (k0b + a) + ((mrev + ((uint64_t)k << p.log_m)) << p.log_r0);
        if (p.post.mode != SCALE_NONE) x = apply_scale<F>(x, p.post, idx, blockIdx.y);
        p.out[ntt_phys(p.out_remap, blockIdx.y, p.n, idx)] = O::store(x);
    }
    clk_end(clk, clk_t0, clk_r0);
}

// out[e] = c * base^(e << shift) in twiddle form, e < count.  base and c are given in F's own multiplicative form
// (Montgomery for Fr, canonical for Goldilocks).
template <class F>
__global__ void pow_table_kernel(F base, F c, uint32_t shift, uint32_t count, typename NttOps<F>::W* out) {
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
    out[e] = NttOps<F>::to_tw(r);
}

// out[k * inner + i] = tab[k * i], k < n / inner: the inter-pass twiddles of pass 0 laid out like the data they multiply
template <class F>
__global__ void twiddle_matrix_kernel(PowTab<F> tab, uint64_t n, uint64_t inner, F* out) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    out[e] = NttOps<F>::tw_pack(powtab_get<F>(tab, (e / inner) * (e % inner)));
}

// data[r][c] *= base^((row0 + r) * c): the twiddle step between the two halves of a four-step (multi-GPU) transform
template <class F>
__global__ void twiddle_rows_kernel(F* data, uint64_t rows, uint64_t cols, uint64_t row0, PowTab<F> tab) {
    typedef NttOps<F> O;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const uint64_t r = i / cols, c = i % cols;
    const typename O::E x = O::mul(O::load(data[i]), powtab_get<F>(tab, (row0 + r) * c));
    data[i] = O::store(x);
}

// Index permutation of 32-byte elements between two strided views of up to four power-of-two dimensions (most significant
// first): element (i0, i1, i2, i3) moves from in[sum i_k in_stride_k] to out[sum i_k out_stride_k].  The copies of the in-process
// multi-GPU transform (csrc/ntt_sharded.inc): packing a slab into per-peer blocks, unpacking what the peers delivered, and the
// transposition behind a natural-order output.  Two lanes per element (16 bytes each), consecutive lanes on consecutive elements
// of the innermost dimension: whichever side has stride 1 there moves whole cache lines, the other whole 32-byte sectors.
struct PermuteSpec {
    uint32_t bits[4];
    uint64_t in_stride[4], out_stride[4];  // in elements
};
__global__ void fr_permute_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, PermuteSpec s, uint64_t total) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t e = t >> 1;
    if (e >= total) return;
    uint64_t src = 0, dst = 0;
#pragma unroll
    for (int d = 3; d >= 0; d--) {
        const uint64_t i = e & ((1ull << s.bits[d]) - 1);
        e >>= s.bits[d];
        src += i * s.in_stride[d];
        dst += i * s.out_stride[d];
    }
    out[2 * dst + (t & 1)] = in[2 * src + (t & 1)];
}

// data[i] *= c * base^(idx0 + i): the coset scaling of a slab of a distributed vector (pre-scaling of coset_fft, post-scaling of
// coset_ifft with c = 1), tables as in SCALE_POW
template <class F>
__global__ void scale_pow_kernel(F* data, uint64_t count, uint64_t idx0, PowTab<F> tab) {
    typedef NttOps<F> O;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    data[i] = O::store(O::mul(O::load(data[i]), powtab_get<F>(tab, idx0 + i)));
}

// c[i] = a[i] * b[i] (pointwise product between the forward and inverse transforms of a polynomial product)
template <class F>
__global__ void pointwise_mul_kernel(const F* a, const F* b, F* c, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = a[i] * b[i];
}

}  // namespace zkp
