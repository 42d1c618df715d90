// msm.hpp -- Pippenger (bucket method) G1 multi-scalar multiplication kernels for gfx950.
//
// Computes the same group element as the reference's KzgScheme::evaluate_in_s (kzg/src/scheme.rs:84-96), which
// does n independent double-and-add scalar multiplications; here:
//   1. msm_digits      scalars (Montgomery Fr) -> canonical -> signed c-bit window digits, window-major
//   2. msm_parthist / msm_partscan / msm_partscatter   level A of the counting sort: partition by the high bucket bits
//   3. msm_binsort     level B: one workgroup per (window, partition) sorts by the low 8 bits inside L2
//   4. msm_order       buckets ranked by decreasing size, so the lanes of a wave walk runs of equal length
//   5. msm_accumulate  one lane per bucket walks its run of sorted indices: gather the 128-B internal affine point
//                      (28-bit limbs, fq28.hpp), XYZZ mixed add
//   6. msm_pyramid     log-depth weighted bucket reduction: sum_b b*B_b = sum(B) + sum_l 2^l * U_l,
//                      U_l = sum of the odd-indexed entries of level l of the pairwise-sum pyramid
//   7. msm_collect     gathers the c per-window results for one small D2H copy
// The O(W * c) serial tail (Horner over the U_l, window combine, one inversion) is latency-bound and runs on the
// host (host_ff.hpp); see DESIGN.md.
#pragma once
#include "g1.hpp"
#include "g1_28.hpp"
#include "fq28_inv.hpp"

namespace zkp {

constexpr int MSM_THREADS = 256;
constexpr int ACC_THREADS = 256;  // workgroup size of msm_accumulate (64 and 128 measure the same)

// Two modes.  Per-window buckets (default): every c-bit window of every MSM of a batch is its own "sort window" with
// n = ns entries and 2^(c-1) buckets.  Shared buckets (bases expanded with zkp_g1_bases_precompute): the W windows of a
// scalar address W pre-multiplied copies of the base (planes 2^(c s) P_i), so ALL of them fall into ONE bucket set:
// the sort window has n = W * ns entries, entry e = s * ns + i selects plane s, point i.
struct MsmGeom {
    uint32_t c;        // window bits
    uint32_t nwin;     // sort windows (bucket sets) in this pass
    uint32_t nb;       // buckets per window = 2^(c-1)   (bucket ids 1..nb)
    uint32_t nchunk;   // chunks per window in the counting sort
    uint64_t n;        // entries per sort window
    uint64_t chunk;    // entries per chunk
    uint64_t ns;       // scalars per MSM
    uint64_t plane_stride;  // points per plane of the expanded bases (shared mode)
    uint32_t nslice;   // c-bit windows per scalar
    uint32_t shared;   // 1: shared bucket set
    uint32_t run_limit;  // buckets with more entries are cut into pieces (msm_order)
    uint32_t piece;      // entries per piece
    uint32_t resume;     // 1: the buckets already hold the sums of earlier passes over other scalar ranges (shared mode)
    uint16_t off[36];    // bit offset of every slice of a scalar (off[nslice] >= 256); widths <= c
    uint32_t interleave; // 1: msm_accumulate walks the bucket sets interleaved (see there)
    uint32_t split_log;  // 2^split_log lanes (quads) share a bucket's run, one contiguous part each (small problems, see msm_accumulate)
    uint32_t more;       // 1: another scalar range follows: the bucket sums go to the hand-over array (msm_accumulate_body), not to `buckets`
};

// Bucket, pyramid and odd-sum arrays are PLANE-MAJOR: a 256-byte XYZZ entry is 16 chunks of 16 bytes, and chunk q of entry e
// lives at base[q * capacity + e] (capacity = nwin * nb entries, the same for all five arrays of a pass).  Lanes of a wave work
// on adjacent entries, so every load / store instruction covers contiguous memory; with entry-major 256-byte records each
// instruction touched 64 different cache lines and the top levels of the bucket reduction were bound by that, not by
// arithmetic (bench_micro/layout_copy.hip: the level-0 access pattern alone 131 us entry-major, 47 us plane-major).
ZKP_DEV uint64_t bucket_cap(const MsmGeom& g) { return (uint64_t)g.nwin * g.nb; }

#ifdef ZKP_MSM_CHECK  // diagnosis builds only (tools/job_r05_range_check.sh): every index the sort and the accumulate form is range-checked,
// a violation is recorded here (first offender per class) instead of being dereferenced
__device__ uint32_t g_msm_check[32];
ZKP_DEV bool msm_check_fail(int cls, uint32_t v0, uint32_t v1) {
    if (atomicAdd(&g_msm_check[cls * 4], 1u) == 0) { g_msm_check[cls * 4 + 1] = v0; g_msm_check[cls * 4 + 2] = v1; }
    return true;
}
#endif
// digit encoding in memory: (|d| << 1) | (d < 0); 0 = skip
constexpr int MSM_MAX_BATCH = 64;
struct DigitSources {
    const Fr* scalars[MSM_MAX_BATCH];  // one scalar vector per MSM of the batch (blockIdx.y), already offset to this range
};
__global__ __launch_bounds__(MSM_THREADS) void msm_digits_kernel(DigitSources src, const uint8_t* __restrict__ base_inf, MsmGeom g,
                                                                uint32_t nwin1, uint32_t* __restrict__ digits) {
    const uint64_t i = (uint64_t)blockIdx.x * MSM_THREADS + threadIdx.x;
    if (i >= g.ns) return;
    const uint32_t win_off = blockIdx.y * nwin1;  // digits laid out [msm][slice][scalar]
    Fr k = from_mont(src.scalars[blockIdx.y][i]);
    const bool skip = base_inf != nullptr && base_inf[i] != 0;  // infinity base contributes nothing
    uint32_t carry = 0;
    for (uint32_t w = 0; w < nwin1; w++) {  // nwin1 = g.nslice windows of this scalar vector
        const uint32_t lo = g.off[w], width = (uint32_t)g.off[w + 1] - lo;
        const uint32_t limb = lo >> 5, sh = lo & 31;
        uint64_t v = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {  // static indexing keeps the limbs in registers
            if (q == (int)limb) v |= (uint64_t)k.l[q];
            if (q == (int)limb + 1) v |= (uint64_t)k.l[q] << 32;
        }
        uint32_t u = ((uint32_t)(v >> sh) & ((1u << width) - 1)) + carry;
        uint32_t enc;
        if (u > (1u << (width - 1))) {  // u in (2^(width-1), 2^width]: use u - 2^width < 0 and carry one into the next slice
            enc = (((1u << width) - u) << 1) | 1u;
            carry = 1;
        } else {
            enc = u << 1;
            carry = 0;
        }
        digits[(uint64_t)(win_off + w) * g.ns + i] = skip ? 0u : enc;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Counting sort of the point indices by bucket, two levels so that every scattered store lands in a region that
// stays resident in L2 until its cache lines are complete (a single-level scatter over 2^15 buckets writes 4 bytes
// per 64-byte fabric transaction: measured 8.6x write amplification, profiles/r01_a_pmc_hbm_msm20.md):
//   level A: partition by the high bits of the bucket id (<= 128 partitions per window), 8-byte entries
//            (index|sign, low bits) written in long runs;
//   level B: one workgroup per (window, partition) sorts its entries by the low 8 bits inside a few hundred KB.
// Bucket ids are 1..nb; (b - 1) = hi * 2^lo_bits + lo.
// ---------------------------------------------------------------------------------------------------------
struct SortGeom {
    uint32_t lo_bits;  // 8 or 9 (at most c - 1): bins of the second pass
    uint32_t nhi;      // partitions per window = nb >> lo_bits
};

constexpr uint32_t SORT_MAX_PART = 8192;  // partitions per window: 2^(c-1) buckets = partitions x (256 .. 1024 bins)
constexpr uint32_t MSM_MAX_WINDOW_BITS = 24;  // widest window of zkp_g1_bases_precompute (bounded by the sort geometry above)

// base[0..nbins] = exclusive prefix of cnt[0..nbins) by ONE wave: lanes own ceil(nbins / 64) consecutive bins each and
// a shuffle scan joins them.  Called by the first wave of the workgroup between two barriers.
ZKP_DEV void wave_exclusive_scan(const uint32_t* cnt, uint32_t* base, uint32_t nbins, uint32_t lane) {
    const uint32_t per = (nbins + 63) / 64;
    const uint32_t b0 = lane * per < nbins ? lane * per : nbins;
    const uint32_t b1 = b0 + per < nbins ? b0 + per : nbins;
    uint32_t sum = 0;
    for (uint32_t k = b0; k < b1; k++) sum += cnt[k];
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= (uint32_t)off) inc += t;
    }
    uint32_t run = inc - sum;
    for (uint32_t k = b0; k < b1; k++) {
        const uint32_t v = cnt[k];
        base[k] = run;
        run += v;
    }
    if (lane == 63) base[nbins] = run;
}

// cntA[(w * nchunk + q) * nhi + hi]
__global__ __launch_bounds__(1024) void msm_parthist_kernel(const uint32_t* __restrict__ digits, MsmGeom g, SortGeom sg,
                                                            uint32_t* __restrict__ cntA) {
    __shared__ uint32_t h[SORT_MAX_PART];
    const uint32_t q = blockIdx.x, w = blockIdx.y;
    for (uint32_t k = threadIdx.x; k < sg.nhi; k += blockDim.x) h[k] = 0;
    __syncthreads();
    const uint64_t begin = (uint64_t)q * g.chunk;
    const uint64_t end = begin + g.chunk < g.n ? begin + g.chunk : g.n;
    const uint32_t* d = digits + (uint64_t)w * g.n;
    const uint64_t nt = blockDim.x;
    for (uint64_t i = begin + threadIdx.x; i < end; i += 4 * nt) {  // 4 independent loads in flight per lane
        uint32_t e[4];
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = i + k * nt < end ? d[i + k * nt] : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t b = e[k] >> 1;
            if (b) atomicAdd(&h[(b - 1) >> sg.lo_bits], 1u);
        }
    }
    __syncthreads();
    uint32_t* out = cntA + ((uint64_t)w * g.nchunk + q) * sg.nhi;
    for (uint32_t k = threadIdx.x; k < sg.nhi; k += blockDim.x) out[k] = h[k];
}

// cntA[w][q][hi] -> exclusive prefix over the chunks q (per partition hi), tot[w][hi] = partition size.
// Workgroup = 16 chunk-groups x 64 partitions: each thread sums its chunk range, the 16 partial sums are joined through
// LDS, then the range is rewritten as prefixes (sequential depth 2 * nchunk / 16 instead of nchunk).
__global__ __launch_bounds__(1024) void msm_partprefix_kernel(uint32_t* __restrict__ cntA, MsmGeom g, SortGeom sg,
                                                              uint32_t* __restrict__ tot) {
    __shared__ uint32_t part[16][64];
    const uint32_t w = blockIdx.y, h = threadIdx.x & 63, qg = threadIdx.x >> 6;
    const uint32_t hi = blockIdx.x * 64 + h;
    const bool live = hi < sg.nhi;
    uint32_t* cw = cntA + (uint64_t)w * g.nchunk * sg.nhi;
    const uint32_t per = (g.nchunk + 15) / 16;
    const uint32_t q0 = qg * per < g.nchunk ? qg * per : g.nchunk;
    const uint32_t q1 = q0 + per < g.nchunk ? q0 + per : g.nchunk;
    uint32_t sum = 0;
    if (live)
        for (uint32_t q = q0; q < q1; q++) sum += cw[(uint64_t)q * sg.nhi + hi];
    part[qg][h] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
    for (uint32_t k = 0; k < 16; k++) {
        if (k < qg) run += part[k][h];
        total += part[k][h];
    }
    if (live) {
        for (uint32_t q = q0; q < q1; q++) {
            uint32_t* p = cw + (uint64_t)q * sg.nhi + hi;
            const uint32_t v = *p;
            *p = run;
            run += v;
        }
        if (qg == 0) tot[(uint64_t)w * sg.nhi + hi] = total;
    }
}

constexpr uint32_t PYR_BAR_STRIDE = 32;  // words between the barrier counters of two windows (msm_pyramid_tail): a 128-byte line each
// One workgroup per window: pstart[w][hi] (nhi + 1 entries) = exclusive prefix of the partition sizes.
__global__ __launch_bounds__(64) void msm_partstart_kernel(const uint32_t* __restrict__ tot, SortGeom sg,
                                                           uint32_t* __restrict__ pstart, uint32_t* __restrict__ ghist,
                                                           uint32_t* __restrict__ tail_barrier) {
    __shared__ uint32_t t[SORT_MAX_PART + 1];  // scanned in place (wave_exclusive_scan reads cnt[k] before it writes base[k])
    uint32_t* base = t;
    const uint32_t w = blockIdx.x;
    // also clears what later kernels of this pass accumulate into (a hipMemsetAsync of 1 KB costs three 5 us fill kernels):
    // the bucket-size histogram filled by msm_binsort, the rank cursors of msm_rank and the arrival counter of msm_pyramid_tail
    for (uint32_t k = threadIdx.x; k < 256; k += 64) {
        ghist[w * 256 + k] = 0;
        ghist[(gridDim.x + w) * 256 + k] = 0;  // the rank cursors of msm_rank live behind the histograms of all windows
    }
    if (threadIdx.x == 0) tail_barrier[w * PYR_BAR_STRIDE] = 0;
    for (uint32_t k = threadIdx.x; k < sg.nhi; k += 64) t[k] = tot[(uint64_t)w * sg.nhi + k];
    __syncthreads();
    wave_exclusive_scan(t, base, sg.nhi, threadIdx.x);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k <= sg.nhi; k += 64) pstart[(uint64_t)w * (sg.nhi + 1) + k] = base[k];
}

// entries[w * n + pos] = (index | sign << 31, low bits of bucket id - 1).
// A wave storing to 64 unrelated addresses is limited by the per-CU rate of uncoalesced lanes (measured ~0.25 lane/clk:
// 77 % of the old kernel's cycles were VMEM issue stalls, profiles/r01_e_sort_counters.txt), so every tile of PS_TILE
// digits is first ranked and staged in LDS in partition order; the copy-out then writes runs of consecutive entries.
// Dynamic LDS: cur[nhi] cnt[nhi] base[nhi+1] | stage[PS_TILE] (uint2) | part[PS_TILE] (u16).
#ifndef ZKP_BS_TILE
#define ZKP_BS_TILE 16384
#endif
// entries per tile of the second pass: 16384 = 64-byte runs per bin with 1024 bins (8192: 32-byte runs; round 4, profiles/r04_m:
// sort 2.49 -> 2.40 ms at 2^24, 0.197 -> 0.188 at 2^20; 4096 -> 8192 had given 3.12 -> 2.78 ms at 2^24 in round 2).  108 KB of LDS:
// one workgroup of 16 waves per CU.
constexpr int BS_TILE = ZKP_BS_TILE;
constexpr int BS_PER = BS_TILE / 1024;
constexpr int SORT_MAX_BINS = 1024;  // low-bit bins of the second pass (one workgroup of 1024 threads owns a partition)
#ifndef ZKP_PS_TILE
#define ZKP_PS_TILE 12288
#endif
// entries per tile of the first pass, as large as the 160 KB of LDS allow next to the 12 bytes per partition: 12288 entries over up to
// 2048 partitions leave as 48..96-byte runs (147 KB; 8192: 32..64-byte runs; round 4, profiles/r04_m: together with the 16384-entry
// second pass the sort goes 0.192 -> 0.176 ms at 2^20, 0.63 -> 0.57 at 2^22, 2.49 -> 2.3 ms at 2^24), 8192 up to 4096 partitions
// (131 KB), 4096 up to 8192 partitions -- 24-bit windows -- (139 KB).  One workgroup per CU, no loss on small problems.
constexpr int PS_TILE_BIG = ZKP_PS_TILE, PS_TILE_MID = 8192, PS_TILE_SMALL = 4096;
ZKP_HD size_t partscatter_lds_bytes(uint32_t nhi, int tile) { return 8 * (size_t)tile + 2 * (size_t)tile + 4 * (size_t)(3 * nhi + 1); }
ZKP_HD int partscatter_tile(uint32_t nhi) { return nhi <= 2048 ? PS_TILE_BIG : nhi <= 4096 ? PS_TILE_MID : PS_TILE_SMALL; }

template <int PS_TILE>
__global__ __launch_bounds__(1024) void msm_partscatter_kernel(const uint32_t* __restrict__ digits, MsmGeom g, SortGeom sg,
                                                               const uint32_t* __restrict__ cntA,
                                                               const uint32_t* __restrict__ pstart,
                                                               uint2* __restrict__ entries) {
    constexpr int PS_PER = PS_TILE / 1024;
    extern __shared__ uint4 zkp_smem[];
    uint2* stage = reinterpret_cast<uint2*>(zkp_smem);                       // 8-byte aligned region first
    uint16_t* part = reinterpret_cast<uint16_t*>(stage + PS_TILE);
    uint32_t* cur = reinterpret_cast<uint32_t*>(part + PS_TILE);
    uint32_t* cnt = cur + sg.nhi;
    uint32_t* base = cnt + sg.nhi;
    const uint32_t q = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    for (uint32_t k = tid; k < sg.nhi; k += 1024)
        cur[k] = pstart[(uint64_t)w * (sg.nhi + 1) + k] + cntA[((uint64_t)w * g.nchunk + q) * sg.nhi + k];
    const uint64_t begin = (uint64_t)q * g.chunk;
    const uint64_t end = begin + g.chunk < g.n ? begin + g.chunk : g.n;
    const uint32_t* d = digits + (uint64_t)w * g.n;
    uint2* out = entries + (uint64_t)w * g.n;
    const uint32_t lo_mask = (1u << sg.lo_bits) - 1;
    for (uint64_t t0 = begin; t0 < end; t0 += PS_TILE) {
        for (uint32_t k = tid; k < sg.nhi; k += 1024) cnt[k] = 0;
        __syncthreads();
        uint32_t e[PS_PER], rk[PS_PER];
#pragma unroll
        for (int k = 0; k < PS_PER; k++) {
            const uint64_t i = t0 + tid + (uint64_t)k * 1024;
            e[k] = i < end ? d[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < PS_PER; k++) {
            const uint32_t b = e[k] >> 1;
            rk[k] = b ? atomicAdd(&cnt[(b - 1) >> sg.lo_bits], 1u) : 0u;
        }
        __syncthreads();
        if (tid < 64) wave_exclusive_scan(cnt, base, sg.nhi, tid);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PS_PER; k++) {
            const uint32_t b = e[k] >> 1;
            if (b) {
                const uint32_t p = (b - 1) >> sg.lo_bits;
                const uint32_t pos = base[p] + rk[k];
                stage[pos] = make_uint2((uint32_t)(t0 - 0 + tid + (uint64_t)k * 1024) | ((e[k] & 1u) << 31), (b - 1) & lo_mask);
                part[pos] = (uint16_t)p;
            }
        }
        __syncthreads();
        const uint32_t total = base[sg.nhi];
        for (uint32_t j = tid; j < total; j += 1024) {
            const uint32_t p = part[j];
#ifdef ZKP_MSM_CHECK
            if ((uint64_t)cur[p] + (j - base[p]) >= g.n && msm_check_fail(0, cur[p] + (j - base[p]), p)) continue;
#endif
            out[cur[p] + (j - base[p])] = stage[j];
        }
        __syncthreads();
        for (uint32_t k = tid; k < sg.nhi; k += 1024) cur[k] += cnt[k];
    }
}

// One workgroup per (partition, window): counting sort by the low bits; writes sorted[] and start[w][b] (nb + 2 entries:
// start[w][b] = first sorted position of bucket b, start[w][nb + 1] = number of non-zero digits of the window).
// Same LDS staging as above for the copy-out.
__global__ __launch_bounds__(1024) void msm_binsort_kernel(const uint2* __restrict__ entries, MsmGeom g, SortGeom sg,
                                                           const uint32_t* __restrict__ pstart,
                                                           uint32_t* __restrict__ start, uint32_t* __restrict__ sorted,
                                                           uint32_t* __restrict__ ghist) {
    __shared__ uint32_t h[SORT_MAX_BINS + 1], cnt[SORT_MAX_BINS], base[SORT_MAX_BINS + 1];
    __shared__ uint32_t szh[256];  // size histogram of this partition's buckets (step (1) of the size ranking below)
    __shared__ uint32_t stage[BS_TILE];
    __shared__ uint16_t bin[BS_TILE];
    const uint32_t hi = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const uint32_t lo_n = 1u << sg.lo_bits;
    const uint32_t* ps = pstart + (uint64_t)w * (sg.nhi + 1);
    const uint32_t begin = ps[hi], end = ps[hi + 1];
    const uint2* in = entries + (uint64_t)w * g.n;
    if (tid < lo_n) cnt[tid] = 0;
    if (tid < 256) szh[tid] = 0;
    __syncthreads();
    const uint32_t nt = blockDim.x;
    for (uint32_t i = begin + tid; i < end; i += 4 * nt) {
        uint32_t y[4];
#pragma unroll
        for (int k = 0; k < 4; k++) y[k] = i + k * nt < end ? in[i + k * nt].y : 0xffffffffu;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (y[k] != 0xffffffffu) atomicAdd(&cnt[y[k]], 1u);
    }
    __syncthreads();
    // cnt[bin] is the size of bucket (hi << lo_bits) + bin + 1: ghist[w][255 - min(size, 255)], one global atomic per class and workgroup
    if (tid < lo_n) atomicAdd(&szh[255 - (cnt[tid] < 255 ? cnt[tid] : 255)], 1u);
    if (tid < 64) wave_exclusive_scan(cnt, h, lo_n, tid);  // h[bin] = first sorted position of the bin, relative to `begin`
    __syncthreads();
    if (tid < 256 && szh[tid]) atomicAdd(&ghist[w * 256 + tid], szh[tid]);
    if (tid < lo_n) h[tid] += begin;
    __syncthreads();
    uint32_t* sw = start + (uint64_t)w * (g.nb + 2);
    if (tid < lo_n) sw[(hi << sg.lo_bits) + tid + 1] = h[tid];
    if (hi == 0 && tid == 0) sw[0] = 0;
    if (hi == sg.nhi - 1 && tid == 0) sw[g.nb + 1] = end;
    __syncthreads();
    uint32_t* out = sorted + (uint64_t)w * g.n;
    for (uint32_t t0 = begin; t0 < end; t0 += BS_TILE) {
        if (tid < lo_n) cnt[tid] = 0;
        __syncthreads();
        uint2 e[BS_PER];
        uint32_t rk[BS_PER];
#pragma unroll
        for (int k = 0; k < BS_PER; k++) {
            const uint32_t i = t0 + tid + k * 1024;
            e[k] = i < end ? in[i] : make_uint2(0u, 0xffffffffu);
        }
#pragma unroll
        for (int k = 0; k < BS_PER; k++) rk[k] = e[k].y != 0xffffffffu ? atomicAdd(&cnt[e[k].y], 1u) : 0u;
        __syncthreads();
        if (tid < 64) wave_exclusive_scan(cnt, base, lo_n, tid);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BS_PER; k++)
            if (e[k].y != 0xffffffffu) {
                const uint32_t pos = base[e[k].y] + rk[k];
                stage[pos] = e[k].x;
                bin[pos] = (uint16_t)e[k].y;
            }
        __syncthreads();
        const uint32_t total = base[lo_n];
        for (uint32_t j = tid; j < total; j += 1024) {
            const uint32_t p = bin[j];
#ifdef ZKP_MSM_CHECK
            if ((uint64_t)h[p] + (j - base[p]) >= g.n && msm_check_fail(1, h[p] + (j - base[p]), p)) continue;
#endif
            out[h[p] + (j - base[p])] = stage[j];
        }
        __syncthreads();
        if (tid < lo_n) h[tid] += cnt[tid];
    }
}

// perm[w][rank] = bucket id, buckets ordered by DEcreasing size (counting sort on min(size, 255)): the 64 lanes of
// a wave then walk runs of (almost) equal length, and the longest runs are dispatched first.
//
// Oversized buckets (more than MSM_RUN_LIMIT entries: skewed scalars, or the sparse top window when c does not divide
// the scalar width) would serialise one lane for the whole run.  They are cut into pieces of MSM_PIECE entries that
// extra lanes accumulate in parallel; msm_combine then adds the pieces of each such bucket (one wave per bucket).
//   over[w]: {n_over, n_pieces};  over_b[w][r] = bucket id (r-th oversized bucket, same order as perm);
//   over_off[w][r] = first piece index;  desc[w][j] = {bucket, piece index, first entry, last entry + 1}
// "Oversized" is relative to the average run: limit = max(128, 4 n / nb), piece = limit / 2 (MsmGeom::run_limit, piece).

// (1) size histogram ghist[w][bin], bin = 255 - min(size, 255): filled by msm_binsort, which has every bucket's size in LDS
// (2) the exclusive prefix of ghist gives every size class its first rank: msm_rank and msm_order scan the 256 counters themselves
// (3) ranks: every workgroup reserves a range per bin with one global atomic, then ranks its buckets inside it
__global__ __launch_bounds__(1024) void msm_rank_kernel(const uint32_t* __restrict__ start, MsmGeom g,
                                                        const uint32_t* __restrict__ ghist, uint32_t* __restrict__ gcur,
                                                        uint32_t* __restrict__ perm) {
    __shared__ uint32_t hist[256], base[256], gh[256], first[257];
    const uint32_t w = blockIdx.y, tid = threadIdx.x;
    const uint32_t* sw = start + (uint64_t)w * (g.nb + 2);
    if (tid < 256) {
        hist[tid] = 0;
        gh[tid] = ghist[w * 256 + tid];
    }
    __syncthreads();
    if (tid < 64) wave_exclusive_scan(gh, first, 256, tid);  // first rank of every size class (gcur counts from zero)
    const uint32_t b = 1 + blockIdx.x * 1024 + tid;
    uint32_t bin = 0, rk = 0;
    if (b <= g.nb) {
        const uint32_t sz = sw[b + 1] - sw[b];
        bin = 255 - (sz < 255 ? sz : 255);
        rk = atomicAdd(&hist[bin], 1u);
    }
    __syncthreads();
    if (tid < 256 && hist[tid]) base[tid] = first[tid] + atomicAdd(&gcur[w * 256 + tid], hist[tid]);
    __syncthreads();
#ifdef ZKP_MSM_CHECK
    if (b <= g.nb && base[bin] + rk >= g.nb && msm_check_fail(5, base[bin] + rk, b)) return;
#endif
    if (b <= g.nb) perm[(uint64_t)w * g.nb + base[bin] + rk] = b;
}

// (4) piece bookkeeping for the oversized buckets (ranks 0 .. over[2w]-1 of perm); one workgroup per window
__global__ __launch_bounds__(1024) void msm_order_kernel(const uint32_t* __restrict__ start, MsmGeom g,
                                                         const uint32_t* __restrict__ ghist,
                                                         const uint32_t* __restrict__ perm, uint32_t* __restrict__ over,
                                                         uint32_t* __restrict__ over_b, uint32_t* __restrict__ over_off,
                                                         uint4* __restrict__ desc, uint32_t over_cap, uint32_t desc_cap) {
    __shared__ uint32_t s_over[2], gh[256], first[257];
    const uint32_t w = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const uint32_t* sw = start + (uint64_t)w * (g.nb + 2);
    const uint32_t* pw = perm + (uint64_t)w * g.nb;
    if (tid < 256) gh[tid] = ghist[w * 256 + tid];
    __syncthreads();
    if (tid < 64) wave_exclusive_scan(gh, first, 256, tid);
    __syncthreads();
    if (tid == 0) {
        // the size histogram saturates at 255: with a larger limit every saturated bucket is a candidate (re-checked below)
        const uint32_t limit_bin = g.run_limit < 254 ? g.run_limit : 254;
        s_over[0] = first[255 - limit_bin];  // buckets with min(size, 255) > limit_bin occupy the first ranks
    }
    __syncthreads();
    // piece bookkeeping for the oversized buckets (ranks 0 .. n_over-1 of perm)
    const uint32_t n_over = s_over[0] < over_cap ? s_over[0] : over_cap;
#ifdef ZKP_MSM_CHECK
    if (tid == 0 && s_over[0] > g.nb) msm_check_fail(7, s_over[0], over_cap);
#endif
    uint32_t* ob = over_b + (uint64_t)w * over_cap;
    uint32_t* oo = over_off + (uint64_t)w * (over_cap + 1);
    {   // pieces per candidate, exclusive prefix over the candidates (each thread owns a contiguous range)
        __shared__ uint32_t part[1024];
        const uint32_t per = (n_over + nt - 1) / nt;
        const uint32_t r0 = tid * per < n_over ? tid * per : n_over;
        const uint32_t r1 = r0 + per < n_over ? r0 + per : n_over;
        uint32_t local = 0;
        for (uint32_t r = r0; r < r1; r++) {
            const uint32_t b = pw[r];
#ifdef ZKP_MSM_CHECK
            if ((b == 0 || b > g.nb) && msm_check_fail(6, b, r)) continue;
#endif
            const uint32_t sz = sw[b + 1] - sw[b];
            local += sz > g.run_limit ? (sz + g.piece - 1) / g.piece : 0;  // candidates at or under the limit: no pieces
        }
        part[tid] = local;
        __syncthreads();
        for (uint32_t off = 1; off < nt; off <<= 1) {
            const uint32_t v = tid >= off ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        uint32_t run = tid ? part[tid - 1] : 0;
        for (uint32_t r = r0; r < r1; r++) {
            const uint32_t b = pw[r];
#ifdef ZKP_MSM_CHECK
            if (b == 0 || b > g.nb) { ob[r] = 1; oo[r] = run; continue; }
#endif
            const uint32_t sz = sw[b + 1] - sw[b];
            ob[r] = b;
            oo[r] = run;
            run += sz > g.run_limit ? (sz + g.piece - 1) / g.piece : 0;
        }
        if (tid == nt - 1) {
            const uint32_t total = part[nt - 1];
            oo[n_over] = total;
            s_over[1] = total;
            over[2 * w] = n_over;
            over[2 * w + 1] = total < desc_cap ? total : desc_cap;
        }
    }
    __syncthreads();
    const uint32_t n_pieces = s_over[1] < desc_cap ? s_over[1] : desc_cap;
    uint4* dw = desc + (uint64_t)w * desc_cap;
    for (uint32_t j = tid; j < n_pieces; j += nt) {
        uint32_t lo = 0, hi = n_over;  // largest r with oo[r] <= j
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (oo[mid] <= j) lo = mid; else hi = mid;
        }
        const uint32_t b = ob[lo], p = j - oo[lo];
        const uint32_t first = sw[b] + p * g.piece;
        const uint32_t last = first + g.piece < sw[b + 1] ? first + g.piece : sw[b + 1];
        dw[j] = make_uint4(b, p, first, last);
    }
}

// bases: 96 B affine (Montgomery radix 2^384, saturated) -> internal 128 B (28-bit limbs, Montgomery radix 2^392).
// x * 2^392 = (x * 2^384) * 2^8: eight modular doublings of the stored residue, then a re-slicing of the bits.
__global__ __launch_bounds__(MSM_THREADS) void g1_to_internal_kernel(const uint4* __restrict__ in, uint64_t n,
                                                                    uint4* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * MSM_THREADS + threadIdx.x;
    if (i >= n) return;
    G1Affine p = G1Affine::load(in + i * 6);
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
        p.x = dbl(p.x);
        p.y = dbl(p.y);
    }
    A28 q;
    q.x = fq28_from_sat(p.x);
    q.y = fq28_from_sat(p.y);
    q.store(out + i * 8);
}

// Expanded bases for the shared-bucket mode: plane s holds 2^off[s] * P_i (off[s] = c s for uniform slices) in the internal affine form.  One thread per
// point walks the whole doubling chain in XYZZ without normalising in between (c doublings per plane), parks the
// unnormalised (X, Y) in the plane's own slot and (ZZ, ZZZ, running product of the ZZZ) in a global scratch area, inverts
// the product ONCE (Montgomery's trick across the planes of the point) and walks back to make every plane affine:
// ~9 c + 10 field products per stored point plus one inversion per POINT (safegcd, fq28_inv.hpp: ~67 products; the Fermat power
// it replaces cost 592).
// planes: nplanes x plane_stride x 128 B, plane 0 already filled; this launch covers points [off, off + cnt);
// scratch: nplanes x 3 x cnt x 64 B.  (Thread-private arrays for the scratch values miscompile on this toolchain: only
// the last plane came out right, bench_micro/batch_inv_check.hip reproduces it; explicit global scratch is also cheaper
// than 4.9 KB of private memory per lane.)  An infinity / garbage base yields ZZZ = 0, which only zeroes its own planes
// (never read: msm_digits skips infinity bases).
struct SliceOffsets {
    uint16_t off[36];  // plane s holds 2^off[s] * P
};
__global__ __launch_bounds__(MSM_THREADS) void g1_expand_planes_kernel(uint4* __restrict__ planes, uint4* __restrict__ scratch,
                                                                      uint64_t off, uint64_t cnt, uint64_t plane_stride,
                                                                      uint32_t nplanes, SliceOffsets so) {
    const uint64_t j = (uint64_t)blockIdx.x * MSM_THREADS + threadIdx.x;
    if (j >= cnt) return;
    const uint64_t i = off + j;
    auto slot = [&](uint32_t s, uint32_t t) { return scratch + (((uint64_t)s * 3 + t) * cnt + j) * 4; };  // t: 0 ZZ 1 ZZZ 2 product
    const A28 p0 = A28::load(planes + i * 8);
    X28 x = g1_28_double_affine(p0);
    Fq28 run = Fq28::one();
#pragma unroll 1
    for (uint32_t s = 1; s < nplanes; s++) {
#pragma unroll 1
        for (uint32_t k = (s == 1 ? 1u : 0u), c = (uint32_t)so.off[s] - so.off[s - 1]; k < c; k++) x = g1_28_double(x);
        A28 raw;
        raw.x = x.x;  // < 14p, limbs < 2^30: any 32-bit limb pattern survives the round trip through memory
        raw.y = x.y;
        raw.store(planes + (s * plane_stride + i) * 8);
        run.store(slot(s, 2));  // product of ZZZ_1 .. ZZZ_{s-1}
        x.zz.store(slot(s, 0));
        x.zzz.store(slot(s, 1));
        run = run * x.zzz;
    }
    Fq28 inv = fq28_inverse_gcd(run);  // 1 / (ZZZ_1 ... ZZZ_last)
#pragma unroll 1
    for (uint32_t s = nplanes - 1; s >= 1; s--) {
        const Fq28 zi3 = inv * Fq28::load(slot(s, 2));  // 1 / ZZZ_s
        inv = inv * Fq28::load(slot(s, 1));
        // affine: x = X / ZZ, y = Y / ZZZ with (ZZ / ZZZ)^2 = 1 / ZZ
        const Fq28 zi = zi3 * Fq28::load(slot(s, 0));
        const Fq28 zi2 = zi * zi;
        uint4* dst = planes + (s * plane_stride + i) * 8;
        const A28 raw = A28::load(dst);
        A28 q;
        q.x = raw.x * zi2;  // 14p * 2p / 2520 p -> tight
        q.y = raw.y * zi3;
        q.store(dst);
    }
}

// Small problems (a 2^16-term commitment): with as many buckets as the machine has lanes the bucket reduction is a chain of c - 1
// dependent levels over mostly idle hardware, and with fewer buckets the lanes run out.  So 2^split_log lanes (or quads) share a
// bucket: each takes one contiguous part of its run and leaves a partial sum -- part 0 in the bucket array, part p > 0 in array
// p - 1 of `parts` (same plane-major geometry) -- and msm_fold_parts adds the parts up before the reduction.  That allows 16-bit
// windows (2^15 buckets, 14 reduction levels) at full occupancy where 18 bits (2^17 buckets, 16 levels) were needed before.
ZKP_DEV uint4* split_dst(const MsmGeom& g, uint4* buckets, uint4* parts, uint32_t part) {
    return part ? parts + (uint64_t)(part - 1) * 16 * bucket_cap(g) : buckets;
}
ZKP_DEV void split_run(const MsmGeom& g, uint32_t part, uint32_t& lo, uint32_t& hi) {
    if (!g.split_log) return;
    const uint32_t len = hi - lo;
    hi = lo + (uint32_t)(((uint64_t)len * (part + 1)) >> g.split_log);
    lo = lo + (uint32_t)(((uint64_t)len * part) >> g.split_log);
}

// (ClkRec, clk_begin, clk_end -- the in-kernel clock stamps of zkp_profile_clock_read -- live in ff.hpp: the NTT passes use them too)

// Physical position of logical entry i of an n-entry level of the reduction pyramid (msm_pyramid_kernel): even entries in the first
// half, odd entries in the second.  The bucket array is level 0 (n = nb, i = bucket - 1).
ZKP_DEV uint64_t pyr_pos(uint32_t i, uint32_t n) { return (uint64_t)(i >> 1) + (uint64_t)(i & 1u) * (n >> 1); }

// One lane per (window, bucket), buckets taken in decreasing-size order: buckets[w * nb + pyr_pos(b - 1, nb)] = sum of the
// bucket's points (internal XYZZ, 256 B).  Lanes past nb take the pieces of oversized buckets (see msm_order).
#ifdef ZKP_ACC_PLAIN_PRODUCTS  // A/B builds (profiles/r05_j): the compiler's schedule of every product, as rounds 1-4
constexpr bool ACC_CHAIN = false;
#else
constexpr bool ACC_CHAIN = true;   // the six plain products of an insertion as strict multiply-add chains (fq28.hpp: fq28_mul_chain)
#endif
ZKP_DEV void msm_accumulate_run(const uint4* __restrict__ bases28, const uint32_t* __restrict__ idx, uint32_t lo,
                                uint32_t hi, const MsmGeom& g, uint4* dst, uint64_t dst_stride, bool resume,
                                const uint4* src = nullptr, uint64_t src_stride = 1) {
#ifdef ZKP_MSM_CHECK
    if ((lo > hi || hi > g.n) && msm_check_fail(3, lo, hi)) return;
#endif
    if (resume && lo == hi) {  // nothing to add: the sum of the earlier ranges is carried over (in place: nothing to do)
        if (src != dst) X28::load_s(src, src_stride).store_s(dst, dst_stride);
        return;
    }
    X28 acc = resume ? X28::load_s(src, src_stride) : X28::infinity();
    auto locate = [&](uint32_t e) {
        uint64_t pt = e & 0x7fffffffu;
#ifdef ZKP_MSM_CHECK
        if (pt >= g.n && msm_check_fail(2, e, (uint32_t)g.n)) pt = 0;
#endif
        if (g.shared) {  // entry = slice * ns + i  ->  plane `slice` of the expanded bases, point i (32-bit divide: ns < 2^31)
            const uint32_t ns32 = (uint32_t)g.ns, s = (uint32_t)pt / ns32;
            pt = (uint64_t)s * g.plane_stride + ((uint32_t)pt - s * ns32);
        }
        return bases28 + pt * 8;
    };
    uint32_t k = lo;
    if (!resume && hi - lo >= 2) {
        // The first point of a run is copied and the second meets an affine accumulator: four of the ten products of the mixed
        // addition have an operand 1 (g1_28_mmadd).  All lanes of a wave are at the start of their runs together, so the peeled
        // iterations are uniform: 4 of (26 x 9.5) products per bucket at 2^20 terms, 4 of 152 per lane in a split 2^16 commitment.
        const uint32_t e0 = idx[k], e1 = idx[k + 1];
        {
            A28 p0 = A28::load(locate(e0));
            if (e0 >> 31) p0.y = neg4(p0.y);
            acc.x = p0.x;
            acc.y = normalise(p0.y);
        }
        A28 p1 = A28::load(locate(e1));
        if (e1 >> 31) p1.y = normalise(neg4(p1.y));
        if (g1_28_mmadd<ACC_CHAIN>(acc, p1)) {  // (reads acc.x, acc.y only; sets ZZ, ZZZ)
            k += 2;
        } else {                     // same x (a repeated or an opposite point): the general addition of the loop below takes it
            acc.zz = Fq28::one();
            acc.zzz = Fq28::one();
            k += 1;
        }
    }
    for (; k < hi; k++) {
        const uint32_t e = idx[k];
        A28 p = A28::load(locate(e));
        if (e >> 31) p.y = neg4(p.y);
        g1_28_madd<ACC_CHAIN>(acc, p);
    }
    acc.store_s(dst, dst_stride);
}

// Workgroup ids interleave the bucket sets (w = id % nwin, slot = id / nwin), so that the hardware's in-order dispatch
// walks ALL bucket sets in decreasing bucket size together (longest-processing-time-first over the whole launch, not per
// bucket set): 2^20 per-window 3.02 -> 2.50 ms.  At 2^24 the interleaved order is 8 % SLOWER (every resident workgroup then
// streams a different 64 MB index array), so the host turns it on up to 2^22 entries per bucket set only.
// Per bucket set: ceil(nb / 256) bucket slots followed by `extra` piece slots (grid-stride over the pieces).
ZKP_DEV void msm_accumulate_body(const uint4* __restrict__ bases28, const uint32_t* __restrict__ sorted,
                                 const uint32_t* __restrict__ start, const uint32_t* __restrict__ perm,
                                 const uint32_t* __restrict__ over, const uint4* __restrict__ desc, uint32_t desc_cap,
                                 uint32_t bucket_blocks, uint32_t extra_blocks, const MsmGeom& g, uint4* __restrict__ buckets,
                                 uint4* __restrict__ pieces, uint4* __restrict__ parts, uint4* __restrict__ carry) {
    const uint32_t per_set = bucket_blocks + extra_blocks;
    const uint32_t w = g.interleave ? blockIdx.x % g.nwin : blockIdx.x / per_set;
    const uint32_t slot = g.interleave ? blockIdx.x / g.nwin : blockIdx.x % per_set;
    const uint32_t* sw = start + (uint64_t)w * (g.nb + 2);
    const uint32_t* idx = sorted + (uint64_t)w * g.n;
    if (slot < bucket_blocks) {
        const uint32_t unit = slot * ACC_THREADS + threadIdx.x;
        const uint32_t rank = unit >> g.split_log, part = unit & ((1u << g.split_log) - 1);
        if (rank >= g.nb) return;
        const uint32_t b = perm[(uint64_t)w * g.nb + rank];
#ifdef ZKP_MSM_CHECK
        if ((b == 0 || b > g.nb) && msm_check_fail(3, b, rank | 0x80000000u)) return;
#endif
        uint32_t lo = sw[b], hi = sw[b + 1];
        // Between two scalar ranges a bucket's sum waits in the hand-over array `carry`, ENTRY-major (256 contiguous bytes per bucket):
        // lanes of a wave own buckets in size order, i.e. scattered ones, and a scattered lane reads or writes two full cache lines there
        // where the plane-major bucket array -- laid out for the reduction, whose lanes walk adjacent buckets -- gives it sixteen 16-byte
        // pieces of sixteen lines (measured on the host-scalar MSM, two ranges at 2^20: profiles/r05_o_range_handover.md).  Only the
        // last range writes `buckets`.
        const uint64_t e = (uint64_t)w * g.nb + pyr_pos(b - 1, g.nb);
        uint4* dst = g.more ? carry + e * 16 : split_dst(g, buckets, parts, part) + e;
        if (hi - lo > g.run_limit && rank < over[2 * w]) {  // cut into pieces, handled by the piece blocks
            if (part) hi = lo;  // (msm_combine writes the bucket itself: the other parts of it are empty)
            else return;
        }
        split_run(g, part, lo, hi);
        msm_accumulate_run(bases28, idx, lo, hi, g, dst, g.more ? 1 : bucket_cap(g), g.resume != 0, carry + e * 16, 1);
    } else {
        const uint32_t n_pieces = over[2 * w + 1];
#ifdef ZKP_MSM_CHECK
        if (n_pieces > desc_cap && msm_check_fail(4, n_pieces, desc_cap)) return;
#endif
        const uint32_t stride = extra_blocks * ACC_THREADS;
        for (uint32_t j = (slot - bucket_blocks) * ACC_THREADS + threadIdx.x; j < n_pieces; j += stride) {
            const uint4 d = desc[(uint64_t)w * desc_cap + j];
            msm_accumulate_run(bases28, idx, d.z, d.w, g, pieces + ((uint64_t)w * desc_cap + j) * 16, 1, false);
        }
    }
}

__global__ __launch_bounds__(ACC_THREADS) __attribute__((amdgpu_waves_per_eu(3))) void msm_accumulate_kernel(const uint4* __restrict__ bases28,
                                                                    const uint32_t* __restrict__ sorted,
                                                                    const uint32_t* __restrict__ start,
                                                                    const uint32_t* __restrict__ perm,
                                                                    const uint32_t* __restrict__ over,
                                                                    const uint4* __restrict__ desc, uint32_t desc_cap,
                                                                    uint32_t bucket_blocks, uint32_t extra_blocks, MsmGeom g,
                                                                    uint4* __restrict__ buckets,
                                                                    uint4* __restrict__ pieces, uint4* __restrict__ parts,
                                                                    uint4* __restrict__ carry, ClkRec* __restrict__ clk) {
    uint64_t t0 = 0, r0 = 0;
    clk_begin(clk, t0, r0);
    msm_accumulate_body(bases28, sorted, start, perm, over, desc, desc_cap, bucket_blocks, extra_blocks, g, buckets, pieces, parts, carry);
    clk_end(clk, t0, r0);
}

// The issue-rate probe of bench_micro/issue_rate.hip as a library kernel (zkp_probe_mad_rate): every lane runs 8 independent chains of
// v_mad_u64_u32, the instruction a field product is made of (392 per Fq28 product), at full occupancy; with the clock stamps the
// caller gets lane-mads per second AND per shader cycle on THIS device, now.
constexpr int MAD_PROBE_ITERS = 8192, MAD_PROBE_CHAINS = 8;
__global__ __launch_bounds__(256) void mad_rate_probe_kernel(uint32_t* __restrict__ out, uint32_t seed, ClkRec* __restrict__ clk) {
    uint64_t t0 = 0, r0 = 0;
    clk_begin(clk, t0, r0);
    const uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[MAD_PROBE_CHAINS];
    uint32_t lo[MAD_PROBE_CHAINS];
#pragma unroll
    for (int c = 0; c < MAD_PROBE_CHAINS; c++) { acc[c] = a + c; lo[c] = b + c; }
#pragma unroll 1
    for (int it = 0; it < MAD_PROBE_ITERS; it++) {
#pragma unroll
        for (int c = 0; c < MAD_PROBE_CHAINS; c++)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(lo[c]) : "vcc");
    }
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < MAD_PROBE_CHAINS; c++) r ^= (uint32_t)acc[c] ^ (uint32_t)(acc[c] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    clk_end(clk, t0, r0);
}

// ---- small problems: FOUR lanes per bucket ------------------------------------------------------------------------
// With few entries (a 2^16-term commitment fills 2^17 buckets with ~7 points each) the lane-per-bucket kernel is bound by
// the LATENCY of its longest run (22 dependent mixed adds of 16 us), not by throughput.  Here a quad shares the accumulator
// (lane 0: X, lane 1: ZZ, lane 2: Y, lane 3: ZZZ) and the 10 products of madd-2008-s run in 4 rounds:
//   round 1   -          | U2 = X2 ZZ1 | -            | S2 = Y2 ZZZ1        then P = U2 - X1 (lane 0), R = S2 - Y1 (lane 2)
//   round 2   PP = P^2   | -           | RR = R^2     | -
//   round 3   PPP = P PP | ZZ3 = ZZ1 PP| -            | Q = X1 PP
//   round 4   V = Y1 PPP | -           | T = R (Q-X3) | ZZZ3 = ZZZ1 PPP      X3 = RR - PPP - 2Q, Y3 = T - V on lane 2
// 16 product slots instead of 10, but a third of the latency: 2^16-term MSMs 402 -> ~200 us.  Every lane runs the same
// instruction stream (role-selected operands, width-4 shuffles).  P = 0 (equal or opposite x) falls back to the scalar
// g1_28_madd, computed redundantly by the four lanes.
ZKP_DEV void msm_accumulate_quad_run(const uint4* __restrict__ bases28, const uint32_t* __restrict__ idx, uint32_t lo,
                                     uint32_t hi, const MsmGeom& g, uint4* __restrict__ dst, uint64_t dst_stride, int j) {
    const bool odd = (j & 1) != 0, up = (j & 2) != 0;
    Fq28 own = Fq28::zero();  // this lane's share of the accumulator
    bool inf = true, fresh = false;  // fresh: the accumulator is the affine point the last insertion copied (ZZ = ZZZ = 1)
    auto point_of = [&](uint32_t e) -> const uint4* {
        uint64_t pt = e & 0x7fffffffu;
        if (g.shared) {
            const uint32_t ns32 = (uint32_t)g.ns, s = (uint32_t)pt / ns32;
            pt = (uint64_t)s * g.plane_stride + ((uint32_t)pt - s * ns32);
        }
        return bases28 + pt * 8;
    };
    // This kernel is bound by the latency of ONE run (a few waves per CU, each a chain of dependent insertions), so the two memory
    // latencies in front of an insertion's arithmetic -- the sorted index, then the gathered point -- are taken off the chain:
    // the point of insertion k + 1 is requested before the products of insertion k start, its index one insertion earlier still.
    // Measured (profiles/r04_h): accumulate 71 -> 64 us at 2^12 terms, 97 -> 91 us at 2^14, 301 -> 297 us at 2^16; nothing from 2^18 on
    // or in a PLONK batch, where two waves per SIMD already cover each other's loads (the lane-per-bucket kernel: no gain, r03_f).
    uint32_t e_next = lo < hi ? idx[lo] : 0u, e_next2 = lo + 1 < hi ? idx[lo + 1] : 0u;
    Fq28 c_next = Fq28::zero();
    if (lo < hi) c_next = Fq28::load(point_of(e_next) + (up ? 4 : 0));
    for (uint32_t k = lo; k < hi; k++) {
        const uint32_t e = e_next;
        const uint4* src = point_of(e);  // (address only: read again in the rare same-x path below)
        Fq28 coord = c_next;                          // lanes 0, 1: X2;  lanes 2, 3: Y2
        if (k + 1 < hi) {
            e_next = e_next2;
            c_next = Fq28::load(point_of(e_next) + (up ? 4 : 0));
        }
        if (k + 2 < hi) e_next2 = idx[k + 2];
        if (up && (e >> 31)) coord = neg4(coord);
        if (inf) {  // uniform over the quad
            own = odd ? Fq28::one() : (up ? normalise(coord) : coord);
            inf = false;
            fresh = true;
            continue;
        }
        // the insertion after a copy meets ZZ = ZZZ = 1: U2 = X2 and S2 = Y2 need no product, the first of the four rounds is skipped
        // (every quad of a wave is at its second insertion together; an accumulator that became infinite later is copied again and
        // is fresh again)
        Fq28 m1;
        if (fresh) m1 = normalise(coord);
        else m1 = coord * own;                                      // lane 1: U2, lane 3: S2
        fresh = false;
        const Fq28 d = sub16(quad_xor1(m1), own);                  // lane 0: P (< 18p), lane 2: R (< 18p)
        const Fq28 m2 = d * d;                                     // lane 0: PP, lane 2: RR
        const Fq28 pp = quad_bcast(m2, 0);
        if (quad_bcast0(tight_is_zero_mod_p(m2) ? 1 : 0)) {        // same x: rare, all four lanes do the scalar add
            X28 acc;
            acc.x = quad_bcast(own, 0); acc.zz = quad_bcast(own, 1); acc.y = quad_bcast(own, 2); acc.zzz = quad_bcast(own, 3);
            A28 q = A28::load(src);
            if (e >> 31) q.y = neg4(q.y);
            g1_28_madd(acc, q);
            inf = acc.is_inf();
            own = up ? (odd ? acc.zzz : acc.y) : (odd ? acc.zz : acc.x);
            continue;
        }
        const Fq28 x1 = quad_bcast(own, 0);
        const Fq28 m3 = fq28_select(j == 0, d, fq28_select(j == 3, x1, own)) * pp;  // PPP | ZZ3 | (Y1 PP, unused) | Q
        const Fq28 ppp = quad_bcast(m3, 0), q = quad_bcast(m3, 3);
        const Fq28 x3 = normalise(sub8w(sub4(m2, ppp), q + q));    // lane 2: RR - PPP - 2Q
        const Fq28 t = sub16(q, x3);
        const Fq28 y1 = quad_bcast(own, 2);
        const Fq28 m4 = fq28_select(j == 0, y1, fq28_select(j == 2, d, own)) * fq28_select(j == 2, t, ppp);  // V | - | T | ZZZ3
        const Fq28 v = quad_bcast(m4, 0);
        const Fq28 x3b = quad_bcast(x3, 2);
        own = up ? (odd ? m4 : normalise(sub4(m4, v))) : (odd ? m3 : x3b);
    }
    uint4* part = dst + (up ? (odd ? 12 : 4) : (odd ? 8 : 0)) * dst_stride;     // X | ZZ | Y | ZZZ
    if (inf) own = Fq28::zero();
    own.store_s(part, dst_stride);
}

// Same slots as msm_accumulate_kernel, 64 buckets (or bucket parts, or pieces) per workgroup
ZKP_DEV void msm_accumulate_quad_body(const uint4* __restrict__ bases28, const uint32_t* __restrict__ sorted,
                                      const uint32_t* __restrict__ start, const uint32_t* __restrict__ perm,
                                      const uint32_t* __restrict__ over, const uint4* __restrict__ desc, uint32_t desc_cap,
                                      uint32_t bucket_blocks, uint32_t extra_blocks, const MsmGeom& g, uint4* __restrict__ buckets,
                                      uint4* __restrict__ pieces, uint4* __restrict__ parts) {
    constexpr uint32_t QUADS = ACC_THREADS / 4;
    const uint32_t per_set = bucket_blocks + extra_blocks;
    const uint32_t w = g.interleave ? blockIdx.x % g.nwin : blockIdx.x / per_set;
    const uint32_t slot = g.interleave ? blockIdx.x / g.nwin : blockIdx.x % per_set;
    const uint32_t* sw = start + (uint64_t)w * (g.nb + 2);
    const uint32_t* idx = sorted + (uint64_t)w * g.n;
    const int j = threadIdx.x & 3;
    const uint32_t quad = threadIdx.x >> 2;
    if (slot < bucket_blocks) {
        const uint32_t unit = slot * QUADS + quad;
        const uint32_t rank = unit >> g.split_log, part = unit & ((1u << g.split_log) - 1);
        if (rank >= g.nb) return;  // whole quads
        const uint32_t b = perm[(uint64_t)w * g.nb + rank];
        uint32_t lo = sw[b], hi = sw[b + 1];
        uint4* dst = split_dst(g, buckets, parts, part) + ((uint64_t)w * g.nb + pyr_pos(b - 1, g.nb));
        if (hi - lo > g.run_limit && rank < over[2 * w]) {
            if (part) hi = lo;
            else return;
        }
        split_run(g, part, lo, hi);
        msm_accumulate_quad_run(bases28, idx, lo, hi, g, dst, bucket_cap(g), j);
    } else {
        const uint32_t n_pieces = over[2 * w + 1];
        const uint32_t stride = extra_blocks * QUADS;
        for (uint32_t p = (slot - bucket_blocks) * QUADS + quad; p < n_pieces; p += stride) {
            const uint4 d = desc[(uint64_t)w * desc_cap + p];
            msm_accumulate_quad_run(bases28, idx, d.z, d.w, g, pieces + ((uint64_t)w * desc_cap + p) * 16, 1, j);
        }
    }
}
__global__ __launch_bounds__(ACC_THREADS) void msm_accumulate_quad_kernel(const uint4* __restrict__ bases28,
                                                                         const uint32_t* __restrict__ sorted,
                                                                         const uint32_t* __restrict__ start,
                                                                         const uint32_t* __restrict__ perm,
                                                                         const uint32_t* __restrict__ over,
                                                                         const uint4* __restrict__ desc, uint32_t desc_cap,
                                                                         uint32_t bucket_blocks, uint32_t extra_blocks, MsmGeom g,
                                                                         uint4* __restrict__ buckets,
                                                                         uint4* __restrict__ pieces, uint4* __restrict__ parts,
                                                                         ClkRec* __restrict__ clk) {
    uint64_t t0 = 0, r0 = 0;
    clk_begin(clk, t0, r0);
    msm_accumulate_quad_body(bases28, sorted, start, perm, over, desc, desc_cap, bucket_blocks, extra_blocks, g, buckets, pieces, parts);
    clk_end(clk, t0, r0);
}

// One step of adding the parts of split buckets up (four lanes per add, msm_pyramid_quad's arithmetic): in step t array
// i * 2^(t+1) += array i * 2^(t+1) + 2^t, one pair per blockIdx.y, for plane-major arrays of `cap` entries each; array 0 is the
// bucket array, array i > 0 is parts array i - 1.  After split_log steps array 0 holds the bucket sums.
__global__ __launch_bounds__(MSM_THREADS) void msm_fold_parts_kernel(uint4* __restrict__ buckets, uint4* __restrict__ parts,
                                                                    uint64_t cap, uint32_t step) {
    const uint64_t e = (uint64_t)blockIdx.x * (MSM_THREADS / 4) + (threadIdx.x >> 2);
    if (e >= cap) return;  // whole quads
    const uint32_t ia = blockIdx.y << (step + 1), ib = ia + (1u << step);
    uint4* pa = (ia ? parts + (uint64_t)(ia - 1) * 16 * cap : buckets) + e;
    const uint4* pb = parts + (uint64_t)(ib - 1) * 16 * cap + e;
    g1_28_add_quad(pa, pb, pa, cap, threadIdx.x & 3);
}

// The same step with ONE lane per add (the streaming add of the reduction pyramid): a fold step over a batch of commitments is 10^5
// adds -- enough to fill the machine, where the cooperative form pays for its exchanges (it reaches half of the multiply-add rate, this
// one the pyramid's) -- profiles/r05_m_fold_lane.md; the host picks by the number of adds in the launch.
__global__ __launch_bounds__(MSM_THREADS) void msm_fold_parts_lane_kernel(uint4* __restrict__ buckets, uint4* __restrict__ parts,
                                                                         uint64_t cap, uint32_t step) {
    const uint64_t e = (uint64_t)blockIdx.x * MSM_THREADS + threadIdx.x;
    if (e >= cap) return;
    const uint32_t ia = blockIdx.y << (step + 1), ib = ia + (1u << step);
    uint4* pa = (ia ? parts + (uint64_t)(ia - 1) * 16 * cap : buckets) + e;
    const uint4* pb = parts + (uint64_t)(ib - 1) * 16 * cap + e;
    g1_28_add_stream_inplace<ACC_CHAIN>(pa, pb, cap);
}

// One wave per oversized bucket: lane i adds pieces i, i + 64, ...; then a 6-step tree through LDS.
__global__ __launch_bounds__(64) void msm_combine_kernel(const uint32_t* __restrict__ over, const uint32_t* __restrict__ over_b,
                                                         const uint32_t* __restrict__ over_off, uint32_t over_cap,
                                                         uint32_t desc_cap, MsmGeom g, const uint4* __restrict__ pieces,
                                                         uint4* __restrict__ buckets, uint4* __restrict__ carry) {
    __shared__ uint4 sh[64 * 16];
    const uint32_t w = blockIdx.y, lane = threadIdx.x;
    for (uint32_t r = blockIdx.x; r < over[2 * w]; r += gridDim.x) {
    const uint32_t b = over_b[(uint64_t)w * over_cap + r];
    const uint32_t* oo = over_off + (uint64_t)w * (over_cap + 1);
    uint32_t p0 = oo[r], p1 = oo[r + 1];
    if (p1 > desc_cap) p1 = desc_cap;
    if (p1 <= p0) continue;  // a candidate that turned out not to be oversized: its lane wrote the bucket directly
    const uint4* pw = pieces + (uint64_t)w * desc_cap * 16;
    X28 acc = X28::infinity();
    for (uint32_t p = p0 + lane; p < p1; p += 64) {
        X28 x = X28::load(pw + (uint64_t)p * 16);
        g1_28_add(acc, x);
    }
    for (uint32_t off = 32; off > 0; off >>= 1) {
        acc.store(sh + lane * 16);
        __syncthreads();
        if (lane < off) {
            X28 x = X28::load(sh + (lane + off) * 16);
            g1_28_add(acc, x);
        }
        __syncthreads();
    }
    if (lane == 0) {  // (the hand-over array between scalar ranges: msm_accumulate_body)
        const uint64_t e = (uint64_t)w * g.nb + pyr_pos(b - 1, g.nb);
        if (g.resume) {
            X28 x = X28::load_s(carry + e * 16, 1);
            g1_28_add(acc, x);
        }
        if (g.more) acc.store_s(carry + e * 16, 1);
        else acc.store_s(buckets + e, bucket_cap(g));
    }
    __syncthreads();
    }
}

// Log-depth weighted bucket reduction.  With A_0[i] = B_{i+1} (i < nb) and A_{l+1}[s] = A_l[2s] + A_l[2s+1]:
//     sum_b b * B_b  =  sum_i A_0[i]  +  sum_l 2^l * U_l,      U_l = sum_s A_l[2s+1]
// (bit l of the 0-based index i is bit 0 of i >> l).  Every level is stored DE-INTERLEAVED (pyr_pos): its even entries in the first
// half of its n_l = nb >> l entries, its odd entries in the second half.  A lane then reads its two operands from two contiguous runs
// (with adjacent entries interleaved every load touched every cache line of the level and used half of it: level 0 fetched 228 MB
// for 134 MB of operands, profiles/r04_e), and the odd half of level l IS the seed of U_l: nothing is copied (rounds 1-3 copied
// A_l[2s+1] into a separate array, 134 MB of extra writes at 2^19 buckets).  Launch l (l = 0 .. c-2) performs, for every window,
//   kind 0        : A_{l+1}[s] = A_l[2s] + A_l[2s+1]                                          s < half = n_l / 2
//   kind j+1 <= l : O_j[s] = O_j[s] + O_j[s + half]   (U_j is a plain sum: any pairing does; O_j has n_l partial sums before the launch;
//                   in launch j + 1 they are the odd half of level j itself, later launches find them in the `odd` buffers)
// so that after the last launch A_{c-1}[0] = sum(B), O_j[0] = U_j for j <= c-3, and U_{c-2} is the single odd entry of level c-2.
// Levels ping-pong between two buffers (level l in parity l & 1, at the start of a window's nb entries); launch j + 1 writes level
// j + 2 into the first quarter of level j's buffer while it reads level j's odd half there -- disjoint.
// O_j lives at entry offset nb - (nb >> j) of an `odd` buffer (reads of launch l come from parity l & 1).
struct PyrLevel {
    uint32_t level;  // l
    uint32_t half;   // nb >> (l+1)
    uint32_t nb;
    uint32_t nwin;
};
ZKP_DEV uint64_t odd_off(uint32_t nb, uint32_t j) { return (uint64_t)nb - (nb >> j); }
// operands and destination of item (kind, s) of launch `level`: entry offsets inside the buffers named by the flags
struct PyrItem {
    uint64_t a, b, d;
    bool src_is_pyr_out;  // kind j + 1 == level: the operands are level j's odd half, in the pyramid buffer this launch also writes
};
ZKP_DEV PyrItem pyr_item(uint32_t nb, uint32_t level, uint32_t half, uint32_t kind, uint32_t s, uint64_t wbase) {
    PyrItem it;
    if (kind == 0) {
        it.a = wbase + s;
        it.b = wbase + half + s;
        it.d = wbase + pyr_pos(s, half);
        it.src_is_pyr_out = false;
    } else {
        const uint64_t o = wbase + odd_off(nb, kind - 1);
        it.src_is_pyr_out = kind == level;
        const uint64_t src = it.src_is_pyr_out ? wbase + 2 * (uint64_t)half : o;  // level j = level - 1 has 4 half entries: odd half at [2 half, 4 half)
        it.a = src + s;
        it.b = src + half + s;
        it.d = o + s;
    }
    return it;
}

__global__ __launch_bounds__(MSM_THREADS) void msm_pyramid_kernel(const uint4* __restrict__ pyr_in,
                                                                  uint4* pyr_out,
                                                                  const uint4* __restrict__ odd_in,
                                                                  uint4* __restrict__ odd_out, PyrLevel L) {
    const uint32_t s = blockIdx.x * MSM_THREADS + threadIdx.x;
    if (s >= L.half) return;
    const uint32_t kind = blockIdx.y, w = blockIdx.z;
    const uint64_t cap = (uint64_t)L.nwin * L.nb;
    const PyrItem it = pyr_item(L.nb, L.level, L.half, kind, s, (uint64_t)w * L.nb);
    const uint4* src = kind == 0 ? pyr_in : (it.src_is_pyr_out ? pyr_out : odd_in);
    g1_28_add_stream<ACC_CHAIN>(src + it.a, src + it.b, (kind ? odd_out : pyr_out) + it.d, cap);
}

// Same level, four lanes per add (g1_28_add_quad): for the levels with too few adds to fill the machine, where the level
// time is the latency of ONE add (16 us on a lone lane, ~5 us on a quad).  grid.x = ceil(half / 64).
__global__ __launch_bounds__(MSM_THREADS) void msm_pyramid_quad_kernel(const uint4* __restrict__ pyr_in,
                                                                       uint4* pyr_out,
                                                                       const uint4* __restrict__ odd_in,
                                                                       uint4* __restrict__ odd_out, PyrLevel L) {
    const uint32_t s = blockIdx.x * (MSM_THREADS / 4) + (threadIdx.x >> 2);
    const int j = threadIdx.x & 3;
    if (s >= L.half) return;  // whole quads leave together
    const uint32_t kind = blockIdx.y, w = blockIdx.z;
    const uint64_t cap = (uint64_t)L.nwin * L.nb;
    const PyrItem it = pyr_item(L.nb, L.level, L.half, kind, s, (uint64_t)w * L.nb);
    const uint4* src = kind == 0 ? pyr_in : (it.src_is_pyr_out ? pyr_out : odd_in);
    g1_28_add_quad(src + it.a, src + it.b, (kind ? odd_out : pyr_out) + it.d, cap, j);
}

// The last levels of the pyramid have at most a few hundred pairwise adds per window: a few workgroups per window run them
// back to back with a device-scope barrier in between instead of one launch per level; four lanes per add.  A level here is the
// latency of ONE cooperative add, and a wave that shares its SIMD with a sibling issues at half the rate: rounds 1-3 ran eight
// workgroups of 512 threads (two waves per SIMD on eight CUs) and a level took 11.2 us against 8.7 us for the same level as its own
// launch with every wave alone on a SIMD (profiles/r04_e).  Now a workgroup is four waves -- one per SIMD of its CU -- and sixteen
// of them share a window (single-wave workgroups spread as well but quadruple the barrier's arrivals: measured slower).
constexpr uint32_t MSM_TAIL_TIMEOUT = 0x80000000u;  // flag in a window's barrier counter, checked by the host
constexpr uint32_t MSM_FLAG_PENDING = 0x40000000u;  // what the host writes into a result flag before the launch: "not written yet"
constexpr uint32_t MSM_FLAG_PENDING_MASK = 0x40000000u;
constexpr uint32_t PYR_TAIL_THREADS = 256;    // four waves = 64 cooperative adds per workgroup and round
constexpr uint32_t PYR_TAIL_BLOCKS = 16;      // workgroups per window at most (1024 adds per round)
// The spinning barrier below needs every workgroup of the launch RESIDENT (a spinner cannot make room for a sibling that was never
// scheduled): the host keeps windows x workgroups x waves below what this device holds at two waves per SIMD and what the kernel's
// own occupancy allows (Ctx::tail_max_waves: from hipDeviceProp and hipOccupancyMaxActiveBlocksPerMultiprocessor at slot creation --
// 2048 on a full MI355X, less on a partition), and runs every level as its own launch when even one workgroup per window is too many.
constexpr uint32_t PYR_TAIL_SPIN_LIMIT = 1u << 24;  // polls (~seconds) before a workgroup gives up
__global__ __launch_bounds__(512) void msm_pyramid_tail_kernel(uint4* __restrict__ pyr0, uint4* __restrict__ pyr1,
                                                               uint4* __restrict__ odd0, uint4* __restrict__ odd1,
                                                               uint32_t level0, uint32_t c, uint32_t nb,
                                                               uint32_t* __restrict__ barrier /* one zeroed counter per window */,
                                                               uint4* __restrict__ result /* as msm_collect_kernel: pinned host memory */,
                                                               uint32_t* __restrict__ flags /* one word per window, next to it */,
                                                               uint32_t expect_blocks /* arrivals per level = gridDim.x (a test hook
                                                                  asks for one more: the starvation path) */,
                                                               uint32_t spin_limit) {
    const uint32_t w = blockIdx.y;
    const uint64_t wbase = (uint64_t)w * nb;
    const uint64_t cap = (uint64_t)gridDim.y * nb;
    const int j = threadIdx.x & 3;
    const uint32_t quad = blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2), nquad = gridDim.x * (blockDim.x >> 2);
    uint32_t epoch = 0;
    for (uint32_t l = level0; l + 1 < c; l++) {
        const uint32_t half = nb >> (l + 1);
        const uint4* pyr_in = (l & 1) ? pyr1 : pyr0;
        uint4* pyr_out = (l & 1) ? pyr0 : pyr1;
        const uint4* odd_in = (l & 1) ? odd1 : odd0;
        uint4* odd_out = (l & 1) ? odd0 : odd1;
        for (uint32_t item = quad; item < (l + 1) * half; item += nquad) {  // a quad per add
            const uint32_t kind = item / half, s = item % half;
            const PyrItem it = pyr_item(nb, l, half, kind, s, wbase);
            const uint4* src = kind == 0 ? pyr_in : (it.src_is_pyr_out ? pyr_out : odd_in);
            g1_28_add_quad(src + it.a, src + it.b, (kind ? odd_out : pyr_out) + it.d, cap, j);
        }
        // barrier over the workgroups of this window: every workgroup arrives once per level.  Release: the workgroup's stores
        // (complete at the workgroup barrier) are written back for the other XCDs before the arrival is counted.  The wait polls with
        // RELAXED loads and takes ONE acquire fence when it is over: an acquire load invalidates the XCD's L2 on every poll, and
        // with many workgroups waiting that kept every cache on the chip cold under the few that were still adding (rounds 1-3;
        // profiles/r04_e: spreading the waiters made a 48-bucket-set reduction 3.29 -> 3.80 ms before this change, 3.26 ms after).
        epoch++;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t* bar = barrier + (uint64_t)w * PYR_BAR_STRIDE;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // bounded spin (~seconds): a scheduling surprise (a sibling workgroup that never became resident) must not hang
            // the GPU.  The workgroup that gives up sets the top bit of the counter: every spinner then leaves at once and
            // the host, which reads the counters back with the results, reports ZKP_E_DEVICE instead of a wrong sum.
            bool arrived = false;
            for (uint32_t spin = 0; spin < spin_limit; spin++) {
                if (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch * expect_blocks) {
                    arrived = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!arrived) __hip_atomic_fetch_or(bar, MSM_TAIL_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // every array is down to one entry and the last barrier has made them visible: gather them (msm_collect_kernel's job)
    if (blockIdx.x == 0) {
        const uint4* pyr_final = ((c - 1) & 1) ? pyr1 : pyr0;
        const uint4* pyr_prev = ((c - 1) & 1) ? pyr0 : pyr1;   // level c - 2: two entries, the odd one is U_{c-2}
        const uint4* odd_final = ((c - 1) & 1) ? odd1 : odd0;
        for (uint32_t t = threadIdx.x; t < c * 16; t += blockDim.x) {
            const uint32_t e = t >> 4, q = t & 15;
            const uint4* src = e == 0 ? pyr_final + wbase : e == c - 1 ? pyr_prev + (wbase + 1) : odd_final + (wbase + odd_off(nb, e - 1));
            result[((uint64_t)w * c + e) * 16 + q] = src[q * cap];
        }
        // the barrier counter goes home with the results: its top bit says that a workgroup gave up waiting (MSM_TAIL_TIMEOUT).  The
        // host polls this word in pinned memory instead of waiting for the stream (api.hip: msm_wait_results): it is written AFTER
        // every result word of this bucket set is on its way (system-scope fence by every writer, workgroup barrier, then the flag)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t v = __hip_atomic_load(barrier + (uint64_t)w * PYR_BAR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flags + w, v & ~MSM_FLAG_PENDING_MASK, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// result[w][0] = sum(B) = A_{c-1}[0];  result[w][1 + j] = U_j,  j < c-1   (c entries of 256 B per window)
__global__ void msm_collect_kernel(const uint4* __restrict__ pyr_final, const uint4* __restrict__ pyr_prev,
                                   const uint4* __restrict__ odd_final, uint32_t nb, uint32_t c, uint4* __restrict__ result,
                                   uint32_t* __restrict__ flags) {
    const uint32_t w = blockIdx.x, j = threadIdx.x;  // j < c
    if (j < c) {
        const uint64_t wbase = (uint64_t)w * nb;
        const uint64_t cap = (uint64_t)gridDim.x * nb;
        // U_{c-2} is the odd entry of the two-entry level c - 2, still in the other pyramid buffer
        const uint4* src = j == 0 ? pyr_final + wbase : j == c - 1 ? pyr_prev + (wbase + 1) : odd_final + (wbase + odd_off(nb, j - 1));
        uint4* dst = result + ((uint64_t)w * c + j) * 16;
        for (int q = 0; q < 16; q++) dst[q] = src[q * cap];
    }
    __threadfence_system();
    __syncthreads();
    if (j == 0) __hip_atomic_store(flags + w, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // no barrier on this path: nothing can have timed out
}

// ---------------------------------------------------------------------------------------------------------
// Fixed-base multiplication P_i = k_i * G (Srs::new_from_secret, kzg/src/srs.rs:48-63, and the benchmark's
// base-point generator).  table[w * 255 + (d-1)] = d * 2^(8w) * G in affine form, 32 windows of 8 bits.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MSM_THREADS) void g1_fixed_base_kernel(const Fr* __restrict__ scalars, uint64_t n,
                                                                   const uint4* __restrict__ table,
                                                                   uint4* __restrict__ out_xy,
                                                                   uint8_t* __restrict__ out_inf) {
    const uint64_t i = (uint64_t)blockIdx.x * MSM_THREADS + threadIdx.x;
    if (i >= n) return;
    Fr k = from_mont(scalars[i]);
    G1Xyzz acc = G1Xyzz::infinity();
#pragma unroll 1
    for (int w = 0; w < 32; w++) {
        uint32_t limb = 0;
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (q == (w >> 2)) limb = k.l[q];
        const uint32_t d = (limb >> ((w & 3) * 8)) & 0xffu;
        if (d) {
            G1Affine p = G1Affine::load(table + ((uint64_t)w * 255 + (d - 1)) * 6);
            g1_madd(acc, p);
        }
    }
    G1Affine r;
    const bool inf = acc.is_inf();
    if (inf) {
        r.x = Fq::zero();
        r.y = Fq::zero();
    } else {
        Fq zi3 = fq_inverse_gcd(acc.zzz);  // safegcd (fq28_inv.hpp): ~33 k instructions against ~700 k for the Fermat power in this limb form
        Fq zi2 = sqr(zi3 * acc.zz);
        r.x = acc.x * zi2;
        r.y = acc.y * zi3;
    }
    r.store(out_xy + i * 6);
    if (out_inf) out_inf[i] = inf ? 1 : 0;
}

// Self-test of the two device inversions (zkp_selftest_fq_inverse_dev): one lane per element, raw limbs in and out.
// form 0: fq_inverse_gcd on the saturated form (12 x u32, Montgomery radix 2^384); form 1: fq28_inverse_gcd on the
// 28-bit-limb form (16 words per element: 14 limbs + 2 pad, Montgomery radix 2^392).
__global__ __launch_bounds__(MSM_THREADS) void fq_inverse_selftest_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                                         uint64_t n, int form) {
    const uint64_t i = (uint64_t)blockIdx.x * MSM_THREADS + threadIdx.x;
    if (i >= n) return;
    if (form == 0) {
        Fq a;
#pragma unroll
        for (int w = 0; w < 12; w++) a.l[w] = in[12 * i + w];
        const Fq r = fq_inverse_gcd(a);
#pragma unroll
        for (int w = 0; w < 12; w++) out[12 * i + w] = r.l[w];
    } else {
        const Fq28 a = Fq28::load(reinterpret_cast<const uint4*>(in) + 4 * i);
        fq28_inverse_gcd(a).store(reinterpret_cast<uint4*>(out) + 4 * i);
    }
}

}  // namespace zkp
