// mfma_redc.hip -- can the i8 matrix cores take the constant-operand half of the Montgomery product?  
//
// A Montgomery product on the 28-bit-limb form (csrc/fq28.hpp) is 196 multiply-adds for x*y and 196 for the reduction
// (m = T N' mod R word by word, T + m p).  Only m * p has a CONSTANT operand, i.e. matrix shape: with the numbers of a wave as the
// columns of B (7-bit digits of m, 56 per number = its 14 limbs spread to bytes) and Toeplitz(p) as a constant A,
//     C[row][n] = sum_k pdigit[row - k] * mdigit_n[k]      (v_mfma_i32_32x32x32_i8, columns = 32 numbers, two batches per wave)
// are the column sums of m_n * p at weight 2^(7 row).  To get there m must exist in full before the product (no word-serial
// interleave): m = T_low * N' mod R is a triangular 14 x 14 product on the VALU (105 multiply-adds), the digits must be spread into
// bytes, two half-wave exchanges put lane n's digits where the B operand wants them, and the 32-bit column sums must be recomposed into
// 28-bit limbs and travel back (v_permlane32_swap) before they meet the high half of x*y.
//
// This program builds that hybrid product for real (bit-identical to fq28_mul_inline, checked on every lane), and times
//   KIND 0  the product as the library has it (fq28_mul_inline)
//   KIND 1  the hybrid: VALU x*y + m, MFMA m*p, VALU recomposition
//   KIND 2  the hybrid's VALU work alone (the MFMAs replaced by register moves)
//   KIND 3  the hybrid's ten MFMAs alone
// as chains of dependent products, one wave per workgroup, W waves per SIMD (the occupancy each kernel's registers allow).
//
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkp-implementation_amd/csrc bench_micro/mfma_redc.hip -o bench_micro/mfma_redc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fq28.hpp"
using namespace zkp;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// N' = -p^-1 mod 2^392 in 28-bit limbs
__device__ __constant__ uint32_t NPRIME[14] = {0xffcfffdu, 0xf3fffcfu, 0x113e889u, 0xdb92d9du, 0xb48286au, 0xf0c8e30u, 0xc16ef2eu,
                                               0x8eb2db4u, 0x9ecca0eu, 0x68cf581u, 0x316fee2u, 0xfc9468bu, 0x106feaau, 0xa0ceb06u};

// Toeplitz(p) tiles (row block, k block) the high half of m * p needs: (1,0) (1,1) (2,0) (2,1) (3,1); rows 0..31 only feed limbs
// 0..7, and of the low half only limb 13 is needed (see carry13 below).  tiles[t][lane] = the lane's 16 bytes of A.
struct ATiles { v4i t[5]; };

ZKP_DEV uint32_t spread7(uint32_t x) {  // 4 x 7 bits -> 4 bytes
    const uint32_t y = (x & 0x3fffu) | ((x & 0x0fffc000u) << 2);
    return (y & 0x007f007fu) | ((y & 0x3f803f80u) << 1);
}

// limb value of four consecutive 7-bit columns (signed i32 sums < 2^20 each, all non-negative here)
ZKP_DEV uint64_t compose4(int c0, int c1, int c2, int c3) {
    const uint32_t lo = (uint32_t)c0 + ((uint32_t)c1 << 7), hi = (uint32_t)c2 + ((uint32_t)c3 << 7);
    return (uint64_t)lo + ((uint64_t)hi << 14);
}

template <bool USE_MFMA>
ZKP_DEV Fq28 hybrid_mul(const Fq28& a, const Fq28& b, const ATiles& A) {
    // 1. x * y: 27 column sums
    uint64_t col[2 * NL28];
#pragma unroll
    for (int k = 0; k < 2 * NL28 - 1; k++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = (k < NL28 ? 0 : k - NL28 + 1); i <= (k < NL28 ? k : NL28 - 1); i++) acc += (uint64_t)a.l[i] * b.l[k - i];
        col[k] = acc;
    }
    col[2 * NL28 - 1] = 0;
    // 2. low half normalised: t = T mod R
    uint32_t t[NL28];
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < NL28; k++) {
        c += col[k];
        t[k] = (uint32_t)c & MASK28;
        c >>= 28;
    }
    const uint64_t carry14 = c;
    // 3. m = t * N' mod R
    uint32_t m[NL28];
    c = 0;
#pragma unroll
    for (int k = 0; k < NL28; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) c += (uint64_t)t[i] * NPRIME[k - i];
        m[k] = (uint32_t)c & MASK28;
        c >>= 28;
    }
    // 4. digits of m as bytes; B operands of the two half-wave batches
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < NL28; i++) w[i] = spread7(m[i]);
    w[14] = w[15] = 0;
    // lanes 0..31 keep words 0..3 / 8..11 (k half h = 0 of their own number) and receive words 0..3 / 8..11 of number n + 32;
    // lanes 32..63 keep words 4..7 / 12..15 of their own number and receive words 4..7 / 12..15 of number n
#pragma unroll
    for (int q = 0; q < 4; q++) {
        auto r0 = __builtin_amdgcn_permlane32_swap(w[q], w[4 + q], false, false);      // vdst = w[q] (upper half out), src = w[4+q] (lower half out)
        w[q] = r0[0]; w[4 + q] = r0[1];
        auto r1 = __builtin_amdgcn_permlane32_swap(w[8 + q], w[12 + q], false, false);
        w[8 + q] = r1[0]; w[12 + q] = r1[1];
    }
    // after the swaps: {w0..3} = B(k block 0) of batch 0, {w4..7} = B(k block 0) of batch 1, {w8..11} / {w12..15} the same for k block 1
    const v4i b00 = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]}, b10 = {(int)w[4], (int)w[5], (int)w[6], (int)w[7]};
    const v4i b01 = {(int)w[8], (int)w[9], (int)w[10], (int)w[11]}, b11 = {(int)w[12], (int)w[13], (int)w[14], (int)w[15]};
    v16i z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0;
    v16i c1[2], c2[2], c3[2];  // row blocks 1..3 of the two batches
    if (USE_MFMA) {
        c1[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[0], b00, z, 0, 0, 0);
        c1[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[1], b01, c1[0], 0, 0, 0);
        c2[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[2], b00, z, 0, 0, 0);
        c2[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[3], b01, c2[0], 0, 0, 0);
        c3[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[4], b01, z, 0, 0, 0);
        c1[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[0], b10, z, 0, 0, 0);
        c1[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[1], b11, c1[1], 0, 0, 0);
        c2[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[2], b10, z, 0, 0, 0);
        c2[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[3], b11, c2[1], 0, 0, 0);
        c3[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[4], b11, z, 0, 0, 0);
    } else {  // same data dependences, no matrix instruction
#pragma unroll
        for (int i = 0; i < 16; i++) {
            c1[0][i] = b00[i & 3] & 0x7f; c2[0][i] = b01[i & 3] & 0x7f; c3[0][i] = b00[(i + 1) & 3] & 0x7f;
            c1[1][i] = b10[i & 3] & 0x7f; c2[1][i] = b11[i & 3] & 0x7f; c3[1][i] = b10[(i + 1) & 3] & 0x7f;
        }
    }
    // 5. column sums -> limb values.  Lane (n, h) holds, per batch, limbs 8 Rb + 2 g + h (g = reg >> 2): needed are limbs 13..27
    //    = row block 1 groups 2, 3; row block 2 groups 0..3; row block 3 groups 0, 1  -> eight values per batch
    uint64_t v[2][8];
#pragma unroll
    for (int bt = 0; bt < 2; bt++) {
        v[bt][0] = compose4(c1[bt][8], c1[bt][9], c1[bt][10], c1[bt][11]);     // limb 12 + h
        v[bt][1] = compose4(c1[bt][12], c1[bt][13], c1[bt][14], c1[bt][15]);   // limb 14 + h
#pragma unroll
        for (int g = 0; g < 4; g++) v[bt][2 + g] = compose4(c2[bt][4 * g], c2[bt][4 * g + 1], c2[bt][4 * g + 2], c2[bt][4 * g + 3]);  // 16 + 2g + h
        v[bt][6] = compose4(c3[bt][0], c3[bt][1], c3[bt][2], c3[bt][3]);       // 24 + h
        v[bt][7] = compose4(c3[bt][4], c3[bt][5], c3[bt][6], c3[bt][7]);       // 26 + h
    }
    // 6. back to "lane = number": batch 0's values in lanes 32..63 (odd limbs of number n) <-> batch 1's in lanes 0..31 (even limbs of n + 32)
#pragma unroll
    for (int q = 0; q < 8; q++) {
        uint32_t a0 = (uint32_t)v[0][q], a1 = (uint32_t)(v[0][q] >> 32), b0 = (uint32_t)v[1][q], b1 = (uint32_t)(v[1][q] >> 32);
        auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        v[0][q] = (uint64_t)r0[0] | ((uint64_t)r1[0] << 32);   // even limbs 12 + 2q of this lane's number
        v[1][q] = (uint64_t)r0[1] | ((uint64_t)r1[1] << 32);   // odd limbs 13 + 2q
    }
    // 7. T_low + (m p)_low is a multiple of R: the carry into limb 13 is whatever makes limb 13 vanish (it is < 2^15)
    const uint64_t s13 = (uint64_t)t[13] + v[1][0];
    const uint64_t c13in = (0 - s13) & MASK28;
    c = ((s13 + c13in) >> 28) + carry14;
    Fq28 r;
#pragma unroll
    for (int k = 0; k < NL28; k++) {
        const int limb = 14 + k;
        c += col[limb] + ((limb & 1) ? v[1][(limb - 13) / 2] : v[0][(limb - 12) / 2]);
        if (k < NL28 - 1) {
            r.l[k] = (uint32_t)c & MASK28;
            c >>= 28;
        } else {
            r.l[k] = (uint32_t)c;
        }
    }
    return r;
}

constexpr int ITERS = 512;

template <int KIND>
__global__ __launch_bounds__(64) void k_chain(const uint4* __restrict__ in, uint4* __restrict__ out, const v4i* __restrict__ tiles,
                                              unsigned long long* __restrict__ cyc, int iters) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    Fq28 x = Fq28::load(in + (uint64_t)i * 8), y = Fq28::load(in + (uint64_t)i * 8 + 4);
    ATiles A;
#pragma unroll
    for (int t = 0; t < 5; t++) A.t[t] = tiles[t * 64 + threadIdx.x];
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (KIND == 3) {
        v16i acc[3];
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[q][e] = 0;
        const v4i b0 = {(int)x.l[0], (int)x.l[1], (int)x.l[2], (int)x.l[3]}, b1 = {(int)y.l[0], (int)y.l[1], (int)y.l[2], (int)y.l[3]};
#pragma unroll 1
        for (int it = 0; it < iters; it++) {  // ten per product, three accumulators
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[0], b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[1], b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[2], b0, acc[2], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[3], b1, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[4], b0, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[0], b1, acc[2], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[1], b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[2], b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[3], b0, acc[2], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.t[4], b1, acc[0], 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 14; e++) x.l[e] = (uint32_t)(acc[0][e] ^ acc[1][e] ^ acc[2][e]);
    } else {
#pragma unroll 1
        for (int it = 0; it < iters; it++) {
            if (KIND == 0) x = fq28_mul_inline(x, y);
            if (KIND == 1) x = hybrid_mul<true>(x, y, A);
            if (KIND == 2) x = hybrid_mul<false>(x, y, A);
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    x.store(out + (uint64_t)i * 4);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static uint32_t pdigit(int j) { return (j < 0 || j >= 56) ? 0u : (Fq28C::MOD[j / 4] >> (7 * (j % 4))) & 0x7fu; }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : ITERS;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    // A tiles: lane l (row 32 Rb + (l & 31), half h = l >> 5), byte j <-> k = 32 Kb + 16 h + j  (B uses the same slots: any k order
    // inside a block gives the same sums as long as A and B agree)
    const int tile_rb[5] = {1, 1, 2, 2, 3}, tile_kb[5] = {0, 1, 0, 1, 1};
    std::vector<uint32_t> tiles(5 * 64 * 4);
    for (int t = 0; t < 5; t++)
        for (int l = 0; l < 64; l++)
            for (int wd = 0; wd < 4; wd++) {
                uint32_t word = 0;
                for (int byte = 0; byte < 4; byte++) {
                    const int j = 4 * wd + byte, row = 32 * tile_rb[t] + (l & 31), k = 32 * tile_kb[t] + 16 * (l >> 5) + j;
                    word |= pdigit(row - k) << (8 * byte);
                }
                tiles[(t * 64 + l) * 4 + wd] = word;
            }
    v4i* d_tiles;
    (void)hipMalloc(&d_tiles, tiles.size() * 4);
    (void)hipMemcpy(d_tiles, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice);
    const char* names[4] = {"fq28_mul_inline (library)", "hybrid: VALU x*y + m, MFMA m*p", "hybrid, VALU part only", "hybrid, the ten MFMAs only"};
    void* funcs[4] = {(void*)k_chain<0>, (void*)k_chain<1>, (void*)k_chain<2>, (void*)k_chain<3>};
    std::vector<std::vector<uint32_t>> results(4);
    for (int waves_cap = 0; waves_cap < 2; waves_cap++) {  // 0: the occupancy the registers allow; 1: everybody at the hybrid's occupancy
        int occ_h = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_h, k_chain<1>, 64, 0);
        for (int kind = 0; kind < 4; kind++) {
            int occ = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)funcs[kind], 64, 0);
            int per_simd = occ / 4;
            if (per_simd > 8) per_simd = 8;
            if (waves_cap) per_simd = per_simd < occ_h / 4 ? per_simd : occ_h / 4;
            if (per_simd < 1) per_simd = 1;
            const int blocks = cus * 4 * per_simd;
            const size_t n = (size_t)blocks * 64;
            std::vector<uint32_t> h_in(n * 32, 0);
            uint64_t s = 0x9e3779b97f4a7c15ull;
            for (size_t e = 0; e < n; e++)
                for (int half = 0; half < 2; half++)
                    for (int l = 0; l < 14; l++) {
                        s = s * 6364136223846793005ull + 1442695040888963407ull;
                        h_in[e * 32 + half * 16 + l] = (uint32_t)(s >> 33) & (l == 13 ? 0x1fffu : MASK28);  // < 2^377: tight
                    }
            uint4 *d_in, *d_out;
            unsigned long long* d_cyc;
            (void)hipMalloc(&d_in, n * 128);
            (void)hipMalloc(&d_out, n * 64);
            (void)hipMalloc(&d_cyc, blocks * 8);
            (void)hipMemcpy(d_in, h_in.data(), n * 128, hipMemcpyHostToDevice);
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            auto launch = [&]() {
                if (kind == 0) hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(64), 0, 0, d_in, d_out, d_tiles, d_cyc, iters);
                if (kind == 1) hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(64), 0, 0, d_in, d_out, d_tiles, d_cyc, iters);
                if (kind == 2) hipLaunchKernelGGL(k_chain<2>, dim3(blocks), dim3(64), 0, 0, d_in, d_out, d_tiles, d_cyc, iters);
                if (kind == 3) hipLaunchKernelGGL(k_chain<3>, dim3(blocks), dim3(64), 0, 0, d_in, d_out, d_tiles, d_cyc, iters);
            };
            launch();
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            for (int rep = 0; rep < 5; rep++) launch();
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            ms /= 5;
            std::vector<unsigned long long> cyc(blocks);
            (void)hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
            double mean = 0;
            for (auto cy : cyc) mean += (double)cy;
            mean /= blocks;
            if (!waves_cap) {
                results[kind].resize(256 * 16);
                (void)hipMemcpy(results[kind].data(), d_out, 256 * 64, hipMemcpyDeviceToHost);  // the first 256 lanes see the same inputs in every kind
            }
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, (const void*)funcs[kind]);
            const double prods = (double)n * iters;
            printf("%-34s %d waves/SIMD (%3d VGPRs)  %8.3f ms  %7.1f cycles per wave-product  %6.2f SIMD-cycles per wave-product  "
                   "%.3e products/s\n", names[kind], per_simd, fa.numRegs, ms, mean / iters, mean / iters / per_simd, prods / (ms * 1e-3));
            (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_cyc);
        }
        printf("\n");
    }
    // bit-exactness: the hybrid's chain of products against the library's, every limb of the first 256 lanes
    size_t bad = 0;
    for (size_t e = 0; e < 256 * 16; e++) bad += results[0][e] != results[1][e];
    printf("hybrid == fq28_mul_inline on %d lanes x %d chained products: %s (%zu differing words)\n", 256, iters, bad ? "NO" : "yes", bad);
    return bad ? 1 : 0;
}
