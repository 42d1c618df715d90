// pyr_timeline.hip -- where does a level of the bucket reduction spend its time?  (msm_pyramid_kernel delivers 60 % of
// msm_accumulate's product rate at the same three waves per SIMD.)
//
// The level kernel of csrc/msm.hpp (one lane = one XYZZ + XYZZ addition, operands and result in plane-major arrays, g1_28_add_stream)
// with two stamps per workgroup: s_memrealtime (the chip-wide 100 MHz counter) when wave 0 starts and when it ends.  From the
// stamps of one launch: when workgroups start (dispatch ramp, second round), how long they live, when the last one ends, and how
// that compares with the kernel's duration between HIP events.  Variants:
//   MODE 0  one addition per lane, `adds` lanes (what the library launches)
//   MODE 1  persistent: 3 workgroups per CU, grid-stride loop over the additions
//   MODE 2  MODE 0 without memory: operands made in registers from the lane index, result xor-reduced into one word (issue only)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkp-implementation_amd/csrc bench_micro/pyr_timeline.hip -o bench_micro/pyr_timeline
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "g1_28.hpp"
using namespace zkp;

constexpr int THREADS = 256;

template <int MODE>
__global__ __launch_bounds__(THREADS) void level_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t adds, uint64_t cap_in,
                                                        uint64_t cap_out, unsigned long long* __restrict__ tl) {
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (MODE == 0) {
        const uint32_t s = blockIdx.x * THREADS + threadIdx.x;
        if (s < adds) g1_28_add_stream(in + 2 * (uint64_t)s, in + 2 * (uint64_t)s + 1, out + s, cap_in);
    } else if (MODE == 1) {
        for (uint32_t s = blockIdx.x * THREADS + threadIdx.x; s < adds; s += gridDim.x * THREADS)
            g1_28_add_stream(in + 2 * (uint64_t)s, in + 2 * (uint64_t)s + 1, out + s, cap_in);
    } else {
        const uint32_t s = blockIdx.x * THREADS + threadIdx.x;
        if (s < adds) {
            X28 a, b;
            Fq28* fa = &a.x;
            Fq28* fb = &b.x;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int l = 0; l < NL28; l++) {
                    fa[q].l[l] = (s * 2654435761u + 40503u * (q * 16 + l)) & (l == 13 ? 0x1fffu : MASK28);
                    fb[q].l[l] = (s * 2246822519u + 30011u * (q * 16 + l) + 7u) & (l == 13 ? 0x1fffu : MASK28);
                }
            g1_28_add(a, b);
            uint32_t x = 0;
#pragma unroll
            for (int l = 0; l < NL28; l++) x ^= a.x.l[l] ^ a.y.l[l] ^ a.zz.l[l] ^ a.zzz.l[l];
            if (x == 0x12345678u) out[s].x = x;  // keeps the arithmetic alive, never true in practice
        }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        tl[4 * blockIdx.x + 0] = r0;
        tl[4 * blockIdx.x + 1] = r1;
        tl[4 * blockIdx.x + 2] = c1 - c0;
        tl[4 * blockIdx.x + 3] = 0;
    }
    (void)cap_out;
}

static double pct(std::vector<double>& v, double p) {
    std::sort(v.begin(), v.end());
    return v[(size_t)(p * (v.size() - 1))];
}

int main(int argc, char** argv) {
    const uint32_t adds = argc > 1 ? (uint32_t)atoi(argv[1]) : 262144;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const uint64_t cap_in = 2ull * adds, cap_out = adds;
    uint4 *d_in, *d_out;
    unsigned long long* d_tl;
    (void)hipMalloc(&d_in, cap_in * 256);
    (void)hipMalloc(&d_out, cap_in * 256);  // g1_28_add_stream uses ONE stride for operands and result (the library's arrays share their capacity)
    std::vector<uint32_t> h(cap_in * 64);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (uint64_t q = 0; q < 16; q++)             // plane-major: chunk q of entry e at [q * cap + e]; chunk = 4 words, words 14, 15 of a field element are pad
        for (uint64_t e = 0; e < cap_in; e++)
            for (int w = 0; w < 4; w++) {
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                const int limb = (int)(q % 4) * 4 + w;
                h[(q * cap_in + e) * 4 + w] = limb >= 14 ? 0u : (uint32_t)(s >> 33) & (limb == 13 ? 0x1fffu : MASK28);
            }
    (void)hipMemcpy(d_in, h.data(), cap_in * 256, hipMemcpyHostToDevice);
    const int max_blocks = std::max<int>((adds + THREADS - 1) / THREADS, prop.multiProcessorCount * 3);
    (void)hipMalloc(&d_tl, (size_t)max_blocks * 32);
    const char* names[3] = {"one add per lane (library)", "persistent, 3 workgroups per CU", "registers only (no memory)"};
    for (int mode = 0; mode < 3; mode++) {
        const int blocks = mode == 1 ? prop.multiProcessorCount * 3 : (int)((adds + THREADS - 1) / THREADS);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(level_kernel<0>, dim3(blocks), dim3(THREADS), 0, 0, d_in, d_out, adds, cap_in, cap_out, d_tl);
            if (mode == 1) hipLaunchKernelGGL(level_kernel<1>, dim3(blocks), dim3(THREADS), 0, 0, d_in, d_out, adds, cap_in, cap_out, d_tl);
            if (mode == 2) hipLaunchKernelGGL(level_kernel<2>, dim3(blocks), dim3(THREADS), 0, 0, d_in, d_out, adds, cap_in, cap_out, d_tl);
        };
        for (int w = 0; w < 3; w++) launch();
        (void)hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            (void)hipEventRecord(e0);
            launch();
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms);
        }
        // back-to-back launches (what a level costs inside a stream of dependent launches)
        (void)hipEventRecord(e0);
        for (int rep = 0; rep < 10; rep++) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms10 = 0;
        (void)hipEventElapsedTime(&ms10, e0, e1);
        std::vector<unsigned long long> tl((size_t)blocks * 4);
        (void)hipMemcpy(tl.data(), d_tl, (size_t)blocks * 32, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int b = 0; b < blocks; b++) {
            t0 = std::min(t0, tl[4 * b]);
            t1 = std::max(t1, tl[4 * b + 1]);
        }
        std::vector<double> start, end, life, cyc;
        for (int b = 0; b < blocks; b++) {
            start.push_back((tl[4 * b] - t0) * 0.01);
            end.push_back((tl[4 * b + 1] - t0) * 0.01);
            life.push_back((tl[4 * b + 1] - tl[4 * b]) * 0.01);
            cyc.push_back((double)tl[4 * b + 2]);
        }
        hipFuncAttributes fa;
        (void)hipFuncGetAttributes(&fa, mode == 0 ? (const void*)level_kernel<0> : mode == 1 ? (const void*)level_kernel<1> : (const void*)level_kernel<2>);
        printf("%-34s %u adds, %d workgroups, %d VGPRs: one launch %.1f us (events), %.1f us each back to back; first start -> last end %.1f us\n",
               names[mode], adds, blocks, fa.numRegs, best * 1e3, ms10 * 100, (t1 - t0) * 0.01);
        printf("    workgroup start  us: p0 %.1f  p25 %.1f  p50 %.1f  p75 %.1f  p90 %.1f  p100 %.1f\n", pct(start, 0), pct(start, .25), pct(start, .5),
               pct(start, .75), pct(start, .9), pct(start, 1));
        printf("    workgroup end    us: p0 %.1f  p25 %.1f  p50 %.1f  p75 %.1f  p90 %.1f  p100 %.1f\n", pct(end, 0), pct(end, .25), pct(end, .5),
               pct(end, .75), pct(end, .9), pct(end, 1));
        printf("    wave-0 lifetime  us: p0 %.1f  p50 %.1f  p100 %.1f   shader kcycles: p0 %.1f  p50 %.1f  p100 %.1f  (clock %.2f GHz at p50)\n", pct(life, 0),
               pct(life, .5), pct(life, 1), pct(cyc, 0) / 1e3, pct(cyc, .5) / 1e3, pct(cyc, 1) / 1e3, pct(cyc, .5) / pct(life, .5) / 1e3);
    }
    return 0;
}
