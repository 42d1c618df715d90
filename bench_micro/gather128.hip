// gather128.hip -- calibration of rocprofv3's FETCH_SIZE for the access pattern of msm_accumulate: every lane reads whole
// 128-byte records at random positions of a large table as eight 16-byte loads (A28::load), a known number of bytes.
// One kernel instantiation per footprint, so that the per-kernel rows of a --pmc pass are the sweep:
//     gather128<0>  2 GiB     gather128<1>  14 GiB     gather128<2>  26 GiB     gather128<3>  52 GiB
//     stream16<0>   the same number of bytes read as a coalesced 16-byte-per-lane stream (the guide's calibrated case)
// Every launch reads N = 2^26 records = 8.59 GB exactly (each lane K = 64 records, three waves per SIMD resident at a time).
//
// hipcc --offload-arch=gfx950 -O3 -std=c++17 bench_micro/gather128.hip -o bench_micro/gather128
// rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o g --output-format csv -- ./bench_micro/gather128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

template <int TAG>
__global__ __launch_bounds__(256) void gather128(const uint4* __restrict__ base, uint64_t nrec, uint32_t K, uint4* __restrict__ sink) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t h = (t + 1) * 0x9E3779B97F4A7C15ull + TAG;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t k = 0; k < K; k++) {
        h = h * 6364136223846793005ull + 1442695040888963407ull;
        const uint64_t r = (uint64_t)(((unsigned __int128)(h >> 16 << 16) * nrec) >> 64);  // uniform in [0, nrec)
        const uint4* p = base + r * 8;
        uint4 v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = p[q];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            acc.x ^= v[q].x; acc.y ^= v[q].y; acc.z ^= v[q].z; acc.w ^= v[q].w;
        }
        // a few hundred cycles of dependent ALU work between gathers, as the mixed add leaves between two of its loads
        uint32_t s = acc.x;
#pragma unroll 1
        for (int j = 0; j < 64; j++) s = s * 1664525u + 1013904223u;
        acc.y ^= s & 1u;
    }
    sink[t] = acc;
}

template <int TAG>
__global__ __launch_bounds__(256) void stream16(const uint4* __restrict__ base, uint64_t n16, uint4* __restrict__ sink) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (uint64_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint64_t i = t; i < n16; i += nt) {
        const uint4 v = base[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    sink[t] = acc;
}

int main(int argc, char** argv) {
    const int max_fp = argc > 1 ? atoi(argv[1]) : 3;
    const uint64_t GiB = 1ull << 30, fp_bytes[4] = {2 * GiB, 14 * GiB, 26 * GiB, 52 * GiB};
    const uint64_t lanes = 1ull << 20;  // 4096 workgroups: the 768 resident ones are replaced as they finish, like msm_accumulate's
    const uint32_t K = 64;
    uint4 *table, *sink;
    CK(hipMalloc(&table, fp_bytes[max_fp]));
    CK(hipMalloc(&sink, 16 * lanes));
    CK(hipMemset(table, 1, fp_bytes[max_fp]));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double bytes = 128.0 * lanes * K;
    for (int rep = 0; rep < 2; rep++)
        for (int fp = 0; fp <= max_fp; fp++) {
            const uint64_t nrec = fp_bytes[fp] / 128;
            CK(hipEventRecord(e0));
            switch (fp) {
                case 0: hipLaunchKernelGGL(gather128<0>, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, table, nrec, K, sink); break;
                case 1: hipLaunchKernelGGL(gather128<1>, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, table, nrec, K, sink); break;
                case 2: hipLaunchKernelGGL(gather128<2>, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, table, nrec, K, sink); break;
                default: hipLaunchKernelGGL(gather128<3>, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, table, nrec, K, sink); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("gather128<%d> footprint %5.1f GiB: %.0f MB in %.3f ms = %.2f TB/s (%.2f G records/s)\n", fp, fp_bytes[fp] / (double)GiB,
                            bytes / 1e6, ms, bytes / ms / 1e9, lanes * (double)K / ms / 1e6);
        }
    for (int rep = 0; rep < 2; rep++) {
        const uint64_t n16 = (uint64_t)(bytes / 16);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream16<0>, dim3(4096), dim3(256), 0, 0, table, n16, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("stream16<0>: %.0f MB in %.3f ms = %.2f TB/s\n", bytes / 1e6, ms, bytes / ms / 1e9);
    }
    return 0;
}
