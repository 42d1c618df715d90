// Is the bucket-reduction pyramid limited by its 256-byte-per-lane (AoS) record accesses?  Copy-only skeleton of level 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void aos(const uint4* __restrict__ in, uint4* __restrict__ out, uint4* __restrict__ seed, uint32_t half) {
    uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= half) return;
    uint4 x[16], y[16];
#pragma unroll
    for (int q = 0; q < 16; q++) { x[q] = in[(uint64_t)(2 * s) * 16 + q]; y[q] = in[(uint64_t)(2 * s + 1) * 16 + q]; }
#pragma unroll
    for (int q = 0; q < 16; q++) {
        seed[(uint64_t)s * 16 + q] = y[q];
        out[(uint64_t)s * 16 + q] = make_uint4(x[q].x ^ y[q].x, x[q].y + y[q].y, x[q].z ^ y[q].z, x[q].w + y[q].w);
    }
}
__global__ __launch_bounds__(256) void soa(const uint4* __restrict__ in, uint4* __restrict__ out, uint4* __restrict__ seed, uint32_t half) {
    uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= half) return;
    const uint64_t N = 2ull * half;
    uint4 x[16], y[16];
#pragma unroll
    for (int q = 0; q < 16; q++) { x[q] = in[q * N + 2 * s]; y[q] = in[q * N + 2 * s + 1]; }
#pragma unroll
    for (int q = 0; q < 16; q++) {
        seed[(uint64_t)q * half + s] = y[q];
        out[(uint64_t)q * half + s] = make_uint4(x[q].x ^ y[q].x, x[q].y + y[q].y, x[q].z ^ y[q].z, x[q].w + y[q].w);
    }
}
int main() {
    const uint32_t nb = 1u << 19, half = nb / 2;
    uint4 *in, *out, *seed;
    CHK(hipMalloc(&in, (size_t)nb * 256)); CHK(hipMalloc(&out, (size_t)half * 256)); CHK(hipMalloc(&seed, (size_t)half * 256));
    CHK(hipMemset(in, 1, (size_t)nb * 256));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipEventRecord(a));
            for (int i = 0; i < 20; i++) {
                if (mode == 0) aos<<<half / 256, 256>>>(in, out, seed, half); else soa<<<half / 256, 256>>>(in, out, seed, half);
            }
            CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
            float ms; CHK(hipEventElapsedTime(&ms, a, b));
            printf("%s: %.1f us per launch, %.2f TB/s (268 MB)\n", mode ? "SoA" : "AoS", ms * 1e3 / 20, 268.4e6 / (ms * 1e-3 / 20) / 1e12);
        }
    }
    return 0;
}
