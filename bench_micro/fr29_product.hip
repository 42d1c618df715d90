// fr29_product.hip -- cost of one Fr29 Montgomery product on gfx950 in the forms the transform passes can use, at 1 / 2 / 4 waves
// per SIMD: (a) the plain C product as the compiler schedules it (235 instructions, 156 multiply-adds), (b) two products per step,
// plain C, (c) fr29_mul2: two strict multiply-add chains interleaved in one asm statement (206 instructions per product).
// SIMD-cycles per product = launch time x clock x SIMDs / products; the multiply-add floor is 162 x (64 lanes / measured lane-mads
// per SIMD per cycle).    hipcc --offload-arch=gfx950 -O3 -std=c++17 bench_micro/fr29_product.hip -o bench_micro/fr29_product
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../zkp-implementation_amd/csrc/fr29.hpp"
using namespace zkp;
#define ITERS 2048

template <int MODE>
__global__ __launch_bounds__(256) void k(Fr29* out, const Fr29* in, unsigned long long* clk) {
    extern __shared__ uint32_t lds[];
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    Fr29 a0 = in[threadIdx.x], a1 = in[256 + threadIdx.x], w0 = in[512 + (threadIdx.x & 63)], w1 = in[640 + (threadIdx.x & 63)];
    for (int it = 0; it < ITERS; it++) {
        Fr29 r0v, r1v;
        if (MODE == 0) {
            r0v = a0 * w0;
            r1v = a1;
        } else if (MODE == 1) {
            r0v = a0 * w0;
            r1v = a1 * w1;
        } else {
            fr29_mul2(a0, w0, a1, w1, r0v, r1v);
        }
        // keep the limb bounds of the contract (a < 2^31): results are tight already
        a0 = r0v;
        a1 = r1v;
    }
    Fr29 r = a0 + a1;
    if (threadIdx.x == 0) lds[0] = r.l[0];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        atomicAdd(&clk[0], (unsigned long long)(t1 - t0));
        atomicAdd(&clk[1], (unsigned long long)(r1 - r0));
    }
}

template <int MODE>
void run(int waves_per_simd, Fr29* d_out, const Fr29* d_in, unsigned long long* d_clk, int cus) {
    size_t lds = 160 * 1024 / waves_per_simd - 1024;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int blocks = cus * waves_per_simd * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d_out, d_in, d_clk);
    hipDeviceSynchronize();
    hipMemset(d_clk, 0, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d_out, d_in, d_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2];
    hipMemcpy(c, d_clk, 16, hipMemcpyDeviceToHost);
    const double mhz = c[1] ? 100.0 * (double)c[0] / (double)c[1] : 0.0;
    const double products = (double)blocks * 256 * ITERS * (MODE == 0 ? 1 : 2);
    const double simd_cycles = ms * 1e-3 * mhz * 1e6 * cus * 4;
    const char* names[3] = {"plain C, one product per step ", "plain C, two products per step", "fr29_mul2 (asm, interleaved)   "};
    printf("%s  waves/SIMD %d : %7.3f ms at %4.0f MHz  %6.1f SIMD-cycles per wave-product\n", names[MODE], waves_per_simd, ms, mhz,
           simd_cycles / (products / 64));
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    Fr29 *d_out, *d_in;
    unsigned long long* d_clk;
    hipMalloc(&d_out, sizeof(Fr29) * 256 * cus * 32);
    hipMalloc(&d_in, sizeof(Fr29) * 1024);
    hipMalloc(&d_clk, 16);
    Fr29 h[1024];
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < 1024; i++)
        for (int j = 0; j < 9; j++) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            h[i].l[j] = (uint32_t)(x >> 35) & (j == 8 ? 0x3fffffu : MASK29);
        }
    hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4}) {
        run<0>(w, d_out, d_in, d_clk, cus);
        run<1>(w, d_out, d_in, d_clk, cus);
        run<2>(w, d_out, d_in, d_clk, cus);
    }
    // bit-identity of the two forms on this data
    Fr29 *o1 = new Fr29[256], *o2 = new Fr29[256];
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 1024, 0, d_out, d_in, d_clk);
    hipMemcpy(o1, d_out, sizeof(Fr29) * 256, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 1024, 0, d_out, d_in, d_clk);
    hipMemcpy(o2, d_out, sizeof(Fr29) * 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; i++)
        for (int j = 0; j < 9; j++) bad += o1[i].l[j] != o2[i].l[j];
    printf("asm form %s the plain C form after %d chained products per lane\n", bad ? "DIFFERS FROM" : "is bit-identical to", ITERS);
    return bad != 0;
}
