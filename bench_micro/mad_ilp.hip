// mad_ilp.hip -- v_mad_u64_u32 throughput vs (waves per SIMD) x (independent chains per wave) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 16384
template <int CHAINS>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    extern __shared__ uint32_t lds[];
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc[c] = a + c;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int rep = 0; rep < 16 / CHAINS; rep++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++)
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) r ^= (uint32_t)acc[c] ^ (uint32_t)(acc[c] >> 32);
    if (threadIdx.x == 0) lds[0] = r;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + lds[0];
}
template <int CHAINS>
void run(int waves_per_simd, uint32_t* d_out) {
    // 256-thread blocks = 1 wave per SIMD each; limit blocks per CU through dynamic LDS
    size_t lds = 160 * 1024 / waves_per_simd - 1024;
    hipFuncSetAttribute((const void*)k<CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int blocks = 256 * waves_per_simd * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), lds, 0, d_out, 7u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), lds, 0, d_out, 9u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * ITERS * 16;
    printf("waves/SIMD %d chains %2d : %7.3f ms  %.1f lane-mads/clk/CU @2.4GHz\n", waves_per_simd, CHAINS, ms, ops / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 4 * 256 * 4);
    for (int w : {1, 2, 4, 8}) { run<1>(w, d); run<2>(w, d); run<4>(w, d); run<8>(w, d); }
    return 0;
}
