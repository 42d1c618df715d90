// fq28_product.hip -- cost of one Fq28 Montgomery product (14 x 28-bit limbs) on gfx950: the compiler's schedule of the C form (486
// instructions: 392 multiply-adds + 25 v_lshl_add_u64 that join re-associated column sums + masks and shifts) against strict
// multiply-add chains written as ONE asm statement (461), one product or two interleaved, at 1..3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 bench_micro/fq28_product.hip -o bench_micro/fq28_product
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../zkp-implementation_amd/csrc/fq28.hpp"
using namespace zkp;
#define ITERS 1024
__device__ __forceinline__ Fq28 mul_asm1(const Fq28& a, const Fq28& b) {
    Fq28 r;
#include "../zkp-implementation_amd/csrc/fq28_mul_asm.inc"
    return r;
}
__device__ __forceinline__ void mul_asm2(const Fq28& a0, const Fq28& b0, const Fq28& a1, const Fq28& b1, Fq28& r0, Fq28& r1) {
#include "../zkp-implementation_amd/csrc/fq28_mul2x_asm.inc"
}
template <int MODE>
__global__ __launch_bounds__(256) void k(Fq28* out, const Fq28* in, unsigned long long* clk) {
    extern __shared__ uint32_t lds[];
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    Fq28 a0 = in[threadIdx.x], a1 = in[256 + threadIdx.x], w0 = in[512 + (threadIdx.x & 63)], w1 = in[640 + (threadIdx.x & 63)];
    for (int it = 0; it < ITERS; it++) {
        Fq28 x, y;
        if (MODE == 0) { x = a0 * w0; y = a1 * w1; }
        else if (MODE == 1) { x = mul_asm1(a0, w0); y = mul_asm1(a1, w1); }
        else mul_asm2(a0, w0, a1, w1, x, y);
        a0 = x;
        a1 = y;
    }
    Fq28 r = a0 + a1;
    if (threadIdx.x == 0) lds[0] = r.l[0];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { atomicAdd(&clk[0], (unsigned long long)(t1 - t0)); atomicAdd(&clk[1], (unsigned long long)(r1 - r0)); }
}
template <int MODE>
void run(int waves, Fq28* d_out, const Fq28* d_in, unsigned long long* d_clk, int cus) {
    size_t lds = 160 * 1024 / waves - 1024;
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int blocks = cus * waves * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d_out, d_in, d_clk);
    (void)hipDeviceSynchronize();
    (void)hipMemset(d_clk, 0, 16);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d_out, d_in, d_clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; (void)hipMemcpy(c, d_clk, 16, hipMemcpyDeviceToHost);
    const double mhz = c[1] ? 100.0 * (double)c[0] / (double)c[1] : 0.0;
    const double products = (double)blocks * 256 * ITERS * 2;
    const char* names[3] = {"plain C (compiler's schedule)   ", "asm, one strict chain           ", "asm, two chains interleaved     "};
    printf("%s waves/SIMD %d : %7.3f ms at %4.0f MHz  %7.1f SIMD-cycles per wave-product\n", names[MODE], waves, ms, mhz,
           ms * 1e-3 * mhz * 1e6 * cus * 4 / (products / 64));
}
int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    Fq28 *d_out, *d_in; unsigned long long* d_clk;
    (void)hipMalloc(&d_out, sizeof(Fq28) * 256 * cus * 16); (void)hipMalloc(&d_in, sizeof(Fq28) * 1024); (void)hipMalloc(&d_clk, 16);
    static Fq28 h[1024];
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < 1024; i++)
        for (int j = 0; j < 14; j++) { x = x * 6364136223846793005ull + 1442695040888963407ull; h[i].l[j] = (uint32_t)(x >> 36) & (j == 13 ? 0xffffu : MASK28); }
    (void)hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
    for (int w : {1, 2, 3}) { run<0>(w, d_out, d_in, d_clk, cus); run<1>(w, d_out, d_in, d_clk, cus); run<2>(w, d_out, d_in, d_clk, cus); }
    static Fq28 o[3][256];
    for (int m = 0; m < 3; m++) {
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 1024, 0, d_out, d_in, d_clk);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 1024, 0, d_out, d_in, d_clk);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 1024, 0, d_out, d_in, d_clk);
        (void)hipMemcpy(o[m], d_out, sizeof(Fq28) * 256, hipMemcpyDeviceToHost);
    }
    int bad = 0;
    for (int i = 0; i < 256; i++) for (int j = 0; j < 14; j++) bad += (o[0][i].l[j] != o[1][i].l[j]) + (o[0][i].l[j] != o[2][i].l[j]);
    printf("asm forms %s the plain C form after %d chained products per lane\n", bad ? "DIFFER FROM" : "are bit-identical to", ITERS);
    return bad != 0;
}
