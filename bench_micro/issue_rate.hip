// issue_rate.hip -- measures the issue rate of the integer / fp64 instructions that bound 256/381-bit Montgomery
// arithmetic on gfx950, to calibrate the integer roofline of DESIGN.md.  Standalone: hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 65536
#define CHAINS 8

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(uint32_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[CHAINS];
    uint32_t lo[CHAINS];
    double d[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { acc[c] = a + c; lo[c] = b + c; d[c] = (double)(a + c); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (KIND == 0) {  // v_mad_u64_u32
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(lo[c]) : "vcc");
            } else if (KIND == 1) {  // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo[c]) : "v"(a));
            } else if (KIND == 2) {  // v_mul_hi_u32
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo[c]) : "v"(a));
            } else if (KIND == 3) {  // v_add_co_u32 + v_addc_co_u32 pair (64-bit add)
                asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo[c]) : "v"(a) : "vcc");
            } else if (KIND == 4) {  // v_lshl_add_u64
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[c]) : "v"(acc[(c + 1) % CHAINS]));
            } else if (KIND == 5) {  // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[c]) : "v"(d[(c + 1) % CHAINS]));
            } else if (KIND == 6) {  // v_mad_u32_u24
                asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(lo[c]) : "v"(a));
            } else if (KIND == 7) {  // v_add_u32 (full-rate reference)
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[c]) : "v"(a));
            } else if (KIND == 8) {  // v_mul_hi_u32_u24
                asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo[c]) : "v"(a));
            } else if (KIND == 10) {  // v_add_u32 in VOP3 (e64) encoding
                asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(lo[c]) : "v"(a));
            } else if (KIND == 11) {  // v_mov_b32
                asm volatile("v_mov_b32 %0, %1" : "=v"(lo[c]) : "v"(lo[(c + 1) % CHAINS]));
            } else if (KIND == 12) {  // v_addc_co_u32 alone (carry in and out through vcc)
                asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo[c]) : "v"(a) : "vcc");
            } else if (KIND == 13) {  // v_mad_u64_u32 with an SGPR-pair carry destination
                asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(lo[c]) : "s20", "s21");
            } else if (KIND == 14) {  // v_lshrrev_b64
                asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(acc[c]));
            } else if (KIND == 15) {  // v_and_b32
                asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo[c]) : "v"(a));
            } else if (KIND == 16) {  // v_add3_u32 (VOP3)
                asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(lo[c]) : "v"(a));
            } else if (KIND == 17) {  // v_alignbit_b32 (VOP3)
                asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(lo[c]) : "v"(a));
            } else if (KIND == 9) {  // v_mad_u64_u32 + v_addc_co_u32 (product-scanning MAC)
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[c]), "+v"(lo[c]) : "v"(a), "v"(b) : "vcc");
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) r ^= (uint32_t)acc[c] ^ (uint32_t)(acc[c] >> 32) ^ lo[c] ^ (uint32_t)d[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
void run(const char* name, int ops_per_item, uint32_t* d_out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 7u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 9u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double lane_ops = (double)blocks * 256 * ITERS * CHAINS * ops_per_item;
    double rate = lane_ops / (ms * 1e-3);
    // per CU per clock at 2.4 GHz, 256 CUs
    printf("%-28s %8.3f ms  %.3e lane-ops/s  = %.1f lanes/clk/CU @2.4GHz\n", name, ms, rate, rate / 256 / 2.4e9);
}

int main() {
    uint32_t* d_out;
    int blocks = 256 * 8;  // 8 blocks of 256 per CU = full occupancy
    hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    run<7>("v_add_u32", 1, d_out, blocks);
    run<0>("v_mad_u64_u32", 1, d_out, blocks);
    run<9>("v_mad_u64_u32+v_addc", 1, d_out, blocks);
    run<1>("v_mul_lo_u32", 1, d_out, blocks);
    run<2>("v_mul_hi_u32", 1, d_out, blocks);
    run<3>("v_add_co+v_addc pair", 1, d_out, blocks);
    run<4>("v_lshl_add_u64", 1, d_out, blocks);
    run<5>("v_fma_f64", 1, d_out, blocks);
    run<6>("v_mad_u32_u24", 1, d_out, blocks);
    run<8>("v_mul_hi_u32_u24", 1, d_out, blocks);
    run<10>("v_add_u32_e64", 1, d_out, blocks);
    run<11>("v_mov_b32", 1, d_out, blocks);
    run<12>("v_addc_co_u32", 1, d_out, blocks);
    run<13>("v_mad_u64_u32 sgpr carry", 1, d_out, blocks);
    run<14>("v_lshrrev_b64", 1, d_out, blocks);
    run<15>("v_and_b32", 1, d_out, blocks);
    run<16>("v_add3_u32", 1, d_out, blocks);
    run<17>("v_alignbit_b32", 1, d_out, blocks);
    run<7>("v_add_u32 (again)", 1, d_out, blocks);
    hipFree(d_out);
    return 0;
}
