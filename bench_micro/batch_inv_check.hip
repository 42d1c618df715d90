// Debug aid: Montgomery's trick over thread-private Fq28 arrays, as g1_expand_planes_kernel uses it.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkp-implementation_amd/csrc bench_micro/batch_inv_check.hip -o bench_micro/bic
#include <hip/hip_runtime.h>
#include <cstdio>
#include "msm.hpp"
using namespace zkp;
__device__ bool same(const Fq28& x, const Fq28& y) {
    uint32_t d = 0;
    for (int i = 0; i < NL28; i++) d |= x.l[i] ^ y.l[i];
    return d == 0;
}
__device__ bool is_one(const Fq28& x) { return tight_is_zero_mod_p(normalise(sub4(x, Fq28::one())) * Fq28::one()); }
__global__ void k(uint32_t n, uint32_t* flags, Fq28* ga, Fq28* gpre) {
    Fq28 a[29], pre[29];
    Fq28 g = Fq28::one();
    g = g + g + g;  // 3
    Fq28 v = normalise(g);
    for (uint32_t s = 1; s < n; s++) {
        v = v * v + Fq28::one();
        v = normalise(v) * Fq28::one() * Fq28::one();  // tight
        a[s] = v;
        pre[s] = s == 1 ? v : pre[s - 1] * v;
        ga[s] = a[s];
        gpre[s] = pre[s];
    }
    Fq28 inv = fq28_inverse_gcd(pre[n - 1]);
    for (uint32_t s = n - 1; s >= 1; s--) {
        uint32_t f = 0;
        f |= same(a[s], ga[s]) ? 1 : 0;
        f |= same(pre[s], gpre[s]) ? 2 : 0;
        f |= is_one(inv * pre[s]) ? 4 : 0;       // inv == 1/pre[s] on entry
        const Fq28 zi = s > 1 ? inv * pre[s - 1] : inv;
        inv = inv * a[s];
        f |= is_one(zi * a[s]) ? 8 : 0;
        if (s > 1) f |= is_one(inv * pre[s - 1]) ? 16 : 0;  // inv == 1/pre[s-1] on exit
        if (s > 1) f |= is_one(pre[s - 1] * a[s] * fq28_inverse_gcd(pre[s])) ? 32 : 0;  // pre[s] == pre[s-1] a[s]
        flags[s] = f;
    }
}
int main() {
    uint32_t* d; Fq28 *ga, *gp;
    (void)hipMalloc(&d, 4 * 32); (void)hipMemset(d, 0xff, 4 * 32);
    (void)hipMalloc(&ga, sizeof(Fq28) * 32); (void)hipMalloc(&gp, sizeof(Fq28) * 32);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 16u, d, ga, gp);
    uint32_t h[32]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int s = 1; s < 16; s++) printf("s=%d flags=%u\n", s, h[s]);
    return 0;
}
