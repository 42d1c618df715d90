#include "host_ff.hpp"
#include <cstdio>
#include <chrono>
#include <random>
using namespace zkp::host;
template <class E, int N> int run(const char* name) {
    std::mt19937_64 g(7);
    int bad = 0;
    E acc = E::one();
    for (int it = 0; it < 20000; it++) {
        E x;
        for (int i = 0; i < N; i++) x.l[i] = g();
        x.l[N - 1] &= (1ull << 60) - 1;  // below the modulus
        if (it == 0) x = E::one();
        if (it == 1) x = E::zero();
        if (it == 2) x = E::zero() - E::one();
        if (it == 3) x = E::from_u64(2);
        E a = x.inverse(), b = x.inverse_fermat();
        if (!(a == b)) bad++;
        if (!x.is_zero() && !((a * x) == E::one())) bad++;
    }
    {   // non-canonical limbs (El::load does not reduce; raw ABI inputs get here): p itself and 2p are zero, p + 5 is 5
        const uint64_t* p = E::M().p;
        E z, z2, f = E::from_u64(5), f2 = f;
        uint64_t c = 0, c2 = 0;
        for (int i = 0; i < N; i++) {
            z.l[i] = p[i];
            unsigned __int128 d = (unsigned __int128)p[i] * 2 + c2;
            z2.l[i] = (uint64_t)d;
            c2 = (uint64_t)(d >> 64);
            unsigned __int128 s = (unsigned __int128)f.l[i] + p[i] + c;
            f2.l[i] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
        if (!z.inverse().is_zero()) bad++;
        if (c2 == 0 && !z2.inverse().is_zero()) bad++;
        if (c == 0 && !(f2.inverse() == f.inverse())) bad++;
    }
    E x = E::from_u64(12345);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; i++) { x = x.inverse() + E::one(); }
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; i++) { x = x.inverse_fermat() + E::one(); }
    auto t2 = std::chrono::steady_clock::now();
    printf("%s: mismatches %d  euclid %.2f us  fermat %.2f us (%llu)\n", name, bad, std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000,
           std::chrono::duration<double, std::micro>(t2 - t1).count() / 2000, (unsigned long long)x.l[0]);
    return bad;
}
int main() { return run<HFr, 4>("Fr") + run<HFq, 6>("Fq"); }
