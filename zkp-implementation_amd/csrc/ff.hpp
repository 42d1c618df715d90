// ff.hpp -- prime-field arithmetic for gfx950 (CDNA4): 32-bit-limb Montgomery for BLS12-381 Fr / Fq,
// and the special-form Goldilocks prime.
//
// CDNA4 has 32x32 integer multipliers only (v_mad_u64_u32 = 32x32+64 -> 64), so every field element is
// held as N 32-bit limbs in VGPRs, fully unrolled.  In memory the elements are exactly the arkworks 0.4
// in-memory form the reference passes around (kzg/src/types.rs:6-10, fri/src/fields/goldilocks.rs:4-8):
// Montgomery residues as little-endian u64 limbs -- a little-endian u64[N/2] and a u32[N] have the same bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zkp {

#define ZKP_DEV __device__ __forceinline__
#define ZKP_HD __host__ __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// Field parameter packs (moduli from ark-bls12-381 0.4.0 FrConfig / FqConfig; values re-derived in
// tests/model/bigmodel.py and SURVEY.md Appendix A).
// ---------------------------------------------------------------------------------------------
struct FrParams {
    static constexpr int N = 8;
    // r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    static constexpr uint32_t MOD[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                        0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    static constexpr uint32_t INV = 0xffffffffu;  // -r^-1 mod 2^32
    // R = 2^256 mod r
    static constexpr uint32_t ONE[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                        0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
    // R^2 mod r
    static constexpr uint32_t R2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                       0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
};

struct FqParams {
    static constexpr int N = 12;
    // p = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    static constexpr uint32_t MOD[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                         0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    static constexpr uint32_t INV = 0xfffcfffdu;  // -p^-1 mod 2^32
    // R = 2^384 mod p
    static constexpr uint32_t ONE[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                         0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
    // R^2 mod p
    static constexpr uint32_t R2[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                        0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
};

// ---------------------------------------------------------------------------------------------
// N x 32-bit Montgomery field element.
// ---------------------------------------------------------------------------------------------
template <class P>
struct Fp {
    static constexpr int N = P::N;
    alignas(16) uint32_t l[N];

    static ZKP_DEV Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    static ZKP_DEV Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::ONE[i];
        return r;
    }
    static ZKP_DEV Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::R2[i];
        return r;
    }
    ZKP_DEV bool is_zero() const {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < N; i++) x |= l[i];
        return x == 0;
    }
    ZKP_DEV bool operator==(const Fp& o) const {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < N; i++) x |= l[i] ^ o.l[i];
        return x == 0;
    }

    // 16-byte vector loads/stores (element size is a multiple of 16 B: Fr 32 B, Fq 48 B)
    static ZKP_DEV Fp load(const void* p) {
        Fp r;
        const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
        for (int i = 0; i < N / 4; i++) {
            uint4 v = q[i];
            r.l[4 * i] = v.x; r.l[4 * i + 1] = v.y; r.l[4 * i + 2] = v.z; r.l[4 * i + 3] = v.w;
        }
        return r;
    }
    ZKP_DEV void store(void* p) const {
        uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
        for (int i = 0; i < N / 4; i++) q[i] = make_uint4(l[4 * i], l[4 * i + 1], l[4 * i + 2], l[4 * i + 3]);
    }
};

// r = a + b (no reduction), returns carry-out
template <int N>
ZKP_DEV uint32_t add_limbs(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        c += (uint64_t)a[i] + b[i];
        r[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
}
// r = a - b, returns borrow (0/1)
template <int N>
ZKP_DEV uint32_t sub_limbs(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        c += (int64_t)a[i] - (int64_t)b[i];
        r[i] = (uint32_t)c;
        c >>= 32;  // arithmetic: 0 or -1
    }
    return (uint32_t)(c & 1);
}

// conditional final subtraction: x in [0, 2p) (+ optional carry bit) -> [0, p)
template <class P>
ZKP_DEV void reduce_once(uint32_t* x, uint32_t carry) {
    constexpr int N = P::N;
    uint32_t t[N];
    uint32_t br = sub_limbs<N>(t, x, P::MOD);
    bool use = (carry != 0) | (br == 0);
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = use ? t[i] : x[i];
}

template <class P>
ZKP_DEV Fp<P> operator+(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint32_t c = add_limbs<P::N>(r.l, a.l, b.l);
    reduce_once<P>(r.l, c);
    return r;
}
template <class P>
ZKP_DEV Fp<P> operator-(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint32_t br = sub_limbs<P::N>(r.l, a.l, b.l);
    uint32_t t[P::N];
    add_limbs<P::N>(t, r.l, P::MOD);
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = br ? t[i] : r.l[i];
    return r;
}
template <class P>
ZKP_DEV Fp<P> neg(const Fp<P>& a) {
    Fp<P> r;
    sub_limbs<P::N>(r.l, P::MOD, a.l);
    bool z = a.is_zero();
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = z ? 0u : r.l[i];
    return r;
}
template <class P>
ZKP_DEV Fp<P> dbl(const Fp<P>& a) { return a + a; }

// Montgomery product a*b*R^-1 mod p, CIOS over 32-bit limbs.  Every inner step is one
// v_mad_u64_u32 (32x32 + 64) plus a carry add; MOD/INV fold to scalar constants.
template <class P>
ZKP_DEV Fp<P> mont_mul(const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            c += (uint64_t)a.l[j] * b.l[i] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[N];
        t[N] = (uint32_t)c;
        t[N + 1] = (uint32_t)(c >> 32);
        uint32_t m = t[0] * P::INV;
        c = ((uint64_t)m * P::MOD[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            c += (uint64_t)m * P::MOD[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[N];
        t[N - 1] = (uint32_t)c;
        t[N] = t[N + 1] + (uint32_t)(c >> 32);
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    reduce_once<P>(r.l, t[N]);
    return r;
}
using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

// Fr products: defined in fr29.hpp (through 9 limbs of 29 bits: 400 instructions instead of the 626 the saturated CIOS form above
// compiles to -- 120 multiply-adds, 135 64-bit additions and 319 moves); mont_mul<FrParams> stays as the statement of the operation.
ZKP_DEV Fr operator*(const Fr& a, const Fr& b);

// Fq products go through ONE out-of-line body: a fully inlined XYZZ mixed add is ~90 KB of code (10 products of
// ~1.2k instructions), more than the instruction cache two CUs share, so the curve code calls this instead.
// Arguments and result travel in VGPRs (ext_vector types; a struct return would go through scratch memory).
typedef uint32_t u32x12 __attribute__((ext_vector_type(12)));
static __device__ __noinline__ u32x12 fq_mul_outlined(u32x12 a, u32x12 b) {
    Fq x, y;
#pragma unroll
    for (int i = 0; i < 12; i++) { x.l[i] = a[i]; y.l[i] = b[i]; }
    Fq z = mont_mul<FqParams>(x, y);
    u32x12 r;
#pragma unroll
    for (int i = 0; i < 12; i++) r[i] = z.l[i];
    return r;
}
ZKP_DEV Fq operator*(const Fq& a, const Fq& b) {
    u32x12 x, y;
#pragma unroll
    for (int i = 0; i < 12; i++) { x[i] = a.l[i]; y[i] = b.l[i]; }
    u32x12 z = fq_mul_outlined(x, y);
    Fq r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = z[i];
    return r;
}
template <class P>
ZKP_DEV Fp<P> sqr(const Fp<P>& a) { return a * a; }

// Montgomery residue -> canonical integer limbs (multiply by 1)
template <class P>
ZKP_DEV Fp<P> from_mont(const Fp<P>& a) {
    Fp<P> o = Fp<P>::zero();
    o.l[0] = 1;
    return a * o;
}
template <class P>
ZKP_DEV Fp<P> to_mont(const Fp<P>& a) { return a * Fp<P>::r2(); }

// a^e for a 64-bit exponent
template <class P>
ZKP_DEV Fp<P> pow_u64(Fp<P> a, uint64_t e) {
    Fp<P> r = Fp<P>::one();
    while (e) {
        if (e & 1) r = r * a;
        a = sqr(a);
        e >>= 1;
    }
    return r;
}

// ---------------------------------------------------------------------------------------------
// Goldilocks p = 2^64 - 2^32 + 1 (fri/src/fields/goldilocks.rs:5).  The reference keeps elements as
// Montgomery residues (R = 2^64); an NTT is linear, so multiplying residues by CANONICAL twiddles with
// plain modular multiplication yields exactly the Montgomery-form outputs arkworks would produce.  The
// special form of p makes the plain reduction (2^64 = 2^32 - 1, 2^96 = -1) cheaper than a Montgomery one.
// ---------------------------------------------------------------------------------------------
struct Gl {
    uint64_t v;
    static constexpr uint64_t MOD = 0xffffffff00000001ull;
    static constexpr uint64_t EPS = 0xffffffffull;  // 2^64 mod p
    static ZKP_HD Gl zero() { return Gl{0}; }
    static ZKP_HD Gl one() { return Gl{1}; }  // canonical one (twiddle domain)
    static ZKP_DEV Gl load(const void* p) { return Gl{*reinterpret_cast<const uint64_t*>(p)}; }
    ZKP_DEV void store(void* p) const { *reinterpret_cast<uint64_t*>(p) = v; }
    ZKP_HD bool is_zero() const { return v == 0; }
    ZKP_HD bool operator==(const Gl& o) const { return v == o.v; }
};
// Canonical in, canonical out.  x - p = x + EPS (mod 2^64), so "subtract p when the sum reaches it" is one more 64-bit
// addition whose carry-out is the comparison (two carries instead of compare + subtract).
ZKP_HD Gl operator+(const Gl& a, const Gl& b) {
    const uint64_t s = a.v + b.v;   // a, b < p  =>  a + b < 2p < 2^65
    const uint64_t t = s + Gl::EPS;  // = a + b - p (mod 2^64)
    return Gl{(s < a.v || t < s) ? t : s};
}
ZKP_HD Gl operator-(const Gl& a, const Gl& b) {
    uint64_t d = a.v - b.v;
    if (a.v < b.v) d -= Gl::EPS;  // + p (mod 2^64)
    return Gl{d};
}
ZKP_HD Gl neg(const Gl& a) { return Gl{a.v ? Gl::MOD - a.v : 0}; }
ZKP_HD Gl gl_reduce128(uint64_t lo, uint64_t hi) {
    // x = lo + c2 * 2^64 + c3 * 2^96  ==  lo - c3 + c2 * (2^32 - 1)  (mod p)
    const uint32_t c2 = (uint32_t)hi, c3 = (uint32_t)(hi >> 32);
    uint64_t y = lo - c3;
    if (lo < c3) y -= Gl::EPS;  // the wrap added 2^64 == EPS (mod p): take it back (y >= 2^64 - 2^32, cannot wrap again)
    uint64_t z = y + (uint64_t)c2 * (uint32_t)Gl::EPS;  // one v_mad_u64_u32
    if (z < y) z += Gl::EPS;    // wrapped value <= 2^64 - 2^33: no second wrap
    const uint64_t w = z + Gl::EPS;  // z - p (mod 2^64), carries iff z >= p
    return Gl{w < z ? w : z};
}
// Schoolbook on 32-bit halves: four v_mad_u64_u32, each addend fits ((2^32-1)^2 + 2 (2^32-1) < 2^64).  Any 64-bit inputs.
ZKP_HD Gl operator*(const Gl& a, const Gl& b) {
    const uint32_t a0 = (uint32_t)a.v, a1 = (uint32_t)(a.v >> 32), b0 = (uint32_t)b.v, b1 = (uint32_t)(b.v >> 32);
    const uint64_t t0 = (uint64_t)a0 * b0;
    const uint64_t t1 = (uint64_t)a0 * b1 + (t0 >> 32);
    const uint64_t t2 = (uint64_t)a1 * b0 + (uint32_t)t1;
    const uint64_t t3 = (uint64_t)a1 * b1 + (t1 >> 32) + (t2 >> 32);
    return gl_reduce128((t2 << 32) | (uint32_t)t0, t3);
}
ZKP_HD Gl sqr(const Gl& a) { return a * a; }
ZKP_HD Gl pow_u64(Gl a, uint64_t e) {
    Gl r = Gl::one();
    while (e) {
        if (e & 1) r = r * a;
        a = a * a;
        e >>= 1;
    }
    return r;
}

// In-kernel clock stamps (zkp_profile_clock_read): while profiling is on, wave 0 of every workgroup reads s_memtime (one tick per
// shader cycle) and s_memrealtime (the constant 100 MHz reference) when it starts and when it ends and adds both deltas to a record
// of their own -- sum(d cycles) / sum(d ref) x 100 MHz is the shader clock the chip held UNDER THIS KERNEL'S LOAD, weighted by wave
// lifetime (MI355X_MICROARCH.md, DVFS: the clock under load is what differs from box to box, not the cycle count).  Two scalar
// reads and two atomics per workgroup that lives for 100 us or more; with a null record (profiling off) nothing executes.
struct ClkRec {
    unsigned long long cycles, ref, waves, pad;
};
ZKP_DEV void clk_begin(const ClkRec* c, uint64_t& t0, uint64_t& r0) {
    if (!c) return;
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the stamps are back before the body's own scalar loads are counted
}
ZKP_DEV void clk_end(ClkRec* c, uint64_t t0, uint64_t r0) {
    if (!c) return;
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    // The stamps must be BACK before the branch below.  They are dead on the path of the waves that do not hold thread 0, and in a kernel
    // larger than a short branch reaches (msm_accumulate: 250 KB of code) LLVM's branch relaxation then takes the stamp's own SGPR pair for
    // the long jump (s_getpc_b64 / s_add_u32 / s_addc_u32 / s_setpc_b64) with the scalar-memory result still in flight: a late return
    // overwrites the high half of the jump target with the high half of the 100 MHz counter and the wave fetches from there -- an
    // intermittent "memory access fault" while profiling, found in round 5 (profiles/r05_l_profile_mode_fault.md;
    // tools/check_smem_long_branch.py scans every kernel of the library for the pattern, tests/test_build_isa.py runs it).
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    if (threadIdx.x == 0) {
        atomicAdd(&c->cycles, (unsigned long long)(t1 - t0));
        atomicAdd(&c->ref, (unsigned long long)(r1 - r0));
        atomicAdd(&c->waves, 1ull);
    }
}


}  // namespace zkp
