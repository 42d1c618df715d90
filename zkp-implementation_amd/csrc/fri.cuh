// fri.cuh -- the FRI commitment step that follows the Goldilocks NTT: SHA-256 Merkle trees over the decimal strings of the
// evaluations (fri/src/hasher.rs:14-36, fri/src/merkle_tree.rs:42-63) and the gather of query decommitments
// (fri/src/prover.rs:84-134, merkle_tree.rs:84-107).
//
// hash(x)       = SHA-256(Display(x))                 -> F::from_le_bytes_mod_order(digest)
// hash_slice(v) = SHA-256(Display(v0) || Display(v1))   (no separator)
// Display = decimal of the canonical integer, leading zeros trimmed: at most 20 digits per element, so a leaf (<= 20
// bytes) and a pair (<= 40 bytes) always fit ONE 64-byte SHA-256 block.  One lane per hash; the variable-length message is
// assembled in a 68-byte LDS slot per lane (byte stores at data-dependent offsets stay out of scratch memory), then the
// 64 rounds run in registers (~2500 VALU instructions per hash, no memory traffic: the kernel is ALU-bound).
// A workgroup of 256 lanes covers 1024 adjacent nodes and climbs up to 10 levels above them (2 inside each lane, 8 through
// LDS), so a tree of 2^21 leaves takes three launches instead of twenty-two.
#pragma once
#include "ff.cuh"

namespace zkp {

constexpr int MERKLE_BLOCK = 256;      // lanes per workgroup
constexpr int MERKLE_SPAN = 1024;      // input nodes per workgroup (4 per lane)
constexpr int MERKLE_MAX_LEVELS = 11;  // leaf hashes + 2 in-lane levels + 8 LDS levels
constexpr int SHA_SLOT = 68;           // bytes of LDS per lane (64 + 4: consecutive slots start on different banks)

__device__ __constant__ const uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

ZKP_DEV uint32_t rotr32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }

// one compression of the initial state with the block w[0..15] (big-endian words); out = digest words
ZKP_DEV void sha256_single_block(uint32_t w[16], uint32_t out[8]) {
    uint32_t a = 0x6a09e667, b = 0xbb67ae85, c = 0x3c6ef372, d = 0xa54ff53a, e = 0x510e527f, f = 0x9b05688c, g = 0x1f83d9ab,
             h = 0x5be0cd19;
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
            const uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const uint32_t t1 = h + (rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25)) + ((e & f) ^ (~e & g)) + SHA256_K[i] + w[i & 15];
        const uint32_t t2 = (rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    out[0] = a + 0x6a09e667; out[1] = b + 0xbb67ae85; out[2] = c + 0x3c6ef372; out[3] = d + 0xa54ff53a;
    out[4] = e + 0x510e527f; out[5] = f + 0x9b05688c; out[6] = g + 0x1f83d9ab; out[7] = h + 0x5be0cd19;
}

// decimal digits of x, most significant first, leading zeros trimmed (zero -> nothing, or "0" when zero_as_0);
// returns the number of bytes written
ZKP_DEV int gl_write_decimal(uint64_t x, uint8_t* dst, bool zero_as_0) {
    const uint32_t hi = (uint32_t)(x / 10000000000ull);  // < 1.85e9
    const uint64_t lo = x - (uint64_t)hi * 10000000000ull;  // < 1e10
    uint32_t part[4] = {hi / 100000u, hi % 100000u, (uint32_t)(lo / 100000u), (uint32_t)(lo % 100000u)};
    uint8_t d[20];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        uint32_t v = part[p];
#pragma unroll
        for (int k = 4; k >= 0; k--) {
            d[5 * p + k] = (uint8_t)(v % 10u);
            v /= 10u;
        }
    }
    int nz = 0;
    bool lead = true;
#pragma unroll
    for (int i = 0; i < 20; i++) {
        lead = lead && d[i] == 0;
        nz += lead ? 1 : 0;
    }
    if (nz == 20 && zero_as_0) nz = 19;
#pragma unroll
    for (int i = 0; i < 20; i++)
        if (i >= nz) dst[i - nz] = (uint8_t)('0' + d[i]);
    return 20 - nz;
}

// F::from_le_bytes_mod_order of the digest: sum of the four little-endian 64-bit limbs l_k 2^(64k), with
// 2^64 = EPS, 2^128 = -2^32, 2^192 = 1 (mod p).  Returns the canonical value.
ZKP_DEV Gl sha_digest_to_gl(const uint32_t dg[8]) {
    uint64_t l[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        l[k] = (uint64_t)__builtin_bswap32(dg[2 * k]) | (uint64_t)__builtin_bswap32(dg[2 * k + 1]) << 32;
    Gl acc{l[0] >= Gl::MOD ? l[0] - Gl::MOD : l[0]};
    acc = acc + Gl{l[1]} * Gl{Gl::EPS};            // operator* accepts any 64-bit operand
    acc = acc - Gl{l[2]} * Gl{1ull << 32};
    acc = acc + Gl{l[3] >= Gl::MOD ? l[3] - Gl::MOD : l[3]};
    return acc;
}

// hash of one or two canonical elements; `slot` is this lane's LDS scratch
ZKP_DEV Gl gl_hash_elems(uint64_t a, uint64_t b, bool two, uint8_t* slot, bool zero_as_0) {
    uint32_t* sw = reinterpret_cast<uint32_t*>(slot);
#pragma unroll
    for (int i = 0; i < 16; i++) sw[i] = 0;
    int len = gl_write_decimal(a, slot, zero_as_0);
    if (two) len += gl_write_decimal(b, slot + len, zero_as_0);
    slot[len] = 0x80;
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = __builtin_bswap32(sw[i]);
    w[15] = (uint32_t)len * 8;  // message bits (< 2^32), big-endian length field
    uint32_t dg[8];
    sha256_single_block(w, dg);
    return sha_digest_to_gl(dg);
}

ZKP_DEV uint64_t gl_canonical_from_mont(uint64_t m) { return (Gl{m} * Gl{0xfffffffe00000001ull}).v; }  // * 2^-64
ZKP_DEV uint64_t gl_mont_from_canonical(uint64_t c) { return (Gl{c} * Gl{Gl::EPS}).v; }               // * 2^64

struct MerkleLaunch {
    const uint64_t* in;   // leaves (leaf_mode) or the nodes of the level below out[0]
    uint64_t n_in;
    int leaf_mode;        // 1: out[0][i] = hash(in[i]); 0: `in` is a node level, out[0] is the level above it
    int levels;           // levels written by this launch (<= MERKLE_MAX_LEVELS in leaf mode, one fewer otherwise)
    int zero_as_0;
    uint64_t* out[MERKLE_MAX_LEVELS];
};

// Workgroup b owns input nodes [1024 b, 1024 b + 1024).  Every lane first walks its own 4-input subtree serially (4 leaf
// hashes, 2 parents, 1 grandparent: full lanes, no barrier), then the 256 grandparents climb up to 8 more levels through
// LDS.  Level s above the input starts at (1024 b) >> s, exact for s <= 10.
__global__ __launch_bounds__(MERKLE_BLOCK) void merkle_levels_kernel(MerkleLaunch p) {
    __shared__ uint64_t cur[MERKLE_BLOCK];
    __shared__ uint32_t slots32[MERKLE_BLOCK * SHA_SLOT / 4];
    const int tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * MERKLE_SPAN;
    uint8_t* slot = reinterpret_cast<uint8_t*>(slots32) + tid * SHA_SLOT;
    const bool z0 = p.zero_as_0 != 0;
    const uint32_t span = (uint32_t)(p.n_in - base < MERKLE_SPAN ? p.n_in - base : MERKLE_SPAN);
    const uint32_t mine = span > 4u * tid ? (span - 4u * tid < 4u ? span - 4u * tid : 4u) : 0u;  // valid inputs of this lane
    int lvl = 0;  // next entry of p.out
    uint64_t v[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < mine; k++) v[k] = gl_canonical_from_mont(p.in[base + 4 * tid + k]);
    if (p.leaf_mode) {
        for (uint32_t k = 0; k < mine; k++) {
            v[k] = gl_hash_elems(v[k], 0, false, slot, z0).v;
            p.out[0][base + 4 * tid + k] = gl_mont_from_canonical(v[k]);
        }
        lvl = 1;
    }
    uint32_t have = mine;  // nodes this lane holds at the current level
    int s = 1;             // level distance from the input level
    for (; s <= 2 && lvl < p.levels; s++, lvl++) {
        const uint32_t next = (have + 1) / 2;
        for (uint32_t j = 0; j < next; j++) {
            const bool two = 2 * j + 1 < have;
            v[j] = gl_hash_elems(v[2 * j], two ? v[2 * j + 1] : 0, two, slot, z0).v;
            p.out[lvl][(base >> s) + (uint64_t)tid * (4u >> s) + j] = gl_mont_from_canonical(v[j]);
        }
        have = next;
    }
    if (lvl >= p.levels) return;  // uniform: depends on the launch parameters only
    uint32_t count = (span + 3) / 4;  // grandparents in this workgroup
    if (tid < (int)count) cur[tid] = v[0];
    __syncthreads();
    for (; lvl < p.levels; s++, lvl++) {
        const uint32_t next = (count + 1) / 2;
        uint64_t h = 0;
        if (tid < (int)next) {
            const bool two = 2 * tid + 1 < (int)count;
            h = gl_hash_elems(cur[2 * tid], two ? cur[2 * tid + 1] : 0, two, slot, z0).v;
            p.out[lvl][(base >> s) + tid] = gl_mont_from_canonical(h);
        }
        __syncthreads();
        if (tid < (int)next) cur[tid] = h;
        __syncthreads();
        count = next;
    }
}

struct FriLayerRef {
    const uint64_t* evals;
    const uint64_t* nodes;  // all Merkle levels, concatenated
    uint64_t size;          // domain size of the layer (a power of two)
};
// One workgroup per (query, layer): writes index, eval, sym_eval, path[depth], sym_path[depth] (prover.rs:100-121).
// rec_off[q * layers + l] = word offset of the record inside `out`.
__global__ __launch_bounds__(64) void fri_gather_kernel(const FriLayerRef* layers, uint32_t n_layers, const uint64_t* challenges,
                                                        const uint64_t* rec_off, uint64_t* out) {
    const uint32_t q = blockIdx.x, l = blockIdx.y;
    const FriLayerRef L = layers[l];
    const uint64_t idx = challenges[q] % L.size, sym = (idx + L.size / 2) % L.size;
    uint32_t depth = 0;
    while ((1ull << depth) < L.size) depth++;
    uint64_t* rec = out + rec_off[(uint64_t)q * n_layers + l];
    for (uint32_t t = threadIdx.x; t < 3 + 2 * depth; t += 64) {
        uint64_t v;
        if (t == 0) v = idx;
        else if (t == 1) v = L.evals[idx];
        else if (t == 2) v = L.evals[sym];
        else {
            const uint32_t i = (t - 3) % depth;
            const uint64_t leaf = (t - 3) < depth ? idx : sym;
            const uint64_t off = 2 * L.size - 2 * (L.size >> i);  // start of level i for a power-of-two tree
            v = L.nodes[off + ((leaf >> i) ^ 1)];
        }
        rec[t] = v;
    }
}

}  // namespace zkp
