// fri.hpp -- the FRI commitment step that follows the Goldilocks NTT: SHA-256 Merkle trees over the decimal strings of the
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
#include "ff.hpp"

namespace zkp {

constexpr int MERKLE_BLOCK = 256;      // lanes per workgroup
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
// gfx950's three-input boolean: one instruction for the xor of the three rotations, for Ch and for Maj
ZKP_DEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
ZKP_DEV uint32_t sha_ch(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xca); }   // e ? f : g
ZKP_DEV uint32_t sha_maj(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xe8); }

// st = compress(st, block w[0..15] of big-endian words)
ZKP_DEV void sha256_compress(uint32_t st[8], uint32_t w[16]) {
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
            const uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const uint32_t t1 = h + xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25)) + sha_ch(e, f, g) + SHA256_K[i] + w[i & 15];
        const uint32_t t2 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22)) + sha_maj(a, b, c);
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// one compression of the initial state with the block w[0..15] (big-endian words); out = digest words
ZKP_DEV void sha256_single_block(uint32_t w[16], uint32_t out[8]) {
    uint32_t a = 0x6a09e667, b = 0xbb67ae85, c = 0x3c6ef372, d = 0xa54ff53a, e = 0x510e527f, f = 0x9b05688c, g = 0x1f83d9ab,
             h = 0x5be0cd19;
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
            const uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const uint32_t t1 = h + xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25)) + sha_ch(e, f, g) + SHA256_K[i] + w[i & 15];
        const uint32_t t2 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22)) + sha_maj(a, b, c);
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
    int levels;           // levels written by this launch (<= 1 + ipl_log + 8 in leaf mode, one fewer otherwise)
    int ipl_log;          // log2 of the inputs per lane: 2 for large levels (throughput), 0 for small ones (shortest chain)
    int zero_as_0;
    uint64_t* out[MERKLE_MAX_LEVELS];
};

// Workgroup b owns input nodes [span b, span (b + 1)), span = 256 << ipl_log.  Every lane first walks its own subtree of
// 2^ipl_log inputs serially (for 4 inputs: 4 leaf hashes, 2 parents, 1 grandparent: full lanes, no barrier), then the 256
// lane results climb up to 8 more levels through LDS.  Level s above the input starts at (span b) >> s.  A tree is a chain of
// depth + 1 dependent hashes (~5 us each): large levels take 4 inputs per lane, small ones 1 so that the chain stays short.
__global__ __launch_bounds__(MERKLE_BLOCK) void merkle_levels_kernel(MerkleLaunch p) {
    __shared__ uint64_t cur[MERKLE_BLOCK];
    __shared__ uint32_t slots32[MERKLE_BLOCK * SHA_SLOT / 4];
    const int tid = threadIdx.x;
    const uint32_t ipl = 1u << p.ipl_log, span_max = (uint32_t)MERKLE_BLOCK << p.ipl_log;
    const uint64_t base = (uint64_t)blockIdx.x * span_max;
    uint8_t* slot = reinterpret_cast<uint8_t*>(slots32) + tid * SHA_SLOT;
    const bool z0 = p.zero_as_0 != 0;
    const uint32_t span = (uint32_t)(p.n_in - base < span_max ? p.n_in - base : span_max);
    const uint32_t mine = span > ipl * tid ? (span - ipl * tid < ipl ? span - ipl * tid : ipl) : 0u;  // valid inputs of this lane
    int lvl = 0;  // next entry of p.out
    uint64_t v[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < mine; k++) v[k] = gl_canonical_from_mont(p.in[base + ipl * tid + k]);
    if (p.leaf_mode) {
        for (uint32_t k = 0; k < mine; k++) {
            v[k] = gl_hash_elems(v[k], 0, false, slot, z0).v;
            p.out[0][base + ipl * tid + k] = gl_mont_from_canonical(v[k]);
        }
        lvl = 1;
    }
    uint32_t have = mine;  // nodes this lane holds at the current level
    int s = 1;             // level distance from the input level
    for (; s <= p.ipl_log && lvl < p.levels; s++, lvl++) {
        const uint32_t next = (have + 1) / 2;
        for (uint32_t j = 0; j < next; j++) {
            const bool two = 2 * j + 1 < have;
            v[j] = gl_hash_elems(v[2 * j], two ? v[2 * j + 1] : 0, two, slot, z0).v;
            p.out[lvl][(base >> s) + (uint64_t)tid * (ipl >> s) + j] = gl_mont_from_canonical(v[j]);
        }
        have = next;
    }
    if (lvl >= p.levels) return;  // uniform: depends on the launch parameters only
    uint32_t count = (span + ipl - 1) / ipl;  // lane results in this workgroup
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

struct FriTranscriptState {
    uint32_t data[8];  // digest so far, big-endian words
    uint64_t index;    // messages digested
};
constexpr int FRI_TAIL_LOG = 11;
constexpr int FRI_TAIL_MAX = 1 << FRI_TAIL_LOG;
constexpr int FRI_TAIL_THREADS = 1024;
struct FriTailParams {
    const uint64_t* poly;   // coefficients entering the first tail layer (memory form)
    uint32_t len;           // how many (<= size)
    uint32_t log_size;      // first tail layer has 2^log_size points; the tail runs log_size layers (sizes 2^log_size .. 2)
    uint64_t coset;         // canonical coset of the first tail layer
    uint64_t omega;         // canonical root of unity of order 2^log_size
    FriTranscriptState* state;  // transcript digest so far and message counter (device memory, updated in place)
    int zero_as_0;
    uint64_t* evals[FRI_TAIL_LOG];
    uint64_t* nodes[FRI_TAIL_LOG];
    uint64_t* out;          // [0 .. log_size) roots, [log_size] final constant, [log_size + 1] remaining coefficient count
};

ZKP_DEV uint32_t rotl32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }
#define ZKP_CHACHA_QR(a, b, c, d)                                                                                         \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; \
    b = rotl32(b, 7);
// One lane: Transcript::digest(root) (transcript.rs:64-72) followed by generate_a_challenge (86-89); returns the challenge
// as a canonical integer and updates data / index.  `buf` = 128 bytes of LDS scratch.
ZKP_DEV uint64_t fri_transcript_challenge(uint32_t data[8], uint64_t& index, uint64_t root_canonical, uint8_t* buf, bool z0) {
    uint32_t* bw = reinterpret_cast<uint32_t*>(buf);
    for (int i = 0; i < 32; i++) bw[i] = 0;
    for (int i = 0; i < 8; i++) bw[i] = __builtin_bswap32(data[i]);        // previous digest, byte order of the digest
    for (int i = 0; i < 8; i++) buf[32 + i] = (uint8_t)(index >> (8 * i));  // index.to_le_bytes()
    const int len = 40 + gl_write_decimal(root_canonical, buf + 40, z0);
    buf[len] = 0x80;
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    const int blocks = len + 9 <= 64 ? 1 : 2;
    for (int b = 0; b < blocks; b++) {
        uint32_t w[16];
        for (int i = 0; i < 16; i++) w[i] = __builtin_bswap32(bw[16 * b + i]);
        if (b == blocks - 1) w[15] = (uint32_t)len * 8;
        sha256_compress(st, w);
    }
    for (int i = 0; i < 8; i++) data[i] = st[i];
    index++;
    // seed = first 8 digest bytes, little-endian (transcript.rs:80-83)
    uint64_t state = (uint64_t)__builtin_bswap32(st[0]) | (uint64_t)__builtin_bswap32(st[1]) << 32;
    uint32_t key[8];
    for (int i = 0; i < 8; i++) {  // rand_core seed_from_u64: PCG32
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
        key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    for (uint64_t counter = 0;; counter++) {  // ChaCha12 blocks; F::rand = next_u64, rejected while >= p
        uint32_t in[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574, key[0], key[1], key[2], key[3], key[4], key[5], key[6],
                           key[7], (uint32_t)counter, (uint32_t)(counter >> 32), 0, 0};
        uint32_t x[16];
        for (int i = 0; i < 16; i++) x[i] = in[i];
        for (int r = 0; r < 6; r++) {
            ZKP_CHACHA_QR(x[0], x[4], x[8], x[12]) ZKP_CHACHA_QR(x[1], x[5], x[9], x[13])
            ZKP_CHACHA_QR(x[2], x[6], x[10], x[14]) ZKP_CHACHA_QR(x[3], x[7], x[11], x[15])
            ZKP_CHACHA_QR(x[0], x[5], x[10], x[15]) ZKP_CHACHA_QR(x[1], x[6], x[11], x[12])
            ZKP_CHACHA_QR(x[2], x[7], x[8], x[13]) ZKP_CHACHA_QR(x[3], x[4], x[9], x[14])
        }
        for (int i = 0; i < 16; i += 2) {
            const uint64_t v = (uint64_t)(x[i] + in[i]) | (uint64_t)(x[i + 1] + in[i + 1]) << 32;
            if (v < Gl::MOD) return gl_canonical_from_mont(v);  // the sampled limb IS the Montgomery residue
        }
    }
}

// dynamic LDS (no static LDS: the kernel raises its dynamic limit to the full 160 KB): coef[2048] u64 | ev[2048] u64
// (evaluations, then the Merkle levels in place) | tw[1024] u64 | 128-byte transcript buffer | broadcast word | one SHA slot
// per thread
constexpr size_t FRI_TAIL_LDS = 8 * FRI_TAIL_MAX * 2 + 8 * (FRI_TAIL_MAX / 2) + 128 + 16 + (size_t)FRI_TAIL_THREADS * SHA_SLOT;
__global__ __launch_bounds__(FRI_TAIL_THREADS) void fri_tail_kernel(FriTailParams p) {
    extern __shared__ uint4 zkp_smem[];
    uint64_t* coef = reinterpret_cast<uint64_t*>(zkp_smem);  // folded coefficients (memory form)
    uint64_t* ev = coef + FRI_TAIL_MAX;
    uint64_t* tw = ev + FRI_TAIL_MAX;  // omega^k, k < 2^(log_size - 1)
    uint8_t* tbuf = reinterpret_cast<uint8_t*>(tw + FRI_TAIL_MAX / 2);
    uint64_t& bcast = *reinterpret_cast<uint64_t*>(tbuf + 128);
    static_assert(FRI_TAIL_THREADS * 2 == FRI_TAIL_MAX, "one parent / folded coefficient per thread");
    const int tid = threadIdx.x;
    uint8_t* slot = tbuf + 144 + tid * SHA_SLOT;
    const bool z0 = p.zero_as_0 != 0;
    uint32_t len = p.len;
    for (uint32_t i = tid; i < FRI_TAIL_MAX; i += FRI_TAIL_THREADS) coef[i] = i < len ? p.poly[i] : 0;
    uint32_t data[8];
    for (int i = 0; i < 8; i++) data[i] = p.state->data[i];
    uint64_t index = p.state->index;
    Gl coset{p.coset}, omega{p.omega};
    // the twiddles of every stage of every tail layer are strided reads of this table: layer j uses omega^(2^j k) = tw[k << j]
    if (tid < (1 << (p.log_size - 1))) tw[tid] = pow_u64(omega, (uint64_t)tid).v;
    __syncthreads();
    for (uint32_t j = 0; j < p.log_size; j++) {
        const uint32_t ls = p.log_size - j, size = 1u << ls;
        // FriLayer::from_poly (fri_layer.rs:40-46): ev[k] = sum_i c_i (coset w^k)^i  = NTT of c_i coset^i; DIT on a bit-reversed load
        for (uint32_t i = tid; i < size; i += FRI_TAIL_THREADS) {
            const uint32_t src = __brev(i) >> (32 - ls);
            ev[i] = src < len ? (Gl{coef[src]} * pow_u64(coset, src)).v : 0;
        }
        __syncthreads();
        for (uint32_t s = 0; s < ls; s++) {
            const uint32_t half = 1u << s;
            for (uint32_t b = tid; b < size / 2; b += FRI_TAIL_THREADS) {  // distinct pairs: no hazard inside a stage
                const uint32_t pos = b & (half - 1), i0 = ((b >> s) << (s + 1)) | pos;
                const Gl w{tw[((uint64_t)pos << (ls - 1 - s)) << j]};
                const Gl u{ev[i0]}, v = Gl{ev[i0 + half]} * w;
                ev[i0] = (u + v).v;
                ev[i0 + half] = (u - v).v;
            }
            __syncthreads();
        }
        // MerkleTree::new (merkle_tree.rs:42-63); ev[] turns into the current level (canonical hashes), in place
        uint64_t* nodes = p.nodes[j];
        for (uint32_t i = tid; i < size; i += FRI_TAIL_THREADS) {
            const uint64_t e = ev[i];
            p.evals[j][i] = e;
            const uint64_t h = gl_hash_elems(gl_canonical_from_mont(e), 0, false, slot, z0).v;
            ev[i] = h;
            nodes[i] = gl_mont_from_canonical(h);
        }
        __syncthreads();
        uint32_t off = size;
        for (uint32_t count = size; count > 1; count >>= 1) {  // count / 2 <= 1024 parents: one per thread, read - barrier - write
            const uint32_t next = count >> 1;
            uint64_t h = 0;
            if (tid < (int)next) h = gl_hash_elems(ev[2 * tid], ev[2 * tid + 1], true, slot, z0).v;
            __syncthreads();
            if (tid < (int)next) {
                ev[tid] = h;
                nodes[off + tid] = gl_mont_from_canonical(h);
            }
            __syncthreads();
            off += next;
        }
        // transcript: digest the root, draw the folding challenge (prover.rs:58-66)
        if (tid == 0) {
            const uint64_t root = ev[0];
            p.out[j] = gl_mont_from_canonical(root);
            bcast = fri_transcript_challenge(data, index, root, tbuf, z0);
        }
        __syncthreads();
        const Gl r{bcast};
        // fold_polynomial (prover.rs:34-42): at most 1024 outputs, one per thread, read - barrier - write
        const uint32_t nl = (len + 1) / 2;
        uint64_t v = 0;
        if (tid < (int)nl) {
            Gl a{coef[2 * tid]};
            if (2 * (uint32_t)tid + 1 < len) a = a + r * Gl{coef[2 * tid + 1]};
            v = a.v;
        }
        __syncthreads();
        coef[tid] = v;                        // entries >= nl become zero
        coef[tid + FRI_TAIL_THREADS] = 0;
        __syncthreads();
        len = nl;
        coset = coset * coset;
        omega = omega * omega;
    }
    if (tid == 0) {
        p.out[p.log_size] = coef[0];
        p.out[p.log_size + 1] = len;
        for (int i = 0; i < 8; i++) p.state->data[i] = data[i];
        p.state->index = index;
    }
}

// One layer's transcript step for the large layers (one lane): digest the root, draw the folding challenge into *r_out
// (canonical), so that the host never has to wait for a root before it can enqueue the next layer.
__global__ void fri_transcript_kernel(FriTranscriptState* state, const uint64_t* root_mont, uint64_t* r_out, uint64_t* root_out,
                                      int zero_as_0) {
    __shared__ uint32_t buf[32];
    *root_out = *root_mont;  // the proof's copy of the layer root, next to the other small outputs (one D2H for all of them)
    uint32_t data[8];
    for (int i = 0; i < 8; i++) data[i] = state->data[i];
    uint64_t index = state->index;
    *r_out = fri_transcript_challenge(data, index, gl_canonical_from_mont(*root_mont), reinterpret_cast<uint8_t*>(buf), zero_as_0 != 0);
    for (int i = 0; i < 8; i++) state->data[i] = data[i];
    state->index = index;
}
// fold_polynomial with the challenge read from device memory
__global__ void fri_fold_dev_kernel(const uint64_t* c, uint64_t d, const uint64_t* r_canonical, uint64_t* out) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * j >= d) return;
    Gl v{c[2 * j]};
    if (2 * j + 1 < d) v = v + Gl{*r_canonical} * Gl{c[2 * j + 1]};
    out[j] = v.v;
}
// fold_polynomial (prover.rs:34-42) with the challenge read from device memory, fused with the preparation of the NEXT
// layer's transform input: next_poly[j] = c[2j] + r c[2j+1] (j < ceil(d / 2)) and next_ev[j] = next_poly[j] * coset^j,
// zero-padded to the next domain (FriLayer::from_poly evaluates on coset * <omega>: scaling the coefficients by coset^j turns
// it into a plain NTT).  r_canonical == nullptr: no fold, `c` is copied (the first layer).  8 consecutive j per thread: one
// power, then steps.
constexpr int FRI_PREP_CHUNK = 8;
__global__ __launch_bounds__(256) void fri_fold_prep_kernel(const uint64_t* __restrict__ c, uint64_t d,
                                                            const uint64_t* __restrict__ r_canonical, uint64_t coset,
                                                            uint64_t next_dom, uint64_t* __restrict__ next_poly,
                                                            uint64_t* __restrict__ next_ev) {
    const uint64_t j0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * FRI_PREP_CHUNK;
    if (j0 >= next_dom) return;
    const bool fold = r_canonical != nullptr;
    const uint64_t nl = fold ? (d + 1) / 2 : d;
    const Gl r{fold ? *r_canonical : 0}, step{coset};
    Gl pw = pow_u64(step, j0);
    for (uint64_t j = j0; j < j0 + FRI_PREP_CHUNK && j < next_dom; j++) {
        uint64_t e = 0;
        if (j < nl) {
            Gl v;
            if (fold) {
                v = Gl{c[2 * j]};
                if (2 * j + 1 < d) v = v + r * Gl{c[2 * j + 1]};
            } else {
                v = Gl{c[j]};
            }
            if (next_poly) next_poly[j] = v.v;
            e = (v * pw).v;
        }
        next_ev[j] = e;
        pw = pw * step;
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
