// transcript_host.hpp -- host-side Fiat-Shamir machinery of the reference, written out so that challenges come out
// bit-identical to what the Rust crates produce:
//   * fri/src/fiat_shamir/transcript.rs:30-139   Transcript<Sha256, F>: SHA-256(prev || index_le64 || Display(F))
//   * plonk/src/challenge.rs:36-77                ChallengeGenerator<Sha256>: SHA-256(prev || serialize_uncompressed(G1))
//   * both seed rand 0.8 `StdRng::seed_from_u64(first 8 digest bytes, LE)` and draw `F::rand`
// Third-party behaviour restated here (crate versions from the reference's Cargo.toml files; none of it can be executed in
// this build environment, see DESIGN.md "FRI commitment path"):
//   rand_core 0.6  seed_from_u64: eight PCG32 outputs (multiplier 6364136223846793005, increment 11634580027462260723,
//                  state advanced BEFORE each output, xorshift-rotate output) form the 32-byte key;
//   rand_chacha 0.3 ChaCha with 12 rounds, 64-bit block counter from 0 in words 12-13, stream id 0 in words 14-15;
//                  BlockRng::next_u64 takes two consecutive u32 words, low word first;
//   ark-ff 0.4.2   UniformRand for Fp<MontBackend>: N x next_u64 limbs (least significant first), top limb masked to the
//                  modulus bit length, rejected while >= p; the accepted integer IS the in-memory (Montgomery) value;
//                  Display: canonical integer in decimal, leading zeros trimmed (zero prints as "");
//   ark-bls12-381 0.4.0 serialize_uncompressed(G1Affine): x || y, 48 big-endian bytes each, bit 6 of byte 0 = infinity.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace zkp {

class Sha256 {
public:
    Sha256() { reset(); }
    void reset() {
        static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                                       0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        std::memcpy(h_, iv, sizeof iv);
        fill_ = 0;
        total_ = 0;
    }
    void update(const void* data, size_t n) {
        const uint8_t* p = static_cast<const uint8_t*>(data);
        total_ += n;
        while (n) {
            const size_t take = std::min<size_t>(64 - fill_, n);
            std::memcpy(block_ + fill_, p, take);
            fill_ += take;
            p += take;
            n -= take;
            if (fill_ == 64) {
                compress();
                fill_ = 0;
            }
        }
    }
    std::array<uint8_t, 32> finish() {
        const uint64_t bits = total_ * 8;
        block_[fill_++] = 0x80;
        if (fill_ > 56) {
            std::memset(block_ + fill_, 0, 64 - fill_);
            compress();
            fill_ = 0;
        }
        std::memset(block_ + fill_, 0, 56 - fill_);
        for (int i = 0; i < 8; i++) block_[56 + i] = static_cast<uint8_t>(bits >> (56 - 8 * i));
        compress();
        std::array<uint8_t, 32> out;
        for (int i = 0; i < 8; i++)
            for (int k = 0; k < 4; k++) out[4 * i + k] = static_cast<uint8_t>(h_[i] >> (24 - 8 * k));
        reset();
        return out;
    }

private:
    static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void compress() {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98,
            0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
            0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8,
            0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
            0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
            0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
            0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
            0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; i++)
            w[i] = uint32_t(block_[4 * i]) << 24 | uint32_t(block_[4 * i + 1]) << 16 | uint32_t(block_[4 * i + 2]) << 8 |
                   block_[4 * i + 3];
        for (int i = 16; i < 64; i++)
            w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] +
                   (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
        uint32_t v[8];
        std::memcpy(v, h_, sizeof v);
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = v[7] + (ror(v[4], 6) ^ ror(v[4], 11) ^ ror(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K[i] + w[i];
            const uint32_t t2 = (ror(v[0], 2) ^ ror(v[0], 13) ^ ror(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
            for (int k = 7; k > 0; k--) v[k] = v[k - 1];
            v[4] += t1;
            v[0] = t1 + t2;
        }
        for (int i = 0; i < 8; i++) h_[i] += v[i];
    }
    uint32_t h_[8];
    uint8_t block_[64];
    size_t fill_;
    uint64_t total_;
};

// rand 0.8.5 StdRng (= rand_chacha::ChaCha12Rng) seeded through SeedableRng::seed_from_u64
class StdRng {
public:
    explicit StdRng(uint64_t seed) {
        uint64_t state = seed;
        for (int i = 0; i < 8; i++) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            const uint32_t xorshifted = static_cast<uint32_t>(((state >> 18) ^ state) >> 27);
            const uint32_t rot = static_cast<uint32_t>(state >> 59);
            key_[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
    }
    uint32_t next_u32() {
        if (pos_ == 16) refill();
        return buf_[pos_++];
    }
    uint64_t next_u64() {
        const uint64_t lo = next_u32();
        return lo | static_cast<uint64_t>(next_u32()) << 32;
    }

private:
    static uint32_t rol(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    static void quarter(uint32_t* x, int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 7);
    }
    void refill() {
        uint32_t in[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
        for (int i = 0; i < 8; i++) in[4 + i] = key_[i];
        in[12] = static_cast<uint32_t>(counter_);
        in[13] = static_cast<uint32_t>(counter_ >> 32);
        in[14] = in[15] = 0;
        uint32_t x[16];
        std::memcpy(x, in, sizeof x);
        for (int r = 0; r < 6; r++) {  // 12 rounds = 6 column/diagonal double rounds
            quarter(x, 0, 4, 8, 12); quarter(x, 1, 5, 9, 13); quarter(x, 2, 6, 10, 14); quarter(x, 3, 7, 11, 15);
            quarter(x, 0, 5, 10, 15); quarter(x, 1, 6, 11, 12); quarter(x, 2, 7, 8, 13); quarter(x, 3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) buf_[i] = x[i] + in[i];
        counter_++;
        pos_ = 0;
    }
    uint32_t key_[8];
    uint32_t buf_[16];
    uint64_t counter_ = 0;
    int pos_ = 16;
};

// ark-ff UniformRand for an N-limb prime field: returns the sampled limbs (= the in-memory Montgomery value)
template <int N>
inline std::array<uint64_t, N> sample_field(StdRng& rng, const uint64_t (&modulus)[N], int modulus_bits) {
    const int shave = 64 * N - modulus_bits;
    for (;;) {
        std::array<uint64_t, N> t;
        for (int i = 0; i < N; i++) t[i] = rng.next_u64();
        t[N - 1] &= shave >= 64 ? 0 : ~0ull >> shave;
        bool less = false;
        for (int i = N - 1; i >= 0; i--) {
            if (t[i] != modulus[i]) {
                less = t[i] < modulus[i];
                break;
            }
        }
        if (less) return t;
    }
}
inline uint64_t sample_goldilocks(StdRng& rng) {
    static const uint64_t m[1] = {0xffffffff00000001ull};
    return sample_field<1>(rng, m, 64)[0];
}
inline std::array<uint64_t, 4> sample_bls_fr(StdRng& rng) {
    static const uint64_t m[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    return sample_field<4>(rng, m, 255);
}

// Display of a Goldilocks element given its canonical value
inline std::string goldilocks_display(uint64_t canonical, bool zero_as_0) {
    if (canonical == 0) return zero_as_0 ? "0" : "";
    return std::to_string(canonical);
}

// fri/src/fiat_shamir/transcript.rs
class FriTranscript {
public:
    explicit FriTranscript(bool zero_as_0) : zero_as_0_(zero_as_0) { digest(0); }  // Transcript::new(F::ZERO)
    void digest(uint64_t canonical) {  // transcript.rs:64-72
        Sha256 h;
        if (has_data_) h.update(data_.data(), 32);
        uint8_t le[8];
        for (int i = 0; i < 8; i++) le[i] = static_cast<uint8_t>(index_ >> (8 * i));
        h.update(le, 8);
        const std::string s = goldilocks_display(canonical, zero_as_0_);
        h.update(s.data(), s.size());
        data_ = h.finish();
        has_data_ = true;
        index_++;
        generated_ = false;
    }
    // transcript.rs:74-84; false = "I'm hungry! Feed me something first"
    bool rng(StdRng* out) {
        if (generated_) return false;
        generated_ = true;
        uint64_t seed = 0;
        for (int i = 0; i < 8; i++) seed |= static_cast<uint64_t>(data_[i]) << (8 * i);
        *out = StdRng(seed);
        return true;
    }

    // hand-over to / from the device-side continuation of the same transcript (csrc/fri.hpp: fri_tail_kernel)
    const std::array<uint8_t, 32>& data() const { return data_; }
    uint64_t index() const { return index_; }
    void resume(const std::array<uint8_t, 32>& data, uint64_t index, bool generated) {
        data_ = data;
        index_ = index;
        has_data_ = true;
        generated_ = generated;
    }

private:
    std::array<uint8_t, 32> data_{};
    bool has_data_ = false, generated_ = true, zero_as_0_;
    uint64_t index_ = 0;
};

// plonk/src/challenge.rs
class PlonkChallengeGenerator {
public:
    // canonical affine coordinates as 6 little-endian limbs each (NOT Montgomery)
    void feed(const uint64_t x[6], const uint64_t y[6], bool infinity) {  // challenge.rs:36-45
        uint8_t bytes[96];
        for (int i = 0; i < 48; i++) {
            bytes[i] = infinity ? 0 : static_cast<uint8_t>(x[5 - i / 8] >> (56 - 8 * (i % 8)));
            bytes[48 + i] = infinity ? 0 : static_cast<uint8_t>(y[5 - i / 8] >> (56 - 8 * (i % 8)));
        }
        if (infinity) bytes[0] |= 1 << 6;
        Sha256 h;
        if (has_data_) h.update(data_.data(), 32);
        h.update(bytes, 96);
        data_ = h.finish();
        has_data_ = true;
        generated_ = false;
    }
    // challenge.rs:47-77; false when nothing was fed since the last call (the reference panics)
    bool generate(size_t n, uint64_t* out_mont /* n x 4 */) {
        if (generated_ || !has_data_) return false;
        generated_ = true;
        uint64_t seed = 0;
        for (int i = 0; i < 8; i++) seed |= static_cast<uint64_t>(data_[i]) << (8 * i);
        StdRng rng(seed);
        for (size_t i = 0; i < n; i++) {
            const auto v = sample_bls_fr(rng);
            std::memcpy(out_mont + 4 * i, v.data(), 32);
        }
        return true;
    }

private:
    std::array<uint8_t, 32> data_{};
    bool has_data_ = false, generated_ = false;
};

}  // namespace zkp
