// chacha.h — ChaCha keystream (djb variant: 64-bit block counter, 64-bit stream id = 0) and the two RNGs built
// on it behind /root/reference/src/marlin/mod.rs:
//   generate_rand() = ark_std::test_rng() = rand 0.8 StdRng = ChaCha12 from a fixed seed      (mod.rs:33-35)
//   FS = SimpleHashFiatShamirRng<Blake2s, ChaChaRng>  (ChaChaRng = ChaCha20)                    (mod.rs:13)
// rand_chacha serves words from a 64-word buffer; consecutive next_u32/next_u64 calls consume consecutive
// 32-bit words of the keystream (a u64 is two consecutive words, low word first), so the generator is modelled
// as a flat word stream with a 64-bit word position — which is also what lets the GPU produce bulk draws
// (the 3|H| mask-polynomial coefficients) from (key, position) without a host round trip.
#pragma once
#include <stdint.h>
#include <string.h>
#include "../ff.cuh"

namespace swm {

SWM_HD uint32_t chacha_rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

// one 16-word block
SWM_HD void chacha_block(const uint32_t key[8], uint64_t counter, int rounds, uint32_t out[16]) {
    uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                       key[4], key[5], key[6], key[7], (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t s[16];
    for (int i = 0; i < 16; i++) s[i] = in[i];
#define SWM_QR(a, b, c, d)                    \
    s[a] += s[b]; s[d] = chacha_rotl(s[d] ^ s[a], 16); \
    s[c] += s[d]; s[b] = chacha_rotl(s[b] ^ s[c], 12); \
    s[a] += s[b]; s[d] = chacha_rotl(s[d] ^ s[a], 8);  \
    s[c] += s[d]; s[b] = chacha_rotl(s[b] ^ s[c], 7);
    for (int r = 0; r < rounds; r += 2) {
        SWM_QR(0, 4, 8, 12) SWM_QR(1, 5, 9, 13) SWM_QR(2, 6, 10, 14) SWM_QR(3, 7, 11, 15)
        SWM_QR(0, 5, 10, 15) SWM_QR(1, 6, 11, 12) SWM_QR(2, 7, 8, 13) SWM_QR(3, 4, 9, 14)
    }
#undef SWM_QR
    for (int i = 0; i < 16; i++) out[i] = s[i] + in[i];
}

// Eight consecutive blocks at once with AVX2 (lane l of every vector belongs to block counter + l), for bulk keystream on
// the host (swm_rng_fill_bytes: the test-harness stand-in for a caller's StdRng; rand_chacha vectorises the same way).
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#define SWM_CHACHA_WIDE 1
}  // namespace swm
#include <immintrin.h>
namespace swm {
__attribute__((target("avx2"))) inline void chacha_blocks8_avx2(const uint32_t key[8], uint64_t counter, int rounds, uint32_t* out) {
    __m256i in[16], s[16];
    in[0] = _mm256_set1_epi32((int)0x61707865u);
    in[1] = _mm256_set1_epi32((int)0x3320646eu);
    in[2] = _mm256_set1_epi32((int)0x79622d32u);
    in[3] = _mm256_set1_epi32((int)0x6b206574u);
    for (int i = 0; i < 8; i++) in[4 + i] = _mm256_set1_epi32((int)key[i]);
    alignas(32) uint32_t lo[8], hi[8];
    for (int l = 0; l < 8; l++) {
        const uint64_t c = counter + (uint64_t)l;
        lo[l] = (uint32_t)c;
        hi[l] = (uint32_t)(c >> 32);
    }
    in[12] = _mm256_load_si256((const __m256i*)lo);
    in[13] = _mm256_load_si256((const __m256i*)hi);
    in[14] = _mm256_setzero_si256();
    in[15] = _mm256_setzero_si256();
    for (int i = 0; i < 16; i++) s[i] = in[i];
#define SWM_ROTL8(v, c) _mm256_or_si256(_mm256_slli_epi32(v, c), _mm256_srli_epi32(v, 32 - c))
#define SWM_QR8(a, b, c, d)                                                    \
    s[a] = _mm256_add_epi32(s[a], s[b]); s[d] = _mm256_xor_si256(s[d], s[a]); s[d] = SWM_ROTL8(s[d], 16); \
    s[c] = _mm256_add_epi32(s[c], s[d]); s[b] = _mm256_xor_si256(s[b], s[c]); s[b] = SWM_ROTL8(s[b], 12); \
    s[a] = _mm256_add_epi32(s[a], s[b]); s[d] = _mm256_xor_si256(s[d], s[a]); s[d] = SWM_ROTL8(s[d], 8);  \
    s[c] = _mm256_add_epi32(s[c], s[d]); s[b] = _mm256_xor_si256(s[b], s[c]); s[b] = SWM_ROTL8(s[b], 7);
    for (int r = 0; r < rounds; r += 2) {
        SWM_QR8(0, 4, 8, 12) SWM_QR8(1, 5, 9, 13) SWM_QR8(2, 6, 10, 14) SWM_QR8(3, 7, 11, 15)
        SWM_QR8(0, 5, 10, 15) SWM_QR8(1, 6, 11, 12) SWM_QR8(2, 7, 8, 13) SWM_QR8(3, 4, 9, 14)
    }
#undef SWM_QR8
#undef SWM_ROTL8
    alignas(32) uint32_t t[16][8];
    for (int i = 0; i < 16; i++) _mm256_store_si256((__m256i*)t[i], _mm256_add_epi32(s[i], in[i]));
    for (int l = 0; l < 8; l++)
        for (int i = 0; i < 16; i++) out[16 * l + i] = t[i][l];
}
inline bool chacha_have_avx2() {
    static const bool have = __builtin_cpu_supports("avx2");
    return have;
}
#endif

// Caller-owned randomness (swm_rng_from_callback): the reference passes `&mut StdRng` into setup / prove / verify
// (src/marlin/mod.rs:49,73,83), so a drop-in has to draw from THAT generator.  rand_core's BlockRng serves
// next_u32 / next_u64 / fill_bytes from one flat stream of 32-bit words (a u64 is two consecutive words, low first;
// fill_bytes(4 k) is k consecutive words), so every draw below is expressed as fill(4) / fill(8) / fill(32 n) and the
// caller's generator advances exactly as it would under arkworks.
typedef void (*RngFillFn)(void* user, uint8_t* dest, size_t len);

struct ChaChaRng {
    uint32_t key[8];
    int rounds;
    uint64_t pos;        // next keystream word
    uint64_t cached_blk; // block currently in `block`
    uint32_t block[16];
    bool have;
    RngFillFn ext = nullptr;  // non-null: every draw is forwarded to the caller's generator
    void* ext_user = nullptr;

    void seed(const uint8_t s[32], int nrounds) {
        for (int i = 0; i < 8; i++)
            key[i] = (uint32_t)s[4 * i] | ((uint32_t)s[4 * i + 1] << 8) | ((uint32_t)s[4 * i + 2] << 16) |
                     ((uint32_t)s[4 * i + 3] << 24);
        rounds = nrounds;
        pos = 0;
        have = false;
        cached_blk = 0;
    }
    uint32_t next_u32() {
        if (ext) {
            uint8_t b[4];
            ext(ext_user, b, 4);
            return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
        }
        uint64_t blk = pos >> 4;
        if (!have || blk != cached_blk) {
            chacha_block(key, blk, rounds, block);
            cached_blk = blk;
            have = true;
        }
        return block[pos++ & 15];
    }
    uint64_t next_u64() {
        if (ext) {
            uint8_t b[8];
            ext(ext_user, b, 8);
            uint64_t v = 0;
            for (int i = 7; i >= 0; i--) v = (v << 8) | b[i];
            return v;
        }
        uint64_t lo = next_u32();
        uint64_t hi = next_u32();
        return (hi << 32) | lo;
    }
    bool gen_bool() { return (next_u32() >> 31) != 0; }  // rand Standard for bool: sign bit of a u32
    void gen_u128(uint64_t out[2]) {                      // rand Standard for u128: low u64 first
        out[0] = next_u64();
        out[1] = next_u64();
    }
    // ark_ff UniformRand for Fp: draw N/2 u64 limbs, clear the top `shave` bits, accept if < p;
    // the accepted limbs ARE the Montgomery representation.
    template <class F>
    F rand_field(int shave) {
        for (;;) {
            F r;
            for (int i = 0; i < F::N / 2; i++) {
                uint64_t v = next_u64();
                r.v[2 * i] = (uint32_t)v;
                r.v[2 * i + 1] = (uint32_t)(v >> 32);
            }
            r.v[F::N - 1] &= 0xffffffffu >> shave;
            bool lt = false;
            for (int i = F::N - 1; i >= 0; i--) {
                if (r.v[i] < F::Params::P[i]) { lt = true; break; }
                if (r.v[i] > F::Params::P[i]) break;
            }
            if (lt) return r;
        }
    }
    Fr rand_fr() { return rand_field<Fr>(3); }
    Fq rand_fq() { return rand_field<Fq>(7); }
};

inline ChaChaRng test_rng() {
    static const uint8_t seed[32] = {1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0,
                                     0, 0, 0, 0, 0,  0, 0, 0, 0,   0, 0, 0, 0,   0,  0, 0};
    ChaChaRng r;
    r.seed(seed, 12);
    return r;
}

}  // namespace swm
