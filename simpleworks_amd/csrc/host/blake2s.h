// blake2s.h — BLAKE2s-256 (RFC 7693), unkeyed.  Host side of the Fiat-Shamir transcript:
// FS = SimpleHashFiatShamirRng<Blake2s, ChaChaRng> (/root/reference/src/marlin/mod.rs:13; blake2 0.9, Cargo.toml:34).
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

namespace swm {

class Blake2s {
   public:
    Blake2s() {
        static const uint32_t iv[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A,
                                       0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
        for (int i = 0; i < 8; i++) h_[i] = iv[i];
        h_[0] ^= 0x01010000 ^ 32;  // digest length 32, no key, fanout = depth = 1
        t_ = 0;
        buflen_ = 0;
    }
    void update(const uint8_t* in, size_t len) {
        while (len > 0) {
            if (buflen_ == 64) {  // buffer full and more input follows: not the last block
                t_ += 64;
                compress(false);
                buflen_ = 0;
            }
            size_t take = 64 - buflen_;
            if (take > len) take = len;
            memcpy(buf_ + buflen_, in, take);
            buflen_ += take;
            in += take;
            len -= take;
        }
    }
    void finalize(uint8_t out[32]) {
        t_ += buflen_;
        memset(buf_ + buflen_, 0, 64 - buflen_);
        compress(true);
        for (int i = 0; i < 8; i++) {
            out[4 * i] = (uint8_t)h_[i];
            out[4 * i + 1] = (uint8_t)(h_[i] >> 8);
            out[4 * i + 2] = (uint8_t)(h_[i] >> 16);
            out[4 * i + 3] = (uint8_t)(h_[i] >> 24);
        }
    }
    static void digest(const uint8_t* in, size_t len, uint8_t out[32]) {
        Blake2s b;
        b.update(in, len);
        b.finalize(out);
    }

   private:
    static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void compress(bool last) {
        static const uint32_t iv[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A,
                                       0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
        static const uint8_t sigma[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        uint32_t m[16], v[16];
        for (int i = 0; i < 16; i++)
            m[i] = (uint32_t)buf_[4 * i] | ((uint32_t)buf_[4 * i + 1] << 8) | ((uint32_t)buf_[4 * i + 2] << 16) |
                   ((uint32_t)buf_[4 * i + 3] << 24);
        for (int i = 0; i < 8; i++) {
            v[i] = h_[i];
            v[i + 8] = iv[i];
        }
        v[12] ^= (uint32_t)t_;
        v[13] ^= (uint32_t)(t_ >> 32);
        if (last) v[14] = ~v[14];
#define SWM_B2S_G(a, b, c, d, x, y)         \
    v[a] = v[a] + v[b] + x;                 \
    v[d] = rotr(v[d] ^ v[a], 16);           \
    v[c] = v[c] + v[d];                     \
    v[b] = rotr(v[b] ^ v[c], 12);           \
    v[a] = v[a] + v[b] + y;                 \
    v[d] = rotr(v[d] ^ v[a], 8);            \
    v[c] = v[c] + v[d];                     \
    v[b] = rotr(v[b] ^ v[c], 7);
        for (int r = 0; r < 10; r++) {
            const uint8_t* s = sigma[r];
            SWM_B2S_G(0, 4, 8, 12, m[s[0]], m[s[1]])
            SWM_B2S_G(1, 5, 9, 13, m[s[2]], m[s[3]])
            SWM_B2S_G(2, 6, 10, 14, m[s[4]], m[s[5]])
            SWM_B2S_G(3, 7, 11, 15, m[s[6]], m[s[7]])
            SWM_B2S_G(0, 5, 10, 15, m[s[8]], m[s[9]])
            SWM_B2S_G(1, 6, 11, 12, m[s[10]], m[s[11]])
            SWM_B2S_G(2, 7, 8, 13, m[s[12]], m[s[13]])
            SWM_B2S_G(3, 4, 9, 14, m[s[14]], m[s[15]])
        }
#undef SWM_B2S_G
        for (int i = 0; i < 8; i++) h_[i] ^= v[i] ^ v[i + 8];
    }
    uint32_t h_[8];
    uint64_t t_;
    uint8_t buf_[64];
    size_t buflen_;
};

}  // namespace swm
