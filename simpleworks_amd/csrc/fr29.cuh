// fr29.cuh — BLS12-377 Fr in 9 x 29-bit limbs with LAZY carries, used inside the transform (ntt.hip, ntt_pass_lazy).
//
// Why (same reasoning as fq28.cuh for Fq): on gfx950 the 32-bit-limb Comba multiplication of ff.cuh pays an add-with-carry per
// partial product (120 x (v_mad_u64_u32 + v_addc_co_u32) + the final conditional subtraction: ~330 instructions per Fr
// product, and ~40 per modular addition or subtraction), and the transform kernel is bound by exactly that: its SIMDs issue in
// every cycle (SQ counters, profiles/r03) at 2 360 instructions per element and pass.  With 29-bit limbs a column of 9 + 9 partial
// products fits a 64-bit accumulator without carry handling (153 multiply-adds + 17 column shifts), and additions /
// subtractions are 9 full-rate 32-bit operations with no comparison against the modulus.
//
// Representation.  value = sum l[i] 2^(29 i), 261 bits in 9 limbs; the modulus is 253 bits, so values up to ~446 r fit.
// Montgomery radix 2^261 — but only the TWIDDLES live in that form: fr29_mul(a, t) = a t 2^-261, so with t = w 2^261 the data
// keep whatever factor they came with (2^256, the memory format of the rest of the library) and need no conversion at all:
// unpacking 8 x 32 -> 9 x 29 bits is all that happens on a load.
// Bounds (asserted limb by limb and column by column by tools/check_ntt29.py, which emulates this file):
//   normalised: limbs 0..7 < 2^29, limb 8 takes what is left;  lazy: limbs < 2^32;
//   fr29_mul(a, b): a lazy, b normalised, a b < 2^261 r  =>  result normalised, < 2r;
//   a - b is a + SPREAD - b, SPREAD = limbs of k r with 2^29 (or 2 x 2^29) borrowed into every limb, k >= the bound of b.
#pragma once
#include "ff.cuh"

namespace swm {

struct Fr29 {
    uint32_t l[9];
};
static constexpr uint32_t M29 = (1u << 29) - 1;
struct Fr29Consts {
    static constexpr uint32_t P[9] = SWM_FR29_P;
    static constexpr uint32_t ONE[9] = SWM_FR29_ONE;
    static constexpr uint32_t P2[9] = SWM_FR29_2P;
};

// 8 x 32-bit words (value < 2^256) -> 9 x 29-bit limbs (limb 8 holds bits 232..255)
__device__ __forceinline__ Fr29 fr29_unpack(const Fr& a) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, off = bit & 31;
        uint32_t v = a.v[w] >> off;
        if (off > 3 && w + 1 < 8) v |= a.v[w + 1] << (32 - off);
        r.l[i] = i < 8 ? (v & M29) : v;
    }
    return r;
}
// normalised limbs, value < 2^256 -> 8 words
__device__ __forceinline__ Fr fr29_pack(const Fr29& a) {
    Fr r;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int bit = 32 * w, i = bit / 29, off = bit - 29 * i;
        uint32_t v = a.l[i] >> off;
        if (i + 1 < 9) v |= a.l[i + 1] << (29 - off);
        if (29 - off + 29 < 32 && i + 2 < 9) v |= a.l[i + 2] << (58 - off);
        r.v[w] = v;
    }
    return r;
}
__device__ __forceinline__ Fr29 fr29_const(const uint32_t (&c)[9]) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = c[i];
    return r;
}
// carry propagation: limbs < 2^32 in, limbs 0..7 < 2^29 out (limb 8 takes what is left)
__device__ __forceinline__ Fr29 fr29_normalize(const Fr29& a) {
    Fr29 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t t = a.l[i] + c;
        r.l[i] = t & M29;
        c = t >> 29;
    }
    r.l[8] = a.l[8] + c;
    return r;
}
__device__ __forceinline__ Fr29 fr29_add(const Fr29& a, const Fr29& b) {  // lazy: limbs add, no carry
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// a - b + k r, the spread handed in by the caller (a kernel argument: uniform, it sits in scalar registers)
struct Spread29 {
    uint32_t l[9];
};
__device__ __forceinline__ Fr29 fr29_sub(const Fr29& a, const Fr29& b, const Spread29& sp) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + sp.l[i] - b.l[i];
    return r;
}
// a b 2^-261 mod r (result normalised, < 2r).  a may be lazy, b normalised (a table entry or a product), a b < 2^261 r.
__device__ __forceinline__ Fr29 fr29_mul(const Fr29& a, const Fr29& b) {
    Fr29 r;
    uint32_t m[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Fr29Consts::P[k - i];
        // r = 1 mod 2^29: m_k = -acc mod 2^29 and acc + m_k * r_0 clears the low limb
        m[k] = (0u - (uint32_t)acc) & M29;
        acc = (acc + m[k]) >> 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * Fr29Consts::P[k - i];
        r.l[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
// The same product as one hand-ordered asm statement (r04; text from tools/gen_mul28_asm.py fr29, reasoning in fq28.cuh at
// fq28_mul_asm): the 153 multiply-adds in one chain through v[0:1], low columns closed by add (2^29 - 1) / v_bfi_b32 / shift,
// high columns by mask / shift: 197 instructions where the compiler's schedule of fr29_mul takes ~238.  Same limbs bit for bit
// (tools/check_ntt29.py asserts the carry rule; every NTT test compares with the oracle).  A statement is its own fence.
#include "fr29_mul_asm.inc"
__device__ __forceinline__ Fr29 fr29_mul_fenced(const Fr29& a, const Fr29& b) {
    Fr29 r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;
    constexpr const uint32_t (&P)[9] = Fr29Consts::P;
    static_assert(P[0] == 1u, "the carry rule needs r = 1 mod 2^29");
    asm(SWM_FR29_MUL_ASM_TEXT
        : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]),
          "=&v"(r.l[8]), "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3), "=&v"(m4), "=&v"(m5), "=&v"(m6), "=&v"(m7), "=&v"(m8)
        : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]),
          "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7]), "v"(b.l[8]),
          "s"(P[1]), "s"(P[2]), "s"(P[3]), "s"(P[4]), "s"(P[5]), "s"(P[6]), "s"(P[7]), "s"(P[8]), "s"(M29), "s"((uint64_t)M29)
        : "v0", "v1", "vcc");
    return r;
}
// normalised value < 4r -> canonical (< r): subtract 2r, then r, each kept when it does not borrow
__device__ __forceinline__ Fr29 fr29_cond_sub(const Fr29& a, const uint32_t (&k)[9]) {
    Fr29 t;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint32_t d = a.l[i] - k[i] - borrow;
        borrow = d >> 31;  // limbs < 2^29 (top < 2^24): a negative difference sets bit 31
        t.l[i] = i < 8 ? (d & M29) : d;
    }
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = borrow ? a.l[i] : t.l[i];
    return r;
}
__device__ __forceinline__ Fr29 fr29_canonical(const Fr29& a, bool below_2r) {
    Fr29 x = below_2r ? a : fr29_cond_sub(a, Fr29Consts::P2);
    return fr29_cond_sub(x, Fr29Consts::P);
}

}  // namespace swm
