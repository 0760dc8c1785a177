// frinv.cuh — inverse of ONE Fr element by the binary extended Euclidean algorithm.
#pragma once
#include "ff.cuh"

namespace swm {

// Inverse of ONE element by the binary extended Euclidean algorithm (the serial tail of a batch inversion, the one
// inversion per digest of pedersen.hip: with one lane of a wave active its data-dependent branches cost nothing, and ~250 shift runs + ~250 subtractions on 8 limbs are far
// shorter than the ~380 Montgomery multiplications of a Fermat inversion).  In and out: Montgomery form, a != 0.
__device__ __forceinline__ bool limbs_is_one(const uint32_t (&x)[8]) {
    uint32_t acc = x[0] ^ 1u;
#pragma unroll
    for (int i = 1; i < 8; i++) acc |= x[i];
    return acc == 0;
}
// u >>= k and x = x / 2^k mod p in one step, 1 <= k <= 31, u divisible by 2^k.  p = 1 mod 2^32, so the multiple of p that
// clears the low k bits of x is m = -x mod 2^k:  x <- (x + m p) >> k  (< p again).  One pass over the limbs per RUN of
// trailing zeros instead of one per zero bit (runs average two bits: the inversion is ~1.6x shorter).
__device__ __forceinline__ void limbs_shr_k(uint32_t (&u)[8], uint32_t (&x)[8], unsigned k) {
    const unsigned r = 32 - k;
#pragma unroll
    for (int i = 0; i < 7; i++) u[i] = (u[i] >> k) | (u[i + 1] << r);
    u[7] >>= k;
    const uint32_t m = (0u - x[0]) & ((1u << k) - 1u);
    uint32_t t[9];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)x[i] + (uint64_t)m * FrParams::P[i];
        t[i] = (uint32_t)c;
        c >>= 32;
    }
    t[8] = (uint32_t)c;
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = (t[i] >> k) | (t[i + 1] << r);
}
__device__ __forceinline__ bool limbs_geq(const uint32_t (&a)[8], const uint32_t (&b)[8]) {
    for (int i = 7; i >= 0; i--) {
        if (a[i] != b[i]) return a[i] > b[i];
    }
    return true;
}
__device__ __forceinline__ void limbs_sub(uint32_t (&a)[8], const uint32_t (&b)[8]) {  // a -= b, a >= b
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)a[i] - b[i] - borrow;
        a[i] = (uint32_t)t;
        borrow = (uint32_t)(t >> 63);
    }
}
__device__ __noinline__ Fr fr_inv_single(const Fr a_mont) {
    uint32_t u[8], v[8];
    Fr x1 = fp_zero<Fr>(), x2 = fp_zero<Fr>();  // plain integers mod p, kept < p; fp_sub is the modular subtraction
    x1.v[0] = 1;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u[i] = a_mont.v[i];
        v[i] = FrParams::P[i];
    }
    static_assert(FrParams::P[0] == 1u, "limbs_shr_k needs p = 1 mod 2^32");
    while (!limbs_is_one(u) && !limbs_is_one(v)) {
        while ((u[0] & 1u) == 0) limbs_shr_k(u, x1.v, u[0] ? (unsigned)__ffs((int)u[0]) - 1u : 31u);
        while ((v[0] & 1u) == 0) limbs_shr_k(v, x2.v, v[0] ? (unsigned)__ffs((int)v[0]) - 1u : 31u);
        if (limbs_geq(u, v)) {
            limbs_sub(u, v);
            x1 = fp_sub(x1, x2);
        } else {
            limbs_sub(v, u);
            x2 = fp_sub(x2, x1);
        }
    }
    Fr inv_plain = limbs_is_one(u) ? x1 : x2;  // (a R)^-1 as a plain residue = a^-1 R^-1
    Fr r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.v[i] = FrParams::R2[i];
    return fp_mul(fp_mul(inv_plain, r2), r2);    // x R^-1 * R^2 * R^-1 = x; twice: a^-1 R^-1 -> a^-1 -> a^-1 R
}

}  // namespace swm
