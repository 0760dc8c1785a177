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
__device__ __noinline__ Fr fr_inv_single_exact(const Fr a_mont) {
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

// ---------------------------------------------------------------------------------------------- r04: the same inverse, six times shorter
// Binary GCD on 64-bit APPROXIMATIONS (Pornin, "Optimized Binary GCD for Modular Inversion", 2020).  The loop above works on
// the full 8-limb values in every one of its ~500 steps (~230 us for the one lane that runs it — the serial tail of every batch
// inversion and 0.45 ms of a small proof).  Here a round takes the low 31 bits and the top 33 bits of (a, b) into two 64-bit
// words, runs 31 steps of the binary GCD on those alone while recording what they do as a 2 x 2 matrix (f0 g0; f1 g1) of signed
// 32-bit entries, and only then applies the matrix to the full values:
//     (a, b) <- ((f0 a + g0 b) / 2^31, (f1 a + g1 b) / 2^31)          exact divisions; a negative result flips its row
//     (u, v) <- ((f0 u + g0 v) / 2^31, (f1 u + g1 v) / 2^31) mod p     Montgomery-style: p = 1 mod 2^32, so t = -S mod 2^31
// with a = u y, b = v y (mod p) throughout.  ceil((2 * 253 - 1) / 31) = 17 rounds end in (a, b) = (0, 1), v = 1 / y.
// The low bits decide every parity exactly; the top bits can mis-order a and b only when they are close, which costs nothing
// in the bound (the paper's argument) — and if a round count ever did not suffice the caller falls back to the loop above,
// which is exact (fr_inv_single checks b == 1).  Host and device (tests/test_host_logic.py runs it on the CPU against Python).
struct BgMat {
    int64_t f0, g0, f1, g1;
};
SWM_HD unsigned bg_bitlen(const uint32_t (&x)[8]) {
    for (int i = 7; i >= 0; i--)
        if (x[i]) {
            unsigned l = 0;
            uint32_t t = x[i];
            while (t) {
                l++;
                t >>= 1;
            }
            return 32u * (unsigned)i + l;
        }
    return 0;
}
// low 31 bits of x below the 33 bits of x that start at bit n - 33 (n >= 64)
SWM_HD uint64_t bg_approx(const uint32_t (&x)[8], unsigned n) {
    const unsigned sh = n - 33, w = sh >> 5, o = sh & 31;
    uint64_t lo = w < 8 ? x[w] : 0, mid = w + 1 < 8 ? x[w + 1] : 0, hi = w + 2 < 8 ? x[w + 2] : 0;
    uint64_t top = (lo >> o) | (mid << (32 - o));           // 64 - o valid bits when o == 0: mid << 32
    if (o) top |= hi << (64 - o);
    top &= (1ull << 33) - 1;
    return (uint64_t)(x[0] & 0x7fffffffu) | (top << 31);
}
// S = f X + g Y as a two's complement number of ten 32-bit limbs (|f| + |g| <= 2^31, X, Y < 2^256)
SWM_HD void bg_lincomb(int64_t f, int64_t g, const uint32_t (&X)[8], const uint32_t (&Y)[8], uint32_t (&S)[10]) {
    const uint32_t af = (uint32_t)(f < 0 ? -f : f), ag = (uint32_t)(g < 0 ? -g : g);
    uint32_t P[10], Q[10];
    uint64_t c = 0, d = 0;
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)af * X[i];
        P[i] = (uint32_t)c;
        c >>= 32;
        d += (uint64_t)ag * Y[i];
        Q[i] = (uint32_t)d;
        d >>= 32;
    }
    P[8] = (uint32_t)c;
    P[9] = 0;
    Q[8] = (uint32_t)d;
    Q[9] = 0;
    // S = (+-P) + (+-Q): negation as complement + 1, folded into the carry chain
    const uint32_t mp = f < 0 ? 0xffffffffu : 0u, mq = g < 0 ? 0xffffffffu : 0u;
    uint64_t carry = (uint64_t)(mp & 1u) + (mq & 1u);
    for (int i = 0; i < 10; i++) {
        carry += (uint64_t)(P[i] ^ mp) + (uint64_t)(Q[i] ^ mq);
        S[i] = (uint32_t)carry;
        carry >>= 32;
    }
}
// S >>= 31 (arithmetic), keeping eight limbs and returning the sign (S / 2^31 fits 256 bits + sign)
SWM_HD bool bg_shift31(const uint32_t (&S)[10], uint32_t (&out)[8]) {
    for (int i = 0; i < 8; i++) out[i] = (S[i] >> 31) | (S[i + 1] << 1);
    return (S[9] >> 31) != 0;
}
SWM_HD void bg_negate(uint32_t (&x)[8]) {
    uint64_t c = 1;
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)(~x[i]);
        x[i] = (uint32_t)c;
        c >>= 32;
    }
}
// (f U + g V) / 2^31 mod p, U, V < p.  p = 1 mod 2^32: the multiple of p that clears the low 31 bits of S is t = -S mod 2^31.
SWM_HD void bg_lincomb_mod(int64_t f, int64_t g, const uint32_t (&U)[8], const uint32_t (&V)[8], uint32_t (&out)[8]) {
    uint32_t S[10];
    bg_lincomb(f, g, U, V, S);
    const uint32_t t = (0u - S[0]) & 0x7fffffffu;
    uint64_t c = 0;
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)S[i] + (uint64_t)t * FrParams::P[i];
        S[i] = (uint32_t)c;
        c >>= 32;
    }
    for (int i = 8; i < 10; i++) {
        c += S[i];
        S[i] = (uint32_t)c;
        c >>= 32;
    }
    const bool neg = bg_shift31(S, out);  // in (-p, 2p)
    if (neg) {
        uint64_t a = 0;
        for (int i = 0; i < 8; i++) {
            a += (uint64_t)out[i] + FrParams::P[i];
            out[i] = (uint32_t)a;
            a >>= 32;
        }
    } else {
        bool ge = true;
        for (int i = 7; i >= 0; i--)
            if (out[i] != FrParams::P[i]) {
                ge = out[i] > FrParams::P[i];
                break;
            }
        if (ge) {
            uint32_t borrow = 0;
            for (int i = 0; i < 8; i++) {
                uint64_t d = (uint64_t)out[i] - FrParams::P[i] - borrow;
                out[i] = (uint32_t)d;
                borrow = (uint32_t)(d >> 63);
            }
        }
    }
}
// a^-1 for a != 0, Montgomery form in and out; *ok = false when the rounds did not end in (0, 1) (never observed; the caller
// then uses the exact loop)
SWM_HD Fr fr_inv_bingcd(const Fr& a_mont, bool* ok) {
    uint32_t a[8], b[8], u[8], v[8];
    for (int i = 0; i < 8; i++) {
        a[i] = a_mont.v[i];
        b[i] = FrParams::P[i];
        u[i] = 0;
        v[i] = 0;
    }
    u[0] = 1;
    for (int round = 0; round < 17; round++) {
        unsigned n = bg_bitlen(a), nb = bg_bitlen(b);
        if (nb > n) n = nb;
        if (n < 64) n = 64;
        uint64_t xa = bg_approx(a, n), xb = bg_approx(b, n);
        BgMat m{1, 0, 0, 1};
        for (int j = 0; j < 31; j++) {
            if (xa & 1) {
                if (xa < xb) {
                    const uint64_t t = xa;
                    xa = xb;
                    xb = t;
                    int64_t s = m.f0;
                    m.f0 = m.f1;
                    m.f1 = s;
                    s = m.g0;
                    m.g0 = m.g1;
                    m.g1 = s;
                }
                xa -= xb;
                m.f0 -= m.f1;
                m.g0 -= m.g1;
            }
            xa >>= 1;
            m.f1 <<= 1;
            m.g1 <<= 1;
        }
        uint32_t S[10], na[8], nbv[8];
        bg_lincomb(m.f0, m.g0, a, b, S);
        if (bg_shift31(S, na)) {
            bg_negate(na);
            m.f0 = -m.f0;
            m.g0 = -m.g0;
        }
        bg_lincomb(m.f1, m.g1, a, b, S);
        if (bg_shift31(S, nbv)) {
            bg_negate(nbv);
            m.f1 = -m.f1;
            m.g1 = -m.g1;
        }
        uint32_t nu[8], nv[8];
        bg_lincomb_mod(m.f0, m.g0, u, v, nu);
        bg_lincomb_mod(m.f1, m.g1, u, v, nv);
        for (int i = 0; i < 8; i++) {
            a[i] = na[i];
            b[i] = nbv[i];
            u[i] = nu[i];
            v[i] = nv[i];
        }
    }
    uint32_t za = 0, zb = b[0] ^ 1u;
    for (int i = 0; i < 8; i++) {
        za |= a[i];
        if (i) zb |= b[i];
    }
    *ok = za == 0 && zb == 0;
    Fr inv_plain, r3;  // v = (a R)^-1 = a^-1 R^-1 as a plain residue; times R^3 under one Montgomery product: a^-1 R
    for (int i = 0; i < 8; i++) {
        inv_plain.v[i] = v[i];
        r3.v[i] = FrParams::R3[i];
    }
    return fp_mul(inv_plain, r3);
}
// the inverse of one element on a single lane: 17 rounds on approximations, the exact loop if they ever did not suffice
__device__ __noinline__ Fr fr_inv_single(const Fr a_mont) {
    bool ok;
    const Fr r = fr_inv_bingcd(a_mont, &ok);
    return ok ? r : fr_inv_single_exact(a_mont);
}

}  // namespace swm
