// ff.cuh — BLS12-377 prime-field arithmetic (Fr: 8 x u32, Fq: 12 x u32, Montgomery form) for gfx950
// device code and for the host-side Marlin logic.
//
// Replaces, for the prove() path of /root/reference/src/marlin/mod.rs:70-77, the arithmetic the reference
// delegates to ark-ff 0.3 Fp256/Fp384 (not vendored).  Memory layout is identical to
// ark-ff's BigInteger256/384 (little-endian 64-bit limbs == little-endian 32-bit limbs on this target), values
// are kept fully reduced in Montgomery form, so buffers can be handed across the C ABI unchanged.
//
// Device multiplication is a product-scanning (Comba/FIPS) Montgomery multiply built from
// v_mad_u64_u32 + v_addc_co_u32 pairs: one 32x32+64 multiply-add and one carry fold per partial
// product, 96-bit column accumulator, no MFMA (carry chains, not a contraction).  Both moduli are
// ≡ 1 mod 2^32, so -p^-1 mod 2^32 = 0xffffffff and the per-column Montgomery factor is a negation.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "constants_gen.h"

// Issue priority of the "light" kernels — everything of a proof that is not the bucket accumulation or the bucket stage: sort
// stages, transforms, mat-vecs, pointwise forms, recurrences.  An accumulation in flight keeps every SIMD's issue port busy with
// three always-ready waves, and the arbiter serves the OLDEST ready wave first: a kernel launched beside it is resident but
// starved (r05 timeline: msm_flat_coarse_hist 1.6 ms beside an accumulation, 30 us alone).  s_setprio raises a wave's priority in
// that arbitration; the light kernels are mostly waiting for memory, so they take few issue cycles from the accumulation and
// stop being the reason the next accumulation starts late.  Measured r05 (builds with -DSWM_LIGHT_PRIO=0 / 1 / 3, alternating on
// one box): prove 2^20 50.4 -> 49.3 ms, "no accumulation in flight" 16.6 -> 12.7 ms of a profiled proof; 2^12 3.55 -> 3.39,
// 2^16 7.9 -> 7.5, 2^18 17.5 -> 17.0 ms, Merkle circuit 15.1 -> 14.8 ms.  0 = off (the r01 - r04 behaviour).
#ifndef SWM_LIGHT_PRIO
#define SWM_LIGHT_PRIO 3
#endif
#define SWM_LIGHT_KERNEL()                                               \
    do {                                                                 \
        if (SWM_LIGHT_PRIO) __builtin_amdgcn_s_setprio(SWM_LIGHT_PRIO);  \
    } while (0)
// the bucket stage and the pre-fold of oversized buckets: VALU-bound chains that share the chip with the NEXT job's accumulation
// (priority 1: above the accumulation, below the light kernels; 0 / 1 / 3 measured within 0.2 ms of one another at 2^20)
#ifndef SWM_TAIL_PRIO
#define SWM_TAIL_PRIO 1
#endif
#define SWM_TAIL_KERNEL()                                              \
    do {                                                               \
        if (SWM_TAIL_PRIO) __builtin_amdgcn_s_setprio(SWM_TAIL_PRIO);  \
    } while (0)

#define SWM_HD __host__ __device__ __forceinline__

namespace swm {

struct FrParams {
    static constexpr int N = 8;
    static constexpr uint32_t P[8] = SWM_FR_MODULUS;
    static constexpr uint32_t R1[8] = SWM_FR_R1;
    static constexpr uint32_t R2[8] = SWM_FR_R2;
    static constexpr uint32_t R3[8] = SWM_FR_R3;
    static constexpr uint32_t PM2[8] = SWM_FR_PM2;
    static constexpr int BITS = 253;
};
struct FqParams {
    static constexpr int N = 12;
    static constexpr uint32_t P[12] = SWM_FQ_MODULUS;
    static constexpr uint32_t R1[12] = SWM_FQ_R1;
    static constexpr uint32_t R2[12] = SWM_FQ_R2;
    static constexpr uint32_t PM2[12] = SWM_FQ_PM2;
    static constexpr int BITS = 377;
};
static_assert(SWM_FR_INV32 == 0xffffffffu && SWM_FQ_INV32 == 0xffffffffu, "moduli must be 1 mod 2^32");

template <class PR>
struct alignas(16) Fp {
    static constexpr int N = PR::N;
    using Params = PR;
    uint32_t v[N];
};
using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

// ------------------------------------------------------------------------------------ basic predicates
template <class F> SWM_HD F fp_zero() {
    F r;
#pragma unroll
    for (int i = 0; i < F::N; i++) r.v[i] = 0;
    return r;
}
template <class F> SWM_HD F fp_one() {
    F r;
#pragma unroll
    for (int i = 0; i < F::N; i++) r.v[i] = F::Params::R1[i];
    return r;
}
template <class F> SWM_HD bool fp_is_zero(const F& a) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) acc |= a.v[i];
    return acc == 0;
}
template <class F> SWM_HD bool fp_eq(const F& a, const F& b) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) acc |= a.v[i] ^ b.v[i];
    return acc == 0;
}
template <class F> SWM_HD bool fp_is_one(const F& a) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) acc |= a.v[i] ^ F::Params::R1[i];
    return acc == 0;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// ---- host side: the same little-endian bytes seen as 64-bit limbs (the verifier's pairing, the prover's host folds)
template <class F> inline void host_limbs64(const F& a, uint64_t* o) {
    for (int i = 0; i < F::N / 2; i++) o[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
}
template <class F> inline void host_store64(F& r, const uint64_t* t) {
    for (int i = 0; i < F::N / 2; i++) {
        r.v[2 * i] = (uint32_t)t[i];
        r.v[2 * i + 1] = (uint32_t)(t[i] >> 32);
    }
}
template <class F> struct HostModulus {  // compile-time constants: the loops below unroll with immediate operands
    static constexpr int M = F::N / 2;
    static constexpr uint64_t limb(int i) { return (uint64_t)F::Params::P[2 * i] | ((uint64_t)F::Params::P[2 * i + 1] << 32); }
    static constexpr uint64_t neg_inv() {  // -p^-1 mod 2^64 by Newton iteration
        uint64_t x = 1;
        for (int i = 0; i < 6; i++) x *= 2 - limb(0) * x;
        return 0 - x;
    }
    static constexpr uint64_t inv = neg_inv();
};
// t (M limbs, + carry bit `extra`) < 2p  ->  t mod p
template <class F> inline void host_cond_sub(uint64_t* t, uint64_t extra) {
    constexpr int M = F::N / 2;
    uint64_t s[M];
    unsigned __int128 borrow = 0;
    for (int i = 0; i < M; i++) {
        unsigned __int128 d = (unsigned __int128)t[i] - HostModulus<F>::limb(i) - (uint64_t)borrow;
        s[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    if (extra || !borrow)
        for (int i = 0; i < M; i++) t[i] = s[i];
}
#endif

// r = a - p if a >= p else a   (a < 2p; `extra` = carry bit above the top limb)
template <class F> SWM_HD void fp_cond_sub_p(F& a, uint32_t extra) {
    uint32_t s[F::N];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
        uint64_t d = (uint64_t)a.v[i] - F::Params::P[i] - borrow;
        s[i] = (uint32_t)d;
        borrow = (uint32_t)(d >> 63);
    }
    bool ge = extra | (borrow == 0);
#pragma unroll
    for (int i = 0; i < F::N; i++) a.v[i] = ge ? s[i] : a.v[i];
}

template <class F> SWM_HD F fp_add(const F& a, const F& b) {
    F r;
#if !defined(__HIP_DEVICE_COMPILE__)
    {
        constexpr int M = F::N / 2;
        uint64_t x[M], y[M];
        host_limbs64(a, x);
        host_limbs64(b, y);
        unsigned __int128 c = 0;
        for (int i = 0; i < M; i++) {
            c += (unsigned __int128)x[i] + y[i];
            x[i] = (uint64_t)c;
            c >>= 64;
        }
        host_cond_sub<F>(x, (uint64_t)c);
        host_store64(r, x);
        return r;
    }
#endif
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
        uint64_t t = (uint64_t)a.v[i] + b.v[i] + carry;
        r.v[i] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
    fp_cond_sub_p(r, carry);  // moduli leave >= 3 spare bits: carry is always 0, kept for generality
    return r;
}
template <class F> SWM_HD F fp_sub(const F& a, const F& b) {
    F r;
#if !defined(__HIP_DEVICE_COMPILE__)
    {
        constexpr int M = F::N / 2;
        uint64_t x[M], y[M];
        host_limbs64(a, x);
        host_limbs64(b, y);
        uint64_t borrow = 0;
        for (int i = 0; i < M; i++) {
            unsigned __int128 d = (unsigned __int128)x[i] - y[i] - borrow;
            x[i] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
        if (borrow) {
            unsigned __int128 c = 0;
            for (int i = 0; i < M; i++) {
                c += (unsigned __int128)x[i] + HostModulus<F>::limb(i);
                x[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        host_store64(r, x);
        return r;
    }
#endif
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
        uint64_t t = (uint64_t)a.v[i] - b.v[i] - borrow;
        r.v[i] = (uint32_t)t;
        borrow = (uint32_t)(t >> 63);
    }
    // add p back when the subtraction went negative
    uint32_t mask = 0u - borrow;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
        uint64_t t = (uint64_t)r.v[i] + (F::Params::P[i] & mask) + carry;
        r.v[i] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
    return r;
}
template <class F> SWM_HD F fp_neg(const F& a) {
    return fp_is_zero(a) ? a : fp_sub(fp_zero<F>(), a);
}
template <class F> SWM_HD F fp_dbl(const F& a) { return fp_add(a, a); }

// ------------------------------------------------------------------------------------ Montgomery multiply
#if defined(__HIP_DEVICE_COMPILE__)
// acc(96 bit) += a * b : one v_mad_u64_u32 (carry-out to an SGPR pair) + one v_addc_co_u32
__device__ __forceinline__ void madc(uint64_t& acc01, uint32_t& acc2, uint32_t a, uint32_t b) {
    uint64_t cy;
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+v"(acc01), "+v"(acc2), "=&s"(cy)
        : "v"(a), "v"(b));
}
// same with the second factor in an SGPR (modulus limbs are wave-uniform constants)
__device__ __forceinline__ void madc_s(uint64_t& acc01, uint32_t& acc2, uint32_t a, uint32_t b_uniform) {
    uint64_t cy;
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2"
        : "+v"(acc01), "+v"(acc2), "=&s"(cy)
        : "v"(a), "s"(b_uniform));
}
#endif

template <class F> SWM_HD F fp_mul(const F& a, const F& b) {
    constexpr int N = F::N;
    F r;
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t m[N];
    uint64_t acc01 = 0;
    uint32_t acc2 = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) madc(acc01, acc2, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) madc_s(acc01, acc2, m[i], F::Params::P[k - i]);
        // m_k = -acc mod 2^32 (inv = -1); acc += m_k * p_0 with p_0 = 1 clears the low word
        uint32_t lo = (uint32_t)acc01;
        m[k] = 0u - lo;
        acc01 = ((acc01 >> 32) | ((uint64_t)acc2 << 32)) + (lo != 0 ? 1u : 0u);
        acc2 = 0;  // the shifted value is < 2^64: acc2 <= 2N fits the upper word
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) madc(acc01, acc2, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = k - N + 1; i < N; i++) madc_s(acc01, acc2, m[i], F::Params::P[k - i]);
        r.v[k - N] = (uint32_t)acc01;
        acc01 = (acc01 >> 32) | ((uint64_t)acc2 << 32);
        acc2 = 0;
    }
    r.v[N - 1] = (uint32_t)acc01;
    fp_cond_sub_p(r, (uint32_t)(acc01 >> 32));
#else
    // host: CIOS on 64-bit limbs (same little-endian bytes as the 32-bit view) with unsigned __int128
    constexpr int M = N / 2;
    constexpr uint64_t inv = HostModulus<F>::inv;
    uint64_t a64[M], b64[M], t[M + 2];
    host_limbs64(a, a64);
    host_limbs64(b, b64);
    for (int i = 0; i < M + 2; i++) t[i] = 0;
    for (int i = 0; i < M; i++) {
        uint64_t c = 0;
        for (int j = 0; j < M; j++) {
            unsigned __int128 x = (unsigned __int128)a64[j] * b64[i] + t[j] + c;
            t[j] = (uint64_t)x;
            c = (uint64_t)(x >> 64);
        }
        unsigned __int128 x = (unsigned __int128)t[M] + c;
        t[M] = (uint64_t)x;
        t[M + 1] = (uint64_t)(x >> 64);
        uint64_t mm = t[0] * inv;
        x = (unsigned __int128)mm * HostModulus<F>::limb(0) + t[0];
        c = (uint64_t)(x >> 64);
        for (int j = 1; j < M; j++) {
            x = (unsigned __int128)mm * HostModulus<F>::limb(j) + t[j] + c;
            t[j - 1] = (uint64_t)x;
            c = (uint64_t)(x >> 64);
        }
        x = (unsigned __int128)t[M] + c;
        t[M - 1] = (uint64_t)x;
        t[M] = t[M + 1] + (uint64_t)(x >> 64);
    }
    host_cond_sub<F>(t, t[M]);
    host_store64(r, t);
#endif
    return r;
}
template <class F> SWM_HD F fp_sqr(const F& a) { return fp_mul(a, a); }

// standard <-> Montgomery
template <class F> SWM_HD F fp_from_std(const F& a) {
    F r2;
#pragma unroll
    for (int i = 0; i < F::N; i++) r2.v[i] = F::Params::R2[i];
    return fp_mul(a, r2);
}
template <class F> SWM_HD F fp_to_std(const F& a) {
    F one = fp_zero<F>();
    one.v[0] = 1;
    return fp_mul(a, one);
}
template <class F> SWM_HD F fp_from_u64(uint64_t x) {
    F r = fp_zero<F>();
    r.v[0] = (uint32_t)x;
    r.v[1] = (uint32_t)(x >> 32);
    return fp_from_std(r);
}

// a^e, e given as little-endian 32-bit limbs (not secret: plain square-and-multiply)
template <class F> SWM_HD F fp_pow(const F& a, const uint32_t* e, int elimbs) {
    F acc = fp_one<F>();
    bool started = false;
    for (int i = elimbs * 32 - 1; i >= 0; i--) {
        if (started) acc = fp_sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            acc = started ? fp_mul(acc, a) : a;
            started = true;
        }
    }
    return acc;
}
// Fermat inverse; 0 -> 0 (matches the "zeros stay zero" convention of ark_ff::batch_inversion callers)
#if !defined(__HIP_DEVICE_COMPILE__)
// host: binary extended Euclid on 64-bit limbs (a few microseconds against ~40 for the Fermat chain): a x1 = u, a x2 = v mod p
// throughout; the Montgomery factor of the input comes out inverted and is put back by two multiplications by R^2.
template <class F> inline F host_inv_euclid(const F& a_mont) {
    constexpr int M = F::N / 2;
    uint64_t u[M], v[M], x1[M], x2[M];
    host_limbs64(a_mont, u);
    for (int i = 0; i < M; i++) {
        v[i] = HostModulus<F>::limb(i);
        x1[i] = x2[i] = 0;
    }
    x1[0] = 1;
    auto is_one = [](const uint64_t* x) {
        uint64_t acc = x[0] ^ 1;
        for (int i = 1; i < M; i++) acc |= x[i];
        return acc == 0;
    };
    auto halve = [&](uint64_t* w, uint64_t* x) {  // w even: w /= 2, x = x / 2 mod p
        for (int i = 0; i < M - 1; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 63);
        w[M - 1] >>= 1;
        uint64_t top = 0;
        if (x[0] & 1) {
            unsigned __int128 c = 0;
            for (int i = 0; i < M; i++) {
                c += (unsigned __int128)x[i] + HostModulus<F>::limb(i);
                x[i] = (uint64_t)c;
                c >>= 64;
            }
            top = (uint64_t)c;
        }
        for (int i = 0; i < M - 1; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 63);
        x[M - 1] = (x[M - 1] >> 1) | (top << 63);
    };
    auto geq = [](const uint64_t* x, const uint64_t* y) {
        for (int i = M - 1; i >= 0; i--)
            if (x[i] != y[i]) return x[i] > y[i];
        return true;
    };
    auto sub = [](uint64_t* x, const uint64_t* y) {  // x -= y, x >= y
        uint64_t borrow = 0;
        for (int i = 0; i < M; i++) {
            unsigned __int128 d = (unsigned __int128)x[i] - y[i] - borrow;
            x[i] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
    };
    auto sub_mod = [&](uint64_t* x, const uint64_t* y) {  // x = x - y mod p, both < p
        uint64_t borrow = 0;
        for (int i = 0; i < M; i++) {
            unsigned __int128 d = (unsigned __int128)x[i] - y[i] - borrow;
            x[i] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
        if (borrow) {
            unsigned __int128 c = 0;
            for (int i = 0; i < M; i++) {
                c += (unsigned __int128)x[i] + HostModulus<F>::limb(i);
                x[i] = (uint64_t)c;
                c >>= 64;
            }
        }
    };
    while (!is_one(u) && !is_one(v)) {
        while (!(u[0] & 1)) halve(u, x1);
        while (!(v[0] & 1)) halve(v, x2);
        if (geq(u, v)) {
            sub(u, v);
            sub_mod(x1, x2);
        } else {
            sub(v, u);
            sub_mod(x2, x1);
        }
    }
    F inv_plain;  // (a R)^-1 as a plain residue = a^-1 R^-1
    host_store64(inv_plain, is_one(u) ? x1 : x2);
    F r2;
    for (int i = 0; i < F::N; i++) r2.v[i] = F::Params::R2[i];
    return fp_mul(fp_mul(inv_plain, r2), r2);
}
#endif
template <class F> SWM_HD F fp_inv(const F& a) {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (fp_is_zero(a)) return a;
    return host_inv_euclid(a);
#endif
    // The exponent limbs are read straight from the constant table, one limb per outer iteration: a private copy
    // indexed by a loop variable lives in scratch memory on the GPU, and its ~250 dependent loads made a single
    // inversion (the serial tail of every batch inversion) twice as slow as its multiplications.
    F acc = a;
    bool started = false;
    for (int limb = F::N - 1; limb >= 0; limb--) {
        const uint32_t w = F::Params::PM2[limb];
        for (int bit = 31; bit >= 0; bit--) {
            if (started) acc = fp_sqr(acc);
            if ((w >> bit) & 1) {
                acc = started ? fp_mul(acc, a) : a;
                started = true;
            }
        }
    }
    return started ? acc : fp_zero<F>();
}

// integer comparison of the standard-form values (ark-ff Ord): -1, 0, 1
template <class F> SWM_HD int fp_cmp_std(const F& a_std, const F& b_std) {
    for (int i = F::N - 1; i >= 0; i--) {
        if (a_std.v[i] > b_std.v[i]) return 1;
        if (a_std.v[i] < b_std.v[i]) return -1;
    }
    return 0;
}

}  // namespace swm
