// fq28.cuh — BLS12-377 Fq in 14 x 28-bit limbs with LAZY carries, used inside the MSM bucket accumulation.
//
// Why: on gfx950 every carry-propagating integer op (v_addc_co_u32, v_lshl_add_u64) issues at the same half rate as
// v_mad_u64_u32 (measured 4.2 cycles per wave64 per SIMD, tools/ubench/valu_rates.hip), so the 32-bit-limb Comba
// multiply in ff.cuh pays one carry fold per partial product: 276 x (mad + addc) ~ 2640 cycles per wave.
// With 28-bit limbs a column of up to 28 partial products fits a 64-bit accumulator without any carry handling:
// 378 v_mad_u64_u32 + 27 column shifts ~ 1.9k cycles, and additions / subtractions become 14 full-rate 32-bit adds
// with no comparison against the modulus.
//
// Representation.  value = sum l[i] 2^(28 i); Montgomery radix R' = 2^392 (NOT the 2^384 of the memory format: inputs
// are pre-scaled by 2^8 once, outputs are scaled back with one multiplication per coordinate).
// Bounds (checked case by case at the call sites, see g1_28 in msm.hip):
//   N  "normalised": limbs < 2^28 (top limb < 2^15), value < 2p         — what mul28 returns
//   D  "lazy":       limbs < 2^30,                    value < 128 p      — what mul28 accepts for BOTH operands:
//        column sum <= 14 * 2^60 + 14 * 2^56 + 2^37 < 2^64;  a b / 2^392 < 2^14 p^2 / 2^392 < p / 2  =>  result < 1.5 p
// Subtraction a - b is a + SPREADk - b, where SPREADk are limbs of k*p with 2^28 borrowed into every limb so that no
// limb goes negative; the value grows by k*p, which the next multiplication absorbs.
#pragma once
#include "ff.cuh"
#include "g1.cuh"

namespace swm {

struct Fq28 {
    uint32_t l[14];
};
static constexpr uint32_t M28 = (1u << 28) - 1;

struct Fq28Consts {
    static constexpr uint32_t P[14] = SWM_FQ28_P;
    static constexpr uint32_t ONE[14] = SWM_FQ28_ONE;
    static constexpr uint32_t TO384[14] = SWM_FQ28_TO384;
    static constexpr uint32_t SPREAD2[14] = SWM_FQ28_SPREAD2_1;
    static constexpr uint32_t SPREAD4[14] = SWM_FQ28_SPREAD4_1;
    static constexpr uint32_t SPREAD8[14] = SWM_FQ28_SPREAD8_1;
    static constexpr uint32_t SPREAD16_3[14] = SWM_FQ28_SPREAD16_3;
    static constexpr uint32_t SPREAD32[14] = SWM_FQ28_SPREAD32_1;
};

// 12 x 32-bit words (value < 2^384) -> 14 x 28-bit limbs (top limb holds bits 364..383)
__device__ __forceinline__ Fq28 fq28_unpack(const Fq& a) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const int bit = 28 * i, w = bit >> 5, off = bit & 31;
        uint32_t v = a.v[w] >> off;
        if (off > 4 && w + 1 < 12) v |= a.v[w + 1] << (32 - off);
        r.l[i] = i < 13 ? (v & M28) : v;
    }
    return r;
}
// normalised limbs (< 2^28, value < 2^384) -> 12 words
__device__ __forceinline__ Fq fq28_pack(const Fq28& a) {
    Fq r;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 32 * w, i = bit / 28, off = bit - 28 * i;
        uint32_t v = a.l[i] >> off;
        if (i + 1 < 14) v |= a.l[i + 1] << (28 - off);
        if (28 - off + 28 < 32 && i + 2 < 14) v |= a.l[i + 2] << (56 - off);
        r.v[w] = v;
    }
    return r;
}

// carry propagation: limbs < 2^32 in, limbs < 2^28 out (top limb takes what is left)
__device__ __forceinline__ Fq28 fq28_normalize(const Fq28& a) {
    Fq28 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        uint32_t t = a.l[i] + c;
        r.l[i] = t & M28;
        c = t >> 28;
    }
    r.l[13] = a.l[13] + c;
    return r;
}
__device__ __forceinline__ Fq28 fq28_add(const Fq28& a, const Fq28& b) {  // lazy: limbs add, no carry
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// a - b + k p  (b limbs must not exceed the spread's borrow: < 2^28 for the _1 spreads, < 3 * 2^28 for SPREAD16_3)
#define FQ28_SUB(a, b, SPREAD) fq28_sub_impl((a), (b), Fq28Consts::SPREAD)
__device__ __forceinline__ Fq28 fq28_sub_impl(const Fq28& a, const Fq28& b, const uint32_t (&sp)[14]) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + sp[i] - b.l[i];
    return r;
}

// Montgomery product a b 2^-392 mod p (result N).  Operands may be lazy (D).
__device__ __forceinline__ Fq28 fq28_mul(const Fq28& a, const Fq28& b) {
    Fq28 r;
    uint32_t m[14];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Fq28Consts::P[k - i];
        // p = 1 mod 2^28: m_k = -acc mod 2^28 and acc + m_k * p_0 clears the low limb
        m[k] = (0u - (uint32_t)acc) & M28;
        acc = (acc + m[k]) >> 28;
    }
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (uint64_t)m[i] * Fq28Consts::P[k - i];
        r.l[k - 14] = (uint32_t)acc & M28;
        acc >>= 28;
    }
    r.l[13] = (uint32_t)acc;
    return r;
}
// The same product as ONE hand-ordered asm statement (r04; text generated by tools/gen_mul28_asm.py).  Written in plain C++
// the compiler re-associates every column sum: it starts each column from zero (to shorten a dependency chain that costs
// nothing on this in-order SIMD with three resident waves — tools/ubench/valu_rates: a dependent v_mad_u64_u32 chain issues
// at the rate of independent ones), adds the carry of the previous column with a 64-bit addition of its own, widens m_k to
// 64 bits with a move and adds it with another one: 515 instructions for the 378 multiply-adds of a product.  Here:
//   * the 64-bit accumulator is the column sum itself: the carry of the previous column is what it holds when the column starts;
//   * low columns: acc + m_k with m_k = -acc mod 2^28 is the next multiple of 2^28, so carry = (acc + (2^28 - 1)) >> 28 and
//     m_k = ~(acc + (2^28 - 1)) mod 2^28: one 64-bit add, one v_bfi_b32, one 64-bit shift (no widening move, no negation);
//     acc + 2^28 - 1 < 2^64 by the column bound above (margin 2^60);
//   * high columns: a mask and a shift.
// 378 + 14 x 3 + 13 x 2 + 1 = 447 instructions, the same limbs as fq28_mul bit for bit (tools/ubench/te28_bench.hip compares
// them on the GPU; tools/check_te28.py emulates the carry rule).  One statement, not one per instruction: after an asm
// statement that defines a VGPR the compiler inserts an s_nop before the next reader (it must assume a dst_sel hazard).
#include "fq28_mul_asm.inc"
__device__ __forceinline__ Fq28 fq28_mul_asm(const Fq28& a, const Fq28& b) {
    Fq28 r;
    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13;
    constexpr const uint32_t (&P)[14] = Fq28Consts::P;
    asm(SWM_FQ28_MUL_ASM_TEXT
        : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]),
          "=&v"(r.l[8]), "=&v"(r.l[9]), "=&v"(r.l[10]), "=&v"(r.l[11]), "=&v"(r.l[12]), "=&v"(r.l[13]),
          "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3), "=&v"(m4), "=&v"(m5), "=&v"(m6), "=&v"(m7), "=&v"(m8), "=&v"(m9),
          "=&v"(m10), "=&v"(m11), "=&v"(m12), "=&v"(m13)
        : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]),
          "v"(a.l[9]), "v"(a.l[10]), "v"(a.l[11]), "v"(a.l[12]), "v"(a.l[13]),
          "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7]), "v"(b.l[8]),
          "v"(b.l[9]), "v"(b.l[10]), "v"(b.l[11]), "v"(b.l[12]), "v"(b.l[13]),
          "s"(P[1]), "s"(P[2]), "s"(P[3]), "s"(P[4]), "s"(P[5]), "s"(P[6]), "s"(P[7]), "s"(P[8]), "s"(P[9]), "s"(P[10]),
          "s"(P[11]), "s"(P[12]), "s"(P[13]), "s"(M28), "s"((uint64_t)M28)
        : "v0", "v1", "vcc");
    return r;
}
// (a b + c d) 2^-392 mod p with ONE Montgomery reduction: 2 x 196 + 182 multiply-adds instead of 2 x 378.  The
// group formulas end in Y3 = R V - Y1 PPP; with c = k p - Y1 that is exactly this shape.
// Bounds: limbs(a), limbs(b) < 1.5 * 2^29 (normalised + one "_1" spread), limbs(c) < 2^29, limbs(d) < 2^28:
//   column sum <= 14 (2.25 * 2^58 + 2^57) + 14 * 2^56 + 2^37 < 2^63.5;  value < (a b + c d) / 2^392 + p < 2p  =>  N.
__device__ __forceinline__ Fq28 fq28_mul2(const Fq28& a, const Fq28& b, const Fq28& c, const Fq28& d) {
    Fq28 r;
    uint32_t m[14];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)c.l[i] * d.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Fq28Consts::P[k - i];
        m[k] = (0u - (uint32_t)acc) & M28;
        acc = (acc + m[k]) >> 28;
    }
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (uint64_t)c.l[i] * d.l[k - i];
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (uint64_t)m[i] * Fq28Consts::P[k - i];
        r.l[k - 14] = (uint32_t)acc & M28;
        acc >>= 28;
    }
    r.l[13] = (uint32_t)acc;
    return r;
}
// a^2 2^-392 mod p with the cross products computed once against the doubled operand: 105 instead of 196 partial
// products for the a*a half.  Requires limbs < 0.75 * 2^30 (every caller: normalised + one spread subtraction, or
// 3 * normalised): column sum <= 7 * (1.5 * 2^30)(0.75 * 2^30) + (0.75 * 2^30)^2 + 14 * 2^56 < 2^64.
__device__ __forceinline__ Fq28 fq28_sqr(const Fq28& a) {
    Fq28 r;
    uint32_t m[14], a2[14];
#pragma unroll
    for (int i = 0; i < 14; i++) a2[i] = a.l[i] + a.l[i];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a2[i] * a.l[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Fq28Consts::P[k - i];
        m[k] = (0u - (uint32_t)acc) & M28;
        acc = (acc + m[k]) >> 28;
    }
#pragma unroll
    for (int k = 14; k < 27; k++) {
#pragma unroll
        for (int i = k - 13; 2 * i < k; i++) acc += (uint64_t)a2[i] * a.l[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = k - 13; i < 14; i++) acc += (uint64_t)m[i] * Fq28Consts::P[k - i];
        r.l[k - 14] = (uint32_t)acc & M28;
        acc >>= 28;
    }
    r.l[13] = (uint32_t)acc;
    return r;
}
struct MulInline {
    static __device__ __forceinline__ Fq28 mul(const Fq28& a, const Fq28& b) { return fq28_mul(a, b); }
    // the dedicated squarer (fq28_sqr, 105 instead of 196 a*a products) measured no faster inside the adders on
    // gfx950 (r01: 2.74 vs 2.63 ms per 2^20-point accumulation), so the policies square with the multiplier
    static __device__ __forceinline__ Fq28 sqr(const Fq28& a) { return fq28_mul(a, a); }
    static __device__ __forceinline__ Fq28 mul2(const Fq28& a, const Fq28& b, const Fq28& c, const Fq28& d) {
        return fq28_mul2(a, b, c, d);
    }
};
// Inline multiplier fenced by scheduling barriers: stops the compiler from interleaving independent multiplications
// of a group operation (which buys no ILP on an in-order SIMD but doubles the live registers and forces spills).
struct MulFenced {
    static __device__ __forceinline__ Fq28 mul(const Fq28& a, const Fq28& b) {
        __builtin_amdgcn_sched_barrier(0);
        Fq28 r = fq28_mul(a, b);
        __builtin_amdgcn_sched_barrier(0);
        return r;
    }
    static __device__ __forceinline__ Fq28 sqr(const Fq28& a) {
        __builtin_amdgcn_sched_barrier(0);
        Fq28 r = fq28_mul(a, a);
        __builtin_amdgcn_sched_barrier(0);
        return r;
    }
    static __device__ __forceinline__ Fq28 mul2(const Fq28& a, const Fq28& b, const Fq28& c, const Fq28& d) {
        __builtin_amdgcn_sched_barrier(0);
        Fq28 r = fq28_mul2(a, b, c, d);
        __builtin_amdgcn_sched_barrier(0);
        return r;
    }
};

// the asm multiplier: one statement per product is its own fence
struct MulAsm {
    static __device__ __forceinline__ Fq28 mul(const Fq28& a, const Fq28& b) { return fq28_mul_asm(a, b); }
    static __device__ __forceinline__ Fq28 sqr(const Fq28& a) { return fq28_mul_asm(a, a); }
    static __device__ __forceinline__ Fq28 mul2(const Fq28& a, const Fq28& b, const Fq28& c, const Fq28& d) {
        return MulFenced::mul2(a, b, c, d);
    }
};

// value == 0 mod p for an N value (normalised limbs, value < 2p): all limbs zero, or equal to p
__device__ __forceinline__ bool fq28_is_zero_mod_p(const Fq28& a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        z |= a.l[i];
        e |= a.l[i] ^ Fq28Consts::P[i];
    }
    return z == 0 || e == 0;
}
// N value (< 2p) -> canonical (< p), normalised limbs
__device__ __forceinline__ Fq28 fq28_canonical(const Fq28& a) {
    // t = a - p with borrow chain
    Fq28 t;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        uint32_t d = a.l[i] - Fq28Consts::P[i] - borrow;
        borrow = (d >> 31) & 1;  // limbs < 2^28 (top < 2^15): a negative difference sets bit 31
        t.l[i] = i < 13 ? (d & M28) : d;
    }
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = borrow ? a.l[i] : t.l[i];
    return r;
}
__device__ __forceinline__ Fq28 fq28_const(const uint32_t (&c)[14]) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = c[i];
    return r;
}

// ------------------------------------------------------------------------------------------------ XYZZ points in the 28-bit domain
// Coordinates are field elements times 2^392.  Invariants between operations ("point form"):
//   x: normalised limbs, value < 18p;  y: normalised limbs, value < 6p;  zz, zzz: N (value < 2p);  identity: zz == 0 exactly.
// In memory a point is a G1XYZZ whose four 384-bit slots hold these (non-canonical, < 2^382) integers.
struct P28 {
    Fq28 x, y, zz, zzz;
};
__device__ __forceinline__ bool p28_is_identity(const P28& p) {
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) z |= p.zz.l[i];
    return z == 0;
}
__device__ __forceinline__ P28 p28_identity() {
    P28 r;
    r.x = fq28_const(Fq28Consts::ONE);
    r.y = r.x;
#pragma unroll
    for (int i = 0; i < 14; i++) r.zz.l[i] = r.zzz.l[i] = 0;
    return r;
}
template <class XYZZ>
__device__ __forceinline__ P28 p28_load(const XYZZ& m) {
    P28 r;
    r.x = fq28_unpack(m.x);
    r.y = fq28_unpack(m.y);
    r.zz = fq28_unpack(m.zz);
    r.zzz = fq28_unpack(m.zzz);
    return r;
}
template <class XYZZ>
__device__ __forceinline__ void p28_store(XYZZ& m, const P28& p) {
    m.x = fq28_pack(p.x);
    m.y = fq28_pack(p.y);
    m.zz = fq28_pack(p.zz);
    m.zzz = fq28_pack(p.zzz);
}

// doubling (EFD dbl-2008-s-1, a = 0); identity stays identity (zz, zzz are multiplied)
template <class M = MulInline>
__device__ __forceinline__ P28 p28_dbl(const P28& p) {
    P28 r;
    Fq28 u = fq28_add(p.y, p.y);             // limbs < 2^29, value < 12p
    Fq28 v = M::sqr(u);                    // N
    Fq28 w = M::mul(u, v);                 // N
    Fq28 s = M::mul(p.x, v);               // N
    Fq28 xx = M::sqr(p.x);                 // N
    Fq28 m = fq28_add(fq28_add(xx, xx), xx); // limbs < 3 * 2^28, value < 6p
    Fq28 mm = M::sqr(m);                   // N
    Fq28 x3;
#pragma unroll
    for (int i = 0; i < 14; i++) x3.l[i] = mm.l[i] + Fq28Consts::SPREAD16_3[i] - s.l[i] - s.l[i];
    r.x = fq28_normalize(x3);                                 // (12p, 18p)
    Fq28 ny;                                                  // 8p - Y1 > 0 (Y1 < 6p), limbs < 2^29
#pragma unroll
    for (int i = 0; i < 14; i++) ny.l[i] = Fq28Consts::SPREAD8[i] - p.y.l[i];
    r.y = M::mul2(m, FQ28_SUB(s, r.x, SPREAD32), ny, w);      // M (S - X3) - W Y1, one reduction; N
    r.zz = M::mul(v, p.zz);
    r.zzz = M::mul(w, p.zzz);
    return r;
}

// acc += q (EFD add-2008-s).  Returns false when the generic formula does not apply (q = +-acc): the caller falls
// back to p28_add_slow.  Identity operands are handled here.
template <class M = MulInline>
__device__ __forceinline__ bool p28_add_fast(P28& a, const P28& q) {
    if (p28_is_identity(q)) return true;
    if (p28_is_identity(a)) {
        a = q;
        return true;
    }
    Fq28 u1 = M::mul(a.x, q.zz), u2 = M::mul(q.x, a.zz);
    Fq28 s1 = M::mul(a.y, q.zzz), s2 = M::mul(q.y, a.zzz);
    Fq28 p = FQ28_SUB(u2, u1, SPREAD4);   // limbs < 2^30, value in (2p, 6p)
    Fq28 r = FQ28_SUB(s2, s1, SPREAD4);
    Fq28 pp = M::sqr(p);
    if (fq28_is_zero_mod_p(pp)) return false;
    Fq28 ppp = M::mul(p, pp);
    Fq28 qq = M::mul(u1, pp);
    Fq28 rr = M::sqr(r);
    Fq28 x3;
#pragma unroll
    for (int i = 0; i < 14; i++) x3.l[i] = rr.l[i] + Fq28Consts::SPREAD16_3[i] - ppp.l[i] - qq.l[i] - qq.l[i];
    x3 = fq28_normalize(x3);              // (10p, 18p)
    Fq28 ns1;                                                 // 4p - S1 > 0 (S1 < 2p), limbs < 2^29
#pragma unroll
    for (int i = 0; i < 14; i++) ns1.l[i] = Fq28Consts::SPREAD4[i] - s1.l[i];
    a.y = M::mul2(r, FQ28_SUB(qq, x3, SPREAD32), ns1, ppp);   // R (Q - X3) - S1 PPP, one reduction; N
    a.x = x3;
    a.zz = M::mul(M::mul(a.zz, q.zz), pp);
    a.zzz = M::mul(M::mul(a.zzz, q.zzz), ppp);
    return true;
}
// q = +-acc: doubling or cancellation, decided on R^2.  Cold, out of line, operands BY VALUE: a by-reference
// parameter would make the caller's accumulators address-taken and pin them in scratch memory for good.
template <class M = MulInline>
__device__ __noinline__ P28 p28_add_slow(const P28 a, const P28 q) {
    Fq28 s1 = M::mul(a.y, q.zzz), s2 = M::mul(q.y, a.zzz);
    Fq28 d = FQ28_SUB(s2, s1, SPREAD4);
    Fq28 rr = M::sqr(d);
    if (fq28_is_zero_mod_p(rr)) return p28_dbl<M>(a);
    return p28_identity();
}
template <class M = MulInline>
__device__ __forceinline__ void p28_add(P28& a, const P28& q) {
    if (!p28_add_fast<M>(a, q)) a = p28_add_slow<M>(a, q);
}

// 28-bit domain (x 2^392) -> memory format of the rest of the library: Montgomery radix 2^384, canonical
template <class XYZZ>
__device__ __forceinline__ void p28_store_384(XYZZ& m, const P28& p) {
    if (p28_is_identity(p)) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            m.x.v[i] = FqParams::R1[i];
            m.y.v[i] = FqParams::R1[i];
            m.zz.v[i] = 0;
            m.zzz.v[i] = 0;
        }
        return;
    }
    Fq28 k = fq28_const(Fq28Consts::TO384);
    m.x = fq28_pack(fq28_canonical(fq28_mul(p.x, k)));
    m.y = fq28_pack(fq28_canonical(fq28_mul(p.y, k)));
    m.zz = fq28_pack(fq28_canonical(fq28_mul(p.zz, k)));
    m.zzz = fq28_pack(fq28_canonical(fq28_mul(p.zzz, k)));
}

// ------------------------------------------------------------------------------------------------ twisted Edwards points in the 28-bit domain
// (the curve form, its constants and the reason for it: g1.cuh, "twisted Edwards form").  Coordinates are field elements
// times 2^392; every coordinate of a point at rest is N (what fq28_mul returns).  In memory a point is a G1XYZZ whose
// slots hold X, Y, T, Z (x, y, zz, zzz).  The law is unified: no identity / doubling / cancellation cases anywhere.
struct Fq28TeConsts {
    static constexpr uint32_t TWO[14] = SWM_FQ28_TWO;
    static constexpr uint32_t K2D[14] = SWM_FQ28_TE_2D;
    static constexpr uint32_t INVD[14] = SWM_FQ28_TE_INVD;
};
struct T28 {
    Fq28 x, y, t, z;
};
#ifndef SWM_TE_EARLY_LOADS
#define SWM_TE_EARLY_LOADS 1
#endif
// one coordinate of a table row (G1TE: fourteen limbs in a 64-byte sector): three 16-byte loads and an 8-byte one
__device__ __forceinline__ Fq28 te28_load_coord(const uint32_t* __restrict__ p) {
    const uint4 v0 = *reinterpret_cast<const uint4*>(p), v1 = *reinterpret_cast<const uint4*>(p + 4),
                v2 = *reinterpret_cast<const uint4*>(p + 8);
    const uint2 v3 = *reinterpret_cast<const uint2*>(p + 12);
    Fq28 r;
    r.l[0] = v0.x, r.l[1] = v0.y, r.l[2] = v0.z, r.l[3] = v0.w;
    r.l[4] = v1.x, r.l[5] = v1.y, r.l[6] = v1.z, r.l[7] = v1.w;
    r.l[8] = v2.x, r.l[9] = v2.y, r.l[10] = v2.z, r.l[11] = v2.w;
    r.l[12] = v3.x, r.l[13] = v3.y;
    return r;
}
// acc += +-P for the table row (m2, s2, k2) = (y2 - x2, y2 + x2, 2 d x2 y2) of P: EFD madd-2008-hwcd-3 with the row
// precomputed, 7 multiplications.  The row of -P is (s2, m2, -k2): a NEGATIVE digit loads the first two coordinates from each
// other's address (no select on limbs) and takes the sign of C = T1 k2 in the choice between D - C and D + C.
// Bounds (N: limbs < 2^28, value < 2p; the row: canonical):
//   Y1 - X1 + 4p: limbs < 2^30, < 6p  |  Y1 + X1 < 4p  |  A, B, C = products: N  |  D = 2 Z1: limbs < 2^29, < 4p
//   B - A + 4p, D - C + 4p: limbs < 2^30, < 8p  |  D + C: limbs < 3 2^28, < 6p  |  H = B + A: limbs < 2^29, < 4p
//   X3 = E F, Y3 = G H, T3 = E H, Z3 = F G: operands within what fq28_mul accepts (limbs < 2^30, value < 128p) -> N.
// The same bounds hold for the first point of a segment (te28_from_row: X1 normalised, in (p, 3p); a SPREADk subtraction
// needs the top limb of what it subtracts below the top limb of k p: 13.8 k for an N value, 27.5 k for 4p).
// (r03 took the sign on the accumulator side — the same seven products, hence the same limbs — with four 14-limb selects.)
template <class M = MulAsm>
__device__ __forceinline__ void te28_madd_row(T28& a, const G1TE* __restrict__ rp, bool neg) {
    const uint32_t* pa = neg ? rp->ypx : rp->ymx;
    const uint32_t* pb = neg ? rp->ymx : rp->ypx;
    Fq28 d, s;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        d.l[i] = a.y.l[i] + Fq28Consts::SPREAD4[i] - a.x.l[i];
        s.l[i] = a.y.l[i] + a.x.l[i];
    }
#if SWM_TE_EARLY_LOADS
    // all three coordinates of the row are requested before the first product: one exposed memory round trip per addition
    // instead of three (the compiler otherwise sinks each coordinate's loads to the product that consumes it, and with tables
    // of tens of GB — 2^22-constraint keys — every round trip is a TLB miss and a DRAM page miss: measured r04)
    const Fq28 ra = te28_load_coord(pa), rb = te28_load_coord(pb), rk = te28_load_coord(rp->kt);
    __builtin_amdgcn_sched_barrier(0);
    const Fq28 A = M::mul(d, ra);
    const Fq28 B = M::mul(s, rb);
    const Fq28 C = M::mul(a.t, rk);
#else
    const Fq28 A = M::mul(d, te28_load_coord(pa));
    const Fq28 B = M::mul(s, te28_load_coord(pb));
    const Fq28 C = M::mul(a.t, te28_load_coord(rp->kt));
#endif
    Fq28 E, H, F, G;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint32_t sp = Fq28Consts::SPREAD4[i];
        E.l[i] = B.l[i] + sp - A.l[i];
        H.l[i] = A.l[i] + B.l[i];
        const uint32_t D = a.z.l[i] + a.z.l[i];
        const uint32_t dm = D + sp - C.l[i], dp = D + C.l[i];
        F.l[i] = neg ? dp : dm;
        G.l[i] = neg ? dm : dp;
    }
    a.x = M::mul(E, F);
    a.y = M::mul(G, H);
    a.t = M::mul(E, H);
    a.z = M::mul(F, G);
}
// the point a row stands for, as an accumulator: (X : Y : T : Z) = (2x : 2y : 2xy : 2) — one multiplication (by 1/d)
// instead of an addition to the identity.  neg: the row of -P is (s2, m2, -k2).
__device__ __forceinline__ T28 te28_from_row(const Fq28& m2, const Fq28& s2, const Fq28& k2, bool neg) {
    T28 r;
    Fq28 d;  // +-(s2 - m2) + 2p = +-2x: in (p, 3p), top limb below SPREAD4's (what the next Y1 - X1 + 4p needs)
#pragma unroll
    for (int i = 0; i < 14; i++) d.l[i] = Fq28Consts::SPREAD2[i] + (neg ? m2.l[i] - s2.l[i] : s2.l[i] - m2.l[i]);
    r.x = fq28_normalize(d);
    r.y = fq28_normalize(fq28_add(s2, m2));    // 2y < 2p
    Fq28 ks;
#pragma unroll
    for (int i = 0; i < 14; i++) ks.l[i] = neg ? Fq28Consts::SPREAD4[i] - k2.l[i] : k2.l[i];
    r.t = fq28_mul(ks, fq28_const(Fq28TeConsts::INVD));  // 2 d x y / d
    r.z = fq28_const(Fq28TeConsts::TWO);
    return r;
}
__device__ __forceinline__ void te28_store_identity(G1XYZZ& m) {
    Fq28 z, one = fq28_const(Fq28Consts::ONE);
#pragma unroll
    for (int i = 0; i < 14; i++) z.l[i] = 0;
    m.x = fq28_pack(z);
    m.y = fq28_pack(one);
    m.zz = fq28_pack(z);
    m.zzz = fq28_pack(one);
}
// *dst = *pa + *pq, operands in memory (LDS slots or HBM), streamed like p28_slot_add: 9 multiplications, four field
// elements live at the peak.  dst may be pa or pq: every load precedes the first store.
template <class M = MulAsm>
__device__ __forceinline__ void te28_slot_add(G1XYZZ* dst, const G1XYZZ* pa, const G1XYZZ* pq) {
    Fq28 A, B;
    {
        Fq28 ax = fq28_unpack(pa->x), ay = fq28_unpack(pa->y), qx = fq28_unpack(pq->x), qy = fq28_unpack(pq->y);
        A = M::mul(FQ28_SUB(ay, ax, SPREAD4), FQ28_SUB(qy, qx, SPREAD4));
        B = M::mul(fq28_add(ay, ax), fq28_add(qy, qx));
    }
    Fq28 C = M::mul(M::mul(fq28_unpack(pa->zz), fq28_unpack(pq->zz)), fq28_const(Fq28TeConsts::K2D));
    Fq28 D = M::mul(fq28_unpack(pa->zzz), fq28_unpack(pq->zzz));
    D = fq28_add(D, D);
    Fq28 E = FQ28_SUB(B, A, SPREAD4), H = fq28_add(B, A);
    Fq28 F = FQ28_SUB(D, C, SPREAD4), G = fq28_add(D, C);
    dst->x = fq28_pack(M::mul(E, F));
    dst->zz = fq28_pack(M::mul(E, H));
    dst->y = fq28_pack(M::mul(G, H));
    dst->zzz = fq28_pack(M::mul(F, G));
}
// te28_slot_add for steps in which a lane reads ANOTHER lane's slot that the same step rewrites (the in-place suffix scan of
// msm_bucket_reduce_low): every load of the workgroup precedes a barrier, every store follows it.  `act` = the lane takes part;
// all lanes of the workgroup must reach the call with the same `barrier`.  Same products, same limbs as te28_slot_add.
template <class M = MulAsm>
__device__ __forceinline__ void te28_slot_add_sync(G1XYZZ* dst, const G1XYZZ* pa, const G1XYZZ* pq, bool act, bool barrier = true) {
    Fq28 E, F, G, H;
#pragma unroll
    for (int i = 0; i < 14; i++) E.l[i] = F.l[i] = G.l[i] = H.l[i] = 0;
    if (act) {
        Fq28 A, B;
        {
            Fq28 ax = fq28_unpack(pa->x), ay = fq28_unpack(pa->y), qx = fq28_unpack(pq->x), qy = fq28_unpack(pq->y);
            A = M::mul(FQ28_SUB(ay, ax, SPREAD4), FQ28_SUB(qy, qx, SPREAD4));
            B = M::mul(fq28_add(ay, ax), fq28_add(qy, qx));
        }
        Fq28 C = M::mul(M::mul(fq28_unpack(pa->zz), fq28_unpack(pq->zz)), fq28_const(Fq28TeConsts::K2D));
        Fq28 D = M::mul(fq28_unpack(pa->zzz), fq28_unpack(pq->zzz));
        D = fq28_add(D, D);
        E = FQ28_SUB(B, A, SPREAD4), H = fq28_add(B, A);
        F = FQ28_SUB(D, C, SPREAD4), G = fq28_add(D, C);
    }
    if (barrier) __syncthreads();  // (uniform over the workgroup; false for steps in which every lane works on slots of its own)
    if (act) {
        dst->x = fq28_pack(M::mul(E, F));
        dst->zz = fq28_pack(M::mul(E, H));
        dst->y = fq28_pack(M::mul(G, H));
        dst->zzz = fq28_pack(M::mul(F, G));
    }
}
// The same addition by FOUR lanes (a quad of one wave; `q` = the lane's index in it): the nine multiplications of
// te28_slot_add as three rounds of one product per lane — (A, B, T1 T2, Z1 Z2), then 2d (T1 T2) on lane 2, then (X3, Y3, T3,
// Z3) — with the four intermediate values passed around the quad by DPP moves and every lane storing one coordinate of the
// result.  Same formulas, same limbs.  For the bucket stage of SMALL MSMs (r04): that stage is a dependent chain of group
// operations on a chip that is mostly idle, and a lone wave issues an instruction every ~5.5 cycles whatever it computes —
// three products per step instead of nine is a step of ~4.5 us instead of ~12.  (For large MSMs every CU already has its
// 256 chains: there the quad form would only add instructions.)  Every lane of the quad is handed the same pointers; all loads
// precede the stores (one wave: lock step), so dst may be pa or pq.
template <int SRC>
__device__ __forceinline__ Fq28 te28_quad_get(const Fq28& v) {  // every lane of the quad: the value held by its lane SRC
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++)  // DPP quad_perm [SRC, SRC, SRC, SRC]: a VALU move, no LDS traffic
        r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], SRC * 0x55, 0xf, 0xf, true);
    return r;
}
template <class M = MulAsm>
__device__ __forceinline__ void te28_quad_add(G1XYZZ* dst, const G1XYZZ* pa, const G1XYZZ* pq, unsigned q) {
    // round 1 operands: lane 0: (Y1 - X1)(Y2 - X2), lane 1: (Y1 + X1)(Y2 + X2), lane 2: T1 T2, lane 3: Z1 Z2 — written as
    // a + s b with (a, b) = (y, x), (y, x), (zz, -), (zzz, -) so that the four lanes run the same instructions
    const unsigned ia = q < 2 ? 1u : q;  // index of coordinate a among (x, y, zz, zzz)
    const Fq* ca = reinterpret_cast<const Fq*>(pa);
    const Fq* cq = reinterpret_cast<const Fq*>(pq);
    const Fq28 a1 = fq28_unpack(ca[ia]), b1 = fq28_unpack(ca[0]), a2 = fq28_unpack(cq[ia]), b2 = fq28_unpack(cq[0]);
    Fq28 u, v;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint32_t sp = Fq28Consts::SPREAD4[i];
        const uint32_t t1 = q == 0 ? sp - b1.l[i] : (q == 1 ? b1.l[i] : 0u), t2 = q == 0 ? sp - b2.l[i] : (q == 1 ? b2.l[i] : 0u);
        u.l[i] = a1.l[i] + t1;
        v.l[i] = a2.l[i] + t2;
    }
    Fq28 p1 = M::mul(u, v);
    const Fq28 p2 = M::mul(p1, fq28_const(Fq28TeConsts::K2D));  // only lane 2 keeps it: C = 2d T1 T2
#pragma unroll
    for (int i = 0; i < 14; i++) p1.l[i] = q == 2 ? p2.l[i] : (q == 3 ? p1.l[i] + p1.l[i] : p1.l[i]);  // lane 3: D = 2 Z1 Z2
    const Fq28 A = te28_quad_get<0>(p1), B = te28_quad_get<1>(p1), C = te28_quad_get<2>(p1), D = te28_quad_get<3>(p1);
    Fq28 l, r;  // lane 0: E F, lane 1: G H, lane 2: E H, lane 3: F G
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint32_t sp = Fq28Consts::SPREAD4[i];
        const uint32_t E = B.l[i] + sp - A.l[i], H = B.l[i] + A.l[i], F = D.l[i] + sp - C.l[i], G = D.l[i] + C.l[i];
        l.l[i] = q == 1 ? G : (q == 3 ? F : E);
        r.l[i] = q == 0 ? F : (q == 3 ? G : H);
    }
    reinterpret_cast<Fq*>(dst)[q == 2 ? 2u : (q == 3 ? 3u : q)] = fq28_pack(M::mul(l, r));  // X3, Y3, T3 (slot zz), Z3 (slot zzz)
}
// Mixed addition by a quad: the accumulator lives ONE COORDINATE PER LANE (`own`: X, Y, T, Z on lanes 0 .. 3), and the seven
// products of te28_madd_row take two rounds — (A, B, C | D = 2 Z1 needs none), then (X3, Y3, T3, Z3).  Same formulas and limbs as
// te28_madd_row.  For the accumulation of SMALL MSMs: a segment's chain of additions is 2 instead of 7 products per entry long.
template <class M = MulAsm>
__device__ __forceinline__ void te28_quad_madd_row(Fq28& own, const G1TE* __restrict__ rp, bool neg, unsigned q) {
    // lane 0 multiplies by y - x of the row (y + x when the point is subtracted), lane 1 by the other one, lane 2 by 2dxy
    const uint32_t* pr = q == 2 ? rp->kt : ((q == 0) != neg ? rp->ymx : rp->ypx);
    const Fq28 v = te28_load_coord(pr);
    Fq28 u;  // lane 0: Y1 - X1 + 4p, lane 1: Y1 + X1, lane 2: T1 (lane 3 idles through the product)
#pragma unroll
    for (int i = 0; i < 14; i++) {
        // the neighbour's coordinate: quad_perm [1, 0, 3, 2]
        const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own.l[i], 0xB1, 0xf, 0xf, true);
        u.l[i] = q == 0 ? other + Fq28Consts::SPREAD4[i] - own.l[i] : (q == 1 ? own.l[i] + other : own.l[i]);
    }
    Fq28 p1 = M::mul(u, v);
#pragma unroll
    for (int i = 0; i < 14; i++) p1.l[i] = q == 3 ? own.l[i] + own.l[i] : p1.l[i];  // D = 2 Z1
    const Fq28 A = te28_quad_get<0>(p1), B = te28_quad_get<1>(p1), C = te28_quad_get<2>(p1), D = te28_quad_get<3>(p1);
    Fq28 l, r;  // lane 0: E F, lane 1: G H, lane 2: E H, lane 3: F G
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint32_t sp = Fq28Consts::SPREAD4[i];
        const uint32_t E = B.l[i] + sp - A.l[i], H = A.l[i] + B.l[i];
        const uint32_t dm = D.l[i] + sp - C.l[i], dp = D.l[i] + C.l[i];
        const uint32_t F = neg ? dp : dm, G = neg ? dm : dp;
        l.l[i] = q == 1 ? G : (q == 3 ? F : E);
        r.l[i] = q == 0 ? F : (q == 3 ? G : H);
    }
    own = M::mul(l, r);
}
// the first point of a segment (te28_from_row), one coordinate per lane
template <class M = MulAsm>
__device__ __forceinline__ Fq28 te28_quad_from_row(const G1TE* __restrict__ rp, bool neg, unsigned q) {
    const Fq28 a = te28_load_coord(q == 2 ? rp->kt : rp->ymx), s2 = te28_load_coord(rp->ypx);  // a: m2, or k2 on lane 2
    Fq28 d, ks;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        d.l[i] = q == 0 ? Fq28Consts::SPREAD2[i] + (neg ? a.l[i] - s2.l[i] : s2.l[i] - a.l[i]) : s2.l[i] + a.l[i];
        ks.l[i] = neg ? Fq28Consts::SPREAD4[i] - a.l[i] : a.l[i];
    }
    d = fq28_normalize(d);  // lane 0: +-2x in (p, 3p), lane 1: 2y
    const Fq28 t = M::mul(ks, fq28_const(Fq28TeConsts::INVD));  // lane 2: 2 d x y / d
    Fq28 r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = q < 2 ? d.l[i] : (q == 2 ? t.l[i] : Fq28TeConsts::TWO[i]);
    return r;
}
// one coordinate per lane of the quad
__device__ __forceinline__ void te28_quad_copy(G1XYZZ* dst, const G1XYZZ* src, unsigned q) {
    reinterpret_cast<Fq*>(dst)[q] = reinterpret_cast<const Fq*>(src)[q];
}
__device__ __forceinline__ void te28_quad_store_identity(G1XYZZ* dst, unsigned q) {
    Fq28 z;
#pragma unroll
    for (int i = 0; i < 14; i++) z.l[i] = (q & 1u) ? Fq28Consts::ONE[i] : 0u;  // (0 : 1 : 0 : 1)
    reinterpret_cast<Fq*>(dst)[q] = fq28_pack(z);
}
// 28-bit domain (x 2^392) -> radix 2^384, canonical; still a twisted Edwards point (the host folds in that form)
__device__ __forceinline__ void te28_store_384(G1XYZZ& m, const G1XYZZ& slot) {
    Fq28 k = fq28_const(Fq28Consts::TO384);
    m.x = fq28_pack(fq28_canonical(fq28_mul(fq28_unpack(slot.x), k)));
    m.y = fq28_pack(fq28_canonical(fq28_mul(fq28_unpack(slot.y), k)));
    m.zz = fq28_pack(fq28_canonical(fq28_mul(fq28_unpack(slot.zz), k)));
    m.zzz = fq28_pack(fq28_canonical(fq28_mul(fq28_unpack(slot.zzz), k)));
}

}  // namespace swm
