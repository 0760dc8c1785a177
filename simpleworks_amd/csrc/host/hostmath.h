// hostmath.h — host-side helpers on top of ff.cuh / g1.cuh: byte codecs, square roots, scalar multiplication,
// ark-ff orderings.  Used by the Marlin host logic (transcript, serialisation, verifier, setup), never on the
// bulk path (bulk = HIP kernels).
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../g1.cuh"
#include "pairing.h"

namespace swm {

// ------------------------------------------------------------------------------------------------ limbs / bytes
template <class F> inline F fp_from_limbs(const uint32_t* l) {
    F r;
    for (int i = 0; i < F::N; i++) r.v[i] = l[i];
    return r;
}
// ark-ff ToBytes / CanonicalSerialize of a field element: little-endian bytes of the STANDARD-form integer
template <class F> inline void fp_to_bytes(const F& a_mont, uint8_t* out) {
    F s = fp_to_std(a_mont);
    for (int i = 0; i < F::N; i++) {
        out[4 * i] = (uint8_t)s.v[i];
        out[4 * i + 1] = (uint8_t)(s.v[i] >> 8);
        out[4 * i + 2] = (uint8_t)(s.v[i] >> 16);
        out[4 * i + 3] = (uint8_t)(s.v[i] >> 24);
    }
}
// returns false when the integer is >= p
template <class F> inline bool fp_from_bytes(const uint8_t* in, F* out_mont) {
    F s;
    for (int i = 0; i < F::N; i++)
        s.v[i] = (uint32_t)in[4 * i] | ((uint32_t)in[4 * i + 1] << 8) | ((uint32_t)in[4 * i + 2] << 16) |
                 ((uint32_t)in[4 * i + 3] << 24);
    bool lt = false;
    for (int i = F::N - 1; i >= 0; i--) {
        if (s.v[i] < F::Params::P[i]) { lt = true; break; }
        if (s.v[i] > F::Params::P[i]) break;
    }
    if (!lt) return false;
    *out_mont = fp_from_std(s);
    return true;
}
// ark-ff Ord on field elements compares the standard-form integers
template <class F> inline int fp_cmp(const F& a_mont, const F& b_mont) {
    return fp_cmp_std(fp_to_std(a_mont), fp_to_std(b_mont));
}
inline Fr fr_from_u128(const uint64_t v[2]) {
    Fr s = fp_zero<Fr>();
    s.v[0] = (uint32_t)v[0];
    s.v[1] = (uint32_t)(v[0] >> 32);
    s.v[2] = (uint32_t)v[1];
    s.v[3] = (uint32_t)(v[1] >> 32);
    return fp_from_std(s);
}
inline Fr fr_pow_u64(const Fr& a, uint64_t e) {
    uint32_t l[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
    return fp_pow(a, l, 2);
}

// ------------------------------------------------------------------------------------------------ square roots
// Tonelli-Shanks in Fq (two-adicity 46); returns false for non-residues.  Which root is returned is unspecified:
// callers pick by ark-ff's ordering rule.
inline bool fq_sqrt(const Fq& a, Fq* out) {
    if (fp_is_zero(a)) {
        *out = a;
        return true;
    }
    static const uint32_t half[12] = SWM_FQ_PM1_HALF, tt[12] = SWM_FQ_TS_T, tp1h[12] = SWM_FQ_TS_T_PLUS1_HALF,
                          cm[12] = SWM_FQ_TS_C_MONT;
    if (!fp_is_one(fp_pow(a, half, 12))) return false;
    Fq c = fp_from_limbs<Fq>(cm);
    Fq x = fp_pow(a, tp1h, 12);
    Fq b = fp_pow(a, tt, 12);
    int m = SWM_FQ_TWO_ADICITY;
    while (!fp_is_one(b)) {
        int i = 0;
        Fq bb = b;
        while (!fp_is_one(bb)) {
            bb = fp_sqr(bb);
            i++;
        }
        Fq g = c;
        for (int k = 0; k < m - i - 1; k++) g = fp_sqr(g);
        x = fp_mul(x, g);
        c = fp_sqr(g);
        b = fp_mul(b, c);
        m = i;
    }
    *out = x;
    return true;
}
inline bool fq_is_qr(const Fq& a) {
    static const uint32_t half[12] = SWM_FQ_PM1_HALF;
    return fp_is_zero(a) || fp_is_one(fp_pow(a, half, 12));
}
// sqrt in Fq2 = Fq[u]/(u^2+5) by the norm method
inline bool fq2_sqrt(const Fq2& a, Fq2* out) {
    if (a.is_zero()) {
        *out = a;
        return true;
    }
    Fq norm = fp_add(fp_sqr(a.c0), fq_mul_small(fp_sqr(a.c1), 5));
    Fq alpha;
    if (!fq_sqrt(norm, &alpha)) return false;
    Fq two_inv = fp_inv(fp_add(fp_one<Fq>(), fp_one<Fq>()));
    Fq delta = fp_mul(fp_add(a.c0, alpha), two_inv);
    if (!fq_is_qr(delta)) delta = fp_mul(fp_sub(a.c0, alpha), two_inv);
    Fq c0;
    if (!fq_sqrt(delta, &c0)) return false;
    Fq2 r;
    if (fp_is_zero(c0)) {
        // a = c1^2 * u^2 = -5 c1^2
        Fq m5 = fp_neg(fq_mul_small(fp_one<Fq>(), 5));
        Fq c1;
        if (!fq_sqrt(fp_mul(a.c0, fp_inv(m5)), &c1)) return false;
        r = {fp_zero<Fq>(), c1};
    } else {
        r = {c0, fp_mul(a.c1, fp_inv(fp_dbl(c0)))};
    }
    if (!(r.square() == a)) return false;
    *out = r;
    return true;
}
// ark-ff QuadExtField Ord: compare c1 first, then c0 [U]
inline bool fq2_less(const Fq2& a, const Fq2& b) {
    int c = fp_cmp(a.c1, b.c1);
    if (c != 0) return c < 0;
    return fp_cmp(a.c0, b.c0) < 0;
}

// ------------------------------------------------------------------------------------------------ G1 host ops
// k given as little-endian 32-bit limbs (standard integer)
inline G1XYZZ g1_mul_limbs(const G1Affine& p, const uint32_t* k, int limbs) {
    G1XYZZ acc = g1_xyzz_identity();
    bool started = false;
    for (int i = limbs * 32 - 1; i >= 0; i--) {
        if (started) acc = g1_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) {
            g1_add_mixed(acc, p);
            started = true;
        }
    }
    return acc;
}
inline G1Affine g1_mul_fr(const G1Affine& p, const Fr& k_mont) {
    Fr s = fp_to_std(k_mont);
    return g1_to_affine(g1_mul_limbs(p, s.v, 8));
}
// sum_i k_i P_i on the host (the verifier's linear combinations of commitments): Straus — one chain of doublings for all
// terms, 4-bit windows, 15 multiples per point.  The same group element as term-by-term double-and-add at a third of the cost
// from a dozen terms on.
inline G1XYZZ g1_msm_host(const std::vector<std::pair<G1Affine, Fr>>& terms) {
    struct Entry {
        G1XYZZ mult[16];
        Fr k;
    };
    std::vector<Entry> es;
    for (auto& t : terms) {
        if (g1_is_inf(t.first) || fp_is_zero(t.second)) continue;
        es.emplace_back();
        Entry& e = es.back();
        e.k = fp_to_std(t.second);
        e.mult[0] = g1_xyzz_identity();
        e.mult[1] = g1_from_affine(t.first);
        for (int m = 2; m < 16; m++) {
            e.mult[m] = e.mult[m - 1];
            g1_add_mixed(e.mult[m], t.first);
        }
    }
    G1XYZZ acc = g1_xyzz_identity();
    for (int w = 63; w >= 0; w--) {  // 256 bits, most significant nibble first
        if (w != 63)
            for (int d = 0; d < 4; d++) acc = g1_dbl(acc);
        for (auto& e : es) {
            const unsigned nib = (e.k.v[w >> 3] >> ((w & 7) * 4)) & 15u;
            if (nib) g1_add(acc, e.mult[nib]);
        }
    }
    return acc;
}
inline G1Affine g1_add_affine(const G1Affine& a, const G1Affine& b) {
    G1XYZZ acc = g1_from_affine(a);
    g1_add_mixed(acc, b);
    return g1_to_affine(acc);
}
inline G1Affine g1_sub_affine(const G1Affine& a, const G1Affine& b) { return g1_add_affine(a, g1_is_inf(b) ? b : g1_neg(b)); }
inline G2Affine g2_mul_fr(const G2Affine& p, const Fr& k_mont) {
    Fr s = fp_to_std(k_mont);
    return g2_mul(p, s.v, 8);
}

}  // namespace swm
