// pairing.h — BLS12-377 extension tower, G2 and the ate pairing on the HOST (CPU C++).
// Needed by verify_proof (/root/reference/src/marlin/mod.rs:79-86 -> MarlinKZG10::check_combinations ->
// KZG10::batch_check: one product of two pairings) and by universal setup (a random G2 element and beta*h).
// The reference delegates to ark-ec 0.3 bls12 (not vendored); this is a restatement with a deliberately plain
// structure (affine Miller loop on the twist with a dense Fq12 product per line, cyclotomic squarings in the final
// exponentiation, no Frobenius-based addition chain): the verifier is milliseconds of host work, not a hot path.  Tower: Fq2 = Fq[u]/(u^2+5), Fq6 = Fq2[v]/(v^3-u), Fq12 = Fq6[w]/(w^2-v);
// G2: y^2 = x^3 + 1/u (D-type twist).
#pragma once
#include <vector>
#include "../g1.cuh"

namespace swm {

inline Fq fq_from_limbs(const uint32_t* l) {
    Fq r;
    for (int i = 0; i < 12; i++) r.v[i] = l[i];
    return r;
}
inline Fq fq_mul_small(const Fq& a, unsigned k) {  // k * a by repeated addition (k <= 8)
    Fq r = fp_zero<Fq>();
    for (unsigned i = 0; i < k; i++) r = fp_add(r, a);
    return r;
}

struct Fq2 {
    Fq c0, c1;
    static Fq2 zero() { return {fp_zero<Fq>(), fp_zero<Fq>()}; }
    static Fq2 one() { return {fp_one<Fq>(), fp_zero<Fq>()}; }
    bool is_zero() const { return fp_is_zero(c0) && fp_is_zero(c1); }
    bool operator==(const Fq2& o) const { return fp_eq(c0, o.c0) && fp_eq(c1, o.c1); }
    Fq2 operator+(const Fq2& o) const { return {fp_add(c0, o.c0), fp_add(c1, o.c1)}; }
    Fq2 operator-(const Fq2& o) const { return {fp_sub(c0, o.c0), fp_sub(c1, o.c1)}; }
    Fq2 operator-() const { return {fp_neg(c0), fp_neg(c1)}; }
    Fq2 operator*(const Fq2& o) const {  // u^2 = -5, Karatsuba
        Fq v0 = fp_mul(c0, o.c0), v1 = fp_mul(c1, o.c1);
        Fq s = fp_mul(fp_add(c0, c1), fp_add(o.c0, o.c1));
        return {fp_sub(v0, fq_mul_small(v1, 5)), fp_sub(fp_sub(s, v0), v1)};
    }
    Fq2 mul_fq(const Fq& k) const { return {fp_mul(c0, k), fp_mul(c1, k)}; }
    Fq2 square() const { return *this * *this; }
    Fq2 mul_by_nonresidue() const {  // * u : (c0 + c1 u) u = -5 c1 + c0 u
        return {fp_neg(fq_mul_small(c1, 5)), c0};
    }
    Fq2 inverse() const {
        Fq n = fp_add(fp_sqr(c0), fq_mul_small(fp_sqr(c1), 5));
        Fq ni = fp_inv(n);
        return {fp_mul(c0, ni), fp_neg(fp_mul(c1, ni))};
    }
};

struct Fq6 {
    Fq2 c0, c1, c2;
    static Fq6 zero() { return {Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
    static Fq6 one() { return {Fq2::one(), Fq2::zero(), Fq2::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
    bool operator==(const Fq6& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
    Fq6 operator+(const Fq6& o) const { return {c0 + o.c0, c1 + o.c1, c2 + o.c2}; }
    Fq6 operator-(const Fq6& o) const { return {c0 - o.c0, c1 - o.c1, c2 - o.c2}; }
    Fq6 operator-() const { return {-c0, -c1, -c2}; }
    Fq6 operator*(const Fq6& o) const {  // v^3 = u
        Fq2 a0b0 = c0 * o.c0, a1b1 = c1 * o.c1, a2b2 = c2 * o.c2;
        Fq2 t0 = (c1 + c2) * (o.c1 + o.c2) - a1b1 - a2b2;  // a1b2 + a2b1
        Fq2 t1 = (c0 + c1) * (o.c0 + o.c1) - a0b0 - a1b1;  // a0b1 + a1b0
        Fq2 t2 = (c0 + c2) * (o.c0 + o.c2) - a0b0 - a2b2;  // a0b2 + a2b0
        return {a0b0 + t0.mul_by_nonresidue(), t1 + a2b2.mul_by_nonresidue(), t2 + a1b1};
    }
    Fq6 mul_by_v() const { return {c2.mul_by_nonresidue(), c0, c1}; }
    Fq6 inverse() const {
        Fq2 t0 = c0.square() - (c1 * c2).mul_by_nonresidue();
        Fq2 t1 = c2.square().mul_by_nonresidue() - c0 * c1;
        Fq2 t2 = c1.square() - c0 * c2;
        Fq2 d = (c0 * t0 + (c2 * t1 + c1 * t2).mul_by_nonresidue()).inverse();
        return {t0 * d, t1 * d, t2 * d};
    }
};

struct Fq12 {
    Fq6 c0, c1;
    static Fq12 one() { return {Fq6::one(), Fq6::zero()}; }
    bool operator==(const Fq12& o) const { return c0 == o.c0 && c1 == o.c1; }
    bool is_one() const { return *this == one(); }
    Fq12 operator+(const Fq12& o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fq12 operator-(const Fq12& o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fq12 operator*(const Fq12& o) const {  // w^2 = v, Karatsuba
        Fq6 v0 = c0 * o.c0, v1 = c1 * o.c1;
        Fq6 s = (c0 + c1) * (o.c0 + o.c1);
        return {v0 + v1.mul_by_v(), s - v0 - v1};
    }
    Fq12 square() const { return *this * *this; }
    Fq12 conjugate() const { return {c0, -c1}; }
    Fq12 inverse() const {
        Fq6 d = (c0 * c0 - (c1 * c1).mul_by_v()).inverse();
        return {c0 * d, -(c1 * d)};
    }
    Fq12 pow(const uint32_t* e, int limbs) const {
        Fq12 acc = one();
        bool started = false;
        for (int i = limbs * 32 - 1; i >= 0; i--) {
            if (started) acc = acc.square();
            if ((e[i >> 5] >> (i & 31)) & 1) {
                acc = started ? acc * *this : *this;
                started = true;
            }
        }
        return acc;
    }
    static Fq12 from_fq(const Fq& a) { return {{{a, fp_zero<Fq>()}, Fq2::zero(), Fq2::zero()}, Fq6::zero()}; }
    static Fq12 from_fq2(const Fq2& a) { return {{a, Fq2::zero(), Fq2::zero()}, Fq6::zero()}; }
};

// ------------------------------------------------------------------------------------------------ G2 (affine over Fq2)
struct G2Affine {
    Fq2 x, y;
    bool inf;
};
inline Fq2 g2_coeff_b() {
    static const uint32_t c1[12] = SWM_G2_B_C1_MONT;
    return {fp_zero<Fq>(), fq_from_limbs(c1)};
}
inline G2Affine g2_identity() { return {Fq2::zero(), Fq2::zero(), true}; }
inline bool g2_is_on_curve(const G2Affine& p) {
    if (p.inf) return true;
    return p.y.square() == p.x.square() * p.x + g2_coeff_b();
}
inline G2Affine g2_add(const G2Affine& p, const G2Affine& q) {
    if (p.inf) return q;
    if (q.inf) return p;
    Fq2 lam;
    if (p.x == q.x) {
        if ((p.y + q.y).is_zero()) return g2_identity();
        Fq2 xx = p.x.square();
        lam = (xx + xx + xx) * (p.y + p.y).inverse();
    } else {
        lam = (q.y - p.y) * (q.x - p.x).inverse();
    }
    Fq2 x3 = lam.square() - p.x - q.x;
    Fq2 y3 = lam * (p.x - x3) - p.y;
    return {x3, y3, false};
}
// k given as little-endian 32-bit limbs
inline G2Affine g2_mul(const G2Affine& p, const uint32_t* k, int limbs) {
    G2Affine acc = g2_identity();
    for (int i = limbs * 32 - 1; i >= 0; i--) {
        acc = g2_add(acc, acc);
        if ((k[i >> 5] >> (i & 31)) & 1) acc = g2_add(acc, p);
    }
    return acc;
}

// ------------------------------------------------------------------------------------------------ pairing
// Ate Miller loop f_{x,Q}(P) with Q untwisted into E(Fq12): psi(x', y') = (x' w^2, y' w^3).  The running point T stays on the
// twist, in affine coordinates over Fq2 (one Fq2 inversion per step: a binary Euclid on the host, ff.cuh).  With the
// slope lambda' of the twist curve, the slope over Fq12 is lambda' w and the line through psi(T) evaluated at P = (xp, yp) is
//     l = yp - lambda' xp w + (lambda' x_T - y_T) w^3            (coefficients at 1, w and w^3 = v w only)
// — the same VALUE the plain Fq12 formulation of r01 produced, so the product is the same field element, not only the same
// pairing.  Several pairings share one accumulator: one Fq12 squaring per loop step whatever their number.
struct MillerPair {
    Fq xp, yp;
    Fq2 xq, yq, tx, ty;
};
inline Fq12 miller_line(const MillerPair& m, const Fq2& lam) {
    Fq12 l;
    l.c0 = {{m.yp, fp_zero<Fq>()}, Fq2::zero(), Fq2::zero()};
    l.c1 = {-(lam.mul_fq(m.xp)), lam * m.tx - m.ty, Fq2::zero()};
    return l;
}
inline Fq12 multi_miller_loop(const std::vector<std::pair<G1Affine, G2Affine>>& pairs) {
    std::vector<MillerPair> ms;
    for (auto& pq : pairs)
        if (!g1_is_inf(pq.first) && !pq.second.inf) ms.push_back({pq.first.x, pq.first.y, pq.second.x, pq.second.y, pq.second.x, pq.second.y});
    Fq12 f = Fq12::one();
    if (ms.empty()) return f;
    const uint64_t X = SWM_BLS_X;
    int top = 63;
    while (!((X >> top) & 1)) top--;
    for (int i = top - 1; i >= 0; i--) {
        f = f * f;
        for (auto& m : ms) {  // doubling step (T never has order two: it lies in the prime-order subgroup)
            Fq2 xx = m.tx.square();
            Fq2 lam = (xx + xx + xx) * (m.ty + m.ty).inverse();
            f = f * miller_line(m, lam);
            Fq2 nx = lam.square() - m.tx - m.tx;
            m.ty = lam * (m.tx - nx) - m.ty;
            m.tx = nx;
        }
        if ((X >> i) & 1)
            for (auto& m : ms) {  // addition step (T = k Q with 1 < k < r: never +-Q)
                Fq2 lam = (m.yq - m.ty) * (m.xq - m.tx).inverse();
                f = f * miller_line(m, lam);
                Fq2 nx = lam.square() - m.tx - m.xq;
                m.ty = lam * (m.tx - nx) - m.ty;
                m.tx = nx;
            }
    }
    return f;
}
inline Fq12 miller_loop(const G1Affine& P, const G2Affine& Q) { return multi_miller_loop({{P, Q}}); }

// q^2-Frobenius: Fq2 is fixed; v^i w^j -> gamma^(2 i + j) v^i w^j, gamma = 5^((q-1)/6) (constants_gen.h)
inline Fq12 frobenius2(const Fq12& a) {
    static const uint32_t gl[12] = SWM_FROB2_GAMMA_MONT;
    static const Fq g1 = fq_from_limbs(gl);
    static const Fq g2 = fp_mul(g1, g1), g3 = fp_mul(g2, g1), g4 = fp_mul(g3, g1), g5 = fp_mul(g4, g1);
    return {{a.c0.c0, a.c0.c1.mul_fq(g2), a.c0.c2.mul_fq(g4)}, {a.c1.c0.mul_fq(g1), a.c1.c1.mul_fq(g3), a.c1.c2.mul_fq(g5)}};
}
// Squaring in the cyclotomic subgroup (Granger-Scott; Guide to Pairing-Based Cryptography alg. 5.5.4, as ark-ff's
// Fp12::cyclotomic_square arranges it for this tower): three Fq4 squarings instead of a full Fq12 one.  Only valid for
// elements of the subgroup, i.e. after the easy part of the final exponentiation.
inline Fq12 cyclotomic_square(const Fq12& a) {
    const Fq2 &r0 = a.c0.c0, &r4 = a.c0.c1, &r3 = a.c0.c2, &r2 = a.c1.c0, &r1 = a.c1.c1, &r5 = a.c1.c2;
    auto sq4 = [](const Fq2& x, const Fq2& y, Fq2* t0, Fq2* t1) {  // (x + y s)^2 with s^2 = u: t0 = x^2 + u y^2, t1 = 2 x y
        Fq2 xy = x * y;
        *t0 = (x + y) * (y.mul_by_nonresidue() + x) - xy - xy.mul_by_nonresidue();
        *t1 = xy + xy;
    };
    Fq2 t0, t1, t2, t3, t4, t5;
    sq4(r0, r1, &t0, &t1);
    sq4(r2, r3, &t2, &t3);
    sq4(r4, r5, &t4, &t5);
    auto three_minus_two = [](const Fq2& t, const Fq2& z) { Fq2 d = t - z; return d + d + t; };   // 3 t - 2 z
    auto three_plus_two = [](const Fq2& t, const Fq2& z) { Fq2 d = t + z; return d + d + t; };    // 3 t + 2 z
    Fq2 z0 = three_minus_two(t0, r0), z1 = three_plus_two(t1, r1);
    Fq2 z2 = three_plus_two(t5.mul_by_nonresidue(), r2), z3 = three_minus_two(t4, r3);
    Fq2 z4 = three_minus_two(t2, r4), z5 = three_plus_two(t3, r5);
    return {{z0, z4, z3}, {z2, z1, z5}};
}
// q-Frobenius: the Fq2 coefficients are conjugated and v^i w^j is multiplied by zeta^(2 i + j), zeta = u^((q-1)/6) in Fq2
inline Fq12 frobenius1(const Fq12& a) {
    static const uint32_t z0[12] = SWM_FROB1_ZETA_C0_MONT, z1[12] = SWM_FROB1_ZETA_C1_MONT;
    static const Fq2 g1 = {fq_from_limbs(z0), fq_from_limbs(z1)};
    static const Fq2 g2 = g1 * g1, g3 = g2 * g1, g4 = g3 * g1, g5 = g4 * g1;
    auto cj = [](const Fq2& x) { return Fq2{x.c0, fp_neg(x.c1)}; };
    return {{cj(a.c0.c0), cj(a.c0.c1) * g2, cj(a.c0.c2) * g4}, {cj(a.c1.c0) * g1, cj(a.c1.c1) * g3, cj(a.c1.c2) * g5}};
}
// g^x for the curve parameter x (64 bits, seven of them set) in the cyclotomic subgroup
inline Fq12 cyclotomic_exp_x(const Fq12& g) {
    const uint64_t X = SWM_BLS_X;
    int top = 63;
    while (!((X >> top) & 1)) top--;
    Fq12 acc = g;
    for (int i = top - 1; i >= 0; i--) {
        acc = cyclotomic_square(acc);
        if ((X >> i) & 1) acc = acc * g;
    }
    return acc;
}
// The plain form of the hard part: (q^4 - q^2 + 1) / r (1270 bits) by a 4-bit window over cyclotomic squarings.  Kept as the
// reference the addition chain below is checked against (tests: swm_selftest_pairing).
inline Fq12 final_exponentiation_windowed(const Fq12& f) {
    static const uint32_t e[SWM_FINAL_EXP_HARD_LIMBS] = SWM_FINAL_EXP_HARD;
    Fq12 g = f.conjugate() * f.inverse();
    g = frobenius2(g) * g;
    Fq12 table[16];
    table[0] = Fq12::one();
    table[1] = g;
    for (int i = 2; i < 16; i++) table[i] = table[i - 1] * g;
    Fq12 acc = Fq12::one();
    bool started = false;
    for (int i = SWM_FINAL_EXP_HARD_LIMBS * 8 - 1; i >= 0; i--) {
        const unsigned nib = (e[i >> 3] >> ((i & 7) * 4)) & 15u;
        if (started)
            for (int k = 0; k < 4; k++) acc = cyclotomic_square(acc);
        if (nib) {
            acc = started ? acc * table[nib] : table[nib];
            started = true;
        }
    }
    return acc;
}
// f^(3 (q^12-1)/r): easy part f^((q^6-1)(q^2+1)) by a conjugation, an inversion and the q^2-Frobenius; hard part by the
// BLS12 addition chain  3 (q^4 - q^2 + 1) / r = (x - 1)^2 (x + q) (x^2 + q^2 - 1) + 3  (Hayashida, Hayasaka, Teruya 2020): five
// exponentiations by the 64-bit x over cyclotomic squarings, inverses by conjugation.  The CUBE of the reduced pairing:
// 3 does not divide r, so it is one exactly when the pairing is — and that is the only question the verifier asks.
inline Fq12 final_exponentiation(const Fq12& f) {
    Fq12 g = f.conjugate() * f.inverse();
    g = frobenius2(g) * g;                                                  // in the cyclotomic subgroup from here
    Fq12 t0 = cyclotomic_exp_x(g) * g.conjugate();                          // g^(x-1)
    Fq12 t1 = cyclotomic_exp_x(t0) * t0.conjugate();                        // g^((x-1)^2)
    Fq12 t2 = cyclotomic_exp_x(t1) * frobenius1(t1);                        // ^(x+q)
    Fq12 t3 = cyclotomic_exp_x(cyclotomic_exp_x(t2)) * frobenius2(t2) * t2.conjugate();   // ^(x^2+q^2-1)
    return t3 * cyclotomic_square(g) * g;                                   // * g^3
}
inline bool product_of_pairings_is_one(const std::vector<std::pair<G1Affine, G2Affine>>& pairs) {
    return final_exponentiation(multi_miller_loop(pairs)).is_one();
}

}  // namespace swm
