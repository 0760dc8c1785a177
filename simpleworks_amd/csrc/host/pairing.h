// pairing.h — BLS12-377 extension tower, G2 and the ate pairing on the HOST (CPU C++).
// Needed by verify_proof (/root/reference/src/marlin/mod.rs:79-86 -> MarlinKZG10::check_combinations ->
// KZG10::batch_check: one product of two pairings) and by universal setup (a random G2 element and beta*h).
// The reference delegates to ark-ec 0.3 bls12 (not vendored); this is a restatement with a deliberately plain
// structure — affine Miller loop with generic Fq12 line evaluation — because the verifier is milliseconds of host
// work, not a hot path.  Tower: Fq2 = Fq[u]/(u^2+5), Fq6 = Fq2[v]/(v^3-u), Fq12 = Fq6[w]/(w^2-v);
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
// Ate Miller loop f_{x,Q}(P) with Q untwisted into E(Fq12): (x', y') -> (x' w^2, y' w^3).
inline Fq12 miller_loop(const G1Affine& P, const G2Affine& Q) {
    if (g1_is_inf(P) || Q.inf) return Fq12::one();
    // w^2 = v -> (0,1,0) in c0 ; w^3 = v w -> (0,1,0) in c1
    Fq12 xq = {{Fq2::zero(), Q.x, Fq2::zero()}, Fq6::zero()};
    Fq12 yq = {Fq6::zero(), {Fq2::zero(), Q.y, Fq2::zero()}};
    Fq12 xp = Fq12::from_fq(P.x), yp = Fq12::from_fq(P.y);
    Fq12 f = Fq12::one(), tx = xq, ty = yq;
    const uint64_t X = SWM_BLS_X;
    int top = 63;
    while (!((X >> top) & 1)) top--;
    for (int i = top - 1; i >= 0; i--) {
        Fq12 txx = tx * tx;
        Fq12 lam = (txx + txx + txx) * (ty + ty).inverse();
        f = f * f * (yp - ty - lam * (xp - tx));
        Fq12 nx = lam * lam - tx - tx;
        ty = lam * (tx - nx) - ty;
        tx = nx;
        if ((X >> i) & 1) {
            lam = (yq - ty) * (xq - tx).inverse();
            f = f * (yp - ty - lam * (xp - tx));
            nx = lam * lam - tx - xq;
            ty = lam * (tx - nx) - ty;
            tx = nx;
        }
    }
    return f;
}
// f^((q^12-1)/r) = (conj(f) * f^-1)^((q^6+1)/r)
inline Fq12 final_exponentiation(const Fq12& f) {
    static const uint32_t e[SWM_FINAL_EXP2_LIMBS] = SWM_FINAL_EXP2;
    Fq12 g = f.conjugate() * f.inverse();
    return g.pow(e, SWM_FINAL_EXP2_LIMBS);
}
inline bool product_of_pairings_is_one(const std::vector<std::pair<G1Affine, G2Affine>>& pairs) {
    Fq12 f = Fq12::one();
    for (auto& pq : pairs) f = f * miller_loop(pq.first, pq.second);
    return final_exponentiation(f).is_one();
}

}  // namespace swm
