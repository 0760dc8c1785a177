// g1.cuh — BLS12-377 G1 (y^2 = x^3 + 1 over Fq) group law for device kernels and host logic.
//
// Replaces ark-ec 0.3 short_weierstrass_jacobian on the path reached from
// /root/reference/src/marlin/mod.rs:75 (prove -> KZG10::commit/open -> VariableBaseMSM) — SURVEY.md A.1/A.2.
// Accumulators use extended Jacobian (XYZZ) coordinates: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2 — a mixed
// addition is 8M + 2S with no field inversion and no special-casing of Z = 1.  Results are canonical
// group elements, so after affine normalisation they are bit-identical to arkworks' Jacobian arithmetic.
// Affine infinity is encoded as x = y = 0 (not on the curve since b = 1); XYZZ/Jacobian infinity has ZZ/Z = 0.
#pragma once
#include "ff.cuh"

namespace swm {

struct alignas(16) G1Affine {
    Fq x, y;
};
struct alignas(16) G1XYZZ {
    Fq x, y, zz, zzz;
};
struct alignas(16) G1Jac {
    Fq x, y, z;
};

SWM_HD bool g1_is_inf(const G1Affine& p) { return fp_is_zero(p.x) && fp_is_zero(p.y); }
SWM_HD bool g1_is_inf(const G1XYZZ& p) { return fp_is_zero(p.zz); }
SWM_HD bool g1_is_inf(const G1Jac& p) { return fp_is_zero(p.z); }

SWM_HD G1XYZZ g1_xyzz_identity() {
    G1XYZZ r;
    r.x = fp_one<Fq>();
    r.y = fp_one<Fq>();
    r.zz = fp_zero<Fq>();
    r.zzz = fp_zero<Fq>();
    return r;
}
SWM_HD G1Affine g1_affine_identity() {
    G1Affine r;
    r.x = fp_zero<Fq>();
    r.y = fp_zero<Fq>();
    return r;
}
SWM_HD G1XYZZ g1_from_affine(const G1Affine& p) {
    if (g1_is_inf(p)) return g1_xyzz_identity();
    G1XYZZ r;
    r.x = p.x;
    r.y = p.y;
    r.zz = fp_one<Fq>();
    r.zzz = fp_one<Fq>();
    return r;
}
SWM_HD G1Affine g1_neg(const G1Affine& p) {
    G1Affine r;
    r.x = p.x;
    r.y = fp_neg(p.y);
    return r;
}
SWM_HD G1XYZZ g1_neg(const G1XYZZ& p) {
    G1XYZZ r = p;
    r.y = fp_neg(p.y);
    return r;
}

// doubling of an affine point -> XYZZ  (EFD mdbl-2008-s-1, a = 0)
SWM_HD G1XYZZ g1_dbl_affine(const G1Affine& p) {
    if (g1_is_inf(p)) return g1_xyzz_identity();
    G1XYZZ r;
    Fq u = fp_dbl(p.y);
    Fq v = fp_sqr(u);
    Fq w = fp_mul(u, v);
    Fq s = fp_mul(p.x, v);
    Fq xx = fp_sqr(p.x);
    Fq m = fp_add(fp_dbl(xx), xx);
    r.x = fp_sub(fp_sub(fp_sqr(m), s), s);
    r.y = fp_sub(fp_mul(m, fp_sub(s, r.x)), fp_mul(w, p.y));
    r.zz = v;
    r.zzz = w;
    return r;
}

// XYZZ doubling (EFD dbl-2008-s-1, a = 0)
SWM_HD G1XYZZ g1_dbl(const G1XYZZ& p) {
    if (g1_is_inf(p)) return p;
    G1XYZZ r;
    Fq u = fp_dbl(p.y);
    Fq v = fp_sqr(u);
    Fq w = fp_mul(u, v);
    Fq s = fp_mul(p.x, v);
    Fq xx = fp_sqr(p.x);
    Fq m = fp_add(fp_dbl(xx), xx);
    r.x = fp_sub(fp_sub(fp_sqr(m), s), s);
    r.y = fp_sub(fp_mul(m, fp_sub(s, r.x)), fp_mul(w, p.y));
    r.zz = fp_mul(v, p.zz);
    r.zzz = fp_mul(w, p.zzz);
    return r;
}

// acc += q (q affine)  (EFD madd-2008-s) with the doubling / cancellation cases handled
SWM_HD void g1_add_mixed(G1XYZZ& acc, const G1Affine& q) {
    if (g1_is_inf(q)) return;
    if (g1_is_inf(acc)) {
        acc = g1_from_affine(q);
        return;
    }
    Fq u2 = fp_mul(q.x, acc.zz);
    Fq s2 = fp_mul(q.y, acc.zzz);
    Fq p = fp_sub(u2, acc.x);
    Fq r = fp_sub(s2, acc.y);
    if (fp_is_zero(p)) {
        if (fp_is_zero(r)) acc = g1_dbl_affine(q);
        else acc = g1_xyzz_identity();
        return;
    }
    Fq pp = fp_sqr(p);
    Fq ppp = fp_mul(p, pp);
    Fq qq = fp_mul(acc.x, pp);
    Fq x3 = fp_sub(fp_sub(fp_sub(fp_sqr(r), ppp), qq), qq);
    Fq y3 = fp_sub(fp_mul(r, fp_sub(qq, x3)), fp_mul(acc.y, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fp_mul(acc.zz, pp);
    acc.zzz = fp_mul(acc.zzz, ppp);
}

// acc += q (both XYZZ)  (EFD add-2008-s)
SWM_HD void g1_add(G1XYZZ& acc, const G1XYZZ& q) {
    if (g1_is_inf(q)) return;
    if (g1_is_inf(acc)) {
        acc = q;
        return;
    }
    Fq u1 = fp_mul(acc.x, q.zz);
    Fq u2 = fp_mul(q.x, acc.zz);
    Fq s1 = fp_mul(acc.y, q.zzz);
    Fq s2 = fp_mul(q.y, acc.zzz);
    Fq p = fp_sub(u2, u1);
    Fq r = fp_sub(s2, s1);
    if (fp_is_zero(p)) {
        if (fp_is_zero(r)) acc = g1_dbl(acc);
        else acc = g1_xyzz_identity();
        return;
    }
    Fq pp = fp_sqr(p);
    Fq ppp = fp_mul(p, pp);
    Fq qq = fp_mul(u1, pp);
    Fq x3 = fp_sub(fp_sub(fp_sub(fp_sqr(r), ppp), qq), qq);
    Fq y3 = fp_sub(fp_mul(r, fp_sub(qq, x3)), fp_mul(s1, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fp_mul(fp_mul(acc.zz, q.zz), pp);
    acc.zzz = fp_mul(fp_mul(acc.zzz, q.zzz), ppp);
}

// XYZZ -> Jacobian (X*ZZ, Y*ZZZ, ZZ): x = X'/Z'^2 = X/ZZ, y = Y'/Z'^3 = Y*ZZZ/ZZ^3 = Y/ZZZ
SWM_HD G1Jac g1_to_jacobian(const G1XYZZ& p) {
    G1Jac r;
    if (g1_is_inf(p)) {
        r.x = fp_one<Fq>();
        r.y = fp_one<Fq>();
        r.z = fp_zero<Fq>();
        return r;
    }
    r.x = fp_mul(p.x, p.zz);
    r.y = fp_mul(p.y, p.zzz);
    r.z = p.zz;
    return r;
}
SWM_HD G1XYZZ g1_from_jacobian(const G1Jac& p) {
    if (g1_is_inf(p)) return g1_xyzz_identity();
    G1XYZZ r;
    r.x = p.x;
    r.y = p.y;
    r.zz = fp_sqr(p.z);
    r.zzz = fp_mul(r.zz, p.z);
    return r;
}
// XYZZ -> affine (one field inversion)
SWM_HD G1Affine g1_to_affine(const G1XYZZ& p) {
    if (g1_is_inf(p)) return g1_affine_identity();
    Fq zi = fp_inv(fp_mul(p.zz, p.zzz));  // 1/(ZZ*ZZZ)
    Fq zz_inv = fp_mul(zi, p.zzz);
    Fq zzz_inv = fp_mul(zi, p.zz);
    G1Affine r;
    r.x = fp_mul(p.x, zz_inv);
    r.y = fp_mul(p.y, zzz_inv);
    return r;
}
SWM_HD bool g1_is_on_curve(const G1Affine& p) {
    if (g1_is_inf(p)) return true;
    Fq lhs = fp_sqr(p.y);
    Fq rhs = fp_add(fp_mul(fp_sqr(p.x), p.x), fp_one<Fq>());
    return fp_eq(lhs, rhs);
}

// ------------------------------------------------------------------------------------------------ twisted Edwards form
// G1 is also the a = -1 twisted Edwards curve  -x^2 + y^2 = 1 + d x^2 y^2  (constants and the map: tools/gen_constants.py):
//     x = f (x_w + 1) / y_w,   y = (s (x_w + 1) - 1) / (s (x_w + 1) + 1),   s = 1/sqrt(3), f = sqrt(-(A + 2)/B).
// In extended coordinates (X : Y : T : Z), T Z = X Y, the UNIFIED addition of Hisil-Wong-Carter-Dawson (EFD
// add-2008-hwcd-3) costs 9 multiplications (8 + the constant 2d) and — with the second operand kept as the affine
// triple (y - x, y + x, 2 d x y) — 7: against 8M + 2S for the XYZZ mixed addition.  The law has no exceptional cases
// among points of odd order (doubling, cancellation and the identity (0, 1) go through the same formulas), so it is used
// only for base sets known to lie in the prime-order subgroup (msm_table_build_te); it is what the precomputed-table
// MSM accumulates and reduces in.  A TE point travels in a G1XYZZ-sized slot: x -> X, y -> Y, zz -> T, zzz -> Z.
// One table row: affine, already in the form the mixed addition consumes — (y - x, y + x, 2 d x y), each coordinate times
// 2^8 (Montgomery radix 2^392) as the FOURTEEN 28-BIT LIMBS the accumulation multiplies (fq28.cuh), in a 64-byte sector of
// its own (limbs 14 and 15 are padding).  r04; r03 kept the coordinates packed (3 x 48 B): every row then cost 84
// shift / mask instructions to unpack and straddled 64-byte sectors (144-B rows: 25 % over-fetch), and the sign of a digit
// cost two 14-limb selects where it now costs the choice of the ADDRESS the first two coordinates are loaded from.
// 192 B per row instead of 144 B: 13 x 3 M rows = 7.5 GB per 2^20 key (5.6 GB before) of the 288 GB.
struct alignas(64) G1TE {
    uint32_t ymx[16], ypx[16], kt[16];
};
struct TeParams {
    static constexpr uint32_t S[12] = SWM_TE_S_MONT;
    static constexpr uint32_t F[12] = SWM_TE_F_MONT;
    static constexpr uint32_t K2D[12] = SWM_TE_2D_MONT;
    static constexpr uint32_t RT3[12] = SWM_TE_RT3_MONT;
    static constexpr uint32_t F_RT3[12] = SWM_TE_F_RT3_MONT;
};
SWM_HD Fq fq_const(const uint32_t (&c)[12]) {
    Fq r;
    for (int i = 0; i < 12; i++) r.v[i] = c[i];
    return r;
}
SWM_HD G1XYZZ g1te_identity() {
    G1XYZZ r;
    r.x = fp_zero<Fq>();
    r.y = fp_one<Fq>();
    r.zz = fp_zero<Fq>();
    r.zzz = fp_one<Fq>();
    return r;
}
// acc += q, both extended (unified: also doubles)
SWM_HD void g1te_add(G1XYZZ& acc, const G1XYZZ& q) {
    Fq a = fp_mul(fp_sub(acc.y, acc.x), fp_sub(q.y, q.x));
    Fq b = fp_mul(fp_add(acc.y, acc.x), fp_add(q.y, q.x));
    Fq c = fp_mul(fp_mul(acc.zz, q.zz), fq_const(TeParams::K2D));
    Fq d = fp_dbl(fp_mul(acc.zzz, q.zzz));
    Fq e = fp_sub(b, a), f = fp_sub(d, c), g = fp_add(d, c), h = fp_add(b, a);
    acc.x = fp_mul(e, f);
    acc.y = fp_mul(g, h);
    acc.zz = fp_mul(e, h);
    acc.zzz = fp_mul(f, g);
}
SWM_HD G1XYZZ g1te_dbl(const G1XYZZ& p) {
    G1XYZZ r = p;
    g1te_add(r, p);
    return r;
}
// extended twisted Edwards -> XYZZ on y^2 = x^3 + 1 (no inversion):
//   u = (Z + Y)/(Z - Y),  x_w = u/s - 1,  y_w = f u Z / (s X);  with z = (Z - Y) X:  zz = z^2, zzz = z^3,
//   X_w = (rt3 (Z + Y) - (Z - Y)) (Z - Y) X^2,   Y_w = (f rt3) (Z + Y) Z zz.        The identity (X = 0) gives zz = 0.
SWM_HD G1XYZZ g1te_to_xyzz(const G1XYZZ& p) {
    Fq n1 = fp_add(p.zzz, p.y), a = fp_sub(p.zzz, p.y);
    Fq z = fp_mul(a, p.x);
    if (fp_is_zero(z)) return g1_xyzz_identity();
    G1XYZZ r;
    r.zz = fp_sqr(z);
    r.zzz = fp_mul(r.zz, z);
    r.x = fp_mul(fp_sub(fp_mul(fq_const(TeParams::RT3), n1), a), fp_mul(z, p.x));
    r.y = fp_mul(fp_mul(fq_const(TeParams::F_RT3), fp_mul(n1, p.zzz)), r.zz);
    return r;
}

// k * p for a small multiplier (window offsets in the bucket reduction; k < 2^31)
SWM_HD G1XYZZ g1_mul_small(const G1XYZZ& p, uint32_t k) {
    G1XYZZ acc = g1_xyzz_identity();
    if (k == 0) return acc;
    int top = 31;
    while (!((k >> top) & 1)) top--;
    for (int i = top; i >= 0; i--) {
        acc = g1_dbl(acc);
        if ((k >> i) & 1) g1_add(acc, p);
    }
    return acc;
}

}  // namespace swm
