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
