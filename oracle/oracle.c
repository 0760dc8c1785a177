/* oracle.c — see oracle.h.  TEST INFRASTRUCTURE ONLY (checker + CPU baseline), never shipped.
 *
 * Plain C restatement of the arkworks 0.3.0 algorithms that simpleworks' src/marlin/mod.rs:52,75,85,92
 * reach (sources not vendored in /root/reference; restated from SURVEY.md Appendix A, tagged [U] there).
 * 64-bit limbs + unsigned __int128 CIOS Montgomery multiplication, deliberately a different limb
 * width and code shape from the product's 32-bit-limb HIP kernels so that the two are independent.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ constants (SURVEY Appendix B [V]) */
static const uint64_t FR_P[4] = {0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL,
                                 0x12ab655e9a2ca556ULL};
static const uint64_t FR_R1[4] = {0x7d1c7ffffffffff3ULL, 0x7257f50f6ffffff2ULL, 0x16d81575512c0feeULL,
                                  0x0d4bda322bbb9a9dULL};
static const uint64_t FR_R2[4] = {0x25d577bab861857bULL, 0xcc2c27b58860591fULL, 0xa7cc008fe5dc8593ULL,
                                  0x011fdae7eff1c939ULL};
static const uint64_t FR_INV = 0x0a117fffffffffffULL;
/* 2^47-th root of unity, Montgomery form (TWO_ADIC_ROOT_OF_UNITY of ark-bls12-377 Fr) */
static const uint64_t FR_ROOT47[4] = {0xaf80da4dda3ad648ULL, 0x5e223adbfc381dacULL, 0x03ba0666b2f92525ULL,
                                      0x0f906c5b3befb0ceULL};

static const uint64_t FQ_P[6] = {0x8508c00000000001ULL, 0x170b5d4430000000ULL, 0x1ef3622fba094800ULL,
                                 0x1a22d9f300f5138fULL, 0xc63b05c06ca1493bULL, 0x01ae3a4617c510eaULL};
static const uint64_t FQ_R1[6] = {0x02cdffffffffff68ULL, 0x51409f837fffffb1ULL, 0x9f7db3a98a7d3ff2ULL,
                                  0x7b4e97b76e7c6305ULL, 0x4cf495bf803c84e8ULL, 0x008d6661e2fdf49aULL};
static const uint64_t FQ_R2[6] = {0xb786686c9400cd22ULL, 0x0329fcaab00431b1ULL, 0x22a5f11162d6b46dULL,
                                  0xbfdf7d03827dc3acULL, 0x837e92f041790bf9ULL, 0x006dfccb1e914b88ULL};
static const uint64_t FQ_INV = 0x8508bfffffffffffULL;

/* ------------------------------------------------------------------ generic N-limb Montgomery helpers */
static inline int limbs_geq(const uint64_t *a, const uint64_t *b, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline int limbs_is_zero(const uint64_t *a, int n) {
    uint64_t acc = 0;
    for (int i = 0; i < n; i++) acc |= a[i];
    return acc == 0;
}
static inline int limbs_eq(const uint64_t *a, const uint64_t *b, int n) {
    uint64_t acc = 0;
    for (int i = 0; i < n; i++) acc |= a[i] ^ b[i];
    return acc == 0;
}
static inline uint64_t limbs_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
    uint64_t borrow = 0;
    for (int i = 0; i < n; i++) {
        u128 t = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1;
    }
    return borrow;
}
static inline uint64_t limbs_add(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
    uint64_t carry = 0;
    for (int i = 0; i < n; i++) {
        u128 t = (u128)a[i] + b[i] + carry;
        r[i] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
    }
    return carry;
}

#define DEFINE_FIELD(NAME, N, P, INV)                                                                       \
    static inline void NAME##_add(uint64_t *r, const uint64_t *a, const uint64_t *b) {                      \
        uint64_t t[N];                                                                                      \
        limbs_add(t, a, b, N); /* both < p < 2^(64N-1): no carry out */                                     \
        if (limbs_geq(t, P, N)) limbs_sub(r, t, P, N);                                                      \
        else memcpy(r, t, sizeof(t));                                                                       \
    }                                                                                                       \
    static inline void NAME##_sub(uint64_t *r, const uint64_t *a, const uint64_t *b) {                      \
        uint64_t t[N];                                                                                      \
        if (limbs_sub(t, a, b, N)) limbs_add(r, t, P, N);                                                   \
        else memcpy(r, t, sizeof(t));                                                                       \
    }                                                                                                       \
    static inline void NAME##_neg(uint64_t *r, const uint64_t *a) {                                         \
        if (limbs_is_zero(a, N)) memset(r, 0, 8 * N);                                                       \
        else limbs_sub(r, P, a, N);                                                                         \
    }                                                                                                       \
    static inline void NAME##_dbl(uint64_t *r, const uint64_t *a) { NAME##_add(r, a, a); }                  \
    /* CIOS Montgomery product a*b*R^-1 mod p */                                                            \
    static inline void NAME##_mul(uint64_t *r, const uint64_t *a, const uint64_t *b) {                      \
        uint64_t t[N + 2];                                                                                  \
        memset(t, 0, sizeof(t));                                                                            \
        for (int i = 0; i < N; i++) {                                                                       \
            uint64_t c = 0;                                                                                 \
            for (int j = 0; j < N; j++) {                                                                   \
                u128 x = (u128)a[j] * b[i] + t[j] + c;                                                      \
                t[j] = (uint64_t)x;                                                                         \
                c = (uint64_t)(x >> 64);                                                                    \
            }                                                                                               \
            u128 x = (u128)t[N] + c;                                                                        \
            t[N] = (uint64_t)x;                                                                             \
            t[N + 1] = (uint64_t)(x >> 64);                                                                 \
            uint64_t m = t[0] * INV;                                                                        \
            x = (u128)m * P[0] + t[0];                                                                      \
            c = (uint64_t)(x >> 64);                                                                        \
            for (int j = 1; j < N; j++) {                                                                   \
                x = (u128)m * P[j] + t[j] + c;                                                              \
                t[j - 1] = (uint64_t)x;                                                                     \
                c = (uint64_t)(x >> 64);                                                                    \
            }                                                                                               \
            x = (u128)t[N] + c;                                                                             \
            t[N - 1] = (uint64_t)x;                                                                         \
            t[N] = t[N + 1] + (uint64_t)(x >> 64);                                                          \
        }                                                                                                   \
        if (t[N] || limbs_geq(t, P, N)) limbs_sub(r, t, P, N);                                              \
        else memcpy(r, t, 8 * N);                                                                           \
    }                                                                                                       \
    static inline void NAME##_sqr(uint64_t *r, const uint64_t *a) { NAME##_mul(r, a, a); }

DEFINE_FIELD(fr, 4, FR_P, FR_INV)
DEFINE_FIELD(fq, 6, FQ_P, FQ_INV)

static void fr_pow(uint64_t *r, const uint64_t *a, const uint64_t *e, int elimbs) {
    uint64_t acc[4], base[4];
    memcpy(acc, FR_R1, 32);
    memcpy(base, a, 32);
    for (int i = 0; i < elimbs * 64; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) fr_mul(acc, acc, base);
        fr_sqr(base, base);
    }
    memcpy(r, acc, 32);
}
static void fr_inverse(uint64_t *r, const uint64_t *a) { /* Fermat: a^(p-2); 0 -> 0 */
    uint64_t e[4];
    uint64_t two[4] = {2, 0, 0, 0};
    limbs_sub(e, FR_P, two, 4);
    fr_pow(r, a, e, 4);
}
static void fq_inverse(uint64_t *r, const uint64_t *a) {
    uint64_t e[6], two[6] = {2, 0, 0, 0, 0, 0};
    limbs_sub(e, FQ_P, two, 6);
    uint64_t acc[6], base[6];
    memcpy(acc, FQ_R1, 48);
    memcpy(base, a, 48);
    for (int i = 0; i < 384; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) fq_mul(acc, acc, base);
        fq_sqr(base, base);
    }
    memcpy(r, acc, 48);
}

/* ------------------------------------------------------------------ exported field helpers */
void oracle_fr_to_mont(const uint64_t *s, uint64_t *m, size_t n) {
    for (size_t i = 0; i < n; i++) fr_mul(m + 4 * i, s + 4 * i, FR_R2);
}
void oracle_fr_from_mont(const uint64_t *m, uint64_t *s, size_t n) {
    static const uint64_t one[4] = {1, 0, 0, 0};
    for (size_t i = 0; i < n; i++) fr_mul(s + 4 * i, m + 4 * i, one);
}
void oracle_fq_to_mont(const uint64_t *s, uint64_t *m, size_t n) {
    for (size_t i = 0; i < n; i++) fq_mul(m + 6 * i, s + 6 * i, FQ_R2);
}
void oracle_fq_from_mont(const uint64_t *m, uint64_t *s, size_t n) {
    static const uint64_t one[6] = {1, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) fq_mul(s + 6 * i, m + 6 * i, one);
}
void oracle_fr_mul(const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) fr_mul(o + 4 * i, a + 4 * i, b + 4 * i);
}
void oracle_fr_add(const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) fr_add(o + 4 * i, a + 4 * i, b + 4 * i);
}
void oracle_fr_sub(const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) fr_sub(o + 4 * i, a + 4 * i, b + 4 * i);
}
void oracle_fq_mul(const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) fq_mul(o + 6 * i, a + 6 * i, b + 6 * i);
}
void oracle_fr_inv(const uint64_t *a, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; i++) fr_inverse(o + 4 * i, a + 4 * i);
}

/* ark_ff::batch_inversion (fields/mod.rs 0.3.0): prefix products skipping zeros, one inversion,
 * backward sweep; zero entries are left untouched. */
void oracle_batch_inverse_fr(uint64_t *v, size_t n) {
    uint64_t *prod = (uint64_t *)malloc(32 * (n ? n : 1));
    uint64_t tmp[4];
    memcpy(tmp, FR_R1, 32);
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        if (limbs_is_zero(v + 4 * i, 4)) continue;
        fr_mul(tmp, tmp, v + 4 * i);
        memcpy(prod + 4 * k, tmp, 32);
        k++;
    }
    fr_inverse(tmp, tmp);
    for (size_t i = n; i-- > 0;) {
        if (limbs_is_zero(v + 4 * i, 4)) continue;
        k--;
        uint64_t newtmp[4], out[4];
        fr_mul(newtmp, tmp, v + 4 * i);
        if (k > 0) fr_mul(out, tmp, prod + 4 * (k - 1));
        else memcpy(out, tmp, 32);
        memcpy(v + 4 * i, out, 32);
        memcpy(tmp, newtmp, 32);
    }
    free(prod);
}

/* ------------------------------------------------------------------ G1 Jacobian (ark-ec 0.3 short_weierstrass_jacobian) */
typedef struct { uint64_t x[6], y[6], z[6]; } jac_t;
typedef struct { uint64_t x[6], y[6]; } aff_t;

static inline void jac_set_inf(jac_t *p) {
    memset(p, 0, sizeof(*p));
    memcpy(p->x, FQ_R1, 48);
    memcpy(p->y, FQ_R1, 48);
}
static inline int jac_is_inf(const jac_t *p) { return limbs_is_zero(p->z, 6); }
static inline int aff_is_inf(const aff_t *p) { return limbs_is_zero(p->x, 6) && limbs_is_zero(p->y, 6); }

/* dbl-2009-l (a = 0) */
static void jac_double(jac_t *r, const jac_t *p) {
    if (jac_is_inf(p)) { *r = *p; return; }
    uint64_t a[6], b[6], c[6], d[6], e[6], f[6], t[6];
    fq_sqr(a, p->x);
    fq_sqr(b, p->y);
    fq_sqr(c, b);
    fq_add(t, p->x, b);
    fq_sqr(t, t);
    fq_sub(t, t, a);
    fq_sub(t, t, c);
    fq_dbl(d, t);
    fq_dbl(e, a);
    fq_add(e, e, a);
    fq_sqr(f, e);
    uint64_t z3[6];
    fq_mul(z3, p->y, p->z);
    fq_dbl(z3, z3);
    fq_sub(t, f, d);
    fq_sub(r->x, t, d);
    fq_sub(t, d, r->x);
    fq_mul(t, e, t);
    fq_dbl(c, c);
    fq_dbl(c, c);
    fq_dbl(c, c);
    fq_sub(r->y, t, c);
    memcpy(r->z, z3, 48);
}

/* madd-2007-bl with doubling fallback (GroupProjective::add_assign_mixed) */
static void jac_add_mixed(jac_t *r, const jac_t *p, const aff_t *q) {
    if (aff_is_inf(q)) { *r = *p; return; }
    if (jac_is_inf(p)) {
        memcpy(r->x, q->x, 48);
        memcpy(r->y, q->y, 48);
        memcpy(r->z, FQ_R1, 48);
        return;
    }
    uint64_t z1z1[6], u2[6], s2[6];
    fq_sqr(z1z1, p->z);
    fq_mul(u2, q->x, z1z1);
    fq_mul(s2, q->y, p->z);
    fq_mul(s2, s2, z1z1);
    if (limbs_eq(p->x, u2, 6) && limbs_eq(p->y, s2, 6)) { jac_double(r, p); return; }
    uint64_t h[6], hh[6], i[6], j[6], rr[6], v[6], t[6], x3[6], y3[6], z3[6];
    fq_sub(h, u2, p->x);
    fq_sqr(hh, h);
    fq_dbl(i, hh);
    fq_dbl(i, i);
    fq_mul(j, h, i);
    fq_sub(rr, s2, p->y);
    fq_dbl(rr, rr);
    fq_mul(v, p->x, i);
    fq_sqr(x3, rr);
    fq_sub(x3, x3, j);
    fq_sub(x3, x3, v);
    fq_sub(x3, x3, v);
    fq_mul(j, p->y, j);
    fq_dbl(j, j);
    fq_sub(t, v, x3);
    fq_mul(y3, rr, t);
    fq_sub(y3, y3, j);
    fq_add(z3, p->z, h);
    fq_sqr(z3, z3);
    fq_sub(z3, z3, z1z1);
    fq_sub(z3, z3, hh);
    memcpy(r->x, x3, 48);
    memcpy(r->y, y3, 48);
    memcpy(r->z, z3, 48);
}

/* add-2007-bl with doubling fallback (GroupProjective::add_assign) */
static void jac_add(jac_t *r, const jac_t *p, const jac_t *q) {
    if (jac_is_inf(p)) { *r = *q; return; }
    if (jac_is_inf(q)) { *r = *p; return; }
    uint64_t z1z1[6], z2z2[6], u1[6], u2[6], s1[6], s2[6];
    fq_sqr(z1z1, p->z);
    fq_sqr(z2z2, q->z);
    fq_mul(u1, p->x, z2z2);
    fq_mul(u2, q->x, z1z1);
    fq_mul(s1, p->y, q->z);
    fq_mul(s1, s1, z2z2);
    fq_mul(s2, q->y, p->z);
    fq_mul(s2, s2, z1z1);
    if (limbs_eq(u1, u2, 6) && limbs_eq(s1, s2, 6)) { jac_double(r, p); return; }
    uint64_t h[6], i[6], j[6], rr[6], v[6], t[6], x3[6], y3[6], z3[6];
    fq_sub(h, u2, u1);
    fq_dbl(i, h);
    fq_sqr(i, i);
    fq_mul(j, h, i);
    fq_sub(rr, s2, s1);
    fq_dbl(rr, rr);
    fq_mul(v, u1, i);
    fq_sqr(x3, rr);
    fq_sub(x3, x3, j);
    fq_sub(x3, x3, v);
    fq_sub(x3, x3, v);
    fq_mul(s1, s1, j);
    fq_dbl(s1, s1);
    fq_sub(t, v, x3);
    fq_mul(y3, rr, t);
    fq_sub(y3, y3, s1);
    fq_add(z3, p->z, q->z);
    fq_sqr(z3, z3);
    fq_sub(z3, z3, z1z1);
    fq_sub(z3, z3, z2z2);
    fq_mul(z3, z3, h);
    memcpy(r->x, x3, 48);
    memcpy(r->y, y3, 48);
    memcpy(r->z, z3, 48);
}

static int jac_to_affine(aff_t *r, const jac_t *p) {
    if (jac_is_inf(p)) { memset(r, 0, sizeof(*r)); return 1; }
    uint64_t zi[6], zi2[6];
    fq_inverse(zi, p->z);
    fq_sqr(zi2, zi);
    fq_mul(r->x, p->x, zi2);
    fq_mul(zi2, zi2, zi);
    fq_mul(r->y, p->y, zi2);
    return 0;
}

void oracle_g1_add_mixed(const uint64_t *j, const uint64_t *a, uint64_t *o) {
    jac_t r;
    jac_add_mixed(&r, (const jac_t *)j, (const aff_t *)a);
    memcpy(o, &r, sizeof(r));
}
void oracle_g1_double(const uint64_t *j, uint64_t *o) {
    jac_t r;
    jac_double(&r, (const jac_t *)j);
    memcpy(o, &r, sizeof(r));
}
void oracle_g1_add(const uint64_t *a, const uint64_t *b, uint64_t *o) {
    jac_t r;
    jac_add(&r, (const jac_t *)a, (const jac_t *)b);
    memcpy(o, &r, sizeof(r));
}
int oracle_g1_to_affine(const uint64_t *j, uint64_t *a) { return jac_to_affine((aff_t *)a, (const jac_t *)j); }
int oracle_g1_is_on_curve(const uint64_t *a) {
    const aff_t *p = (const aff_t *)a;
    if (aff_is_inf(p)) return 1;
    uint64_t l[6], r[6];
    fq_sqr(l, p->y);
    fq_sqr(r, p->x);
    fq_mul(r, r, p->x);
    fq_add(r, r, FQ_R1); /* b = 1 */
    return limbs_eq(l, r, 6);
}

/* Batch Jacobian -> affine with one inversion per block (used by the fixed-base generator). */
static void batch_to_affine(const jac_t *in, aff_t *out, size_t n) {
    uint64_t(*pref)[6] = malloc(48 * (n ? n : 1));
    uint64_t acc[6];
    memcpy(acc, FQ_R1, 48);
    for (size_t i = 0; i < n; i++) {
        if (!jac_is_inf(&in[i])) fq_mul(acc, acc, in[i].z);
        memcpy(pref[i], acc, 48);
    }
    uint64_t inv[6];
    fq_inverse(inv, acc);
    for (size_t i = n; i-- > 0;) {
        if (jac_is_inf(&in[i])) { memset(&out[i], 0, sizeof(aff_t)); continue; }
        uint64_t zi[6], zi2[6];
        if (i > 0) fq_mul(zi, inv, pref[i - 1]);
        else memcpy(zi, inv, 48);
        fq_mul(inv, inv, in[i].z);
        fq_sqr(zi2, zi);
        fq_mul(out[i].x, in[i].x, zi2);
        fq_mul(zi2, zi2, zi);
        fq_mul(out[i].y, in[i].y, zi2);
    }
    free(pref);
}

void oracle_g1_fixed_base_mul(const uint64_t *base12, const uint64_t *scalars4, size_t n, uint64_t *out12,
                              int threads) {
    /* 8-bit windows: table[k][d] = d * 2^(8k) * B for d in 1..255 (affine) */
    enum { WIN = 8, NW = 32, TS = 255 };
    aff_t *table = malloc(sizeof(aff_t) * NW * TS);
    jac_t *tj = malloc(sizeof(jac_t) * NW * TS);
    jac_t cur;
    memcpy(cur.x, base12, 48);
    memcpy(cur.y, base12 + 6, 48);
    memcpy(cur.z, FQ_R1, 48);
    for (int k = 0; k < NW; k++) {
        aff_t curaff;
        jac_to_affine(&curaff, &cur);
        jac_t acc;
        jac_set_inf(&acc);
        for (int d = 0; d < TS; d++) {
            jac_add_mixed(&acc, &acc, &curaff);
            tj[k * TS + d] = acc;
        }
        for (int s = 0; s < WIN; s++) jac_double(&cur, &cur);
    }
    batch_to_affine(tj, table, (size_t)NW * TS);
    free(tj);
    const size_t BLK = 1024;
    size_t nblk = (n + BLK - 1) / BLK;
    (void)threads;
#pragma omp parallel for schedule(dynamic) num_threads(threads > 0 ? threads : 1)
    for (size_t b = 0; b < nblk; b++) {
        size_t lo = b * BLK, hi = lo + BLK < n ? lo + BLK : n;
        jac_t *tmp = malloc(sizeof(jac_t) * (hi - lo));
        for (size_t i = lo; i < hi; i++) {
            jac_t acc;
            jac_set_inf(&acc);
            const uint8_t *sb = (const uint8_t *)(scalars4 + 4 * i);
            for (int k = 0; k < NW; k++) {
                unsigned d = sb[k];
                if (d) jac_add_mixed(&acc, &acc, &table[k * TS + d - 1]);
            }
            tmp[i - lo] = acc;
        }
        batch_to_affine(tmp, (aff_t *)(out12 + 12 * lo), hi - lo);
        free(tmp);
    }
    free(table);
}

/* ------------------------------------------------------------------ K1: VariableBaseMSM::multi_scalar_mul (ark-ec 0.3.0) */
unsigned oracle_msm_window(size_t n) {
    if (n < 32) return 3;
    unsigned lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    return lg * 69 / 100 + 2;
}

static inline unsigned scalar_window(const uint64_t *s, unsigned w_start, unsigned c) {
    /* (scalar >> w_start) mod 2^c  — ark: scalar.divn(w_start); scalar.as_ref()[0] % (1 << c) */
    unsigned limb = w_start / 64, off = w_start % 64;
    uint64_t v = s[limb] >> off;
    if (off && limb + 1 < 4) v |= s[limb + 1] << (64 - off);
    return (unsigned)(v & (((uint64_t)1 << c) - 1));
}

void oracle_msm_g1(const uint64_t *bases12, const uint64_t *scalars4, size_t n, uint64_t *out18, int threads) {
    const aff_t *bases = (const aff_t *)bases12;
    unsigned c = oracle_msm_window(n);
    const unsigned num_bits = 253;
    unsigned nwin = (num_bits + c - 1) / c;
    jac_t *window_sums = malloc(sizeof(jac_t) * nwin);
    static const uint64_t one[4] = {1, 0, 0, 0};
    (void)threads;
#pragma omp parallel for schedule(dynamic) num_threads(threads > 0 ? threads : 1)
    for (unsigned wi = 0; wi < nwin; wi++) {
        unsigned w_start = wi * c;
        jac_t res;
        jac_set_inf(&res);
        size_t nb = ((size_t)1 << c) - 1;
        jac_t *buckets = malloc(sizeof(jac_t) * nb);
        for (size_t b = 0; b < nb; b++) jac_set_inf(&buckets[b]);
        for (size_t i = 0; i < n; i++) {
            const uint64_t *s = scalars4 + 4 * i;
            if (limbs_is_zero(s, 4)) continue; /* zero scalars are filtered out up front */
            if (limbs_eq(s, one, 4)) {
                if (w_start == 0) jac_add_mixed(&res, &res, &bases[i]); /* unit scalars only in window 0 */
            } else {
                unsigned d = scalar_window(s, w_start, c);
                if (d) jac_add_mixed(&buckets[d - 1], &buckets[d - 1], &bases[i]);
            }
        }
        /* running-sum trick, highest bucket first */
        jac_t running;
        jac_set_inf(&running);
        for (size_t b = nb; b-- > 0;) {
            jac_add(&running, &running, &buckets[b]);
            jac_add(&res, &res, &running);
        }
        window_sums[wi] = res;
        free(buckets);
    }
    /* lowest + fold(high -> low){ total += ws; total doubled c times } */
    jac_t total;
    jac_set_inf(&total);
    for (unsigned wi = nwin; wi-- > 1;) {
        jac_add(&total, &total, &window_sums[wi]);
        for (unsigned k = 0; k < c; k++) jac_double(&total, &total);
    }
    jac_add(&total, &total, &window_sums[0]);
    memcpy(out18, &total, sizeof(total));
    free(window_sums);
}

/* ------------------------------------------------------------------ K2: Radix2EvaluationDomain (ark-poly 0.3.0) */
static void fr_root_of_unity(uint64_t *w, unsigned log_n) {
    memcpy(w, FR_ROOT47, 32);
    for (unsigned i = log_n; i < 47; i++) fr_sqr(w, w);
}
static inline size_t bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

/* out[i] = scale * base^i for i in [0, n): every thread starts its chunk from base^start (square-and-multiply) and
 * continues with one multiplication per element, like ark-poly's compute_powers under the `parallel` feature. */
static void fr_pow_u64(uint64_t *o, const uint64_t *b, uint64_t e) {
    uint64_t acc[4], x[4];
    memcpy(acc, FR_R1, 32);
    memcpy(x, b, 32);
    while (e) {
        if (e & 1) fr_mul(acc, acc, x);
        fr_sqr(x, x);
        e >>= 1;
    }
    memcpy(o, acc, 32);
}
static void fr_scale_by_powers(uint64_t *a, size_t n, const uint64_t *scale, const uint64_t *base, int nt) {
    size_t chunk = (n + (size_t)nt - 1) / (size_t)nt;
    if (chunk < 1024) chunk = 1024;
#pragma omp parallel for schedule(static) num_threads(nt) if (n >= 4096)
    for (size_t lo = 0; lo < n; lo += chunk) {
        size_t hi = lo + chunk < n ? lo + chunk : n;
        uint64_t p[4];
        fr_pow_u64(p, base, lo);
        fr_mul(p, p, scale);
        for (size_t i = lo; i < hi; i++) {
            fr_mul(a + 4 * i, a + 4 * i, p);
            fr_mul(p, p, base);
        }
    }
}

void oracle_ntt_fr(uint64_t *a, unsigned log_n, int inverse, int coset, int threads) {
    size_t n = (size_t)1 << log_n;
    uint64_t w[4];
    fr_root_of_unity(w, log_n);
    uint64_t gen[4] = {22, 0, 0, 0}, gen_m[4], gen_inv[4];
    fr_mul(gen_m, gen, FR_R2);
    fr_inverse(gen_inv, gen_m);
    if (inverse) fr_inverse(w, w);
    int nt = threads > 0 ? threads : 1;
    /* coset_fft: multiply coefficient i by g^i first */
    if (coset && !inverse) fr_scale_by_powers(a, n, FR_R1, gen_m, nt);
    /* derange (bit reversal) then Cooley-Tukey DIT, natural order out.  ark-poly's fft does
     * Gentleman-Sande then derange; the outputs are the same vector. */
#pragma omp parallel for schedule(static) num_threads(nt) if (n >= 4096)
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev(i, log_n);
        if (i < j) {
            uint64_t t[4];
            memcpy(t, a + 4 * i, 32);
            memcpy(a + 4 * i, a + 4 * j, 32);
            memcpy(a + 4 * j, t, 32);
        }
    }
    /* one table of the n/2 powers of the root (ark-poly: roots_of_unity), indexed with the stage's stride */
    size_t half_n = n / 2 ? n / 2 : 1;
    uint64_t *tw = malloc(32 * half_n);
    for (size_t k = 0; k < half_n; k++) memcpy(tw + 4 * k, FR_R1, 32);
    fr_scale_by_powers(tw, half_n, FR_R1, w, nt);
    for (unsigned s = 1; s <= log_n; s++) {
        size_t len = (size_t)1 << s, half = len >> 1, stride = n / len;
        size_t total = n / 2; /* butterflies of this stage, flattened so that early stages parallelise too */
#pragma omp parallel for schedule(static) num_threads(nt) if (n >= 4096)
        for (size_t t = 0; t < total; t++) {
            size_t blk = t / half, k = t % half;
            uint64_t *base = a + 4 * blk * len;
            uint64_t u[4], v[4];
            memcpy(u, base + 4 * k, 32);
            fr_mul(v, base + 4 * (k + half), tw + 4 * (k * stride));
            fr_add(base + 4 * k, u, v);
            fr_sub(base + 4 * (k + half), u, v);
        }
    }
    free(tw);
    if (inverse) {
        /* multiply by size_inv; coset_ifft then multiplies coefficient i by g^-i */
        uint64_t ninv[4] = {n, 0, 0, 0};
        fr_mul(ninv, ninv, FR_R2);
        fr_inverse(ninv, ninv);
        fr_scale_by_powers(a, n, ninv, coset ? gen_inv : FR_R1, nt);
    }
}

/* ------------------------------------------------------------------ K3: sparse M*z (ark-marlin prover_init inner_prod_fn) */
void oracle_spmv_fr(const uint32_t *rowptr, const uint32_t *col, const uint64_t *val4, const uint64_t *z4,
                    uint64_t *out4, size_t rows) {
    oracle_spmv_fr_mt(rowptr, col, val4, z4, out4, rows, 1);
}
/* rows in parallel, as ark-marlin's cfg_iter!(matrix) under the `parallel` feature */
void oracle_spmv_fr_mt(const uint32_t *rowptr, const uint32_t *col, const uint64_t *val4, const uint64_t *z4,
                       uint64_t *out4, size_t rows, int threads) {
    int nt = threads > 0 ? threads : 1;
#pragma omp parallel for schedule(static, 1024) num_threads(nt) if (rows >= 4096)
    for (size_t r = 0; r < rows; r++) {
        uint64_t acc[4] = {0, 0, 0, 0};
        for (uint32_t k = rowptr[r]; k < rowptr[r + 1]; k++) {
            const uint64_t *coeff = val4 + 4 * (size_t)k;
            const uint64_t *zv = z4 + 4 * (size_t)col[k];
            if (limbs_eq(coeff, FR_R1, 4)) {
                fr_add(acc, acc, zv); /* coeff.is_one() shortcut */
            } else {
                uint64_t t[4];
                fr_mul(t, zv, coeff);
                fr_add(acc, acc, t);
            }
        }
        memcpy(out4 + 4 * r, acc, 32);
    }
}

/* ------------------------------------------------------------------ Pedersen CRH on ed-on-BLS12-377 + Merkle tree
 * Checker for simpleworks_amd/csrc/pedersen.hip.  Restates ark-crypto-primitives 0.3 [U] (not vendored in /root/reference):
 *   crh/pedersen CRH::evaluate: input zero-padded to WINDOW_SIZE x NUM_WINDOWS bits, bits LSB-first inside a byte; per
 *     window `encoded += generators[w][j]` for every set bit j, the windows' points summed;
 *   crh/injective_map TECompressor: the affine x coordinate;
 *   merkle_tree MerkleTree::new: leaf digests, then two-to-one hashes of to_bytes![left] || to_bytes![right] level by level;
 * as reached from /root/reference/src/merkle_tree/simple_merkle_tree.rs:47-49 with the windows of
 * /root/reference/src/merkle_tree/common.rs:11-30 and from /root/reference/src/hash/mod.rs:23-28.
 * Group law as ark-ec 0.3 twisted_edwards_extended GroupProjective::add_assign writes it (add-2008-hwcd) — bit by bit,
 * not the product's tabulated window multiples. */
typedef struct {
    uint64_t x[4], y[4], t[4], z[4];
} ed_t;
static void ed_set_identity(ed_t *p) {
    memset(p->x, 0, 32);
    memset(p->t, 0, 32);
    memcpy(p->y, FR_R1, 32);
    memcpy(p->z, FR_R1, 32);
}
static void ed_add(ed_t *r, const ed_t *p, const ed_t *q, const uint64_t *d_mont) {
    uint64_t A[4], B[4], C[4], D[4], E[4], F[4], G[4], H[4], s1[4], s2[4];
    fr_mul(A, p->x, q->x);
    fr_mul(B, p->y, q->y);
    fr_mul(C, p->t, q->t);
    fr_mul(C, C, d_mont);
    fr_mul(D, p->z, q->z);
    fr_add(H, B, A); /* H = B - a A, a = -1 */
    fr_add(s1, p->x, p->y);
    fr_add(s2, q->x, q->y);
    fr_mul(E, s1, s2);
    fr_sub(E, E, A);
    fr_sub(E, E, B);
    fr_sub(F, D, C);
    fr_add(G, D, C);
    fr_mul(r->x, E, F);
    fr_mul(r->y, G, H);
    fr_mul(r->t, E, H);
    fr_mul(r->z, F, G);
}
static ed_t *ed_load_generators(const uint64_t *gens_xy_std, size_t count) {
    ed_t *g = (ed_t *)malloc(count * sizeof(ed_t));
    for (size_t i = 0; i < count; i++) {
        fr_mul(g[i].x, gens_xy_std + 8 * i, FR_R2);
        fr_mul(g[i].y, gens_xy_std + 8 * i + 4, FR_R2);
        fr_mul(g[i].t, g[i].x, g[i].y);
        memcpy(g[i].z, FR_R1, 32);
    }
    return g;
}
static void pedersen_one(const ed_t *gens, size_t nw, size_t ws, const uint8_t *msg, size_t len, const uint64_t *d_mont,
                         uint8_t *digest32) {
    ed_t sum;
    ed_set_identity(&sum);
    for (size_t w = 0; w < nw; w++) {
        ed_t encoded;
        ed_set_identity(&encoded);
        int any = 0;
        for (size_t j = 0; j < ws; j++) {
            size_t k = w * ws + j;
            if (k < 8 * len && ((msg[k >> 3] >> (k & 7)) & 1)) {
                ed_add(&encoded, &encoded, &gens[w * ws + j], d_mont);
                any = 1;
            }
        }
        if (any) ed_add(&sum, &sum, &encoded, d_mont); /* adding the identity changes nothing: skipped */
    }
    uint64_t zi[4], x[4];
    static const uint64_t one[4] = {1, 0, 0, 0};
    fr_inverse(zi, sum.z);
    fr_mul(x, sum.x, zi);
    fr_mul(x, x, one); /* out of Montgomery form */
    for (int i = 0; i < 32; i++) digest32[i] = (uint8_t)(x[i / 8] >> (8 * (i % 8)));
}
static void ed_d_mont(uint64_t *d) {
    static const uint64_t d_std[4] = {3021, 0, 0, 0};
    fr_mul(d, d_std, FR_R2);
}
/* gens_xy_std: [nw][ws] affine points, 8 limbs each (x, y in standard form); digests: 32 little-endian bytes each */
void oracle_pedersen_hash(const uint64_t *gens_xy_std, size_t nw, size_t ws, const uint8_t *inputs, size_t len, size_t count,
                          uint8_t *digests, int threads) {
    uint64_t d[4];
    ed_d_mont(d);
    ed_t *g = ed_load_generators(gens_xy_std, nw * ws);
    int nt = threads > 0 ? threads : 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nt) if (count >= 64)
    for (size_t i = 0; i < count; i++) pedersen_one(g, nw, ws, inputs + i * len, len, d, digests + 32 * i);
    free(g);
}
/* nodes: n leaf digests | n / 2 | ... | root, 32 bytes each */
void oracle_merkle_tree(const uint64_t *leaf_gens, size_t nw_leaf, const uint64_t *inner_gens, size_t nw_inner, size_t ws,
                        const uint8_t *leaves, size_t leaf_len, size_t n, uint8_t *nodes, int threads) {
    oracle_pedersen_hash(leaf_gens, nw_leaf, ws, leaves, leaf_len, n, nodes, threads);
    size_t off = 0;
    for (size_t cnt = n; cnt > 1; cnt >>= 1) {
        oracle_pedersen_hash(inner_gens, nw_inner, ws, nodes + 32 * off, 64, cnt >> 1, nodes + 32 * (off + cnt), threads);
        off += cnt;
    }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
