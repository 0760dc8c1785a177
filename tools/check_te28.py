#!/usr/bin/env python3
"""Bit-level emulation of the lazy 28-bit twisted Edwards arithmetic of simpleworks_amd/csrc/fq28.cuh (te28_from_row,
te28_madd_row, te28_slot_add): every uint32 limb operation and every 64-bit column sum is checked for wrap-around, and the
results are compared with the group law computed with Python integers on y^2 = x^3 + 1 (through the map of
tools/gen_constants.py).  CPU only; run: python tools/check_te28.py [chains]"""
import os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
from pyref.bls12_377 import Q, R, fq_sqrt, g1_add, g1_mul_fast, g1_neg

M28 = (1 << 28) - 1
R392 = (1 << 392) % Q
inv = lambda a: pow(a % Q, Q - 2, Q)


def limbs(v):
    assert 0 <= v < 1 << 392
    return [(v >> (28 * i)) & M28 for i in range(13)] + [v >> (28 * 13)]


def val(l):
    return sum(x << (28 * i) for i, x in enumerate(l))


def spread(kp, sub):
    q = [(kp >> (28 * i)) & M28 for i in range(13)] + [kp >> (28 * 13)]
    return [q[0] + sub * (1 << 28)] + [q[i] + sub * (1 << 28) - sub for i in range(1, 13)] + [q[13] - sub]


P28 = limbs(Q)
SPREAD2, SPREAD4 = spread(2 * Q, 1), spread(4 * Q, 1)
ONE = limbs(R392)
TWO = limbs(2 * R392 % Q)


def u32(x):
    assert 0 <= x < 1 << 32, "uint32 wrap"
    return x


def mul(a, b):
    """fq28_mul, column by column; checks the documented operand bound and the 64-bit column sums."""
    assert all(x < 1 << 30 for x in a) and all(x < 1 << 30 for x in b), "operand limb >= 2^30"
    assert val(a) < 128 * Q and val(b) < 128 * Q, "operand value >= 128p"
    m, r, acc = [0] * 14, [0] * 14, 0
    for k in range(14):
        for i in range(k + 1):
            acc += a[i] * b[k - i]
        for i in range(k):
            acc += m[i] * P28[k - i]
        assert acc < 1 << 64
        m[k] = (-acc) & M28
        # the carry rule of the asm multiplier (fq28_mul_asm): one 64-bit add, one bit-select, one shift
        t = acc + M28
        assert t < 1 << 64 and (~t) & M28 == m[k] and t >> 28 == (acc + m[k]) >> 28
        acc = (acc + m[k]) >> 28
    for k in range(14, 27):
        for i in range(k - 13, 14):
            acc += a[i] * b[k - i]
        for i in range(k - 13, 14):
            acc += m[i] * P28[k - i]
        assert acc < 1 << 64
        r[k - 14] = acc & M28
        acc >>= 28
    r[13] = acc
    assert acc < 1 << 15 and val(r) < 2 * Q, "result is not N"
    assert (val(r) * (1 << 392) - val(a) * val(b)) % Q == 0
    return r


def sub(a, b, sp):
    return [u32(a[i] + sp[i] - b[i]) for i in range(14)]


def add(a, b):
    return [u32(a[i] + b[i]) for i in range(14)]


def normalize(a):
    r, c = [0] * 14, 0
    for i in range(13):
        t = u32(a[i] + c)
        r[i] = t & M28
        c = t >> 28
    r[13] = u32(a[13] + c)
    return r


# ---- curve constants (same derivation as tools/gen_constants.py)
rt3 = min(fq_sqrt(3), Q - fq_sqrt(3))
te_s = inv(rt3)
te_A = (-3 * te_s) % Q
a1 = (te_A + 2) * inv(te_s) % Q
d1 = (te_A - 2) * inv(te_s) % Q
te_f = min(fq_sqrt(-a1 % Q), Q - fq_sqrt(-a1 % Q))
te_d = d1 * inv(-a1) % Q
K2D = limbs(2 * te_d * R392 % Q)
INVD = limbs(inv(te_d) * R392 % Q)


def w2te(P):
    if P is None:
        return (0, 1)
    u = te_s * (P[0] + 1) % Q
    return (te_f * (P[0] + 1) * inv(P[1]) % Q, (u - 1) * inv(u + 1) % Q)


def te2w(T):
    if T == (0, 1):
        return None
    u = (1 + T[1]) * inv(1 - T[1]) % Q
    return ((u * rt3 - 1) % Q, te_f * rt3 * u * inv(T[0]) % Q)


def row_of(P):  # table row in the 28-bit domain (coordinates x 2^392, canonical)
    x, y = w2te(P)
    return [limbs(v * R392 % Q) for v in ((y - x) % Q, (y + x) % Q, 2 * te_d * x * y % Q)]


def from_row(m2, s2, k2, neg):
    d = [u32(SPREAD2[i] + (m2[i] - s2[i] if neg else s2[i] - m2[i])) for i in range(14)]
    x = normalize(d)
    y = normalize(add(s2, m2))
    ks = [u32(SPREAD4[i] - k2[i]) if neg else k2[i] for i in range(14)]
    return [x, y, mul(ks, INVD), TWO]


def madd_row(a, row, neg):
    x, y, t, z = a
    d = [u32(y[i] + SPREAD4[i] - x[i]) for i in range(14)]
    s = add(y, x)
    # r04: the row of -P is (y + x, y - x, -k): the first two coordinates are LOADED from each other's address
    A, B, C = mul(d, row[1] if neg else row[0]), mul(s, row[0] if neg else row[1]), mul(t, row[2])
    E = sub(B, A, SPREAD4)
    H = add(A, B)
    D = add(z, z)
    dm, dp = sub(D, C, SPREAD4), add(D, C)
    F, G = (dp, dm) if neg else (dm, dp)
    return [mul(E, F), mul(G, H), mul(E, H), mul(F, G)]


def slot_add(a, q):
    A = mul(sub(a[1], a[0], SPREAD4), sub(q[1], q[0], SPREAD4))
    B = mul(add(a[1], a[0]), add(q[1], q[0]))
    C = mul(mul(a[2], q[2]), K2D)
    D = mul(a[3], q[3])
    D = add(D, D)
    E, H = sub(B, A, SPREAD4), add(B, A)
    F, G = sub(D, C, SPREAD4), add(D, C)
    return [mul(E, F), mul(G, H), mul(E, H), mul(F, G)]


def to_w(a):  # extended point in the 28-bit domain -> affine on the Weierstrass curve (None = identity)
    X, Y, T, Z = (val(c) * inv(R392) % Q for c in a)
    assert T * Z % Q == X * Y % Q, "T Z != X Y"
    zi = inv(Z)
    return te2w((X * zi % Q, Y * zi % Q))


IDENT = [limbs(0), ONE, limbs(0), ONE]


def main():
    chains = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    random.seed(7)
    G = (81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
         241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030)
    n_add = 0
    partials = []
    for c in range(chains):
        pts = [g1_mul_fast(G, random.randrange(1, R)) for _ in range(8)]
        if c == 1:
            pts[3] = pts[2]            # doubling inside a chain
        if c == 2:
            pts[1] = g1_neg(pts[0])    # cancellation: the running sum passes through the identity
        signs = [random.random() < 0.5 for _ in pts]
        acc = from_row(*row_of(pts[0]), signs[0])
        ref = g1_neg(pts[0]) if signs[0] else pts[0]
        assert to_w(acc) == ref
        for P, sg in zip(pts[1:], signs[1:]):
            acc = madd_row(acc, row_of(P), sg)
            ref = g1_add(ref, g1_neg(P) if sg else P)
            assert to_w(acc) == ref, "madd mismatch"
            n_add += 1
        partials.append((acc, ref))
    # bucket stage: general additions of partial sums, of a point with itself, with the identity, with single-row partials
    tot, ref = IDENT, None
    for acc, r in partials:
        tot = slot_add(tot, acc)
        ref = g1_add(ref, r)
        assert to_w(tot) == ref
    dbl = slot_add(tot, tot)
    assert to_w(dbl) == g1_add(ref, ref)
    assert to_w(slot_add(IDENT, IDENT)) is None
    one_row = from_row(*row_of(G), True)            # X in (p, 3p): the widest value a stored partial sum can carry
    assert to_w(slot_add(one_row, tot)) == g1_add(g1_neg(G), ref)
    assert to_w(slot_add(one_row, one_row)) == g1_neg(g1_add(G, G))
    # worst-case limb patterns for the mixed addition: accumulator coordinates with all limbs at 2^28 - 1 are not field
    # elements below 2p, so the bound is exercised with the largest N values instead (2p - 1, limbs normalised)
    big = limbs(2 * Q - 1)
    row = row_of(G)
    for neg in (False, True):
        madd_row([big, big, big, big], row, neg)
        madd_row([normalize(limbs(3 * Q - 1)), big, big, TWO], row, neg)   # from_row's X range
    print("te28 emulation OK: %d mixed additions, bucket-stage additions, worst-case limbs; no uint32 / uint64 wrap" % n_add)


if __name__ == "__main__":
    main()
