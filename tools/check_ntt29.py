#!/usr/bin/env python3
"""Bit-level emulation of the lazy 29-bit transform arithmetic of simpleworks_amd/csrc/fr29.cuh + ntt.hip (ntt_pass_lazy): every
uint32 limb operation and every 64-bit column sum is checked for wrap-around, the value bounds the host plan assumes are asserted,
and whole small transforms (every pass plan up to 2^12, forward / inverse / coset) are compared with a direct DFT in Python
integers.  CPU only; run: python tools/check_ntt29.py"""
import random, sys

R = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
M29 = (1 << 29) - 1
R261 = (1 << 261) % R
GEN = 22
ROOT47 = pow(GEN, (R - 1) >> 47, R)


def limbs(v):
    assert 0 <= v < 1 << 264
    return [(v >> (29 * i)) & M29 for i in range(8)] + [v >> 232]


def val(l):
    return sum(x << (29 * i) for i, x in enumerate(l))


def spread(k, sub):
    """limbs of k*r with sub * 2^29 borrowed into every limb below the top one"""
    q = limbs(k * R)
    l = [q[0] + sub * (1 << 29)] + [q[i] + sub * (1 << 29) - sub for i in range(1, 8)] + [q[8] - sub]
    assert val(l) == k * R and l[8] > 0
    return l


P29 = limbs(R)
ONE = limbs(R261)
assert P29[0] == 1


def u32(x):
    assert 0 <= x < 1 << 32, "uint32 wrap"
    return x


def mul(a, b):
    """fr29_mul: a may be lazy (limbs < 2^32), b normalised (limbs < 2^29: a table entry, or the product of two); a b < 2^261 r."""
    assert all(0 <= x < 1 << 32 for x in a) and all(0 <= x < 1 << 29 for x in b[:8]) and b[8] < 1 << 23
    assert val(a) * val(b) < (1 << 261) * R, "product too large for a result below 2r"
    m, r, acc = [0] * 9, [0] * 9, 0
    for k in range(9):
        for i in range(k + 1):
            acc += a[i] * b[k - i]
        for i in range(k):
            acc += m[i] * P29[k - i]
        assert acc < 1 << 64, "column sum wraps"
        m[k] = (-acc) & M29
        t = acc + M29  # the carry rule of fr29_mul_asm: one 64-bit add, one bit-select, one shift
        assert t < 1 << 64 and (~t) & M29 == m[k] and t >> 29 == (acc + m[k]) >> 29, "asm carry rule"
        acc = (acc + m[k]) >> 29
    for k in range(9, 17):
        for i in range(k - 8, 9):
            acc += a[i] * b[k - i]
        for i in range(k - 8, 9):
            acc += m[i] * P29[k - i]
        assert acc < 1 << 64, "column sum wraps"
        r[k - 9] = acc & M29
        acc >>= 29
    r[8] = acc
    assert val(r) < 2 * R and (val(r) * (1 << 261) - val(a) * val(b)) % R == 0
    return r


def add(a, b):
    return [u32(x + y) for x, y in zip(a, b)]


def sub(a, b, sp):
    return [u32(x + s - y) for x, y, s in zip(a, b, sp)]


def normalize(a):
    r, c = [0] * 9, 0
    for i in range(8):
        t = u32(a[i] + c)
        r[i] = t & M29
        c = t >> 29
    r[8] = u32(a[8] + c)
    return r


def canonical(a, bound):
    """normalised value < bound * r (bound <= 4) -> < r: conditional subtractions of 2r and r"""
    v = val(a)
    assert v < bound * R
    if bound > 2 and v >= 2 * R:
        v -= 2 * R
    if v >= R:
        v -= R
    return limbs(v)


# ---- the host plan of one pass: spreads and y0 reductions per step (mirrors ntt.hip plan_lazy_pass)
def plan(log_r, b0):
    steps, B = [], b0
    if log_r & 1:
        steps.append(("r2", B))
        B = max(2 * B, 2)
    for s in range(log_r // 2):
        last = s == log_r // 2 - 1
        red = 4 * B > 64 or last
        assert B <= 64
        steps.append(("r4", B, red))
        B = 4 if red else 4 * B
    if log_r == 1:
        B = max(B, 2)
    return steps, B


def tile_transform(x, log_r, tw_small, b0):
    """x: list of 2^log_r Fr29 (normalised, value < b0 r).  DIF in place (bit-reversed out), as the kernel's LDS phase."""
    Rn = 1 << log_r
    steps, _ = plan(log_r, b0)
    h = Rn >> 1
    t = list(x)
    for st in steps:
        if st[0] == "r2":
            B = st[1]
            S = spread(2 * B, 1)
            for pos in range(h):
                u, v = t[pos], t[pos + h]
                assert val(u) < B * R and val(v) < B * R
                t[pos] = normalize(add(u, v))
                t[pos + h] = mul(sub(u, v, S), tw_small[pos])
            h >>= 1
        else:
            _, B, red = st
            S1, S2, S4 = spread(2 * B, 1), spread(4 * B, 2), spread(4, 1)
            hh, s1 = h >> 1, (Rn >> 1) // h
            s2 = 2 * s1
            for qq in range(Rn >> 2):
                p, blk = qq & (hh - 1), qq // hh
                i0 = blk * 2 * h + p
                i1, i2, i3 = i0 + hh, i0 + h, i0 + h + hh
                x0, x1, x2, x3 = t[i0], t[i1], t[i2], t[i3]
                for e in (x0, x1, x2, x3):
                    assert val(e) < B * R and all(l < 1 << 29 for l in e[:8])
                a0 = add(x0, x2)
                a2 = mul(sub(x0, x2, S1), tw_small[p * s1])
                a1 = add(x1, x3)
                a3 = mul(sub(x1, x3, S1), tw_small[(p + hh) * s1])
                w = tw_small[p * s2]
                y0 = add(a0, a1)
                y0 = mul(y0, ONE) if red else normalize(y0)
                y1 = mul(sub(a0, a1, S2), w)
                y2 = normalize(add(a2, a3))
                y3 = mul(sub(a2, a3, S4), w)
                t[i0], t[i1], t[i2], t[i3] = y0, y1, y2, y3
            h >>= 2
    return t


def bitrev(x, bits):
    return int(bin(x)[2:].zfill(bits)[::-1], 2) if bits else 0


def root(log_n, inverse):
    w = pow(ROOT47, 1 << (47 - log_n), R)
    return pow(w, R - 2, R) if inverse else w


def ntt_lazy(x, log_n, inverse, coset, maxr=10):
    """whole transform with the kernel's pass plan (Stockham index maps of ntt_pass), lazy arithmetic, natural in / out"""
    n = 1 << log_n
    npass = 1 if log_n <= maxr else (log_n + maxr - 1) // maxr
    radices, rem = [], log_n
    for p in range(npass):
        r = (rem + (npass - p) - 1) // (npass - p)
        radices.append(r)
        rem -= r
    w = root(log_n, inverse)
    g = pow(GEN, R - 2, R) if inverse else GEN
    to29 = lambda v: limbs(v % R * R261 % R)          # twiddles: x 2^261 (Montgomery form of the lazy domain)
    src = [limbs(v) for v in x]                       # data stays in whatever domain it came in (the factor is invariant)
    log_ns = 0
    for p, log_r in enumerate(radices):
        Rn, stride = 1 << log_r, n >> log_r
        ns_mask, tw_shift = (1 << log_ns) - 1, log_n - log_ns - log_r
        coset_in = coset and not inverse and p == 0
        last = p == npass - 1
        b0 = 4 if p > 0 else 1
        tw_small = [to29(pow(root(log_r, inverse), e, R)) for e in range(max(Rn >> 1, 1))]
        dst = [None] * n
        b_in = 2 if (p > 0 or coset_in) else 1
        _, b_out = plan(log_r, b_in)
        for j in range(stride):
            tile = []
            for t in range(Rn):
                idx = j + t * stride
                xx = src[idx]
                assert val(xx) < b0 * R
                if coset_in:
                    xx = mul(xx, to29(pow(g, idx, R)))
                if log_ns:
                    k = j & ns_mask
                    xx = mul(xx, to29(pow(w, (k * t) << tw_shift, R)))      # always (w^0 = one): a uniform bound after the load
                tile.append(xx)
            tile = tile_transform(tile, log_r, tw_small, b_in)
            k = j & ns_mask
            for u in range(Rn):
                o = ((j - k) << log_r) + k + (u << log_ns)
                y = tile[bitrev(u, log_r)]
                assert val(y) < b_out * R <= 4 * R
                if last and inverse:
                    s = pow(n, R - 2, R)
                    if coset:
                        s = s * pow(g, o, R) % R
                    y = canonical(mul(y, to29(s)), 2)
                elif last:
                    y = canonical(normalize(y), b_out)
                dst[o] = y
        src = dst
        log_ns += log_r
    return [val(v) for v in src]


def dft(x, log_n, inverse, coset):
    n = 1 << log_n
    w = root(log_n, inverse)
    if coset and not inverse:
        x = [v * pow(GEN, i, R) % R for i, v in enumerate(x)]
    out = [sum(x[i] * pow(w, i * k, R) for i in range(n)) % R for k in range(n)]
    if inverse:
        ni = pow(n, R - 2, R)
        out = [v * ni % R for v in out]
        if coset:
            gi = pow(GEN, R - 2, R)
            out = [v * pow(gi, i, R) % R for i, v in enumerate(out)]
    return out


def main():
    random.seed(11)
    # worst-case limbs through one multiplication: lazy operand with every lower limb at 2.5 * 2^30 and the value at the 446 r edge
    big = [min((5 << 29) - 1, (1 << 32) - 1)] * 8 + [0]
    big[8] = ((1 << 261) - 1 - val(big)) >> 232
    mul(big, limbs(R - 1))
    for log_r in range(1, 13):                       # every tile size (11, 12: SWM_NTT_MAXR), random data at the bound of the plan
        for b0 in (1, 2):
            tw = [limbs(pow(root(log_r, False), e, R) * R261 % R) for e in range(max((1 << log_r) >> 1, 1))]
            xs = [limbs(random.randrange(b0 * R)) for _ in range(1 << log_r)]
            xs[0] = limbs(b0 * R - 1)
            tile_transform(xs, log_r, tw, b0)
    checked = 0
    for log_n, maxr in ((1, 10), (2, 10), (3, 10), (5, 10), (6, 10), (7, 3), (8, 4), (9, 3), (6, 2), (10, 10)):
        x = [random.randrange(R) for _ in range(1 << log_n)]
        for inverse in (False, True):
            for coset in (False, True):
                assert ntt_lazy(x, log_n, inverse, coset, maxr) == dft(x, log_n, inverse, coset), (log_n, maxr, inverse, coset)
                checked += 1
    print("fr29 emulation OK: %d transforms equal the direct DFT; no uint32 / uint64 wrap, all plan bounds hold" % checked)


if __name__ == "__main__":
    main()
