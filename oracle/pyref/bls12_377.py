"""Python big-int model of BLS12-377 (Fr, Fq, G1, G2, pairing).

TEST INFRASTRUCTURE ONLY.  This module is part of the oracle: it may be imported by
tests/, tests/golden/gen_golden.py, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product path (simpleworks_amd/).

The reference (lambdaclass/simpleworks) delegates all of this to arkworks 0.3 crates that are
not vendored under /root/reference (Cargo.toml:15-30), so every routine here restates the
*published* algorithm (ark-ff / ark-ec / ark-bls12-377 0.3.0) and is anchored on the reference's
call sites: src/marlin/mod.rs:12-14 (curve = Bls12_377) and src/gadgets/mod.rs:29 (field).
Parity status: **unpinned** against arkworks (no Rust toolchain, no known-answer vectors in the
reference's tests); pinned against mathematical invariants checked in tests/ (group orders,
bilinearity, on-curve checks) and SURVEY.md Appendix B constants.
"""

# ----------------------------------------------------------------------------- parameters
X = 0x8508C00000000001  # BLS parameter
R = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001  # scalar field Fr
Q = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001  # base field Fq

assert R == X**4 - X**2 + 1
assert Q == (X - 1) ** 2 * R // 3 + X

FR_BITS = 253
FQ_BITS = 377
FR_LIMBS = 4  # u64
FQ_LIMBS = 6  # u64
FR_MONT_R = (1 << 256) % R
FQ_MONT_R = (1 << 384) % Q
FR_TWO_ADICITY = 47
FR_GENERATOR = 22
FR_ROOT_2_47 = pow(FR_GENERATOR, (R - 1) >> 47, R)

G1_B = 1
G1_COFACTOR = 0x170B5D44300000000000000000000000
G1_GEN = (
    81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
    241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030,
)
assert G1_COFACTOR == (X - 1) ** 2 // 3

# Fq2 = Fq[u]/(u^2 + 5); Fq6 = Fq2[v]/(v^3 - u); Fq12 = Fq6[w]/(w^2 - v)   (ark-bls12-377 0.3)
FQ2_NONRESIDUE = Q - 5
# G2 (D-type twist): y^2 = x^3 + 1/u  over Fq2
G2_COFACTOR = (X**8 - 4 * X**7 + 5 * X**6 - 4 * X**4 + 6 * X**3 - 4 * X**2 - 4 * X + 13) // 9


# ----------------------------------------------------------------------------- limb helpers
def to_limbs(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs(limbs):
    return sum(int(l) << (64 * i) for i, l in enumerate(limbs))


def fr_to_mont(a):
    return a * FR_MONT_R % R


def fr_from_mont(a):
    return a * pow(FR_MONT_R, -1, R) % R


def fq_to_mont(a):
    return a * FQ_MONT_R % Q


def fq_from_mont(a):
    return a * pow(FQ_MONT_R, -1, Q) % Q


def hex_le_bytes(v, nbytes):
    return v.to_bytes(nbytes, "little").hex()


# ----------------------------------------------------------------------------- Fr helpers
def fr_inv(a):
    return pow(a, -1, R)


def fr_root_of_unity(log_n):
    """ark-poly Radix2EvaluationDomain::new: group_gen = TWO_ADIC_ROOT^(2^(47-log_n))."""
    assert 0 <= log_n <= FR_TWO_ADICITY
    return pow(FR_ROOT_2_47, 1 << (FR_TWO_ADICITY - log_n), R)


# ----------------------------------------------------------------------------- G1 (affine, ints mod Q)
# A point is None (infinity) or (x, y) with standard-form integers.
def g1_is_on_curve(P):
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - G1_B) % Q == 0


def g1_neg(P):
    if P is None:
        return None
    return (P[0], (-P[1]) % Q)


def g1_add(P, S):
    if P is None:
        return S
    if S is None:
        return P
    x1, y1 = P
    x2, y2 = S
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    y3 = (lam * (x1 - x3) - y1) % Q
    return (x3, y3)


def g1_mul(P, k):
    """Double-and-add over Jacobian-free affine arithmetic (oracle: clarity over speed)."""
    if k < 0:
        return g1_mul(g1_neg(P), -k)
    acc = None
    add = P
    while k:
        if k & 1:
            acc = g1_add(acc, add)
        add = g1_add(add, add)
        k >>= 1
    return acc


# Jacobian version for bulk work in the pure-Python Marlin reference (same results, much faster)
def g1_jac_double(P):
    X1, Y1, Z1 = P
    if Z1 == 0:
        return P
    A = X1 * X1 % Q
    B = Y1 * Y1 % Q
    C = B * B % Q
    D = 2 * ((X1 + B) * (X1 + B) - A - C) % Q
    E = 3 * A % Q
    F = E * E % Q
    X3 = (F - 2 * D) % Q
    Y3 = (E * (D - X3) - 8 * C) % Q
    Z3 = 2 * Y1 * Z1 % Q
    return (X3, Y3, Z3)


def g1_jac_add_affine(P, S):
    """P Jacobian + S affine (None = infinity)."""
    if S is None:
        return P
    X1, Y1, Z1 = P
    x2, y2 = S
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % Q
    U2 = x2 * Z1Z1 % Q
    S2 = y2 * Z1 * Z1Z1 % Q
    if U2 == X1:
        if S2 == Y1:
            return g1_jac_double(P)
        return (1, 1, 0)
    H = (U2 - X1) % Q
    HH = H * H % Q
    I = 4 * HH % Q
    J = H * I % Q
    r = 2 * (S2 - Y1) % Q
    V = X1 * I % Q
    X3 = (r * r - J - 2 * V) % Q
    Y3 = (r * (V - X3) - 2 * Y1 * J) % Q
    Z3 = ((Z1 + H) * (Z1 + H) - Z1Z1 - HH) % Q
    return (X3, Y3, Z3)


def g1_jac_to_affine(P):
    X1, Y1, Z1 = P
    if Z1 == 0:
        return None
    zi = pow(Z1, -1, Q)
    zi2 = zi * zi % Q
    return (X1 * zi2 % Q, Y1 * zi2 * zi % Q)


def g1_mul_fast(P, k):
    if P is None or k == 0:
        return None
    if k < 0:
        return g1_mul_fast(g1_neg(P), -k)
    acc = (1, 1, 0)
    for bit in bin(k)[2:]:
        acc = g1_jac_double(acc)
        if bit == "1":
            acc = g1_jac_add_affine(acc, P)
    return g1_jac_to_affine(acc)


def g1_msm_naive(bases, scalars):
    acc = (1, 1, 0)
    for P, s in zip(bases, scalars):
        t = g1_mul_fast(P, s % R)
        acc = g1_jac_add_affine(acc, t)
    return g1_jac_to_affine(acc)


def ark_msm_window(n):
    """ark-ec 0.3 VariableBaseMSM window rule [U]."""
    if n < 32:
        return 3
    return (n - 1).bit_length() * 69 // 100 + 2


def g1_msm_pippenger(bases, scalars):
    """Restatement of ark_ec::msm::VariableBaseMSM::multi_scalar_mul (0.3.0) [U]."""
    size = min(len(bases), len(scalars))
    pairs = [(scalars[i], bases[i]) for i in range(size) if scalars[i] != 0]
    c = ark_msm_window(size)
    num_bits = FR_BITS
    window_sums = []
    for w in range(0, num_bits, c):
        res = (1, 1, 0)
        buckets = [(1, 1, 0)] * ((1 << c) - 1)
        for s, b in pairs:
            if s == 1:
                if w == 0:
                    res = g1_jac_add_affine(res, b)
            else:
                d = (s >> w) % (1 << c)
                if d != 0:
                    buckets[d - 1] = g1_jac_add_affine(buckets[d - 1], b)
        running = (1, 1, 0)
        for b in reversed(buckets):
            running = g1_jac_add_affine(running, g1_jac_to_affine(b))
            res = g1_jac_add_affine(res, g1_jac_to_affine(running))
        window_sums.append(res)
    lowest = window_sums[0]
    total = (1, 1, 0)
    for ws in reversed(window_sums[1:]):
        total = g1_jac_add_affine(total, g1_jac_to_affine(ws))
        for _ in range(c):
            total = g1_jac_double(total)
    total = g1_jac_add_affine(total, g1_jac_to_affine(lowest))
    return g1_jac_to_affine(total)


# ----------------------------------------------------------------------------- Fq2 / Fq6 / Fq12
class Fq2:
    __slots__ = ("c0", "c1")

    def __init__(self, c0, c1=0):
        self.c0 = c0 % Q
        self.c1 = c1 % Q

    def __eq__(self, o):
        return self.c0 == o.c0 and self.c1 == o.c1

    def __hash__(self):
        return hash((self.c0, self.c1))

    def is_zero(self):
        return self.c0 == 0 and self.c1 == 0

    def __add__(self, o):
        return Fq2(self.c0 + o.c0, self.c1 + o.c1)

    def __sub__(self, o):
        return Fq2(self.c0 - o.c0, self.c1 - o.c1)

    def __neg__(self):
        return Fq2(-self.c0, -self.c1)

    def __mul__(self, o):
        if isinstance(o, int):
            return Fq2(self.c0 * o, self.c1 * o)
        # u^2 = -5
        return Fq2(self.c0 * o.c0 - 5 * self.c1 * o.c1, self.c0 * o.c1 + self.c1 * o.c0)

    def square(self):
        return self * self

    def conj(self):
        return Fq2(self.c0, -self.c1)

    def inv(self):
        n = pow(self.c0 * self.c0 + 5 * self.c1 * self.c1, -1, Q)
        return Fq2(self.c0 * n, -self.c1 * n)

    def pow(self, e):
        acc = Fq2(1)
        b = self
        while e:
            if e & 1:
                acc = acc * b
            b = b * b
            e >>= 1
        return acc

    def legendre_is_qr(self):
        if self.is_zero():
            return True
        norm = (self.c0 * self.c0 + 5 * self.c1 * self.c1) % Q
        return pow(norm, (Q - 1) // 2, Q) == 1

    def sqrt(self):
        """Any square root or None (root choice is fixed afterwards by the caller's ordering rule)."""
        if self.is_zero():
            return Fq2(0)
        if not self.legendre_is_qr():
            return None
        # complex method: alpha = norm^(1/2) in Fq; delta = (c0 + alpha)/2 must be QR else use (c0 - alpha)/2
        norm = (self.c0 * self.c0 + 5 * self.c1 * self.c1) % Q
        alpha = fq_sqrt(norm)
        assert alpha is not None
        two_inv = pow(2, -1, Q)
        delta = (self.c0 + alpha) * two_inv % Q
        if pow(delta, (Q - 1) // 2, Q) != 1 and delta != 0:
            delta = (self.c0 - alpha) * two_inv % Q
        c0 = fq_sqrt(delta)
        assert c0 is not None
        if c0 == 0:
            # then c1^2 * (-5) = self.c0
            c1 = fq_sqrt(self.c0 * pow(Q - 5, -1, Q) % Q)
            r = Fq2(0, c1)
        else:
            c1 = self.c1 * pow(2 * c0, -1, Q) % Q
            r = Fq2(c0, c1)
        assert r * r == self
        return r

    def ark_lt(self, o):
        """ark-ff 0.3 QuadExtField Ord: compare c1 first, then c0 [U]."""
        if self.c1 != o.c1:
            return self.c1 < o.c1
        return self.c0 < o.c0


def fq_sqrt(a):
    """Tonelli-Shanks in Fq (two-adicity 46)."""
    a %= Q
    if a == 0:
        return 0
    if pow(a, (Q - 1) // 2, Q) != 1:
        return None
    s = 46
    t = (Q - 1) >> s
    # find non-residue
    z = 2
    while pow(z, (Q - 1) // 2, Q) == 1:
        z += 1
    c = pow(z, t, Q)
    x = pow(a, (t + 1) // 2, Q)
    b = pow(a, t, Q)
    m = s
    while b != 1:
        i = 0
        bb = b
        while bb != 1:
            bb = bb * bb % Q
            i += 1
        g = pow(c, 1 << (m - i - 1), Q)
        x = x * g % Q
        c = g * g % Q
        b = b * c % Q
        m = i
    assert x * x % Q == a
    return x


XI = Fq2(0, 1)  # Fq6 non-residue


class Fq6:
    __slots__ = ("c0", "c1", "c2")

    def __init__(self, c0, c1=None, c2=None):
        self.c0 = c0
        self.c1 = c1 if c1 is not None else Fq2(0)
        self.c2 = c2 if c2 is not None else Fq2(0)

    def __eq__(self, o):
        return self.c0 == o.c0 and self.c1 == o.c1 and self.c2 == o.c2

    def is_zero(self):
        return self.c0.is_zero() and self.c1.is_zero() and self.c2.is_zero()

    def __add__(self, o):
        return Fq6(self.c0 + o.c0, self.c1 + o.c1, self.c2 + o.c2)

    def __sub__(self, o):
        return Fq6(self.c0 - o.c0, self.c1 - o.c1, self.c2 - o.c2)

    def __neg__(self):
        return Fq6(-self.c0, -self.c1, -self.c2)

    def __mul__(self, o):
        a0, a1, a2 = self.c0, self.c1, self.c2
        b0, b1, b2 = o.c0, o.c1, o.c2
        # v^3 = XI
        c0 = a0 * b0 + (a1 * b2 + a2 * b1) * XI
        c1 = a0 * b1 + a1 * b0 + (a2 * b2) * XI
        c2 = a0 * b2 + a1 * b1 + a2 * b0
        return Fq6(c0, c1, c2)

    def mul_by_v(self):
        return Fq6(self.c2 * XI, self.c0, self.c1)

    def inv(self):
        a0, a1, a2 = self.c0, self.c1, self.c2
        t0 = a0 * a0 - (a1 * a2) * XI
        t1 = (a2 * a2) * XI - a0 * a1
        t2 = a1 * a1 - a0 * a2
        d = (a0 * t0 + ((a2 * t1) + (a1 * t2)) * XI).inv()
        return Fq6(t0 * d, t1 * d, t2 * d)


class Fq12:
    __slots__ = ("c0", "c1")

    def __init__(self, c0, c1=None):
        self.c0 = c0
        self.c1 = c1 if c1 is not None else Fq6(Fq2(0))

    @staticmethod
    def one():
        return Fq12(Fq6(Fq2(1)))

    def __eq__(self, o):
        return self.c0 == o.c0 and self.c1 == o.c1

    def __add__(self, o):
        return Fq12(self.c0 + o.c0, self.c1 + o.c1)

    def __sub__(self, o):
        return Fq12(self.c0 - o.c0, self.c1 - o.c1)

    def __neg__(self):
        return Fq12(-self.c0, -self.c1)

    def __mul__(self, o):
        # w^2 = v
        a0, a1 = self.c0, self.c1
        b0, b1 = o.c0, o.c1
        return Fq12(a0 * b0 + (a1 * b1).mul_by_v(), a0 * b1 + a1 * b0)

    def inv(self):
        d = (self.c0 * self.c0 - (self.c1 * self.c1).mul_by_v()).inv()
        return Fq12(self.c0 * d, -(self.c1 * d))

    def pow(self, e):
        acc = Fq12.one()
        for bit in bin(e)[2:]:
            acc = acc * acc
            if bit == "1":
                acc = acc * self
        return acc

    def is_one(self):
        return self == Fq12.one()

    def flat(self):
        """12 Fq coefficients in tower order c0.c0.c0, c0.c0.c1, c0.c1.c0 ... c1.c2.c1."""
        out = []
        for six in (self.c0, self.c1):
            for two in (six.c0, six.c1, six.c2):
                out += [two.c0, two.c1]
        return out


def fq12_from_fq(a):
    return Fq12(Fq6(Fq2(a)))


def fq12_from_fq2(a):
    return Fq12(Fq6(a))


# w and its powers as Fq12 elements: w = (0, 1) in Fq6[w]
FQ12_W = Fq12(Fq6(Fq2(0)), Fq6(Fq2(1)))
FQ12_W2 = FQ12_W * FQ12_W
FQ12_W3 = FQ12_W2 * FQ12_W

# ----------------------------------------------------------------------------- G2 (affine over Fq2)
G2_B = XI.inv()  # 1/u  (b / xi with b = 1, D-type twist)


def g2_is_on_curve(P):
    if P is None:
        return True
    x, y = P
    return y * y == x * x * x + G2_B


def g2_neg(P):
    return None if P is None else (P[0], -P[1])


def g2_add(P, S):
    if P is None:
        return S
    if S is None:
        return P
    x1, y1 = P
    x2, y2 = S
    if x1 == x2:
        if (y1 + y2).is_zero():
            return None
        lam = (x1 * x1 * 3) * (y1 * 2).inv()
    else:
        lam = (y2 - y1) * (x2 - x1).inv()
    x3 = lam * lam - x1 - x2
    y3 = lam * (x1 - x3) - y1
    return (x3, y3)


def g2_mul(P, k):
    acc = None
    add = P
    while k:
        if k & 1:
            acc = g2_add(acc, add)
        add = g2_add(add, add)
        k >>= 1
    return acc


# ----------------------------------------------------------------------------- pairing
def _untwist(Qp):
    """E'(Fq2) -> E(Fq12): (x', y') -> (x' w^2, y' w^3)   (D-type twist, w^6 = xi)."""
    x, y = Qp
    return (fq12_from_fq2(x) * FQ12_W2, fq12_from_fq2(y) * FQ12_W3)


def miller_loop(P, Qp):
    """Ate Miller loop f_{x,Q}(P) with generic affine line functions in Fq12 (oracle: clarity)."""
    if P is None or Qp is None:
        return Fq12.one()
    xq, yq = _untwist(Qp)
    xp, yp = fq12_from_fq(P[0]), fq12_from_fq(P[1])
    f = Fq12.one()
    tx, ty = xq, yq
    three = fq12_from_fq(3)
    two = fq12_from_fq(2)
    for bit in bin(X)[3:]:
        lam = (tx * tx * three) * (ty * two).inv()
        f = f * f * (yp - ty - lam * (xp - tx))
        nx = lam * lam - tx - tx
        ty = lam * (tx - nx) - ty
        tx = nx
        if bit == "1":
            lam = (yq - ty) * (xq - tx).inv()
            f = f * (yp - ty - lam * (xp - tx))
            nx = lam * lam - tx - xq
            ty = lam * (tx - nx) - ty
            tx = nx
    return f


FINAL_EXP = (Q**12 - 1) // R


def final_exponentiation(f):
    return f.pow(FINAL_EXP)


def pairing(P, Qp):
    return final_exponentiation(miller_loop(P, Qp))


def product_of_pairings_is_one(pairs):
    f = Fq12.one()
    for P, Qp in pairs:
        f = f * miller_loop(P, Qp)
    return final_exponentiation(f).is_one()
