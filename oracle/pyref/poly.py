"""Python restatement of the ark-poly 0.3 routines on the Marlin prove() path (oracle; test infra only).

Follows SURVEY.md Appendix A.3 [U] (ark_poly::Radix2EvaluationDomain, DensePolynomial); reference
call sites: /root/reference/src/marlin/mod.rs:75,92 (prove / index reach these through ark-marlin).
All values are standard-form integers mod R.  Polynomials are coefficient lists, low degree first.
"""
from .bls12_377 import R, FR_GENERATOR, fr_root_of_unity, fr_inv


class Domain:
    """Radix2EvaluationDomain::new(n): size = next_pow2(n) (min 1)."""

    def __init__(self, n):
        size = 1
        log = 0
        while size < n:
            size <<= 1
            log += 1
        self.size = size
        self.log = log
        self.gen = fr_root_of_unity(log)
        self.gen_inv = fr_inv(self.gen)
        self.size_inv = fr_inv(size % R)

    def element(self, i):
        return pow(self.gen, i, R)

    def elements(self):
        out = [1] * self.size
        for i in range(1, self.size):
            out[i] = out[i - 1] * self.gen % R
        return out

    def vanishing(self, tau):
        return (pow(tau, self.size, R) - 1) % R

    # -- transforms (natural order in, natural order out)
    def _dft(self, a, w):
        n = self.size
        a = list(a) + [0] * (n - len(a))
        assert len(a) == n
        # iterative radix-2 DIT after bit reversal
        j = 0
        for i in range(1, n):
            bit = n >> 1
            while j & bit:
                j ^= bit
                bit >>= 1
            j ^= bit
            if i < j:
                a[i], a[j] = a[j], a[i]
        length = 2
        while length <= n:
            wl = pow(w, n // length, R)
            half = length >> 1
            for s in range(0, n, length):
                t = 1
                for k in range(s, s + half):
                    u = a[k]
                    v = a[k + half] * t % R
                    a[k] = (u + v) % R
                    a[k + half] = (u - v) % R
                    t = t * wl % R
            length <<= 1
        return a

    def fft(self, coeffs):
        return self._dft(coeffs, self.gen)

    def ifft(self, evals):
        out = self._dft(evals, self.gen_inv)
        return [x * self.size_inv % R for x in out]

    def coset_fft(self, coeffs):
        g = 1
        scaled = []
        for c in coeffs:
            scaled.append(c * g % R)
            g = g * FR_GENERATOR % R
        return self.fft(scaled)

    def coset_ifft(self, evals):
        out = self.ifft(evals)
        gi = fr_inv(FR_GENERATOR)
        g = 1
        res = []
        for c in out:
            res.append(c * g % R)
            g = g * gi % R
        return res

    def reindex_by_subdomain(self, other, index):
        """ark-poly EvaluationDomain::reindex_by_subdomain."""
        period = self.size // other.size
        if index < other.size:
            return index * period
        i = index - other.size
        x = period - 1
        return i + (i // x) + 1

    # -- bivariate Lagrange helpers (ark-marlin ahp/mod.rs UnnormalizedBivariateLagrangePoly)
    def eval_unnormalized_bivariate_lagrange_poly(self, x, y):
        if x != y:
            return (self.vanishing(x) - self.vanishing(y)) * fr_inv((x - y) % R) % R
        return self.size * pow(x, self.size - 1, R) % R

    def batch_eval_unnormalized_bivariate_lagrange_poly_with_diff_inputs(self, x):
        vx = self.vanishing(x)
        return [vx * fr_inv((x - y) % R) % R for y in self.elements()]

    def batch_eval_unnormalized_bivariate_lagrange_poly_with_same_inputs(self):
        elems = [e * self.size % R for e in self.elements()]
        return [elems[0]] + elems[1:][::-1]


def poly_trim(p):
    p = list(p)
    while p and p[-1] % R == 0:
        p.pop()
    return p


def poly_degree(p):
    p = poly_trim(p)
    return max(len(p) - 1, 0)


def poly_eval(p, x):
    acc = 0
    for c in reversed(p):
        acc = (acc * x + c) % R
    return acc


def poly_add(a, b):
    n = max(len(a), len(b))
    return poly_trim([((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % R for i in range(n)])


def poly_sub(a, b):
    n = max(len(a), len(b))
    return poly_trim([((a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0)) % R for i in range(n)])


def poly_scale(a, k):
    return poly_trim([c * k % R for c in a])


def poly_mul(a, b):
    """DensePolynomial * DensePolynomial: FFT on next_pow2(len_a + len_b - 1)."""
    a = poly_trim(a)
    b = poly_trim(b)
    if not a or not b:
        return []
    d = Domain(len(a) + len(b) - 1)
    ea = d.fft(a)
    eb = d.fft(b)
    return poly_trim(d.ifft([x * y % R for x, y in zip(ea, eb)]))


def poly_mul_by_vanishing(p, domain):
    """p * (X^n - 1)."""
    p = poly_trim(p)
    n = domain.size
    out = [0] * (len(p) + n)
    for i, c in enumerate(p):
        out[i] = (out[i] - c) % R
        out[i + n] = (out[i + n] + c) % R
    return poly_trim(out)


def poly_divide_by_vanishing(p, domain):
    """DensePolynomial::divide_by_vanishing_poly -> (quotient, remainder)."""
    p = poly_trim(p)
    n = domain.size
    if len(p) < n + 1 and poly_degree(p) < n:
        return [], p
    # quotient[j] = sum_{i >= 1} p[j + i n] (ark-poly adds the shifted tails one by one: len(p)^2 / n steps, which for the
    # division of w by v_X, |X| = 4, is quadratic in |H|); the same sums from the top down in one pass
    quotient = list(p[n:])
    for j in range(len(quotient) - n - 1, -1, -1):
        quotient[j] = (quotient[j] + quotient[j + n]) % R
    remainder = list(p[:n])
    for j in range(min(n, len(quotient))):
        remainder[j] = (remainder[j] + quotient[j]) % R
    return poly_trim(quotient), poly_trim(remainder)


def poly_div_linear(p, z):
    """Quotient of p / (X - z) (remainder dropped), as ark-poly's generic long division gives."""
    p = poly_trim(p)
    if len(p) <= 1:
        return []
    q = [0] * (len(p) - 1)
    carry = 0
    for i in range(len(p) - 1, 0, -1):
        carry = (p[i] + carry * z) % R
        q[i - 1] = carry
    return poly_trim(q)


def batch_inverse(v):
    """ark_ff::batch_inversion semantics: zeros stay zero."""
    out = list(v)
    prod = 1
    pref = []
    for x in out:
        if x % R != 0:
            prod = prod * x % R
        pref.append(prod)
    inv = fr_inv(prod)
    for i in range(len(out) - 1, -1, -1):
        if out[i] % R == 0:
            continue
        prev = 1
        k = i - 1
        while k >= 0 and out[k] % R == 0:
            k -= 1
        prev = pref[k] if k >= 0 else 1
        x = out[i]
        out[i] = inv * prev % R
        inv = inv * x % R
    return out
