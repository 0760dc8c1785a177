"""Pure-Python restatement of the Marlin instance behind /root/reference/src/marlin/mod.rs
(oracle; TEST INFRASTRUCTURE ONLY — small circuits, pure big-int loops).

    MarlinInst = Marlin<Fr, MarlinKZG10<Bls12_377, DensePolynomial<Fr>>,
                        SimpleHashFiatShamirRng<Blake2s, ChaChaRng>>        (src/marlin/mod.rs:12-14)

Mirrors, function for function, the reference surface:
    generate_rand                         src/marlin/mod.rs:33-35
    generate_universal_srs                src/marlin/mod.rs:45-55   -> universal_setup
    generate_proving_and_verifying_keys   src/marlin/mod.rs:88-94   -> index
    generate_proof                        src/marlin/mod.rs:70-77   -> prove
    verify_proof                          src/marlin/mod.rs:79-86   -> verify
    serialize_proof / serialize_verifying_key   src/marlin/serialization.rs:5-31

The arithmetic lives in ark-marlin (git fork, branch use-constraint-system-directly, Cargo.toml:30),
ark-poly-commit 0.3, ark-poly 0.3, ark-ec 0.3, none of which is vendored in /root/reference; this file
restates their published 0.3.0 algorithms as recorded in SURVEY.md Appendix A [U].
**Parity unpinned** vs arkworks byte-for-byte (no golden proof exists in the reference's tests,
SURVEY.md §8c); what IS pinned: the proof this prover emits is accepted by the independent
pairing-based verifier below, tampered proofs are rejected, and the product (C++/HIP) must emit
the same bytes as this file on the same circuit and seeds.
"""
from . import bls12_377 as bls
from .bls12_377 import R, Q, Fq2
from .poly import (Domain, poly_trim, poly_degree, poly_eval, poly_add, poly_sub, poly_scale, poly_mul,
                   poly_mul_by_vanishing, poly_divide_by_vanishing, poly_div_linear, batch_inverse)
from .rng import test_rng, FiatShamirRng

PROTOCOL_NAME = b"MARLIN-2019"
INDEXER_POLYNOMIALS = ["a_row", "a_col", "a_val", "a_row_col", "b_row", "b_col", "b_val", "b_row_col",
                       "c_row", "c_col", "c_val", "c_row_col"]
PROVER_POLYNOMIALS = ["w", "z_a", "z_b", "mask_poly", "t", "g_1", "h_1", "g_2", "h_2"]
LC_WITH_ZERO_EVAL = ["inner_sumcheck", "outer_sumcheck"]
ZK_BOUND = 1


class SynthesisError(Exception):
    pass


class MarlinError(Exception):
    pass


# ----------------------------------------------------------------------------- constraint system
class ConstraintSystem:
    """Mirror of ark_relations::r1cs::ConstraintSystem as used through ConstraintSystemRef
    (src/marlin/mod.rs:16): instance variable 0 is the constant one; a linear combination is a list
    of (coeff, variable) with variable = ('i', k) or ('w', k)."""

    def __init__(self):
        self.instance = [1]
        self.witness = []
        self.a = []
        self.b = []
        self.c = []

    def new_input_variable(self, value):
        self.instance.append(value % R)
        return ("i", len(self.instance) - 1)

    def new_witness_variable(self, value):
        self.witness.append(value % R)
        return ("w", len(self.witness) - 1)

    @staticmethod
    def one():
        return ("i", 0)

    def enforce_constraint(self, a, b, c):
        self.a.append(list(a))
        self.b.append(list(b))
        self.c.append(list(c))

    @property
    def num_constraints(self):
        return len(self.a)

    def _col(self, var):
        kind, k = var
        return k if kind == "i" else len(self.instance) + k

    def to_matrices(self):
        """Row-sparse (coeff, column) with instance columns first."""
        def conv(rows):
            out = []
            for row in rows:
                acc = {}
                for coeff, var in row:
                    col = self._col(var)
                    acc[col] = (acc.get(col, 0) + coeff) % R
                out.append([(acc[c], c) for c in sorted(acc) if acc[c] != 0])  # LCs are kept sorted by variable
            return out
        return conv(self.a), conv(self.b), conv(self.c)

    def assignment(self):
        return list(self.instance) + list(self.witness)

    def is_satisfied(self):
        z = self.assignment()
        a, b, c = self.to_matrices()
        for ra, rb, rc in zip(a, b, c):
            ea = sum(v * z[col] for v, col in ra) % R
            eb = sum(v * z[col] for v, col in rb) % R
            ec = sum(v * z[col] for v, col in rc) % R
            if ea * eb % R != ec:
                return False
        return True

    def copy(self):
        o = ConstraintSystem()
        o.instance = list(self.instance)
        o.witness = list(self.witness)
        o.a = [list(r) for r in self.a]
        o.b = [list(r) for r in self.b]
        o.c = [list(r) for r in self.c]
        return o


def pad_and_square(cs):
    """ark-marlin constraint_systems.rs: pad_input_for_indexer_and_prover + make_matrices_square."""
    cs = cs.copy()
    dx = Domain(len(cs.instance))
    while len(cs.instance) < dx.size:
        cs.new_input_variable(0)
    nvars = len(cs.instance) + len(cs.witness)
    ncons = cs.num_constraints
    if nvars > ncons:
        for _ in range(nvars - ncons):
            cs.enforce_constraint([], [], [])
    else:
        for _ in range(ncons - nvars):
            cs.new_witness_variable(1)
    assert len(cs.instance) + len(cs.witness) == cs.num_constraints
    return cs


def balance_matrices(a, b):
    a_density = sum(len(r) for r in a)
    b_density = sum(len(r) for r in b)
    max_density = max(a_density, b_density)
    a_is_denser = a_density == max_density
    for i in range(len(a)):
        if a_is_denser:
            la, lb = len(a[i]), len(b[i])
            a[i], b[i] = b[i], a[i]
            a_density = a_density - la + lb
            b_density = b_density - lb + la
            max_density = max(a_density, b_density)
            a_is_denser = a_density == max_density


# ----------------------------------------------------------------------------- ToBytes / CanonicalSerialize
def tb_fr(v):
    return (v % R).to_bytes(32, "little")


def tb_fq(v):
    return (v % Q).to_bytes(48, "little")


def tb_g1(P):
    """ark-ec SW affine ToBytes: x || y || infinity  (97 B); zero point = (0, 1, true)."""
    if P is None:
        return tb_fq(0) + tb_fq(1) + b"\x01"
    return tb_fq(P[0]) + tb_fq(P[1]) + b"\x00"


def tb_commitment(comm):
    """marlin_pc::Commitment ToBytes: comm || bool(shifted present) || shifted-or-empty."""
    c, s = comm
    return tb_g1(c) + (b"\x01" if s is not None else b"\x00") + tb_g1(s if s is not None else None)


def ser_u64(v):
    return int(v).to_bytes(8, "little")


def ser_fr(v):
    return tb_fr(v)


def ser_g1(P):
    """ark-serialize compressed SW point."""
    if P is None:
        b = bytearray(48)
        b[47] |= 0x40
        return bytes(b)
    x, y = P
    b = bytearray(tb_fq(x))
    if y > (Q - y) % Q:
        b[47] |= 0x80
    return bytes(b)


def deser_g1(b):
    """CanonicalDeserialize::deserialize for G1Affine (ark-ec 0.3, the CHECKED form): flags 0xC0 invalid, x a field
    element, on the curve, in the prime-order subgroup."""
    b = bytearray(b)
    flags = b[47] & 0xC0
    b[47] &= 0x3F
    if flags == 0xC0:
        raise MarlinError("invalid G1 flags")
    x = int.from_bytes(b, "little")
    if x >= Q:
        raise MarlinError("invalid G1 encoding")
    if flags & 0x40:
        return None
    y = bls.fq_sqrt((x * x * x + 1) % Q)
    if y is None:
        raise MarlinError("G1 x not on curve")
    neg = (Q - y) % Q
    big, small = (y, neg) if y > neg else (neg, y)
    P = (x, big if flags & 0x80 else small)
    if bls.g1_mul_fast(P, R) is not None:
        raise MarlinError("G1 point not in the prime-order subgroup")
    return P


def ser_g2(P):
    if P is None:
        b = bytearray(96)
        b[95] |= 0x40
        return bytes(b)
    x, y = P
    b = bytearray(tb_fq(x.c0) + tb_fq(x.c1))
    if (-y).ark_lt(y):
        b[95] |= 0x80
    return bytes(b)


def ser_commitment(comm):
    c, s = comm
    out = ser_g1(c)
    out += b"\x01" + ser_g1(s) if s is not None else b"\x00"
    return out


# ----------------------------------------------------------------------------- KZG10 / MarlinKZG10
class UniversalSRS:
    def __init__(self, powers_of_g, powers_of_gamma_g, h, beta_h):
        self.powers_of_g = powers_of_g
        self.powers_of_gamma_g = powers_of_gamma_g
        self.h = h
        self.beta_h = beta_h

    @property
    def max_degree(self):
        return len(self.powers_of_g) - 1


def _g1_rand(rng):
    """ark-ec GroupProjective rand: x = Fq::rand; greatest = bool; get_point_from_x; scale by cofactor."""
    while True:
        x = rng.rand_fq()
        greatest = rng.gen_bool()
        y = bls.fq_sqrt((x * x * x + 1) % Q)
        if y is None:
            continue
        neg = (Q - y) % Q
        # ark: let y = if (y < negy) ^ greatest { y } else { negy }
        y = y if ((y < neg) != greatest) else neg
        return bls.g1_mul_fast((x, y), bls.G1_COFACTOR)


def _g2_rand(rng):
    while True:
        x = rng.rand_fq2()
        greatest = rng.gen_bool()
        y = (x * x * x + bls.G2_B).sqrt()
        if y is None:
            continue
        neg = -y
        y = y if (y.ark_lt(neg) != greatest) else neg
        return bls.g2_mul((x, y), bls.G2_COFACTOR)


def ahp_max_degree(num_constraints, num_variables, num_non_zero):
    dim = max(num_variables, num_constraints)
    h = Domain(dim).size
    k = Domain(num_non_zero).size
    return max(2 * h + ZK_BOUND - 2, 3 * h + 2 * ZK_BOUND - 3, h, h, 3 * k - 3)


def universal_setup(num_constraints, num_variables, num_non_zero, rng):
    """KZG10::setup(max_degree, false, rng)."""
    max_degree = ahp_max_degree(num_constraints, num_variables, num_non_zero)
    if max_degree < 1:
        raise MarlinError("DegreeIsZero")
    beta = rng.rand_fr()
    g = _g1_rand(rng)
    gamma_g = _g1_rand(rng)
    h = _g2_rand(rng)
    powers_of_beta = [1]
    for _ in range(max_degree):
        powers_of_beta.append(powers_of_beta[-1] * beta % R)
    powers_of_g = [bls.g1_mul_fast(g, p) for p in powers_of_beta]
    powers_of_gamma_g = [bls.g1_mul_fast(gamma_g, p) for p in powers_of_beta]
    powers_of_gamma_g.append(bls.g1_mul_fast(powers_of_gamma_g[-1], beta))
    beta_h = bls.g2_mul(h, beta)
    return UniversalSRS(powers_of_g, powers_of_gamma_g, h, beta_h)


class CommitterKey:
    pass


def trim(srs, supported_degree, supported_hiding_bound, enforced_degree_bounds):
    if supported_degree > srs.max_degree:
        raise MarlinError("TrimmingDegreeTooLarge")
    ck = CommitterKey()
    ck.powers = srs.powers_of_g[: supported_degree + 1]
    ck.powers_of_gamma_g = srs.powers_of_gamma_g[: supported_hiding_bound + 2]
    ck.max_degree = srs.max_degree
    ck.supported_degree = supported_degree
    bounds = sorted(set(enforced_degree_bounds))
    ck.enforced_degree_bounds = bounds
    lowest_shifted_power = srs.max_degree - bounds[-1]
    ck.shifted_powers = srs.powers_of_g[lowest_shifted_power:]
    vk = {
        "g": srs.powers_of_g[0], "gamma_g": srs.powers_of_gamma_g[0], "h": srs.h, "beta_h": srs.beta_h,
        "degree_bounds_and_shift_powers": [(d, srs.powers_of_g[srs.max_degree - d]) for d in bounds],
        "max_degree": srs.max_degree, "supported_degree": supported_degree,
    }
    return ck, vk


def ck_shifted_powers(ck, degree_bound):
    if degree_bound is None:
        return ck.shifted_powers
    assert degree_bound in ck.enforced_degree_bounds
    return ck.shifted_powers[ck.enforced_degree_bounds[-1] - degree_bound:]


def _msm(bases, scalars):
    return bls.g1_msm_naive(bases[: len(scalars)], scalars)


def kzg_commit(powers_of_g, powers_of_gamma_g, poly, hiding_bound, rng):
    """kzg10::KZG10::commit -> (G1 affine, blinding polynomial)."""
    poly = poly_trim(poly)
    if poly and len(poly) - 1 >= len(powers_of_g):
        raise MarlinError("TooManyCoefficients")
    nlz = 0
    while nlz < len(poly) and poly[nlz] == 0:
        nlz += 1
    commitment = _msm(powers_of_g[nlz:], poly[nlz:])
    blinding = []
    if hiding_bound is not None:
        blinding = poly_trim([rng.rand_fr() for _ in range(hiding_bound + 2)])
        if poly_degree(blinding) + 1 > len(powers_of_gamma_g) and blinding:
            raise MarlinError("HidingBoundToolarge")
    random_commitment = _msm(powers_of_gamma_g, blinding)
    return bls.g1_add(commitment, random_commitment), blinding


def pc_commit(ck, labeled_polys, rng):
    """MarlinKZG10::commit. labeled_polys: list of (label, coeffs, degree_bound, hiding_bound).
    Returns commitments [(comm, shifted_comm)] and randomness [(rand, shifted_rand)]."""
    comms, rands = [], []
    for label, poly, degree_bound, hiding_bound in labeled_polys:
        comm, rand = kzg_commit(ck.powers, ck.powers_of_gamma_g, poly, hiding_bound, rng)
        if degree_bound is not None:
            sp = ck_shifted_powers(ck, degree_bound)
            scomm, srand = kzg_commit(sp, ck.powers_of_gamma_g, poly, hiding_bound, rng)
        else:
            scomm, srand = None, None
        comms.append((comm, scomm))
        rands.append((rand, srand))
    return comms, rands


def _open_with_witness(powers_of_g, powers_of_gamma_g, point, blinding, witness, hiding_witness):
    witness = poly_trim(witness)
    nlz = 0
    while nlz < len(witness) and witness[nlz] == 0:
        nlz += 1
    w = _msm(powers_of_g[nlz:], witness[nlz:])
    random_v = None
    if hiding_witness is not None:
        random_v = poly_eval(blinding, point)
        w = bls.g1_add(w, _msm(powers_of_gamma_g, poly_trim(hiding_witness)))
    return w, random_v


def pc_open(ck, polys, point, xi, rands):
    """MarlinKZG10::open_individual_opening_challenges with challenges xi^0, xi^1, ...
    polys: list of (label, coeffs, degree_bound, hiding_bound); rands: [(rand, shifted_rand)]."""
    p, r = [], []
    shifted_w, shifted_r, shifted_r_witness = [], [], []
    enforce = False
    ctr = 0
    for (label, poly, degree_bound, _hb), (rand, shifted_rand) in zip(polys, rands):
        ch = pow(xi, ctr, R)
        ctr += 1
        p = poly_add(p, poly_scale(poly, ch))
        r = poly_add(r, poly_scale(rand, ch))
        if degree_bound is not None:
            enforce = True
            witness = poly_div_linear(poly, point)
            srw = poly_div_linear(shifted_rand, point) if poly_trim(shifted_rand) else None
            ch1 = pow(xi, ctr, R)
            ctr += 1
            if witness:
                shifted_witness = [0] * (ck.enforced_degree_bounds[-1] - degree_bound) + witness
            else:
                shifted_witness = []
            shifted_w = poly_add(shifted_w, poly_scale(shifted_witness, ch1))
            shifted_r = poly_add(shifted_r, poly_scale(shifted_rand, ch1))
            if srw is not None:
                shifted_r_witness = poly_add(shifted_r_witness, poly_scale(srw, ch1))
    witness = poly_div_linear(p, point)
    hiding_witness = poly_div_linear(r, point) if poly_trim(r) else None
    w, random_v = _open_with_witness(ck.powers, ck.powers_of_gamma_g, point, r, witness, hiding_witness)
    if enforce:
        sw, srv = _open_with_witness(ck_shifted_powers(ck, None), ck.powers_of_gamma_g, point, shifted_r,
                                     shifted_w, shifted_r_witness)
        w = bls.g1_add(w, sw)
        if srv is not None and random_v is not None:
            random_v = (random_v + srv) % R
    return w, random_v


# ----------------------------------------------------------------------------- indexer
class Index:
    pass


def _arithmetize(matrix, domain_k, domain_h, domain_x, domain_b):
    """ark-marlin 0.3.0 arithmetize_matrix (SURVEY A.7 'Indexer'): arithmetises M* (transpose, scaled)."""
    elems = domain_h.elements()
    eq_vals = dict(zip(elems, domain_h.batch_eval_unnormalized_bivariate_lagrange_poly_with_same_inputs()))
    row_vec, col_vec, val_vec, inverses = [], [], [], []
    for r, row in enumerate(matrix):
        row = sorted(row, key=lambda t: t[1])
        for val, i in row:
            row_val = elems[r]
            col_val = elems[domain_h.reindex_by_subdomain(domain_x, i)]
            row_vec.append(col_val)
            col_vec.append(row_val)
            val_vec.append(val)
            inverses.append(eq_vals[col_val])
    inverses = batch_inverse(inverses)
    val_vec = [v * i % R for v, i in zip(val_vec, inverses)]
    while len(row_vec) < domain_k.size:
        col_vec.append(elems[0])
        row_vec.append(elems[0])
        val_vec.append(0)
    row_col_vec = [a * b % R for a, b in zip(row_vec, col_vec)]
    ar = {}
    for name, vec in (("row", row_vec), ("col", col_vec), ("val", val_vec), ("row_col", row_col_vec)):
        poly = poly_trim(domain_k.ifft(vec))
        ar[name] = poly
        ar[name + "_K"] = vec
        ar[name + "_B"] = domain_b.fft(poly)
    return ar


def index(srs, cs):
    """Marlin::index_from_constraint_system (fork) / Marlin::index (upstream) [U]."""
    ics = pad_and_square(cs)
    a, b, c = ics.to_matrices()
    num_non_zero = max(sum(len(r) for r in m) for m in (a, b, c))
    balance_matrices(a, b)
    num_constraints = ics.num_constraints
    num_variables = len(ics.instance) + len(ics.witness)
    if num_constraints != num_variables:
        raise MarlinError("NonSquareMatrix")
    idx = Index()
    idx.num_variables = num_variables
    idx.num_constraints = num_constraints
    idx.num_non_zero = num_non_zero
    idx.num_instance_variables = len(ics.instance)
    idx.a, idx.b, idx.c = a, b, c
    dh, dk, dx = Domain(num_constraints), Domain(num_non_zero), Domain(len(ics.instance))
    db = Domain(3 * dk.size - 3)
    idx.arith = {n: _arithmetize(m, dk, dh, dx, db) for n, m in (("a", a), ("b", b), ("c", c))}
    max_deg = ahp_max_degree(num_constraints, num_variables, num_non_zero)
    if srs.max_degree < max_deg:
        raise MarlinError("IndexTooLarge")
    ck, pcvk = trim(srs, max_deg, 1, [dh.size - 2, dk.size - 2])
    polys = [(m + "_" + n, idx.arith[m][n], None, None) for m in "abc" for n in ("row", "col", "val", "row_col")]
    comms, rands = pc_commit(ck, polys, None)
    vk = {"num_variables": num_variables, "num_constraints": num_constraints, "num_non_zero": num_non_zero,
          "num_instance_variables": idx.num_instance_variables, "index_comms": comms, "verifier_key": pcvk}
    pk = {"index": idx, "index_comm_rands": rands, "vk": vk, "ck": ck}
    return pk, vk


def tb_index_vk(vk):
    out = ser_u64(vk["num_variables"]) + ser_u64(vk["num_constraints"]) + ser_u64(vk["num_non_zero"])
    for c in vk["index_comms"]:
        out += tb_commitment(c)
    return out


# ----------------------------------------------------------------------------- AHP verifier messages
def _sample_outside(domain, fs):
    t = fs.rand_fr()
    while domain.vanishing(t) == 0:
        t = fs.rand_fr()
    return t


def lc_eval_by_terms(poly_eval_fn):
    """EvaluationsProvider for a set of polynomials (prover side): evaluate the LC term by term."""
    def provider(label, lc, point):
        acc = 0
        for coeff, term in lc:
            acc += coeff * (1 if term is None else poly_eval_fn(term, point))
        return acc % R
    return provider


def construct_linear_combinations(info, public_input, lc_eval, st):
    """AHPForR1CS::construct_linear_combinations (0.3.0).  lc_eval(label, lc, point) -> Fr is the
    EvaluationsProvider: the prover evaluates term by term, the verifier looks the LC label up in
    the proof's evaluations.  Returns a label-sorted list of (label, [(coeff, poly-label-or-None)])."""
    def _lc_eval(lc_label_and_terms, _unused, point):
        return lc_eval(lc_label_and_terms[0], lc_label_and_terms[1], point)
    dh, dk = Domain(info["num_constraints"]), Domain(info["num_non_zero"])
    alpha, eta_a, eta_b, eta_c, beta, gamma = (st[k] for k in ("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma"))
    dx = Domain(len(public_input) + 1)
    x_poly = poly_trim(dx.ifft([1] + list(public_input)))
    lcs = {}
    lcs["z_b"] = [(1, "z_b")]
    lcs["g_1"] = [(1, "g_1")]
    lcs["t"] = [(1, "t")]
    r_alpha_at_beta = dh.eval_unnormalized_bivariate_lagrange_poly(alpha, beta)
    v_H_at_alpha = dh.vanishing(alpha)
    v_H_at_beta = dh.vanishing(beta)
    v_X_at_beta = dx.vanishing(beta)
    z_b_at_beta = _lc_eval(("z_b", lcs["z_b"]), None, beta)
    t_at_beta = _lc_eval(("t", lcs["t"]), None, beta)
    g_1_at_beta = _lc_eval(("g_1", lcs["g_1"]), None, beta)
    x_at_beta = poly_eval(x_poly, beta)
    lcs["outer_sumcheck"] = [
        (1, "mask_poly"),
        (r_alpha_at_beta * (eta_a + eta_c * z_b_at_beta) % R, "z_a"),
        (r_alpha_at_beta * eta_b * z_b_at_beta % R, None),
        (-t_at_beta * v_X_at_beta % R, "w"),
        (-t_at_beta * x_at_beta % R, None),
        (-v_H_at_beta % R, "h_1"),
        (-beta * g_1_at_beta % R, None),
    ]
    beta_alpha = beta * alpha % R
    lcs["g_2"] = [(1, "g_2")]
    for m in "abc":
        lcs[m + "_denom"] = [(beta_alpha, None), (-alpha % R, m + "_row"), (-beta % R, m + "_col"), (1, m + "_row_col")]
    a_d = _lc_eval(("a_denom", lcs["a_denom"]), None, gamma)
    b_d = _lc_eval(("b_denom", lcs["b_denom"]), None, gamma)
    c_d = _lc_eval(("c_denom", lcs["c_denom"]), None, gamma)
    g_2_at_gamma = _lc_eval(("g_2", lcs["g_2"]), None, gamma)
    v_K_at_gamma = dk.vanishing(gamma)
    scale = v_H_at_alpha * v_H_at_beta % R
    inner = [
        (eta_a * b_d * c_d % R * scale % R, "a_val"),
        (eta_b * a_d * c_d % R * scale % R, "b_val"),
        (eta_c * b_d * a_d % R * scale % R, "c_val"),
    ]
    b_at_gamma = a_d * b_d * c_d % R
    b_expr = b_at_gamma * ((gamma * g_2_at_gamma + t_at_beta * pow(dk.size, -1, R)) % R) % R
    inner.append((-b_expr % R, None))
    inner.append((-v_K_at_gamma % R, "h_2"))
    lcs["inner_sumcheck"] = inner
    return sorted(lcs.items())


QUERY_SET = [("g_1", "beta"), ("z_b", "beta"), ("t", "beta"), ("outer_sumcheck", "beta"),
             ("g_2", "gamma"), ("a_denom", "gamma"), ("b_denom", "gamma"), ("c_denom", "gamma"),
             ("inner_sumcheck", "gamma")]


def _fs_init(vk, public_input):
    data = PROTOCOL_NAME + tb_index_vk(vk) + b"".join(tb_fr(x) for x in public_input)
    return FiatShamirRng(data)


# ----------------------------------------------------------------------------- prover
def prove(pk, cs, zk_rng, trace=None):
    """Marlin::prove_from_constraint_system (src/marlin/mod.rs:75) following SURVEY A.6/A.7 [U].
    `trace`, if a dict, receives intermediate values for parity debugging of the product."""
    idx = pk["index"]
    ck = pk["ck"]
    vk = pk["vk"]
    pcs = pad_and_square(cs)
    formatted_input = list(pcs.instance)
    witness = list(pcs.witness)
    if idx.num_constraints != pcs.num_constraints or len(formatted_input) + len(witness) != idx.num_variables:
        raise MarlinError("InstanceDoesNotMatchIndex")
    z = formatted_input + witness
    z_a = [sum(v * z[c] for v, c in row) % R for row in idx.a]
    z_b = [sum(v * z[c] for v, c in row) % R for row in idx.b]
    dh, dk, dx = Domain(idx.num_constraints), Domain(idx.num_non_zero), Domain(len(formatted_input))
    H, K = dh.size, dk.size
    public_input = formatted_input[1:]
    fs = _fs_init(vk, public_input)

    # ---- round 1
    x_poly = poly_trim(dx.ifft(formatted_input))
    x_evals = dh.fft(x_poly)
    ratio = H // dx.size
    w_extended = witness + [0] * (H - dx.size - len(witness))
    w_evals = [0 if k % ratio == 0 else (w_extended[k - k // ratio - 1] - x_evals[k]) % R for k in range(H)]
    rho_w = zk_rng.rand_fr()
    w_poly = poly_add(dh.ifft(w_evals), poly_mul_by_vanishing([rho_w], dh))
    w_poly, rem = poly_divide_by_vanishing(w_poly, dx)
    assert not rem
    rho_a = zk_rng.rand_fr()
    z_a_poly = poly_add(dh.ifft(z_a), poly_mul_by_vanishing([rho_a], dh))
    rho_b = zk_rng.rand_fr()
    z_b_poly = poly_add(dh.ifft(z_b), poly_mul_by_vanishing([rho_b], dh))
    mask_degree = 3 * H + 2 * ZK_BOUND - 3
    mask_poly = poly_trim([zk_rng.rand_fr() for _ in range(mask_degree + 1)])
    sigma = poly_divide_by_vanishing(mask_poly, dh)[1]
    mask_poly[0] = (mask_poly[0] - (sigma[0] if sigma else 0)) % R
    oracles1 = [("w", w_poly, None, 1), ("z_a", z_a_poly, None, 1), ("z_b", z_b_poly, None, 1),
                ("mask_poly", mask_poly, None, None)]
    comms1, rands1 = pc_commit(ck, oracles1, zk_rng)
    fs.absorb(b"".join(tb_commitment(c) for c in comms1))
    alpha = _sample_outside(dh, fs)
    eta_a, eta_b, eta_c = fs.rand_fr(), fs.rand_fr(), fs.rand_fr()

    # ---- round 2
    z_c_poly = poly_mul(z_a_poly, z_b_poly)
    summed = [c * eta_c % R for c in z_c_poly]
    for i in range(min(len(summed), len(z_a_poly), len(z_b_poly))):
        summed[i] = (summed[i] + eta_a * z_a_poly[i] + eta_b * z_b_poly[i]) % R
    summed = poly_trim(summed)
    r_alpha_evals = dh.batch_eval_unnormalized_bivariate_lagrange_poly_with_diff_inputs(alpha)
    r_alpha_poly = poly_trim(dh.ifft(r_alpha_evals))
    t_evals = [0] * H
    for matrix, eta in ((idx.a, eta_a), (idx.b, eta_b), (idx.c, eta_c)):
        for r, row in enumerate(matrix):
            for coeff, c in row:
                k = dh.reindex_by_subdomain(dx, c)
                t_evals[k] = (t_evals[k] + eta * coeff % R * r_alpha_evals[r]) % R
    t_poly = poly_trim(dh.ifft(t_evals))
    z_poly = poly_mul_by_vanishing(w_poly, dx)
    z_poly = z_poly + [0] * (len(x_poly) - len(z_poly))
    for i, xc in enumerate(x_poly):
        z_poly[i] = (z_poly[i] + xc) % R
    z_poly = poly_trim(z_poly)
    mul_size = max(len(mask_poly), len(r_alpha_poly) + len(summed), len(t_poly) + len(z_poly))
    dm = Domain(mul_size)
    ra, sm, zp, tp = dm.fft(r_alpha_poly), dm.fft(summed), dm.fft(z_poly), dm.fft(t_poly)
    rhs = poly_trim(dm.ifft([(a * b - c * d) % R for a, b, c, d in zip(ra, sm, zp, tp)]))
    q_1 = poly_add(mask_poly, rhs)
    h_1, x_g_1 = poly_divide_by_vanishing(q_1, dh)
    if x_g_1 and x_g_1[0] != 0:
        # arkworks: debug_assert in construct_linear_combinations fires (unsatisfied witness)
        raise MarlinError("outer sumcheck does not hold: constraint system is not satisfied")
    g_1 = poly_trim(x_g_1[1:])
    assert poly_degree(g_1) <= H - 2
    oracles2 = [("t", t_poly, None, None), ("g_1", g_1, H - 2, 1), ("h_1", h_1, None, None)]
    comms2, rands2 = pc_commit(ck, oracles2, zk_rng)
    fs.absorb(b"".join(tb_commitment(c) for c in comms2))
    beta = _sample_outside(dh, fs)

    # ---- round 3
    v_H_at_alpha, v_H_at_beta = dh.vanishing(alpha), dh.vanishing(beta)
    ar = idx.arith
    inv = {}
    for m in "abc":
        inv[m] = batch_inverse([(beta - ar[m]["row_K"][i]) * (alpha - ar[m]["col_K"][i]) % R for i in range(K)])
    etas = {"a": eta_a, "b": eta_b, "c": eta_c}
    f_vals = []
    for i in range(K):
        t = sum(etas[m] * ar[m]["val_K"][i] % R * inv[m][i] for m in "abc") % R
        f_vals.append(v_H_at_alpha * v_H_at_beta % R * t % R)
    f = poly_trim(dk.ifft(f_vals))
    g_2 = poly_trim(f[1:])
    db = Domain(3 * K - 3)
    B = db.size
    den = {}
    for m in "abc":
        den[m] = [(beta * alpha - ar[m]["row_B"][i] * alpha - beta * ar[m]["col_B"][i] + ar[m]["row_col_B"][i]) % R
                  for i in range(B)]
    a_on_B = []
    for i in range(B):
        t = (eta_a * ar["a"]["val_B"][i] % R * den["b"][i] % R * den["c"][i]
             + eta_b * ar["b"]["val_B"][i] % R * den["a"][i] % R * den["c"][i]
             + eta_c * ar["c"]["val_B"][i] % R * den["a"][i] % R * den["b"][i]) % R
        a_on_B.append(v_H_at_beta * v_H_at_alpha % R * t % R)
    a_poly = poly_trim(db.ifft(a_on_B))
    b_poly = poly_trim(db.ifft([den["a"][i] * den["b"][i] % R * den["c"][i] % R for i in range(B)]))
    h_2, rem2 = poly_divide_by_vanishing(poly_sub(a_poly, poly_mul(b_poly, f)), dk)
    assert poly_degree(g_2) <= K - 2
    oracles3 = [("g_2", g_2, K - 2, None), ("h_2", h_2, None, None)]
    comms3, rands3 = pc_commit(ck, oracles3, zk_rng)
    fs.absorb(b"".join(tb_commitment(c) for c in comms3))
    gamma = fs.rand_fr()

    # ---- evaluations + opening
    polys = {m + "_" + n: (ar[m][n], None, None) for m in "abc" for n in ("row", "col", "val", "row_col")}
    for label, p, db_, hb in oracles1 + oracles2 + oracles3:
        polys[label] = (p, db_, hb)
    rands = dict(zip(INDEXER_POLYNOMIALS, pk["index_comm_rands"]))
    rands.update(zip(["w", "z_a", "z_b", "mask_poly"], rands1))
    rands.update(zip(["t", "g_1", "h_1"], rands2))
    rands.update(zip(["g_2", "h_2"], rands3))
    st = {"alpha": alpha, "eta_a": eta_a, "eta_b": eta_b, "eta_c": eta_c, "beta": beta, "gamma": gamma}
    points = {"beta": beta, "gamma": gamma}
    info = {"num_constraints": idx.num_constraints, "num_non_zero": idx.num_non_zero}

    provider = lc_eval_by_terms(lambda label, point: poly_eval(polys[label][0], point))
    lcs = construct_linear_combinations(info, public_input, provider, st)
    lc_map = dict(lcs)
    evaluations = []
    for label, pl in QUERY_SET:
        if label in LC_WITH_ZERO_EVAL:
            if provider(label, lc_map[label], points[pl]) != 0:
                raise MarlinError(label + " does not evaluate to zero: constraint system is not satisfied")
            continue
        evaluations.append((label, provider(label, lc_map[label], points[pl])))
    evaluations.sort()
    evaluations = [e for _, e in evaluations]
    fs.absorb(b"".join(tb_fr(e) for e in evaluations))
    xi = fs.gen_u128() % R

    # MarlinKZG10::open_combinations: materialise LC polynomials + randomness, then batch-open per point
    lc_polys, lc_rands = {}, {}
    for label, lc in lcs:
        poly, rand = [], []
        degree_bound, hiding_bound = None, None
        srand = None
        terms = [(c, t) for c, t in lc if t is not None]
        for coeff, term in terms:
            p, db_, hb = polys[term]
            if len(lc) == 1 and db_ is not None:
                assert coeff == 1
                degree_bound = db_
                srand = rands[term][1]
            elif db_ is not None:
                raise MarlinError("EquationHasDegreeBounds")
            if hb is not None:
                hiding_bound = hb if hiding_bound is None else max(hiding_bound, hb)
            poly = poly_add(poly, poly_scale(p, coeff))
            rand = poly_add(rand, poly_scale(rands[term][0], coeff))
        lc_polys[label] = (label, poly, degree_bound, hiding_bound)
        lc_rands[label] = (rand, srand)
    pc_proofs = []
    for pl in ("beta", "gamma"):
        labels = sorted(l for l, p in QUERY_SET if p == pl)
        w, random_v = pc_open(ck, [lc_polys[l] for l in labels], points[pl], xi, [lc_rands[l] for l in labels])
        pc_proofs.append((w, random_v))
    proof = {"commitments": [comms1, comms2, comms3], "evaluations": evaluations, "pc_proof": pc_proofs}
    if trace is not None:
        trace.update(st)
        trace.update({"xi": xi, "z_a": z_a, "z_b": z_b, "w_poly": w_poly, "z_a_poly": z_a_poly, "z_b_poly": z_b_poly,
                      "mask_poly": mask_poly, "t_poly": t_poly, "g_1": g_1, "h_1": h_1, "g_2": g_2, "h_2": h_2,
                      "f": f, "public_input": public_input})
    return proof


def serialize_proof(proof):
    """CanonicalSerialize of ark_marlin::Proof (src/marlin/serialization.rs:5-12)."""
    out = ser_u64(len(proof["commitments"]))
    for rnd in proof["commitments"]:
        out += ser_u64(len(rnd))
        for c in rnd:
            out += ser_commitment(c)
    out += ser_u64(len(proof["evaluations"]))
    for e in proof["evaluations"]:
        out += ser_fr(e)
    out += ser_u64(3) + b"\x00\x00\x00"  # prover_messages: 3 x EmptyMessage -> Option::None
    out += ser_u64(len(proof["pc_proof"]))
    for w, rv in proof["pc_proof"]:
        out += ser_g1(w)
        out += (b"\x01" + ser_fr(rv)) if rv is not None else b"\x00"
    out += b"\x00"  # BatchLCProof.evals = None
    return out


class _Reader:
    def __init__(self, b):
        self.b = bytes(b)
        self.p = 0

    def take(self, n):
        if self.p + n > len(self.b):
            raise MarlinError("Error deserializing proof: unexpected end of input")
        v = self.b[self.p:self.p + n]
        self.p += n
        return v

    def u64(self):
        return int.from_bytes(self.take(8), "little")

    def fr(self):
        v = int.from_bytes(self.take(32), "little")
        if v >= R:
            raise MarlinError("invalid Fr encoding")
        return v

    def boolean(self):
        v = self.take(1)[0]
        if v > 1:
            raise MarlinError("invalid bool")
        return v == 1


def deserialize_proof(data):
    rd = _Reader(data)
    commitments = []
    for _ in range(rd.u64()):
        rnd = []
        for _ in range(rd.u64()):
            c = deser_g1(rd.take(48))
            s = deser_g1(rd.take(48)) if rd.boolean() else None
            rnd.append((c, s))
        commitments.append(rnd)
    evaluations = [rd.fr() for _ in range(rd.u64())]
    for _ in range(rd.u64()):
        if rd.boolean():
            for _ in range(rd.u64()):
                rd.fr()
    pc = []
    for _ in range(rd.u64()):
        w = deser_g1(rd.take(48))
        rv = rd.fr() if rd.boolean() else None
        pc.append((w, rv))
    if rd.boolean():
        for _ in range(rd.u64()):
            rd.fr()
    return {"commitments": commitments, "evaluations": evaluations, "pc_proof": pc}


def serialize_verifying_key(vk):
    """CanonicalSerialize of IndexVerifierKey (src/marlin/serialization.rs:19-26) [U field order]."""
    out = ser_u64(vk["num_variables"]) + ser_u64(vk["num_constraints"]) + ser_u64(vk["num_non_zero"])
    out += ser_u64(vk["num_instance_variables"])
    out += ser_u64(len(vk["index_comms"]))
    for c in vk["index_comms"]:
        out += ser_commitment(c)
    pv = vk["verifier_key"]
    out += ser_g1(pv["g"]) + ser_g1(pv["gamma_g"]) + ser_g2(pv["h"]) + ser_g2(pv["beta_h"])
    out += b"\x01" + ser_u64(len(pv["degree_bounds_and_shift_powers"]))
    for d, p in pv["degree_bounds_and_shift_powers"]:
        out += ser_u64(d) + ser_g1(p)
    out += ser_u64(pv["max_degree"]) + ser_u64(pv["supported_degree"])
    return out


def ser_usize_opt(v):
    return b"\x00" if v is None else b"\x01" + ser_u64(v)


def ser_domain(d):
    """CanonicalSerialize of GeneralEvaluationDomain::Radix2 (ark-poly 0.3.0 [U]): a u8 variant tag, then the derived
    serialisation of Radix2EvaluationDomain { size: u64, log_size_of_group: u32, size_as_field_element, size_inv,
    group_gen, group_gen_inv, generator_inv }."""
    out = b"\x00" + ser_u64(d.size) + d.log.to_bytes(4, "little")
    out += ser_fr(d.size % R) + ser_fr(d.size_inv) + ser_fr(d.gen) + ser_fr(d.gen_inv) + ser_fr(pow(22, -1, R))
    return out


def ser_fr_vec(v):
    return ser_u64(len(v)) + b"".join(ser_fr(x) for x in v)


def ser_labeled_poly(label, coeffs, degree_bound=None, hiding_bound=None):
    """LabeledPolynomial { label: String, polynomial: Rc<DensePolynomial>, degree_bound, hiding_bound } [U]."""
    lb = label.encode()
    return ser_u64(len(lb)) + lb + ser_fr_vec(poly_trim(coeffs)) + ser_usize_opt(degree_bound) + ser_usize_opt(hiding_bound)


def ser_evals(vals, domain):
    """Evaluations { evals: Vec<F>, domain: GeneralEvaluationDomain } [U]."""
    return ser_fr_vec(vals) + ser_domain(domain)


def ser_matrix(m):
    """Matrix<F> = Vec<Vec<(F, usize)>>."""
    out = ser_u64(len(m))
    for row in m:
        out += ser_u64(len(row))
        for val, col in row:
            out += ser_fr(val) + ser_u64(col)
    return out


def serialize_proving_key(pk):
    """CanonicalSerialize of IndexProverKey (src/marlin/serialization.rs:33-39) — field order of ark-marlin 0.3.0's
    derives as recalled [U]:
        IndexProverKey { index_vk, index_comm_rands: Vec<marlin_pc::Randomness>, index: Index, committer_key }
        Index { index_info, a, b, c, a_star_arith, b_star_arith, c_star_arith }
        MatrixArithmetization { row, col, val, row_col: LabeledPolynomial, evals_on_K: MatrixEvals { row, col, val },
                                evals_on_B: MatrixEvals, row_col_evals_on_B: Evaluations }
        marlin_pc::CommitterKey { powers, shifted_powers: Option<Vec>, powers_of_gamma_g, enforced_degree_bounds: Option<Vec<usize>>, max_degree }
    """
    vk, idx, ck = pk["vk"], pk["index"], pk["ck"]
    out = serialize_verifying_key(vk)
    out += ser_u64(len(pk["index_comm_rands"]))
    for _ in pk["index_comm_rands"]:  # index polynomials are committed without hiding: empty blinding polynomial, no shifted rand
        out += ser_fr_vec([]) + b"\x00"
    out += ser_u64(idx.num_variables) + ser_u64(idx.num_constraints) + ser_u64(idx.num_non_zero) + ser_u64(idx.num_instance_variables)
    out += ser_matrix(idx.a) + ser_matrix(idx.b) + ser_matrix(idx.c)
    dk = Domain(idx.num_non_zero)
    db = Domain(3 * dk.size - 3)
    for m in "abc":
        ar = idx.arith[m]
        for n in ("row", "col", "val", "row_col"):
            out += ser_labeled_poly(m + "_" + n, ar[n])
        for dom, suf in ((dk, "_K"), (db, "_B")):
            for n in ("row", "col", "val"):
                out += ser_evals(ar[n + suf], dom)
        out += ser_evals(ar["row_col_B"], db)
    out += ser_u64(len(ck.powers)) + b"".join(ser_g1(p) for p in ck.powers)
    out += b"\x01" + ser_u64(len(ck.shifted_powers)) + b"".join(ser_g1(p) for p in ck.shifted_powers)
    out += ser_u64(len(ck.powers_of_gamma_g)) + b"".join(ser_g1(p) for p in ck.powers_of_gamma_g)
    out += b"\x01" + ser_u64(len(ck.enforced_degree_bounds)) + b"".join(ser_u64(d) for d in ck.enforced_degree_bounds)
    out += ser_u64(ck.max_degree)
    return out


# ----------------------------------------------------------------------------- verifier
def verify(vk, public_input, proof, rng):
    """Marlin::verify (src/marlin/mod.rs:79-86) + MarlinKZG10::check_combinations + KZG10::batch_check."""
    public_input = [x % R for x in public_input]
    dx = Domain(len(public_input) + 1)
    public_input = public_input + [0] * (max(len(public_input), dx.size - 1) - len(public_input))
    fs = _fs_init(vk, public_input)
    comms1, comms2, comms3 = proof["commitments"]
    dh, dk = Domain(vk["num_constraints"]), Domain(vk["num_non_zero"])
    fs.absorb(b"".join(tb_commitment(c) for c in comms1))
    alpha = _sample_outside(dh, fs)
    eta_a, eta_b, eta_c = fs.rand_fr(), fs.rand_fr(), fs.rand_fr()
    fs.absorb(b"".join(tb_commitment(c) for c in comms2))
    beta = _sample_outside(dh, fs)
    fs.absorb(b"".join(tb_commitment(c) for c in comms3))
    gamma = fs.rand_fr()
    st = {"alpha": alpha, "eta_a": eta_a, "eta_b": eta_b, "eta_c": eta_c, "beta": beta, "gamma": gamma}
    points = {"beta": beta, "gamma": gamma}
    degree_bounds = [None] * 12 + [None, None, None, None, None, dh.size - 2, None, dk.size - 2, None]
    all_comms = list(vk["index_comms"]) + list(comms1) + list(comms2) + list(comms3)
    labels = INDEXER_POLYNOMIALS + PROVER_POLYNOMIALS
    if len(all_comms) != len(labels):
        return False
    commitments = {l: (c, d) for l, c, d in zip(labels, all_comms, degree_bounds)}
    fs.absorb(b"".join(tb_fr(e) for e in proof["evaluations"]))
    xi = fs.gen_u128() % R
    evaluations = {}
    eval_labels = []
    for label, pl in QUERY_SET:
        if label in LC_WITH_ZERO_EVAL:
            evaluations[(label, pl)] = 0
        else:
            eval_labels.append((label, pl))
    eval_labels.sort()
    if len(eval_labels) != len(proof["evaluations"]):
        return False
    for q, e in zip(eval_labels, proof["evaluations"]):
        evaluations[q] = e
    pl_of = dict(QUERY_SET)

    def provider(label, lc, point):
        return evaluations[(label, pl_of[label])]

    info = {"num_constraints": vk["num_constraints"], "num_non_zero": vk["num_non_zero"]}
    lcs = construct_linear_combinations(info, public_input, provider, st)
    # check_combinations: build LC commitments, fold constants into the claimed evaluations
    lc_comms = {}
    for label, lc in lcs:
        comm, shifted, degree_bound = None, None, None
        for coeff, term in lc:
            if term is None:
                evaluations[(label, pl_of[label])] = (evaluations[(label, pl_of[label])] - coeff) % R
                continue
            (c, s), d = commitments[term]
            if len(lc) == 1 and d is not None:
                assert coeff == 1
                degree_bound = d
            elif d is not None:
                return False
            comm = bls.g1_add(comm, bls.g1_mul_fast(c, coeff))
            if s is not None:
                shifted = bls.g1_add(shifted, bls.g1_mul_fast(s, coeff))
        lc_comms[label] = ((comm, shifted if degree_bound is not None else None), degree_bound)
    pv = vk["verifier_key"]
    shift_power = dict(pv["degree_bounds_and_shift_powers"])
    combined = []
    if len(proof["pc_proof"]) != 2:
        return False
    for pl in ("beta", "gamma"):
        labels_here = sorted(l for l, p in QUERY_SET if p == pl)
        cc, cv = None, 0
        ctr = 0
        for l in labels_here:
            (c, s), d = lc_comms[l]
            v = evaluations[(l, pl)]
            ch = pow(xi, ctr, R)
            ctr += 1
            cc = bls.g1_add(cc, bls.g1_mul_fast(c, ch))
            cv = (cv + v * ch) % R
            if d is not None:
                ch1 = pow(xi, ctr, R)
                ctr += 1
                adj = bls.g1_add(s, bls.g1_neg(bls.g1_mul_fast(shift_power[d], v)))
                cc = bls.g1_add(cc, bls.g1_mul_fast(adj, ch1))
        combined.append((cc, points[pl], cv))
    # KZG10::batch_check
    total_c, total_w = None, None
    randomizer = 1
    g_mult, gamma_g_mult = 0, 0
    for (c, zpt, v), (w, random_v) in zip(combined, proof["pc_proof"]):
        tmp = bls.g1_add(bls.g1_mul_fast(w, zpt), c)
        g_mult = (g_mult + randomizer * v) % R
        if random_v is not None:
            gamma_g_mult = (gamma_g_mult + randomizer * random_v) % R
        total_c = bls.g1_add(total_c, bls.g1_mul_fast(tmp, randomizer))
        total_w = bls.g1_add(total_w, bls.g1_mul_fast(w, randomizer))
        randomizer = rng.gen_u128() % R
    total_c = bls.g1_add(total_c, bls.g1_neg(bls.g1_mul_fast(pv["g"], g_mult)))
    total_c = bls.g1_add(total_c, bls.g1_neg(bls.g1_mul_fast(pv["gamma_g"], gamma_g_mult)))
    return bls.product_of_pairings_is_one([(bls.g1_neg(total_w), pv["beta_h"]), (total_c, pv["h"])])


# ----------------------------------------------------------------------------- reference-shaped wrappers
def generate_rand():
    return test_rng()


def generate_universal_srs(num_constraints, num_variables, num_non_zero, rng):
    return universal_setup(num_constraints, num_variables, num_non_zero, rng)


def generate_proving_and_verifying_keys(universal_srs, constraint_system):
    return index(universal_srs, constraint_system)


def generate_proof(constraint_system, proving_key, rng):
    return prove(proving_key, constraint_system, rng)


def verify_proof(verifying_key, public_inputs, proof, rng):
    return verify(verifying_key, public_inputs, proof, rng)


# ----------------------------------------------------------------------------- workloads
def manual_constraints_circuit(a, b):
    """examples/manual-constraints.rs:15-31: one public input a, one witness b, (a - b) * 1 = 0."""
    cs = ConstraintSystem()
    va = cs.new_input_variable(a)
    vb = cs.new_witness_variable(b)
    cs.enforce_constraint([(1, va), (R - 1, vb)], [(1, cs.one())], [])
    return cs


def synthetic_circuit(n, a, b):
    """SURVEY §8d synthetic R1CS sized so that |H| = |K| = n exactly: instance [1, c, d, 0-pad],
    witnesses a, b and n - 6 copies of a; n - 1 rows a*b = c and one row c*b = d."""
    assert n >= 8 and n & (n - 1) == 0
    cs = ConstraintSystem()
    va = cs.new_witness_variable(a)
    vb = cs.new_witness_variable(b)
    c = a * b % R
    d = c * b % R
    vc = cs.new_input_variable(c)
    vd = cs.new_input_variable(d)
    for _ in range(n - 6):
        cs.new_witness_variable(a)
    for _ in range(n - 1):
        cs.enforce_constraint([(1, va)], [(1, vb)], [(1, vc)])
    cs.enforce_constraint([(1, vc)], [(1, vb)], [(1, vd)])
    return cs


def random_sparse_circuit(seed, num_inputs=5, free_witnesses=4, num_constraints=12, repeated_rows=0):
    """A small circuit with multi-term linear combinations, several public inputs, |K| != |H| and more non-zeros in B
    than in A (so that the indexer's balance_matrices swap is taken): every row is
    (sum of 1-3 terms) * (sum of 2-4 terms) = fresh product witness.  random.Random(seed) drives the shape, so the
    product-side builder (simpleworks_amd/workloads.py) lays out the identical system."""
    import random
    rnd = random.Random(seed)
    cs = ConstraintSystem()
    vars_, vals = [cs.one()], [1]
    for _ in range(num_inputs):
        v = rnd.randrange(R)
        vars_.append(cs.new_input_variable(v))
        vals.append(v)
    for _ in range(free_witnesses):
        v = rnd.randrange(R)
        vars_.append(cs.new_witness_variable(v))
        vals.append(v)
    rows = []
    for _ in range(num_constraints):
        def lc(lo, hi):
            terms, total = [], 0
            for _ in range(rnd.randint(lo, hi)):
                k = rnd.randrange(len(vars_))
                coeff = rnd.choice([1, 2, R - 1, rnd.randrange(R)])
                terms.append((coeff, vars_[k]))
                total = (total + coeff * vals[k]) % R
            return terms, total
        a, va = lc(1, 3)
        b, vb = lc(2, 4)
        prod = va * vb % R
        w = cs.new_witness_variable(prod)
        vars_.append(w)
        vals.append(prod)
        cs.enforce_constraint(a, b, [(1, w)])
        rows.append((a, b, [(1, w)]))
    for i in range(repeated_rows):  # more constraints than variables: |H| comes from the row count
        cs.enforce_constraint(*rows[i % len(rows)])
    return cs
