"""Workload definitions for the configurations BASELINE.json names.

manual_constraints_circuit  config #1: /root/reference/examples/manual-constraints.rs:15-31 — one public input a, one
                            witness b, one row (a - b) * 1 = 0.
random_sparse_circuit       parity case beyond the BASELINE shapes: multi-term rows, 5 public inputs, |K| != |H|, nnz(B) > nnz(A).
synthetic_r1cs              configs #2-#4: the ark-marlin test/bench circuit shape — witnesses a, b, public c = a*b and
                            d = c*b, n - 1 rows a*b = c and one row c*b = d, padded with copies of a so that
                            |H| = |K| = n exactly (instance [1, c, d, 0-pad], n - 4 witnesses).  Built with numpy so that
                            n = 2^20 .. 2^22 take milliseconds to lay out.
"""
import numpy as np

from .marlin import ConstraintSystem, PackedR1cs, R_MODULUS, _MONT_R, _to_mont_limbs


def manual_constraints_circuit(a, b):
    cs = ConstraintSystem()
    va = cs.new_input_variable(a)
    vb = cs.new_witness_variable(b)
    cs.enforce_constraint([(1, va), (R_MODULUS - 1, vb)], [(1, cs.one())], [])
    return cs


def build_test_circuit(cs, a, b):
    """examples/test-circuit.rs:13-26 (BASELINE configs[0]): two PRIVATE u8 values as UInt8::new_witness — eight booleans each,
    least significant first, every one with its booleanity row (1 - x) x = 0 — and a.enforce_equal(&b) bit by bit,
    (a_i - b_i) 1 = 0.  24 constraints, 16 witnesses, no public input (the reference verifies with `&[]`, :80).  `cs` is any
    builder with ark-relations' vocabulary."""
    one = cs.one()
    bits = []
    for v in (a, b):
        row = []
        for i in range(8):
            x = cs.new_witness_variable((v >> i) & 1)
            cs.enforce_constraint([(1, one), (R_MODULUS - 1, x)], [(1, x)], [])
            row.append(x)
        bits.append(row)
    for xa, xb in zip(*bits):
        cs.enforce_constraint([(1, xa), (R_MODULUS - 1, xb)], [(1, one)], [])
    return []


def test_circuit(a, b):
    cs = ConstraintSystem()
    build_test_circuit(cs, a, b)
    return cs


test_circuit.__test__ = False  # a circuit constructor, not a pytest case


def synthetic_circuit(n, a, b):
    """Same circuit through the ConstraintSystem builder (small n)."""
    assert n >= 8 and n & (n - 1) == 0
    cs = ConstraintSystem()
    va = cs.new_witness_variable(a)
    vb = cs.new_witness_variable(b)
    c = a * b % R_MODULUS
    d = c * b % R_MODULUS
    vc = cs.new_input_variable(c)
    vd = cs.new_input_variable(d)
    for _ in range(n - 6):
        cs.new_witness_variable(a)
    for _ in range(n - 1):
        cs.enforce_constraint([(1, va)], [(1, vb)], [(1, vc)])
    cs.enforce_constraint([(1, vc)], [(1, vb)], [(1, vd)])
    return cs


def random_sparse_circuit(seed, num_inputs=5, free_witnesses=4, num_constraints=12, repeated_rows=0):
    """Small circuit with multi-term linear combinations, several public inputs, |K| != |H| and nnz(B) > nnz(A): rows
    (1-3 terms) * (2-4 terms) = fresh product witness, shape drawn from random.Random(seed).  The test suite's
    reference model builds the identical system from the same arguments; tests/golden/marlin.json holds its proofs."""
    import random
    rnd = random.Random(seed)
    cs = ConstraintSystem()
    vars_, vals = [cs.one()], [1]
    for _ in range(num_inputs):
        v = rnd.randrange(R_MODULUS)
        vars_.append(cs.new_input_variable(v))
        vals.append(v)
    for _ in range(free_witnesses):
        v = rnd.randrange(R_MODULUS)
        vars_.append(cs.new_witness_variable(v))
        vals.append(v)
    rows = []
    for _ in range(num_constraints):
        def lc(lo, hi):
            terms, total = [], 0
            for _ in range(rnd.randint(lo, hi)):
                k = rnd.randrange(len(vars_))
                coeff = rnd.choice([1, 2, R_MODULUS - 1, rnd.randrange(R_MODULUS)])
                terms.append((coeff, vars_[k]))
                total = (total + coeff * vals[k]) % R_MODULUS
            return terms, total
        a, va = lc(1, 3)
        b, vb = lc(2, 4)
        prod = va * vb % R_MODULUS
        w = cs.new_witness_variable(prod)
        vars_.append(w)
        vals.append(prod)
        cs.enforce_constraint(a, b, [(1, w)])
        rows.append((a, b, [(1, w)]))
    for i in range(repeated_rows):  # more constraints than variables: |H| comes from the row count
        cs.enforce_constraint(*rows[i % len(rows)])
    return cs


def synthetic_r1cs(n, a, b):
    """Vectorised layout of synthetic_circuit(n, a, b) as a PackedR1cs; returns (packed, public_inputs)."""
    assert n >= 8 and n & (n - 1) == 0
    a %= R_MODULUS
    b %= R_MODULUS
    c = a * b % R_MODULUS
    d = c * b % R_MODULUS
    instance = _to_mont_limbs([1, c, d])
    am = _to_mont_limbs([a, b])
    witness = np.empty((n - 4, 4), dtype=np.uint64)
    witness[:] = am[0]
    witness[1] = am[1]
    one = _to_mont_limbs([1])[0]
    ninst = 3
    col_a, col_b, col_c, col_d = ninst + 0, ninst + 1, 1, 2
    rowptr = np.arange(n + 1, dtype=np.uint32)
    val = np.empty((n, 4), dtype=np.uint64)
    val[:] = one

    def mat(cols_main, col_last):
        col = np.full(n, cols_main, dtype=np.uint32)
        col[-1] = col_last
        return rowptr, col, val

    packed = PackedR1cs(instance, witness, mat(col_a, col_c), mat(col_b, col_b), mat(col_c, col_d))
    return packed, [c, d]


# ------------------------------------------------------------------------------------------------ R1CS dump ("SWMR1CS1")
# A synthesised constraint system as ONE flat little-endian file: what a Rust-side caller writes with
# swmarlin_sys::r1cs_dump::dump_r1cs(&cs, path) after `cs.finalize()` (ark-relations' `to_matrices` + the two assignment
# vectors), and what `load_r1cs` / `bench.py --r1cs FILE` read back — the way the reference's REAL circuits (e.g.
# MerkleTreeVerificationU8, /root/reference/src/merkle_tree/merkle_tree_verification_u8.rs:25-58, whose constraint layout
# comes out of ark-r1cs-std and cannot be reproduced here) reach this library without a Rust toolchain on the GPU box.
#   0   8  magic "SWMR1CS1"
#   8   8  num_instance   (instance assignment, the leading one included)          u64
#  16   8  num_witness                                                              u64
#  24   8  num_constraints                                                          u64
#  32  24  nnz of A, B, C                                                           3 x u64
#  56   8  flags: bit 0 = field elements are Montgomery limbs (R = 2^256; always set)
#  64      instance  num_instance x 32 B | witness  num_witness x 32 B
#          per matrix A, B, C:  rowptr (num_constraints + 1) x u32 | col nnz x u32 (variable index: instance variables first,
#          then witness variables — ark-relations' Matrix convention) | zero padding to a multiple of 8 | val nnz x 32 B
#  end 32  BLAKE2s-256 of everything before it
R1CS_MAGIC = b"SWMR1CS1"


def dump_r1cs(packed, path):
    """Writes PackedR1cs `packed` (or anything with .pack()) to `path` in the SWMR1CS1 layout; returns the byte count."""
    import hashlib
    packed = packed.pack()
    head = np.array([packed.instance.shape[0], packed.witness.shape[0], packed.num_constraints] +
                    [int(m[1].shape[0]) for m in packed.mats] + [1], dtype="<u8")
    parts = [R1CS_MAGIC, head.tobytes(), packed.instance.astype("<u8").tobytes(), packed.witness.astype("<u8").tobytes()]
    for rowptr, col, val in packed.mats:
        idx = rowptr.astype("<u4").tobytes() + col.astype("<u4").tobytes()
        parts += [idx, b"\0" * (-len(idx) % 8), val.astype("<u8").tobytes()]
    body = b"".join(parts)
    with open(path, "wb") as f:
        f.write(body)
        f.write(hashlib.blake2s(body).digest())
    return len(body) + 32


def pack_model_system(cs):
    """A constraint system with .instance / .witness (ints) and .to_matrices() -> rows of (coefficient, column) — the oracle's
    model, or anything shaped like ark-relations' ConstraintSystem after finalize — as a PackedR1cs."""
    mats = []
    for rows in cs.to_matrices():
        rowptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)
        col = np.array([c for r in rows for _, c in r], dtype=np.uint32)
        mats.append((rowptr, col, _to_mont_limbs([v for r in rows for v, _ in r])))
    return PackedR1cs(_to_mont_limbs(cs.instance), _to_mont_limbs(cs.witness), *mats)


def load_r1cs(path):
    """SWMR1CS1 file -> (PackedR1cs, public_inputs): the public inputs are the instance assignment without its leading one,
    as ints in standard form (what verify_proof takes).  Every inconsistency of the file is a ValueError."""
    import hashlib
    data = open(path, "rb").read()
    if len(data) < 64 + 32 or data[:8] != R1CS_MAGIC:
        raise ValueError("not an SWMR1CS1 file")
    if hashlib.blake2s(data[:-32]).digest() != data[-32:]:
        raise ValueError("SWMR1CS1: checksum mismatch (truncated or corrupted file)")
    ninst, nwit, nrows, na, nb, nc, flags = (int(x) for x in np.frombuffer(data, dtype="<u8", count=7, offset=8))
    if flags != 1:
        raise ValueError("SWMR1CS1: unknown flags %#x" % flags)
    if ninst < 1 or max(ninst, nwit, nrows, na, nb, nc) >= 1 << 31:
        raise ValueError("SWMR1CS1: implausible header")
    want = 64 + 32 * (ninst + nwit) + sum(4 * (nrows + 1 + k) + (-4 * (nrows + 1 + k) % 8) + 32 * k for k in (na, nb, nc)) + 32
    if want != len(data):
        raise ValueError("SWMR1CS1: %d bytes, the header describes %d" % (len(data), want))
    off = 64
    instance = np.frombuffer(data, dtype="<u8", count=4 * ninst, offset=off).reshape(-1, 4)
    off += 32 * ninst
    witness = np.frombuffer(data, dtype="<u8", count=4 * nwit, offset=off).reshape(-1, 4)
    off += 32 * nwit
    mats = []
    for k in (na, nb, nc):
        rowptr = np.frombuffer(data, dtype="<u4", count=nrows + 1, offset=off)
        col = np.frombuffer(data, dtype="<u4", count=k, offset=off + 4 * (nrows + 1))
        off += 4 * (nrows + 1 + k) + (-4 * (nrows + 1 + k) % 8)
        val = np.frombuffer(data, dtype="<u8", count=4 * k, offset=off).reshape(-1, 4)
        off += 32 * k
        if int(rowptr[0]) != 0 or int(rowptr[-1]) != k or (np.diff(rowptr.astype(np.int64)) < 0).any():
            raise ValueError("SWMR1CS1: row pointers are not a monotone prefix of the non-zeros")
        if k and int(col.max()) >= ninst + nwit:
            raise ValueError("SWMR1CS1: column index beyond the variables")
        mats.append((rowptr, col, val))
    mont_r_inv = pow(_MONT_R, -1, R_MODULUS)

    def std(limbs):  # Montgomery limbs -> int in standard form; a non-canonical residue is refused
        v = sum(int(limbs[i]) << (64 * i) for i in range(4))
        if v >= R_MODULUS:
            raise ValueError("SWMR1CS1: field element out of range")
        return v * mont_r_inv % R_MODULUS
    if std(instance[0]) != 1:
        raise ValueError("SWMR1CS1: the instance assignment does not start with one")
    public = [std(instance[i]) for i in range(1, ninst)]
    return PackedR1cs(instance, witness, *mats), public


# ===================================================================================================================
# BASELINE config #5 stand-in: Pedersen-hash Merkle-membership circuit (examples/merkle-tree, SimpleMerkleTree)
#
# What the reference proves there (src/merkle_tree/merkle_tree_verification_u8.rs:25-58): a u8 leaf is a member of a
# Pedersen Merkle tree — public inputs [root, 8 leaf bits LSB-first] (src/merkle_tree/simple_merkle_tree.rs:129-143,
# src/gadgets/traits.rs:150-164), private authentication path; leaf hash = Pedersen CRH with 144 windows of 4 bits,
# inner nodes = Pedersen CRH with 128 windows of 4 bits over left || right digests (src/merkle_tree/common.rs:11-52),
# both on ed-on-BLS12-377 and compressed to the x coordinate (TECompressor); a tree over 2^18 leaves has height 19
# (simple_merkle_tree.rs:155-163): one leaf hash + 18 two-to-one hashes in the circuit.
#
# The reference synthesises that R1CS with ark-r1cs-std / ark-crypto-primitives gadgets, which are not available here
# (SURVEY.md §8d), so the constraint-by-constraint layout below is OURS: same statement, same I/O convention, same hash
# (window sizes, bit order, curve), our own gadget for the conditional point addition.  The Pedersen generators:
# MerkleParams.setup(rng) samples them the way CRH::setup does from the caller's generator (SimpleMerkleTree's default);
# MerkleParams() derives them from a fixed seed (the committed fixtures) — circuit constants either way.  What matters
# to the hot path is the SHAPE this gives the prover, which the one-term-per-row synthetic circuit lacks:
#   * multi-term rows (1-3 terms; bit-packing rows with 257 terms), booleanity rows b (1 - b) = 0,
#   * rows with empty A and B, `0 * 0 = a - b` — the shape simpleworks' own UInt gadgets emit
#     (src/gadgets/uint8.rs:117-118, :165-167) — from the canonical-range rows of the digests and from the optional
#     block of simpleworks-style UInt8 operations (shift / xor / and) on the path bytes,
#   * a 0/1-heavy witness (bits, and products with a zero bit), |K| != |H|.
# ===================================================================================================================
ED_A = R_MODULUS - 1          # twisted Edwards a = -1
ED_D = 3021                   # d
ED_COFACTOR = 4
ED_SUBGROUP_ORDER = 2111115437357092606062206234695386632838870926408408195193685246394721360383
ED_GENERATOR = (4497879464030519973909970603271755437257548612157028181994697785683032656389,
                4357141146396347889246900916607623952598927460421559113092863576544024487809)


def ed_add(p, q):
    """Unified twisted-Edwards addition on ed-on-BLS12-377 (a = -1, d = 3021) in affine coordinates."""
    x1, y1 = p
    x2, y2 = q
    t = ED_D * x1 % R_MODULUS * x2 % R_MODULUS * y1 % R_MODULUS * y2 % R_MODULUS
    x3 = (x1 * y2 + y1 * x2) * pow(1 + t, -1, R_MODULUS) % R_MODULUS
    y3 = (y1 * y2 + x1 * x2) * pow(1 - t, -1, R_MODULUS) % R_MODULUS
    return x3, y3


def ed_mul(p, k):
    acc = (0, 1)
    while k:
        if k & 1:
            acc = ed_add(acc, p)
        p = ed_add(p, p)
        k >>= 1
    return acc


def ed_on_curve(p):
    x, y = p
    return (ED_A * x * x + y * y - 1 - ED_D * x * x % R_MODULUS * y * y) % R_MODULUS == 0


def _fr_sqrt(v):
    """Tonelli-Shanks in Fr (two-adicity 47)."""
    v %= R_MODULUS
    if v == 0:
        return 0
    if pow(v, (R_MODULUS - 1) // 2, R_MODULUS) != 1:
        return None
    s, q = 47, (R_MODULUS - 1) >> 47
    z = pow(22, q, R_MODULUS)  # 22 generates Fr*
    m, c, t, r = s, z, pow(v, q, R_MODULUS), pow(v, (q + 1) // 2, R_MODULUS)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % R_MODULUS
            i += 1
        b = pow(c, 1 << (m - i - 1), R_MODULUS)
        m, c = i, b * b % R_MODULUS
        t, r = t * c % R_MODULUS, r * b % R_MODULUS
    return r


class _SplitMix:
    def __init__(self, seed):
        self.s = seed & ((1 << 64) - 1)

    def next_u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & ((1 << 64) - 1)
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & ((1 << 64) - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & ((1 << 64) - 1)
        return z ^ (z >> 31)

    def fr(self):
        while True:
            v = 0
            for k in range(4):
                v |= self.next_u64() << (64 * k)
            v &= (1 << 253) - 1
            if v < R_MODULUS:
                return v


def pedersen_generators(num_windows, window_size, seed):
    """ark-crypto-primitives pedersen::CRH::setup shape: one random subgroup element per window and its doublings,
    generators[w][j] = 2^j * base_w.  Bases by try-and-increment on y from a seeded generator, cofactor cleared."""
    g = _SplitMix(seed)
    out = []
    while len(out) < num_windows:
        y = g.fr()
        num = (1 - y * y) % R_MODULUS
        den = (ED_A - ED_D * y * y) % R_MODULUS
        x = _fr_sqrt(num * pow(den, -1, R_MODULUS))
        if x is None or x == 0:
            continue
        p = ed_mul((x, y), ED_COFACTOR)
        if p == (0, 1):
            continue
        row = [p]
        for _ in range(window_size - 1):
            row.append(ed_add(row[-1], row[-1]))
        out.append(row)
    return out


def pedersen_hash_bits(bits, gens):
    """Pedersen CRH + TECompressor: x coordinate of sum_k bits[k] * gens[k // ws][k % ws] (missing bits are zero)."""
    ws = len(gens[0])
    assert len(bits) <= ws * len(gens)
    acc = (0, 1)
    for k, b in enumerate(bits):
        if b:
            acc = ed_add(acc, gens[k // ws][k % ws])
    return acc[0]


def _bits_le(v, n):
    return [(v >> i) & 1 for i in range(n)]


class MerkleParams:
    """Hash parameters of the membership circuit.  digest_bits = 256 with 144 / 128 windows of 4 is the reference's
    configuration (src/merkle_tree/common.rs:16-30); smaller values give toy instances for the pure-Python prover
    (only the low digest_bits bits of a child digest enter its parent's hash; the rest is carried in one witness)."""

    def __init__(self, digest_bits=256, leaf_windows=144, inner_windows=128, window_size=4, seed=0x5157_4D41_524C_494E,
                 generators=None):
        assert 2 * digest_bits <= inner_windows * window_size and 8 <= leaf_windows * window_size
        self.digest_bits = digest_bits
        if generators is not None:
            self.leaf_gens, self.inner_gens = generators
        else:
            self.leaf_gens = pedersen_generators(leaf_windows, window_size, seed)
            self.inner_gens = pedersen_generators(inner_windows, window_size, seed ^ 0xA5A5_A5A5_5A5A_5A5A)
        self._crh = None  # (leaf, two-to-one) PedersenCRH handles of the GPU tree builder, created on first use

    @classmethod
    def setup(cls, rng):
        """The reference's order of sampling (src/merkle_tree/simple_merkle_tree.rs:43-45): <LeafHash as CRH>::setup(&mut rng),
        then <TwoToOneHash as TwoToOneCRH>::setup(&mut rng), from the caller's generator."""
        from .hash import pedersen_setup, LEAF_WINDOWS, TWO_TO_ONE_WINDOWS, WINDOW_SIZE
        leaf = pedersen_setup(rng, LEAF_WINDOWS, WINDOW_SIZE)
        inner = pedersen_setup(rng, TWO_TO_ONE_WINDOWS, WINDOW_SIZE)
        return cls(generators=(leaf, inner))

    def crh(self, ctx=None):
        """The two hash parameter sets resident on the GPU (simpleworks_amd.hash.PedersenCRH)."""
        if self._crh is None:
            from .hash import PedersenCRH
            self._crh = (PedersenCRH(self.leaf_gens, ctx), PedersenCRH(self.inner_gens, ctx))
        return self._crh

    def leaf_hash(self, leaf_u8):
        return pedersen_hash_bits(_bits_le(leaf_u8, 8), self.leaf_gens)

    def inner_hash(self, left, right):
        d = self.digest_bits
        return pedersen_hash_bits(_bits_le(left, d) + _bits_le(right, d), self.inner_gens)

    def root_from_path(self, leaf_u8, leaf_index, siblings):
        cur = self.leaf_hash(leaf_u8)
        for lvl, s in enumerate(siblings):
            cur = self.inner_hash(s, cur) if (leaf_index >> lvl) & 1 else self.inner_hash(cur, s)
        return cur

    def build_tree(self, leaves_u8, ctx=None):
        """All levels of the tree over len(leaves) = 2^h leaves, bottom up (MerkleTree::new,
        src/merkle_tree/simple_merkle_tree.rs:47-49); returns the list of levels, levels[-1][0] is the root.
        The reference's configuration (256-bit digests) is built on the GPU (swm_merkle_tree_build); the truncated-digest toy
        parameter sets of the pure-Python prover fixtures, which are not the reference's hash, stay on the loop below."""
        assert len(leaves_u8) & (len(leaves_u8) - 1) == 0
        if self.digest_bits == 256:
            from .hash import MerkleTree
            leaf, inner = self.crh(ctx)
            return MerkleTree.new(leaf, inner, [int(v) for v in leaves_u8]).int_levels()
        levels = [[self.leaf_hash(v) for v in leaves_u8]]
        while len(levels[-1]) > 1:
            prev = levels[-1]
            levels.append([self.inner_hash(prev[2 * i], prev[2 * i + 1]) for i in range(len(prev) // 2)])
        return levels

    @staticmethod
    def path_of(levels, index):
        return [levels[lvl][(index >> lvl) ^ 1] for lvl in range(len(levels) - 1)]


class _LC:
    """A linear combination with its value: what a gadget coordinate is between constraints."""
    __slots__ = ("terms", "value")

    def __init__(self, terms, value):
        self.terms, self.value = terms, value % R_MODULUS

    def scaled(self, k):
        k %= R_MODULUS
        return _LC([(c * k % R_MODULUS, v) for c, v in self.terms], self.value * k)

    def plus(self, o):
        return _LC(self.terms + o.terms, self.value + o.value)

    def minus(self, o):
        return self.plus(o.scaled(R_MODULUS - 1))


def _cond_add_const(cs, one, acc, point, bit):
    """acc + bit * point for a CONSTANT point and a boolean variable `bit` (an _LC over one variable):
        t = X Y;  bt = bit t;  m1 = bit ((cy - 1) X + cx Y);  m2 = bit ((cy - 1) Y + cx X)
        X3 (1 + k bt) = X + m1;   Y3 (1 - k bt) = Y + m2,   k = d cx cy        (a = -1)
    six rows, six new witnesses.  acc = None is the identity: the sum is linear in the bit and costs nothing."""
    cx, cy = point
    if acc is None:
        return (bit.scaled(cx), _LC([(1, one)], 1).plus(bit.scaled(cy - 1)))
    X, Y = acc

    def witness(v):
        return _LC([(1, cs.new_witness_variable(v % R_MODULUS))], v)

    def product(a, b):
        w = witness(a.value * b.value)
        cs.enforce_constraint(a.terms, b.terms, w.terms)
        return w
    t = product(X, Y)
    bt = product(bit, t)
    m1 = product(bit, X.scaled(cy - 1).plus(Y.scaled(cx)))
    m2 = product(bit, Y.scaled(cy - 1).plus(X.scaled(cx)))
    k = ED_D * cx % R_MODULUS * cy % R_MODULUS
    kbt = bt.scaled(k)
    den_x = _LC([(1, one)], 1).plus(kbt)
    den_y = _LC([(1, one)], 1).minus(kbt)
    num_x, num_y = X.plus(m1), Y.plus(m2)
    X3 = witness(num_x.value * pow(den_x.value, -1, R_MODULUS))
    Y3 = witness(num_y.value * pow(den_y.value, -1, R_MODULUS))
    cs.enforce_constraint(X3.terms, den_x.terms, num_x.terms)
    cs.enforce_constraint(Y3.terms, den_y.terms, num_y.terms)
    return (X3, Y3)


def _boolean_witness(cs, one, v):
    """Boolean::new_witness: b (1 - b) = 0."""
    b = _LC([(1, cs.new_witness_variable(v))], v)
    cs.enforce_constraint(b.terms, [(1, one), (R_MODULUS - 1, b.terms[0][1])], [])
    return b


def build_merkle_membership(cs, params, leaf_u8, leaf_index, siblings, gadget_byte_ops=0, root=None):
    """Emits the membership circuit into `cs` (any builder with ark-relations' vocabulary: new_input_variable,
    new_witness_variable, enforce_constraint(a, b, c), one()).  Public inputs, in order: root, then the 8 leaf bits
    LSB-first — the vector SimpleMerkleTree::verify rebuilds (src/merkle_tree/simple_merkle_tree.rs:129-143).
    `root` overrides the public root (a wrong one gives an unsatisfied system: the final row fails).
    gadget_byte_ops > 0 appends that many simpleworks-style UInt8 operations (shl / xor / and, cycling) on the bytes of
    the decomposed digests: 8 new boolean witnesses and 8-16 rows each, half of them with empty A and B
    (src/gadgets/uint8.rs:117-118, :165-167).  Returns the public-input list [root, b0..b7]."""
    one = cs.one()
    d = params.digest_bits
    true_root = params.root_from_path(leaf_u8, leaf_index, siblings)
    pub_root = true_root if root is None else root % R_MODULUS
    root_v = _LC([(1, cs.new_input_variable(pub_root))], pub_root)
    leaf_bits = []
    for i in range(8):  # UInt8::new_input: 8 booleans, least significant first
        v = (leaf_u8 >> i) & 1
        b = _LC([(1, cs.new_input_variable(v))], v)
        cs.enforce_constraint(b.terms, [(1, one), (R_MODULUS - 1, b.terms[0][1])], [])
        leaf_bits.append(b)
    acc = None
    ws = len(params.leaf_gens[0])
    for k, b in enumerate(leaf_bits):
        acc = _cond_add_const(cs, one, acc, params.leaf_gens[k // ws][k % ws], b)
    cur = acc[0]
    byte_pool = []
    ws = len(params.inner_gens[0])
    for lvl, sib in enumerate(siblings):
        dirbit = _boolean_witness(cs, one, (leaf_index >> lvl) & 1)
        s = _LC([(1, cs.new_witness_variable(sib % R_MODULUS))], sib)
        # left = dir ? sibling : cur  (one row), right = cur + sibling - left (linear)
        left_v = s.value if dirbit.value else cur.value
        left = _LC([(1, cs.new_witness_variable(left_v))], left_v)
        cs.enforce_constraint(dirbit.terms, s.minus(cur).terms, left.minus(cur).terms)
        right = cur.plus(s).minus(left)
        bits = []
        for child in (left, right):
            cb = [_boolean_witness(cs, one, (child.value >> i) & 1) for i in range(d)]
            packed = _LC([], 0)
            for i, b in enumerate(cb):
                packed = packed.plus(b.scaled(1 << i))
            if d < 256:  # toy digests: the bits above digest_bits travel in one unconstrained witness
                hi = child.value >> d
                packed = packed.plus(_LC([(1 << d, cs.new_witness_variable(hi))], hi << d))
            cs.enforce_constraint(packed.minus(child).terms, [(1, one)], [])
            for i in range(253, d):  # canonical range: r < 2^253, the top bits are zero — rows `0 * 0 = bit`
                cs.enforce_constraint([], [], cb[i].terms)
            bits += cb
            for i in range(0, d - 7, 8):
                byte_pool.append(cb[i:i + 8])
        acc = None
        for k, b in enumerate(bits):
            acc = _cond_add_const(cs, one, acc, params.inner_gens[k // ws][k % ws], b)
        cur = acc[0]
    cs.enforce_constraint(cur.minus(root_v).terms, [(1, one)], [])  # is_member.enforce_equal(TRUE)
    # optional block of simpleworks UInt8 gadget rows over the path bytes
    for op in range(gadget_byte_ops):
        a = byte_pool[(7 * op) % len(byte_pool)]
        b = byte_pool[(11 * op + 3) % len(byte_pool)]
        kind = op % 3
        if kind == 0:  # shift_left by k (src/gadgets/uint8.rs:141-185): new UInt8 witness + rows 0 * 0 = lc
            k = 1 + op % 7
            c = [_boolean_witness(cs, one, a[i - k].value if i >= k else 0) for i in range(8)]
            for i in range(8):
                cs.enforce_constraint([], [], c[i].terms if i < k else a[i - k].minus(c[i]).terms)
        elif kind == 1:  # xor (ark-r1cs-std Boolean::xor): (a + a) * b = a + b - c
            c = []
            for i in range(8):
                v = a[i].value ^ b[i].value
                ci = _LC([(1, cs.new_witness_variable(v))], v)
                cs.enforce_constraint(a[i].scaled(2).terms, b[i].terms, a[i].plus(b[i]).minus(ci).terms)
                c.append(ci)
        else:  # and: a * b = c
            c = []
            for i in range(8):
                v = a[i].value & b[i].value
                ci = _LC([(1, cs.new_witness_variable(v))], v)
                cs.enforce_constraint(a[i].terms, b[i].terms, ci.terms)
                c.append(ci)
        byte_pool.append(c)
    return [pub_root] + [(leaf_u8 >> i) & 1 for i in range(8)]


def merkle_membership_circuit(height=19, leaf_u8=0xA7, leaf_index=None, seed=7, gadget_byte_ops=2400, params=None,
                              root=None, siblings=None):
    """BASELINE config #5 stand-in as a ConstraintSystem.  height = merkle_tree_height(number of leaves)
    (src/merkle_tree/simple_merkle_tree.rs:155-163): height - 1 two-to-one hashes; 19 for 2^18 leaves.  `siblings`: the
    authentication path of a real tree (MerkleParams.build_tree + path_of; bench.py --circuit merkle does that); without it
    the siblings are random digests (the other leaves are not needed to prove one path).  Returns (cs, public_inputs, params)."""
    params = params or MerkleParams()
    g = _SplitMix(seed)
    levels = height - 1
    if siblings is None:
        siblings = [g.fr() for _ in range(levels)]
    assert len(siblings) == levels
    if leaf_index is None:
        leaf_index = g.next_u64() % (1 << levels)
    cs = ConstraintSystem()
    public = build_merkle_membership(cs, params, leaf_u8, leaf_index, siblings, gadget_byte_ops, root)
    return cs, public, params


# ===================================================================================================================
# The reference's driver of that circuit: SimpleMerkleTree (src/merkle_tree/simple_merkle_tree.rs:35-153), call for call
# ===================================================================================================================
class MerkleTreeVerificationU8:
    """The ConstraintSynthesizer the driver hands to MarlinInst (src/merkle_tree/merkle_tree_verification_u8.rs:25-58):
    constants = hash parameters, public = root + leaf, witness = authentication path."""

    def __init__(self, params, root, leaf, leaf_index, authentication_path, gadget_byte_ops=0):
        self.params, self.root, self.leaf, self.leaf_index = params, root, leaf, leaf_index
        self.authentication_path, self.gadget_byte_ops = list(authentication_path), gadget_byte_ops

    def generate_constraints(self, cs):
        build_merkle_membership(cs, self.params, self.leaf, self.leaf_index, self.authentication_path,
                                self.gadget_byte_ops, root=self.root)


def merkle_tree_height(leaves_length):
    """src/merkle_tree/simple_merkle_tree.rs:155-163."""
    result = 0
    while leaves_length:
        result += 1
        leaves_length >>= 1
    return result


class SimpleMerkleTree:
    """SimpleMerkleTree::{new, get_merkle_path, prove, verify} with the same call sequence into MarlinInst: a fresh test_rng
    and universal_setup(100_000, 25_000, 300_000) in new(), keys from a DUMMY circuit over a blank tree of the same height
    (the circuit's shape depends on the height only), a fresh test_rng per prove / verify, proofs as serialised bytes,
    verify(proof_bytes, leaf_u8) rebuilding the public input [root, 8 bits LSB-first].  Hash parameters: MerkleParams (the
    reference samples them from the same rng, after universal_setup: so does this, MerkleParams.setup)."""

    def __init__(self, leaves_u8, params=None, srs_sizes=(100_000, 25_000, 300_000), gadget_byte_ops=0, ctx=None):
        from . import marlin as M
        from . import serialization as S
        self._M, self._S = M, S
        rng = M.generate_rand()                                                  # ark_std::test_rng()
        universal_srs = M.MarlinInst.universal_setup(*srs_sizes, rng, ctx)       # simple_merkle_tree.rs:39
        self.params = params or MerkleParams.setup(rng)                          # LeafHash / TwoToOneHash setup(&mut rng), :43-45
        self.leaves = list(leaves_u8)
        if self.params.digest_bits == 256:                                       # MerkleTree::new, :47-49 (on the GPU)
            from .hash import MerkleTree
            leaf_crh, inner_crh = self.params.crh(ctx)
            self.tree = MerkleTree.new(leaf_crh, inner_crh, [int(v) for v in self.leaves])
            self.levels = self.tree.levels                                       # digests as bytes: converted where they are used
        else:
            self.tree = None
            self.levels = self.params.build_tree(self.leaves, ctx)
        height = merkle_tree_height(len(self.leaves))
        blank_path = [0] * (height - 1)                                          # MerkleTree::blank(..).generate_proof(0)
        blank_root = self.params.root_from_path(0, 0, blank_path)
        dummy = MerkleTreeVerificationU8(self.params, blank_root, 0, 0, blank_path, gadget_byte_ops)
        self.gadget_byte_ops = gadget_byte_ops
        self.proving_key, self.verifying_key = M.MarlinInst.index(universal_srs, dummy)   # :83
        universal_srs.free()

    def root(self):
        return self.tree.root() if self.tree is not None else self.levels[-1][0]

    def get_merkle_path(self, leaf_index):
        if self.tree is not None:
            return leaf_index, self.tree.generate_proof(leaf_index)              # tree.generate_proof(leaf_index), :99-103
        return leaf_index, MerkleParams.path_of(self.levels, leaf_index)

    def prove(self, leaf, merkle_path):
        leaf_index, siblings = merkle_path
        circuit = MerkleTreeVerificationU8(self.params, self.root(), leaf, leaf_index, siblings, self.gadget_byte_ops)
        rng = self._M.generate_rand()
        proof = self._M.MarlinInst.prove(self.proving_key, circuit, rng)         # :119
        return self._S.serialize_proof(proof)                                    # proof.serialize(&mut bytes)

    def verify(self, proof_bytes, input_u8):
        input_vec = [self.root()] + [(input_u8 >> i) & 1 for i in range(8)]      # :129-143
        proof = self._S.deserialize_proof(proof_bytes)
        rng = self._M.generate_rand()
        return self._M.MarlinInst.verify(self.verifying_key, input_vec, proof, rng)   # :148

    def free(self):
        self.proving_key.free()
