"""Native Pedersen hash and Pedersen Merkle tree, computed on the GPU (csrc/pedersen.hip through include/swmarlin.h).

Caller-facing mirror of what the reference reaches through ark-crypto-primitives 0.3:
    src/hash/mod.rs:13-28                           pedersen_hash(input): LeafWindow 144 x 4, parameters from a fresh test_rng
    src/merkle_tree/simple_merkle_tree.rs:43-49     <LeafHash as CRH>::setup(&mut rng), <TwoToOneHash as TwoToOneCRH>::setup(&mut rng),
                                                    MerkleTree::<MerkleConfig>::new(&leaf_crh_params, &two_to_one_crh_params, leaves)
    src/merkle_tree/simple_merkle_tree.rs:99-103    tree.generate_proof(leaf_index), tree.root()
    src/merkle_tree/common.rs:11-30                 the two window shapes

Host side (this file): sampling the parameters — CRH::setup is a few hundred curve operations, done with Python integers the
way ark-ec samples a twisted Edwards point [U] — and the tree's bookkeeping.  Every hash runs on the GPU; there is no CPU
evaluation path here (the checker lives under oracle/).
"""
import numpy as np

from .marlin import R_MODULUS, default_context, generate_rand

ED_D = 3021            # ed-on-BLS12-377: -x^2 + y^2 = 1 + 3021 x^2 y^2 over BLS12-377 Fr
ED_COFACTOR = 4
LEAF_WINDOWS, TWO_TO_ONE_WINDOWS, WINDOW_SIZE = 144, 128, 4   # src/merkle_tree/common.rs:16-30, src/hash/mod.rs:16-19
_MONT_RINV = pow(1 << 256, -1, R_MODULUS)


def ed_add(p, q):
    """Unified affine addition (a = -1)."""
    x1, y1 = p
    x2, y2 = q
    t = ED_D * x1 % R_MODULUS * x2 % R_MODULUS * y1 % R_MODULUS * y2 % R_MODULUS
    x3 = (x1 * y2 + y1 * x2) * pow(1 + t, -1, R_MODULUS) % R_MODULUS
    y3 = (y1 * y2 + x1 * x2) * pow(1 - t, -1, R_MODULUS) % R_MODULUS
    return x3, y3


def fr_sqrt(v):
    """Tonelli-Shanks in Fr (two-adicity 47); None for a non-residue."""
    v %= R_MODULUS
    if v == 0:
        return 0
    if pow(v, (R_MODULUS - 1) // 2, R_MODULUS) != 1:
        return None
    q = (R_MODULUS - 1) >> 47
    m, c, t, r = 47, pow(22, q, R_MODULUS), pow(v, q, R_MODULUS), pow(v, (q + 1) // 2, R_MODULUS)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % R_MODULUS
            i += 1
        b = pow(c, 1 << (m - i - 1), R_MODULUS)
        m, c = i, b * b % R_MODULUS
        t, r = t * c % R_MODULUS, r * b % R_MODULUS
    return r


def _rand_fr(rng):
    """ark-ff UniformRand for Fr: the accepted limbs are the Montgomery representation (swm_rng_rand_fr returns them)."""
    limbs = rng.rand_fr_mont()
    return sum(int(l) << (64 * i) for i, l in enumerate(limbs)) * _MONT_RINV % R_MODULUS


def _gen_bool(rng):
    """rand 0.8 Standard for bool: the sign bit of next_u32 (fill_bytes(4) consumes exactly that word)."""
    return rng.fill_bytes(4)[3] >> 7 == 1


def ed_rand(rng):
    """ark-ec 0.3 twisted_edwards_extended, Distribution<GroupProjective<P>> for Standard [U]: x = Fq::rand, greatest =
    rng.gen(), y from get_point_from_x (y^2 = (a x^2 - 1) / (d x^2 - 1), the root with (y < -y) ^ greatest), then
    scale_by_cofactor; repeat while x is not an abscissa of the curve."""
    while True:
        x = _rand_fr(rng)
        greatest = _gen_bool(rng)
        x2 = x * x % R_MODULUS
        den = (ED_D * x2 - 1) % R_MODULUS
        if den == 0:
            continue
        y = fr_sqrt((-x2 - 1) * pow(den, -1, R_MODULUS))
        if y is None:
            continue
        negy = (-y) % R_MODULUS
        y = y if (y < negy) ^ greatest else negy
        p = (x, y)
        for _ in range(2):  # cofactor 4
            p = ed_add(p, p)
        return p


def pedersen_setup(rng, num_windows, window_size=WINDOW_SIZE):
    """pedersen::CRH::setup -> Parameters.generators [U]: per window a random point and its doublings."""
    gens = []
    for _ in range(num_windows):
        base = ed_rand(rng)
        row = []
        for _ in range(window_size):
            row.append(base)
            base = ed_add(base, base)
        gens.append(row)
    return gens


class PedersenCRH:
    """PedersenCRHCompressor<EdwardsProjective, TECompressor, W> with its Parameters resident on the GPU."""

    def __init__(self, generators, ctx=None):
        self.ctx = ctx or default_context()
        self.generators = generators
        self.num_windows, self.window_size = len(generators), len(generators[0])
        raw = b"".join(x.to_bytes(32, "little") + y.to_bytes(32, "little") for row in generators for x, y in row)
        self.h = self.ctx.pedersen_create(raw, self.num_windows, self.window_size)

    @classmethod
    def setup(cls, rng, num_windows, window_size=WINDOW_SIZE, ctx=None):
        return cls(pedersen_setup(rng, num_windows, window_size), ctx)

    def evaluate_many(self, inputs):
        """inputs: uint8 [count, input_len] -> uint8 [count, 32] (CRH::evaluate + to_bytes! of each digest)."""
        return self.ctx.pedersen_hash(self.h, inputs)

    def evaluate(self, data):
        """CRH::evaluate(&params, input) -> Fq (as an integer)."""
        a = np.frombuffer(bytes(data), dtype=np.uint8).reshape(1, -1)
        return int.from_bytes(self.evaluate_many(a)[0].tobytes(), "little")

    def free(self):
        if self.h:
            self.ctx.pedersen_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def pedersen_hash(data, ctx=None):
    """src/hash/mod.rs:23-28: parameters from a fresh test_rng (144 windows of 4 bits), then evaluate."""
    crh = PedersenCRH.setup(generate_rand(), LEAF_WINDOWS, WINDOW_SIZE, ctx)
    try:
        return crh.evaluate(data)
    finally:
        crh.free()


def _leaf_bytes(leaves):
    rows = [bytes([v]) if isinstance(v, (int, np.integer)) else bytes(v) for v in leaves]   # to_bytes![leaf]; u8 -> one byte
    if len({len(r) for r in rows}) != 1:
        raise ValueError("leaves must serialise to the same number of bytes")
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), -1)


class MerkleTree:
    """ark_crypto_primitives::merkle_tree::MerkleTree over (LeafHash, TwoToOneHash): built by swm_merkle_tree_build."""

    def __init__(self, levels):
        self.levels = levels          # levels[0] = leaf digests ... levels[-1] = [root], uint8 [count, 32] each

    @staticmethod
    def new(leaf_crh, two_to_one_crh, leaves):
        a = _leaf_bytes(leaves)
        nodes = leaf_crh.ctx.merkle_tree_build(leaf_crh.h, two_to_one_crh.h, a)
        levels, off, cnt = [], 0, a.shape[0]
        while cnt >= 1:
            levels.append(nodes[off:off + cnt])
            off += cnt
            cnt >>= 1
        return MerkleTree(levels)

    def height(self):
        """tree.height() of ark-crypto-primitives: levels including the leaves."""
        return len(self.levels)

    def node(self, level, index):
        return int.from_bytes(self.levels[level][index].tobytes(), "little")

    def root(self):
        return self.node(len(self.levels) - 1, 0)

    def generate_proof(self, index):
        """Path of leaf `index`: the sibling digest at every level, bottom up (leaf sibling first), as integers."""
        if not 0 <= index < len(self.levels[0]):
            raise IndexError("leaf index out of range")
        return [self.node(lvl, (index >> lvl) ^ 1) for lvl in range(len(self.levels) - 1)]

    def int_levels(self):
        return [[int.from_bytes(r.tobytes(), "little") for r in lvl] for lvl in self.levels]
