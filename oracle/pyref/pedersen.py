"""Pure-Python restatement of the native Pedersen hash and Merkle tree of simpleworks (oracle; TEST INFRASTRUCTURE ONLY).

What the reference calls (ark-crypto-primitives 0.3 / ark-ec 0.3 / ark-ed-on-bls12-377 0.3, none vendored in /root/reference;
behaviour restated from their published sources, [U] = not checkable against the crates here):
    /root/reference/src/hash/mod.rs:23-28                         pedersen_hash(input): setup from a fresh test_rng, evaluate
    /root/reference/src/merkle_tree/simple_merkle_tree.rs:43-49   LeafHash::setup, TwoToOneHash::setup, MerkleTree::new
    /root/reference/src/merkle_tree/common.rs:11-30               windows 144 x 4 (leaves), 128 x 4 (two digests)
    /root/reference/examples/merkle-tree/main.rs:103-121          the 8-leaf tree of its test, proof for index 4

Parity status: the reference holds no golden digest or root for these calls (its tests only check that the circuit is
satisfied), so this model is pinned by algebra only: points on the curve and in the prime-order subgroup, the independent C
restatement in oracle.c (per-bit additions in extended coordinates) and the GPU's tabulated-window evaluation all agree.
"""
from .bls12_377 import R

ED_A = R - 1
ED_D = 3021
ED_COFACTOR = 4
ED_SUBGROUP_ORDER = 2111115437357092606062206234695386632838870926408408195193685246394721360383
LEAF_WINDOWS, TWO_TO_ONE_WINDOWS, WINDOW_SIZE = 144, 128, 4


def ed_add(p, q):
    """ark-ec GroupAffine add for a twisted Edwards curve (the unified law; complete here: a = -1 is a square, d is not)."""
    x1, y1 = p
    x2, y2 = q
    k = ED_D * x1 * x2 * y1 * y2 % R
    return ((x1 * y2 + y1 * x2) * pow(1 + k, -1, R) % R, (y1 * y2 - ED_A * x1 * x2) * pow(1 - k, -1, R) % R)


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
    return (ED_A * x * x + y * y - 1 - ED_D * x * x * y * y) % R == 0


def fr_sqrt(v):
    """Some square root, or None (ark-ff sqrt is Tonelli-Shanks; which root it returns does not matter below)."""
    v %= R
    if v == 0:
        return 0
    if pow(v, (R - 1) // 2, R) != 1:
        return None
    q = (R - 1) >> 47
    m, c, t, r = 47, pow(22, q, R), pow(v, q, R), pow(v, (q + 1) // 2, R)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % R
            i += 1
        b = pow(c, 1 << (m - i - 1), R)
        m, c = i, b * b % R
        t, r = t * c % R, r * b % R
    return r


def get_point_from_x(x, greatest):
    """ark-ec 0.3 twisted_edwards_extended GroupAffine::get_point_from_x [U]: y^2 = (a x^2 - 1) / (d x^2 - 1);
    y = if (y < -y) ^ greatest { y } else { -y }, compared as integers in standard form."""
    x2 = x * x % R
    den = (ED_D * x2 - 1) % R
    if den == 0:
        return None
    y = fr_sqrt((ED_A * x2 - 1) * pow(den, -1, R))
    if y is None:
        return None
    negy = (-y) % R
    return (x, y if (y < negy) ^ greatest else negy)


def ed_rand(rng):
    """Distribution<GroupProjective<P>> for Standard [U]: loop { x = BaseField::rand(rng); greatest = rng.gen();
    if let Some(p) = get_point_from_x(x, greatest) { return p.scale_by_cofactor() } }"""
    while True:
        x = rng.rand_fr()
        greatest = rng.gen_bool()
        p = get_point_from_x(x, greatest)
        if p is not None:
            return ed_mul(p, ED_COFACTOR)


def pedersen_setup(rng, num_windows, window_size=WINDOW_SIZE):
    """pedersen::CRH::setup -> create_generators [U]: per window `base = C::rand(rng)`, then window_size times
    { push(base); base.double_in_place() }."""
    gens = []
    for _ in range(num_windows):
        base = ed_rand(rng)
        row = []
        for _ in range(window_size):
            row.append(base)
            base = ed_add(base, base)
        gens.append(row)
    return gens


def pedersen_evaluate(gens, data):
    """CRH::evaluate + TECompressor: bytes -> bits LSB-first, zero-padded to the window grid; per window the generators of
    the set bits are added; the x coordinate of the sum of the windows."""
    ws = len(gens[0])
    if len(data) * 8 > ws * len(gens):
        raise ValueError("incorrect input length")
    bits = [(byte >> i) & 1 for byte in data for i in range(8)]
    bits += [0] * (ws * len(gens) - len(bits))
    total = (0, 1)
    for w, row in enumerate(gens):
        encoded = (0, 1)
        for j in range(ws):
            if bits[w * ws + j]:
                encoded = ed_add(encoded, row[j])
        total = ed_add(total, encoded)
    return total[0]


def merkle_tree(leaf_gens, inner_gens, leaves):
    """MerkleTree::new: levels bottom-up; leaves are byte strings (to_bytes![leaf]); an inner node hashes the 32-byte
    little-endian encodings of its children, left then right."""
    assert len(leaves) >= 2 and len(leaves) & (len(leaves) - 1) == 0
    levels = [[pedersen_evaluate(leaf_gens, bytes(l)) for l in leaves]]
    while len(levels[-1]) > 1:
        prev = levels[-1]
        levels.append([pedersen_evaluate(inner_gens, prev[2 * i].to_bytes(32, "little") + prev[2 * i + 1].to_bytes(32, "little"))
                       for i in range(len(prev) // 2)])
    return levels


def merkle_path(levels, index):
    return [levels[l][(index >> l) ^ 1] for l in range(len(levels) - 1)]


def root_from_path(leaf_gens, inner_gens, leaf, index, path):
    cur = pedersen_evaluate(leaf_gens, bytes(leaf))
    for l, sib in enumerate(path):
        a, b = (sib, cur) if (index >> l) & 1 else (cur, sib)
        cur = pedersen_evaluate(inner_gens, a.to_bytes(32, "little") + b.to_bytes(32, "little"))
    return cur
