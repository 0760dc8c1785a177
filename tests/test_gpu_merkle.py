"""BASELINE config #5 stand-in on the GPU: the Pedersen-hash Merkle-membership circuit of simpleworks_amd/workloads.py
(statement and I/O convention of /root/reference/src/merkle_tree/{merkle_tree_verification_u8,simple_merkle_tree}.rs).
  * tiny instance: proof and verifying-key bytes equal to the pure-Python model's (tests/golden/marlin_merkle.json);
  * full size (tree height 19 = 2^18 leaves, 256-bit digests, + the block of simpleworks UInt8 gadget rows): is_satisfied,
    the mat-vecs of A, B and A^T and MSMs with this circuit's scalar distributions bit-exact against the C oracle,
    prove -> verify, tampered / wrong public input rejected, wrong root fails at prove time."""
import numpy as np
import pytest

from oracle_lib import Oracle, golden, h2i

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    from simpleworks_amd import marlin
    return marlin


@pytest.fixture(scope="module")
def S():
    from simpleworks_amd import serialization
    return serialization


@pytest.fixture(scope="module")
def W():
    from simpleworks_amd import workloads
    return workloads


@pytest.fixture(scope="module")
def full(W):
    cs, public, params = W.merkle_membership_circuit()  # height 19, gadget block on: the bench's `--circuit merkle`
    return cs, cs.pack(), public, params


def test_merkle_tiny_golden_bytes(M, S, W):
    case = golden("marlin_merkle.json")["merkle_tiny"]
    kw = case["circuit"]
    P = W.MerkleParams(kw["digest_bits"], kw["leaf_windows"], kw["inner_windows"], kw["window_size"], kw["seed"])
    levels = P.build_tree(kw["leaves"])
    cs = M.ConstraintSystem()
    public = W.build_merkle_membership(cs, P, kw["leaves"][kw["leaf_index"]], kw["leaf_index"],
                                       P.path_of(levels, kw["leaf_index"]), kw["gadget_byte_ops"])
    assert public == [h2i(x) for x in case["public_input"]]
    assert cs.is_satisfied()
    rng = M.generate_rand()
    srs = M.generate_universal_srs(*case["srs"], rng)
    assert srs.max_degree == case["max_degree"]
    pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
    assert S.serialize_verifying_key(vk).hex() == case["vk"]
    proof = M.generate_proof(cs, pk, rng)
    assert S.serialize_proof(proof).hex() == case["proof"]
    assert M.verify_proof(vk, public, proof, rng)
    pk.free()
    srs.free()


def test_merkle_real_tree_height5(M, S, W):
    """A real tree over 16 leaves with the reference's hash parameters (256-bit digests, 144 / 128 windows of 4): the
    root the circuit exposes is the root of the natively built tree; the proof verifies against [root, 8 LSB-first bits]
    (src/merkle_tree/simple_merkle_tree.rs:129-143) and against nothing else."""
    P = W.MerkleParams()
    leaves = [(37 * i + 11) & 0xFF for i in range(16)]
    levels = P.build_tree(leaves)
    idx = 9
    cs = M.ConstraintSystem()
    public = W.build_merkle_membership(cs, P, leaves[idx], idx, P.path_of(levels, idx), gadget_byte_ops=16)
    assert public[0] == levels[-1][0]
    assert public[1:] == [(leaves[idx] >> i) & 1 for i in range(8)]
    packed = cs.pack()
    assert packed.is_satisfied()
    nnz = max(int(m[0][-1]) for m in packed.mats)
    rng = M.generate_rand()
    srs = M.generate_universal_srs(cs.num_constraints, len(cs.instance) + len(cs.witness), nnz, rng)
    pk, vk = M.generate_proving_and_verifying_keys(srs, packed)
    proof = M.generate_proof(packed, pk, rng)
    assert M.verify_proof(vk, public, S.deserialize_proof(S.serialize_proof(proof)), M.generate_rand())
    other_leaf = [public[0]] + [(leaves[idx + 1] >> i) & 1 for i in range(8)]
    assert not M.verify_proof(vk, other_leaf, proof, M.generate_rand())
    assert not M.verify_proof(vk, [(public[0] + 1) % M.R_MODULUS] + public[1:], proof, M.generate_rand())
    # the same path against a wrong root: the final equality row fails, proving fails at prove time
    bad = M.ConstraintSystem()
    W.build_merkle_membership(bad, P, leaves[idx], idx, P.path_of(levels, idx), gadget_byte_ops=16, root=public[0] + 1)
    assert not bad.is_satisfied()
    with pytest.raises(M.MarlinError) as e:
        M.generate_proof(bad, pk, rng)
    assert e.value.code == -5
    pk.free()
    srs.free()


def _csr_transpose(rowptr, col, val, ncols):
    rows = len(rowptr) - 1
    row_of = np.repeat(np.arange(rows, dtype=np.uint32), np.diff(rowptr).astype(np.int64))
    order = np.argsort(col, kind="stable")
    tptr = np.zeros(ncols + 1, dtype=np.uint32)
    tptr[1:] = np.cumsum(np.bincount(col, minlength=ncols))
    return tptr, np.ascontiguousarray(row_of[order]), np.ascontiguousarray(val[order])


def test_merkle_full_shape_and_matvec(M, W, full):
    cs, packed, public, params = full
    nv = len(cs.instance) + len(cs.witness)
    assert cs.num_constraints > 65536 and nv < 131072            # |H| = 2^17
    nnz = [int(m[0][-1]) for m in packed.mats]
    assert 131072 < max(nnz) <= 262144                           # |K| = 2^18 != |H|
    assert sum(v in (0, 1) for v in cs.witness) * 2 >= len(cs.witness)
    empty_ab = int(np.sum((np.diff(packed.mats[0][0]) == 0) & (np.diff(packed.mats[1][0]) == 0) & (np.diff(packed.mats[2][0]) > 0)))
    assert empty_ab > 5000                                       # rows `0 * 0 = a - b`
    assert int(np.diff(packed.mats[0][0]).max()) >= 257          # bit-packing rows
    assert packed.is_satisfied()
    # K3 on this circuit's matrices: A z, B z and A^T r bit-exact against the C oracle
    ctx = M.default_context()
    orc = Oracle()
    z = np.ascontiguousarray(np.concatenate([packed.instance, packed.witness]))
    for rowptr, col, val in packed.mats[:2]:
        assert np.array_equal(ctx.spmv_fr(rowptr, col, val, z), orc.spmv(rowptr, col, val, z))
    from pyref.prng import fr_array
    r = orc.fr_to_mont(fr_array(cs.num_constraints, 99))
    tptr, tcol, tval = _csr_transpose(*packed.mats[0], nv)
    assert np.array_equal(ctx.spmv_fr(tptr, tcol, tval, r), orc.spmv(tptr, tcol, tval, r))
    # an unsatisfied witness is noticed
    w2 = packed.witness.copy()
    w2[len(w2) // 2, 0] ^= np.uint64(1)
    broken = M.PackedR1cs(packed.instance, w2, *packed.mats)
    assert not broken.is_satisfied()


def test_merkle_full_msm_scalar_shapes(M, full):
    """K1 with the two scalar distributions this circuit produces: the assignment z itself (>= 50 % zeros and ones: the
    oversized-bucket path) and the coefficients of the z_A polynomial (iNTT of A z: dense, what the prover commits)."""
    cs, packed, public, params = full
    ctx = M.default_context()
    orc = Oracle()
    n = 1 << 17
    G = orc.points_to_mont([tuple(h2i(v) for v in golden("g1.json")["generator"])])
    bases = orc.srs_bases(n, h2i(golden("msm.json")["tau"]), G)
    bh = ctx.srs_upload(bases)
    z = np.ascontiguousarray(np.concatenate([packed.instance, packed.witness]))
    zs = np.zeros((n, 4), dtype=np.uint64)
    zs[: z.shape[0]] = orc.fr_from_mont(z)
    rowptr, col, val = packed.mats[0]
    za = np.zeros((n, 4), dtype=np.uint64)
    za[: cs.num_constraints] = orc.spmv(rowptr, col, val, z)
    za_coeffs = orc.fr_from_mont(orc.ntt(za, 17, 1, 0, orc.lib.oracle_max_threads()))
    for sc in (zs, za_coeffs):
        xy, inf = ctx.g1_normalize(ctx.msm_g1(bh, sc))
        ref = orc.jac_to_affine_int(orc.msm(bases, sc, threads=orc.lib.oracle_max_threads()))
        got = None if inf else orc.points_from_mont(xy.reshape(1, 12))[0]
        assert got == ref
    bh.free()


def test_merkle_full_prove_verify(M, S, W, full):
    cs, packed, public, params = full
    nnz = max(int(m[0][-1]) for m in packed.mats)
    rng = M.generate_rand()
    srs = M.generate_universal_srs(cs.num_constraints, len(cs.instance) + len(cs.witness), nnz, rng)
    pk, vk = M.generate_proving_and_verifying_keys(srs, packed)
    srs.free()
    proof = M.generate_proof(packed, pk, rng)
    assert len(proof.data) == 951
    assert M.verify_proof(vk, public, proof, M.generate_rand())
    flipped = list(public)
    flipped[1 + 3] ^= 1                                          # another leaf value
    assert not M.verify_proof(vk, flipped, proof, M.generate_rand())
    t = bytearray(proof.data)
    t[100] ^= 0x01
    try:
        ok = M.verify_proof(vk, public, M.MarlinProof(bytes(t)), M.generate_rand())
    except M.MarlinError:
        ok = False                                               # the tampered commitment no longer decodes
    assert not ok
    # wrong root at full size
    bad, _, _ = W.merkle_membership_circuit(params=params, root=public[0] + 5)
    with pytest.raises(M.MarlinError) as e:
        M.generate_proof(bad, pk, rng)
    assert e.value.code == -5
    pk.free()


# ---- the reference's own call sequences, through MarlinInst with a ConstraintSynthesizer (what its callers actually hit)
def test_simple_merkle_tree_call_sequence_through_marlin_inst(M, W):
    """SimpleMerkleTree::{new, get_merkle_path, prove, verify} (/root/reference/src/merkle_tree/simple_merkle_tree.rs:35-153),
    call for call: universal_setup(100_000, 25_000, 300_000), keys indexed from a DUMMY circuit over a blank tree,
    MarlinInst::prove with the real circuit, proofs as bytes, verify(bytes, leaf) over [root, 8 LSB-first bits] — and its
    tests `:275-291` (a valid proof verifies) and `:224-273` (a wrong leaf does not)."""
    leaves = [3, 200, 77, 9, 0, 255, 16, 42]
    tree = W.SimpleMerkleTree(leaves)
    assert W.merkle_tree_height(len(leaves)) == 4
    path = tree.get_merkle_path(5)
    proof = tree.prove(leaves[5], path)
    assert isinstance(proof, bytes) and len(proof) > 900
    assert tree.verify(proof, leaves[5])
    assert not tree.verify(proof, leaves[5] ^ 1)          # another leaf value: rejected by the pairing check
    proof2 = tree.prove(leaves[0], tree.get_merkle_path(0))
    assert tree.verify(proof2, leaves[0]) and not tree.verify(proof2, leaves[5])
    with pytest.raises(M.MarlinError) as e:               # a leaf that is not at that position: fails at prove time
        tree.prove(leaves[5] ^ 0x10, path)
    assert e.value.code == -5
    tree.free()


def test_marlin_inst_with_a_synthesizer_matches_golden_bytes(M, S):
    """examples/manual-constraints.rs:86-100: MarlinInst::{universal_setup, index, prove, verify} with a circuit OBJECT
    (generate_constraints), rng passed along — the bytes of tests/golden/marlin.json."""
    case = golden("marlin.json")["manual_constraints"]

    class ManualConstraints:  # examples/manual-constraints.rs:15-31
        def __init__(self, a, b):
            self.a, self.b = a, b

        def generate_constraints(self, cs):
            a = cs.new_input_variable(self.a)
            b = cs.new_witness_variable(self.b)
            cs.enforce_constraint([(1, a), (M.R_MODULUS - 1, b)], [(1, cs.one())], [])

    rng = M.generate_rand()
    universal_srs = M.MarlinInst.universal_setup(100, 25, 300, rng)
    circuit = ManualConstraints(1, 1)
    index_pk, index_vk = M.MarlinInst.index(universal_srs, circuit)
    proof = M.MarlinInst.prove(index_pk, circuit, rng)
    assert S.serialize_verifying_key(index_vk).hex() == case["vk"]
    assert S.serialize_proof(proof).hex() == case["proof"]
    assert M.MarlinInst.verify(index_vk, [1], proof, rng)
    # the fork's entry points on an already synthesised system give the same bytes
    rng = M.generate_rand()
    srs2 = M.MarlinInst.universal_setup(100, 25, 300, rng)
    cs = M.ConstraintSystem()
    circuit.generate_constraints(cs)
    pk2, vk2 = M.MarlinInst.index_from_constraint_system(srs2, cs)
    assert S.serialize_proof(M.MarlinInst.prove_from_constraint_system(pk2, cs, rng)).hex() == case["proof"]
    for h in (index_pk, pk2, universal_srs, srs2):
        h.free()


def test_simple_merkle_tree_golden_bytes(M, S, W):
    """SimpleMerkleTree::new + ::prove with the reference's own sizes and order of draws — ONE test_rng for
    universal_setup(100_000, 25_000, 300_000), LeafHash::setup, TwoToOneHash::setup; the tree built natively; keys from the dummy
    circuit; a fresh test_rng for the proof (src/merkle_tree/simple_merkle_tree.rs:35-127) — against the same sequence run
    through the Python model (tests/golden/gen_golden_large.py simple_merkle_tree): same generators, same root and path (GPU
    tree vs the model's), same verifying-key bytes, same proof bytes."""
    import hashlib
    case = golden("marlin_large.json")["simple_merkle_tree"]
    tree = W.SimpleMerkleTree(case["leaves"], srs_sizes=tuple(case["srs"]))
    raw = lambda gens: b"".join(x.to_bytes(32, "little") + y.to_bytes(32, "little") for row in gens for x, y in row)
    assert hashlib.sha256(raw(tree.params.leaf_gens)).hexdigest() == case["leaf_generators_sha256"]
    assert hashlib.sha256(raw(tree.params.inner_gens)).hexdigest() == case["two_to_one_generators_sha256"]
    assert tree.root() == h2i(case["root"])
    idx, path = tree.get_merkle_path(case["index"])
    assert path == [h2i(v) for v in case["path"]]
    assert S.serialize_verifying_key(tree.verifying_key).hex() == case["vk"]
    proof = tree.prove(case["leaves"][case["index"]], (idx, path))
    assert proof.hex() == case["proof"]
    assert tree.verify(proof, case["leaves"][case["index"]]) and not tree.verify(proof, case["leaves"][0])
    tree.free()


def test_simple_merkle_tree_config5_2p18_leaves(M, W):
    """BASELINE configs[4] end to end, the reference's call sequence at its size (examples/merkle-tree, 2^18 leaves):
    SimpleMerkleTree::new — universal_setup(100_000, 25_000, 300_000), both CRH setups, the tree over 2^18 u8 leaves (on the GPU),
    keys from the dummy circuit of height 19 — then get_merkle_path, prove, verify (src/merkle_tree/simple_merkle_tree.rs:35-153)."""
    import time
    rng = np.random.default_rng(2018)
    leaves = [int(v) for v in rng.integers(0, 256, size=1 << 18)]
    t0 = time.time()
    tree = W.SimpleMerkleTree(leaves)
    t1 = time.time()
    assert W.merkle_tree_height(len(leaves)) == 19 and len(tree.levels) == 19
    i = 0x2B3C5
    path = tree.get_merkle_path(i)
    assert tree.params.root_from_path(leaves[i], *path) == tree.root()      # the path of the GPU tree folds to its root
    proof = tree.prove(leaves[i], path)
    t2 = time.time()
    assert tree.verify(proof, leaves[i])
    assert not tree.verify(proof, leaves[i] ^ 0x10)
    other = (i + 12345) % len(leaves)
    with pytest.raises(M.MarlinError):                                      # a path that is not this leaf's: unsatisfied
        tree.prove(leaves[i] ^ 1, path)
    assert tree.verify(tree.prove(leaves[other], tree.get_merkle_path(other)), leaves[other])
    print("SimpleMerkleTree 2^18 leaves: new %.1f s (incl. two circuit syntheses in Python), prove %.1f s" % (t1 - t0, t2 - t1))
    tree.free()
