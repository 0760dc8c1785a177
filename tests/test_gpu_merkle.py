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
