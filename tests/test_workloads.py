"""Host logic of the BASELINE workloads (CPU): the Pedersen-Merkle membership circuit (config #5 stand-in) against a
native model of the tree and against the pure-Python constraint system of the oracle."""
import json
import os

from oracle_lib import GOLDEN, h2i
from pyref import marlin as PM

from simpleworks_amd import workloads as W
from simpleworks_amd.marlin import R_MODULUS


def _tiny():
    with open(os.path.join(GOLDEN, "marlin_merkle.json")) as f:
        case = json.load(f)["merkle_tiny"]
    kw = case["circuit"]
    P = W.MerkleParams(kw["digest_bits"], kw["leaf_windows"], kw["inner_windows"], kw["window_size"], kw["seed"])
    return case, kw, P


def test_edwards_parameters():
    assert W.ed_on_curve(W.ED_GENERATOR)
    assert W.ed_mul(W.ED_GENERATOR, W.ED_SUBGROUP_ORDER) == (0, 1)
    gens = W.pedersen_generators(3, 4, seed=5)
    for row in gens:
        assert all(W.ed_on_curve(p) for p in row)
        assert W.ed_mul(row[0], W.ED_SUBGROUP_ORDER) == (0, 1) and row[0] != (0, 1)
        assert row[3] == W.ed_mul(row[0], 8)


def test_merkle_tiny_matches_native_tree_and_io_convention():
    case, kw, P = _tiny()
    levels = P.build_tree(kw["leaves"])
    idx = kw["leaf_index"]
    cs = PM.ConstraintSystem()
    public = W.build_merkle_membership(cs, P, kw["leaves"][idx], idx, P.path_of(levels, idx), kw["gadget_byte_ops"])
    # public input = [root, 8 bits of the u8 leaf, least significant first] (simple_merkle_tree.rs:129-143)
    assert public == [levels[-1][0]] + [(kw["leaves"][idx] >> i) & 1 for i in range(8)]
    assert public == [h2i(x) for x in case["public_input"]]
    assert cs.instance == [1] + public
    assert cs.is_satisfied()
    assert cs.num_constraints == case["srs"][0]
    # every other leaf index gives another path but the same root
    for j in range(len(kw["leaves"])):
        assert P.root_from_path(kw["leaves"][j], j, P.path_of(levels, j)) == levels[-1][0]
    # wrong root, wrong leaf value, wrong direction bits: unsatisfied
    for bad_kw in (dict(root=public[0] + 1),):
        bad = PM.ConstraintSystem()
        W.build_merkle_membership(bad, P, kw["leaves"][idx], idx, P.path_of(levels, idx), 0, **bad_kw)
        assert not bad.is_satisfied()
    bad = PM.ConstraintSystem()
    W.build_merkle_membership(bad, P, kw["leaves"][idx] ^ 4, idx, P.path_of(levels, idx), 0, root=public[0])
    assert not bad.is_satisfied()
    bad = PM.ConstraintSystem()
    W.build_merkle_membership(bad, P, kw["leaves"][idx], idx ^ 1, P.path_of(levels, idx), 0, root=public[0])
    assert not bad.is_satisfied()


def test_merkle_full_shape():
    """The shape BASELINE config #5 gives the prover (SURVEY.md §8d estimated 5-7 x 10^4 constraints for height 19)."""
    cs, public, P = W.merkle_membership_circuit(gadget_byte_ops=0)
    assert 60000 < cs.num_constraints < 65536 and len(cs.witness) + len(cs.instance) < 65536   # |H| = 2^16
    nnz = [sum(len(r) for r in m) for m in cs.rows]
    assert 65536 < max(nnz) <= 131072                                                          # |K| = 2^17
    assert len(public) == 9 and all(b in (0, 1) for b in public[1:])
    cs2, _, _ = W.merkle_membership_circuit(params=P)                                          # + UInt8 gadget block
    assert 65536 < cs2.num_constraints < 131072
    assert sum(v in (0, 1) for v in cs2.witness) * 2 >= len(cs2.witness)
    assert sum(1 for a, b, c in zip(*cs2.rows) if not a and not b and c) > 5000
    assert max(len(r) for r in cs2.rows[0]) >= 257
    # satisfied (pure Python evaluation of all rows)
    for cs_ in (cs, cs2):
        def ev(lc):
            return sum(c * (cs_.instance[k] if kind == "i" else cs_.witness[k]) for c, (kind, k) in lc) % R_MODULUS
        assert all(ev(a) * ev(b) % R_MODULUS == ev(c) for a, b, c in zip(*cs_.rows))


# ---- SWMR1CS1: the flat dump a Rust-side caller writes (swmarlin_sys::r1cs_dump) and `bench.py --r1cs` reads
def test_r1cs_dump_round_trip_and_rejections(tmp_path):
    import numpy as np
    import pytest
    cs, public = W.synthetic_r1cs(1 << 12, 0x1234, 0x5678)
    path = str(tmp_path / "synthetic_2p12.r1cs")
    size = W.dump_r1cs(cs, path)
    assert size == os.path.getsize(path)
    back, public2 = W.load_r1cs(path)
    assert public2 == [int(x) for x in public]
    assert back.num_constraints == cs.num_constraints
    assert np.array_equal(back.instance, cs.instance) and np.array_equal(back.witness, cs.witness)
    for m0, m1 in zip(cs.mats, back.mats):
        assert all(np.array_equal(x, y) for x, y in zip(m0, m1))
    # a model constraint system with multi-term rows and |K| != |H| (the matrices as ark-relations' to_matrices returns them)
    model = PM.random_sparse_circuit(seed=20261002)
    packed = W.pack_model_system(model)
    W.dump_r1cs(packed, path)
    back, public2 = W.load_r1cs(path)
    assert public2 == model.instance[1:]
    for m0, m1 in zip(packed.mats, back.mats):
        assert all(np.array_equal(x, y) for x, y in zip(m0, m1))
    # every inconsistency is a ValueError, never a wrong system
    data = bytearray(open(path, "rb").read())

    def refuse(mutate):
        d = bytearray(data)
        mutate(d)
        bad = str(tmp_path / "bad.r1cs")
        with open(bad, "wb") as f:
            f.write(d)
        with pytest.raises(ValueError):
            W.load_r1cs(bad)
    refuse(lambda d: d.__setitem__(0, ord("X")))                       # magic
    refuse(lambda d: d.__setitem__(100, d[100] ^ 1))                   # a flipped bit: checksum
    refuse(lambda d: d.__delitem__(slice(len(d) - 40, len(d) - 32)))   # truncated body
    refuse(lambda d: d.__delitem__(slice(64, len(d))))                 # header only

    def resealed(mutate):   # a file whose checksum is right but whose content is not
        import hashlib

        def f(d):
            del d[-32:]
            mutate(d)
            d += hashlib.blake2s(bytes(d)).digest()
        return f
    refuse(resealed(lambda d: d.__setitem__(slice(8, 16), (0).to_bytes(8, "little"))))              # no instance variable
    refuse(resealed(lambda d: d.__setitem__(slice(32, 40), (1 << 40).to_bytes(8, "little"))))       # implausible nnz
    refuse(resealed(lambda d: d.__setitem__(slice(64, 96), b"\xff" * 32)))                           # non-canonical field element
    refuse(resealed(lambda d: d.__setitem__(slice(56, 64), (3).to_bytes(8, "little"))))              # unknown flags
