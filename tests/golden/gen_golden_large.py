#!/usr/bin/env python3
"""Golden Marlin proofs at REAL sizes (BASELINE config #2: "2^16 ... bit-exact vs CPU"), written to
tests/golden/marlin_large.json.  Run from the repo root, build container only (minutes of CPU):

    python tests/golden/gen_golden_large.py [case ...]

The prover is the independent Python model (oracle/pyref/marlin.py: setup, index, prove, verify, codecs — the same code
that produced tests/golden/marlin.json).  What would take days in pure Python is handed to the C restatement of the
arkworks kernels (oracle/oracle.c: VariableBaseMSM, Radix2EvaluationDomain FFTs, the fixed-base SRS powers), both of
them checker code: nothing of the product (simpleworks_amd/, libswmarlin.so) runs here.  The accelerated pipeline is
first replayed on the small cases of marlin.json with EVERY transform and MSM routed through the C kernels and must
reproduce their proof and verifying-key bytes exactly; only then are the large cases generated.

Cases: the synthetic circuit of bench.py at 2^12 and 2^16 constraints, and the Pedersen-Merkle membership circuit with
the reference's hash shape (144 / 128 windows of 4 bits, 256-bit digests; src/merkle_tree/common.rs:11-52) at height 5.
Nothing here reads /root/reference.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from pyref import bls12_377 as bls  # noqa: E402
from pyref import marlin as M  # noqa: E402
from pyref import poly as P  # noqa: E402
from pyref.bls12_377 import R  # noqa: E402
from pyref.prng import Xoshiro256ss  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

ORC = Oracle()
THREADS = max(1, min(os.cpu_count() or 1, ORC.lib.oracle_max_threads()))
NTT_MIN = 256       # transforms below this size stay in Python (set to 1 for the self-check)
MSM_MIN = 64


def to_limbs(vals, nl):
    w = 8 * nl
    return np.frombuffer(b"".join(int(v).to_bytes(w, "little") for v in vals), dtype=np.uint64).reshape(-1, nl).copy()


def from_limbs(arr):
    arr = np.ascontiguousarray(arr)
    w = 8 * arr.shape[1]
    b = arr.tobytes()
    return [int.from_bytes(b[i:i + w], "little") for i in range(0, len(b), w)]


class MontPoints:
    """A sequence of G1 affine points held as an (n, 12) Montgomery limb array (what oracle_msm_g1 consumes).  Slices stay
    arrays; single points are converted to the model's (x, y) integers on demand."""

    def __init__(self, arr):
        self.arr = arr

    def __len__(self):
        return self.arr.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return MontPoints(self.arr[i])
        return ORC.points_from_mont(np.ascontiguousarray(self.arr[i]).reshape(1, 12))[0]


# ---- accelerators (the Python originals stay reachable for the self-check)
_py_msm = M._msm
_py_fft, _py_ifft, _py_cfft, _py_cifft = P.Domain.fft, P.Domain.ifft, P.Domain.coset_fft, P.Domain.coset_ifft
_py_setup = M.universal_setup


def fast_msm(bases, scalars):
    n = len(scalars)
    if not isinstance(bases, MontPoints):
        return _py_msm(bases, scalars)
    if n == 0:
        return None
    if n < MSM_MIN:
        return _py_msm([bases[i] for i in range(n)], scalars)
    jac = ORC.msm(np.ascontiguousarray(bases.arr[:n]), to_limbs([s % R for s in scalars], 4), threads=THREADS)
    return ORC.jac_to_affine_int(jac)


def _c_ntt(dom, vals, inverse, coset, fallback):
    if dom.size < NTT_MIN:
        return fallback(dom, vals)
    vals = list(vals)
    assert len(vals) <= dom.size
    a = [v % R for v in vals] + [0] * (dom.size - len(vals))
    out = ORC.ntt(ORC.fr_to_mont(to_limbs(a, 4)), dom.log, inverse, coset, THREADS)
    return from_limbs(ORC.fr_from_mont(out))


def fast_setup(num_constraints, num_variables, num_non_zero, rng):
    """M.universal_setup with the same draws in the same order; [beta^i] g from the C fixed-base routine.  Only the
    first four powers of gamma_g are materialised (trim keeps hiding_bound + 2 = 3 of them)."""
    max_degree = M.ahp_max_degree(num_constraints, num_variables, num_non_zero)
    if max_degree < 1:
        raise M.MarlinError("DegreeIsZero")
    beta = rng.rand_fr()
    g = M._g1_rand(rng)
    gamma_g = M._g1_rand(rng)
    h = M._g2_rand(rng)
    powers = MontPoints(ORC.srs_bases(max_degree + 1, beta, ORC.points_to_mont([g])))
    pg = [gamma_g]
    for _ in range(3):
        pg.append(bls.g1_mul_fast(pg[-1], beta))
    return M.UniversalSRS(powers, pg, h, bls.g2_mul(h, beta))


def install():
    M._msm = fast_msm
    M.universal_setup = fast_setup
    P.Domain.fft = lambda self, c: _c_ntt(self, c, 0, 0, _py_fft)
    P.Domain.ifft = lambda self, e: _c_ntt(self, e, 1, 0, _py_ifft)
    P.Domain.coset_fft = lambda self, c: _c_ntt(self, c, 0, 1, _py_cfft)
    P.Domain.coset_ifft = lambda self, e: _c_ntt(self, e, 1, 1, _py_cifft)


def run_case(cs, srs_sizes, public, label):
    t0 = time.time()
    rng = M.generate_rand()
    srs = M.generate_universal_srs(*srs_sizes, rng)
    t1 = time.time()
    pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
    t2 = time.time()
    trace = {}
    proof = M.prove(pk, cs, rng, trace)
    t3 = time.time()
    pbytes = M.serialize_proof(proof)
    vbytes = M.serialize_verifying_key(vk)
    assert M.verify_proof(vk, public, M.deserialize_proof(pbytes), rng), "the model's own verifier rejects the proof"
    print(" %s: setup %.0fs, index %.0fs, prove %.0fs, verify %.0fs; proof %d B" %
          (label, t1 - t0, t2 - t1, t3 - t2, time.time() - t3, len(pbytes)), flush=True)
    return {"srs": list(srs_sizes), "max_degree": srs.max_degree, "public_input": [hex(x) for x in public],
            "num_constraints": vk["num_constraints"], "num_variables": vk["num_variables"],
            "num_non_zero": vk["num_non_zero"],
            "challenges": {k: hex(trace[k]) for k in ("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma", "xi")},
            "proof": pbytes.hex(), "vk": vbytes.hex(),
            "proof_sha256": hashlib.sha256(pbytes).hexdigest(), "vk_sha256": hashlib.sha256(vbytes).hexdigest()}


def self_check():
    """Every transform and MSM through the C kernels on the small committed cases: same bytes as the pure-Python run."""
    global NTT_MIN, MSM_MIN
    keep = NTT_MIN, MSM_MIN
    NTT_MIN, MSM_MIN = 1, 1
    small = json.load(open(os.path.join(HERE, "marlin.json")))
    for name in ("synthetic_32", "random_sparse", "manual_constraints"):
        case = small[name]
        if name == "manual_constraints":
            cs = M.manual_constraints_circuit(1, 1)
        elif name == "random_sparse":
            cs = M.random_sparse_circuit(**case["circuit"])
        else:
            cs = M.synthetic_circuit(case["num_constraints"], int(case["a"], 16), int(case["b"], 16))
        got = run_case(cs, tuple(case["srs"]), [int(x, 16) for x in case["public_input"]], "self-check " + name)
        assert got["proof"] == case["proof"] and got["vk"] == case["vk"], "accelerated pipeline differs from the model: " + name
    NTT_MIN, MSM_MIN = keep


def case_synthetic(log_n):
    n = 1 << log_n
    g = Xoshiro256ss(1000 + log_n)
    a, b = g.fr(), g.fr()
    cs = M.synthetic_circuit(n, a, b)
    out = run_case(cs, (n, n, n), cs.instance[1:], "synthetic 2^%d" % log_n)
    out["a"], out["b"] = hex(a), hex(b)
    return out


MERKLE_H5 = dict(height=5, leaf_u8=0xA7, seed=7, gadget_byte_ops=96)


def case_merkle_h5():
    from simpleworks_amd import workloads as W
    kw = MERKLE_H5
    # the circuit DESCRIPTION is host logic shared with the product (as in gen_golden.py: merkle_tiny_system); the
    # prover that turns it into proof bytes is the independent model
    g = W._SplitMix(kw["seed"])
    levels = kw["height"] - 1
    siblings = [g.fr() for _ in range(levels)]
    leaf_index = g.next_u64() % (1 << levels)
    params = W.MerkleParams()
    cs = M.ConstraintSystem()
    public = W.build_merkle_membership(cs, params, kw["leaf_u8"], leaf_index, siblings, kw["gadget_byte_ops"])
    assert cs.is_satisfied()
    a_m, b_m, c_m = cs.to_matrices()
    nnz = max(sum(len(r) for r in m) for m in (a_m, b_m, c_m))
    nv = len(cs.instance) + len(cs.witness)
    out = run_case(cs, (cs.num_constraints, nv, nnz), public, "merkle height 5")
    out["circuit"] = kw
    return out


def case_test_circuit():
    """examples/test-circuit.rs:72-81 (BASELINE configs[0]): universal_setup(100, 25, 300), index, prove, verify(&[]) with ONE
    test_rng; the circuit description is host logic shared with the product (workloads.build_test_circuit)."""
    from simpleworks_amd import workloads as W
    cs = M.ConstraintSystem()
    W.build_test_circuit(cs, 1, 1)
    assert cs.is_satisfied() and cs.num_constraints == 24 and len(cs.instance) == 1
    bad = M.ConstraintSystem()
    W.build_test_circuit(bad, 1, 2)
    assert not bad.is_satisfied()          # :52-61 different_values_should_fail
    return run_case(cs, (100, 25, 300), [], "test-circuit")


SIMPLE_MERKLE_TREE = dict(leaves=[3, 200, 77, 9, 0, 255, 16, 42], index=5, srs=[100_000, 25_000, 300_000])


def case_simple_merkle_tree():
    """SimpleMerkleTree::new followed by ::prove, call for call (src/merkle_tree/simple_merkle_tree.rs:35-127), in the model:
    ONE test_rng for universal_setup(100_000, 25_000, 300_000), LeafHash::setup and TwoToOneHash::setup (:38-45); the tree
    over the leaves (:47-49); keys indexed from the DUMMY circuit over a blank path (:59-83); the proof of leaf `index` with
    a FRESH test_rng (:116-119).  The circuit description is host logic shared with the product (as in case_merkle_h5)."""
    from pyref import pedersen as PP
    from simpleworks_amd import workloads as W
    kw = SIMPLE_MERKLE_TREE
    t0 = time.time()
    rng = M.generate_rand()
    srs = M.generate_universal_srs(*kw["srs"], rng)
    leaf = PP.pedersen_setup(rng, PP.LEAF_WINDOWS)
    inner = PP.pedersen_setup(rng, PP.TWO_TO_ONE_WINDOWS)
    params = W.MerkleParams(generators=(leaf, inner))
    levels = PP.merkle_tree(leaf, inner, [bytes([v]) for v in kw["leaves"]])
    root, path = levels[-1][0], PP.merkle_path(levels, kw["index"])
    height = W.merkle_tree_height(len(kw["leaves"]))
    blank_path = [0] * (height - 1)
    blank_root = params.root_from_path(0, 0, blank_path)
    dummy = M.ConstraintSystem()
    W.build_merkle_membership(dummy, params, 0, 0, blank_path, 0, blank_root)
    pk, vk = M.generate_proving_and_verifying_keys(srs, dummy)
    t1 = time.time()
    real = M.ConstraintSystem()
    public = W.build_merkle_membership(real, params, kw["leaves"][kw["index"]], kw["index"], path, 0, root)
    assert real.is_satisfied() and public[0] == root
    proof = M.prove(pk, real, M.generate_rand(), {})
    pbytes, vbytes = M.serialize_proof(proof), M.serialize_verifying_key(vk)
    assert M.verify_proof(vk, public, M.deserialize_proof(pbytes), M.generate_rand())
    assert not M.verify_proof(vk, [root] + [(kw["leaves"][0] >> i) & 1 for i in range(8)], M.deserialize_proof(pbytes), M.generate_rand())
    print(" simple merkle tree: setup + index %.0fs, prove + verify %.0fs; %d constraints" %
          (t1 - t0, time.time() - t1, vk["num_constraints"]), flush=True)
    raw = lambda gens: b"".join(x.to_bytes(32, "little") + y.to_bytes(32, "little") for row in gens for x, y in row)
    return dict(kw, max_degree=srs.max_degree, root=hex(root), path=[hex(v) for v in path],
                leaf_generators_sha256=hashlib.sha256(raw(leaf)).hexdigest(),
                two_to_one_generators_sha256=hashlib.sha256(raw(inner)).hexdigest(),
                num_constraints=vk["num_constraints"], num_variables=vk["num_variables"], num_non_zero=vk["num_non_zero"],
                proof=pbytes.hex(), vk_sha256=hashlib.sha256(vbytes).hexdigest(), vk=vbytes.hex())


CASES = {"synthetic_2p12": lambda: case_synthetic(12), "synthetic_2p16": lambda: case_synthetic(16),
         "merkle_h5": case_merkle_h5}
# not part of the default run (tens of minutes and several GB of Python integers): python gen_golden_large.py synthetic_2p18
EXTRA_CASES = {"synthetic_2p14": lambda: case_synthetic(14), "synthetic_2p17": lambda: case_synthetic(17),
               "synthetic_2p18": lambda: case_synthetic(18), "synthetic_2p19": lambda: case_synthetic(19), "synthetic_2p20": lambda: case_synthetic(20),
               "simple_merkle_tree": case_simple_merkle_tree, "test_circuit": case_test_circuit}

if __name__ == "__main__":
    install()
    which = sys.argv[1:] or list(CASES)
    path = os.path.join(HERE, "marlin_large.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    self_check()
    for name in which:
        out[name] = (CASES.get(name) or EXTRA_CASES[name])()
        with open(path, "w") as f:
            json.dump(out, f, indent=0, separators=(",", ":"))
        print("wrote", name, flush=True)
