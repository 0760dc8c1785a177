#!/usr/bin/env python3
"""Writes tests/golden/pin_circuits.json: the small circuits behind tests/golden/marlin.json and pk_bytes.json as plain data —
instance and witness assignments and every constraint's three linear combinations — so that a program WITHOUT the Python
builders (swmarlin-sys/tests/pin_golden.rs: arkworks' own CPU prover, no GPU, no libswmarlin) can replay them into an
ark-relations ConstraintSystem and compare what arkworks emits with the committed golden bytes.
Variables: ["i", k] = instance variable k (k = 0 is the constant one), ["w", k] = witness variable k; numbers are hex strings.
Run from the repo root: python3 tests/golden/gen_pin_circuits.py

  python3 tests/golden/gen_pin_circuits.py --r1cs-dir DIR
additionally writes the circuits that are too large for a JSON fixture as SWMR1CS1 files (simpleworks_amd/workloads.py:
dump_r1cs; read on the Rust side by swmarlin_sys::r1cs_dump::R1csFile) into DIR — synthetic_2p12.r1cs, merkle_h5.r1cs,
test_circuit.r1cs, the circuits behind tests/golden/marlin_large.json and marlin_merkle.json — for the pin kit
(SWM_PIN_R1CS_DIR=DIR cargo test ... --test pin_golden) and for `bench.py --r1cs`.  Pure Python: no GPU, no library.

Also writes tests/golden/tobytes.json: arkworks' ToBytes layouts ([U] in SURVEY A.8) as the model emits them, for the pin kit's
per-item tests."""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyref import marlin as M
from pyref.prng import Xoshiro256ss


def dump_cs(cs):
    lc = lambda terms: [[hex(c % M.R), [v[0], v[1]]] for c, v in terms]
    return {"instance": [hex(v) for v in cs.instance], "witness": [hex(v) for v in cs.witness],
            "constraints": [[lc(a), lc(b), lc(c)] for a, b, c in zip(cs.a, cs.b, cs.c)]}


def main():
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "marlin.json")))
    out = {"marlin": {}, "pk_bytes": {}}
    out["marlin"]["manual_constraints"] = dump_cs(M.manual_constraints_circuit(1, 1))
    g = Xoshiro256ss(5)
    for n in (8, 16, 32):
        a, b = g.fr(), g.fr()
        assert hex(a) == golden["synthetic_%d" % n]["a"] and hex(b) == golden["synthetic_%d" % n]["b"]
        out["marlin"]["synthetic_%d" % n] = dump_cs(M.synthetic_circuit(n, a, b))
    for name in ("random_sparse", "random_tall"):
        out["marlin"][name] = dump_cs(M.random_sparse_circuit(**golden[name]["circuit"]))
    for name, cs in out["marlin"].items():   # the dump is the system the golden bytes were made from
        assert [int(x, 16) for x in cs["instance"][1:]] == [int(x, 16) for x in golden[name]["public_input"]], name
    # tests/golden/pk_bytes.json (gen_golden.py gen_pk_bytes): fresh test_rng per case, synthetic_8 with a = 3, b = 5
    out["pk_bytes"]["manual_constraints"] = out["marlin"]["manual_constraints"]
    out["pk_bytes"]["synthetic_8"] = dump_cs(M.synthetic_circuit(8, 3, 5))
    out["pk_bytes"]["random_sparse"] = out["marlin"]["random_sparse"]
    # tests/golden/marlin_merkle.json "test_circuit" (examples/test-circuit.rs, BASELINE configs[0]): 24 constraints
    sys.path.insert(0, ROOT)
    from simpleworks_amd import workloads as W   # the circuit DESCRIPTIONS are host logic shared with the product
    tc = M.ConstraintSystem()
    W.build_test_circuit(tc, 1, 1)
    out["marlin_merkle"] = {"test_circuit": dump_cs(tc)}
    with open(os.path.join(ROOT, "tests", "golden", "pin_circuits.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote tests/golden/pin_circuits.json:", {k: len(v["constraints"]) for k, v in out["marlin"].items()})
    tobytes()
    if "--r1cs-dir" in sys.argv:
        r1cs_files(sys.argv[sys.argv.index("--r1cs-dir") + 1], tc)


def tobytes():
    """ark-ff / ark-ec / ark-poly-commit ToBytes layouts as the model writes them into the Fiat-Shamir transcript."""
    small = json.load(open(os.path.join(ROOT, "tests", "golden", "marlin.json")))["manual_constraints"]
    g1 = json.load(open(os.path.join(ROOT, "tests", "golden", "g1.json")))
    G = tuple(int(v, 16) for v in g1["generator"])
    rng = M.generate_rand()
    srs = M.generate_universal_srs(*small["srs"], rng)
    pk, vk = M.generate_proving_and_verifying_keys(srs, M.manual_constraints_circuit(1, 1))
    assert M.serialize_verifying_key(vk).hex() == small["vk"]
    out = {"fr_5": M.tb_fr(5).hex(), "fr_minus_1": M.tb_fr(M.R - 1).hex(), "g1_generator": M.tb_g1(G).hex(), "g1_zero": M.tb_g1(None).hex(),
           "commitment_without_shift": M.tb_commitment((G, None)).hex(), "commitment_with_shift": M.tb_commitment((G, G)).hex(),
           "index_vk_manual_constraints": M.tb_index_vk(vk).hex(),
           "fs_seed_manual_constraints": (b"MARLIN-2019" + M.tb_index_vk(vk) + b"".join(M.tb_fr(int(x, 16)) for x in small["public_input"])).hex()}
    with open(os.path.join(ROOT, "tests", "golden", "tobytes.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote tests/golden/tobytes.json")


def r1cs_files(d, test_circuit):
    sys.path.insert(0, ROOT)
    from simpleworks_amd import workloads as W
    os.makedirs(d, exist_ok=True)

    def write(name, cs):   # (the matrices as ark-relations' to_matrices returns them after finalize: sorted by column, merged)
        n = W.dump_r1cs(W.pack_model_system(cs), os.path.join(d, name + ".r1cs"))
        print(" wrote %s.r1cs: %d constraints, %d bytes" % (name, cs.num_constraints, n))
    large = json.load(open(os.path.join(ROOT, "tests", "golden", "marlin_large.json")))
    for lg in (12, 16):
        c = large["synthetic_2p%d" % lg]
        write("synthetic_2p%d" % lg, M.synthetic_circuit(1 << lg, int(c["a"], 16), int(c["b"], 16)))
    kw = large["merkle_h5"]["circuit"]
    g = W._SplitMix(kw["seed"])
    levels = kw["height"] - 1
    siblings = [g.fr() for _ in range(levels)]
    leaf_index = g.next_u64() % (1 << levels)
    cs = M.ConstraintSystem()
    public = W.build_merkle_membership(cs, W.MerkleParams(), kw["leaf_u8"], leaf_index, siblings, kw["gadget_byte_ops"])
    assert [hex(x) for x in public] == large["merkle_h5"]["public_input"]
    write("merkle_h5", cs)
    write("test_circuit", test_circuit)


if __name__ == "__main__":
    main()
