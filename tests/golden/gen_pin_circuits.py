#!/usr/bin/env python3
"""Writes tests/golden/pin_circuits.json: the small circuits behind tests/golden/marlin.json and pk_bytes.json as plain data —
instance and witness assignments and every constraint's three linear combinations — so that a program WITHOUT the Python
builders (swmarlin-sys/tests/pin_golden.rs: arkworks' own CPU prover, no GPU, no libswmarlin) can replay them into an
ark-relations ConstraintSystem and compare what arkworks emits with the committed golden bytes.
Variables: ["i", k] = instance variable k (k = 0 is the constant one), ["w", k] = witness variable k; numbers are hex strings.
Run from the repo root: python3 tests/golden/gen_pin_circuits.py"""
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
    with open(os.path.join(ROOT, "tests", "golden", "pin_circuits.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote tests/golden/pin_circuits.json:", {k: len(v["constraints"]) for k, v in out["marlin"].items()})


if __name__ == "__main__":
    main()
