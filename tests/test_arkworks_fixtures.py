"""One run of the pin kit made permanent (VERDICT r05 "next" #6).  swmarlin-sys/tests/pin_golden.rs records what ARKWORKS ITSELF
produced — `MarlinInst = Marlin<Fr, MultiPC, FS>` of /root/reference/src/marlin/mod.rs:12-14 with ark_std::test_rng(), the codecs of
src/marlin/serialization.rs:5-45 — under tests/golden/arkworks/ in the schema of tests/golden/arkworks_schema.json.  Whoever first has
cargo commits that directory; from then on

  * this file compares the Python model's golden files against arkworks' own values (the oracle is then PINNED), and
  * the GPU golden-bytes tests compare the HIP path against them as well (oracle_lib.expected_bytes).

No toolchain here, so the directory is absent and the comparison is skipped BY NAME; what runs everywhere is the schema test: the
model's own output, projected onto the arkworks schema, written and loaded back through the same loader and comparer, agrees with
itself — and a changed byte, a missing key or an unknown case is reported by name."""
import copy
import json
import os

import pytest

from oracle_lib import ARKWORKS_DIR, arkworks_diff, arkworks_golden, arkworks_project, arkworks_schema, expected_bytes, golden

FILES = sorted(arkworks_schema())


def test_schema_names_files_and_keys_that_exist():
    for name in FILES:
        model = golden(name)
        proj = arkworks_project(name, model)
        assert proj, name
        spec = arkworks_schema()[name]
        if "per_case" in spec:
            for case, rec in proj.items():
                assert set(spec["per_case"]["required"]) <= set(rec), (name, case)
    # the pin kit writes exactly these files (one `record("<file>", ...)` family per schema entry)
    text = open(os.path.join(os.path.dirname(ARKWORKS_DIR), "..", "..", "swmarlin-sys", "tests", "pin_golden.rs")).read()
    for name in FILES:
        assert 'record("%s"' % name in text or ('let out_file = "%s"' % name) in text, "pin_golden.rs never records %s" % name


def test_round_trip_of_the_models_own_output_through_the_arkworks_schema(tmp_path):
    for name in FILES:
        model = golden(name)
        proj = arkworks_project(name, model)
        proj["_generator"] = "schema round trip (tests/test_arkworks_fixtures.py)"
        (tmp_path / name).write_text(json.dumps(proj))
        back = arkworks_golden(name, str(tmp_path))
        assert back is not None and "_generator" not in back
        assert arkworks_diff(name, model, back) == [], name
    # differences are reported by file, case and key
    m = golden("marlin.json")
    bad = copy.deepcopy(arkworks_project("marlin.json", m))
    bad["synthetic_8"]["proof"] = "00" + bad["synthetic_8"]["proof"][2:]
    del bad["synthetic_16"]["vk"]
    bad["not_a_case"] = {"vk": "", "proof": ""}
    d = arkworks_diff("marlin.json", m, bad)
    assert any("synthetic_8.proof differs" in x for x in d) and any("synthetic_16.vk missing" in x for x in d)
    assert any("not_a_case" in x for x in d) and len(d) == 3
    r = copy.deepcopy(arkworks_project("rng.json", golden("rng.json")))
    r["test_rng_u64"][3] = "0x0"
    r["fs_then_u128"] = "0x1"
    assert len(arkworks_diff("rng.json", golden("rng.json"), r)) == 2
    p = copy.deepcopy(arkworks_project("pedersen.json", golden("pedersen.json")))
    p["tree"]["root"] = "0x2"
    assert arkworks_diff("pedersen.json", golden("pedersen.json"), p) == ["pedersen.json: tree.root differs (arkworks 0x2..., model %s...)"
                                                                          % str(golden("pedersen.json")["tree"]["root"])[:24]]


def test_expected_bytes_adds_arkworks_when_present(tmp_path, monkeypatch):
    import oracle_lib
    assert [s for s, _ in expected_bytes("marlin.json", "synthetic_8", "proof")][0] == "model"
    proj = arkworks_project("marlin.json", golden("marlin.json"))
    (tmp_path / "marlin.json").write_text(json.dumps(proj))
    monkeypatch.setattr(oracle_lib, "ARKWORKS_DIR", str(tmp_path))
    got = oracle_lib.expected_bytes("marlin.json", "synthetic_8", "proof")
    assert [s for s, _ in got] == ["model", "arkworks"] and got[0][1] == got[1][1]


@pytest.mark.parametrize("name", FILES)
def test_model_against_arkworks_own_output(name):
    ark = arkworks_golden(name)
    if ark is None:
        pytest.skip("tests/golden/arkworks/%s is absent: nobody has run swmarlin-sys/tests/pin_golden.rs (cargo test --features pin) "
                    "yet — the oracle stays UNPINNED against arkworks" % name)
    assert arkworks_diff(name, golden(name), ark) == []
