"""CPU test: libswmarlin.so loads without a GPU and exports every symbol include/swmarlin.h declares; the product
fails loudly (no CPU fallback) when there is no device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "swmarlin.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(swm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import simpleworks_amd._lib as L
    lib = L.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), name
        assert name in L.ABI, "python binding lacks %s" % name
    for name in L.ABI:
        assert name in declared, "%s bound but not declared in swmarlin.h" % name
    assert lib.swm_version() >= 100
    assert lib.swm_strerror(-2).decode().startswith("no usable gfx950")


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import simpleworks_amd as swm
    with pytest.raises(swm.SwmError) as e:
        swm.Context(0)
    assert e.value.code == -2


def test_product_does_not_reference_oracle():
    """The product package must never import / link / call the oracle (tier rule 3)."""
    pkg = os.path.join(ROOT, "simpleworks_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in src and "oracle_lib" not in src and "pyref" not in src, os.path.join(dirpath, f)
