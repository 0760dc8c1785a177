"""Which librccl carries the library's exchanges is decided deterministically and reported (include/swmarlin.h
swm_rccl_info): SWM_RCCL_PATH or an error; else the copy the process has already mapped (torch's, when torch was imported first) —
never a second one beside it; else the loader's librccl.so.1.  CPU only: resolution needs no GPU.  One subprocess per case (the
resolution happens once per process)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
if sys.argv[2] == "torch":
    import torch
from simpleworks_amd._lib import rccl_info
ok, text = rccl_info()
copies = sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl.so" in l})
print("OK" if ok else "NO", "|", text, "|", ";".join(copies))
"""


def _probe(mode, **env_extra):
    env = dict(os.environ)
    env.pop("SWM_RCCL_PATH", None)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", _PROBE, ROOT, mode], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    ok, text, copies = [x.strip() for x in out.stdout.strip().splitlines()[-1].split("|")]
    return ok == "OK", text, [c for c in copies.split(";") if c]


def test_forced_path_that_does_not_load_is_an_error_not_a_fallback():
    ok, text, copies = _probe("plain", SWM_RCCL_PATH="/nonexistent/librccl.so")
    assert not ok and "SWM_RCCL_PATH=/nonexistent/librccl.so does not load" in text and copies == []


def test_loader_copy_without_torch_and_the_description_names_it():
    ok, text, copies = _probe("plain")
    if not ok:
        pytest.skip("no librccl on the loader's path here: %s" % text)
    assert text.startswith("librccl /") and " version " in text and "(loader search path)" in text
    assert len(copies) == 1 and copies[0] in text


def test_forced_path_is_used():
    ok0, text0, copies0 = _probe("plain")
    if not ok0:
        pytest.skip("no librccl to point at")
    ok, text, copies = _probe("plain", SWM_RCCL_PATH=copies0[0])
    assert ok and "(SWM_RCCL_PATH)" in text and copies == copies0 and copies0[0] in text


def test_torch_first_means_torchs_copy_and_no_second_one():
    torch = pytest.importorskip("torch")
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    ok, text, copies = _probe("torch")
    mapped_by_torch = [c for c in copies if os.path.dirname(torch.__file__) in c]
    if not mapped_by_torch:
        pytest.skip("this torch build does not map a librccl of its own at import (bundled copy %s)" % ("exists" if os.path.exists(bundled) else "absent"))
    assert ok and "(already mapped in the process)" in text, text
    assert len(copies) == 1, copies                       # never a second RCCL beside torch's
    assert copies[0] in text
