"""csrc/ reads its environment in ONE place (csrc/switches.h) and every switch it reads is either exercised for golden proof bytes by
tests/test_gpu_switches.py or is a declared diagnostic that cannot change a result (VERDICT r05 "next" #5).  CPU only: source audit."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "simpleworks_amd", "csrc")

# switches that cannot change what a call returns: traces, kernel-trace delimiters, host-thread counts, which librccl is loaded
DIAGNOSTICS = {"SWM_TRACE", "SWM_PROOF_MARKS", "SWM_POOL_WORKERS", "SWM_POOL_SPIN_US", "SWM_RCCL_PATH", "LOCAL_WORLD_SIZE"}
# exercised by the sharded-prover tests (tests/test_gpu_marlin.py, tests/test_gpu_multi.py, tests/test_dist.py), not by the switch test
SHARDING = {"SWM_SHARD_RANGE", "SWM_SHARD_BUCKETS", "SWM_SHARD_BLOCK_LOG", "SWM_SHARD_R1_OFF", "SWM_SHARD_R2_OFF", "SWM_SHARD_FORCE"}
# compiled only with -DSWM_MEASURE_HOOKS (tools/ubench/shard_emulate.py builds that second library); never in the shipped one
MEASURE_HOOKS = {"SWM_SHARD_EMULATE"}


def _sources():
    for dirpath, _, files in os.walk(CSRC):
        if os.path.basename(dirpath) == "build":
            continue
        for f in files:
            if f.endswith((".hip", ".cuh", ".h", ".inc", ".cpp")):
                yield os.path.join(dirpath, f)


def _read_switches():
    names = {}
    for path in _sources():
        text = open(path, errors="replace").read()
        for m in re.finditer(r'env_(?:switch|flag|path)\(\s*"([A-Z0-9_]+)"', text):
            names.setdefault(m.group(1), set()).add(os.path.relpath(path, CSRC))
    return names


def test_getenv_only_in_switches_h():
    offenders = []
    for path in _sources():
        if os.path.basename(path) == "switches.h":
            continue
        text = re.sub(r"//[^\n]*", "", open(path, errors="replace").read())
        if re.search(r"\bgetenv\s*\(", text):
            offenders.append(os.path.relpath(path, CSRC))
    assert not offenders, "getenv outside csrc/switches.h: %s" % offenders
    n = len(re.findall(r"\bgetenv\s*\(", open(os.path.join(CSRC, "switches.h")).read()))
    assert 1 <= n <= 4, n   # (r05: 95 getenv calls over seven files)


def test_every_switch_is_tested_or_a_declared_diagnostic():
    import test_gpu_switches as T
    tested = set()
    for env in T.SETTINGS:
        tested |= set(env)
    read = _read_switches()
    assert len(read) >= 15, read
    for name, where in sorted(read.items()):
        assert name in tested or name in DIAGNOSTICS or name in SHARDING or name in MEASURE_HOOKS, \
            "%s is read in %s but neither exercised by tests/test_gpu_switches.py nor a declared diagnostic" % (name, sorted(where))
    for name in tested:
        assert name in read, "tests/test_gpu_switches.py sets %s, which the library no longer reads" % name
    # the sharding switches are referenced by the sharded-prover tests
    body = "".join(open(os.path.join(ROOT, "tests", f)).read() for f in ("test_gpu_marlin.py", "test_gpu_multi.py", "test_dist.py"))
    for name in SHARDING:
        assert name in body, "%s: no sharded-prover test sets it" % name


def test_measurement_hook_is_not_in_the_shipped_library():
    """SWM_SHARD_EMULATE answers exchanges with the rank's own data (wrong proofs by construction): its code sits between
    #ifdef SWM_MEASURE_HOOKS / #endif, the Makefile does not define that macro, and the built library does not contain the name."""
    for path in _sources():
        text = open(path, errors="replace").read()
        for m in re.finditer(r'env_flag\("SWM_SHARD_EMULATE"\)', text):
            before = text[: m.start()]
            assert before.count("#ifdef SWM_MEASURE_HOOKS") > before.count("#endif  // SWM_MEASURE_HOOKS") or \
                before.rfind("#ifdef SWM_MEASURE_HOOKS") > before.rfind("#endif"), path
    assert "SWM_MEASURE_HOOKS" not in open(os.path.join(CSRC, "Makefile")).read()
    lib = os.path.join(ROOT, "simpleworks_amd", "libswmarlin.so")
    if os.path.exists(lib):
        assert b"SWM_SHARD_EMULATE" not in open(lib, "rb").read()


def test_refused_values_fall_back_to_the_default(tmp_path):
    """env_switch: a value that does not parse or lies outside the declared range is refused, never clamped (compiled on the host
    from the shipped header)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        import pytest
        pytest.skip("needs g++")
    src = tmp_path / "t.cpp"
    src.write_text('#include "switches.h"\n#include <stdio.h>\nint main() { printf("%ld %ld %d %d\\n", swm::env_switch("SWM_T_A", 7, 0, 9), '
                   'swm::env_switch("SWM_T_B", 3, 1, 3), (int)swm::env_flag("SWM_T_C"), (int)swm::env_flag("SWM_T_D")); return 0; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++17", "-I", CSRC, str(src), "-o", str(exe)])

    def run(**env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("SWM_T_")}
        e.update(env)
        out = subprocess.run([str(exe)], env=e, capture_output=True, text=True)
        return out.stdout.split(), out.stderr
    assert run()[0] == ["7", "3", "0", "0"]
    assert run(SWM_T_A="5", SWM_T_B="2", SWM_T_C="1", SWM_T_D="0")[0] == ["5", "2", "1", "0"]
    vals, err = run(SWM_T_A="10", SWM_T_B="2x")
    assert vals[:2] == ["7", "3"] and err.count("refused") == 2
    assert run(SWM_T_A="-1", SWM_T_B="")[0][:2] == ["7", "3"]
    assert run(SWM_T_A="0x9")[0][0] == "9"
