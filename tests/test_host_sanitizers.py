"""The host-side, attacker-facing surface of the library under AddressSanitizer + UndefinedBehaviorSanitizer.

The reference's deserialisers return Err on bad bytes and the crate forbids panics (/root/reference/src/marlin/serialization.rs:14-17,
26-31,40-45; /root/reference/src/lib.rs:28).  Here the entry points that parse untrusted bytes — swm_proof_validate,
swm_verify_proof, swm_vk_deserialize / swm_vk_serialize, the host prefix of swm_pk_deserialize, plus swm_rng_* and swm_blake2s —
are compiled FROM THE TEXT THE LIBRARY SHIPS (csrc/host/host_abi.inc, csrc/host/*.h) by g++ -fsanitize=address,undefined into
tests/native/host_fuzz.cpp and fed >= 10 000 seeded mutations of the golden proof / verifying key / proving key of the reference's
manual-constraints example (truncations, bit flips, inflated length fields, flag bits, non-canonical field elements, inserted /
removed / trailing bytes).  Every rejection must be a status code; no input may trip a sanitizer.  CPU only: no GPU, no HIP call."""
import json
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "host_fuzz.cpp")
R_MODULUS = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
PROCS = 8
PER_PROC = (500, 400, 375)   # proof / vk / pk mutations per process: 8 x 1275 = 10 200


def _golden():
    with open(os.path.join(ROOT, "tests", "golden", "marlin.json")) as f:
        return json.load(f)["manual_constraints"]


def _build(tmp_path):
    exe = str(tmp_path / "host_fuzz")
    cmd = ["g++", "-std=c++17", "-O2", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "simpleworks_amd", "csrc"),
           "-I", os.path.join(ROOT, "include"), SRC, "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if out.returncode != 0 and ("asan" in out.stderr.lower() or "ubsan" in out.stderr.lower()) and "error:" not in out.stderr:
        pytest.skip("this g++ has no ASan / UBSan runtime")
    assert out.returncode == 0, out.stderr[-4000:]
    return exe


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include/hip"), reason="needs g++ and the HIP headers")
def test_codecs_and_verifier_survive_10k_mutations_under_asan_ubsan(tmp_path):
    case = _golden()
    with open(os.path.join(ROOT, "tests", "golden", "pk_bytes.json")) as f:
        pk_hex = json.load(f)["manual_constraints"]["bytes"]
    files = {"vk.bin": bytes.fromhex(case["vk"]), "proof.bin": bytes.fromhex(case["proof"]), "pk.bin": bytes.fromhex(pk_hex),
             # public inputs as the ABI takes them: Montgomery limbs
             "pi.bin": b"".join(((int(x, 16) << 256) % R_MODULUS).to_bytes(32, "little") for x in case["public_input"])}
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    exe = _build(tmp_path)
    scale = float(os.environ.get("SWM_FUZZ_SCALE", "1"))
    counts = [str(max(1, int(c * scale))) for c in PER_PROC]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")

    def run(seed):
        return subprocess.run([exe] + [str(tmp_path / n) for n in ("vk.bin", "proof.bin", "pi.bin", "pk.bin")] + [str(seed)] + counts,
                              capture_output=True, text=True, timeout=1800, env=env)
    with ThreadPoolExecutor(PROCS) as ex:
        outs = list(ex.map(run, range(1, PROCS + 1)))
    total = parsed = 0
    for seed, out in enumerate(outs, 1):
        assert out.returncode == 0 and out.stdout.startswith("OK "), "seed %d: %s%s" % (seed, out.stdout[-500:], out.stderr[-4000:])
        assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr and "LeakSanitizer" not in out.stderr, out.stderr[-4000:]
        nums = [int(t) for t in out.stdout.replace(";", " ").replace(":", " ").split() if t.isdigit()]
        p_ok, p_rej, _p_ver, v_ok, v_rej, _v_acc, k_ok, k_rej = nums[:8]
        total += p_ok + p_rej + v_ok + v_rej + k_ok + k_rej
        parsed += p_ok + v_ok + k_ok
    if scale >= 1:
        assert total >= 10000, total
    assert 0 < parsed < total   # both outcomes were exercised: inputs the parsers take and inputs they refuse
