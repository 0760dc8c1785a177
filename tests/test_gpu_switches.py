"""The switches of csrc/switches.h that select between maintained code paths (per-window schedule, XYZZ tables, table widths,
bucket-stage kernels, one-stream schedule, mask commitment in pieces, the older transform ...).  Each must still produce the
golden proof bytes and accepted proofs: one subprocess per setting (the switches are read once per process).  Until r05 this
list had 35 entries, most of them the launch geometry of rejected experiments; r06 removed those switches with their code."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, os, sys
sys.path.insert(0, %r)
from simpleworks_amd import marlin as M, workloads as W, serialization as S
case = json.load(open(os.path.join(%r, "tests", "golden", "marlin.json")))["synthetic_32"]
rng = M.generate_rand()
srs = M.generate_universal_srs(*case["srs"], rng)
cs = W.synthetic_circuit(case["num_constraints"], int(case["a"], 16), int(case["b"], 16))
pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
proof = M.generate_proof(cs, pk, rng)
assert S.serialize_proof(proof).hex() == case["proof"], "golden proof bytes"
for lg in (12, 17) + ((20,) if os.environ.get("SWM_TEST_ALSO_2P20") else ()):
    n = 1 << lg
    rng = M.generate_rand()
    srs = M.generate_universal_srs(n, n, n, rng)
    cs, public = W.synthetic_r1cs(n, 3 + lg, 5)
    pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
    a = S.serialize_proof(M.generate_proof(cs, pk, M.generate_rand()))
    b = S.serialize_proof(M.generate_proof(cs, pk, M.generate_rand()))
    assert a == b, "same rng, same proof"
    # the same stream behind the fill_bytes callback (a caller-owned generator): the mask commitment in pieces, the host
    # counting only the runs that could complete the draw
    c = S.serialize_proof(M.generate_proof(cs, pk, M.rng_behind_callback(M.generate_rand())))
    assert c == a, "callback rng, same proof"
    assert M.verify_proof(vk, public, S.deserialize_proof(a), M.generate_rand())
    print("sha", lg, __import__("hashlib").sha256(a).hexdigest())
ctx = M.default_context()
ctx.profile()
print("twins", ctx.last_work["msm_twins"])
"""


def _run(env_extra, twins=None):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    if twins is not None:
        twins.append(int([l for l in out.stdout.splitlines() if l.startswith("twins ")][0].split()[1]))
    return [l for l in out.stdout.splitlines() if l.startswith("sha ")]


# Every switch that selects between MAINTAINED paths of the product (csrc/switches.h), each with the values that leave the default
# path.  tests/test_switch_audit.py (CPU) holds csrc/ to this list: a switch read anywhere in the library must be named here or in
# its allow-list of diagnostics.
SETTINGS = (
    {"SWM_MSM_NO_TABLE": "1"},                                  # per-window schedule (what table-less base sets run)
    {"SWM_MSM_TE": "0"},                                        # XYZZ tables (what sets outside the prime-order subgroup get)
    {"SWM_MSM_TABLE_C": "15"},                                  # narrower window tables (what a rank of a sharded proof takes)
    {"SWM_MSM_TABLE_C": "18", "SWM_MSM_BATCH_BELOW": "0"},      # ... wider ones; every job with a bucket stage of its own
    {"SWM_MSM_TABLE_C": "18", "SWM_MSM_TWIN": "0"},             # ... and every job with a sort of its own (test_twin_jobs below)
    {"SWM_MSM_BATCH_BELOW": "4000000"},                         # every job in the round's joint bucket stage
    {"SWM_MSM_LOW": "1"}, {"SWM_MSM_LOW": "0"},                 # the low-LDS bucket stage everywhere / nowhere (default: joint stages)
    {"SWM_MSM_QUAD": "0"},                                      # small MSMs: one lane per chain / per segment
    {"SWM_PROVE_ONE_STREAM_LOG": "0"},                          # small proofs on the three-stage pipeline of the large ones
    {"SWM_PROVE_ONE_STREAM_LOG": "30"},                         # ... and every proof on one stream per lane
    {"SWM_MASK_PIECES": "1"}, {"SWM_MASK_PIECES": "2"},         # caller-owned generator: the mask commitment in one / two pieces
    {"SWM_NTT_LAZY": "0"},                                      # 8 x 32-bit transform instead of the 9 x 29-bit lazy one
    {"SWM_NTT_PASS_TABLES": "0"},                               # twiddles from the two-level tables (transforms beyond the table budget)
    {"SWM_RALPHA_TRANSFORMS": "1"},                             # r(alpha, X) on 4|H| by two transforms (the 2^-231 case of the closed form)
    {"SWM_SAMPLE_TIGHT": "1"},                                  # the bulk sampler's retry branch, several rounds per draw
    # values outside a switch's declared range are REFUSED (a line on stderr, the default is used): same bytes as no switch at all
    {"SWM_MSM_TABLE_C": "99", "SWM_MSM_LOW": "7", "SWM_MASK_PIECES": "x3", "SWM_NTT_LAZY": "-1"},
)


@pytest.mark.gpu
def test_switches_select_equivalent_paths():
    ref = _run({})
    assert len(ref) == 2
    for env in SETTINGS:
        assert _run(env) == ref, env


@pytest.mark.gpu
def test_twin_jobs_share_one_sort():
    """The plain and the shifted commitment of a bounded polynomial are MSMs of the same scalars: when the two tables have the
    same width the second job takes the first one's sort (msm.h: MsmTwin).  The path must have run (the context counts the jobs
    that took it), and not with SWM_MSM_TWIN=0 — same bytes: with one width forced for every table, and with the default widths
    up to 2^20 constraints."""
    on, off = [], []
    a = _run({"SWM_MSM_TABLE_C": "18"}, on)
    b = _run({"SWM_MSM_TABLE_C": "18", "SWM_MSM_TWIN": "0"}, off)
    assert a == b and len(a) == 2
    assert on[0] == 6, on   # two bounded polynomials (g_1, g_2) in each of the three 2^17 proofs (the 2^12 proofs' jobs are too small)
    assert off[0] == 0, off
    # default widths (g_1 and g_2 of each of the three proofs per size: the SRS has the circuit's own degree, so the shifted powers
    # are a sub-range of the powers and both jobs of a pair read the one table)
    on, off = [], []
    a = _run({"SWM_TEST_ALSO_2P20": "1"}, on)
    b = _run({"SWM_TEST_ALSO_2P20": "1", "SWM_MSM_TWIN": "0"}, off)
    assert a == b and len(a) == 3
    assert on[0] == 12 and off[0] == 0, (on, off)


SCRIPT_LARGE_SRS = r"""
import json, os, sys, hashlib
sys.path.insert(0, %r)
from simpleworks_amd import marlin as M, workloads as W, serialization as S
n = 1 << 16
rng = M.generate_rand()
srs = M.generate_universal_srs(2 * n, 2 * n, 2 * n, rng)     # a universal SRS larger than the circuit needs
cs, public = W.synthetic_r1cs(n, 19, 5)
pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
a = S.serialize_proof(M.generate_proof(cs, pk, M.generate_rand()))
b = S.serialize_proof(M.generate_proof(cs, pk, M.generate_rand()))
assert a == b
assert M.verify_proof(vk, public, S.deserialize_proof(a), M.generate_rand())
bad = list(public); bad[0] = (bad[0] + 1) %% M.R_MODULUS
assert not M.verify_proof(vk, bad, S.deserialize_proof(a), M.generate_rand())
# the key survives its codec with both tables rebuilt
pk2 = S.deserialize_proving_key(S.serialize_proving_key(pk))
assert S.serialize_proof(M.generate_proof(cs, pk2, M.generate_rand())) == a
print("sha", hashlib.sha256(a).hexdigest())
ctx = M.default_context()
ctx.profile()
print("twins", ctx.last_work["msm_twins"])
"""


@pytest.mark.gpu
def test_larger_universal_srs_uses_the_shifted_powers_table():
    """Under a universal SRS larger than the key's degree the shifted powers are NOT a sub-range of the powers: the key keeps a
    scaled copy and a window table of their own (install_committer_key), the shifted commitments of g_1 / g_2 read THAT table, and
    a twin pair spans two tables (different strides) when their widths agree.  Every other test generates the SRS for the
    circuit's own degree, where all of that is bypassed.  Proofs must verify, a wrong input must not, and the bytes must not depend
    on the table widths or on whether the shifted job sorted for itself."""
    def run(env_extra):
        env = dict(os.environ)
        env.update(env_extra)
        env["SWM_TRACE"] = "1"
        out = subprocess.run([sys.executable, "-c", SCRIPT_LARGE_SRS % ROOT], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
        sha = [l for l in out.stdout.splitlines() if l.startswith("sha ")]
        twins = int([l for l in out.stdout.splitlines() if l.startswith("twins ")][0].split()[1])
        widths = sorted(set(l.split("base set of ")[1] for l in out.stderr.splitlines() if "base set of" in l))
        return sha, twins, widths
    sha0, tw0, w0 = run({})
    assert any(w.startswith("65535 points") for w in w0), w0           # the shifted set got a table of its own ...
    assert tw0 == 0, tw0                                               # ... narrower than the powers': no twin jobs
    sha1, tw1, w1 = run({"SWM_MSM_TABLE_C": "18"})
    assert sha1 == sha0 and tw1 == 6, (tw1, w1)                        # one width: g_1, g_2 of three proofs pair up across two tables
    sha2, tw2, _ = run({"SWM_MSM_TABLE_C": "18", "SWM_MSM_TWIN": "0"})
    assert sha2 == sha0 and tw2 == 0
