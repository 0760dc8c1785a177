"""The experiment switches DESIGN.md quotes A/B numbers for select alternative code paths (per-window schedule, stream-
ordered result copies, plain stream order, 64-lane bucket stage, single stage stream ...).  Each must still produce the
golden proof bytes and accepted proofs: one subprocess per setting (the switches are read once per process)."""
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
for lg in (12, 17):
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
"""


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return [l for l in out.stdout.splitlines() if l.startswith("sha ")]


@pytest.mark.gpu
def test_switches_select_equivalent_paths():
    ref = _run({})
    assert len(ref) == 2
    for env in ({"SWM_MSM_NO_TABLE": "1"}, {"SWM_MSM_ZERO_COPY": "0"}, {"SWM_MSM_QUEUE_ORDER": "0"},
                {"SWM_RED_LANES": "64"}, {"SWM_MSM_PIPE": "0"}, {"SWM_MSM_LAT_BELOW": "0", "SWM_MSM_BATCH_BELOW": "0"},
                {"SWM_MSM_SMALL_LANES": "1", "SWM_COMMIT_LATE": "1"},
                {"SWM_MSM_TE": "0"},          # XYZZ tables instead of the twisted Edwards ones
                {"SWM_SAMPLE_TIGHT": "1"},    # the bulk sampler's retry branch, several rounds per draw
                {"SWM_NTT_LAZY": "0"},        # 8 x 32-bit Comba transform instead of the 9 x 29-bit lazy one
                {"SWM_NTT_PASS_TABLES": "0"},  # lazy transform with the two-level twiddle product on every pass
                {"SWM_RALPHA_TRANSFORMS": "1"},  # r(alpha, X) on 4|H| by two transforms instead of the closed form
                {"SWM_MSM_QUAD": "0", "SWM_BINV_SMALL": "0"},  # small MSMs: one lane per chain of the bucket stage; 16-element inversion chunks
                {"SWM_MSM_QUAD_RB": "64", "SWM_MSM_QUAD_MAXB": "1048576"},   # quad bucket stage in its other shapes, also at 2^17
                {"SWM_MSM_QUAD_RB": "256", "SWM_MSM_QUAD_BLOCKS": "16"},
                {"SWM_MSM_QUAD_ACC": "0", "SWM_FLAT_PART_TILE": "8192"},  # one lane per segment in small accumulations; 8 K-digit partition tiles
                {"SWM_MSM_TABLE_C": "15"},      # narrower window tables (what a rank of a sharded proof takes)
                # r05: the low-LDS bucket stage everywhere / nowhere (default: joint launches only; alone, re-shaped up to 512 workgroups, a barrier
                # behind every step), two bucket-stage streams, smaller joint stages, the mask commitment enqueued last
                {"SWM_MSM_LOW": "1"}, {"SWM_MSM_LOW": "0"},
                {"SWM_MSM_LOW": "1", "SWM_MSM_LOW_BLOCKS": "512", "SWM_MSM_LAT_BELOW": "0", "SWM_MSM_BATCH_BELOW": "0", "SWM_LOW_SYNC_ALL": "1"},
                {"SWM_MSM_TAILS": "2", "SWM_MSM_LAT_BELOW": "0", "SWM_MSM_BATCH_BELOW": "0"},
                {"SWM_MSM_JOINT_BLOCKS": "16", "SWM_MASK_COMMIT": "2"},
                {"SWM_SORT_NARROW": "2", "SWM_MSM_LAT_BELOW": "0"},   # 256-lane partition and bin sort (co-resident with an accumulation)
                {"SWM_MSM_PREFIX_TABLES": "1"},                        # narrower tables over prefixes of the powers (|H| + 1 points)
                {"SWM_MSM_JOINT_ADAPT": "0", "SWM_R1_ORDER": "1"},     # 64 workgroups per job in a joint stage; z_B committed ahead of z_A
                # the caller-owned generator's draw: one piece / two pieces of the mask commitment, every run counted on the host
                {"SWM_MASK_PIECES": "1", "SWM_EXT_COUNT_ALL": "1"}, {"SWM_MASK_PIECES": "2"},
                {"SWM_EXT_READBACK": "1", "SWM_EXT_RING": "2"},   # stream-synchronising read-backs of the device's total; two host chunks
                {"SWM_PROVE_ONE_STREAM_LOG": "0"},                 # commitments of small proofs pipelined over three streams from 131 072 points (r02 - r04)
                {"SWM_REC_LAZY": "0", "SWM_BINV_LAZY": "0"},      # recurrences and batch inversion on the 8 x 32-bit Comba multiplier
                # the first commitment of a round (w, t, h_1, g_1, g_2, h_2 with mask 0xfa: at most 8 MSMs in flight) as two MSMs, also for the small circuits
                {"SWM_HEAD_SPLIT": "2"}, {"SWM_HEAD_SPLIT": "3", "SWM_HEAD_MASK": "0xfa", "SWM_HEAD_MIN": "64"}):
        assert _run(env) == ref, env
