"""bench.py's N > 1 code path (process group, the library's RCCL communicator, ONE proof over all ranks, the `sharded` object of
the JSON line) executed on a single-GPU box: a FRESH child process with SWM_BENCH_FORCE_DIST=1 and a world of one — so that the
driver's pytest has run those lines before a multi-GPU SCALE run ever does."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_bench_multi_rank_path_with_a_world_of_one():
    env = dict(os.environ)
    env.update({"SWM_BENCH_FORCE_DIST": "1", "SWM_SHARD_FORCE": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--log-n", "16",
                          "--no-cpu-baseline", "--no-drop-in"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    sh = line["sharded"]
    assert sh and "error" not in sh, sh
    assert sh["proof_bytes_identical_on_all_ranks"] is True
    assert sh["ranks"] == 1 and sh["exchanges_per_proof"] >= 4   # the per-round all-gathers went through the library's communicator
    assert sh["ms_per_proof"] > 0
    assert sh["rccl"].startswith("librccl ") and " version " in sh["rccl"], sh["rccl"]   # which library carried the exchange, and how it was found


@pytest.mark.gpu
def test_bench_two_ranks_through_the_drivers_launcher():
    """The driver's N > 1 command line — python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 2 … — on a single-GPU box: both ranks on device 0, the collectives over gloo (the hooks
    SWM_BENCH_BACKEND / SWM_BENCH_DEVICE), the sharded leg off (two ranks of one RCCL communicator cannot share a device).  Rank 0
    must print ONE JSON line whose value is the whole job: both ranks' proofs over the max-over-ranks time."""
    pytest.importorskip("torch")
    env = dict(os.environ)
    env.update({"SWM_BENCH_BACKEND": "gloo", "SWM_BENCH_DEVICE": "0", "SWM_BENCH_NO_SHARDED": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--log-n", "14", "--no-cpu-baseline", "--no-drop-in"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines          # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert abs(line["value"] - 2 * (1 << 14) / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]   # whole-job units / max-over-ranks time
    assert line["sharded"] is None


@pytest.mark.gpu
def test_bench_times_an_r1cs_dump(tmp_path):
    """`bench.py --r1cs FILE`: the constraint system of an SWMR1CS1 dump (how the reference's own circuits — MerkleTreeVerificationU8 as
    ark-r1cs-std lays it out — are timed on a box without Rust) through the whole bench path: load, is_satisfied, SRS, index, proofs,
    verification of the last proof; the line names the file."""
    sys.path.insert(0, ROOT)
    from simpleworks_amd import workloads as W
    cs, _, _ = W.merkle_membership_circuit(height=4, gadget_byte_ops=16)
    path = str(tmp_path / "merkle_h4.r1cs")
    W.dump_r1cs(cs, path)
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--r1cs", path, "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--no-drop-in"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "merkle_h4.r1cs" in line["config"]["workload"] and line["config"]["per_gpu_units"] == cs.pack().num_constraints
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_bench_overlapped_contexts():
    """`bench.py --overlap 2`: two contexts on the one GPU, a key and a host thread each, proving concurrently after the timed loop; the
    line carries the `overlapped` object and `value` is still the one-proof-at-a-time figure."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", "14", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--no-drop-in", "--overlap", "2"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    ov = line["overlapped"]
    assert ov["contexts"] == 2 and ov["proofs"] == 6 and ov["ms_per_proof"] > 0
    assert abs(line["value"] - (1 << 14) / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
