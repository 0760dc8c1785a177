"""The RCCL path with ONE RANK PER GPU: every multi-rank test elsewhere pins Context(0) and exchanges through a
thread barrier, gloo, or a one-rank communicator.  These tests run whenever two or more GPUs are visible — so that the driver's
pytest on any multi-GPU box executes csrc/capi.hip's ncclAllGather / grouped ncclSend + ncclRecv for real before a SCALE run
does — and are skipped (by name, with the reason) on the single-GPU boxes this library is developed on:

  * swm_ntt_fr_sharded_dev over the library's own communicator against the oracle (all four direction / layout combinations);
  * ONE proof of the 2^12 synthetic circuit over all ranks: golden proof bytes on every rank, the exchanges counted;
  * bench.py --gpus N through the driver's launcher (python -m torch.distributed.run), the `sharded` leg on.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpus():
    try:
        import torch
        return torch.cuda.device_count()   # (counting devices does not initialise the GPU)
    except Exception:  # noqa: BLE001
        return 0


NGPUS = _ngpus()
needs_two_gpus = pytest.mark.skipif(NGPUS < 2, reason="needs >= 2 GPUs: RCCL with one rank per GPU (%d visible)" % NGPUS)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_WORKER = r"""
import json, os, sys, time
root, rank, world, idfile, torch_first = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
sys.path[:0] = [root, os.path.join(root, "oracle"), os.path.join(root, "tests")]
if torch_first:                             # torch maps ITS librccl: the library must then use that very copy, not a second one
    import torch
import numpy as np
import simpleworks_amd as swm
from simpleworks_amd import marlin as M, serialization as S, workloads as W
from simpleworks_amd._lib import rccl_unique_id, rccl_info
from simpleworks_amd.dist import blocks_rows, cyclic_rows
from oracle_lib import Oracle, golden, h2i
from pyref.prng import fr_array

ctx = swm.Context(rank)                     # one rank per GPU
if rank == 0:                               # the unique id travels through a file: no torch.distributed in this test
    uid = rccl_unique_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 120:
            raise SystemExit("rank %d: no unique id after 120 s" % rank)
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
ctx.rccl_init(uid, rank, world)             # ncclCommInitRank; sharding is on
out = {"rank": rank, "ntt": True, "rccl": rccl_info()[1],
       "rccl_copies_mapped": len({l.split()[-1] for l in open("/proc/self/maps") if "librccl.so" in l})}
orc = Oracle()
log_n = 14
n = 1 << log_n
x = orc.fr_to_mont(fr_array(n, 914))
for inverse in (False, True):
    ref = orc.ntt(x, log_n, int(inverse), 0, 4)
    for blocks_in in (False, True):
        rows_in = (blocks_rows if blocks_in else cyclic_rows)(log_n, world, rank)
        d = ctx.to_device(np.ascontiguousarray(x[rows_in]))
        ctx.ntt_fr_sharded_dev(d, log_n, inverse, blocks_in)    # grouped ncclSend / ncclRecv between the GPUs
        got = d.download((n // world, 4))
        d.free()
        rows_out = (cyclic_rows if blocks_in else blocks_rows)(log_n, world, rank)
        out["ntt"] = out["ntt"] and bool(np.array_equal(got, ref[rows_out]))
M.set_default_context(ctx)
case = golden("marlin_large.json")["synthetic_2p12"]
rng = M.generate_rand()
srs = M.generate_universal_srs(*case["srs"], rng)
cs, public = W.synthetic_r1cs(case["num_constraints"], h2i(case["a"]), h2i(case["b"]))
pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
c0, b0 = ctx.exchange_stats()
proof = M.generate_proof(cs, pk, rng)
c1, b1 = ctx.exchange_stats()
out["vk"] = S.serialize_verifying_key(vk).hex() == case["vk"]
out["proof"] = S.serialize_proof(proof).hex() == case["proof"]
out["verifies"] = bool(M.verify_proof(vk, public, proof, M.generate_rand()))
out["exchanges"], out["bytes"] = c1 - c0, b1 - b0
pk.free()
srs.free()
print(json.dumps(out), flush=True)
"""


@needs_two_gpus
@pytest.mark.gpu
@pytest.mark.parametrize("torch_first", [0, 1])   # both resolution orders of librccl (swm_rccl_info): the loader's copy / torch's copy
@pytest.mark.parametrize("world", sorted({2, min(NGPUS, 8)} if NGPUS >= 2 else {2}))
def test_rccl_one_rank_per_gpu_sharded_ntt_and_golden_proof(tmp_path, world, torch_first):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    idfile = str(tmp_path / "rccl_id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), idfile, str(torch_first)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e.decode()[-3000:]
        outs.append(json.loads([l for l in o.decode().splitlines() if l.startswith("{")][-1]))
    assert sorted(o["rank"] for o in outs) == list(range(world))
    for o in outs:
        assert o["ntt"], "rank %d: the sharded transform differs from the oracle" % o["rank"]
        assert o["vk"] and o["proof"] and o["verifies"], o
        # per-round partial sums (4) + round 1 (4) + rounds 2 and 3 (all-to-alls and all-gathers): as the thread-rank test counts them
        assert o["exchanges"] >= 4 + 4 + 6 + 3, o
    assert len({o["exchanges"] for o in outs}) == 1
    # every rank names the library that carried its exchanges — the same one on all of them, and only ONE copy mapped per process
    assert len({o["rccl"] for o in outs}) == 1 and outs[0]["rccl"].startswith("librccl "), [o["rccl"] for o in outs]
    assert all(o["rccl_copies_mapped"] == 1 for o in outs), outs
    if torch_first:
        assert "already mapped" in outs[0]["rccl"], outs[0]["rccl"]


@needs_two_gpus
@pytest.mark.gpu
def test_bench_through_the_launcher_one_rank_per_gpu_with_the_sharded_leg():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2`: the driver's SCALE command line, RCCL,
    the sharded leg ON (at 2^14 so that the test stays short).  The line must carry a sharded object without an error."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SWM_BENCH_BACKEND", "SWM_BENCH_DEVICE", "SWM_BENCH_NO_SHARDED"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--log-n", "14", "--no-cpu-baseline", "--no-drop-in"], env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    sh = line["sharded"]
    assert sh and "error" not in sh, sh
    assert sh["ranks"] == 2 and sh["proof_bytes_identical_on_all_ranks"] is True and sh["proof_verifies"] is True
    assert sh["exchanges_per_proof"] >= 4
    assert sh["rccl"].startswith("librccl ") and sh["rccl_same_on_all_ranks"] is True


@pytest.mark.gpu
def test_the_multi_gpu_tests_name_what_they_skip():
    """On a single-GPU box the two tests above are skipped; this one records that fact in the run (and passes), so that a green
    suite cannot be mistaken for a multi-GPU measurement."""
    if NGPUS < 2:
        print("single-GPU box (%d device): test_rccl_one_rank_per_gpu_* and test_bench_through_the_launcher_* were skipped" % NGPUS)
    assert NGPUS >= 1
