"""Per-rank time of ONE 2^log_n proof split over G ranks, measured on ONE GPU: the context is told it is rank R of G and its
exchange callback hands back this rank's own contribution in every slot (so the proof is not a valid one — only the time is
of interest: everything a rank computes, with the exchange reduced to a host memcpy).  What the number says: the time one
rank needs when the other G - 1 GPUs work beside it, i.e. the strong-scaling bound of the split (xGMI exchanges of 192 B x
commitments per round are microseconds).  Prints one JSON line per (G, rank).
usage: shard_emulate.py [log_n=20] [steps=5] [G ...]        env: SWM_SHARD_RANGE=1 / SWM_SHARD_BUCKETS=1 -> point-range / bucket-range split instead of the cyclic one, SWM_SHARD_R1_OFF"""
# SWM_SHARD_EMULATE is a MEASUREMENT HOOK that makes proofs wrong by construction: it is compiled only into a second library
# (-DSWM_MEASURE_HOOKS: `bash tools/buildvar.sh hooks -DSWM_MEASURE_HOOKS` -> build/libswmarlin_hooks.so), never into the shipped one
def _use_hooks_library():
    import os, subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
    lib = os.path.join(root, "build", "libswmarlin_hooks.so")
    if os.environ.get("SWM_SHARD_EMULATE") and not os.environ.get("SWM_LIB_PATH"):
        if not os.path.exists(lib):
            subprocess.check_call(["bash", os.path.join(root, "tools", "buildvar.sh"), "hooks", "-DSWM_MEASURE_HOOKS"])
        os.environ["SWM_LIB_PATH"] = os.path.abspath(lib)
_use_hooks_library()

import json, os, sys, time
# the sharded rounds 1 and 2 exchange DATA the rest of the proof depends on (an emulated exchange makes the prover's own checks
# fail): they are measured apart (tools/ubench/ntt_sharded_one.py); here every transform runs whole on the rank
# (SWM_SHARD_EMULATE=1: try them sharded as well, with the library's device exchanges handing back this rank's own chunk)
ROUNDS_SHARDED = bool(os.environ.get("SWM_SHARD_EMULATE"))
if not ROUNDS_SHARDED:
    os.environ.setdefault("SWM_SHARD_R1_OFF", "1")
    os.environ.setdefault("SWM_SHARD_R2_OFF", "1")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import simpleworks_amd as swm
from simpleworks_amd import marlin as M, workloads as W

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << lg
ctx = M.default_context()
cs, public = W.synthetic_r1cs(n, 0x1234567, 0x7654321)
mode = ("buckets" if os.environ.get("SWM_SHARD_BUCKETS", "0") not in ("", "0") else
        "ranges" if os.environ.get("SWM_SHARD_RANGE", "0") not in ("", "0") else "cyclic")
for G in ([int(a) for a in sys.argv[3:]] or [1, 2, 4, 8]):
    # the key is built by a context that already knows its world: the table width follows the rank's share (msm_install_bases)
    ctx.set_msm_sharding(0, G, None if G == 1 else (lambda send, G=G: bytes(send) * G))
    rng = M.generate_rand()
    srs = M.generate_universal_srs(n, n, n, rng)
    pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
    srs.free()
    for rank in sorted({0, G - 1}):
        if G > 1:
            ctx.set_msm_sharding(rank, G, lambda send, G=G: bytes(send) * G)
        for _ in range(2):
            M.generate_proof(cs, pk, rng)
        ctx.synchronize()
        ctx.profile_reset()
        ctx.profile_enable(2)
        t0 = time.perf_counter()
        for _ in range(steps):
            M.generate_proof(cs, pk, rng)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ctx.profile_enable(False)
        prof = ctx.profile()
        print(json.dumps({"log_n": lg, "split": mode, "rounds": "sharded" if ROUNDS_SHARDED else "whole", "ranks": G, "rank": rank, "ms_per_proof_on_this_rank": dt * 1e3,
                          "msm_adds": ctx.last_work.get("msm_adds"), "accumulate_ms": prof.get("msm_accumulate", {}).get("total_ms", 0) / steps}),
              flush=True)
    pk.free()
ctx.set_msm_sharding(0, 1, None)
