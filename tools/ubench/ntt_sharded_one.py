"""What ONE rank of a transform split over G ranks computes (local passes, twiddle, the all-to-all as device copies of the same
size, the length-G cross transform), beside the whole transform on one GPU.  SWM_SHARD_EMULATE makes the library's device
exchange hand back the rank's own chunk: wrong values, the right work.  Prints one JSON line per (log_n, G).
usage: SWM_SHARD_EMULATE=1 python3 tools/ubench/ntt_sharded_one.py [log_n ...]"""
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
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
import simpleworks_amd as swm
assert os.environ.get("SWM_SHARD_EMULATE"), "set SWM_SHARD_EMULATE=1"
ctx = swm.Context(0)
for lg in [int(a) for a in sys.argv[1:]] or [22, 24]:
    x = np.random.default_rng(5).integers(0, 1 << 60, size=(1 << lg, 4), dtype=np.uint64)
    d = ctx.to_device(x)

    def timed(fn, reps=10):
        for _ in range(3):
            fn()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    ctx.set_msm_sharding(0, 1, None)
    full = timed(lambda: ctx.ntt_fr_dev(d, lg, 0, 0))
    for G in (2, 4, 8):
        ctx.set_msm_sharding(G - 1, G, lambda send, G=G: bytes(send) * G)
        fwd = timed(lambda: ctx.ntt_fr_sharded_dev(d, lg, False, False))
        inv = timed(lambda: ctx.ntt_fr_sharded_dev(d, lg, True, True))
        print(json.dumps({"log_n": lg, "ranks": G, "whole_transform_ms": full, "one_rank_forward_ms": fwd, "one_rank_inverse_ms": inv}), flush=True)
    ctx.set_msm_sharding(0, 1, None)
