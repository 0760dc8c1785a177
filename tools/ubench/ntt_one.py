"""One transform size alone on the chip, for the PMC passes of tools/collect_profiles.sh: prints a JSON line with the
wall time per transform.  usage: ntt_one.py [log_n=22] [reps=10]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import simpleworks_amd as swm
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = swm.Context(0)
x = np.random.default_rng(5).integers(0, 1 << 60, size=(1 << lg, 4), dtype=np.uint64)
d = ctx.to_device(x)
for _ in range(3):
    ctx.ntt_fr_dev(d, lg, 0, 0)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.ntt_fr_dev(d, lg, 0, 0)
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"log_n": lg, "transforms": reps + 3, "ms_per_transform": dt * 1e3, "algorithmic_GBps": (64 << lg) / dt / 1e9}))
