"""One mat-vec shape alone on the chip (the synthetic R1CS: one non-zero per row), for the PMC passes of
tools/collect_profiles.sh.  usage: spmv_one.py [log_rows=20] [reps=10]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import simpleworks_amd as swm
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = swm.Context(0)
rng = np.random.default_rng(5)
rows = 1 << lg
rowptr = np.arange(rows + 1, dtype=np.uint32)
col = rng.integers(0, rows, rows, dtype=np.uint32)
val = rng.integers(0, 1 << 60, size=(rows, 4), dtype=np.uint64)
z = rng.integers(0, 1 << 60, size=(rows, 4), dtype=np.uint64)
d = [ctx.to_device(a) for a in (rowptr, col, val, z)]
out = ctx.alloc(rows * 32)
for _ in range(3):
    ctx.spmv_fr_dev(d[0], d[1], d[2], d[3], out, rows)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.spmv_fr_dev(d[0], d[1], d[2], d[3], out, rows)
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"rows": rows, "nnz": rows, "matvecs": reps + 3, "ms_per_matvec": dt * 1e3,
                  "algorithmic_GBps": (68 * rows + 36 * rows) / dt / 1e9}))
