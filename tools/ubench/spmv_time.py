import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
import numpy as np
import simpleworks_amd as swm
from pyref.prng import fr_array
ctx = swm.Context(0)
rng = np.random.default_rng(5)
for lg in (16, 20, 22):
    rows = 1 << lg
    for name, counts in (("1 per row (synthetic R1CS)", np.ones(rows, dtype=np.int64)),
                         ("Poisson(3) per row", rng.poisson(3.0, rows))):
        rowptr = np.zeros(rows + 1, dtype=np.uint32); rowptr[1:] = np.cumsum(counts)
        nnz = int(rowptr[-1])
        col = rng.integers(0, rows, nnz, dtype=np.uint32)
        base = fr_array(1 << 16, 7)
        val = np.ascontiguousarray(np.tile(base, ((nnz >> 16) + 1, 1))[:nnz])
        z = np.ascontiguousarray(np.tile(base, ((rows >> 16) + 1, 1))[:rows])
        d = [ctx.to_device(a) for a in (rowptr, col, val, z)]
        out = ctx.alloc(rows * 32)
        for _ in range(3): ctx.spmv_fr_dev(d[0], d[1], d[2], d[3], out, rows)
        ctx.synchronize(); t0 = time.perf_counter(); reps = 20
        for _ in range(reps): ctx.spmv_fr_dev(d[0], d[1], d[2], d[3], out, rows)
        ctx.synchronize(); dt = (time.perf_counter() - t0) / reps
        alg = 68 * nnz + 36 * rows
        print(f"spmv rows=2^{lg} nnz={nnz:9d} {name:28s}: {dt*1e3:7.3f} ms  {alg/dt/1e9:7.1f} GB/s algorithmic ({alg/dt/8e12*100:4.1f} % of HBM)")
        for b in d: b.free()
        out.free()
