import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
import numpy as np
import simpleworks_amd as swm
from pyref.prng import fr_array
ctx = swm.Context(0)
for lg in (16, 18, 20, 22, 24):
    n = 1 << lg
    x = fr_array(min(n, 1 << 20), 5)
    x = np.tile(x, (n // x.shape[0], 1))
    d = ctx.to_device(x)
    for inv, coset in ((0, 0), (1, 0), (0, 1)):
        for _ in range(3): ctx.ntt_fr_dev(d, lg, inv, coset)
        ctx.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps): ctx.ntt_fr_dev(d, lg, inv, coset)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"ntt 2^{lg} inv={inv} coset={coset}: {dt*1e3:7.3f} ms  {n*64/dt/1e9:7.1f} GB/s algorithmic  {n/dt/1e9:6.2f} G elem/s")
