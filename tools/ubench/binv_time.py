import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
import numpy as np
import simpleworks_amd as swm
from pyref.prng import fr_array
ctx = swm.Context(0)
base = fr_array(1 << 20, 5)
for n in (16, 4096, 1 << 14, 1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 22):
    x = np.ascontiguousarray(np.tile(base, ((n + (1 << 20) - 1) >> 20, 1))[:n])
    d = ctx.to_device(x)
    for _ in range(3): ctx.batch_inverse_fr_dev(d, n)
    ctx.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps): ctx.batch_inverse_fr_dev(d, n)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"batch_inverse n={n}: {dt*1e3:7.3f} ms  {n/dt/1e9:6.2f} G elem/s")
