"""Wall time of the one-time calls at 2^16 and 2^20 constraints: generate_universal_srs, generate_proving_and_verifying_keys
(arithmetisation, 12 index commitments, window tables).  r03, one MI355X: 30 / 41 ms and 0.12 - 0.18 / 0.55 s."""
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simpleworks_amd import marlin as M, workloads as W
ctx = M.default_context()
for lg in (16, 20):
    n = 1 << lg
    rng = M.generate_rand()
    ctx.synchronize(); t0 = time.perf_counter()
    srs = M.generate_universal_srs(n, n, n, rng)
    ctx.synchronize(); t1 = time.perf_counter()
    cs, public = W.synthetic_r1cs(n, 3, 5)
    t2 = time.perf_counter()
    pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
    ctx.synchronize(); t3 = time.perf_counter()
    print("2^%d: universal_setup %.0f ms, circuit (python) %.0f ms, index %.0f ms" % (lg, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), flush=True)
    for _ in range(2):
        t4 = time.perf_counter(); M.generate_proving_and_verifying_keys(srs, cs)[0].free(); ctx.synchronize(); print("   index again %.0f ms" % ((time.perf_counter()-t4)*1e3))
    pk.free(); srs.free()
