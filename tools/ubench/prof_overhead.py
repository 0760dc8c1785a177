import sys, time
sys.path.insert(0, ".")
import simpleworks_amd as swm
from simpleworks_amd import marlin as M, workloads as W
ctx = swm.Context(0); M.set_default_context(ctx)
n = 1 << 20
rng = M.generate_rand()
srs = M.generate_universal_srs(n, n, n, rng)
cs, public = W.synthetic_r1cs(n, 3, 5)
pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
for prof in (False, True, False, True):
    ctx.profile_reset(); ctx.profile_enable(prof)
    for _ in range(2): M.generate_proof(cs, pk, rng)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(6): M.generate_proof(cs, pk, rng)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 6
    ctx.profile_enable(False)
    print("profiling", prof, "prove %.2f ms" % (dt * 1e3))
