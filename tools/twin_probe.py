"""Does the twin path (msm.h: MsmTwin) engage?  python tools/twin_probe.py [log2 constraints]: jobs that took their twin's sort."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpleworks_amd import marlin as M, workloads as W
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
rng = M.generate_rand()
srs = M.generate_universal_srs(n, n, n, rng)
cs, public = W.synthetic_r1cs(n, 3 + lg, 5)
pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
ctx = M.default_context()
for i in range(3):
    ctx.profile_reset()
    t = time.time()
    proof = M.generate_proof(cs, pk, M.generate_rand())
    dt = time.time() - t
    ctx.profile()
    print(lg, "twins", ctx.last_work["msm_twins"], "msm calls", ctx.last_work["msm_calls"], "ms", round(dt * 1e3, 2))
assert M.verify_proof(vk, public, proof, M.generate_rand())
