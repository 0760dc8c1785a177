import os, sys
sys.path.insert(0, os.getcwd())
from simpleworks_amd import marlin as M, workloads as W
n = 1 << 20
rng = M.generate_rand()
srs = M.generate_universal_srs(n, n, n, rng)
cs, public = W.synthetic_r1cs(n, 0x1234567, 0x7654321)
pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
for _ in range(4):
    M.generate_proof(cs, pk, rng)
os.environ["SWM_TRACE_ON"] = "1"
sys.stderr.write("=== traced proof\n")
M.generate_proof(cs, pk, rng)
