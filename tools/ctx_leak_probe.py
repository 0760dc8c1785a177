import sys, os
sys.path.insert(0, os.getcwd())
import simpleworks_amd as swm
from simpleworks_amd import marlin as M, workloads as W
base = swm.Context(0)
n = 1 << 12
rng = M.generate_rand()
srs = M.generate_universal_srs(n, n, n, rng, ctx=base)
cs, public = W.synthetic_r1cs(n, 11, 13)
pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
M.generate_proof(cs, pk, M.generate_rand())
f0 = base.mem_info()[0]
for i in range(41):
    c = swm.Context(0)
    k = pk.attach(c)
    if i % 2 == 0:
        M.generate_proof(cs, k, M.generate_rand())
    k.free()
    c.close() if hasattr(c, "close") else None
    del c
    if i % 10 == 0:
        print(i, f0 - base.mem_info()[0])
