import sys, time
sys.path.insert(0, ".")
from simpleworks_amd import marlin as M, workloads as W
rng = M.generate_rand()
keys = []
for lg in (10, 12, 14, 16, 17, 18, 20):
    n = 1 << lg
    srs = M.generate_universal_srs(n, n, n, rng)
    cs, public = W.synthetic_r1cs(n, 0x99 + lg, 0x1234)
    pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
    srs.free()
    keys.append((lg, cs, public, pk, vk))
t0 = time.time(); cnt = 0; bad = 0
for rep in range(40):
    for lg, cs, public, pk, vk in keys:
        if lg == 20 and rep % 2: continue
        proof = M.generate_proof(cs, pk, rng)
        ok = M.verify_proof(vk, public, proof, M.generate_rand())
        cnt += 1; bad += (not ok)
print("soak: %d proofs, %d rejected, %.1f s" % (cnt, bad, time.time() - t0))
