"""Soak: proofs of seven sizes (2^10 .. 2^20 constraints) interleaved on one context, 40 rounds, every proof verified; every fourth
round draws through the caller-owned generator's callback.  usage: python3 tools/soak.py"""
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
        # every fourth repetition through the fill_bytes callback (the caller-owned generator's path: host ring, mask in pieces)
        proof = M.generate_proof(cs, pk, M.rng_behind_callback(M.generate_rand()) if rep % 4 == 3 else rng)
        ok = M.verify_proof(vk, public, proof, M.generate_rand())
        cnt += 1; bad += (not ok)
print("soak: %d proofs, %d rejected, %.1f s" % (cnt, bad, time.time() - t0))
