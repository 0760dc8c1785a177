"""Set-up of a key at 2^lg constraints: SRS generation, indexing (key + window tables) and the first proof, with the HBM the key took.
usage: setup_time.py [lg=20]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpleworks_amd import marlin as M, workloads as W
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
ctx = M.default_context()
cs, public = W.synthetic_r1cs(n, 3 + lg, 5)
ctx.synchronize()
f0 = ctx.mem_info()[0]
t = time.perf_counter(); rng = M.generate_rand(); srs = M.generate_universal_srs(n, n, n, rng); ctx.synchronize(); t_srs = time.perf_counter() - t
t = time.perf_counter(); pk, vk = M.generate_proving_and_verifying_keys(srs, cs); ctx.synchronize(); t_idx = time.perf_counter() - t
srs.free()
ctx.synchronize()
f1 = ctx.mem_info()[0]
t = time.perf_counter(); p = M.generate_proof(cs, pk, M.generate_rand()); t_first = time.perf_counter() - t
t = time.perf_counter(); p = M.generate_proof(cs, pk, M.generate_rand()); t_second = time.perf_counter() - t
assert M.verify_proof(vk, public, p, M.generate_rand())
print("2^%d: srs %.3f s, index %.3f s, key %.2f GB of HBM, first proof %.1f ms, second %.1f ms" % (lg, t_srs, t_idx, (f0 - f1) / 2**30, t_first * 1e3, t_second * 1e3))
