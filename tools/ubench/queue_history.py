"""Hardware-queue placement under a different stream-creation history: k extra streams created (and used) by the host
application before the first proof.  usage: queue_history.py <k> [log_n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
k = int(sys.argv[1]); lg = int(sys.argv[2]) if len(sys.argv) > 2 else 18
keep = []
x = torch.zeros(16, device="cuda")
for i in range(k):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x = x + 1
    keep.append(s)
torch.cuda.synchronize()
from simpleworks_amd import marlin as M, workloads as W
n = 1 << lg
rng = M.generate_rand()
srs = M.generate_universal_srs(n, n, n, rng)
cs, public = W.synthetic_r1cs(n, 3 + lg, 5)
pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
srs.free()
M.generate_proof(cs, pk, rng)
ts = []
for _ in range(7):
    t0 = time.perf_counter(); proof = M.generate_proof(cs, pk, rng); ts.append((time.perf_counter() - t0) * 1e3)
assert M.verify_proof(vk, public, proof, M.generate_rand())
ts.sort()
print("extra streams %d: prove 2^%d median %.2f ms" % (k, lg, ts[3]), flush=True)
