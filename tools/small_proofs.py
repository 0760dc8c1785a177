#!/usr/bin/env python3
"""Latency of small proofs (2^10 .. 2^16 constraints): wall time per proof, median of `reps`, on one context.
usage: small_proofs.py [log_n ...]   (SWM_TRACE=1 adds the per-phase breakdown of the library on stderr)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simpleworks_amd import marlin as M, workloads as W  # noqa: E402

logs = [int(a) for a in sys.argv[1:]] or [12, 14, 16]
reps = int(os.environ.get("REPS", "7"))
for lg in logs:
    n = 1 << lg
    rng = M.generate_rand()
    srs = M.generate_universal_srs(n, n, n, rng)
    cs, public = W.synthetic_r1cs(n, 3 + lg, 5)
    pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
    srs.free()
    M.generate_proof(cs, pk, rng)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        proof = M.generate_proof(cs, pk, rng)
        ts.append((time.perf_counter() - t0) * 1e3)
    assert M.verify_proof(vk, public, proof, M.generate_rand())
    ts.sort()
    print("prove 2^%d: median %.2f ms  min %.2f  max %.2f" % (lg, ts[len(ts) // 2], ts[0], ts[-1]), flush=True)
    pk.free()
