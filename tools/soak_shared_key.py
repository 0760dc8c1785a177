"""Soak of the shared-key path: T host threads, a context each, ONE resident proving key, P proofs per thread through the C++ drop-in
harness (tests/native/dropin_harness.cpp); every thread draws from the same ChaCha stream position, so all threads must end on the
same proof bytes, and those must verify.   usage: soak_shared_key.py [log_n=16] [threads=4] [proofs=50]"""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import simpleworks_amd as swm
from simpleworks_amd import marlin as M, workloads as W
import dropin_lib
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
P = int(sys.argv[3]) if len(sys.argv) > 3 else 50
n = 1 << lg
ctx = swm.Context(0)
rng = M.generate_rand()
srs = M.generate_universal_srs(n, n, n, rng, ctx=ctx)
cs, public = W.synthetic_r1cs(n, 11, 13)
pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
srs.free()
free0 = ctx.mem_info()[0]
t0 = time.perf_counter()
rep, proofs = dropin_lib.run(pk, vk, cs.pack_assignment(), threads=T, proofs_per_thread=P, rng_key=M.TEST_RNG_SEED, rng_word_pos=rng.word_pos())
dt = time.perf_counter() - t0
assert len(set(proofs)) == 1, "threads disagree on the proof bytes"
assert M.verify_proof(vk, public, M.MarlinProof(proofs[0]), M.generate_rand()), "proof does not verify"
assert pk.refcount == 1
free1 = ctx.mem_info()[0]
# a second batch: what the first one did not give back is the runtime's (stream and queue pools, constant in the proof size), not a
# leak of the library's — the second batch must not add to it
rep2, proofs2 = dropin_lib.run(pk, vk, cs.pack_assignment(), threads=T, proofs_per_thread=P, rng_key=M.TEST_RNG_SEED, rng_word_pos=rng.word_pos())
assert proofs2[0] == proofs[0]
free2 = ctx.mem_info()[0]
print(json.dumps({"log_n": lg, "threads": T, "proofs": rep["proofs"], "ms_per_proof": rep["ms_per_proof"], "latency_ms_per_proof": rep["latency_ms_per_proof"],
                  "identical_on_all_threads": True, "verifies": True, "hbm_not_returned_bytes": free0 - free1, "hbm_not_returned_after_second_batch_bytes": free0 - free2, "seconds": round(dt, 2)}))
