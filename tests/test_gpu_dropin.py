"""generate_proof through the binding costs what the headline says (VERDICT r05 "next" #1):
  * the prover reads ONLY the assignment of the constraint system it is handed (NULL matrix pointers are fine: the matrices
    are the key's, as in ark-marlin's prover_init) — /root/reference/src/marlin/mod.rs:70-77 hands over a live
    ConstraintSystemRef, and the binding no longer flattens A, B, C per proof;
  * a proving key is resident per DEVICE, reference-counted and read-only: two host threads with a context each prove with
    ONE swm_pk at the same time and get the golden bytes; the key outlives the context that built it;
  * tests/native/dropin_harness.cpp replays the Rust shim's per-proof call sequence (key lookup by vk digest, assignment
    pack, swm_generate_proof, proof bytes out) from T threads sharing one key."""
import ctypes
import threading

import numpy as np
import pytest

from oracle_lib import golden, h2i

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    from simpleworks_amd import marlin
    return marlin


@pytest.fixture(scope="module")
def W():
    from simpleworks_amd import workloads
    return workloads


def _setup(M, W, name, ctx=None):
    case = golden("marlin_large.json")[name]
    rng = M.generate_rand()
    srs = M.generate_universal_srs(*case["srs"], rng, ctx=ctx)
    cs, public = W.synthetic_r1cs(case["num_constraints"], h2i(case["a"]), h2i(case["b"]))
    return case, rng, srs, cs, public


def test_assignment_only_null_matrices_and_shape_mismatch(M, W):
    import simpleworks_amd as swm
    ctx = swm.Context(0)
    case, rng, srs, cs, public = _setup(M, W, "synthetic_2p12", ctx)
    pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
    pos = rng.word_pos()
    a = cs.pack_assignment()
    s = a.struct()
    for f in ("a_rowptr", "a_col", "a_val", "b_rowptr", "b_col", "b_val", "c_rowptr", "c_col", "c_val"):
        assert getattr(s, f) is None          # the nine matrix pointers are NULL
    proof = M.generate_proof(a, pk, M.rng_from_chacha(M.TEST_RNG_SEED, pos))
    assert proof.data.hex() == case["proof"]
    # ... and the full struct (matrices present, as r01 - r05 bindings passed it) gives the same bytes
    s2 = cs.struct()
    buf = (ctypes.c_uint8 * 2048)()
    n = ctypes.c_size_t(0)
    r2 = M.rng_from_chacha(M.TEST_RNG_SEED, pos)
    assert ctx.lib.swm_generate_proof(ctx.h, pk.h, ctypes.byref(s2), r2.h, buf, len(buf), ctypes.byref(n)) == 0
    assert bytes(buf[: n.value]).hex() == case["proof"]
    assert M.verify_proof(vk, public, proof, M.generate_rand())
    # the uncompressed form (swm_generate_proof_ex) is the same proof: converted on the host it gives the golden bytes
    from simpleworks_amd import serialization as S
    unc = M.generate_proof_uncompressed(a, pk, M.rng_from_chacha(M.TEST_RNG_SEED, pos))
    assert len(unc) > len(proof.data) and S.proof_recode(unc, False).hex() == case["proof"]
    assert S.proof_recode(proof.data, True) == unc
    # a shape that is not the key's: InstanceDoesNotMatchIndex
    bad = M.AssignmentOnly(a.instance, a.witness[:-1], a.num_constraints - 1)
    with pytest.raises(M.MarlinError) as e:
        M.generate_proof(bad, pk, M.rng_from_chacha(M.TEST_RNG_SEED, pos))
    assert e.value.code == -8
    # an unsatisfying assignment still fails at prove time without the matrices
    w2 = a.witness.copy()
    w2[0, 0] ^= 1                             # witness 0 is the circuit's `a` (every row reads it)
    with pytest.raises(M.MarlinError) as e:
        M.generate_proof(M.AssignmentOnly(a.instance, w2, a.num_constraints), pk, M.rng_from_chacha(M.TEST_RNG_SEED, pos))
    assert e.value.code == -5
    pk.free()
    srs.free()
    ctx.close()


@pytest.mark.parametrize("name", ["synthetic_2p12", "synthetic_2p16"])
def test_two_threads_two_contexts_one_key_golden_bytes(M, W, name):
    import simpleworks_amd as swm
    ctx_a, ctx_b = swm.Context(0), swm.Context(0)
    free0, _ = ctx_a.mem_info()
    case, rng, srs, cs, public = _setup(M, W, name, ctx_a)
    pk_a, vk = M.generate_proving_and_verifying_keys(srs, cs)
    srs.free()
    ctx_a.synchronize()
    free_key, _ = ctx_a.mem_info()
    pos = rng.word_pos()
    assert pk_a.refcount == 1
    pk_b = pk_a.attach(ctx_b)
    assert pk_b.h.value == pk_a.h.value and pk_a.refcount == 2
    free_att, _ = ctx_a.mem_info()
    assert free_att == free_key                 # attaching allocates nothing: the second holder reads the same tables
    a = cs.pack_assignment()
    # the working set of ONE context (pool blocks, MSM scratch, result slots): proofs alone on the first context (several: the result
    # slots rotate through a proof's jobs, and each slot's scratch grows to the largest job it has met)
    for _ in range(4):
        assert M.generate_proof(a, pk_a, M.rng_from_chacha(M.TEST_RNG_SEED, pos)).data.hex() == case["proof"]
    ctx_a.synchronize()
    free_solo, _ = ctx_a.mem_info()
    scratch_solo = free_key - free_solo
    out, errs = {}, []

    def run(tag, pk):
        try:
            for _ in range(3):                  # three proofs each, concurrently, every one from the golden stream position
                out[tag] = M.generate_proof(a, pk, M.rng_from_chacha(M.TEST_RNG_SEED, pos)).data.hex()
                assert out[tag] == case["proof"], tag
        except Exception as e:                  # noqa: BLE001
            errs.append((tag, e))
    ths = [threading.Thread(target=run, args=("a", pk_a)), threading.Thread(target=run, args=("b", pk_b))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert out["a"] == out["b"] == case["proof"]
    ctx_a.synchronize()
    ctx_b.synchronize()
    free_end, total = ctx_a.mem_info()
    key_bytes, both = free0 - free_key, free0 - free_end
    print("\n[shared key %s] key %.1f MB resident; one context's working set %.1f MB; after two contexts proved with it: %.1f MB in "
          "use = %.2f x the key" % (name, key_bytes / 1e6, scratch_solo / 1e6, both / 1e6, both / max(key_bytes, 1)))
    if name == "synthetic_2p16":
        # one key + the state of two contexts, not two keys: the second context added a working set like the first one's (measured
        # alone above) plus its own transform tables (the first context's were allocated while it built the key and count as "key"
        # here) — well under another copy of the key's window tables
        added = free_solo - free_end
        assert added < scratch_solo + key_bytes // 2, (added, scratch_solo, key_bytes)
    # the key outlives the context that built it
    pk_a.free()
    ctx_a.close()
    assert pk_b.refcount == 1
    p = M.generate_proof(a, pk_b, M.rng_from_chacha(M.TEST_RNG_SEED, pos))
    assert p.data.hex() == case["proof"]
    pk_b.free()
    ctx_b.close()


def test_attach_checks_the_device(M, W):
    import torch
    import simpleworks_amd as swm
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: a key on device 0 must be refused by a context on device 1")
    ctx0, ctx1 = swm.Context(0), swm.Context(1)
    case, rng, srs, cs, public = _setup(M, W, "synthetic_2p12", ctx0)
    pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
    with pytest.raises(M.MarlinError) as e:
        pk.attach(ctx1)
    assert e.value.code == -8
    pk.free()
    srs.free()


def test_dropin_harness_threads_share_one_key(M, W):
    import dropin_lib
    import simpleworks_amd as swm
    ctx = swm.Context(0)
    case, rng, srs, cs, public = _setup(M, W, "synthetic_2p12", ctx)
    pk, vk = M.generate_proving_and_verifying_keys(srs, cs)
    srs.free()
    pos = rng.word_pos()
    a = cs.pack_assignment()
    for mode in ("view", "copy"):
        rep, proofs = dropin_lib.run(pk, vk, a, threads=3, proofs_per_thread=1, rng_key=M.TEST_RNG_SEED, rng_word_pos=pos, pack_mode=mode)
        assert rep["status"] == 0 and rep["threads"] == 3 and rep["proofs"] == 3
        assert all(p.hex() == case["proof"] for p in proofs), mode      # every thread's proof is the golden one
        assert rep["pk_refcount"] == 4                                   # the caller's reference + one per thread
    assert pk.refcount == 1                                              # every thread let go
    rep, proofs = dropin_lib.run(pk, vk, a, threads=4, proofs_per_thread=4, rng_key=M.TEST_RNG_SEED, rng_word_pos=pos)
    assert len(set(proofs)) == 1 and rep["proofs"] == 12                 # same key, assignment and stream: same bytes on every thread
    assert M.verify_proof(vk, public, M.MarlinProof(proofs[0]), M.generate_rand())
    assert rep["binding_overhead_ms"] < 1.0, rep
    print("\n[dropin 2^12, 4 threads] %s" % rep)
    pk.free()
    ctx.close()
