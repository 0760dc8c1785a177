// dropin_harness.cpp — the per-proof call sequence of the Rust binding, replayed against the C ABI without Rust.
//
// What it stands for: swmarlin-sys/src/marlin.rs `prove()` behind the reference's
//     pub fn generate_proof(constraint_system: ConstraintSystemRef, proving_key: ProvingKey, rng: &mut StdRng)
//                                                                     (/root/reference/src/marlin/mod.rs:70-77)
// i.e. everything a caller pays per proof ON TOP of swm_generate_proof.  One proving thread of the binding does, per proof:
//   1. key lookup   — serialise the verifying key of the ProvingKey it was handed, Blake2s it, look the digest up in the
//                     process-wide per-device cache of resident keys (a mutex + a map), swm_pk_attach on first sight;
//   2. pack         — hand the two assignment vectors of the live ConstraintSystem to the library: a pointer view of
//                     `Vec<Fr>` (Fp256 = its four Montgomery limbs; mode 0, the binding's default after its layout check),
//                     or a copy into a flat u64 buffer (mode 1, the fallback when the check fails).  NO matrices: the nine
//                     matrix pointers of struct swm_r1cs are NULL (include/swmarlin.h, "ASSIGNMENT ONLY");
//   3. prove        — swm_generate_proof on the thread's own context with the SHARED key;
//   4. proof out    — the bytes come back in the serialize_uncompressed form (swm_generate_proof_ex, SWM_PROOF_UNCOMPRESSED) and
//                     the binding reads them with Proof::deserialize_unchecked: a field-by-field copy, no arithmetic.  What the
//                     r05 shim did instead — the checked Proof::deserialize of the compressed bytes — is measured beside it
//                     through its stand-in, the library's own checked reader (swm_proof_validate: every point decompressed
//                     and subgroup-checked, as ark-serialize's checked path does), outside the timed loop.
// T threads run that loop concurrently, each with a context of its own, all holding ONE swm_pk.  The harness reports wall
// time, per-step host times and the HBM in use before / after the threads attached (one key, not T).
//
// Test infrastructure (tests/test_gpu_dropin.py) and a measurement leg of bench.py (`drop_in`); not part of the product.
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "swmarlin.h"

namespace {

using clk = std::chrono::steady_clock;
double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

// the binding's process-wide cache of resident keys: digest of (vk bytes) -> handle, per device
struct KeyCache {
    std::mutex mu;
    std::map<std::string, swm_pk*> keys;
    swm_pk* get(const uint8_t digest[32]) {
        std::lock_guard<std::mutex> g(mu);
        auto it = keys.find(std::string((const char*)digest, 32));
        return it == keys.end() ? nullptr : it->second;
    }
    void put(const uint8_t digest[32], swm_pk* pk) {
        std::lock_guard<std::mutex> g(mu);
        keys[std::string((const char*)digest, 32)] = pk;
    }
};

struct ThreadOut {
    int status = SWM_OK;
    std::string error;
    double key_ms = 0, pack_ms = 0, prove_ms = 0, validate_ms = 0;
    std::vector<uint8_t> last_proof, last_unc;
    swm_rng* rng = nullptr;  // the thread's generator: one stream over warm-up and timed proofs
};

}  // namespace

extern "C" {

// One run.  `pk` / `vk`: the key every thread proves with (the caller keeps its own reference).  instance / witness: the
// assignment as the live constraint system holds it (n x 4 Montgomery limbs).  pack_mode 0: pointer view, 1: copy per proof.
// rng_key / rng_word_pos: every thread's generator is the ChaCha12 stream (key, position) — "the caller's StdRng" handed over by
// state (swm_rng_from_chacha); with the position of ark_std::test_rng() after the universal set-up every thread's FIRST proof is
// the golden one.  proofs_out: threads x 1024 bytes, proof_lens: threads entries — the last proof of every thread (with
// proofs_per_thread == 1 the caller compares them with its golden bytes; otherwise with each other: same key, same assignment,
// same stream).  json: the report.  Returns SWM_OK or the first failing status.
int dropin_run(int device, swm_pk* pk, const swm_vk* vk, const uint64_t* instance, size_t num_instance, const uint64_t* witness,
               size_t num_witness, size_t num_constraints, int threads, int proofs_per_thread, int pack_mode,
               const uint8_t rng_key[32], uint64_t rng_word_pos, uint8_t* proofs_out, size_t* proof_lens, char* json,
               size_t json_cap) {
    if (!pk || !vk || !instance || threads < 1 || threads > 64 || proofs_per_thread < 1 || !json || !rng_key) return SWM_ERR_INVALID_ARG;
    KeyCache cache;
    // the cache already holds the key: the binding put it there when it indexed (swmarlin-sys: index() -> pk_put)
    uint8_t vkb[4096], dg[32];
    size_t vklen = 0;
    int rc = swm_vk_serialize(vk, vkb, sizeof vkb, &vklen);
    if (rc != SWM_OK) return rc;
    swm_blake2s(vkb, vklen, dg);
    cache.put(dg, pk);

    swm_ctx* probe = nullptr;
    rc = swm_init(device, &probe);
    if (rc != SWM_OK) return rc;
    size_t free0 = 0, free1 = 0, total = 0;
    swm_device_mem_info(probe, &free0, &total);

    std::vector<ThreadOut> outs(threads);
    std::vector<swm_ctx*> ctxs(threads, nullptr);
    std::vector<swm_pk*> held(threads, nullptr);
    // set-up per thread (the binding's thread-local State::new + the first resident_pk): not part of the per-proof cost
    for (int t = 0; t < threads; t++) {
        rc = swm_init(device, &ctxs[t]);
        if (rc != SWM_OK) break;
        rc = swm_pk_attach(ctxs[t], pk);
        if (rc != SWM_OK) break;
        held[t] = pk;
    }
    if (rc == SWM_OK) {
        auto body = [&](int t, int count, bool timed) {
            ThreadOut& o = outs[t];
            swm_rng* rng = o.rng;
            int r = SWM_OK;
            if (!rng) r = swm_rng_from_chacha(rng_key, rng_word_pos, 12, &rng);
            o.rng = rng;
            std::vector<uint64_t> inst_copy, wit_copy;
            uint8_t proof[2048];
            for (int i = 0; i < count && r == SWM_OK; i++) {
                // 1. key lookup (resident_pk): vk bytes -> digest -> cache
                auto t0 = clk::now();
                uint8_t b[4096], d[32];
                size_t bl = 0;
                r = swm_vk_serialize(vk, b, sizeof b, &bl);
                if (r != SWM_OK) break;
                swm_blake2s(b, bl, d);
                swm_pk* h = cache.get(d);
                if (!h) { r = SWM_ERR_INTERNAL; break; }
                if (timed) o.key_ms += ms_since(t0);
                // 2. pack (PackedR1cs::assignment_only)
                t0 = clk::now();
                swm_r1cs cs;
                memset(&cs, 0, sizeof cs);  // the nine matrix pointers stay NULL
                cs.num_instance = num_instance;
                cs.num_witness = num_witness;
                cs.num_constraints = num_constraints;
                if (pack_mode == 1) {
                    inst_copy.assign(instance, instance + 4 * num_instance);
                    wit_copy.assign(witness, witness + 4 * num_witness);
                    cs.instance = inst_copy.data();
                    cs.witness = num_witness ? wit_copy.data() : nullptr;
                } else {
                    cs.instance = instance;
                    cs.witness = num_witness ? witness : nullptr;
                }
                if (timed) o.pack_ms += ms_since(t0);
                // 3. prove; 4. proof out: uncompressed bytes, copied out as deserialize_unchecked would walk them
                t0 = clk::now();
                size_t len = 0;
                uint8_t unc[4096];
                r = swm_generate_proof_ex(ctxs[t], h, &cs, rng, SWM_PROOF_UNCOMPRESSED, unc, sizeof unc, &len);
                if (r != SWM_OK) {
                    o.error = swm_last_error(ctxs[t]);
                    break;
                }
                o.last_unc.assign(unc, unc + len);
                if (timed) o.prove_ms += ms_since(t0);
            }
            if (r == SWM_OK && !o.last_unc.empty()) {
                // outside the timed loop: the compressed bytes (what the caller compares with its golden proof), and what the
                // checked deserialisation of THOSE would have cost per proof (the r05 shim's Proof::deserialize; proxy)
                size_t len = 0;
                r = swm_proof_recode(o.last_unc.data(), o.last_unc.size(), 0, proof, sizeof proof, &len);
                if (r == SWM_OK) {
                    auto t0 = clk::now();
                    r = swm_proof_validate(proof, len);
                    o.validate_ms = ms_since(t0) * count;  // reported per proof below
                    o.last_proof.assign(proof, proof + len);
                }
            }
            o.status = r;
        };
        // warm-up: one proof per thread, one thread at a time (first-touch of every context's scratch and twiddle tables)
        const int warm = proofs_per_thread >= 6 ? 3 : (proofs_per_thread > 1 ? 1 : 0);  // (a fresh context's first proofs grow its scratch)
        for (int t = 0; t < threads && warm; t++) body(t, warm, false);
        swm_device_mem_info(probe, &free1, &total);
        const int count = proofs_per_thread - warm;
        auto w0 = clk::now();
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back(body, t, count, true);
        for (auto& x : th) x.join();
        const double wall = ms_since(w0);
        if (proofs_per_thread == 1) swm_device_mem_info(probe, &free1, &total);
        double key = 0, pack = 0, prove = 0, val = 0;
        for (int t = 0; t < threads; t++) {
            if (outs[t].status != SWM_OK && rc == SWM_OK) rc = outs[t].status;
            key += outs[t].key_ms; pack += outs[t].pack_ms; prove += outs[t].prove_ms; val += outs[t].validate_ms;
            if (proofs_out && proof_lens) {
                proof_lens[t] = std::min<size_t>(outs[t].last_proof.size(), 1024);
                memcpy(proofs_out + (size_t)t * 1024, outs[t].last_proof.data(), proof_lens[t]);
            }
        }
        const double np = (double)threads * count;
        std::string err;
        for (int t = 0; t < threads; t++)
            if (!outs[t].error.empty()) err = outs[t].error;
        for (auto& c : err)
            if (c == '"' || c == '\\' || c == '\n') c = ' ';
        snprintf(json, json_cap,
                 "{\"threads\": %d, \"proofs\": %d, \"pack_mode\": \"%s\", \"wall_ms\": %.3f, \"ms_per_proof\": %.3f, "
                 "\"latency_ms_per_proof\": %.3f, \"key_lookup_ms\": %.4f, \"host_pack_ms\": %.4f, \"prove_call_ms\": %.3f, "
                 "\"checked_deserialize_proxy_ms\": %.4f, \"binding_overhead_ms\": %.4f, \"hbm_used_by_attach_and_contexts_bytes\": %lld, "
                 "\"hbm_free_before_bytes\": %zu, \"pk_refcount\": %d, \"status\": %d, \"error\": \"%s\"}",
                 threads, (int)np, pack_mode == 1 ? "copy" : "view", wall, wall / np, wall / count, key / np, pack / np, prove / np,
                 val / np, (key + pack) / np, (long long)free0 - (long long)free1, free0, swm_pk_refcount(pk), rc, err.c_str());
    }
    for (int t = 0; t < threads; t++) {
        if (outs[t].rng) swm_rng_free(outs[t].rng);
        if (held[t]) swm_pk_destroy(ctxs[t], held[t]);  // drops this thread's reference; the caller's keeps the key alive
        if (ctxs[t]) swm_destroy(ctxs[t]);
    }
    swm_destroy(probe);
    return rc;
}

}  // extern "C"
