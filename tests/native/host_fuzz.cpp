// Mutation harness for the host-side, attacker-facing surface of libswmarlin: the proof / verifying-key / proving-key codecs
// and the verifier (csrc/host/marlin_types.h, ahp.h, pairing.h, pk_codec.h) behind the SAME entry-point text the library
// ships (csrc/host/host_abi.inc, compiled here by g++ with -fsanitize=address,undefined; no GPU, no HIP runtime call).
// The reference forbids panics on bad bytes (/root/reference/src/lib.rs:28) and every deserialiser returns Err
// (src/marlin/serialization.rs:14-17,26-31,40-45): here every rejection must be a status code, and no input may trip a
// sanitizer.
//
//   host_fuzz <vk.bin> <proof.bin> <public_inputs.bin> <pk.bin> <seed> <n_proof> <n_vk> <n_pk>
// prints "OK ..." with the tallies, exits 1 on the first violation.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "swmarlin.h"
#include "host/ahp.h"
#include "host/pk_codec.h"
#include "host/host_handles.h"

namespace swm {
// what capi.hip provides in the library: the detail text of a failure, and the stream drain of a failed GPU entry point
int set_err(swm_ctx*, int code, const char*, ...) { return code; }
void drain_streams(swm_ctx*) {}
}  // namespace swm
using namespace swm;
#include "host/host_abi.inc"

// the host part of swm_pk_deserialize (marlin.hip: pk_deserialize up to the committer key) under the library's guard
static int pk_prefix_status(const uint8_t* bytes, size_t len, size_t* consumed) {
    swm_ctx* none = nullptr;
    SWM_GUARD(none, {
        ByteReader r(bytes, len);
        PkPrefix pre = pk_parse_prefix(r);
        *consumed = r.pos;
    });
}

struct Rng {  // xorshift64*: the mutations are a function of the seed alone
    uint64_t s;
    uint64_t next() {
        s ^= s >> 12;
        s ^= s << 25;
        s ^= s >> 27;
        return s * 0x2545F4914F6CDD1Dull;
    }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
};

static std::vector<uint8_t> read_file(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path);
        exit(2);
    }
    std::vector<uint8_t> b;
    uint8_t buf[4096];
    size_t k;
    while ((k = fread(buf, 1, sizeof(buf), f)) > 0) b.insert(b.end(), buf, buf + k);
    fclose(f);
    return b;
}

// one mutation of `src`; `kind` is returned for the tallies
static std::vector<uint8_t> mutate(const std::vector<uint8_t>& src, Rng& g, int* kind) {
    std::vector<uint8_t> b = src;
    const uint64_t n = b.size();
    *kind = (int)g.below(9);
    switch (*kind) {
    case 0:  // truncation
        b.resize(g.below(n));
        break;
    case 1: {  // 1 - 3 bit flips
        for (uint64_t k = 1 + g.below(3); k-- > 0 && n;) b[g.below(n)] ^= (uint8_t)(1u << g.below(8));
        break;
    }
    case 2: {  // a length-like field inflated: eight bytes overwritten with a huge / off-by-one count
        static const uint64_t vals[] = {~0ull, 1ull << 63, 1ull << 32, (1ull << 32) - 1, 1ull << 31, 65, 17, 1025, (1ull << 20) + 1, 0};
        if (n >= 8) {
            uint64_t v = vals[g.below(sizeof(vals) / sizeof(vals[0]))];
            size_t at = (size_t)g.below(n - 7);
            if (g.below(2)) at &= ~(size_t)7;
            memcpy(&b[at], &v, 8);
        }
        break;
    }
    case 3:  // one byte replaced
        if (n) b[g.below(n)] = (uint8_t)g.next();
        break;
    case 4: {  // a chunk duplicated in place (shifts everything behind it)
        if (n) {
            size_t at = (size_t)g.below(n), len = (size_t)(1 + g.below(64));
            len = std::min<size_t>(len, n - at);
            b.insert(b.begin() + at, src.begin() + at, src.begin() + at + len);
        }
        break;
    }
    case 5: {  // a chunk removed
        if (n) {
            size_t at = (size_t)g.below(n), len = std::min<size_t>((size_t)(1 + g.below(64)), n - at);
            b.erase(b.begin() + at, b.begin() + at + len);
        }
        break;
    }
    case 6:  // trailing bytes
        for (uint64_t k = 1 + g.below(40); k-- > 0;) b.push_back((uint8_t)g.next());
        break;
    case 7: {  // flag bits of a compressed point / option tags: a byte ORed with 0x40 / 0x80 / 0xC0, or set to 2
        if (n) {
            size_t at = (size_t)g.below(n);
            static const uint8_t m[] = {0x40, 0x80, 0xC0};
            if (g.below(4) == 0) b[at] = 2;
            else b[at] |= m[g.below(3)];
        }
        break;
    }
    default: {  // a 32- or 48-byte window set to all ones / the modulus boundary region (non-canonical field elements)
        if (n >= 48) {
            size_t w = g.below(2) ? 32 : 48, at = (size_t)g.below(n - w + 1);
            memset(&b[at], 0xFF, w);
            if (g.below(2)) b[at + w - 1] = (uint8_t)g.below(64);
        }
        break;
    }
    }
    return b;
}

static bool status_ok(int rc) {  // the status codes of include/swmarlin.h an entry point may return for bad bytes
    return rc == SWM_OK || rc == SWM_ERR_SERIALIZATION || rc == SWM_ERR_INVALID_ARG;
}
#define FAIL(...)                     \
    do {                              \
        fprintf(stderr, __VA_ARGS__); \
        fprintf(stderr, "\n");        \
        return 1;                     \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 9) {
        fprintf(stderr, "usage: host_fuzz vk proof public_inputs pk seed n_proof n_vk n_pk\n");
        return 2;
    }
    const std::vector<uint8_t> vkb = read_file(argv[1]), prb = read_file(argv[2]), pib = read_file(argv[3]), pkb = read_file(argv[4]);
    Rng g{strtoull(argv[5], nullptr, 0) * 0x9E3779B97F4A7C15ull + 0x1234567ull};
    const long n_proof = atol(argv[6]), n_vk = atol(argv[7]), n_pk = atol(argv[8]);
    const size_t npi = pib.size() / 32;
    std::vector<uint64_t> pi(4 * npi + 4);
    memcpy(pi.data(), pib.data(), 32 * npi);

    // ---- the unmutated inputs are what they claim to be
    swm_vk* vk = nullptr;
    if (swm_vk_deserialize(vkb.data(), vkb.size(), &vk) != SWM_OK) FAIL("golden vk rejected");
    swm_rng* rng = nullptr;
    if (swm_rng_test_new(&rng) != SWM_OK) FAIL("rng");
    int ok = 0;
    if (swm_proof_validate(prb.data(), prb.size()) != SWM_OK) FAIL("golden proof rejected by validate");
    if (swm_verify_proof(vk, pi.data(), npi, prb.data(), prb.size(), rng, &ok) != SWM_OK || ok != 1) FAIL("golden proof does not verify");
    {
        std::vector<uint8_t> out(vkb.size() + 16);
        size_t len = 0;
        if (swm_vk_serialize(vk, out.data(), out.size(), &len) != SWM_OK || len != vkb.size() || memcmp(out.data(), vkb.data(), len))
            FAIL("vk round trip differs");
        if (swm_vk_serialize(vk, out.data(), len - 1, &len) != SWM_ERR_INVALID_ARG) FAIL("short vk buffer not refused");
    }
    size_t consumed = 0;
    if (pk_prefix_status(pkb.data(), pkb.size(), &consumed) != SWM_OK || consumed == 0 || consumed >= pkb.size()) FAIL("golden pk prefix rejected");

    long tally[3][3] = {{0}};  // [proof | vk | pk][accepted-by-parser, rejected, verified]
    long kinds[9] = {0};
    // ---- mutated proofs: parse, and if the parser takes them, verify against the golden key
    for (long i = 0; i < n_proof; i++) {
        int kind;
        std::vector<uint8_t> m = mutate(prb, g, &kind);
        kinds[kind]++;
        // (a copy sized exactly to the mutation: ASan then sees any read past its end)
        std::vector<uint8_t> exact(m.begin(), m.end());
        static const uint8_t empty_buf[1] = {0};
        const uint8_t* p = exact.empty() ? empty_buf : exact.data();
        const int rc = swm_proof_validate(p, exact.size());
        if (!status_ok(rc)) FAIL("proof mutation %ld (kind %d): validate returned %d", i, kind, rc);
        tally[0][rc == SWM_OK ? 0 : 1]++;
        // what the parser takes goes to the verifier (which parses again: the two must agree); of the rejected ones every
        // eighth as well — the verifier must refuse them with the same status
        if (rc != SWM_OK && (i & 7) != 0) continue;
        ok = -1;
        const int rv = swm_verify_proof(vk, pi.data(), npi, p, exact.size(), rng, &ok);
        if (rv != rc) FAIL("proof mutation %ld (kind %d): validate returned %d but verify %d", i, kind, rc, rv);
        if (rv == SWM_OK && ok != 0 && ok != 1) FAIL("proof mutation %ld: ok flag %d", i, ok);
        if (rv == SWM_OK && ok == 1) {
            tally[0][2]++;
            // a proof that still verifies must decode to the SAME proof (e.g. the ignored x of a point at infinity cannot
            // occur in a valid proof; re-encoding is the check): anything else would be malleability in the codec
            if (exact != prb) FAIL("proof mutation %ld (kind %d) differs from the golden bytes and still verifies", i, kind);
        }
    }
    // ---- the uncompressed proof form (swm_generate_proof_ex / swm_proof_recode): golden -> uncompressed -> golden, then a
    // quarter as many mutations of the uncompressed bytes through the checked reader of that form
    {
        std::vector<uint8_t> unc(4096), back(4096);
        size_t ul = 0, bl = 0;
        if (swm_proof_recode(prb.data(), prb.size(), 1, unc.data(), unc.size(), &ul) != SWM_OK || ul <= prb.size()) FAIL("recode to uncompressed");
        unc.resize(ul);
        if (swm_proof_recode(unc.data(), unc.size(), 0, back.data(), back.size(), &bl) != SWM_OK || bl != prb.size() ||
            memcmp(back.data(), prb.data(), bl))
            FAIL("uncompressed -> compressed is not the golden proof");
        if (swm_proof_recode(unc.data(), unc.size(), 0, back.data(), bl - 1, &bl) != SWM_ERR_INVALID_ARG) FAIL("short recode buffer not refused");
        for (long i = 0; i < n_proof / 4; i++) {
            int kind;
            std::vector<uint8_t> m = mutate(unc, g, &kind);
            std::vector<uint8_t> exact(m.begin(), m.end());
            static const uint8_t empty_buf[1] = {0};
            const uint8_t* p = exact.empty() ? empty_buf : exact.data();
            std::vector<uint8_t> out(4096);
            size_t ol = 0;
            const int rc = swm_proof_recode(p, exact.size(), 0, out.data(), out.size(), &ol);
            if (!status_ok(rc)) FAIL("uncompressed proof mutation %ld (kind %d): recode returned %d", i, kind, rc);
            tally[0][rc == SWM_OK ? 0 : 1]++;
            if (rc == SWM_OK && swm_proof_validate(out.data(), ol) != SWM_OK) FAIL("uncompressed proof mutation %ld: recoded bytes do not parse", i);
        }
    }
    // ---- mutated verifying keys: parse; survivors are re-serialised, re-parsed and used to verify the golden proof
    for (long i = 0; i < n_vk; i++) {
        int kind;
        std::vector<uint8_t> m = mutate(vkb, g, &kind);
        static const uint8_t empty_buf[1] = {0};
        const uint8_t* p = m.empty() ? empty_buf : m.data();
        swm_vk* v = nullptr;
        const int rc = swm_vk_deserialize(p, m.size(), &v);
        if (!status_ok(rc)) FAIL("vk mutation %ld (kind %d): deserialize returned %d", i, kind, rc);
        if ((rc == SWM_OK) != (v != nullptr)) FAIL("vk mutation %ld: status %d with handle %p", i, rc, (void*)v);
        tally[1][rc == SWM_OK ? 0 : 1]++;
        if (v) {
            size_t len = 0;
            if (swm_vk_serialize(v, nullptr, 0, &len) != SWM_OK) FAIL("vk mutation %ld: length query failed", i);
            std::vector<uint8_t> out(len);
            if (swm_vk_serialize(v, out.data(), out.size(), &len) != SWM_OK || len != out.size()) FAIL("vk mutation %ld: re-serialise failed", i);
            swm_vk* v2 = nullptr;
            if (swm_vk_deserialize(out.data(), out.size(), &v2) != SWM_OK) FAIL("vk mutation %ld: its own bytes are refused", i);
            swm_vk_destroy(v2);
            ok = -1;
            const int rv = swm_verify_proof(v, pi.data(), npi, prb.data(), prb.size(), rng, &ok);
            if (!status_ok(rv)) FAIL("vk mutation %ld (kind %d): verify returned %d", i, kind, rv);
            // (a changed key may still accept: num_instance_variables, max_degree, supported_degree and unused degree bounds
            // do not enter the checks — as in ark-marlin, whose transcript absorbs index_info[0..3] and the commitments only)
            if (rv == SWM_OK && ok == 1) tally[1][2]++;
            swm_vk_destroy(v);
        }
    }
    // ---- mutated proving keys: the host prefix of swm_pk_deserialize (everything in front of the committer key)
    for (long i = 0; i < n_pk; i++) {
        int kind;
        std::vector<uint8_t> m = mutate(pkb, g, &kind);
        if (g.below(2)) {  // half of the mutations inside the prefix, where this parser looks
            m = pkb;
            std::vector<uint8_t> head(pkb.begin(), pkb.begin() + consumed);
            std::vector<uint8_t> mh = mutate(head, g, &kind);
            m.assign(mh.begin(), mh.end());
            m.insert(m.end(), pkb.begin() + consumed, pkb.end());
        }
        static const uint8_t empty_buf[1] = {0};
        size_t used = 0;
        const int rc = pk_prefix_status(m.empty() ? empty_buf : m.data(), m.size(), &used);
        if (!status_ok(rc)) FAIL("pk mutation %ld (kind %d): prefix parse returned %d", i, kind, rc);
        if (rc == SWM_OK && used > m.size()) FAIL("pk mutation %ld: consumed %zu of %zu bytes", i, used, m.size());
        tally[2][rc == SWM_OK ? 0 : 1]++;
    }
    // ---- generators and hashes: argument checks, and the bulk paths of fill_bytes against the word-by-word stream
    {
        swm_rng* r2 = nullptr;
        uint8_t key[32];
        for (int i = 0; i < 32; i++) key[i] = (uint8_t)(7 * i + 1);
        if (swm_rng_from_chacha(key, 0, 13, &r2) != SWM_ERR_INVALID_ARG || r2) FAIL("odd round count accepted");
        if (swm_rng_from_chacha(nullptr, 0, 12, &r2) != SWM_ERR_INVALID_ARG) FAIL("null key accepted");
        for (int t = 0; t < 200; t++) {
            swm_rng *a = nullptr, *b = nullptr;
            const uint64_t pos = g.below(1000);
            if (swm_rng_from_chacha(key, pos, 12, &a) != SWM_OK || swm_rng_from_chacha(key, pos, 12, &b) != SWM_OK) FAIL("rng_from_chacha");
            const size_t len = (size_t)g.below(3000);
            std::vector<uint8_t> x(len), y(len);
            if (swm_rng_fill_bytes(a, len ? x.data() : nullptr, len) != SWM_OK) FAIL("fill_bytes");
            for (size_t i = 0; i < len; i += 4) {  // the same stream word by word
                uint8_t w[4];
                if (swm_rng_fill_bytes(b, w, 4) != SWM_OK) FAIL("fill_bytes(4)");
                memcpy(&y[i], w, std::min<size_t>(4, len - i));
            }
            if (x != y) FAIL("fill_bytes: bulk and word-by-word streams differ (pos %llu, len %zu)", (unsigned long long)pos, len);
            swm_rng_free(a);
            swm_rng_free(b);
        }
        for (int t = 0; t < 300; t++) {
            const size_t len = (size_t)g.below(700);
            std::vector<uint8_t> d(len);
            for (auto& c : d) c = (uint8_t)g.next();
            uint8_t h1[32], h2[32];
            if (swm_blake2s(len ? d.data() : nullptr, len, h1) != SWM_OK || swm_blake2s(len ? d.data() : nullptr, len, h2) != SWM_OK || memcmp(h1, h2, 32))
                FAIL("blake2s");
        }
        uint8_t blk[64];
        if (swm_chacha_block(key, ~0ull, 20, blk) != SWM_OK) FAIL("chacha_block");
    }
    swm_rng_free(rng);
    swm_vk_destroy(vk);
    printf("OK proof: %ld parsed / %ld rejected / %ld verified; vk: %ld parsed / %ld rejected / %ld accept the golden proof; "
           "pk prefix: %ld parsed / %ld rejected; mutation kinds of the proof leg:",
           tally[0][0], tally[0][1], tally[0][2], tally[1][0], tally[1][1], tally[1][2], tally[2][0], tally[2][1]);
    for (int k = 0; k < 9; k++) printf(" %ld", kinds[k]);
    printf("\n");
    return 0;
}
