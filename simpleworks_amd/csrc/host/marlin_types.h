// marlin_types.h — host data model of the Marlin instance fixed by /root/reference/src/marlin/mod.rs:12-26
// (MarlinInst = Marlin<Fr, MarlinKZG10<Bls12_377, DensePolynomial<Fr>>, SimpleHashFiatShamirRng<Blake2s, ChaChaRng>>)
// and its wire formats:
//   * ToBytes (transcript input) and
//   * ark-serialize CanonicalSerialize (src/marlin/serialization.rs:5-45).
// The arkworks crates are not vendored; struct field orders follow ark-marlin / ark-poly-commit 0.3.0 [U].
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "hostmath.h"

namespace swm {

struct MarlinError : std::runtime_error {
    int code;
    MarlinError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// marlin_pc::Commitment { comm, shifted_comm: Option }
struct Commitment {
    G1Affine comm;
    bool has_shifted = false;
    G1Affine shifted;
};
// kzg10::Proof { w, random_v: Option<Fr> }
struct PcProof {
    G1Affine w;
    bool has_random_v = false;
    Fr random_v;
};
// ark_marlin::Proof (prover_messages are three EmptyMessage; BatchLCProof.evals = None)
struct Proof {
    std::vector<std::vector<Commitment>> commitments;
    std::vector<Fr> evaluations;
    std::vector<PcProof> pc_proof;
};
struct IndexInfo {
    uint64_t num_variables = 0, num_constraints = 0, num_non_zero = 0, num_instance_variables = 0;
};
struct PcVerifierKey {
    G1Affine g, gamma_g;
    G2Affine h, beta_h;
    std::vector<std::pair<uint64_t, G1Affine>> degree_bounds_and_shift_powers;
    uint64_t max_degree = 0, supported_degree = 0;
};
struct VerifyingKey {
    IndexInfo info;
    std::vector<Commitment> index_comms;
    PcVerifierKey vk;
};

// ------------------------------------------------------------------------------------------------ byte sinks
struct ByteWriter {
    std::vector<uint8_t> b;
    void raw(const void* p, size_t n) { b.insert(b.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
    void u8(uint8_t v) { b.push_back(v); }
    void u64(uint64_t v) {
        for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i)));
    }
    void fr(const Fr& a) {
        uint8_t t[32];
        fp_to_bytes(a, t);
        raw(t, 32);
    }
    void fq(const Fq& a) {
        uint8_t t[48];
        fp_to_bytes(a, t);
        raw(t, 48);
    }
    // ---- ToBytes (transcript)
    void tb_g1(const G1Affine& p) {  // x || y || infinity; the zero point is (0, 1, true)
        if (g1_is_inf(p)) {
            fq(fp_zero<Fq>());
            fq(fp_one<Fq>());
            u8(1);
        } else {
            fq(p.x);
            fq(p.y);
            u8(0);
        }
    }
    void tb_commitment(const Commitment& c) {  // comm || bool || shifted-or-empty
        tb_g1(c.comm);
        u8(c.has_shifted ? 1 : 0);
        tb_g1(c.has_shifted ? c.shifted : g1_affine_identity());
    }
    // ---- CanonicalSerialize
    // uncompressed = true: G1 points as serialize_uncompressed writes them [U: ark-ec 0.3 short_weierstrass_jacobian.rs] — x, then
    // y with the flags in the top bits of ITS last byte: infinity (bit 6) or none (SWFlags::default() = NegativeY = no bit); the
    // identity is GroupAffine::zero() = (0, 1, infinity).  Field elements, lengths and Option tags are the same in both forms.
    bool uncompressed = false;
    void ser_g1(const G1Affine& p) {  // compressed: x with flags in the top bits of the last byte
        if (uncompressed) {
            uint8_t u[96];
            if (g1_is_inf(p)) {
                fp_to_bytes(fp_zero<Fq>(), u);
                fp_to_bytes(fp_one<Fq>(), u + 48);
                u[95] |= 0x40;
            } else {
                fp_to_bytes(p.x, u);
                fp_to_bytes(p.y, u + 48);
            }
            raw(u, 96);
            return;
        }
        uint8_t t[48];
        if (g1_is_inf(p)) {
            memset(t, 0, 48);
            t[47] |= 0x40;
        } else {
            fp_to_bytes(p.x, t);
            if (fp_cmp(p.y, fp_neg(p.y)) > 0) t[47] |= 0x80;
        }
        raw(t, 48);
    }
    void ser_g2(const G2Affine& p) {
        uint8_t t[96];
        if (p.inf) {
            memset(t, 0, 96);
            t[95] |= 0x40;
        } else {
            fp_to_bytes(p.x.c0, t);
            fp_to_bytes(p.x.c1, t + 48);
            if (fq2_less(-p.y, p.y)) t[95] |= 0x80;
        }
        raw(t, 96);
    }
    void ser_commitment(const Commitment& c) {
        ser_g1(c.comm);
        u8(c.has_shifted ? 1 : 0);
        if (c.has_shifted) ser_g1(c.shifted);
    }
};

struct ByteReader {
    const uint8_t* p;
    size_t n, pos = 0;
    bool uncompressed = false;  // G1 points in the serialize_uncompressed form (ByteWriter::uncompressed); still CHECKED here
    ByteReader(const uint8_t* d, size_t len) : p(d), n(len) {}
    const uint8_t* take(size_t k) {
        if (k > n - pos) throw MarlinError(SWM_ERR_SERIALIZATION, "unexpected end of input");  // (pos <= n always: no wrap for a huge k)
        const uint8_t* r = p + pos;
        pos += k;
        return r;
    }
    uint64_t u64() {
        const uint8_t* t = take(8);
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v |= (uint64_t)t[i] << (8 * i);
        return v;
    }
    bool boolean() {
        uint8_t v = *take(1);
        if (v > 1) throw MarlinError(SWM_ERR_SERIALIZATION, "invalid bool");
        return v == 1;
    }
    Fr fr() {
        Fr r;
        if (!fp_from_bytes(take(32), &r)) throw MarlinError(SWM_ERR_SERIALIZATION, "Fr out of range");
        return r;
    }
    G1Affine g1() {
        if (uncompressed) {  // deserialize_uncompressed: x, y | flags, then on-curve (this reader: always) and subgroup checks
            uint8_t u[96];
            memcpy(u, take(96), 96);
            uint8_t fl = u[95] & 0xC0;
            u[95] &= 0x3F;
            if (fl == 0xC0) throw MarlinError(SWM_ERR_SERIALIZATION, "G1 invalid flags");
            G1Affine r;
            if (!fp_from_bytes(u, &r.x) || !fp_from_bytes(u + 48, &r.y)) throw MarlinError(SWM_ERR_SERIALIZATION, "G1 coordinate out of range");
            if (fl & 0x40) return g1_affine_identity();
            if (!fp_eq(fp_sqr(r.y), fp_add(fp_mul(fp_sqr(r.x), r.x), fp_one<Fq>()))) throw MarlinError(SWM_ERR_SERIALIZATION, "G1 point not on curve");
            if (!g1_is_inf(g1_mul_limbs(r, FrParams::P, 8))) throw MarlinError(SWM_ERR_SERIALIZATION, "G1 point not in the prime-order subgroup");
            return r;
        }
        uint8_t t[48];
        memcpy(t, take(48), 48);
        uint8_t flags = t[47] & 0xC0;
        t[47] &= 0x3F;
        // ark-serialize 0.3 SWFlags::from_u8: 0xC0 (infinity AND sign) is not a valid flag combination -> InvalidData
        if (flags == 0xC0) throw MarlinError(SWM_ERR_SERIALIZATION, "G1 invalid flags");
        G1Affine r;
        if (!fp_from_bytes(t, &r.x)) throw MarlinError(SWM_ERR_SERIALIZATION, "G1 x out of range");
        if (flags & 0x40) return g1_affine_identity();  // ark-ec reads x (must be a field element) and then ignores it
        Fq y2 = fp_add(fp_mul(fp_sqr(r.x), r.x), fp_one<Fq>());
        Fq y;
        if (!fq_sqrt(y2, &y)) throw MarlinError(SWM_ERR_SERIALIZATION, "G1 x not on curve");
        Fq ny = fp_neg(y);
        bool y_is_larger = fp_cmp(y, ny) > 0;
        bool want_larger = (flags & 0x80) != 0;
        r.y = (y_is_larger == want_larger) ? y : ny;
        // CanonicalDeserialize::deserialize is the CHECKED form: the point must lie in the prime-order subgroup
        // (BLS12-377 G1 has a ~2^125 cofactor); ark-ec tests [r]P == O
        if (!g1_is_inf(g1_mul_limbs(r, FrParams::P, 8))) throw MarlinError(SWM_ERR_SERIALIZATION, "G1 point not in the prime-order subgroup");
        return r;
    }
    G2Affine g2() {
        uint8_t t[96];
        memcpy(t, take(96), 96);
        uint8_t flags = t[95] & 0xC0;
        t[95] &= 0x3F;
        if (flags == 0xC0) throw MarlinError(SWM_ERR_SERIALIZATION, "G2 invalid flags");
        G2Affine r;
        r.inf = false;
        if (!fp_from_bytes(t, &r.x.c0) || !fp_from_bytes(t + 48, &r.x.c1))
            throw MarlinError(SWM_ERR_SERIALIZATION, "G2 x out of range");
        if (flags & 0x40) return g2_identity();
        Fq2 y;
        if (!fq2_sqrt(r.x.square() * r.x + g2_coeff_b(), &y)) throw MarlinError(SWM_ERR_SERIALIZATION, "G2 x not on curve");
        Fq2 ny = -y;
        bool y_is_larger = fq2_less(ny, y);
        bool want_larger = (flags & 0x80) != 0;
        r.y = (y_is_larger == want_larger) ? y : ny;
        if (!g2_mul(r, FrParams::P, 8).inf) throw MarlinError(SWM_ERR_SERIALIZATION, "G2 point not in the prime-order subgroup");
        return r;
    }
    Commitment commitment() {
        Commitment c;
        c.comm = g1();
        c.has_shifted = boolean();
        if (c.has_shifted) c.shifted = g1();
        return c;
    }
};

// ------------------------------------------------------------------------------------------------ proof / vk codecs
inline std::vector<uint8_t> serialize_proof(const Proof& pr, bool uncompressed = false) {
    ByteWriter w;
    w.uncompressed = uncompressed;
    w.u64(pr.commitments.size());
    for (auto& rnd : pr.commitments) {
        w.u64(rnd.size());
        for (auto& c : rnd) w.ser_commitment(c);
    }
    w.u64(pr.evaluations.size());
    for (auto& e : pr.evaluations) w.fr(e);
    w.u64(3);  // prover_messages: 3 x EmptyMessage -> Option::None
    w.u8(0);
    w.u8(0);
    w.u8(0);
    w.u64(pr.pc_proof.size());
    for (auto& p : pr.pc_proof) {
        w.ser_g1(p.w);
        w.u8(p.has_random_v ? 1 : 0);
        if (p.has_random_v) w.fr(p.random_v);
    }
    w.u8(0);  // BatchLCProof.evals = None
    return w.b;
}

inline Proof deserialize_proof(const uint8_t* data, size_t len, bool uncompressed = false) {
    ByteReader r(data, len);
    r.uncompressed = uncompressed;
    Proof pr;
    uint64_t nr = r.u64();
    if (nr > 16) throw MarlinError(SWM_ERR_SERIALIZATION, "bad round count");
    for (uint64_t i = 0; i < nr; i++) {
        uint64_t nc = r.u64();
        if (nc > 64) throw MarlinError(SWM_ERR_SERIALIZATION, "bad commitment count");
        std::vector<Commitment> rnd;
        for (uint64_t j = 0; j < nc; j++) rnd.push_back(r.commitment());
        pr.commitments.push_back(rnd);
    }
    uint64_t ne = r.u64();
    if (ne > 1024) throw MarlinError(SWM_ERR_SERIALIZATION, "bad evaluation count");
    for (uint64_t i = 0; i < ne; i++) pr.evaluations.push_back(r.fr());
    uint64_t nm = r.u64();
    if (nm > 16) throw MarlinError(SWM_ERR_SERIALIZATION, "bad message count");
    // ark-marlin 0.3 provers only ever send ProverMsg::EmptyMessage (Option::None on the wire).  A FieldElements
    // message would be absorbed into the Fiat-Shamir transcript by the reference verifier; this verifier has no such
    // absorb, so a proof carrying one is refused instead of being verified against a different transcript.
    for (uint64_t i = 0; i < nm; i++)
        if (r.boolean()) throw MarlinError(SWM_ERR_SERIALIZATION, "non-empty prover message");
    uint64_t np = r.u64();
    if (np > 16) throw MarlinError(SWM_ERR_SERIALIZATION, "bad opening count");
    for (uint64_t i = 0; i < np; i++) {
        PcProof p;
        p.w = r.g1();
        p.has_random_v = r.boolean();
        if (p.has_random_v) p.random_v = r.fr();
        pr.pc_proof.push_back(p);
    }
    if (r.boolean()) {
        uint64_t k = r.u64();
        if (k > (1u << 20)) throw MarlinError(SWM_ERR_SERIALIZATION, "bad evals length");
        for (uint64_t j = 0; j < k; j++) r.fr();
    }
    if (r.pos != len) throw MarlinError(SWM_ERR_SERIALIZATION, "trailing bytes");
    return pr;
}

inline std::vector<uint8_t> serialize_verifying_key(const VerifyingKey& vk) {
    ByteWriter w;
    w.u64(vk.info.num_variables);
    w.u64(vk.info.num_constraints);
    w.u64(vk.info.num_non_zero);
    w.u64(vk.info.num_instance_variables);
    w.u64(vk.index_comms.size());
    for (auto& c : vk.index_comms) w.ser_commitment(c);
    w.ser_g1(vk.vk.g);
    w.ser_g1(vk.vk.gamma_g);
    w.ser_g2(vk.vk.h);
    w.ser_g2(vk.vk.beta_h);
    w.u8(1);
    w.u64(vk.vk.degree_bounds_and_shift_powers.size());
    for (auto& ds : vk.vk.degree_bounds_and_shift_powers) {
        w.u64(ds.first);
        w.ser_g1(ds.second);
    }
    w.u64(vk.vk.max_degree);
    w.u64(vk.vk.supported_degree);
    return w.b;
}

inline VerifyingKey read_verifying_key(ByteReader& r) {
    VerifyingKey vk;
    vk.info.num_variables = r.u64();
    vk.info.num_constraints = r.u64();
    vk.info.num_non_zero = r.u64();
    vk.info.num_instance_variables = r.u64();
    uint64_t nc = r.u64();
    if (nc > 64) throw MarlinError(SWM_ERR_SERIALIZATION, "bad index commitment count");
    for (uint64_t i = 0; i < nc; i++) vk.index_comms.push_back(r.commitment());
    vk.vk.g = r.g1();
    vk.vk.gamma_g = r.g1();
    vk.vk.h = r.g2();
    vk.vk.beta_h = r.g2();
    if (r.boolean()) {
        uint64_t k = r.u64();
        if (k > 64) throw MarlinError(SWM_ERR_SERIALIZATION, "bad degree-bound count");
        for (uint64_t i = 0; i < k; i++) {
            uint64_t d = r.u64();
            G1Affine p = r.g1();
            vk.vk.degree_bounds_and_shift_powers.push_back({d, p});
        }
    }
    vk.vk.max_degree = r.u64();
    vk.vk.supported_degree = r.u64();
    return vk;
}
inline VerifyingKey deserialize_verifying_key(const uint8_t* data, size_t len) {
    ByteReader r(data, len);
    VerifyingKey vk = read_verifying_key(r);
    if (r.pos != len) throw MarlinError(SWM_ERR_SERIALIZATION, "trailing bytes");
    return vk;
}

// ToBytes of IndexVerifierKey: index_info (3 x u64, without num_instance_variables) || index_comms
inline void tb_index_vk(ByteWriter& w, const VerifyingKey& vk) {
    w.u64(vk.info.num_variables);
    w.u64(vk.info.num_constraints);
    w.u64(vk.info.num_non_zero);
    for (auto& c : vk.index_comms) w.tb_commitment(c);
}

}  // namespace swm
