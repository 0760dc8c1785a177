// ahp.h — host side of the Marlin AHP: evaluation-domain scalars, the Fiat-Shamir transcript, the verifier's
// linear combinations and the pairing-based verifier.
// Mirrors ark-marlin 0.3.0 (fork branch use-constraint-system-directly, /root/reference/Cargo.toml:30; sources not
// vendored, restated from SURVEY.md A.6-A.8 [U]) for the instance fixed at /root/reference/src/marlin/mod.rs:12-14.
// verify() is the counterpart of verify_proof (src/marlin/mod.rs:79-86): milliseconds of host work, no GPU.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "blake2s.h"
#include "chacha.h"
#include "marlin_types.h"
#include "../switches.h"

namespace swm {

static const char* const kIndexerPolys[12] = {"a_row", "a_col", "a_val", "a_row_col", "b_row", "b_col",
                                              "b_val", "b_row_col", "c_row", "c_col", "c_val", "c_row_col"};
static const char* const kProverPolys[9] = {"w", "z_a", "z_b", "mask_poly", "t", "g_1", "h_1", "g_2", "h_2"};

// ------------------------------------------------------------------------------------------------ domains (scalars only)
struct HDomain {
    uint64_t size;
    unsigned log;
    Fr gen, gen_inv, size_inv, size_fr;
    explicit HDomain(uint64_t n) {
        // Fr has two-adicity 47: ark-poly's Radix2EvaluationDomain::new returns None beyond that and ark-marlin turns it into
        // PolynomialDegreeTooLarge.  Without the bound a size above 2^63 (a mutated verifying key's num_constraints: found by the
        // r06 sanitizer run) overflows the doubling below into an endless loop.
        if (n > (1ull << 47)) throw MarlinError(SWM_ERR_INVALID_ARG, "PolynomialDegreeTooLarge: no radix-2 domain of that size in Fr");
        size = 1;
        log = 0;
        while (size < n) {
            size <<= 1;
            log++;
        }
        static const uint32_t root[8] = SWM_FR_ROOT47_MONT;
        gen = fp_from_limbs<Fr>(root);
        for (unsigned i = log; i < 47; i++) gen = fp_sqr(gen);
        gen_inv = fp_inv(gen);
        size_fr = fp_from_u64<Fr>(size);
        size_inv = fp_inv(size_fr);
    }
    Fr element(uint64_t i) const { return fr_pow_u64(gen, i); }
    Fr vanishing(const Fr& tau) const { return fp_sub(fr_pow_u64(tau, size), fp_one<Fr>()); }
    // ark-poly reindex_by_subdomain
    uint64_t reindex_by_subdomain(const HDomain& other, uint64_t index) const {
        uint64_t period = size / other.size;
        if (index < other.size) return index * period;
        uint64_t i = index - other.size, x = period - 1;
        return i + (i / x) + 1;
    }
    Fr eval_unnormalized_bivariate_lagrange_poly(const Fr& x, const Fr& y) const {
        if (!fp_eq(x, y)) return fp_mul(fp_sub(vanishing(x), vanishing(y)), fp_inv(fp_sub(x, y)));
        return fp_mul(size_fr, fr_pow_u64(x, size - 1));
    }
    // small host transforms (public-input interpolation; natural order)
    void dft(std::vector<Fr>& a, bool inverse) const {
        a.resize(size, fp_zero<Fr>());
        uint64_t n = size;
        for (uint64_t i = 1, j = 0; i < n; i++) {
            uint64_t bit = n >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) std::swap(a[i], a[j]);
        }
        Fr w = inverse ? gen_inv : gen;
        for (uint64_t len = 2; len <= n; len <<= 1) {
            Fr wl = fr_pow_u64(w, n / len);
            for (uint64_t s = 0; s < n; s += len) {
                Fr t = fp_one<Fr>();
                for (uint64_t k = s; k < s + len / 2; k++) {
                    Fr u = a[k], v = fp_mul(a[k + len / 2], t);
                    a[k] = fp_add(u, v);
                    a[k + len / 2] = fp_sub(u, v);
                    t = fp_mul(t, wl);
                }
            }
        }
        if (inverse)
            for (auto& x : a) x = fp_mul(x, size_inv);
    }
};

inline Fr host_poly_eval(const std::vector<Fr>& p, const Fr& x) {
    Fr acc = fp_zero<Fr>();
    for (size_t i = p.size(); i-- > 0;) acc = fp_add(fp_mul(acc, x), p[i]);
    return acc;
}

// ------------------------------------------------------------------------------------------------ Fiat-Shamir
// SimpleHashFiatShamirRng<Blake2s, ChaChaRng>: seed = Blake2s(bytes [|| old seed]); rng = ChaCha20::from_seed(seed)
struct FiatShamirRng {
    uint8_t seed[32];
    ChaChaRng r;
    void initialize(const std::vector<uint8_t>& bytes) {
        Blake2s::digest(bytes.data(), bytes.size(), seed);
        r.seed(seed, 20);
    }
    void absorb(const std::vector<uint8_t>& bytes) {
        Blake2s b;
        b.update(bytes.data(), bytes.size());
        b.update(seed, 32);
        b.finalize(seed);
        r.seed(seed, 20);
    }
    Fr rand_fr() { return r.rand_fr(); }
    Fr sample_outside(const HDomain& d) {
        Fr t = rand_fr();
        while (fp_is_zero(d.vanishing(t))) t = rand_fr();
        return t;
    }
    Fr challenge_u128() {
        uint64_t v[2];
        r.gen_u128(v);
        return fr_from_u128(v);
    }
};

inline void fs_init(FiatShamirRng& fs, const VerifyingKey& vk, const std::vector<Fr>& public_input) {
    ByteWriter w;
    w.raw("MARLIN-2019", 11);
    tb_index_vk(w, vk);
    for (auto& x : public_input) w.fr(x);
    fs.initialize(w.b);
}
inline void fs_absorb_commitments(FiatShamirRng& fs, const std::vector<Commitment>& comms) {
    ByteWriter w;
    for (auto& c : comms) w.tb_commitment(c);
    fs.absorb(w.b);
}
inline void fs_absorb_evals(FiatShamirRng& fs, const std::vector<Fr>& evals) {
    ByteWriter w;
    for (auto& e : evals) w.fr(e);
    fs.absorb(w.b);
}

struct VerifierState {
    Fr alpha, eta_a, eta_b, eta_c, beta, gamma;
};

// ------------------------------------------------------------------------------------------------ linear combinations
// term label "" = LCTerm::One
typedef std::vector<std::pair<Fr, std::string>> LcTerms;
typedef std::map<std::string, LcTerms> LcSet;  // iteration order = label order (as lc_s.sort_by label)

struct QueryEntry {
    const char* label;
    const char* point;  // "beta" | "gamma"
};
static const QueryEntry kQuerySet[9] = {{"g_1", "beta"}, {"z_b", "beta"}, {"t", "beta"}, {"outer_sumcheck", "beta"},
                                        {"g_2", "gamma"}, {"a_denom", "gamma"}, {"b_denom", "gamma"},
                                        {"c_denom", "gamma"}, {"inner_sumcheck", "gamma"}};
inline bool lc_has_zero_eval(const std::string& l) { return l == "inner_sumcheck" || l == "outer_sumcheck"; }

// EvaluationsProvider: eval(label, terms, point)
template <class Provider>
LcSet construct_linear_combinations(const IndexInfo& info, const std::vector<Fr>& public_input, Provider eval,
                                    const VerifierState& st) {
    HDomain dh(info.num_constraints), dk(info.num_non_zero), dx(public_input.size() + 1);
    std::vector<Fr> x_poly;
    x_poly.push_back(fp_one<Fr>());
    for (auto& v : public_input) x_poly.push_back(v);
    dx.dft(x_poly, true);
    const Fr one = fp_one<Fr>();
    LcSet lcs;
    lcs["z_b"] = {{one, "z_b"}};
    lcs["g_1"] = {{one, "g_1"}};
    lcs["t"] = {{one, "t"}};
    Fr r_alpha_at_beta = dh.eval_unnormalized_bivariate_lagrange_poly(st.alpha, st.beta);
    Fr v_H_at_alpha = dh.vanishing(st.alpha), v_H_at_beta = dh.vanishing(st.beta), v_X_at_beta = dx.vanishing(st.beta);
    Fr z_b_at_beta = eval("z_b", lcs["z_b"], st.beta);
    Fr t_at_beta = eval("t", lcs["t"], st.beta);
    Fr g_1_at_beta = eval("g_1", lcs["g_1"], st.beta);
    Fr x_at_beta = host_poly_eval(x_poly, st.beta);
    lcs["outer_sumcheck"] = {
        {one, "mask_poly"},
        {fp_mul(r_alpha_at_beta, fp_add(st.eta_a, fp_mul(st.eta_c, z_b_at_beta))), "z_a"},
        {fp_mul(fp_mul(r_alpha_at_beta, st.eta_b), z_b_at_beta), ""},
        {fp_neg(fp_mul(t_at_beta, v_X_at_beta)), "w"},
        {fp_neg(fp_mul(t_at_beta, x_at_beta)), ""},
        {fp_neg(v_H_at_beta), "h_1"},
        {fp_neg(fp_mul(st.beta, g_1_at_beta)), ""},
    };
    Fr beta_alpha = fp_mul(st.beta, st.alpha);
    lcs["g_2"] = {{one, "g_2"}};
    const char* ms[3] = {"a", "b", "c"};
    for (auto m : ms) {
        std::string s(m);
        lcs[s + "_denom"] = {{beta_alpha, ""}, {fp_neg(st.alpha), s + "_row"}, {fp_neg(st.beta), s + "_col"},
                             {one, s + "_row_col"}};
    }
    Fr a_d = eval("a_denom", lcs["a_denom"], st.gamma), b_d = eval("b_denom", lcs["b_denom"], st.gamma),
       c_d = eval("c_denom", lcs["c_denom"], st.gamma);
    Fr g_2_at_gamma = eval("g_2", lcs["g_2"], st.gamma);
    Fr v_K_at_gamma = dk.vanishing(st.gamma);
    Fr scale = fp_mul(v_H_at_alpha, v_H_at_beta);
    LcTerms inner = {
        {fp_mul(fp_mul(fp_mul(st.eta_a, b_d), c_d), scale), "a_val"},
        {fp_mul(fp_mul(fp_mul(st.eta_b, a_d), c_d), scale), "b_val"},
        {fp_mul(fp_mul(fp_mul(st.eta_c, b_d), a_d), scale), "c_val"},
    };
    Fr b_at_gamma = fp_mul(fp_mul(a_d, b_d), c_d);
    Fr b_expr = fp_mul(b_at_gamma, fp_add(fp_mul(st.gamma, g_2_at_gamma), fp_mul(t_at_beta, dk.size_inv)));
    inner.push_back({fp_neg(b_expr), ""});
    inner.push_back({fp_neg(v_K_at_gamma), "h_2"});
    lcs["inner_sumcheck"] = inner;
    return lcs;
}

// ------------------------------------------------------------------------------------------------ verifier
// Marlin::verify + MarlinKZG10::check_combinations + KZG10::batch_check.  `rng` is the caller's generator
// (the batch_check randomiser is drawn from it, as in arkworks).
inline bool verify(const VerifyingKey& vk, std::vector<Fr> public_input, const Proof& proof, ChaChaRng& rng) {
    const bool vtrace = env_flag("SWM_TRACE");
    auto vnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double vt0 = vnow();
    auto vmark = [&](const char* what) {
        if (vtrace) {
            double t = vnow();
            fprintf(stderr, "[swm trace] verify: %-28s %7.3f ms\n", what, t - vt0);
            vt0 = t;
        }
    };
    HDomain dx(public_input.size() + 1);
    size_t padded = std::max<size_t>(public_input.size(), dx.size - 1);
    public_input.resize(padded, fp_zero<Fr>());
    if (proof.commitments.size() != 3 || proof.commitments[0].size() != 4 || proof.commitments[1].size() != 3 ||
        proof.commitments[2].size() != 2 || vk.index_comms.size() != 12 || proof.pc_proof.size() != 2)
        return false;
    HDomain dh(vk.info.num_constraints), dk(vk.info.num_non_zero);
    FiatShamirRng fs;
    fs_init(fs, vk, public_input);
    VerifierState st;
    fs_absorb_commitments(fs, proof.commitments[0]);
    st.alpha = fs.sample_outside(dh);
    st.eta_a = fs.rand_fr();
    st.eta_b = fs.rand_fr();
    st.eta_c = fs.rand_fr();
    fs_absorb_commitments(fs, proof.commitments[1]);
    st.beta = fs.sample_outside(dh);
    fs_absorb_commitments(fs, proof.commitments[2]);
    st.gamma = fs.rand_fr();
    // labelled commitments with their degree bounds
    struct LComm {
        Commitment c;
        bool has_bound;
        uint64_t bound;
    };
    std::map<std::string, LComm> commitments;
    for (int i = 0; i < 12; i++) commitments[kIndexerPolys[i]] = {vk.index_comms[i], false, 0};
    {
        int k = 0;
        for (auto& rnd : proof.commitments)
            for (auto& c : rnd) {
                std::string l = kProverPolys[k++];
                LComm lc{c, false, 0};
                if (l == "g_1") lc = {c, true, dh.size - 2};
                if (l == "g_2") lc = {c, true, dk.size - 2};
                if (lc.has_bound != c.has_shifted) return false;
                commitments[l] = lc;
            }
    }
    fs_absorb_evals(fs, proof.evaluations);
    Fr xi = fs.challenge_u128();
    // evaluations keyed by LC label
    std::map<std::string, Fr> evaluations;
    std::map<std::string, std::string> point_of;
    std::vector<std::string> eval_labels;
    for (auto& q : kQuerySet) {
        point_of[q.label] = q.point;
        if (lc_has_zero_eval(q.label)) evaluations[q.label] = fp_zero<Fr>();
        else eval_labels.push_back(q.label);
    }
    std::sort(eval_labels.begin(), eval_labels.end());
    if (eval_labels.size() != proof.evaluations.size()) return false;
    for (size_t i = 0; i < eval_labels.size(); i++) evaluations[eval_labels[i]] = proof.evaluations[i];
    auto provider = [&](const std::string& label, const LcTerms&, const Fr&) { return evaluations.at(label); };
    vmark("transcript");
    LcSet lcs = construct_linear_combinations(vk.info, public_input, provider, st);
    vmark("linear combinations");
    // check_combinations: combine commitments, fold constant terms into the claimed evaluations
    std::map<std::string, LComm> lc_comms;
    for (auto& kv : lcs) {
        const std::string& label = kv.first;
        std::vector<std::pair<G1Affine, Fr>> comm_terms, shifted_terms;  // summed by one Straus chain each (g1_msm_host)
        bool has_bound = false;
        uint64_t bound = 0;
        for (auto& term : kv.second) {
            if (term.second.empty()) {
                evaluations[label] = fp_sub(evaluations[label], term.first);
                continue;
            }
            const LComm& cur = commitments.at(term.second);
            if (kv.second.size() == 1 && cur.has_bound) {
                if (!fp_is_one(term.first)) return false;
                has_bound = true;
                bound = cur.bound;
            } else if (cur.has_bound) {
                return false;  // EquationHasDegreeBounds
            }
            comm_terms.push_back({cur.c.comm, term.first});
            if (cur.c.has_shifted) shifted_terms.push_back({cur.c.shifted, term.first});
        }
        LComm out;
        out.c.comm = g1_to_affine(g1_msm_host(comm_terms));
        out.c.has_shifted = has_bound;
        if (has_bound) out.c.shifted = g1_to_affine(g1_msm_host(shifted_terms));
        out.has_bound = has_bound;
        out.bound = bound;
        lc_comms[label] = out;
    }
    vmark("combined commitments");
    // batch_check: per query point (beta, then gamma), labels in sorted order, challenges xi^0, xi^1, ...
    struct Combined {
        G1Affine c;
        Fr z, v;
    };
    std::vector<Combined> combined;
    const char* points[2] = {"beta", "gamma"};
    for (auto pl : points) {
        std::vector<std::string> labels;
        for (auto& q : kQuerySet)
            if (std::string(q.point) == pl) labels.push_back(q.label);
        std::sort(labels.begin(), labels.end());
        std::vector<std::pair<G1Affine, Fr>> cc_terms;
        Fr cv = fp_zero<Fr>();
        Fr ch = fp_one<Fr>();  // xi^ctr
        for (auto& l : labels) {
            const LComm& lc = lc_comms.at(l);
            const Fr& v = evaluations.at(l);
            cc_terms.push_back({lc.c.comm, ch});
            cv = fp_add(cv, fp_mul(v, ch));
            ch = fp_mul(ch, xi);
            if (lc.has_bound) {
                const G1Affine* sp = nullptr;
                for (auto& ds : vk.vk.degree_bounds_and_shift_powers)
                    if (ds.first == lc.bound) sp = &ds.second;
                if (!sp) return false;
                // (shifted - v * shift_power) * ch as two terms of the same sum
                cc_terms.push_back({lc.c.shifted, ch});
                cc_terms.push_back({*sp, fp_neg(fp_mul(v, ch))});
                ch = fp_mul(ch, xi);
            }
        }
        combined.push_back({g1_to_affine(g1_msm_host(cc_terms)), std::string(pl) == "beta" ? st.beta : st.gamma, cv});
    }
    vmark("batched openings");
    // total_c = sum_i randomizer_i (z_i w_i + c_i) - g_mult g - gamma_g_mult gamma_g,  total_w = sum_i randomizer_i w_i
    std::vector<std::pair<G1Affine, Fr>> c_terms, w_terms;
    Fr randomizer = fp_one<Fr>(), g_mult = fp_zero<Fr>(), gamma_g_mult = fp_zero<Fr>();
    for (size_t i = 0; i < combined.size(); i++) {
        const PcProof& pp = proof.pc_proof[i];
        g_mult = fp_add(g_mult, fp_mul(randomizer, combined[i].v));
        if (pp.has_random_v) gamma_g_mult = fp_add(gamma_g_mult, fp_mul(randomizer, pp.random_v));
        c_terms.push_back({pp.w, fp_mul(randomizer, combined[i].z)});
        c_terms.push_back({combined[i].c, randomizer});
        w_terms.push_back({pp.w, randomizer});
        uint64_t rv[2];
        rng.gen_u128(rv);
        randomizer = fr_from_u128(rv);
    }
    c_terms.push_back({vk.vk.g, fp_neg(g_mult)});
    c_terms.push_back({vk.vk.gamma_g, fp_neg(gamma_g_mult)});
    G1Affine tw = g1_to_affine(g1_msm_host(w_terms)), tc = g1_to_affine(g1_msm_host(c_terms));
    if (!g1_is_inf(tw)) tw = g1_neg(tw);
    vmark("pairing inputs");
    const bool ok = product_of_pairings_is_one({{tw, vk.vk.beta_h}, {tc, vk.vk.h}});
    vmark("two pairings");
    return ok;
}

}  // namespace swm
