/* oracle.h — CPU restatement (plain C) of the arithmetic on simpleworks' Marlin prove() path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (simpleworks_amd/, libswmarlin.so) never links or calls it.
 *
 * The reference (/root/reference, lambdaclass/simpleworks) holds none of this arithmetic itself:
 * src/marlin/mod.rs:52,75,85,92 delegate to arkworks 0.3 crates that are not vendored (Cargo.toml:15-30,
 * no Cargo.lock).  Each function below restates the published arkworks 0.3.0 algorithm recorded in
 * SURVEY.md Appendix A and cites the reference call site that reaches it.
 *
 * Parity status: UNPINNED against arkworks itself (no Rust toolchain here, no known-answer vectors in
 * the reference's tests — SURVEY.md §8c).  Pinned instead against the independent Python big-int
 * model (oracle/pyref) through the committed fixtures in tests/golden/, and against mathematical
 * invariants (canonical outputs: an MSM, an NTT, a mat-vec have exactly one correct value).
 *
 * Data formats (ark-ff 0.3 BigInteger256/384, little-endian u64 limbs):
 *   Fr element  : 4 x u64, Montgomery form (R = 2^256) unless a parameter says "standard form"
 *   Fq element  : 6 x u64, Montgomery form (R = 2^384)
 *   G1 affine   : x,y = 12 x u64 Montgomery; the point at infinity is encoded as x = y = 0
 *   G1 Jacobian : X,Y,Z = 18 x u64 Montgomery; infinity has Z = 0
 */
#ifndef SWM_ORACLE_H
#define SWM_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- field helpers (ark-ff 0.3 Fp256/Fp384 Montgomery) */
void oracle_fr_to_mont(const uint64_t *std4, uint64_t *mont4, size_t n);
void oracle_fr_from_mont(const uint64_t *mont4, uint64_t *std4, size_t n);
void oracle_fq_to_mont(const uint64_t *std6, uint64_t *mont6, size_t n);
void oracle_fq_from_mont(const uint64_t *mont6, uint64_t *std6, size_t n);
void oracle_fr_mul(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n); /* elementwise, Montgomery */
void oracle_fr_add(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void oracle_fr_sub(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void oracle_fq_mul(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
void oracle_fr_inv(const uint64_t *a, uint64_t *out, size_t n); /* per-element inverse, 0 -> 0 */
/* ark_ff::batch_inversion: Montgomery trick, zeros stay zero (K4; reached from src/marlin/mod.rs:75) */
void oracle_batch_inverse_fr(uint64_t *v, size_t n);

/* ---- G1 (ark-ec 0.3 short Weierstrass Jacobian, a = 0, b = 1) */
void oracle_g1_add_mixed(const uint64_t *jac18, const uint64_t *aff12, uint64_t *out18);
void oracle_g1_double(const uint64_t *jac18, uint64_t *out18);
void oracle_g1_add(const uint64_t *a18, const uint64_t *b18, uint64_t *out18);
/* Jacobian -> affine (Montgomery x,y; infinity -> zeros). Returns 1 if the point is infinity. */
int oracle_g1_to_affine(const uint64_t *jac18, uint64_t *aff12);
int oracle_g1_is_on_curve(const uint64_t *aff12);
/* out[i] = [scalars[i]] * base, affine (fixed-base windowed; used to build SRS-shaped test bases
 * P_i = [tau^i]G exactly as KZG10::setup produces them, src/marlin/mod.rs:45-55).  scalars standard form. */
void oracle_g1_fixed_base_mul(const uint64_t *base12, const uint64_t *scalars4, size_t n, uint64_t *out12,
                              int threads);

/* ---- K1: ark_ec::msm::VariableBaseMSM::multi_scalar_mul (0.3.0), SURVEY A.2.
 * Reached from src/marlin/mod.rs:75 (prove) and :92 (index) through KZG10::commit/open.
 * bases: n affine points; scalars: n x 4 limbs STANDARD form; out: Jacobian.
 * threads > 1 runs windows in parallel exactly where arkworks' `parallel` feature uses rayon. */
void oracle_msm_g1(const uint64_t *bases12, const uint64_t *scalars4, size_t n, uint64_t *out18, int threads);
/* arkworks' window-size rule (c = 3 if n < 32 else ceil_log2(n)*69/100 + 2) */
unsigned oracle_msm_window(size_t n);

/* ---- K2: ark_poly::Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place (0.3.0), SURVEY A.3.
 * data: 2^log_n Montgomery Fr elements, natural order in and out. */
void oracle_ntt_fr(uint64_t *data4, unsigned log_n, int inverse, int coset, int threads);

/* ---- K3: row-sparse M*z as in ark-marlin prover_init; CSR, Montgomery values. */
void oracle_spmv_fr(const uint32_t *rowptr, const uint32_t *col, const uint64_t *val4, const uint64_t *z4,
                    uint64_t *out4, size_t rows);
/* the same with the rows spread over `threads` OpenMP threads (ark-marlin's cfg_iter! under `parallel`) */
void oracle_spmv_fr_mt(const uint32_t *rowptr, const uint32_t *col, const uint64_t *val4, const uint64_t *z4,
                       uint64_t *out4, size_t rows, int threads);

/* ---- Pedersen CRH (+ TECompressor) on ed-on-BLS12-377 and the Merkle tree over it: ark-crypto-primitives 0.3 [U], as reached
 * from src/hash/mod.rs:23-28 and src/merkle_tree/simple_merkle_tree.rs:47-49 (windows: src/merkle_tree/common.rs:11-30).
 * gens_xy_std: [nw][ws] affine generators, 8 limbs each (x, y, standard form); inputs: count x len bytes; digests /
 * nodes: 32 little-endian bytes each; nodes = n leaf digests | n / 2 | ... | root. */
void oracle_pedersen_hash(const uint64_t *gens_xy_std, size_t nw, size_t ws, const uint8_t *inputs, size_t len, size_t count,
                          uint8_t *digests, int threads);
void oracle_merkle_tree(const uint64_t *leaf_gens, size_t nw_leaf, const uint64_t *inner_gens, size_t nw_inner, size_t ws,
                        const uint8_t *leaves, size_t leaf_len, size_t n, uint8_t *nodes, int threads);

/* number of OpenMP threads the library can use on this host */
int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
