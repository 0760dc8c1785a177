/* swmarlin.h — C ABI of libswmarlin.so: the MI355X-native (gfx950 HIP) replacement for the arithmetic that
 * simpleworks' Marlin wrapper delegates to arkworks.
 *
 * Drop-in boundary.  The reference's Rust surface
 *     /root/reference/src/marlin/mod.rs:33-94            generate_rand, generate_universal_srs, generate_proof,
 *                                                        verify_proof, generate_proving_and_verifying_keys
 *     /root/reference/src/marlin/serialization.rs:5-45   (de)serialisers
 * stays source compatible; a thin Rust shim copies arkworks values into packed limb buffers and
 * binds exactly the entry points declared here.  No torch / C++ types cross this boundary: plain pointers, sizes
 * and opaque handles only.
 *
 * Conventions
 *   - every function returns an int status: SWM_OK (0) or a negative SWM_ERR_* code; swm_strerror() names it and
 *     swm_last_error(ctx) carries detail.  Nothing throws or aborts across the ABI (src/lib.rs:28 forbids panics).
 *   - Fr element  = 4 x uint64 little-endian limbs (ark-ff BigInteger256); Montgomery form unless stated.
 *   - Fq element  = 6 x uint64 limbs, Montgomery form.  G1 affine = x,y (12 limbs), infinity encoded as x = y = 0.
 *   - G1 Jacobian = X,Y,Z (18 limbs, Montgomery), infinity has Z = 0 — what ark-ec's G1Projective holds.
 *   - the caller owns every host buffer for the duration of a call; the library never retains host pointers.
 *   - functions suffixed _dev take DEVICE pointers (hipMalloc'd or a torch tensor's data_ptr) and enqueue on the
 *     context's stream without a host copy of the bulk data; the plain forms take HOST pointers and stage.
 *   - a context is bound to one GPU and one HIP stream; use one context per host thread.
 *   - there is NO CPU fallback: every compute entry point fails with SWM_ERR_NO_DEVICE when no gfx950 GPU is usable.
 */
#ifndef SWMARLIN_H
#define SWMARLIN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SWM_OK 0
#define SWM_ERR_INVALID_ARG (-1)
#define SWM_ERR_NO_DEVICE (-2)
#define SWM_ERR_HIP (-3)
#define SWM_ERR_OOM (-4)
#define SWM_ERR_UNSATISFIED (-5)      /* witness does not satisfy the constraint system (prove-time failure) */
#define SWM_ERR_INDEX_TOO_LARGE (-6)  /* SRS too small for the circuit (ark-marlin Error::IndexTooLarge) */
#define SWM_ERR_SERIALIZATION (-7)
#define SWM_ERR_MISMATCH (-8)         /* instance does not match index / key */
#define SWM_ERR_INTERNAL (-9)

typedef struct swm_ctx swm_ctx;
typedef struct swm_bases swm_bases;

/* ---------------------------------------------------------------------------------------------- context */
int swm_version(void);
const char *swm_strerror(int code);
/* Creates a context on HIP device `device` with its own stream. */
int swm_init(int device, swm_ctx **out);
void swm_destroy(swm_ctx *ctx);
/* detail of the last failure on this context; ctx == NULL: of the last context-free call (verify, codecs) on the
 * calling thread */
const char *swm_last_error(swm_ctx *ctx);
/* Enqueue on an externally owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL restores the own stream. */
int swm_set_stream(swm_ctx *ctx, void *hip_stream);
int swm_synchronize(swm_ctx *ctx);
/* device memory helpers so that a caller without a HIP binding can keep operands resident in HBM */
int swm_malloc(swm_ctx *ctx, size_t bytes, void **dptr);
int swm_free(swm_ctx *ctx, void *dptr);
int swm_memcpy_h2d(swm_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int swm_memcpy_d2h(swm_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* ---------------------------------------------------------------------------------------------- K1: G1 MSM
 * Replaces ark_ec::msm::VariableBaseMSM::multi_scalar_mul(bases, scalars) (ark-ec 0.3.0), reached from
 * src/marlin/mod.rs:75 (prove) and :92 (index) through ark_poly_commit::kzg10::KZG10::commit / open.
 * The result is the same group element arkworks computes (compare after affine normalisation). */
/* Upload n affine bases (n x 12 limbs, Montgomery) once; they stay resident in HBM (SRS powers [tau^i]G).
 * x = y = 0 encodes the point at infinity. */
int swm_srs_upload(swm_ctx *ctx, const uint64_t *xy, size_t n, swm_bases **out);
int swm_srs_free(swm_ctx *ctx, swm_bases *bases);
size_t swm_srs_len(const swm_bases *bases);
/* out_jac = sum_i scalars[i] * bases[offset + i].  scalars: n x 4 limbs, STANDARD form (what arkworks passes:
 * p.coeffs.map(|s| s.into_repr())), host memory.  Every scalar must be a canonical field element (< r), as arkworks'
 * BigInteger256 scalars are: a scalar >= r makes the call fail with SWM_ERR_INVALID_ARG (checked on the device, no
 * result is written).  Bases that are the point at infinity (x = y = 0) contribute nothing, as in arkworks. */
int swm_msm_g1(swm_ctx *ctx, const swm_bases *bases, size_t offset, const uint64_t *scalars, size_t n,
               uint64_t out_jac[18]);
/* same with the scalars already in HBM; scalars_montgomery != 0 means they are Montgomery-form Fr (polynomial
 * coefficients as the NTT leaves them) and are converted on the fly. */
int swm_msm_g1_dev(swm_ctx *ctx, const swm_bases *bases, size_t offset, const void *d_scalars, size_t n,
                   int scalars_montgomery, uint64_t out_jac[18]);
/* Jacobian -> affine on the host (x = y = 0 for infinity); returns 1 in *is_inf for the identity. */
int swm_g1_normalize(const uint64_t jac[18], uint64_t out_xy[12], int *is_inf);
/* out = a + b on the host (Jacobian in/out): folds the per-GPU partial sums of a point-range-sharded MSM after the
 * all-gather (EC addition is not an RCCL reduction op). */
int swm_g1_add_jac(const uint64_t a[18], const uint64_t b[18], uint64_t out[18]);

/* ---------------------------------------------------------------------------------------------- K2: Fr NTT
 * Replaces ark_poly::Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place (ark-poly 0.3.0):
 * natural order in, natural order out, size 2^log_n, root = TWO_ADIC_ROOT^(2^(47-log_n)); inverse scales by 1/n;
 * coset pre-scales coefficient i by 22^i (forward) / post-scales by 22^-i (inverse). */
int swm_ntt_fr(swm_ctx *ctx, uint64_t *data, unsigned log_n, int inverse, int coset);
int swm_ntt_fr_dev(swm_ctx *ctx, void *d_data, unsigned log_n, int inverse, int coset);
/* ONE transform over the G ranks of the context's sharding (swm_set_msm_sharding / swm_rccl_init; G a power of two <= 16,
 * 2^log_n >= G^2): the four-step split with a single all-to-all (SURVEY.md §8e "NTT partitioning (ii)").  `d_local` holds
 * this rank's n / G elements, in place.  Layouts (m = n / G, blk = m / G):
 *     CYCLIC   local[j] = v[rank + G j]                       BLOCKS   local[k1 blk + t] = v[m k1 + rank blk + t]
 * blocks_in = 0: CYCLIC in -> BLOCKS out;  blocks_in = 1: BLOCKS in -> CYCLIC out.  inverse as swm_ntt_fr_dev (scales by
 * 1 / n).  Every rank calls with the same arguments; the exchange is ncclSend / ncclRecv (grouped) on the context's stream
 * when a communicator is set, the all-gather callback otherwise.  Evaluations in BLOCKS and coefficients in CYCLIC layout is
 * what the sharded prover keeps: its commitment MSMs take CYCLIC coefficients where they are. */
int swm_ntt_fr_sharded_dev(swm_ctx *ctx, void *d_local, unsigned log_n, int inverse, int blocks_in);

/* ---------------------------------------------------------------------------------------------- K3: R1CS mat-vec
 * Replaces the row-wise inner products of ark-marlin's prover_init (z_A = A z, z_B = B z) and the evaluation
 * inside ConstraintSystem::is_satisfied (src/merkle_tree/simple_merkle_tree.rs:197-199).
 * CSR: rowptr[rows+1], col[nnz] (uint32), val[nnz] (Fr Montgomery); z: dense Fr vector; out: rows Fr. */
int swm_spmv_fr(swm_ctx *ctx, const uint32_t *rowptr, const uint32_t *col, const uint64_t *val, const uint64_t *z,
                size_t z_len, uint64_t *out, size_t rows, size_t nnz);
int swm_spmv_fr_dev(swm_ctx *ctx, const void *d_rowptr, const void *d_col, const void *d_val, const void *d_z,
                    void *d_out, size_t rows);

/* ---------------------------------------------------------------------------------------------- K4: support kernels
 * ark_ff::batch_inversion (zeros stay zero) and pointwise products, on Montgomery Fr vectors. */
int swm_batch_inverse_fr(swm_ctx *ctx, uint64_t *data, size_t n);
int swm_batch_inverse_fr_dev(swm_ctx *ctx, void *d_data, size_t n);
int swm_vec_mul_fr(swm_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
int swm_vec_mul_fr_dev(swm_ctx *ctx, const void *d_a, const void *d_b, void *d_out, size_t n);

/* ---------------------------------------------------------------------------------------------- Marlin surface
 * The functions of /root/reference/src/marlin/mod.rs:33-94 and src/marlin/serialization.rs:5-45, one entry point
 * each.  Keys are opaque handles (the proving key owns device-resident matrices, index polynomials and SRS powers);
 * proofs and verifying keys cross the boundary in the ark-serialize wire format the reference's serialisers emit. */
typedef struct swm_rng swm_rng; /* rand::rngs::StdRng as produced by generate_rand(): ChaCha12, fixed test seed */
typedef struct swm_srs swm_srs; /* Box<UniversalSRS> */
typedef struct swm_pk swm_pk;   /* ProvingKey  = IndexProverKey */
typedef struct swm_vk swm_vk;   /* VerifyingKey = IndexVerifierKey */

/* generate_rand() (src/marlin/mod.rs:33-35): ark_std::test_rng() */
int swm_rng_test_new(swm_rng **out);
/* StdRng::from_seed(seed) for callers that want their own randomness */
int swm_rng_from_seed(const uint8_t seed[32], swm_rng **out);
/* The CALLER's generator behind the same handle: the reference's functions take `&mut StdRng`
 * (src/marlin/mod.rs:49,73,83); a shim that wants to keep that signature AND the draw stream wraps its rng in a
 * fill_bytes trampoline.  Every draw the library makes is then a call fill_bytes(user, dest, 4 | 8 | 32 n) and consumes
 * the caller's stream word for word as arkworks would (rand_core BlockRng: next_u32 = 4 bytes, next_u64 = 8 bytes,
 * both little-endian, consecutive words).  The callback is invoked on the calling thread, only from inside
 * swm_generate_universal_srs / swm_generate_proof / swm_verify_proof / swm_rng_next_u64 / swm_rng_rand_fr, and must not
 * call back into the library.  `user` must stay valid until swm_rng_free.  Bulk draws (the 3|H| mask coefficients of a
 * proof) travel through the callback too (~170 MB at |H| = 2^20, requested while the GPU works on the rest of round 1):
 * as fast as the caller's generator is.  A ChaCha generator is better handed over by state (swm_rng_from_chacha). */
typedef void (*swm_fill_bytes_fn)(void *user, uint8_t *dest, size_t len);
int swm_rng_from_callback(swm_fill_bytes_fn fill_bytes, void *user, swm_rng **out);
/* The caller's generator by STATE instead of by callback, for callers whose generator is a ChaCha stream — rand 0.8's
 * StdRng (= rand_chacha::ChaCha12Rng, the type of every `rng` parameter in src/marlin/mod.rs:49,73,83) is: `key` =
 * get_seed(), `word_pos` = get_word_pos() (32-bit words of keystream consumed so far), `rounds` = 8, 12 or 20, stream
 * id 0.  The library then draws exactly the words the caller's generator would have produced — bulk draws on the GPU,
 * nothing through a callback — and swm_rng_word_pos returns where the stream stands afterwards, which the caller writes
 * back with set_word_pos: caller-visible behaviour identical to swm_rng_from_callback at the cost of the built-in rng. */
int swm_rng_from_chacha(const uint8_t key[32], uint64_t word_pos, int rounds, swm_rng **out);
int swm_rng_word_pos(const swm_rng *rng, uint64_t *word_pos); /* SWM_ERR_INVALID_ARG for a callback generator */
/* RngCore::fill_bytes on the handle (rand_core BlockRng: whole 32-bit words are consumed, little-endian; a tail of
 * 1-3 bytes takes the low bytes of one more word) */
int swm_rng_fill_bytes(swm_rng *rng, uint8_t *dest, size_t len);
/* The same as a swm_fill_bytes_fn (user = a swm_rng* that is NOT the one it is installed in): lets a harness put one
 * library generator behind the callback of another handle — a stand-in for "the caller's StdRng" whose stream is known
 * (tests, `bench.py --rng callback`).  Pure host code, no context: the one callback that may enter the library. */
void swm_rng_fill_bytes_cb(void *user, uint8_t *dest, size_t len);
void swm_rng_free(swm_rng *rng);
int swm_rng_next_u64(swm_rng *rng, uint64_t *out);
int swm_rng_rand_fr(swm_rng *rng, uint64_t out_mont[4]); /* ark_ff UniformRand for Fr (Montgomery limbs) */

/* A synthesised constraint system (what ConstraintSystemRef<Fr> holds after generate_constraints): instance
 * assignment (instance[0] must be one), witness assignment, and A, B, C as CSR over columns
 * [instance..., witness...] with Montgomery coefficients.  Padding (public input to a power of two, square
 * matrices) is applied inside index/prove exactly as ark-marlin does. */
typedef struct swm_r1cs {
    size_t num_instance, num_witness, num_constraints;
    const uint64_t *instance; /* num_instance x 4 */
    const uint64_t *witness;  /* num_witness x 4 */
    const uint32_t *a_rowptr, *a_col; const uint64_t *a_val;
    const uint32_t *b_rowptr, *b_col; const uint64_t *b_val;
    const uint32_t *c_rowptr, *c_col; const uint64_t *c_val;
} swm_r1cs;

/* generate_universal_srs (src/marlin/mod.rs:45-55): MarlinInst::universal_setup(nc, nv, nnz, rng) */
int swm_generate_universal_srs(swm_ctx *ctx, size_t num_constraints, size_t num_variables, size_t num_non_zero,
                               swm_rng *rng, swm_srs **out);
void swm_srs_destroy(swm_ctx *ctx, swm_srs *srs);
size_t swm_srs_max_degree(const swm_srs *srs);
/* UniversalSRS <-> arkworks' in-memory kzg10::UniversalParams (what a binding that keeps the reference's
 * `Box<UniversalSRS>` return type / `&UniversalSRS` parameter needs, src/marlin/mod.rs:50,89).
 * export: powers_of_g[first .. first + count) as affine Montgomery x,y (count x 12 limbs); the three gamma-powers
 *         KZG hiding with bound 1 uses (36 limbs; arkworks' map holds max_degree + 2 of them, MarlinKZG10::trim reads
 *         indices 0..=2 only); h and beta_h in G2 as x.c0, x.c1, y.c0, y.c1 (4 x 6 Montgomery limbs, all-zero =
 *         infinity).  Any of the three small outputs may be NULL.
 * import: the same values in; the result behaves like a setup produced here.  Points are taken as given (an in-memory
 *         struct is not validated by arkworks either). */
int swm_srs_export(swm_ctx *ctx, const swm_srs *srs, size_t first, size_t count, uint64_t *powers_xy,
                   uint64_t gamma_xy[36], uint64_t h[24], uint64_t beta_h[24]);
int swm_srs_import(swm_ctx *ctx, const uint64_t *powers_xy, size_t n_powers, const uint64_t gamma_xy[36],
                   const uint64_t h[24], const uint64_t beta_h[24], swm_srs **out);
/* i-th power of g (affine Montgomery x,y) — test hook */
int swm_srs_power_of_g(swm_ctx *ctx, const swm_srs *srs, size_t i, uint64_t out_xy[12]);

/* generate_proving_and_verifying_keys (src/marlin/mod.rs:88-94): MarlinInst::index_from_constraint_system */
int swm_generate_proving_and_verifying_keys(swm_ctx *ctx, const swm_srs *srs, const swm_r1cs *cs, swm_pk **pk,
                                            swm_vk **vk);
/* A proving key is resident per DEVICE, read-only once built, and reference-counted: the handle returned by
 * swm_generate_proving_and_verifying_keys / swm_pk_deserialize carries one reference, and ANY context on the key's device may
 * prove with it — several contexts (one per host thread) at the same time: scratch, streams and result slots are the
 * context's, the key is only read.  One resident copy (committer key, window tables, index tables: 13 GB at 2^20
 * constraints) serves every proving thread of the process, and outlives the context that built it.
 *   swm_pk_retain    one more reference (a second owner of the same handle, e.g. a process-wide key cache);
 *   swm_pk_attach    the same, after checking that `ctx` runs on the key's device (SWM_ERR_MISMATCH otherwise);
 *   swm_pk_destroy   drops one reference (draining `ctx`, which may be NULL, first); the last one frees the key;
 *   swm_pk_device / swm_pk_refcount   the key's device / its current reference count (diagnostics, tests). */
void swm_pk_destroy(swm_ctx *ctx, swm_pk *pk);
int swm_pk_retain(swm_pk *pk);
int swm_pk_attach(swm_ctx *ctx, swm_pk *pk);
int swm_pk_device(const swm_pk *pk);
int swm_pk_refcount(const swm_pk *pk);
void swm_vk_destroy(swm_vk *vk);

/* generate_proof (src/marlin/mod.rs:70-77): MarlinInst::prove_from_constraint_system(&pk, cs, rng).
 * Writes the CanonicalSerialize bytes of the proof (<= 1024 B).  SWM_ERR_UNSATISFIED when the witness does not
 * satisfy the constraints (the reference panics on a debug assertion inside ark-marlin at this point).
 * ASSIGNMENT ONLY: the prover reads num_instance, num_witness, num_constraints, instance and witness of `cs` and nothing
 * else — the matrices are the key's (as in ark-marlin, whose prover_init takes them from the index and never calls
 * to_matrices() at prove time), so the nine matrix pointers of `cs` may be NULL and a binding need not flatten A, B, C per
 * proof.  A shape that does not match the key is SWM_ERR_MISMATCH (ark-marlin: InstanceDoesNotMatchIndex).  `ctx` must run
 * on the key's device (SWM_ERR_MISMATCH otherwise); it need not be the context that built the key. */
int swm_generate_proof(swm_ctx *ctx, const swm_pk *pk, const swm_r1cs *cs, swm_rng *rng, uint8_t *proof_out,
                       size_t cap, size_t *len);
/* The same proof in the form CanonicalSerialize::serialize_uncompressed writes (flags = SWM_PROOF_UNCOMPRESSED; <= 2048 B): every
 * G1 point as x, y with the infinity flag in the top bits of y's last byte, everything else unchanged.  For a binding that turns
 * the bytes back into an arkworks `Proof` in the same process: `Proof::deserialize_unchecked` on this form costs microseconds,
 * where the checked `Proof::deserialize` of the compressed form takes a square root and a subgroup check per commitment (~2.3 ms
 * for a proof, measured on the library's own checked reader: `drop_in.checked_deserialize_proxy_ms` in the bench line) — the bytes
 * come from this library, not from an untrusted peer.  flags = 0 is swm_generate_proof.
 * swm_proof_recode converts between the two forms on the host (checked parse of the input form; out == NULL reports the length):
 * serialize(deserialize_unchecked(uncompressed bytes)) on the Rust side gives the bytes swm_generate_proof would have written. */
#define SWM_PROOF_UNCOMPRESSED 1u
int swm_generate_proof_ex(swm_ctx *ctx, const swm_pk *pk, const swm_r1cs *cs, swm_rng *rng, unsigned flags,
                          uint8_t *proof_out, size_t cap, size_t *len);
int swm_proof_recode(const uint8_t *bytes, size_t len, int to_uncompressed, uint8_t *out, size_t cap, size_t *out_len);

/* verify_proof (src/marlin/mod.rs:79-86).  public_inputs: n x 4 Montgomery limbs (without the leading one).
 * Host-only (two pairings); needs no GPU and no context. */
int swm_verify_proof(const swm_vk *vk, const uint64_t *public_inputs, size_t n, const uint8_t *proof, size_t len,
                     swm_rng *rng, int *ok);

/* serialization.rs: serialize_/deserialize_verifying_key, deserialize_proof (validation of the byte string) */
int swm_vk_serialize(const swm_vk *vk, uint8_t *out, size_t cap, size_t *len);
int swm_vk_deserialize(const uint8_t *bytes, size_t len, swm_vk **out);
int swm_proof_validate(const uint8_t *bytes, size_t len);
/* serialize_proving_key / deserialize_proving_key (src/marlin/serialization.rs:33-45): the CanonicalSerialize bytes of
 * ark_marlin::IndexProverKey — index_vk, index_comm_rands, index (info, matrices, the three matrix arithmetisations with
 * their polynomials and evaluation tables), committer_key (trimmed powers, shifted powers, gamma powers, degree bounds,
 * max_degree), points compressed — so that ProvingKey::deserialize(bytes) / proving_key.serialize() on the Rust side
 * move a key across the boundary.  Field order as recalled from ark-marlin / ark-poly-commit 0.3.0 (not vendored in the
 * reference: unpinned, like the proof layout).  ~2.5 GB at |K| = 2^20.  Deserialisation performs arkworks' checks
 * (canonical field elements, points on the curve and in the prime-order subgroup — on the GPU for the committer key)
 * and recomputes everything that is derived from the matrices instead of trusting it.
 * swm_pk_serialize with out == NULL only reports the length. */
int swm_pk_serialize(swm_ctx *ctx, const swm_pk *pk, uint8_t *out, size_t cap, size_t *len);
int swm_pk_deserialize(swm_ctx *ctx, const uint8_t *bytes, size_t len, swm_pk **out);

/* K3 as the reference exercises it: ConstraintSystem::is_satisfied (src/merkle_tree/simple_merkle_tree.rs:197-199):
 * A z o B z == C z on the GPU.  *ok = 1 when satisfied, else *first_bad = index of the first unsatisfied row. */
int swm_r1cs_is_satisfied(swm_ctx *ctx, const swm_r1cs *cs, int *ok, size_t *first_bad);

/* transcript primitives, exposed for known-answer tests */
int swm_blake2s(const uint8_t *data, size_t len, uint8_t out[32]);
int swm_chacha_block(const uint8_t key[32], uint64_t counter, int rounds, uint8_t out[64]);

/* ---------------------------------------------------------------------------------------------- Pedersen CRH + Merkle tree
 * The tree BASELINE config #5 builds before it proves membership ("below the line").  Replaces, on the GPU,
 *   /root/reference/src/merkle_tree/simple_merkle_tree.rs:47-49   MerkleTree::<MerkleConfig>::new(&leaf_params, &two_to_one_params, leaves)
 *   /root/reference/src/merkle_tree/common.rs:11-30               LeafHash / TwoToOneHash = PedersenCRHCompressor<EdwardsProjective, TECompressor, W>
 *   /root/reference/src/hash/mod.rs:13-28                         the Pedersen CRH called on its own
 * swm_pedersen = pedersen::Parameters<EdwardsProjective> [U]: generators[w][j] = 2^j g_w, w < num_windows, j < window_size,
 * handed over as affine (x, y), 2 x 32 little-endian bytes each in standard form (to_bytes! of the affine point), window
 * after window.  Refused with SWM_ERR_INVALID_ARG: a coordinate >= r, a point off the curve, a generator that is not twice
 * its predecessor.  A digest is the affine x coordinate of the sum (TECompressor), 32 little-endian bytes.
 * swm_pedersen_hash: `count` inputs of `input_len` bytes each, back to back; input_len x 8 <= num_windows x window_size
 * (else SWM_ERR_INVALID_ARG, where ark-crypto-primitives panics); bits LSB-first inside a byte, missing bits zero.
 * swm_merkle_tree_build: n_leaves (a power of two >= 2) leaves of leaf_len bytes each (to_bytes! of a leaf: 1 for u8);
 * nodes = n leaf digests | n / 2 two-to-one digests of (left || right) | ... | root: (2 n - 1) x 32 bytes.  Sibling of
 * node i of a level is i ^ 1; the path of leaf i is level[l][(i >> l) ^ 1].  _dev: device pointers (digests 4-byte aligned). */
typedef struct swm_pedersen swm_pedersen;
int swm_pedersen_create(swm_ctx *ctx, const uint8_t *generators_xy, size_t num_windows, size_t window_size, swm_pedersen **out);
void swm_pedersen_destroy(swm_ctx *ctx, swm_pedersen *params);
int swm_pedersen_hash(swm_ctx *ctx, const swm_pedersen *params, const uint8_t *inputs, size_t input_len, size_t count,
                      uint8_t *digests);
int swm_pedersen_hash_dev(swm_ctx *ctx, const swm_pedersen *params, const void *d_inputs, size_t input_len, size_t count,
                          void *d_digests);
int swm_merkle_tree_build(swm_ctx *ctx, const swm_pedersen *leaf_params, const swm_pedersen *two_to_one_params,
                          const uint8_t *leaves, size_t leaf_len, size_t n_leaves, uint8_t *nodes);
int swm_merkle_tree_build_dev(swm_ctx *ctx, const swm_pedersen *leaf_params, const swm_pedersen *two_to_one_params,
                              const void *d_leaves, size_t leaf_len, size_t n_leaves, void *d_nodes);

/* ---------------------------------------------------------------------------------------------- one proof over several GPUs
 * every commitment MSM of swm_generate_proof / swm_generate_proving_and_verifying_keys is split by
 * point range — rank g of `world` takes coefficients and SRS powers [g n / world, (g+1) n / world) — and the
 * per-rank partial sums (one 192-byte XYZZ point per MSM) are exchanged through `allgather`, which must behave like
 * MPI_Allgather on `bytes` bytes per rank (recv holds world * bytes, rank order).  EC addition is not an RCCL
 * reduction, so the exchange is an all-gather followed by the same rank-ordered sum on every rank; every rank then
 * holds the same commitment and emits the same proof bytes as a single-GPU run.  Everything else of the prover
 * (transforms, pointwise work, transcript) is replicated: it is cheaper to recompute a 2^20-point NTT (0.15 ms)
 * than to move its 32 MB over xGMI.  world = 1 (the default) or allgather = NULL switches sharding off.
 * The callback is invoked on the thread that called into the library, between kernels (the stream is idle). */
typedef int (*swm_allgather_fn)(void *user, const void *send, size_t bytes, void *recv);
int swm_set_msm_sharding(swm_ctx *ctx, unsigned rank, unsigned world, swm_allgather_fn allgather, void *user);

/* The same exchange through RCCL INSIDE the library (one process per GPU, backend RCCL over xGMI): the partial sums of
 * ALL commitments of a prover round travel in ONE ncclAllGather on the context's stream (k x 192 bytes per rank;
 * 3-4 exchanges per proof), no callback, no host framework in the data path.  librccl is resolved at run time (swm_rccl_info:
 * SWM_RCCL_PATH, else the copy already mapped in the process, else librccl.so.1).
 *   swm_rccl_unique_id  rank 0 creates the 128-byte ncclUniqueId and hands it to the other ranks by any means;
 *   swm_rccl_init       every rank: ncclCommInitRank(world, id, rank) for this context, then sharding is on;
 *   swm_set_rccl_comm   alternatively adopt a communicator the caller owns (same RCCL build); NULL switches back;
 *   swm_exchange_stats  all-gathers issued so far and bytes contributed per rank (what a bench reports). */
int swm_rccl_unique_id(uint8_t out[128]);
int swm_rccl_init(swm_ctx *ctx, const uint8_t id[128], unsigned rank, unsigned world);
int swm_set_rccl_comm(swm_ctx *ctx, void *nccl_comm, unsigned rank, unsigned world);
int swm_exchange_stats(swm_ctx *ctx, uint64_t *calls, uint64_t *bytes_per_rank);
/* Which RCCL carries the exchanges of this process, and how it was found: "librccl <path> version <ncclGetVersion> (<how>)", or why
 * none is usable (the return value is then SWM_ERR_INTERNAL and buf says why).  Resolution is deterministic: (1) SWM_RCCL_PATH —
 * that file or an error; (2) a librccl the process has ALREADY mapped (torch's copy when torch was imported first; read from
 * /proc/self/maps) — never a second copy beside it; (3) librccl.so.1, then librccl.so, on the loader's search path.  The same
 * string is part of every RCCL error message (swm_last_error) and of bench.py's `sharded` object. */
int swm_rccl_info(char *buf, size_t cap);

/* ---------------------------------------------------------------------------------------------- measurement
 * Per-kernel HIP-event log on the context's stream (SURVEY.md §5 "per-kernel event log"): when enabled every
 * kernel launch is bracketed by hipEventRecord on the stream it is launched on (on = 1), or only the launches of the
 * kernels bench.py prices against a roofline — msm_accumulate, ntt_pass, spmv_* — (on = 2; bracketing all ~700 launches
 * of a 2^20 proof costs ~2 % of the proof).  swm_profile_json writes
 * {"kernels":[{"name":..,"calls":..,"total_ms":..,"avg_ms":..}, ...],
 *  "work":{"msm_calls","msm_points","msm_digits" (points x windows),"msm_adds" (non-zero digits = mixed additions),
 *          "ntt_calls","ntt_elements","spmv_calls","spmv_rows","spmv_nnz"}} into buf. */
int swm_profile_enable(swm_ctx *ctx, int on);
/* hipMemGetInfo on the context's device: HBM free / total in bytes (what a resident key costs, tests and bench) */
int swm_device_mem_info(swm_ctx *ctx, size_t *free_bytes, size_t *total_bytes);
int swm_profile_reset(swm_ctx *ctx);
int swm_profile_json(swm_ctx *ctx, char *buf, size_t buflen);

/* Device self-test of the field / curve primitives the kernels are built from: computes a[i]*b[i] in Fq (which = 0),
 * or in Fr (which = 1), element-wise on the GPU.  Inputs/outputs are host buffers in Montgomery form.
 * which = 2, 5, 6 exercise the MSM's 28-bit lazy-limb multipliers of csrc/fq28.cuh (plain, squarer, fused two-product)
 * on packed 384-bit integers; results are canonical residues times 2^-392. */
/* the device-buffer exchanges of the sharded transform over whatever sharding the context has: alltoall != 0 — chunk c of
 * d_send (bytes_per_peer bytes) to rank c, chunk i of d_recv from rank i; else an all-gather of bytes_per_peer bytes */
int swm_selftest_exchange(swm_ctx *ctx, const void *d_send, void *d_recv, size_t bytes_per_peer, int alltoall);
int swm_selftest_mul(swm_ctx *ctx, int which, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
/* out[i] = jacobian(a[i] (+) b[i]) with a, b affine (n x 12 limbs): exercises the mixed/XYZZ adders incl. doubling. */
int swm_selftest_g1_add(swm_ctx *ctx, const uint64_t *a_xy, const uint64_t *b_xy, uint64_t *out_jac, size_t n);
/* Throughput probe: `iters` dependent Montgomery multiplications per thread on `threads` threads; returns ms. */
int swm_selftest_mul_throughput(swm_ctx *ctx, int which, size_t threads, int iters, float *ms);
/* Host pairing code of the verifier against itself (no GPU): bit k of *failed is set when identity k does not hold —
 * 0 cyclotomic squaring == plain squaring on the cyclotomic subgroup; 1 the 4-bit-window hard part == the plain power by
 * (q^6 + 1) / r; 2 the addition chain == the cube of that; 3 q-Frobenius twice == q^2-Frobenius; 4 e(2P, Q) == e(P, Q)^2 and
 * e(P, Q) != 1; 5 e(P, Q) e(-P, Q) == 1 and e(P, Q)^2 != 1 through product_of_pairings_is_one; 6 the shared Miller
 * accumulator of two pairs == the product of two single loops. */
int swm_selftest_pairing(unsigned *failed);
/* Host-side self-test of the single-element Fr inversion the device kernels use (frinv.cuh, fr_inv_bingcd: binary GCD on
 * 64-bit approximations; tail of ark_ff::batch_inversion's one field inversion): out = a^-1 for n Montgomery-form elements
 * a != 0, computed on the CPU by the same function the GPU lanes run; *fallbacks counts inputs whose 17 rounds did not end
 * in (0, 1) (the kernels then use the exact loop; expected 0). */
int swm_selftest_fr_inv(const uint64_t *a_mont, uint64_t *out_mont, size_t n, unsigned *fallbacks);

#ifdef __cplusplus
}
#endif
#endif
