//! Raw bindings: one declaration per symbol of include/swmarlin.h that the Marlin surface needs (plus the K1-K4 kernel
//! entry points for callers that keep their own arkworks pipeline).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_uint, c_void};

#[repr(C)] pub struct swm_ctx { _p: [u8; 0] }
#[repr(C)] pub struct swm_rng { _p: [u8; 0] }
#[repr(C)] pub struct swm_srs { _p: [u8; 0] }
#[repr(C)] pub struct swm_pk { _p: [u8; 0] }
#[repr(C)] pub struct swm_vk { _p: [u8; 0] }
#[repr(C)] pub struct swm_bases { _p: [u8; 0] }
#[repr(C)] pub struct swm_pedersen { _p: [u8; 0] }

pub const SWM_OK: c_int = 0;
pub const SWM_ERR_UNSATISFIED: c_int = -5;
pub const SWM_PROOF_UNCOMPRESSED: c_uint = 1;

/// struct swm_r1cs: a synthesised constraint system as flat arrays (instance[0] is the constant one).
#[repr(C)]
pub struct swm_r1cs {
    pub num_instance: usize,
    pub num_witness: usize,
    pub num_constraints: usize,
    pub instance: *const u64,
    pub witness: *const u64,
    pub a_rowptr: *const u32, pub a_col: *const u32, pub a_val: *const u64,
    pub b_rowptr: *const u32, pub b_col: *const u32, pub b_val: *const u64,
    pub c_rowptr: *const u32, pub c_col: *const u32, pub c_val: *const u64,
}

pub type swm_fill_bytes_fn = unsafe extern "C" fn(user: *mut c_void, dest: *mut u8, len: usize);
pub type swm_allgather_fn = unsafe extern "C" fn(user: *mut c_void, send: *const c_void, bytes: usize, recv: *mut c_void) -> c_int;

extern "C" {
    pub fn swm_version() -> c_int;
    pub fn swm_strerror(code: c_int) -> *const c_char;
    pub fn swm_init(device: c_int, out: *mut *mut swm_ctx) -> c_int;
    pub fn swm_destroy(ctx: *mut swm_ctx);
    pub fn swm_last_error(ctx: *mut swm_ctx) -> *const c_char;

    // generate_rand / caller-owned randomness (src/marlin/mod.rs:33-35, :49, :73, :83)
    pub fn swm_rng_test_new(out: *mut *mut swm_rng) -> c_int;
    pub fn swm_rng_from_seed(seed: *const u8, out: *mut *mut swm_rng) -> c_int;
    pub fn swm_rng_from_callback(fill_bytes: swm_fill_bytes_fn, user: *mut c_void, out: *mut *mut swm_rng) -> c_int;
    pub fn swm_rng_from_chacha(key: *const u8, word_pos: u64, rounds: c_int, out: *mut *mut swm_rng) -> c_int;
    pub fn swm_rng_word_pos(rng: *const swm_rng, word_pos: *mut u64) -> c_int;
    pub fn swm_rng_fill_bytes(rng: *mut swm_rng, dest: *mut u8, len: usize) -> c_int;
    pub fn swm_rng_fill_bytes_cb(user: *mut c_void, dest: *mut u8, len: usize);
    pub fn swm_rng_free(rng: *mut swm_rng);

    // generate_universal_srs (src/marlin/mod.rs:45-55)
    pub fn swm_generate_universal_srs(ctx: *mut swm_ctx, nc: usize, nv: usize, nnz: usize, rng: *mut swm_rng,
                                      out: *mut *mut swm_srs) -> c_int;
    pub fn swm_srs_destroy(ctx: *mut swm_ctx, srs: *mut swm_srs);
    pub fn swm_srs_max_degree(srs: *const swm_srs) -> usize;
    pub fn swm_srs_export(ctx: *mut swm_ctx, srs: *const swm_srs, first: usize, count: usize, powers_xy: *mut u64,
                          gamma_xy: *mut u64, h: *mut u64, beta_h: *mut u64) -> c_int;
    pub fn swm_srs_import(ctx: *mut swm_ctx, powers_xy: *const u64, n_powers: usize, gamma_xy: *const u64,
                          h: *const u64, beta_h: *const u64, out: *mut *mut swm_srs) -> c_int;

    // generate_proving_and_verifying_keys (src/marlin/mod.rs:88-94)
    pub fn swm_generate_proving_and_verifying_keys(ctx: *mut swm_ctx, srs: *const swm_srs, cs: *const swm_r1cs,
                                                   pk: *mut *mut swm_pk, vk: *mut *mut swm_vk) -> c_int;
    pub fn swm_pk_destroy(ctx: *mut swm_ctx, pk: *mut swm_pk);
    pub fn swm_pk_retain(pk: *mut swm_pk) -> c_int;
    pub fn swm_pk_attach(ctx: *mut swm_ctx, pk: *mut swm_pk) -> c_int;
    pub fn swm_pk_device(pk: *const swm_pk) -> c_int;
    pub fn swm_pk_refcount(pk: *const swm_pk) -> c_int;
    pub fn swm_vk_destroy(vk: *mut swm_vk);

    // generate_proof (src/marlin/mod.rs:70-77) and verify_proof (:79-86)
    pub fn swm_generate_proof(ctx: *mut swm_ctx, pk: *const swm_pk, cs: *const swm_r1cs, rng: *mut swm_rng,
                              proof_out: *mut u8, cap: usize, len: *mut usize) -> c_int;
    /// flags: SWM_PROOF_UNCOMPRESSED = 1 (the proof as serialize_uncompressed bytes, <= 2048 B)
    pub fn swm_generate_proof_ex(ctx: *mut swm_ctx, pk: *const swm_pk, cs: *const swm_r1cs, rng: *mut swm_rng, flags: c_uint,
                                 proof_out: *mut u8, cap: usize, len: *mut usize) -> c_int;
    pub fn swm_proof_recode(bytes: *const u8, len: usize, to_uncompressed: c_int, out: *mut u8, cap: usize, out_len: *mut usize) -> c_int;
    pub fn swm_verify_proof(vk: *const swm_vk, public_inputs: *const u64, n: usize, proof: *const u8, len: usize,
                            rng: *mut swm_rng, ok: *mut c_int) -> c_int;

    // src/marlin/serialization.rs:5-45 (ark-serialize bytes in both directions)
    pub fn swm_vk_serialize(vk: *const swm_vk, out: *mut u8, cap: usize, len: *mut usize) -> c_int;
    pub fn swm_vk_deserialize(bytes: *const u8, len: usize, out: *mut *mut swm_vk) -> c_int;
    pub fn swm_proof_validate(bytes: *const u8, len: usize) -> c_int;
    pub fn swm_pk_serialize(ctx: *mut swm_ctx, pk: *const swm_pk, out: *mut u8, cap: usize, len: *mut usize) -> c_int;
    pub fn swm_pk_deserialize(ctx: *mut swm_ctx, bytes: *const u8, len: usize, out: *mut *mut swm_pk) -> c_int;
    pub fn swm_r1cs_is_satisfied(ctx: *mut swm_ctx, cs: *const swm_r1cs, ok: *mut c_int, first_bad: *mut usize) -> c_int;

    // one proof over several GPUs
    pub fn swm_set_msm_sharding(ctx: *mut swm_ctx, rank: c_uint, world: c_uint, allgather: Option<swm_allgather_fn>,
                                user: *mut c_void) -> c_int;
    pub fn swm_set_rccl_comm(ctx: *mut swm_ctx, nccl_comm: *mut c_void, rank: c_uint, world: c_uint) -> c_int;

    // K1-K4 for callers that keep arkworks' prover and only swap kernels
    pub fn swm_srs_upload(ctx: *mut swm_ctx, xy: *const u64, n: usize, out: *mut *mut swm_bases) -> c_int;
    pub fn swm_srs_free(ctx: *mut swm_ctx, bases: *mut swm_bases) -> c_int;
    pub fn swm_msm_g1(ctx: *mut swm_ctx, bases: *const swm_bases, offset: usize, scalars: *const u64, n: usize,
                      out_jac: *mut u64) -> c_int; // VariableBaseMSM::multi_scalar_mul
    pub fn swm_ntt_fr(ctx: *mut swm_ctx, data: *mut u64, log_n: c_uint, inverse: c_int, coset: c_int) -> c_int;
    pub fn swm_ntt_fr_sharded_dev(ctx: *mut swm_ctx, d_local: *mut c_void, log_n: c_uint, inverse: c_int, blocks_in: c_int) -> c_int;
    pub fn swm_spmv_fr(ctx: *mut swm_ctx, rowptr: *const u32, col: *const u32, val: *const u64, z: *const u64,
                       z_len: usize, out: *mut u64, rows: usize, nnz: usize) -> c_int;
    pub fn swm_batch_inverse_fr(ctx: *mut swm_ctx, data: *mut u64, n: usize) -> c_int;

    // the native Pedersen hash and MerkleTree::new of src/merkle_tree/simple_merkle_tree.rs:47-49, src/hash/mod.rs:23-28
    pub fn swm_pedersen_create(ctx: *mut swm_ctx, generators_xy: *const u8, num_windows: usize, window_size: usize,
                               out: *mut *mut swm_pedersen) -> c_int;
    pub fn swm_pedersen_destroy(ctx: *mut swm_ctx, params: *mut swm_pedersen);
    pub fn swm_pedersen_hash(ctx: *mut swm_ctx, params: *const swm_pedersen, inputs: *const u8, input_len: usize, count: usize,
                             digests: *mut u8) -> c_int;
    pub fn swm_merkle_tree_build(ctx: *mut swm_ctx, leaf_params: *const swm_pedersen, two_to_one_params: *const swm_pedersen,
                                 leaves: *const u8, leaf_len: usize, n_leaves: usize, nodes: *mut u8) -> c_int;
}
