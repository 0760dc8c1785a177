//! Drop-in for /root/reference/src/marlin/mod.rs:12-94 — same type aliases, same five functions, same signatures
//! (`&mut StdRng`, arkworks `ProvingKey` / `VerifyingKey` by value, `Box<UniversalSRS>`), same error convention
//! (`anyhow!("{:?}", e)`), with the arithmetic running in libswmarlin.so on an MI355X.
//!
//! How arkworks values cross the boundary
//! * `&mut StdRng`         -> through `swm_rng_from_callback` over a `fill_bytes` trampoline (the caller's stream, word for
//!                            word, at the caller's speed: r05 measured 77 - 79 ms per 2^20 proof against the 50-ms headline, which
//!                            is the built-in generator — see `drop_in_rng` in the bench line); with the cargo feature
//!                            `adopt-std-rng` by STATE — `swm_rng_from_chacha(seed, word position)`, position written back
//!                            afterwards: the library then produces the ChaCha12 stream itself (on the GPU for the 3|H| mask
//!                            coefficients).  Opt-in because it views the StdRng through a layout rand does not guarantee.
//! * `ConstraintSystemRef` -> the indexer takes `PackedR1cs::from_cs` (matrices + assignments as flat arrays); the PROVER takes
//!                            `AssignmentOnly::from_cs`: the two assignment vectors where they are and the shape, no
//!                            `to_matrices()`, no CSR copies (the matrices are the key's; `swm_generate_proof` accepts NULL
//!                            matrix pointers).
//! * `UniversalSRS`        -> `swm_srs_export` / `swm_srs_import` (the fields of kzg10::UniversalParams).
//! * `ProvingKey`, `VerifyingKey`, `MarlinProof` -> their CanonicalSerialize bytes, which the library reads and writes
//!                            (`swm_pk_*`, `swm_vk_*`, proof bytes) — the interchange format src/marlin/serialization.rs defines.
//! Device-resident twins of keys are cached PER PROCESS AND DEVICE (`KEYS`: a mutex-guarded LRU of at most `PK_CACHE`
//! reference-counted handles, by a digest of the verifying key and the committer-key shape): a key is resident per device,
//! read-only, and every proving thread attaches its own context to the one copy (`swm_pk_attach`) — eight proving threads
//! hold ONE 13-GB table set at 2^20 constraints, not eight, and the import (serialise + upload + re-derive tables) is paid
//! once per key and process.  SRS twins stay per thread (they are only read by `index`), keyed by their first powers.
//! `generate_proof(cs, proving_key, rng)` — which takes the key BY VALUE in the reference — therefore pays, per proof, a
//! verifying-key digest, a cache lookup and the GPU proof (tests/native/dropin_harness.cpp measures exactly that sequence
//! against the C ABI: `drop_in` in the bench line).
use crate::convert::*;
use crate::ffi::*;
use anyhow::{anyhow, Result};
use ark_bls12_377::{Bls12_377, Fr, FrParameters, G1Affine, G2Affine, Parameters};
use ark_ec::bls12::Bls12;
use ark_ff::Fp256;
use ark_marlin::{IndexProverKey, IndexVerifierKey, Marlin, Proof, SimpleHashFiatShamirRng};
use ark_poly::univariate::DensePolynomial;
use ark_poly_commit::marlin_pc::MarlinKZG10;
use ark_serialize::{CanonicalDeserialize, CanonicalSerialize};
use blake2::Blake2s;
use digest::Digest;
use rand::rngs::StdRng;
use rand::RngCore;
use rand_chacha::ChaChaRng;
use ark_relations::r1cs::{ConstraintSynthesizer, ConstraintSystem, OptimizationGoal, SynthesisMode};
use std::cell::RefCell;
use std::collections::{BTreeMap, HashMap};
use std::ffi::CStr;
use std::os::raw::{c_int, c_void};

pub type MultiPC = MarlinKZG10<Bls12_377, DensePolynomial<Fr>>;
pub type FS = SimpleHashFiatShamirRng<Blake2s, ChaChaRng>;
/// arkworks' own CPU prover (`Marlin<Fr, MultiPC, FS>`): what the reference calls `MarlinInst`.  Kept under another name for
/// A/B checks; the name `MarlinInst` below is the MI355X implementation with the same associated functions.
pub type ArkMarlinInst = Marlin<Fr, MultiPC, FS>;
/// The error type of `Marlin::{universal_setup, index, prove, verify}` for this instantiation.
pub type MarlinError = ark_marlin::Error<ark_poly_commit::Error>;
pub type UniversalSRS = ark_marlin::UniversalSRS<Fr, MultiPC>;
pub type ConstraintSystemRef = ark_relations::r1cs::ConstraintSystemRef<Fr>;
pub type VerifyingKey =
    IndexVerifierKey<Fp256<FrParameters>, MarlinKZG10<Bls12<Parameters>, DensePolynomial<Fp256<FrParameters>>>>;
pub type ProvingKey =
    IndexProverKey<Fp256<FrParameters>, MarlinKZG10<Bls12<Parameters>, DensePolynomial<Fp256<FrParameters>>>>;
pub type MarlinProof = Proof<Fr, MarlinKZG10<Bls12<Parameters>, DensePolynomial<Fp256<FrParameters>>>>;
/// `crate::gadgets::ConstraintF` in simpleworks (= ark_ed_on_bls12_377::Fq = BLS12-377 Fr, src/gadgets/mod.rs:29)
pub type ConstraintF = Fr;

// ------------------------------------------------------------------------------------------------ errors
#[derive(Debug)]
pub struct SwmError {
    pub code: c_int,
    pub what: &'static str,
    pub detail: String,
}
fn cstr(p: *const std::os::raw::c_char) -> String {
    if p.is_null() {
        return String::new();
    }
    unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
}
fn check(rc: c_int, what: &'static str, ctx: *mut swm_ctx) -> std::result::Result<(), SwmError> {
    if rc == SWM_OK {
        return Ok(());
    }
    let detail = format!("{}: {}", cstr(unsafe { swm_strerror(rc) }), cstr(unsafe { swm_last_error(ctx) }));
    Err(SwmError { code: rc, what, detail })
}

// ------------------------------------------------------------------------------------------------ resident keys: per process and device
/// Resident keys of this process, most recently used last; at most `PK_CACHE` per device (each holds its committer key and
/// MSM tables in HBM: 13 GB at 2^20 constraints).  The cache owns ONE reference per entry (`swm_pk` is reference-counted);
/// a thread that proves with a key holds a reference of its own for the duration of the call (`swm_pk_attach` ..
/// `swm_pk_destroy`), so evicting an entry another thread is proving with only drops the cache's reference — the key is freed
/// when the last holder lets go.
const PK_CACHE: usize = 4;
struct SharedPk(*mut swm_pk);
unsafe impl Send for SharedPk {} // an opaque, internally synchronised (atomic refcount, read-only) library handle
struct KeyCache {
    entries: Vec<(c_int, [u8; 32], SharedPk)>, // (device, digest, handle)
}
static KEYS: std::sync::Mutex<KeyCache> = std::sync::Mutex::new(KeyCache { entries: Vec::new() });
impl KeyCache {
    /// The cached handle with one more reference taken for the caller's context (`swm_pk_attach`), or None.
    fn get_attached(&mut self, ctx: *mut swm_ctx, device: c_int, key: &[u8; 32]) -> std::result::Result<Option<*mut swm_pk>, SwmError> {
        let i = match self.entries.iter().position(|(d, k, _)| *d == device && k == key) {
            Some(i) => i,
            None => return Ok(None),
        };
        let e = self.entries.remove(i);
        let h = e.2 .0;
        self.entries.push(e);
        check(unsafe { swm_pk_attach(ctx, h) }, "swm_pk_attach", ctx)?;
        Ok(Some(h))
    }
    /// Takes over the caller's reference to `h`.
    fn put(&mut self, device: c_int, key: [u8; 32], h: *mut swm_pk) {
        if let Some(i) = self.entries.iter().position(|(d, k, _)| *d == device && *k == key) {
            let (_, _, old) = self.entries.remove(i);
            unsafe { swm_pk_destroy(std::ptr::null_mut(), old.0) };
        }
        if self.entries.iter().filter(|(d, _, _)| *d == device).count() >= PK_CACHE {
            let i = self.entries.iter().position(|(d, _, _)| *d == device).expect("counted above");
            let (_, _, old) = self.entries.remove(i);
            unsafe { swm_pk_destroy(std::ptr::null_mut(), old.0) };
        }
        self.entries.push((device, key, SharedPk(h)));
    }
}

// ------------------------------------------------------------------------------------------------ per-thread state
// A ConstraintSystemRef is Rc<RefCell<..>> (!Send): one proof is driven by one thread, and so is one context.
struct State {
    ctx: *mut swm_ctx,
    device: c_int,
    srs: HashMap<[u8; 32], *mut swm_srs>,
}
impl State {
    fn new() -> std::result::Result<Self, SwmError> {
        let device = std::env::var("SWM_DEVICE").ok().and_then(|s| s.parse::<c_int>().ok()).unwrap_or(0);
        let mut ctx = std::ptr::null_mut();
        check(unsafe { swm_init(device, &mut ctx) }, "swm_init", std::ptr::null_mut())?;
        Ok(State { ctx, device, srs: HashMap::new() })
    }
}
impl Drop for State {
    fn drop(&mut self) {
        unsafe {
            for (_, srs) in self.srs.drain() {
                swm_srs_destroy(self.ctx, srs);
            }
            swm_destroy(self.ctx);
        }
    }
}
thread_local! {
    static STATE: RefCell<Option<State>> = RefCell::new(None);
}
/// Runs one C-ABI call on this thread's context (crate::merkle uses it); `f` returns the library's status code.
pub(crate) fn with_context(f: impl FnOnce(*mut swm_ctx) -> c_int, what: &'static str) -> Result<()> {
    with_state(|st| check(f(st.ctx), what, st.ctx)).map_err(|e| anyhow!("{:?}", e))
}
fn with_state<T>(f: impl FnOnce(&mut State) -> std::result::Result<T, SwmError>) -> std::result::Result<T, SwmError> {
    STATE.with(|cell| {
        let mut slot = cell.borrow_mut();
        if slot.is_none() {
            *slot = Some(State::new()?);
        }
        f(slot.as_mut().expect("state initialised above"))
    })
}

// ------------------------------------------------------------------------------------------------ the caller's rng
unsafe extern "C" fn fill_bytes_trampoline<R: RngCore>(user: *mut c_void, dest: *mut u8, len: usize) {
    let rng = &mut *(user as *mut R);
    rng.fill_bytes(std::slice::from_raw_parts_mut(dest, len));
}
/// rand 0.8 defines `pub struct StdRng(rand_chacha::ChaCha12Rng)` (private field, no accessor).  A ChaCha12 generator is
/// fully described by (seed, stream, word position), which `ChaCha12Rng` exposes — so a `StdRng` can be handed to the library
/// BY STATE (`swm_rng_from_chacha`): the library then produces the caller's stream on the GPU instead of pulling ~170 MB per
/// 2^20-constraint proof through `fill_bytes`, and the position is written back afterwards (`set_word_pos`): the caller
/// observes exactly what it would have observed through the callback.  The view below relies on the single-field tuple struct
/// having its field's layout; `std_rng_view_is_sound` checks that once per process on a generator with a known stream
/// (size, alignment, seed, position, next words) and the callback path is used if anything differs.
#[cfg(feature = "adopt-std-rng")]
fn std_rng_as_chacha(rng: &mut StdRng) -> &mut rand_chacha::ChaCha12Rng {
    unsafe { &mut *(rng as *mut StdRng as *mut rand_chacha::ChaCha12Rng) }
}
#[cfg(feature = "adopt-std-rng")]
fn std_rng_view_is_sound() -> bool {
    use rand::SeedableRng;
    use std::sync::OnceLock;
    static OK: OnceLock<bool> = OnceLock::new();
    *OK.get_or_init(|| {
        if std::mem::size_of::<StdRng>() != std::mem::size_of::<rand_chacha::ChaCha12Rng>()
            || std::mem::align_of::<StdRng>() != std::mem::align_of::<rand_chacha::ChaCha12Rng>()
        {
            return false;
        }
        let seed = [7u8; 32];
        let mut a = StdRng::from_seed(seed);
        let mut b = rand_chacha::ChaCha12Rng::from_seed(seed);
        for _ in 0..5 {
            if a.next_u32() != b.next_u32() {
                return false;
            }
        }
        let view = std_rng_as_chacha(&mut a);
        if view.get_seed() != seed || view.get_stream() != 0 || view.get_word_pos() != 5 {
            return false;
        }
        view.set_word_pos(37);
        b.set_word_pos(37);
        a.next_u64() == b.next_u64()
    })
}

/// Runs `f` with a library handle that draws from `rng` (any `RngCore`, as `Marlin`'s associated functions accept; the
/// reference always passes its `StdRng`); the handle does not outlive the borrow.  A `StdRng` on stream 0 is handed over by
/// state (see above), everything else through the `fill_bytes` callback.
fn with_rng<R: RngCore + 'static, T>(rng: &mut R, f: impl FnOnce(*mut swm_rng) -> std::result::Result<T, SwmError>) -> std::result::Result<T, SwmError> {
    // State adoption reads the StdRng through a pointer cast whose layout rand does not guarantee (a tuple struct without
    // #[repr(transparent)]): formally undefined behaviour, so it is opt-in (cargo feature `adopt-std-rng`) until this crate
    // has been built and tested against the rand version in use.  The default is the fill_bytes callback: the same stream.
    #[cfg(feature = "adopt-std-rng")]
    if let Some(std_rng) = (rng as &mut dyn std::any::Any).downcast_mut::<StdRng>() {
        if std_rng_view_is_sound() {
            let inner = std_rng_as_chacha(std_rng);
            let pos = inner.get_word_pos();
            if inner.get_stream() == 0 && pos < (1u128 << 62) {
                let seed = inner.get_seed();
                let mut h = std::ptr::null_mut();
                check(unsafe { swm_rng_from_chacha(seed.as_ptr(), pos as u64, 12, &mut h) }, "swm_rng_from_chacha", std::ptr::null_mut())?;
                let out = f(h);
                let mut end = 0u64;
                let rc = unsafe { swm_rng_word_pos(h, &mut end) };
                unsafe { swm_rng_free(h) };
                check(rc, "swm_rng_word_pos", std::ptr::null_mut())?;
                inner.set_word_pos(end as u128); // the caller's generator continues where the library stopped
                return out;
            }
        }
    }
    let mut h = std::ptr::null_mut();
    check(
        unsafe { swm_rng_from_callback(fill_bytes_trampoline::<R>, rng as *mut R as *mut c_void, &mut h) },
        "swm_rng_from_callback",
        std::ptr::null_mut(),
    )?;
    let out = f(h);
    unsafe { swm_rng_free(h) };
    out
}

fn digest32(bytes: &[u8]) -> [u8; 32] {
    let mut out = [0u8; 32];
    out.copy_from_slice(Blake2s::digest(bytes).as_slice());
    out
}
fn srs_key(srs: &UniversalSRS) -> [u8; 32] {
    // the first powers identify a setup (beta and g); cheap to compute, no need to hash 100 MB
    let mut limbs = vec![0u64; 12 * 4];
    for (i, p) in srs.powers_of_g.iter().take(4).enumerate() {
        g1_limbs(p, &mut limbs[12 * i..12 * i + 12]);
    }
    let mut bytes = Vec::with_capacity(8 * limbs.len() + 8);
    bytes.extend_from_slice(&(srs.powers_of_g.len() as u64).to_le_bytes());
    for w in limbs {
        bytes.extend_from_slice(&w.to_le_bytes());
    }
    digest32(&bytes)
}
/// Identity of a proving key for the resident-twin cache: its verifying key AND its committer key (a key with the same
/// index under another SRS, or trimmed differently, is a different key).
fn pk_key(pk: &ProvingKey, vk_bytes: &[u8]) -> [u8; 32] {
    let ck = &pk.committer_key;
    let mut bytes = Vec::with_capacity(vk_bytes.len() + 8 * (3 + 12 * 4));
    bytes.extend_from_slice(vk_bytes);
    let shifted: &[G1Affine] = ck.shifted_powers.as_deref().unwrap_or(&[]);
    for v in [ck.powers.len() as u64, shifted.len() as u64, ck.max_degree as u64] {
        bytes.extend_from_slice(&v.to_le_bytes());
    }
    let mut limbs = [0u64; 12];
    for p in ck.powers.iter().take(2).chain(ck.powers.last()).chain(shifted.last()) {
        g1_limbs(p, &mut limbs);
        for w in limbs {
            bytes.extend_from_slice(&w.to_le_bytes());
        }
    }
    digest32(&bytes)
}
fn vk_key(vk: &VerifyingKey) -> std::result::Result<([u8; 32], Vec<u8>), SwmError> {
    let mut bytes = Vec::new();
    vk.serialize(&mut bytes).map_err(|e| SwmError { code: -7, what: "VerifyingKey::serialize", detail: format!("{:?}", e) })?;
    Ok((digest32(&bytes), bytes))
}

// ------------------------------------------------------------------------------------------------ the five functions
/// Return a pseudorandom number generator (src/marlin/mod.rs:33-35).
pub fn generate_rand() -> StdRng {
    ark_std::test_rng()
}

/// Generate the universal prover and verifier keys for the argument system (src/marlin/mod.rs:45-55).
pub fn generate_universal_srs(
    num_constraints: usize,
    num_variables: usize,
    num_non_zero: usize,
    rng: &mut StdRng,
) -> Result<Box<UniversalSRS>> {
    universal_setup(num_constraints, num_variables, num_non_zero, rng).map(Box::new).map_err(|e| anyhow!("{:?}", e))
}

fn universal_setup<R: RngCore + 'static>(nc: usize, nv: usize, nnz: usize, rng: &mut R) -> std::result::Result<UniversalSRS, SwmError> {
    with_state(|st| {
        let ctx = st.ctx;
        let handle = with_rng(rng, |r| {
            let mut h = std::ptr::null_mut();
            check(unsafe { swm_generate_universal_srs(ctx, nc, nv, nnz, r, &mut h) }, "swm_generate_universal_srs", ctx)?;
            Ok(h)
        })?;
        let n = unsafe { swm_srs_max_degree(handle) } + 1;
        let mut powers = vec![0u64; 12 * n];
        let (mut gamma, mut h, mut bh) = ([0u64; 36], [0u64; 24], [0u64; 24]);
        let rc = unsafe { swm_srs_export(ctx, handle, 0, n, powers.as_mut_ptr(), gamma.as_mut_ptr(), h.as_mut_ptr(), bh.as_mut_ptr()) };
        if let Err(e) = check(rc, "swm_srs_export", ctx) {
            unsafe { swm_srs_destroy(ctx, handle) };
            return Err(e);
        }
        let powers_of_g: Vec<G1Affine> = powers.chunks_exact(12).map(g1_from_limbs).collect();
        // arkworks keeps max_degree + 2 gamma powers; MarlinKZG10::trim reads indices 0..=hiding_bound + 1 = 0..=2 only
        let mut powers_of_gamma_g = BTreeMap::new();
        for i in 0..3 {
            powers_of_gamma_g.insert(i, g1_from_limbs(&gamma[12 * i..12 * i + 12]));
        }
        let (h_pt, beta_h): (G2Affine, G2Affine) = (g2_from_limbs(&h), g2_from_limbs(&bh));
        // (built under its concrete name: a struct expression cannot go through the associated-type alias UniversalSRS)
        let srs: UniversalSRS = ark_poly_commit::kzg10::UniversalParams::<Bls12_377> {
            powers_of_g,
            powers_of_gamma_g,
            h: h_pt,
            beta_h,
            neg_powers_of_h: BTreeMap::new(),
            prepared_h: h_pt.into(),
            prepared_beta_h: beta_h.into(),
        };
        st.srs.insert(srs_key(&srs), handle); // index() finds the resident twin instead of uploading it again
        Ok(srs)
    })
}

fn resident_srs(st: &mut State, srs: &UniversalSRS) -> std::result::Result<*mut swm_srs, SwmError> {
    let key = srs_key(srs);
    if let Some(h) = st.srs.get(&key) {
        return Ok(*h);
    }
    let n = srs.powers_of_g.len();
    let mut powers = vec![0u64; 12 * n];
    for (i, p) in srs.powers_of_g.iter().enumerate() {
        g1_limbs(p, &mut powers[12 * i..12 * i + 12]);
    }
    let mut gamma = [0u64; 36];
    for i in 0..3usize {
        let p = srs.powers_of_gamma_g.get(&i).ok_or(SwmError { code: -1, what: "UniversalSRS", detail: "missing gamma power".into() })?;
        g1_limbs(p, &mut gamma[12 * i..12 * i + 12]);
    }
    let (mut h, mut bh) = ([0u64; 24], [0u64; 24]);
    g2_limbs(&srs.h, &mut h);
    g2_limbs(&srs.beta_h, &mut bh);
    let mut out = std::ptr::null_mut();
    check(unsafe { swm_srs_import(st.ctx, powers.as_ptr(), n, gamma.as_ptr(), h.as_ptr(), bh.as_ptr(), &mut out) }, "swm_srs_import", st.ctx)?;
    st.srs.insert(key, out);
    Ok(out)
}

/// src/marlin/mod.rs:88-94
pub fn generate_proving_and_verifying_keys(
    universal_srs: &UniversalSRS,
    constraint_system: ConstraintSystemRef,
) -> Result<(ProvingKey, VerifyingKey)> {
    index(universal_srs, constraint_system).map_err(|e| anyhow!("{:?}", e))
}

fn bytes_of(mut call: impl FnMut(*mut u8, usize, *mut usize) -> c_int, what: &'static str, ctx: *mut swm_ctx) -> std::result::Result<Vec<u8>, SwmError> {
    let mut len = 0usize;
    check(call(std::ptr::null_mut(), 0, &mut len), what, ctx)?;
    let mut buf = vec![0u8; len];
    check(call(buf.as_mut_ptr(), len, &mut len), what, ctx)?;
    buf.truncate(len);
    Ok(buf)
}

fn index(srs: &UniversalSRS, cs: ConstraintSystemRef) -> std::result::Result<(ProvingKey, VerifyingKey), SwmError> {
    let packed = PackedR1cs::from_cs(&cs).map_err(|e| SwmError { code: -1, what: "to_matrices", detail: format!("{:?}", e) })?;
    with_state(|st| {
        let ctx = st.ctx;
        let srs_h = resident_srs(st, srs)?;
        let (mut pk_h, mut vk_h) = (std::ptr::null_mut(), std::ptr::null_mut());
        let r1cs = packed.as_ffi();
        check(unsafe { swm_generate_proving_and_verifying_keys(ctx, srs_h, &r1cs, &mut pk_h, &mut vk_h) },
              "swm_generate_proving_and_verifying_keys", ctx)?;
        let vk_bytes = bytes_of(|p, cap, len| unsafe { swm_vk_serialize(vk_h, p, cap, len) }, "swm_vk_serialize", ctx);
        unsafe { swm_vk_destroy(vk_h) };
        let pk_bytes = bytes_of(|p, cap, len| unsafe { swm_pk_serialize(ctx, pk_h, p, cap, len) }, "swm_pk_serialize", ctx);
        let (vk_bytes, pk_bytes) = match (vk_bytes, pk_bytes) {
            (Ok(v), Ok(p)) => (v, p),
            (Err(e), _) | (_, Err(e)) => {
                unsafe { swm_pk_destroy(ctx, pk_h) };
                return Err(e);
            }
        };
        let de = |e: ark_serialize::SerializationError| SwmError { code: -7, what: "CanonicalDeserialize", detail: format!("{:?}", e) };
        let vk = VerifyingKey::deserialize(&mut vk_bytes.as_slice()).map_err(de);
        let pk = ProvingKey::deserialize(&mut pk_bytes.as_slice()).map_err(de);
        match (pk, vk) {
            (Ok(pk), Ok(vk)) => {
                // the process-wide cache takes over this thread's reference: any thread's generate_proof finds the twin
                KEYS.lock().unwrap_or_else(|p| p.into_inner()).put(st.device, pk_key(&pk, &vk_bytes), pk_h);
                Ok((pk, vk))
            }
            (Err(e), _) | (_, Err(e)) => {
                unsafe { swm_pk_destroy(ctx, pk_h) };
                Err(e)
            }
        }
    })
}

/// The resident twin of `pk` with one reference taken for this thread's context; the caller releases it with
/// `swm_pk_destroy(st.ctx, h)` when its proof is done.
fn resident_pk(st: &mut State, pk: &ProvingKey) -> std::result::Result<*mut swm_pk, SwmError> {
    let (_, vk_bytes) = vk_key(&pk.index_vk)?;
    let key = pk_key(pk, &vk_bytes);
    if let Some(h) = KEYS.lock().unwrap_or_else(|p| p.into_inner()).get_attached(st.ctx, st.device, &key)? {
        return Ok(h);
    }
    // a key this process has not seen on this device (deserialised from disk, built by arkworks): move it in through its
    // bytes — outside the lock (seconds at 2^20); two threads racing on the same new key both import, the second put wins
    let mut bytes = Vec::new();
    pk.serialize(&mut bytes).map_err(|e| SwmError { code: -7, what: "ProvingKey::serialize", detail: format!("{:?}", e) })?;
    let mut h = std::ptr::null_mut();
    check(unsafe { swm_pk_deserialize(st.ctx, bytes.as_ptr(), bytes.len(), &mut h) }, "swm_pk_deserialize", st.ctx)?;
    check(unsafe { swm_pk_retain(h) }, "swm_pk_retain", st.ctx)?; // one reference for the cache, one for this call
    KEYS.lock().unwrap_or_else(|p| p.into_inner()).put(st.device, key, h);
    Ok(h)
}

/// Return the marlin proof for the given circuit/constraint system (src/marlin/mod.rs:70-77).
pub fn generate_proof(
    constraint_system: ConstraintSystemRef,
    proving_key: ProvingKey,
    rng: &mut StdRng,
) -> Result<MarlinProof> {
    prove(&proving_key, constraint_system, rng).map_err(|e| anyhow!("{:?}", e))
}

fn prove<R: RngCore + 'static>(pk: &ProvingKey, cs: ConstraintSystemRef, rng: &mut R) -> std::result::Result<MarlinProof, SwmError> {
    // assignment only: finalize() as ark-marlin's prover does, then no to_matrices() / CSR copies (the matrices are the key's)
    let packed = AssignmentOnly::from_cs(&cs).map_err(|e| SwmError { code: -1, what: "ConstraintSystemRef::borrow", detail: format!("{:?}", e) })?;
    with_state(|st| {
        let ctx = st.ctx;
        let pk_h = resident_pk(st, pk)?;
        let r1cs = packed.as_ffi();
        let mut buf = [0u8; 4096];
        let mut len = 0usize;
        // the proof comes back in the serialize_uncompressed form and is read with deserialize_unchecked: the bytes were
        // written by the library in this process a microsecond ago, so the square root and the subgroup check per commitment
        // of the checked compressed path (~2.3 ms per proof: `drop_in.checked_deserialize_proxy_ms` in the bench line) buy nothing.
        // `proof.serialize(..)` of the value built here gives the compressed bytes swm_generate_proof would have written.
        let rc = with_rng(rng, |r| {
            check(unsafe { swm_generate_proof_ex(ctx, pk_h, &r1cs, r, SWM_PROOF_UNCOMPRESSED, buf.as_mut_ptr(), buf.len(), &mut len) }, "swm_generate_proof_ex", ctx)
        });
        unsafe { swm_pk_destroy(ctx, pk_h) }; // this call's reference; the cache keeps the key resident
        rc?;
        MarlinProof::deserialize_unchecked(&mut &buf[..len]).map_err(|e| SwmError { code: -7, what: "Proof::deserialize_unchecked", detail: format!("{:?}", e) })
    })
}

/// src/marlin/mod.rs:79-86.  Host arithmetic (two pairings) in the library as in the reference; needs no GPU.
pub fn verify_proof(
    verifying_key: VerifyingKey,
    public_inputs: &[ConstraintF],
    proof: &MarlinProof,
    rng: &mut StdRng,
) -> Result<bool> {
    verify(&verifying_key, public_inputs, proof, rng).map_err(|e| anyhow!("{:?}", e))
}

fn verify<R: RngCore + 'static>(vk: &VerifyingKey, public_inputs: &[Fr], proof: &MarlinProof, rng: &mut R) -> std::result::Result<bool, SwmError> {
    let (_, vk_bytes) = vk_key(vk)?;
    let mut proof_bytes = Vec::new();
    proof.serialize(&mut proof_bytes).map_err(|e| SwmError { code: -7, what: "Proof::serialize", detail: format!("{:?}", e) })?;
    let inputs: Vec<u64> = public_inputs.iter().flat_map(|f| fr_limbs(f)).collect();
    let mut vk_h = std::ptr::null_mut();
    check(unsafe { swm_vk_deserialize(vk_bytes.as_ptr(), vk_bytes.len(), &mut vk_h) }, "swm_vk_deserialize", std::ptr::null_mut())?;
    let mut ok: c_int = 0;
    let out = with_rng(rng, |r| {
        check(
            unsafe {
                swm_verify_proof(vk_h, if inputs.is_empty() { std::ptr::null() } else { inputs.as_ptr() }, public_inputs.len(),
                                 proof_bytes.as_ptr(), proof_bytes.len(), r, &mut ok)
            },
            "swm_verify_proof",
            std::ptr::null_mut(),
        )
    });
    unsafe { swm_vk_destroy(vk_h) };
    out.map(|_| ok != 0)
}

// ------------------------------------------------------------------------------------------------ MarlinInst
/// `MarlinInst` as every in-tree caller of the reference uses it — `SimpleMerkleTree::{new, prove, verify}`
/// (/root/reference/src/merkle_tree/simple_merkle_tree.rs:39,83,119,148), `examples/manual-constraints.rs:89-99`,
/// `examples/merkle-tree/main.rs:212-257`, `examples/simple-payments/transaction.rs:96-125`, `examples/test-circuit.rs:74-80`,
/// `examples/schnorr-signature/main.rs:191-252` — is `ark_marlin::Marlin<Fr, MultiPC, FS>`: associated functions taking a
/// `ConstraintSynthesizer`.  This unit struct has the same associated functions with the same argument and `Result` types
/// (errors are `ark_marlin::Error<ark_poly_commit::Error>`, which those callers format with `{:?}` or `unwrap`), and sends
/// the work to the MI355X library: after `pub use swmarlin_sys::marlin::*` those call sites compile unchanged and no longer
/// run arkworks' CPU prover.
///
/// Synthesis follows ark-marlin 0.3 (`Marlin::index` / `Marlin::prove`): a fresh `ConstraintSystem`, optimisation goal
/// `Weight`, mode `Setup` for the indexer and `Prove { construct_matrices: true }` for the prover, `generate_constraints`,
/// then `finalize` + `to_matrices` (`PackedR1cs::from_cs`); padding to a square system happens inside the library.
pub struct MarlinInst;

/// Library status -> the variant arkworks would have returned where one exists, else a polynomial-commitment error that
/// carries the library's message (callers only ever format the error).
fn to_marlin_error(e: SwmError) -> MarlinError {
    match e.code {
        -6 => ark_marlin::Error::IndexTooLarge, // SWM_ERR_INDEX_TOO_LARGE
        _ => ark_marlin::Error::PolynomialCommitmentError(ark_poly_commit::Error::IncorrectInputLength(format!(
            "swmarlin {}: {} ({})",
            e.what, e.detail, e.code
        ))),
    }
}
fn synthesize<C: ConstraintSynthesizer<Fr>>(c: C, mode: SynthesisMode) -> std::result::Result<ConstraintSystemRef, MarlinError> {
    let cs = ConstraintSystem::<Fr>::new_ref();
    cs.set_optimization_goal(OptimizationGoal::Weight);
    cs.set_mode(mode);
    c.generate_constraints(cs.clone()).map_err(ark_marlin::Error::R1CSError)?;
    Ok(cs)
}

impl MarlinInst {
    /// `Marlin::universal_setup` (simple_merkle_tree.rs:39).
    pub fn universal_setup<R: RngCore + 'static>(
        num_constraints: usize,
        num_variables: usize,
        num_non_zero: usize,
        rng: &mut R,
    ) -> std::result::Result<UniversalSRS, MarlinError> {
        universal_setup(num_constraints, num_variables, num_non_zero, rng).map_err(to_marlin_error)
    }

    /// `Marlin::index` (simple_merkle_tree.rs:83): index the circuit `c` under `srs`.
    pub fn index<C: ConstraintSynthesizer<Fr>>(srs: &UniversalSRS, c: C) -> std::result::Result<(ProvingKey, VerifyingKey), MarlinError> {
        let cs = synthesize(c, SynthesisMode::Setup)?;
        index(srs, cs).map_err(to_marlin_error)
    }

    /// The fork's entry point for an already synthesised system (/root/reference/src/marlin/mod.rs:92).
    pub fn index_from_constraint_system(srs: &UniversalSRS, cs: ConstraintSystemRef) -> std::result::Result<(ProvingKey, VerifyingKey), MarlinError> {
        index(srs, cs).map_err(to_marlin_error)
    }

    /// `Marlin::prove` (simple_merkle_tree.rs:119).  An unsatisfied witness is an `Err` here (ark-marlin panics on a
    /// debug assertion, which the reference's `#[should_panic]` test reaches through `unwrap`: same outcome).
    pub fn prove<C: ConstraintSynthesizer<Fr>, R: RngCore + 'static>(index_pk: &ProvingKey, c: C, zk_rng: &mut R) -> std::result::Result<MarlinProof, MarlinError> {
        let cs = synthesize(c, SynthesisMode::Prove { construct_matrices: true })?;
        prove(index_pk, cs, zk_rng).map_err(to_marlin_error)
    }

    /// The fork's entry point for an already synthesised system (/root/reference/src/marlin/mod.rs:75).
    pub fn prove_from_constraint_system<R: RngCore + 'static>(index_pk: &ProvingKey, cs: ConstraintSystemRef, zk_rng: &mut R) -> std::result::Result<MarlinProof, MarlinError> {
        prove(index_pk, cs, zk_rng).map_err(to_marlin_error)
    }

    /// `Marlin::verify` (simple_merkle_tree.rs:148).
    pub fn verify<R: RngCore + 'static>(index_vk: &VerifyingKey, public_input: &[Fr], proof: &MarlinProof, rng: &mut R) -> std::result::Result<bool, MarlinError> {
        verify(index_vk, public_input, proof, rng).map_err(to_marlin_error)
    }
}

#[cfg(test)]
mod tests {
    //! The reference's own plumbing test (examples/manual-constraints.rs:86-100) against this module: needs an MI355X.
    use super::*;
    use ark_relations::{lc, r1cs::{ConstraintSynthesizer, ConstraintSystem, SynthesisError, Variable}};

    #[derive(Clone)]
    struct ManualConstraints { a: Fr, b: Fr }
    impl ConstraintSynthesizer<Fr> for ManualConstraints {
        fn generate_constraints(self, cs: ark_relations::r1cs::ConstraintSystemRef<Fr>) -> std::result::Result<(), SynthesisError> {
            let a = cs.new_input_variable(|| Ok(self.a))?;
            let b = cs.new_witness_variable(|| Ok(self.b))?;
            cs.enforce_constraint(lc!() + a - b, lc!() + Variable::One, lc!())
        }
    }

    /// examples/manual-constraints.rs:86-100 verbatim (MarlinInst with a ConstraintSynthesizer, `rng` reborrowed).
    #[test]
    fn manual_constraints_through_marlin_inst() {
        let rng = &mut ark_std::test_rng();
        let universal_srs = MarlinInst::universal_setup(100, 25, 300, rng).unwrap();
        let number = Fr::from(1u64);
        let circuit = ManualConstraints { a: number, b: number };
        let (index_pk, index_vk) = MarlinInst::index(&universal_srs, circuit.clone()).unwrap();
        let proof = MarlinInst::prove(&index_pk, circuit.clone(), rng).unwrap();
        assert!(MarlinInst::verify(&index_vk, &[number], &proof, rng).unwrap());
        // and the bytes are what arkworks' CPU prover emits for the same circuit, key and rng stream
        let rng2 = &mut ark_std::test_rng();
        let srs2 = ArkMarlinInst::universal_setup(100, 25, 300, rng2).unwrap();
        let (pk2, _vk2) = ArkMarlinInst::index(&srs2, circuit.clone()).unwrap();
        let proof2 = ArkMarlinInst::prove(&pk2, circuit, rng2).unwrap();
        let (mut b1, mut b2) = (Vec::new(), Vec::new());
        proof.serialize(&mut b1).unwrap();
        proof2.serialize(&mut b2).unwrap();
        assert_eq!(b1, b2);
    }

    #[test]
    fn manual_constraints_prove_and_verify() {
        let mut rng = generate_rand();
        let srs = generate_universal_srs(100, 25, 300, &mut rng).unwrap();
        let cs = ConstraintSystem::<Fr>::new_ref();
        ManualConstraints { a: Fr::from(1u64), b: Fr::from(1u64) }.generate_constraints(cs.clone()).unwrap();
        let (pk, vk) = generate_proving_and_verifying_keys(&srs, cs.clone()).unwrap();
        let proof = generate_proof(cs, pk, &mut rng).unwrap();
        assert!(verify_proof(vk, &[Fr::from(1u64)], &proof, &mut rng).unwrap());
    }
}
