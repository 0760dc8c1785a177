//! swmarlin-sys: `ffi` = the raw C ABI of include/swmarlin.h; `marlin` = safe functions with exactly the signatures of
//! simpleworks' `src/marlin/mod.rs` (/root/reference/src/marlin/mod.rs:12-94), implemented on the MI355X library.
//!
//! simpleworks' own `src/marlin/mod.rs` shrinks to
//! ```ignore
//! pub use swmarlin_sys::marlin::*;
//! pub mod serialization;            // unchanged: ark-serialize on arkworks types
//! ```
//! and every caller (`SimpleMerkleTree`, the examples, external VMs) compiles unchanged: the types are still the
//! arkworks types, the rng is still the caller's `&mut StdRng`, and `MarlinInst::{universal_setup, index, prove, verify}`
//! — what those callers actually invoke, with a `ConstraintSynthesizer` — is a unit struct of this crate that forwards to
//! the library (the reference's alias of arkworks' CPU prover is kept as `ArkMarlinInst`).
//!
//! EXPERIMENTAL: written without a Rust toolchain or the arkworks sources at hand; never compiled.  See INTEGRATION.md.
// (feature `pin`: the GPU-free pin kit, tests/pin_golden.rs — arkworks only, nothing of the library is compiled or linked)
#[cfg(not(feature = "pin"))]
pub mod ffi;
#[cfg(not(feature = "pin"))]
pub mod marlin;
#[cfg(not(feature = "pin"))]
pub mod merkle;
#[cfg(not(feature = "pin"))]
mod convert;
/// SWMR1CS1 dumps of synthesised constraint systems (`dump_r1cs(&cs, path)`): arkworks only, available with and without `pin`
pub mod r1cs_dump;
