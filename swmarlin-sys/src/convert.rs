//! arkworks values <-> the packed limb buffers of the C ABI.  No `#[repr(C)]` exists on arkworks types, so everything is
//! copied; nothing aliases a Rust struct.  Field elements travel as their Montgomery limbs (`Fp256(BigInteger256([u64; 4]))`
//! holds exactly the bytes the library uses), points as affine Montgomery coordinates with all-zero for infinity.
use crate::ffi;
use ark_bls12_377::{Fq, Fq2, Fr, G1Affine, G2Affine};
use ark_ff::{BigInteger256, BigInteger384, Fp256, Fp384, Zero};
use ark_relations::r1cs::{ConstraintSystemRef, Matrix, SynthesisError};

pub fn fr_limbs(f: &Fr) -> [u64; 4] {
    (f.0).0
}
pub fn fr_from_limbs(l: [u64; 4]) -> Fr {
    Fp256::new(BigInteger256(l)) // `new` takes the Montgomery representation as is
}
pub fn fq_from_limbs(l: &[u64]) -> Fq {
    let mut a = [0u64; 6];
    a.copy_from_slice(&l[..6]);
    Fp384::new(BigInteger384(a))
}
pub fn g1_from_limbs(l: &[u64]) -> G1Affine {
    if l[..12].iter().all(|w| *w == 0) {
        return G1Affine::zero();
    }
    G1Affine::new(fq_from_limbs(&l[0..6]), fq_from_limbs(&l[6..12]), false)
}
pub fn g1_limbs(p: &G1Affine, out: &mut [u64]) {
    if p.infinity {
        out[..12].iter_mut().for_each(|w| *w = 0);
    } else {
        out[0..6].copy_from_slice(&(p.x.0).0);
        out[6..12].copy_from_slice(&(p.y.0).0);
    }
}
pub fn g2_from_limbs(l: &[u64]) -> G2Affine {
    if l[..24].iter().all(|w| *w == 0) {
        return G2Affine::zero();
    }
    let x = Fq2::new(fq_from_limbs(&l[0..6]), fq_from_limbs(&l[6..12]));
    let y = Fq2::new(fq_from_limbs(&l[12..18]), fq_from_limbs(&l[18..24]));
    G2Affine::new(x, y, false)
}
pub fn g2_limbs(p: &G2Affine, out: &mut [u64]) {
    if p.infinity {
        out[..24].iter_mut().for_each(|w| *w = 0);
    } else {
        out[0..6].copy_from_slice(&(p.x.c0.0).0);
        out[6..12].copy_from_slice(&(p.x.c1.0).0);
        out[12..18].copy_from_slice(&(p.y.c0.0).0);
        out[18..24].copy_from_slice(&(p.y.c1.0).0);
    }
}

/// `cs.to_matrices()` + the assignments, flattened to what `struct swm_r1cs` points at.  The buffers live as long as
/// this value; `as_ffi` hands out pointers into them for the duration of one call.  What the INDEXER needs.
pub struct PackedR1cs {
    instance: Vec<u64>,
    witness: Vec<u64>,
    mats: [(Vec<u32>, Vec<u32>, Vec<u64>); 3],
    num_constraints: usize,
}

/// What the PROVER needs from a live constraint system: the two assignment vectors and the shape — nothing else
/// (`swm_generate_proof` reads num_instance, num_witness, num_constraints, instance, witness; the matrices are the key's, as
/// in ark-marlin, whose `prover_init` never calls `to_matrices()` either).  No `to_matrices()` / CSR copies
/// per proof: at 2^20 constraints those were 3 x 2^20 `Vec` clones plus ~100 MB of copies on one host thread, more than
/// the 49-ms GPU proof they fed.
///
/// The assignment itself is handed over WHERE IT IS when `Vec<Fr>` is what it looks like — `Fp256<P>(BigInteger256([u64; 4]),
/// PhantomData)`: 32 bytes, 8-aligned, the four Montgomery limbs in order — which `fr_view_is_sound` checks once per
/// process on known values; otherwise the vectors are flattened (a 32-MB copy at 2^20: ~3 ms).  The borrow of the
/// `ConstraintSystem` is held for as long as this value lives, so the vectors cannot move under the pointers.
pub struct AssignmentOnly<'a> {
    borrow: std::cell::Ref<'a, ark_relations::r1cs::ConstraintSystem<Fr>>,
    copies: Option<(Vec<u64>, Vec<u64>)>,
    num_constraints: usize,
}

/// `size_of::<Fr>() == 32`, `align_of::<Fr>() == 8`, and the bytes of an `Fr` ARE its Montgomery limbs in order.
pub fn fr_view_is_sound() -> bool {
    use std::sync::OnceLock;
    static OK: OnceLock<bool> = OnceLock::new();
    *OK.get_or_init(|| {
        if std::mem::size_of::<Fr>() != 32 || std::mem::align_of::<Fr>() != std::mem::align_of::<u64>() {
            return false;
        }
        let probe: Vec<Fr> = vec![Fr::from(1u64), Fr::from(0x1234_5678_9abc_def0u64), -Fr::from(7u64)];
        let view = unsafe { std::slice::from_raw_parts(probe.as_ptr() as *const u64, 4 * probe.len()) };
        probe.iter().enumerate().all(|(i, f)| view[4 * i..4 * i + 4] == fr_limbs(f))
    })
}

impl<'a> AssignmentOnly<'a> {
    pub fn from_cs(cs: &'a ConstraintSystemRef<Fr>) -> Result<Self, SynthesisError> {
        // `finalize()` stays: with OptimizationGoal::Weight it OUTLINES linear combinations into new witness variables and
        // constraints — the system the key was indexed from went through it (`PackedR1cs::from_cs`), so the assignment and the
        // constraint count have to as well (ark-marlin's own prover calls it at the same point).  It is part of synthesis, not of
        // what this crate adds; what is gone is `to_matrices()` and the three CSR copies.
        cs.finalize();
        let num_constraints = cs.num_constraints();
        let borrow = cs.borrow().ok_or(SynthesisError::MissingCS)?;
        let copies = if fr_view_is_sound() {
            None
        } else {
            let flat = |v: &[Fr]| -> Vec<u64> { v.iter().flat_map(|f| fr_limbs(f)).collect() };
            Some((flat(&borrow.instance_assignment), flat(&borrow.witness_assignment)))
        };
        Ok(AssignmentOnly { borrow, copies, num_constraints })
    }

    /// `struct swm_r1cs` with the nine matrix pointers NULL (include/swmarlin.h: "ASSIGNMENT ONLY").
    pub fn as_ffi(&self) -> ffi::swm_r1cs {
        let (ni, nw) = (self.borrow.instance_assignment.len(), self.borrow.witness_assignment.len());
        let (instance, witness) = match &self.copies {
            Some((i, w)) => (i.as_ptr(), if w.is_empty() { std::ptr::null() } else { w.as_ptr() }),
            None => (
                self.borrow.instance_assignment.as_ptr() as *const u64,
                if nw == 0 { std::ptr::null() } else { self.borrow.witness_assignment.as_ptr() as *const u64 },
            ),
        };
        let z32 = std::ptr::null::<u32>();
        let z64 = std::ptr::null::<u64>();
        ffi::swm_r1cs {
            num_instance: ni,
            num_witness: nw,
            num_constraints: self.num_constraints,
            instance,
            witness,
            a_rowptr: z32, a_col: z32, a_val: z64,
            b_rowptr: z32, b_col: z32, b_val: z64,
            c_rowptr: z32, c_col: z32, c_val: z64,
        }
    }
}

fn csr(m: &Matrix<Fr>) -> (Vec<u32>, Vec<u32>, Vec<u64>) {
    let nnz: usize = m.iter().map(|r| r.len()).sum();
    let (mut rowptr, mut col, mut val) = (Vec::with_capacity(m.len() + 1), Vec::with_capacity(nnz), Vec::with_capacity(4 * nnz));
    rowptr.push(0u32);
    for row in m {
        for (coeff, j) in row {
            col.push(*j as u32);
            val.extend_from_slice(&fr_limbs(coeff));
        }
        rowptr.push(col.len() as u32);
    }
    (rowptr, col, val)
}

impl PackedR1cs {
    /// What ark-marlin's `prove_from_constraint_system` / `index_from_constraint_system` (the fork the reference pins,
    /// /root/reference/Cargo.toml:30) read from a live constraint system: finalise (inline the linear combinations), take
    /// the matrices and the two assignment vectors.  Padding and squaring happen inside the library, as in ark-marlin.
    pub fn from_cs(cs: &ConstraintSystemRef<Fr>) -> Result<Self, SynthesisError> {
        cs.finalize();
        let m = cs.to_matrices().ok_or(SynthesisError::MissingCS)?;
        let b = cs.borrow().ok_or(SynthesisError::MissingCS)?;
        let flat = |v: &[Fr]| -> Vec<u64> { v.iter().flat_map(|f| fr_limbs(f)).collect() };
        Ok(PackedR1cs {
            instance: flat(&b.instance_assignment),
            witness: flat(&b.witness_assignment),
            mats: [csr(&m.a), csr(&m.b), csr(&m.c)],
            num_constraints: m.num_constraints,
        })
    }

    pub fn as_ffi(&self) -> ffi::swm_r1cs {
        let p32 = |v: &Vec<u32>| if v.is_empty() { std::ptr::null() } else { v.as_ptr() };
        let p64 = |v: &Vec<u64>| if v.is_empty() { std::ptr::null() } else { v.as_ptr() };
        ffi::swm_r1cs {
            num_instance: self.instance.len() / 4,
            num_witness: self.witness.len() / 4,
            num_constraints: self.num_constraints,
            instance: self.instance.as_ptr(),
            witness: p64(&self.witness),
            a_rowptr: self.mats[0].0.as_ptr(), a_col: p32(&self.mats[0].1), a_val: p64(&self.mats[0].2),
            b_rowptr: self.mats[1].0.as_ptr(), b_col: p32(&self.mats[1].1), b_val: p64(&self.mats[1].2),
            c_rowptr: self.mats[2].0.as_ptr(), c_col: p32(&self.mats[2].1), c_val: p64(&self.mats[2].2),
        }
    }
}
