//! SWMR1CS1: a synthesised constraint system as ONE flat little-endian file — how a circuit that only exists on the Rust side
//! (e.g. `MerkleTreeVerificationU8`, /root/reference/src/merkle_tree/merkle_tree_verification_u8.rs:25-58, whose constraint
//! layout comes out of ark-r1cs-std) reaches the MI355X library's Python harness and `bench.py --r1cs FILE` on a box without a
//! Rust toolchain, and how the pin kit (tests/pin_golden.rs) replays the larger golden circuits of the library's repository.
//! The reader on the other side is `simpleworks_amd/workloads.py::load_r1cs`; the layout is documented there and in
//! INTEGRATION.md:
//!
//! ```text
//!   0   8  magic "SWMR1CS1"
//!   8   8  num_instance (instance assignment, the leading one included)      u64
//!  16   8  num_witness                                                        u64
//!  24   8  num_constraints                                                    u64
//!  32  24  non-zeros of A, B, C                                               3 x u64
//!  56   8  flags: bit 0 = field elements are Montgomery limbs (R = 2^256); always 1
//!  64      instance (num_instance x 32 B) | witness (num_witness x 32 B)
//!          per matrix A, B, C: rowptr (num_constraints + 1) x u32 | col nnz x u32 | zero padding to 8 | val nnz x 32 B
//!  end 32  BLAKE2s-256 of everything before it
//! ```
//! Columns follow ark-relations' `Matrix`: instance variables first (column 0 is the constant one), then witness variables.
//! Compiled with and without the `pin` feature: arkworks only, nothing of the library.
//!
//! EXPERIMENTAL like the rest of this crate: written without a compiler at hand.
use ark_bls12_377::Fr;
use ark_ff::{BigInteger256, Fp256};
use ark_relations::r1cs::{ConstraintSynthesizer, ConstraintSystemRef, LinearCombination, Matrix, SynthesisError, Variable};
use blake2::Blake2s;
use digest::Digest;
use std::io;
use std::path::Path;

pub const MAGIC: &[u8; 8] = b"SWMR1CS1";

/// `cs.to_matrices()` and the two assignment vectors of a finalised constraint system.
#[derive(Clone)]
pub struct R1csFile {
    /// instance assignment, the leading one included
    pub instance: Vec<Fr>,
    pub witness: Vec<Fr>,
    /// A, B, C as ark-relations lays them out: per row (coefficient, column)
    pub mats: [Matrix<Fr>; 3],
}

fn bad(msg: &str) -> io::Error {
    io::Error::new(io::ErrorKind::InvalidData, format!("SWMR1CS1: {}", msg))
}
fn put_fr(out: &mut Vec<u8>, f: &Fr) {
    for w in (f.0).0.iter() {
        out.extend_from_slice(&w.to_le_bytes());
    }
}
fn get_u64(b: &[u8], at: usize) -> u64 {
    let mut w = [0u8; 8];
    w.copy_from_slice(&b[at..at + 8]);
    u64::from_le_bytes(w)
}
fn get_u32(b: &[u8], at: usize) -> u32 {
    let mut w = [0u8; 4];
    w.copy_from_slice(&b[at..at + 4]);
    u32::from_le_bytes(w)
}
fn get_fr(b: &[u8], at: usize) -> Fr {
    let l = [get_u64(b, at), get_u64(b, at + 8), get_u64(b, at + 16), get_u64(b, at + 24)];
    Fp256::new(BigInteger256(l)) // `new` takes the Montgomery representation as is
}

impl R1csFile {
    /// What ark-marlin's `*_from_constraint_system` entry points read from a live system (the fork the reference pins,
    /// /root/reference/Cargo.toml:30): finalise — inline the linear combinations —, take the matrices and the assignments.
    pub fn from_cs(cs: &ConstraintSystemRef<Fr>) -> Result<Self, SynthesisError> {
        cs.finalize();
        let m = cs.to_matrices().ok_or(SynthesisError::MissingCS)?;
        let b = cs.borrow().ok_or(SynthesisError::MissingCS)?;
        Ok(R1csFile { instance: b.instance_assignment.clone(), witness: b.witness_assignment.clone(), mats: [m.a, m.b, m.c] })
    }

    pub fn num_constraints(&self) -> usize {
        self.mats[0].len()
    }

    pub fn to_bytes(&self) -> Vec<u8> {
        let nnz = |m: &Matrix<Fr>| m.iter().map(|r| r.len()).sum::<usize>();
        let mut out = Vec::new();
        out.extend_from_slice(MAGIC);
        for v in [self.instance.len(), self.witness.len(), self.num_constraints(), nnz(&self.mats[0]), nnz(&self.mats[1]), nnz(&self.mats[2]), 1usize].iter() {
            out.extend_from_slice(&(*v as u64).to_le_bytes());
        }
        for f in self.instance.iter().chain(self.witness.iter()) {
            put_fr(&mut out, f);
        }
        for m in self.mats.iter() {
            let mut run = 0u32;
            out.extend_from_slice(&run.to_le_bytes());
            for row in m {
                run += row.len() as u32;
                out.extend_from_slice(&run.to_le_bytes());
            }
            for row in m {
                for (_, j) in row {
                    out.extend_from_slice(&(*j as u32).to_le_bytes());
                }
            }
            while out.len() % 8 != 0 {
                out.push(0);
            }
            for row in m {
                for (c, _) in row {
                    put_fr(&mut out, c);
                }
            }
        }
        let h = Blake2s::digest(&out);
        out.extend_from_slice(&h);
        out
    }

    pub fn write(&self, path: impl AsRef<Path>) -> io::Result<()> {
        std::fs::write(path, self.to_bytes())
    }

    pub fn from_bytes(data: &[u8]) -> io::Result<Self> {
        if data.len() < 64 + 32 || &data[..8] != MAGIC {
            return Err(bad("not an SWMR1CS1 file"));
        }
        let body = &data[..data.len() - 32];
        if Blake2s::digest(body).as_slice() != &data[data.len() - 32..] {
            return Err(bad("checksum mismatch (truncated or corrupted file)"));
        }
        let (ninst, nwit, nrows) = (get_u64(data, 8) as usize, get_u64(data, 16) as usize, get_u64(data, 24) as usize);
        let nnz = [get_u64(data, 32) as usize, get_u64(data, 40) as usize, get_u64(data, 48) as usize];
        if get_u64(data, 56) != 1 || ninst < 1 || [ninst, nwit, nrows, nnz[0], nnz[1], nnz[2]].iter().any(|v| *v >= 1 << 31) {
            return Err(bad("implausible header"));
        }
        let mut want = 64 + 32 * (ninst + nwit);
        for k in nnz.iter() {
            let idx = 4 * (nrows + 1 + k);
            want += idx + (8 - idx % 8) % 8 + 32 * k;
        }
        if want != body.len() {
            return Err(bad("the header does not describe the file's length"));
        }
        let mut off = 64;
        let mut take_frs = |n: usize| {
            let v: Vec<Fr> = (0..n).map(|i| get_fr(data, off + 32 * i)).collect();
            off += 32 * n;
            v
        };
        let instance = take_frs(ninst);
        let witness = take_frs(nwit);
        let mut mats: Vec<Matrix<Fr>> = Vec::new();
        for k in nnz.iter() {
            let rowptr: Vec<usize> = (0..=nrows).map(|i| get_u32(data, off + 4 * i) as usize).collect();
            let cols_at = off + 4 * (nrows + 1);
            let idx = 4 * (nrows + 1 + k);
            let vals_at = off + idx + (8 - idx % 8) % 8;
            if rowptr[0] != 0 || rowptr[nrows] != *k || rowptr.windows(2).any(|w| w[0] > w[1]) {
                return Err(bad("row pointers are not a monotone prefix of the non-zeros"));
            }
            let mut m: Matrix<Fr> = Vec::with_capacity(nrows);
            for r in 0..nrows {
                let mut row = Vec::with_capacity(rowptr[r + 1] - rowptr[r]);
                for e in rowptr[r]..rowptr[r + 1] {
                    let col = get_u32(data, cols_at + 4 * e) as usize;
                    if col >= ninst + nwit {
                        return Err(bad("column index beyond the variables"));
                    }
                    row.push((get_fr(data, vals_at + 32 * e), col));
                }
                m.push(row);
            }
            mats.push(m);
            off = vals_at + 32 * k;
        }
        let c = mats.pop().unwrap();
        let b = mats.pop().unwrap();
        let a = mats.pop().unwrap();
        Ok(R1csFile { instance, witness, mats: [a, b, c] })
    }

    pub fn read(path: impl AsRef<Path>) -> io::Result<Self> {
        Self::from_bytes(&std::fs::read(path)?)
    }

    /// the public inputs as `verify` takes them: the instance assignment without its leading one
    pub fn public_inputs(&self) -> Vec<Fr> {
        self.instance[1..].to_vec()
    }
}

/// `dump_r1cs(&cs, "merkle_h19.r1cs")` after the circuit's `generate_constraints(cs.clone())`
pub fn dump_r1cs(cs: &ConstraintSystemRef<Fr>, path: impl AsRef<Path>) -> io::Result<()> {
    R1csFile::from_cs(cs).map_err(|e| bad(&format!("{:?}", e)))?.write(path)
}

/// Replays the file into a constraint system: the variables in file order, one `enforce_constraint` per row with the row's
/// terms as they are stored.  (`to_matrices` of the replayed system returns the stored matrices again: ark-relations keeps a
/// linear combination sorted by variable, which the stored rows already are.)
impl ConstraintSynthesizer<Fr> for R1csFile {
    fn generate_constraints(self, cs: ConstraintSystemRef<Fr>) -> Result<(), SynthesisError> {
        let mut vars = vec![Variable::One];
        for v in self.instance.iter().skip(1) {
            let v = *v;
            vars.push(cs.new_input_variable(|| Ok(v))?);
        }
        for v in self.witness.iter() {
            let v = *v;
            vars.push(cs.new_witness_variable(|| Ok(v))?);
        }
        let lc = |row: &Vec<(Fr, usize)>| {
            let mut l = LinearCombination::<Fr>::zero();
            for (c, j) in row {
                l = l + (*c, vars[*j]);
            }
            l
        };
        for r in 0..self.num_constraints() {
            cs.enforce_constraint(lc(&self.mats[0][r]), lc(&self.mats[1][r]), lc(&self.mats[2][r]))?;
        }
        Ok(())
    }
}
