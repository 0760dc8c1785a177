//! The native Pedersen hash and the Pedersen Merkle tree of simpleworks on the MI355X library: what
//! `MerkleTree::<MerkleConfig>::new(&leaf_crh_params, &two_to_one_crh_params, leaves)` computes on one CPU thread
//! (/root/reference/src/merkle_tree/simple_merkle_tree.rs:47-49: 2^19 - 1 Pedersen hashes at 2^18 leaves) and
//! `pedersen_hash` of /root/reference/src/hash/mod.rs:23-28.  The parameters stay the arkworks values (the circuit gadgets
//! need them as such); this module only evaluates the hash and the tree.
//!
//! EXPERIMENTAL like the rest of the crate: written without a Rust toolchain, never compiled.  The same entry points are
//! exercised through the Python mirror (simpleworks_amd/hash.py, tests/test_gpu_pedersen.py).
use crate::ffi::*;
use crate::marlin::with_context;
use anyhow::{anyhow, Result};
use ark_crypto_primitives::crh::pedersen;
use ark_ec::ProjectiveCurve;
use ark_ed_on_bls12_377::{EdwardsProjective, Fq};
use ark_ff::{FromBytes, ToBytes};

/// `pedersen::Parameters<EdwardsProjective>` resident on the GPU (a table of 2^window multiples per window).
pub struct GpuPedersen {
    handle: *mut swm_pedersen,
    pub num_windows: usize,
    pub window_size: usize,
}

impl GpuPedersen {
    /// `params.generators[w][j]` = 2^j g_w, handed over as affine (x, y): 2 x 32 little-endian bytes each.
    pub fn new(params: &pedersen::Parameters<EdwardsProjective>) -> Result<Self> {
        let num_windows = params.generators.len();
        let window_size = params.generators.first().map(|r| r.len()).unwrap_or(0);
        let mut raw = Vec::with_capacity(64 * num_windows * window_size);
        for row in &params.generators {
            if row.len() != window_size {
                return Err(anyhow!("ragged generator table"));
            }
            for g in row {
                let a = g.into_affine();
                a.x.write(&mut raw).map_err(|e| anyhow!("{:?}", e))?;
                a.y.write(&mut raw).map_err(|e| anyhow!("{:?}", e))?;
            }
        }
        let mut handle = std::ptr::null_mut();
        with_context(|ctx| unsafe { swm_pedersen_create(ctx, raw.as_ptr(), num_windows, window_size, &mut handle) }, "swm_pedersen_create")?;
        Ok(GpuPedersen { handle, num_windows, window_size })
    }

    /// `CRH::evaluate` + `TECompressor` for `inputs.len() / input_len` inputs of `input_len` bytes each, back to back.
    pub fn evaluate_many(&self, inputs: &[u8], input_len: usize) -> Result<Vec<Fq>> {
        if input_len == 0 || inputs.len() % input_len != 0 {
            return Err(anyhow!("inputs are not a whole number of {}-byte messages", input_len));
        }
        let count = inputs.len() / input_len;
        let mut digests = vec![0u8; 32 * count];
        with_context(|ctx| unsafe { swm_pedersen_hash(ctx, self.handle, inputs.as_ptr(), input_len, count, digests.as_mut_ptr()) }, "swm_pedersen_hash")?;
        digests.chunks(32).map(|c| Fq::read(c).map_err(|e| anyhow!("{:?}", e))).collect()
    }

    /// `CRH::evaluate(&params, input)`
    pub fn evaluate(&self, input: &[u8]) -> Result<Fq> {
        Ok(self.evaluate_many(input, input.len())?[0])
    }
}

impl Drop for GpuPedersen {
    fn drop(&mut self) {
        let h = self.handle;
        let _ = with_context(|ctx| { unsafe { swm_pedersen_destroy(ctx, h) }; 0 }, "swm_pedersen_destroy");
    }
}

/// All nodes of the tree, bottom level first: levels[0] = the n leaf digests, ..., levels.last() = [root].
/// `leaves`: n = 2^k (k >= 1) leaves of `leaf_len` bytes each (`to_bytes![leaf]`; 1 for u8 leaves), back to back.
pub struct GpuMerkleTree {
    pub levels: Vec<Vec<Fq>>,
}

impl GpuMerkleTree {
    pub fn new(leaf: &GpuPedersen, two_to_one: &GpuPedersen, leaves: &[u8], leaf_len: usize) -> Result<Self> {
        if leaf_len == 0 || leaves.len() % leaf_len != 0 {
            return Err(anyhow!("leaves are not a whole number of {}-byte items", leaf_len));
        }
        let n = leaves.len() / leaf_len;
        let mut nodes = vec![0u8; 32 * (2 * n - 1)];
        with_context(
            |ctx| unsafe { swm_merkle_tree_build(ctx, leaf.handle, two_to_one.handle, leaves.as_ptr(), leaf_len, n, nodes.as_mut_ptr()) },
            "swm_merkle_tree_build",
        )?;
        let mut levels = Vec::new();
        let (mut off, mut cnt) = (0usize, n);
        while cnt >= 1 {
            let lvl: Result<Vec<Fq>> =
                nodes[32 * off..32 * (off + cnt)].chunks(32).map(|c| Fq::read(c).map_err(|e| anyhow!("{:?}", e))).collect();
            levels.push(lvl?);
            off += cnt;
            cnt >>= 1;
        }
        Ok(GpuMerkleTree { levels })
    }

    pub fn root(&self) -> Fq {
        self.levels[self.levels.len() - 1][0]
    }

    /// The sibling digest at every level, leaf level first: what `Path::{leaf_sibling_hash, auth_path}` hold
    /// (ark-crypto-primitives stores the upper ones top-down: reverse `[1..]` for `auth_path`).
    pub fn siblings(&self, index: usize) -> Vec<Fq> {
        (0..self.levels.len() - 1).map(|l| self.levels[l][(index >> l) ^ 1]).collect()
    }
}
