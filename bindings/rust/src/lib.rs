//! Reference-side binding of `libimt_hip.so` (the MI355X indexed-Merkle-tree hot path).
//!
//! * [`IndexedMerkleTree`] and [`IndexedMerkleTreeLeaf`] have the EXACT signatures of the reference's
//!   `src/utils.rs:5-108`: `new(hash: &'a mut Poseidon<F, T, RATE>, leaves: Vec<F>) -> Result<_, &'static str>`,
//!   `get_root`, `get_proof`, `verify_proof`, same error strings, same `(proof, proof_helper)` convention
//!   (helper = 1 iff the node is a left child, `:79`).  The reference's own callers
//!   (`src/indexed_merkle_tree.rs:377-425, 417, 519, 706-735`) compile against it unchanged: replace
//!   `use crate::utils::{IndexedMerkleTree, IndexedMerkleTreeLeaf as IMTLeaf}` by `use imt_hip::{..}`.
//!   The struct deliberately has NO `Drop` impl and owns no device handle: those callers re-borrow the hasher
//!   (`native_hasher.update(..)`, `:407`) between the last use of one tree and the assignment of the next
//!   (`:417`), which only type-checks when dropping the tree does not touch its `&'a mut` field.
//! * [`gpu`] — batched entry points over the process-wide GPU context (hashes, path roots, non-membership,
//!   insert-witness checks, the stateful depth-32 [`gpu::IndexedTree`]).
//! * [`sliced`] — several GPUs, the single sorted list: the time-slice schedule and raw wrappers of `imt_itree_slice_*`
//!   for a host that brings its own RCCL calls.
//! * [`chip`] — circuit side: `TracedPoseidonHasher::hash_fix_len_array` assigns the GPU's witness trace instead of
//!   recomputing Poseidon, and `IndexedMerkleTreeChip` is the insert / update / non-membership sugar `north_star`
//!   names (the reference itself only has the free functions `insert_leaf` / `verify_non_inclusion`).
//!
//! NOT COMPILED in the repository this file ships in (no Rust toolchain in its build image); the FFI block is
//! checked mechanically against `include/imt.h` by `tests/test_rust_binding.py`.

pub mod chip;
pub mod ffi;
pub mod gpu;
pub mod sliced;

use halo2_base::utils::ScalarField;
use pse_poseidon::Poseidon;
use serde::{Deserialize, Serialize};

/// `src/utils.rs:5-10`.  `hash` is held (and untouched) so that the type, its lifetime and its constructor are
/// the reference's; the Poseidon tables the GPU uses are the same `Poseidon::<F, 3, 2>::new(8, 57)` instance
/// (`src/indexed_merkle_tree.rs:370`), built once per process in [`gpu::context`].
#[derive(Debug)]
pub struct IndexedMerkleTree<'a, F: ScalarField, const T: usize, const RATE: usize> {
    hash: &'a mut Poseidon<F, T, RATE>,
    tree: Vec<Vec<F>>,
    root: F,
}

/// `src/utils.rs:12-17`: the serde leaf record; also the snapshot format of [`gpu::IndexedTree`].
#[derive(Clone, Debug, Serialize, Deserialize)]
pub struct IndexedMerkleTreeLeaf<F: ScalarField> {
    pub val: F,
    pub next_val: F,
    pub next_idx: F,
}

impl<'a, F: ScalarField, const T: usize, const RATE: usize> IndexedMerkleTree<'a, F, T, RATE> {
    /// `src/utils.rs:20-57`: all levels are hashed on the GPU in one call (`imt_tree_build`) and kept on the host
    /// like the reference's `Vec<Vec<F>>`.
    pub fn new(
        hash: &'a mut Poseidon<F, T, RATE>,
        leaves: Vec<F>,
    ) -> Result<IndexedMerkleTree<'a, F, T, RATE>, &'static str> {
        if leaves.is_empty() {
            return Err("Cannot create Merkle Tree with no leaves");
        }
        if leaves.len() == 1 {
            return Ok(IndexedMerkleTree { hash, tree: vec![leaves.clone()], root: leaves[0] });
        }
        if leaves.len() % 2 == 1 {
            return Err("Leaves must be even");
        }
        // an even length that is not a power of two panics in the reference (index out of bounds at :45)
        assert!(leaves.len().is_power_of_two(), "index out of bounds: leaf count is not a power of two");
        gpu::assert_supported::<F, T, RATE>();
        let n = leaves.len();
        let flat = gpu::tree_build(&leaves).expect("imt_tree_build");
        let mut tree = Vec::new();
        let (mut off, mut len) = (0usize, n);
        loop {
            tree.push(flat[off..off + len].to_vec());
            if len == 1 {
                break;
            }
            off += len;
            len /= 2;
        }
        let root = tree.last().unwrap()[0];
        Ok(IndexedMerkleTree { hash, tree, root })
    }

    /// `src/utils.rs:59-61`
    pub fn get_root(&self) -> F {
        self.root
    }

    /// `src/utils.rs:63-85`
    pub fn get_proof(&self, index: usize) -> (Vec<F>, Vec<F>) {
        let mut proof = Vec::new();
        let mut proof_helper = Vec::new();
        let mut current_index = index;
        for level in &self.tree[..self.tree.len() - 1] {
            let is_left_node = current_index % 2 == 0;
            let sibling_index = if is_left_node { current_index + 1 } else { current_index - 1 };
            proof.push(level[sibling_index]);
            proof_helper.push(if is_left_node { F::from(1) } else { F::from(0) });
            current_index /= 2;
        }
        (proof, proof_helper)
    }

    /// `src/utils.rs:87-107`: the path is recomputed on the GPU (`imt_verify_proof_batch`, one item).  For many
    /// proofs use [`gpu::verify_proofs`], which sends them in one launch.
    pub fn verify_proof(&mut self, leaf: &F, index: usize, root: &F, proof: &[F]) -> bool {
        let _ = &self.hash; // same exclusive borrow as the reference's `self.hash.update(..)`
        gpu::verify_proofs(&[*leaf], &[index as u64], root, proof, proof.len())[0]
    }
}
