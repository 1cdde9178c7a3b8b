//! Safe wrappers over the C ABI on a process-wide context.  Field elements cross as canonical 32-byte
//! little-endian (`PrimeField::to_repr`); for `halo2curves::bn256::Fr` the zero-copy variants pass the slice itself
//! with `IMT_FMT_MONT256` (`Fr` is `#[repr(transparent)]` over its Montgomery `[u64; 4]`).

use crate::ffi::*;
use halo2_base::utils::ScalarField;
use std::ffi::CStr;
use std::os::raw::c_void;
use std::sync::{Mutex, OnceLock};

/// bn256::Fr, the only field the library implements (`src/indexed_merkle_tree.rs:383`)
const BN256_FR_MODULUS: &str = "0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001";

pub struct Gpu {
    pub ctx: *mut imt_ctx,
}
unsafe impl Send for Gpu {}

static GPU: OnceLock<Mutex<Gpu>> = OnceLock::new();

/// The process-wide context on device `IMT_HIP_DEVICE` (default 0).  An `imt_ctx` belongs to one thread at a
/// time (imt.h), hence the mutex.
pub fn context() -> &'static Mutex<Gpu> {
    GPU.get_or_init(|| {
        let device: i32 = std::env::var("IMT_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { imt_ctx_create(device, &mut ctx) };
        assert_eq!(rc, IMT_OK, "imt_ctx_create({device}) failed with {rc}: no usable MI355X (there is no CPU fallback)");
        Mutex::new(Gpu { ctx })
    })
}

pub fn assert_supported<F: ScalarField, const T: usize, const RATE: usize>() {
    assert!(T == 3 && RATE == 2, "libimt_hip implements Poseidon T = 3, RATE = 2, R_F = 8, R_P = 57 only");
    assert_eq!(F::MODULUS.to_lowercase(), BN256_FR_MODULUS, "libimt_hip implements bn256::Fr only");
}

#[derive(Debug)]
pub struct ImtError {
    pub code: i32,
    pub message: String,
}
fn check(g: &Gpu, rc: i32) -> Result<(), ImtError> {
    if rc == IMT_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(imt_last_error(g.ctx)) }.to_string_lossy().into_owned();
    Err(ImtError { code: rc, message })
}

pub fn to_bytes<F: ScalarField>(v: &[F]) -> Vec<u8> {
    let mut out = Vec::with_capacity(v.len() * 32);
    for x in v {
        out.extend_from_slice(&x.to_bytes_le()[..32]);
    }
    out
}
pub fn from_bytes<F: ScalarField>(b: &[u8]) -> Vec<F> {
    b.chunks_exact(32).map(|c| F::from_bytes_le(c)).collect()
}

/// `hash.update(&[a, b]); hash.squeeze_and_reset()` for n pairs (`src/utils.rs:46-47`)
pub fn hash2_batch<F: ScalarField>(pairs: &[[F; 2]]) -> Result<Vec<F>, ImtError> {
    let g = context().lock().unwrap();
    let input: Vec<u8> = pairs.iter().flat_map(|p| to_bytes(p)).collect();
    let mut out = vec![0u8; pairs.len() * 32];
    check(&g, unsafe {
        imt_hash2_batch(g.ctx, input.as_ptr() as *const c_void, out.as_mut_ptr() as *mut c_void, pairs.len(), IMT_FMT_CANONICAL)
    })?;
    Ok(from_bytes(&out))
}

/// leaf hash `[val, next_val, next_idx]` (`src/indexed_merkle_tree.rs:663-668`)
pub fn hash3_batch<F: ScalarField>(triples: &[[F; 3]]) -> Result<Vec<F>, ImtError> {
    let g = context().lock().unwrap();
    let input: Vec<u8> = triples.iter().flat_map(|p| to_bytes(p)).collect();
    let mut out = vec![0u8; triples.len() * 32];
    check(&g, unsafe {
        imt_hash3_batch(g.ctx, input.as_ptr() as *const c_void, out.as_mut_ptr() as *mut c_void, triples.len(), IMT_FMT_CANONICAL)
    })?;
    Ok(from_bytes(&out))
}

/// all levels of the dense tree, bottom-up, concatenated (2n - 1 elements)
pub fn tree_build<F: ScalarField>(leaves: &[F]) -> Result<Vec<F>, ImtError> {
    let g = context().lock().unwrap();
    let input = to_bytes(leaves);
    let mut levels = vec![0u8; (2 * leaves.len() - 1) * 32];
    let mut root = [0u8; 32];
    check(&g, unsafe {
        imt_tree_build(g.ctx, input.as_ptr() as *const c_void, leaves.len(), levels.as_mut_ptr() as *mut c_void,
                       root.as_mut_ptr() as *mut c_void, IMT_FMT_CANONICAL)
    })?;
    Ok(from_bytes(&levels))
}

/// `verify_proof` for n (leaf, index) pairs that share one root; `proofs` is item-major `[n][depth]`
pub fn verify_proofs<F: ScalarField>(leaves: &[F], index: &[u64], root: &F, proofs: &[F], depth: usize) -> Vec<bool> {
    let g = context().lock().unwrap();
    let n = leaves.len();
    assert!(index.len() == n && proofs.len() == n * depth);
    let (l, r, p) = (to_bytes(leaves), to_bytes(std::slice::from_ref(root)), to_bytes(proofs));
    let mut ok = vec![0u8; n];
    check(&g, unsafe {
        imt_verify_proof_batch(g.ctx, l.as_ptr() as *const c_void, index.as_ptr(), r.as_ptr() as *const c_void,
                               p.as_ptr() as *const c_void, depth as u32, n, ok.as_mut_ptr(), IMT_FMT_CANONICAL | IMT_SIB_ITEM_MAJOR)
    })
    .expect("imt_verify_proof_batch");
    ok.into_iter().map(|b| b != 0).collect()
}

/// Everything `insert_leaf` takes for one insertion (`src/indexed_merkle_tree.rs:231-245`), as field values.
#[derive(Clone, Debug)]
pub struct InsertWitness<F> {
    pub old_root: F,
    pub low_leaf: [F; 3],
    pub low_leaf_proof: Vec<F>,
    pub low_leaf_proof_helper: Vec<F>,
    pub new_root: F,
    pub interim_root: F,
    pub new_leaf: [F; 3],
    pub new_leaf_index: u64,
    pub new_leaf_proof: Vec<F>,
    pub new_leaf_proof_helper: Vec<F>,
    pub is_new_leaf_largest: bool,
}

/// Stateful depth-`depth` indexed tree on the GPU: `update_idx_leaf` + rebuild + `get_proof`, batched
/// (`src/indexed_merkle_tree.rs:632-671, :715-735`).
pub struct IndexedTree {
    handle: *mut imt_itree,
    pub depth: usize,
}
unsafe impl Send for IndexedTree {}

impl IndexedTree {
    pub fn new(depth: usize, capacity: u64) -> Result<Self, ImtError> {
        let g = context().lock().unwrap();
        let mut handle = std::ptr::null_mut();
        check(&g, unsafe { imt_itree_new(g.ctx, depth as u32, capacity, &mut handle) })?;
        Ok(IndexedTree { handle, depth })
    }
    pub fn size(&self) -> u64 {
        unsafe { imt_itree_size(self.handle) }
    }
    pub fn root<F: ScalarField>(&self) -> Result<F, ImtError> {
        let g = context().lock().unwrap();
        let mut r = [0u8; 32];
        check(&g, unsafe { imt_itree_root(self.handle, r.as_mut_ptr() as *mut c_void, IMT_FMT_CANONICAL) })?;
        Ok(F::from_bytes_le(&r))
    }
    /// n sequential insertions; one [`InsertWitness`] each (66 hashes per insertion at depth 32, all on the GPU)
    pub fn insert_batch<F: ScalarField>(&mut self, vals: &[F]) -> Result<Vec<InsertWitness<F>>, ImtError> {
        let g = context().lock().unwrap();
        let (n, d) = (vals.len(), self.depth);
        let first = self.size();
        let v = to_bytes(vals);
        let mut low_index = vec![0u64; n];
        let mut is_largest = vec![0u8; n];
        let (mut low_leaf, mut new_leaf) = (vec![0u8; n * 96], vec![0u8; n * 96]);
        let (mut old_root, mut interim_root, mut new_root) = (vec![0u8; n * 32], vec![0u8; n * 32], vec![0u8; n * 32]);
        let (mut low_sib, mut new_sib) = (vec![0u8; n * d * 32], vec![0u8; n * d * 32]);
        let out = imt_insert_out {
            low_index: low_index.as_mut_ptr(),
            low_leaf: low_leaf.as_mut_ptr() as *mut c_void,
            is_largest: is_largest.as_mut_ptr(),
            old_root: old_root.as_mut_ptr() as *mut c_void,
            interim_root: interim_root.as_mut_ptr() as *mut c_void,
            new_root: new_root.as_mut_ptr() as *mut c_void,
            new_leaf: new_leaf.as_mut_ptr() as *mut c_void,
            low_sib: low_sib.as_mut_ptr() as *mut c_void,
            new_sib: new_sib.as_mut_ptr() as *mut c_void,
        };
        check(&g, unsafe {
            imt_itree_insert_batch(self.handle, v.as_ptr() as *const c_void, n, &out, IMT_FMT_CANONICAL | IMT_SIB_ITEM_MAJOR)
        })?;
        let helper = |idx: u64| -> Vec<F> { (0..d).map(|l| if (idx >> l) & 1 == 0 { F::from(1) } else { F::from(0) }).collect() };
        let three = |b: &[u8]| -> [F; 3] { [F::from_bytes_le(&b[0..32]), F::from_bytes_le(&b[32..64]), F::from_bytes_le(&b[64..96])] };
        Ok((0..n)
            .map(|i| InsertWitness {
                old_root: F::from_bytes_le(&old_root[i * 32..][..32]),
                low_leaf: three(&low_leaf[i * 96..][..96]),
                low_leaf_proof: from_bytes(&low_sib[i * d * 32..][..d * 32]),
                low_leaf_proof_helper: helper(low_index[i]),
                new_root: F::from_bytes_le(&new_root[i * 32..][..32]),
                interim_root: F::from_bytes_le(&interim_root[i * 32..][..32]),
                new_leaf: three(&new_leaf[i * 96..][..96]),
                new_leaf_index: first + i as u64,
                new_leaf_proof: from_bytes(&new_sib[i * d * 32..][..d * 32]),
                new_leaf_proof_helper: helper(first + i as u64),
                is_new_leaf_largest: is_largest[i] != 0,
            })
            .collect())
    }
}
/// Everything `verify_non_inclusion` takes for one candidate value (`src/indexed_merkle_tree.rs:127-137`).
#[derive(Clone, Debug)]
pub struct NonInclusionWitness<F> {
    pub root: F,
    pub low_leaf: [F; 3],
    pub low_leaf_index: u64,
    pub low_leaf_proof: Vec<F>,
    pub low_leaf_proof_helper: Vec<F>,
    pub new_leaf_value: F,
    pub is_new_leaf_largest: bool,
}

impl IndexedTree {
    /// Non-membership witnesses of n candidate values against the current tree, from the device-resident sorted
    /// index (`imt_itree_non_membership_witness`); `Err(code -10)` if a candidate is 0 or already stored.
    pub fn non_inclusion_witnesses<F: ScalarField>(&self, vals: &[F]) -> Result<Vec<NonInclusionWitness<F>>, ImtError> {
        let root: F = self.root()?;
        let g = context().lock().unwrap();
        let (n, d) = (vals.len(), self.depth);
        let v = to_bytes(vals);
        let mut low_index = vec![0u64; n];
        let mut is_largest = vec![0u8; n];
        let mut low_leaf = vec![0u8; n * 96];
        let mut low_sib = vec![0u8; n * d * 32];
        check(&g, unsafe {
            imt_itree_non_membership_witness(self.handle, v.as_ptr() as *const c_void, n, low_index.as_mut_ptr(),
                                             low_leaf.as_mut_ptr() as *mut c_void, is_largest.as_mut_ptr(),
                                             low_sib.as_mut_ptr() as *mut c_void, IMT_FMT_CANONICAL | IMT_SIB_ITEM_MAJOR)
        })?;
        Ok((0..n)
            .map(|i| NonInclusionWitness {
                root,
                low_leaf: [F::from_bytes_le(&low_leaf[i * 96..][..32]), F::from_bytes_le(&low_leaf[i * 96 + 32..][..32]),
                           F::from_bytes_le(&low_leaf[i * 96 + 64..][..32])],
                low_leaf_index: low_index[i],
                low_leaf_proof: from_bytes(&low_sib[i * d * 32..][..d * 32]),
                low_leaf_proof_helper: (0..d).map(|l| if (low_index[i] >> l) & 1 == 0 { F::from(1) } else { F::from(0) }).collect(),
                new_leaf_value: vals[i],
                is_new_leaf_largest: is_largest[i] != 0,
            })
            .collect())
    }
}

impl IndexedTree {
    /// Checkpoint: every leaf `{val, next_val, next_idx}` in index order (the reference's serde leaf, `src/utils.rs:12-17`),
    /// read from the device-resident index (`imt_itree_get_leaves` with `index = NULL`).
    pub fn snapshot<F: ScalarField>(&self) -> Result<Vec<[F; 3]>, ImtError> {
        let g = context().lock().unwrap();
        let n = self.size() as usize;
        let mut pre = vec![0u8; n * 96];
        check(&g, unsafe {
            imt_itree_get_leaves(self.handle, std::ptr::null(), n, pre.as_mut_ptr() as *mut c_void, IMT_FMT_CANONICAL)
        })?;
        Ok((0..n)
            .map(|i| [F::from_bytes_le(&pre[i * 96..][..32]), F::from_bytes_le(&pre[i * 96 + 32..][..32]),
                      F::from_bytes_le(&pre[i * 96 + 64..][..32])])
            .collect())
    }
    /// Resume: replace the contents with a snapshot.  The GPU checks that the leaves are one sorted linked list from the
    /// `{0,..}` sentinel and rebuilds every level; `Err(code -10)` names the first leaf with a broken link and leaves the
    /// tree as it was (`imt_itree_load`).
    pub fn load<F: ScalarField>(&mut self, leaves: &[[F; 3]]) -> Result<(), ImtError> {
        let g = context().lock().unwrap();
        let mut pre = Vec::with_capacity(leaves.len() * 96);
        for l in leaves {
            pre.extend_from_slice(&to_bytes(&l[..]));
        }
        check(&g, unsafe { imt_itree_load(self.handle, pre.as_ptr() as *const c_void, leaves.len() as u64, IMT_FMT_CANONICAL) })
    }
}

impl Drop for IndexedTree {
    fn drop(&mut self) {
        let _g = context().lock().unwrap();
        unsafe { imt_itree_free(self.handle) }
    }
}

/// Witness traces of every hash of one `compute_merkle_root` call per item (leaf hash first when `leaf3` is
/// given), item-major: `rows[i]` = `1209 + depth * 1208` (or `depth * 1208`) field elements, the order in which
/// `chip::TracedPoseidonHasher` consumes them.  `sib` is item-major `[n][depth]`.
pub fn path_traces<F: ScalarField>(
    leaf: Option<&[F]>,
    leaf3: Option<&[[F; 3]]>,
    index: &[u64],
    sib: &[F],
    depth: usize,
) -> Result<Vec<Vec<F>>, ImtError> {
    let g = context().lock().unwrap();
    let n = index.len();
    let rows = (if leaf3.is_some() { unsafe { imt_hash_trace_rows(3) } } else { 0 }) + depth * unsafe { imt_hash_trace_rows(2) };
    let l = leaf.map(to_bytes);
    let l3: Option<Vec<u8>> = leaf3.map(|t| t.iter().flat_map(|p| to_bytes(p)).collect());
    let s = to_bytes(sib);
    // MONT256 rows could be transmuted into Fr without arithmetic; the generic path parses canonical bytes
    let mut trace = vec![0u8; n * rows * 32];
    check(&g, unsafe {
        imt_path_trace_batch(
            g.ctx,
            l.as_ref().map_or(std::ptr::null(), |v| v.as_ptr() as *const c_void),
            l3.as_ref().map_or(std::ptr::null(), |v| v.as_ptr() as *const c_void),
            index.as_ptr(),
            s.as_ptr() as *const c_void,
            depth as u32,
            n,
            trace.as_mut_ptr() as *mut c_void,
            std::ptr::null_mut(),
            IMT_FMT_CANONICAL | IMT_SIB_ITEM_MAJOR | IMT_TRACE_ITEM_MAJOR,
        )
    })?;
    Ok(trace.chunks_exact(rows * 32).map(from_bytes).collect())
}

/// All 3 + 4 d hash traces of `insert_leaf` for every witness, in the circuit's call order, one `Vec<F>` of
/// `imt_insert_trace_rows(depth)` rows per insertion (one GPU call for the whole slice: a call costs ~13 ms however
/// few items it carries, so batch).  5.1 MB per insertion at depth 32.
pub fn insert_traces<F: ScalarField>(w: &[InsertWitness<F>], depth: usize) -> Result<Vec<Vec<F>>, ImtError> {
    let g = context().lock().unwrap();
    let n = w.len();
    let rows = unsafe { imt_insert_trace_rows(depth as u32) };
    let idx = |h: &Vec<F>| h.iter().enumerate().fold(0u64, |a, (l, x)| if *x == F::ZERO { a | (1u64 << l) } else { a });
    let low_leaf: Vec<u8> = w.iter().flat_map(|x| to_bytes(&x.low_leaf)).collect();
    let new_leaf: Vec<u8> = w.iter().flat_map(|x| to_bytes(&x.new_leaf)).collect();
    let low_index: Vec<u64> = w.iter().map(|x| idx(&x.low_leaf_proof_helper)).collect();
    let new_path: Vec<u64> = w.iter().map(|x| idx(&x.new_leaf_proof_helper)).collect();
    let new_index: Vec<u64> = w.iter().map(|x| x.new_leaf_index).collect();
    let low_sib: Vec<u8> = w.iter().flat_map(|x| to_bytes(&x.low_leaf_proof)).collect();
    let new_sib: Vec<u8> = w.iter().flat_map(|x| to_bytes(&x.new_leaf_proof)).collect();
    let mut trace = vec![0u8; n * rows * 32];
    check(&g, unsafe {
        imt_insert_trace_batch(
            g.ctx,
            low_leaf.as_ptr() as *const c_void,
            low_index.as_ptr(),
            low_sib.as_ptr() as *const c_void,
            new_leaf.as_ptr() as *const c_void,
            new_index.as_ptr(),
            new_path.as_ptr(),
            new_sib.as_ptr() as *const c_void,
            depth as u32,
            n,
            trace.as_mut_ptr() as *mut c_void,
            IMT_FMT_CANONICAL | IMT_TRACE_ITEM_MAJOR, // = IMT_SIB_ITEM_MAJOR: per-item proofs, per-item traces
        )
    })?;
    Ok(trace.chunks_exact(rows * 32).map(from_bytes).collect())
}

/// The static cell map of one hash (`imt_hash_trace_layout`): cells, constants, output row.
pub fn trace_layout<F: ScalarField>(arity: usize) -> Result<(Vec<imt_trace_cell>, Vec<F>, usize), ImtError> {
    let g = context().lock().unwrap();
    let (mut nc, mut nk, mut row) = (0usize, 0usize, 0u32);
    check(&g, unsafe {
        imt_hash_trace_layout(g.ctx, arity as i32, std::ptr::null_mut(), 0, &mut nc, std::ptr::null_mut(), 0, &mut nk, &mut row, IMT_FMT_CANONICAL)
    })?;
    let mut cells = vec![imt_trace_cell::default(); nc];
    let mut consts = vec![0u8; nk * 32];
    check(&g, unsafe {
        imt_hash_trace_layout(g.ctx, arity as i32, cells.as_mut_ptr(), nc, &mut nc, consts.as_mut_ptr() as *mut c_void, nk, &mut nk, &mut row, IMT_FMT_CANONICAL)
    })?;
    Ok((cells, from_bytes(&consts), row as usize))
}

/// f3: the static cell map of one `is_less_than` (`src/indexed_merkle_tree.rs:98-125`) at the RangeChip's `lookup_bits`
/// (`imt_less_than_trace_layout`): cells, constants, output row, and the rows whose cells also go to the lookup table
/// (`imt_less_than_lookup_rows`: the limbs of both range checks).  Inputs 0..3 = a_q, a_r, b_q, b_r.
pub fn less_than_layout<F: ScalarField>(lookup_bits: usize) -> Result<(Vec<imt_trace_cell>, Vec<F>, usize, Vec<u32>), ImtError> {
    let g = context().lock().unwrap();
    let (mut nc, mut nk, mut row) = (0usize, 0usize, 0u32);
    check(&g, unsafe {
        imt_less_than_trace_layout(g.ctx, lookup_bits as u32, std::ptr::null_mut(), 0, &mut nc, std::ptr::null_mut(), 0, &mut nk, &mut row, IMT_FMT_CANONICAL)
    })?;
    let mut cells = vec![imt_trace_cell::default(); nc];
    let mut consts = vec![0u8; nk * 32];
    check(&g, unsafe {
        imt_less_than_trace_layout(g.ctx, lookup_bits as u32, cells.as_mut_ptr(), nc, &mut nc, consts.as_mut_ptr() as *mut c_void, nk, &mut nk, &mut row, IMT_FMT_CANONICAL)
    })?;
    let mut nl = 0usize;
    check(&g, unsafe { imt_less_than_lookup_rows(lookup_bits as u32, std::ptr::null_mut(), 0, &mut nl) })?;
    let mut lookup = vec![0u32; nl];
    check(&g, unsafe { imt_less_than_lookup_rows(lookup_bits as u32, lookup.as_mut_ptr(), nl, &mut nl) })?;
    Ok((cells, from_bytes(&consts), row as usize, lookup))
}

/// f3: every new advice value of `is_less_than(a, b)` for n pairs of field elements, item-major: `rows[i]` =
/// `imt_less_than_trace_rows(lookup_bits)` elements in assignment order (`imt_less_than_trace_batch`).
pub fn less_than_traces<F: ScalarField>(a: &[F], b: &[F], lookup_bits: usize) -> Result<Vec<Vec<F>>, ImtError> {
    assert_eq!(a.len(), b.len());
    let g = context().lock().unwrap();
    let n = a.len();
    let rows = unsafe { imt_less_than_trace_rows(lookup_bits as u32) };
    let mut trace = vec![0u8; n * rows * 32];
    check(&g, unsafe {
        imt_less_than_trace_batch(g.ctx, to_bytes(a).as_ptr() as *const c_void, to_bytes(b).as_ptr() as *const c_void, n,
                                  lookup_bits as u32, trace.as_mut_ptr() as *mut c_void, std::ptr::null_mut(),
                                  IMT_FMT_CANONICAL | IMT_TRACE_ITEM_MAJOR)
    })?;
    Ok((0..n).map(|i| from_bytes(&trace[i * rows * 32..][..rows * 32])).collect())
}

/// f3 for `verify_non_inclusion` on its own (`src/indexed_merkle_tree.rs:127-229`, BASELINE config 3's gadget): every new
/// advice value outside its hashes for n candidates, item-major: `rows[i]` = `imt_non_inclusion_gadget_rows(depth,
/// lookup_bits)` elements in assignment order (`imt_non_inclusion_gadget_trace_batch`), from the witnesses
/// [`IndexedTree::non_inclusion_witnesses`] returned.  The hash blocks in between are `path_traces(None, Some(low_leaf), ..)`;
/// `imt_non_inclusion_column_segments` says how the two interleave, `imt_insert_gadget_lookup_rows` which rows are lookup cells.
pub fn non_inclusion_gadget_traces<F: ScalarField>(w: &[NonInclusionWitness<F>], depth: usize, lookup_bits: usize) -> Result<Vec<Vec<F>>, ImtError> {
    let g = context().lock().unwrap();
    let n = w.len();
    let rows = unsafe { imt_non_inclusion_gadget_rows(depth as u32, lookup_bits as u32) };
    let mut low_leaf = Vec::with_capacity(n * 96);
    let mut low_sib = Vec::with_capacity(n * depth * 32);
    let mut vals = Vec::with_capacity(n * 32);
    let (mut low_index, mut largest) = (Vec::with_capacity(n), Vec::with_capacity(n));
    for x in w {
        low_leaf.extend_from_slice(&to_bytes(&x.low_leaf[..]));
        low_sib.extend_from_slice(&to_bytes(&x.low_leaf_proof));
        vals.extend_from_slice(&to_bytes(&[x.new_leaf_value]));
        low_index.push(x.low_leaf_index);
        largest.push(x.is_new_leaf_largest as u8);
    }
    let mut trace = vec![0u8; n * rows * 32];
    check(&g, unsafe {
        imt_non_inclusion_gadget_trace_batch(g.ctx, low_leaf.as_ptr() as *const c_void, low_index.as_ptr(), low_sib.as_ptr() as *const c_void,
                                             vals.as_ptr() as *const c_void, largest.as_ptr(), depth as u32, lookup_bits as u32, n,
                                             trace.as_mut_ptr() as *mut c_void, IMT_FMT_CANONICAL | IMT_TRACE_ITEM_MAJOR)
    })?;
    Ok((0..n).map(|i| from_bytes(&trace[i * rows * 32..][..rows * 32])).collect())
}
