//! Circuit side.
//!
//! The reference's gadget recomputes every Poseidon on the CPU while it assigns:
//! `hasher.hash_fix_len_array(ctx, gate, &inp)` at `src/indexed_merkle_tree.rs:92` (inside the path loop `:90-93`),
//! `:194`, `:271-275`, `:299-303` -- 3 + 4 d hashes per `insert_leaf`.  [`TracedPoseidonHasher`] has the same method
//! and lays down the SAME cells (same gates, same copy constraints, same constants), but takes every new value
//! from a witness trace the GPU produced (`imt_path_trace_batch` / `imt_hash_trace_batch`) instead of computing it.
//! To use it, the reference's private helpers only change the TYPE of their `hasher` parameter (or become generic
//! over [`FixLenHasher`], which both hashers implement) -- see INTEGRATION.md.
//!
//! [`IndexedMerkleTreeChip`] is the optional sugar `north_star` names: the reference has no chip struct, only the
//! free functions `insert_leaf` (`:231`) and `verify_non_inclusion` (`:127`).

use crate::ffi::{imt_trace_cell, IMT_CELL_CONST, IMT_CELL_COPY, IMT_CELL_INIT, IMT_CELL_INPUT, IMT_CELL_WITNESS};
use crate::gpu::{self, InsertWitness};
use halo2_base::gates::GateInstructions;
use halo2_base::poseidon::hasher::PoseidonHasher;
use halo2_base::utils::BigPrimeField;
use halo2_base::QuantumCell::{Constant, Existing, Witness};
use halo2_base::{AssignedValue, Context};
use std::cell::RefCell;
use std::collections::VecDeque;

/// What `compute_merkle_root` & co. need from a hasher (`PoseidonHasher::hash_fix_len_array`'s signature).
pub trait FixLenHasher<F: BigPrimeField> {
    fn hash_fix_len_array(
        &self,
        ctx: &mut Context<F>,
        gate: &impl GateInstructions<F>,
        inputs: &[AssignedValue<F>],
    ) -> AssignedValue<F>;
}

impl<F: BigPrimeField, const T: usize, const RATE: usize> FixLenHasher<F> for PoseidonHasher<F, T, RATE> {
    fn hash_fix_len_array(&self, ctx: &mut Context<F>, gate: &impl GateInstructions<F>, inputs: &[AssignedValue<F>]) -> AssignedValue<F> {
        PoseidonHasher::hash_fix_len_array(self, ctx, gate, inputs)
    }
}

/// One arity's static cell map.
pub struct TraceLayout<F> {
    pub cells: Vec<imt_trace_cell>,
    pub constants: Vec<F>,
    pub out_row: usize,
    pub rows: usize,
}

/// Assigns precomputed traces.  Traces are queued in the order the circuit will ask for them (one `Vec<F>` of 1208 /
/// 1209 rows per hash); `init_state` are the hasher's three initial-state cells ([2^64, 0, 0], loaded once with
/// `ctx.load_constant`, exactly what `PoseidonHasher::initialize_consts` does at `:442`).
pub struct TracedPoseidonHasher<F: BigPrimeField> {
    layout2: TraceLayout<F>,
    layout3: TraceLayout<F>,
    init_state: [AssignedValue<F>; 3],
    queue: RefCell<VecDeque<Vec<F>>>,
}

impl<F: BigPrimeField> TracedPoseidonHasher<F> {
    pub fn new(ctx: &mut Context<F>) -> Self {
        let mk = |arity: usize| {
            let (cells, constants, out_row) = gpu::trace_layout::<F>(arity).expect("imt_hash_trace_layout");
            let rows = cells.iter().filter(|c| c.kind == IMT_CELL_WITNESS).count();
            TraceLayout { cells, constants, out_row, rows }
        };
        let init_state = [
            ctx.load_constant(F::from_u128(1u128 << 64)),
            ctx.load_constant(F::ZERO),
            ctx.load_constant(F::ZERO),
        ];
        TracedPoseidonHasher { layout2: mk(2), layout3: mk(3), init_state, queue: RefCell::new(VecDeque::new()) }
    }

    /// Queue traces in call order.  `rows` may hold several hashes back to back (what `gpu::path_traces` returns
    /// per item): it is split by the row counts of `arities`.
    pub fn push_traces(&self, rows: &[F], arities: &[usize]) {
        let mut off = 0;
        let mut q = self.queue.borrow_mut();
        for &a in arities {
            let n = if a == 3 { self.layout3.rows } else { self.layout2.rows };
            q.push_back(rows[off..off + n].to_vec());
            off += n;
        }
        assert_eq!(off, rows.len(), "trace length does not match the hash sequence");
    }

    pub fn pending(&self) -> usize {
        self.queue.borrow().len()
    }
}

impl<F: BigPrimeField> FixLenHasher<F> for TracedPoseidonHasher<F> {
    /// Cell for cell what `PoseidonHasher::hash_fix_len_array` assigns, region by region (a region = one
    /// `ctx.assign_region` call of the original gadget: a `gate.add` / `sum` / `mul` / `mul_add` / `inner_product`).
    fn hash_fix_len_array(&self, ctx: &mut Context<F>, _gate: &impl GateInstructions<F>, inputs: &[AssignedValue<F>]) -> AssignedValue<F> {
        let layout = match inputs.len() {
            2 => &self.layout2,
            3 => &self.layout3,
            n => panic!("no trace layout for {n} inputs"),
        };
        let rows = self.queue.borrow_mut().pop_front().expect("no precomputed trace queued for this hash");
        assert_eq!(rows.len(), layout.rows);
        let assigned = assign_layout(ctx, &layout.cells, &layout.constants, inputs, &self.init_state, &rows);
        assigned[layout.out_row].unwrap()
    }
}

/// Lay one gadget's column down from its cell map: region by region (a region = one `ctx.assign_region` call of the
/// original gadget), every NEW value taken from `rows`, every copy an `Existing(..)` of a cell assigned earlier.
/// Returns the assigned cell of every trace row.  Shared by the hash layouts (f1) and the comparison layout (f3).
pub fn assign_layout<F: BigPrimeField>(
    ctx: &mut Context<F>,
    cells: &[imt_trace_cell],
    constants: &[F],
    inputs: &[AssignedValue<F>],
    init_state: &[AssignedValue<F>],
    rows: &[F],
) -> Vec<Option<AssignedValue<F>>> {
    let mut assigned: Vec<Option<AssignedValue<F>>> = vec![None; rows.len()];
    let mut a = 0usize;
    while a < cells.len() {
        let mut b = a + 1;
        while b < cells.len() && cells[b].region == 0 {
            b += 1;
        }
        let region = &cells[a..b];
        let quantum = region.iter().map(|c| match c.kind {
            IMT_CELL_CONST => Constant(constants[c.index as usize]),
            IMT_CELL_INPUT => Existing(inputs[c.index as usize]),
            IMT_CELL_INIT => Existing(init_state[c.index as usize]),
            IMT_CELL_WITNESS => Witness(rows[c.index as usize]),
            IMT_CELL_COPY => Existing(assigned[c.index as usize].expect("copy of a row that is not assigned yet")),
            k => panic!("unknown cell kind {k}"),
        });
        let gates = region.iter().enumerate().filter(|(_, c)| c.gate == 1).map(|(i, _)| i as isize);
        let start = ctx.advice.len();
        ctx.assign_region(quantum.collect::<Vec<_>>(), gates.collect::<Vec<_>>());
        for (i, c) in region.iter().enumerate() {
            if c.kind == IMT_CELL_WITNESS {
                assigned[c.index as usize] = Some(ctx.get((start + i) as isize));
            }
        }
        a = b;
    }
    assigned
}

/// f3: the reference's `is_less_than` (`src/indexed_merkle_tree.rs:98-125`: `range.is_less_than(.., 128)` and
/// `gate.is_equal` for the high and the low 128-bit limbs, `not` x4, `and` x4, `or`) with every new value -- the shifted
/// differences, their `lookup_bits`-wide limbs, the running sums, both inverses, the booleans -- taken from rows the GPU
/// produced (`gpu::less_than_traces`, or the two K-row stretches of `imt_insert_gadget_trace_batch`) instead of being
/// computed while assigning.  Same cells, same gates, same copies; the limb cells are registered with the RangeChip's
/// lookup exactly where `range_check` would (`imt_less_than_lookup_rows`).  In the reference the function is private, so
/// using this means replacing its body by `traced.is_less_than(ctx, range, [a_q, a_r, b_q, b_r], rows)`.
pub struct TracedLessThan<F: BigPrimeField> {
    cells: Vec<imt_trace_cell>,
    constants: Vec<F>,
    out_row: usize,
    lookup_rows: Vec<u32>,
    pub rows: usize,
}

impl<F: BigPrimeField> TracedLessThan<F> {
    pub fn new(lookup_bits: usize) -> Self {
        let (cells, constants, out_row, lookup_rows) = gpu::less_than_layout::<F>(lookup_bits).expect("imt_less_than_trace_layout");
        let rows = cells.iter().filter(|c| c.kind == IMT_CELL_WITNESS).count();
        TracedLessThan { cells, constants, out_row, lookup_rows, rows }
    }

    pub fn is_less_than(
        &self,
        ctx: &mut Context<F>,
        range: &halo2_base::gates::RangeChip<F>,
        limbs: [AssignedValue<F>; 4],
        rows: &[F],
    ) -> AssignedValue<F> {
        assert_eq!(rows.len(), self.rows, "one comparison is {} rows", self.rows);
        let assigned = assign_layout(ctx, &self.cells, &self.constants, &limbs, &[], rows);
        for &r in &self.lookup_rows {
            range.add_cell_to_lookup(ctx, assigned[r as usize].expect("a lookup row is a trace row"));
        }
        assigned[self.out_row].unwrap()
    }
}

/// The ONE extra constraint a circuit needs when the tree behind the root is the SUBTREE layout (`sharded.py`,
/// `bench.py --gpus N` mode "subtrees": `world = 2^k` sorted lists, list `g` = the values with `v mod world == g`, stored
/// under leaf indices `[g << (d - k), (g + 1) << (d - k))`).  The reference's `verify_non_inclusion` (`:127-229`) and
/// `insert_leaf` (`:231-314`) never tie a leaf's POSITION to a VALUE -- with one list they need not.  With `world` lists
/// under one root they must: a low leaf taken from a list that does not own `v` (for instance another list's sentinel
/// `{0, w, idx}` with `w > v`) satisfies every constraint of `verify_non_inclusion` for a `v` that IS stored in its own
/// list, and `insert_leaf` would accept `v` a second time.  So, next to every `verify_non_inclusion(.., low_leaf_proof_helper,
/// new_leaf_value, ..)` and for both paths of every `insert_leaf`, also call
///
/// ```ignore
/// constrain_owner_subtree(ctx, range, &new_leaf_value, &low_leaf_proof_helper, k);   // the low leaf's list owns v
/// constrain_owner_subtree(ctx, range, &new_leaf.val,   &new_leaf_proof_helper, k);   // insert_leaf: v lands in its own list
/// ```
///
/// It constrains `v mod 2^k` to equal the subtree number the path's top `k` helper bits spell (`helper = 1` means "left
/// child", `src/utils.rs:79`, so bit `j` of the subtree number is `1 - helper[d - k + j]`).  `v mod 2^k` is taken from the
/// CANONICAL integer of `v`: the value is split into two range-checked 128-bit limbs `(q, r)` with `q 2^128 + r = v`
/// in the field AND `(q, r) <= ((p - 1) >> 128, (p - 1) mod 2^128)` lexicographically.  Without the second condition
/// the decomposition is not unique -- for every `v < 2^254 - p` (a quarter of the field, all small values) the limbs of
/// `v + p` satisfy the same field equation, and since `p` is odd they name ANOTHER subtree: the forgery this constraint
/// exists to stop would go through (a single `range.div_mod(v, 2^k, 254)` has exactly that hole: halo2-base's div_mod
/// is only sound below the field's capacity of 253 bits).  The residue itself is then a `div_mod` of the 128-bit low
/// limb.  The single-list layout (`sliced.py`, `bench.py`'s `value`) needs none of this.
pub fn constrain_owner_subtree<F: BigPrimeField>(
    ctx: &mut Context<F>,
    range: &halo2_base::gates::RangeChip<F>,
    value: &AssignedValue<F>,
    proof_helper: &[AssignedValue<F>],
    k: usize,
) {
    constrain_owner_subtree_with(ctx, range, value, proof_helper, k, None)
}

/// The same with the prover's choice of limbs made explicit (`limbs = Some((q, r))`): what a negative test needs to
/// show that the limbs of `v + p` are refused (`tests/mockprover.rs::owner_constraint_refuses_the_limbs_of_v_plus_p`).
pub fn constrain_owner_subtree_with<F: BigPrimeField>(
    ctx: &mut Context<F>,
    range: &halo2_base::gates::RangeChip<F>,
    value: &AssignedValue<F>,
    proof_helper: &[AssignedValue<F>],
    k: usize,
    limbs: Option<(num_bigint::BigUint, num_bigint::BigUint)>,
) {
    use halo2_base::gates::RangeInstructions;
    use halo2_base::utils::{biguint_to_fe, fe_to_biguint, modulus};
    use num_bigint::BigUint;
    if k == 0 {
        return;
    }
    let gate = range.gate();
    let d = proof_helper.len();
    assert!(k <= d && k <= 128, "more subtree bits than tree levels (or than a limb)");
    let one = BigUint::from(1u64);
    let mask = (&one << 128) - &one;
    // canonical 128-bit limbs of the value
    let v = fe_to_biguint(value.value());
    let (q_bu, r_bu) = limbs.unwrap_or((&v >> 128, &v & &mask));
    let q = ctx.load_witness(biguint_to_fe::<F>(&q_bu));
    let r = ctx.load_witness(biguint_to_fe::<F>(&r_bu));
    range.range_check(ctx, q, 128);
    range.range_check(ctx, r, 128);
    let recomposed = gate.mul_add(ctx, q, Constant(biguint_to_fe::<F>(&(&one << 128))), r);
    ctx.constrain_equal(&recomposed, value);
    // (q, r) <= limbs of p - 1:  q < pq  or  (q == pq and r < pr + 1)
    let p_minus_1 = modulus::<F>() - &one;
    let (pq, pr) = (&p_minus_1 >> 128, &p_minus_1 & &mask);
    let q_lt = range.is_less_than(ctx, q, Constant(biguint_to_fe::<F>(&pq)), 128);
    let q_eq = gate.is_equal(ctx, q, Constant(biguint_to_fe::<F>(&pq)));
    let r_le = range.is_less_than(ctx, r, Constant(biguint_to_fe::<F>(&(&pr + &one))), 128);
    let tail = gate.and(ctx, q_eq, r_le);
    let canonical = gate.or(ctx, q_lt, tail);
    gate.assert_is_const(ctx, &canonical, &F::ONE);
    // the residue of the canonical integer (k <= 128: it lives in the low limb)
    let (_, residue) = range.div_mod(ctx, r, &one << k, 128);
    let bits: Vec<AssignedValue<F>> = proof_helper[d - k..].iter().map(|h| gate.not(ctx, *h)).collect();
    let weights = (0..k).map(|j| Constant(F::from(1u64 << j)));
    let subtree = gate.inner_product(ctx, bits, weights);
    ctx.constrain_equal(&residue, &subtree);
}

/// insert / update / non-membership in one object (the names `north_star` uses).  `assign_insert` loads one
/// [`InsertWitness`] into the circuit and calls the reference's own `insert_leaf` through the closure the caller
/// passes (the function is private to the reference's module, `src/indexed_merkle_tree.rs:231`), with a
/// [`TracedPoseidonHasher`] primed with the 3 + 4 d traces of that insertion.
pub struct IndexedMerkleTreeChip {
    pub tree: gpu::IndexedTree,
}

/// the assigned cells `insert_leaf` takes, in its argument order (`:235-244`)
pub struct AssignedInsert<F: BigPrimeField> {
    pub old_root: AssignedValue<F>,
    pub low_leaf: [AssignedValue<F>; 3],
    pub low_leaf_proof: Vec<AssignedValue<F>>,
    pub low_leaf_proof_helper: Vec<AssignedValue<F>>,
    pub new_root: AssignedValue<F>,
    pub new_leaf: [AssignedValue<F>; 3],
    pub new_leaf_index: AssignedValue<F>,
    pub new_leaf_proof: Vec<AssignedValue<F>>,
    pub new_leaf_proof_helper: Vec<AssignedValue<F>>,
    pub is_new_leaf_largest: AssignedValue<F>,
}

impl IndexedMerkleTreeChip {
    pub fn new(depth: usize, capacity: u64) -> Self {
        IndexedMerkleTreeChip { tree: gpu::IndexedTree::new(depth, capacity).expect("imt_itree_new") }
    }

    /// n insertions on the GPU: the witnesses `insert_leaf` takes, one per value
    pub fn insert<F: BigPrimeField>(&mut self, vals: &[F]) -> Vec<InsertWitness<F>> {
        self.tree.insert_batch(vals).expect("imt_itree_insert_batch")
    }

    /// The 3 + 4 d traces of every witness in one GPU call (`imt_insert_trace_batch`), in `insert_leaf`'s own order:
    /// low leaf + path (`:193-204`), rewritten low leaf + path (`:271-284`), the zero leaf's path at the new slot
    /// (`:286-294`; no leaf hash, the zero-leaf hash is a constant), new leaf + path (`:299-312`).
    pub fn insert_traces<F: BigPrimeField>(&self, w: &[InsertWitness<F>]) -> Vec<Vec<F>> {
        gpu::insert_traces(w, self.tree.depth).expect("imt_insert_trace_batch")
    }

    /// queue one insertion's traces (one element of `insert_traces`) on `hasher`, in call order
    pub fn prime_insert_traces<F: BigPrimeField>(&self, hasher: &TracedPoseidonHasher<F>, rows: &[F]) {
        let d = self.tree.depth;
        let with_leaf = std::iter::once(3).chain(std::iter::repeat(2).take(d));
        let arities: Vec<usize> = with_leaf.clone().chain(with_leaf.clone()).chain(std::iter::repeat(2).take(d)).chain(with_leaf).collect();
        hasher.push_traces(rows, &arities);
    }

    /// non-membership (`verify_non_inclusion`, `:127-229`): witnesses for n candidate values against the current tree
    pub fn non_inclusion<F: BigPrimeField>(&self, vals: &[F]) -> Vec<gpu::NonInclusionWitness<F>> {
        self.tree.non_inclusion_witnesses(vals).expect("imt_itree_non_membership_witness")
    }

    /// queue the 1 + d traces `verify_non_inclusion` consumes (low-leaf hash `:193-194`, its path `:196-204`) for each
    /// witness, in order; one GPU call for all of them
    pub fn prime_non_inclusion_traces<F: BigPrimeField>(&self, hasher: &TracedPoseidonHasher<F>, w: &[gpu::NonInclusionWitness<F>]) {
        let d = self.tree.depth;
        let leaves: Vec<[F; 3]> = w.iter().map(|x| x.low_leaf).collect();
        let index: Vec<u64> = w.iter().map(|x| x.low_leaf_index).collect();
        let sib: Vec<F> = w.iter().flat_map(|x| x.low_leaf_proof.iter().copied()).collect();
        let traces = gpu::path_traces(None, Some(&leaves), &index, &sib, d).expect("imt_path_trace_batch");
        let arities: Vec<usize> = std::iter::once(3).chain(std::iter::repeat(2).take(d)).collect();
        for rows in &traces {
            hasher.push_traces(rows, &arities);
        }
    }

    /// loads one witness as `insert_leaf`'s arguments (`ctx.load_witness` per value, as the reference's tests do at
    /// `:444-474`)
    pub fn assign_insert<F: BigPrimeField>(&self, ctx: &mut Context<F>, w: &InsertWitness<F>) -> AssignedInsert<F> {
        let mut lw = |v: F| ctx.load_witness(v);
        AssignedInsert {
            old_root: lw(w.old_root),
            low_leaf: [lw(w.low_leaf[0]), lw(w.low_leaf[1]), lw(w.low_leaf[2])],
            low_leaf_proof: w.low_leaf_proof.iter().map(|v| lw(*v)).collect(),
            low_leaf_proof_helper: w.low_leaf_proof_helper.iter().map(|v| lw(*v)).collect(),
            new_root: lw(w.new_root),
            new_leaf: [lw(w.new_leaf[0]), lw(w.new_leaf[1]), lw(w.new_leaf[2])],
            new_leaf_index: lw(F::from(w.new_leaf_index)),
            new_leaf_proof: w.new_leaf_proof.iter().map(|v| lw(*v)).collect(),
            new_leaf_proof_helper: w.new_leaf_proof_helper.iter().map(|v| lw(*v)).collect(),
            is_new_leaf_largest: lw(F::from(w.is_new_leaf_largest)),
        }
    }
}

