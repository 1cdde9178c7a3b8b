//! The pin this repository cannot run itself (its build image has no cargo / rustc): ONE `cargo test` by anyone who
//! has a Rust toolchain and an MI355X closes the two items the Python / C++ suites leave open --
//!
//!  * f1, the ORDER of the witness trace: `traced_hasher_lays_down_the_same_column` runs halo2-base's own
//!    `PoseidonHasher::hash_fix_len_array` and [`TracedPoseidonHasher`] (GPU rows, `imt_hash_trace_batch`) on the same
//!    inputs in two contexts and compares the advice columns VALUE BY VALUE and the copy constraints they recorded.
//!    Needs nothing of the reference crate.
//!  * BASELINE config 5, the MockProver half: `insert_leaf_rounds_are_satisfied` replays the value sequence of the
//!    reference's `test_insert_leaf_multiple_round` (`src/indexed_merkle_tree.rs:679-803`: 30, 10, 20, 5, 50, 35 into
//!    a depth-3 tree) and `insert_leaf_at_depth_32_fits_k17` one depth-32 insertion, each through the reference's OWN
//!    `insert_leaf` gadget (`:231-314`) inside `base_test().expect_satisfied(true)` (`:434-438`, `:747-751`), with
//!    every Poseidon witness taken from the GPU trace.  The zero-leaf constant the gadget enforces (`:247-250`)
//!    is the reference's one absolute known answer, so a satisfied circuit pins the GPU values to it.
//!    These two need the reference crate with ONE mechanical edit: its four helpers take `&PoseidonHasher<F, T, RATE>`
//!    (`:69`, `:82`, `:131`, `:235`); make that parameter `&impl imt_hip::chip::FixLenHasher<F>` (README.md here has
//!    the sed line).  They are behind the cargo feature `reference-gadget`.
//!
//!  * f3, the ORDER of the comparison's rows: `traced_less_than_lays_down_the_same_column` runs the gadget sequence of
//!    the reference's private `is_less_than` (`:98-125`: halo2-base's `range.is_less_than(.., 128)` + `gate.is_equal` per
//!    128-bit limb, `not` x4, `and` x4, `or`) and [`TracedLessThan`] (GPU rows, `imt_less_than_trace_batch`) in two
//!    contexts and compares the advice columns value by value, the copy constraints and the cells sent to the lookup.
//!    Needs nothing of the reference crate.
//!  * the soundness of the subtree layout's extra constraint: `owner_constraint_refuses_the_limbs_of_v_plus_p` (no GPU,
//!    no reference crate).
//!
//! Run:  IMT_HIP_LIB_DIR=<repo>/indexed-merkle-tree-halo2_amd/csrc cargo test --release --features reference-gadget
//!
//! tests/test_rust_binding.py checks that every `imt_hip::` item and every chip / gpu method named here exists in
//! `src/` with that name.
use halo2_base::gates::RangeInstructions;
use halo2_base::halo2_proofs::halo2curves::bn256::Fr;
use halo2_base::poseidon::hasher::spec::OptimizedPoseidonSpec;
use halo2_base::poseidon::hasher::PoseidonHasher;
use halo2_base::utils::testing::base_test;
use halo2_base::Context;
use imt_hip::chip::{FixLenHasher, IndexedMerkleTreeChip, TracedLessThan, TracedPoseidonHasher};
use imt_hip::gpu;

const T: usize = 3;
const RATE: usize = 2;
const R_F: usize = 8;
const R_P: usize = 57;

fn advice_values(ctx: &Context<Fr>) -> Vec<Fr> {
    ctx.advice.iter().map(|a| a.evaluate()).collect()
}

/// f1: same cells, same order, same equality constraints as the gadget the reference calls.
#[test]
fn traced_hasher_lays_down_the_same_column() {
    let cases: Vec<Vec<Fr>> = vec![
        vec![Fr::from(0), Fr::from(0), Fr::from(0)],
        vec![Fr::from(1), Fr::from(2)],
        vec![Fr::from(1), Fr::from(2), Fr::from(3)],
        vec![Fr::from(u64::MAX), -Fr::from(1)],
        vec![-Fr::from(2), Fr::from(1u64 << 63), -Fr::from(1)],
    ];
    for inputs in cases {
        let arity = inputs.len();
        // the GPU's rows for this hash
        let rows = if arity == 2 {
            gpu::path_traces(Some(&[inputs[0]]), None, &[0u64], &[inputs[1]], 1).expect("imt_path_trace_batch").remove(0)
        } else {
            gpu::path_traces::<Fr>(None, Some(&[[inputs[0], inputs[1], inputs[2]]]), &[0u64], &[], 0).expect("imt_path_trace_batch").remove(0)
        };
        let mut columns: Vec<Vec<Fr>> = Vec::new();
        let mut copies: Vec<usize> = Vec::new();
        let mut outputs: Vec<Fr> = Vec::new();
        for traced in [false, true] {
            base_test().k(12).lookup_bits(8).expect_satisfied(true).run(|ctx, range| {
                let gate = range.gate();
                // the three initial-state constants first, in both runs, so that the columns line up
                let ins: Vec<_> = inputs.iter().map(|v| ctx.load_witness(*v)).collect();
                let out = if traced {
                    let hasher = TracedPoseidonHasher::<Fr>::new(ctx);
                    hasher.push_traces(&rows, &[arity]);
                    let o = hasher.hash_fix_len_array(ctx, gate, &ins);
                    assert_eq!(hasher.pending(), 0);
                    o
                } else {
                    let mut hasher = PoseidonHasher::<Fr, T, RATE>::new(OptimizedPoseidonSpec::new::<R_F, R_P, 0>());
                    hasher.initialize_consts(ctx, gate);
                    hasher.hash_fix_len_array(ctx, gate, &ins)
                };
                outputs.push(*out.value());
                columns.push(advice_values(ctx));
                copies.push(ctx.copy_manager.lock().unwrap().advice_equalities.len());
            });
        }
        assert_eq!(outputs[0], outputs[1], "hash value");
        assert_eq!(columns[0].len(), columns[1].len(), "number of advice cells for {arity} inputs");
        for (i, (a, b)) in columns[0].iter().zip(columns[1].iter()).enumerate() {
            assert_eq!(a, b, "advice cell {i} of a {arity}-input hash: the trace order differs from halo2-base's");
        }
        assert_eq!(copies[0], copies[1], "number of copy constraints");
    }
}

/// f3: same cells, same order, same equality constraints, same lookup cells as the gadgets the reference's
/// `is_less_than` calls (`src/indexed_merkle_tree.rs:98-125`).
#[test]
fn traced_less_than_lays_down_the_same_column() {
    use halo2_base::gates::GateInstructions;
    use halo2_base::utils::{biguint_to_fe, fe_to_biguint};
    use num_bigint::BigUint;
    let lookup_bits = 18usize;                            // the reference's tests, `:436`
    let big = |x: u128, y: u128| -> Fr { biguint_to_fe(&((BigUint::from(x) << 128) + BigUint::from(y))) };
    let cases: Vec<(Fr, Fr)> = vec![
        (Fr::from(5), Fr::from(9)), (Fr::from(9), Fr::from(5)), (Fr::from(7), Fr::from(7)), (Fr::from(0), -Fr::from(1)),
        (big(3, u128::MAX), big(4, 0)), (big(4, 0), big(3, u128::MAX)), (big(6, 11), big(6, 12)), (big(6, 12), big(6, 11)),
    ];
    let traced = TracedLessThan::<Fr>::new(lookup_bits);
    let one = BigUint::from(1u64);
    let mask = (&one << 128) - &one;
    for (a, b) in cases {
        let rows = gpu::less_than_traces(&[a], &[b], lookup_bits).expect("imt_less_than_trace_batch").remove(0);
        let limbs: Vec<Fr> = [a, b].iter().flat_map(|v| { let x = fe_to_biguint(v); [biguint_to_fe(&(&x >> 128)), biguint_to_fe(&(&x & &mask))] }).collect();
        let mut columns: Vec<Vec<Fr>> = Vec::new();
        let mut copies: Vec<usize> = Vec::new();
        let mut lookups: Vec<usize> = Vec::new();
        let mut outputs: Vec<Fr> = Vec::new();
        for use_trace in [false, true] {
            base_test().k(12).lookup_bits(lookup_bits).expect_satisfied(true).run(|ctx, range| {
                let gate = range.gate();
                let l: Vec<_> = limbs.iter().map(|v| ctx.load_witness(*v)).collect();
                let (a_q, a_r, b_q, b_r) = (l[0], l[1], l[2], l[3]);
                let out = if use_trace {
                    traced.is_less_than(ctx, range, [a_q, a_r, b_q, b_r], &rows)
                } else {
                    // the call sequence of the reference's is_less_than, on halo2-base's own chips
                    let msb_lt = range.is_less_than(ctx, a_q, b_q, 128);
                    let msb_eq = gate.is_equal(ctx, a_q, b_q);
                    let lsb_lt = range.is_less_than(ctx, a_r, b_r, 128);
                    let lsb_eq = gate.is_equal(ctx, a_r, b_r);
                    let msb_ne = gate.not(ctx, msb_eq);
                    let msb_ge = gate.not(ctx, msb_lt);
                    let msb_eq2 = gate.not(ctx, msb_ne);
                    let lsb_ne = gate.not(ctx, lsb_eq);
                    let mut rhs = msb_ge;
                    for x in [lsb_lt, msb_eq2, lsb_ne] {
                        rhs = gate.and(ctx, rhs, x);
                    }
                    let lhs = gate.and(ctx, msb_lt, msb_ne);
                    gate.or(ctx, lhs, rhs)
                };
                outputs.push(*out.value());
                columns.push(advice_values(ctx));
                copies.push(ctx.copy_manager.lock().unwrap().advice_equalities.len());
                lookups.push(ctx.copy_manager.lock().unwrap().assigned_advices.len());
            });
        }
        assert_eq!(outputs[0], outputs[1], "comparison result");
        assert_eq!(outputs[0], if fe_to_biguint(&a) < fe_to_biguint(&b) { Fr::from(1) } else { Fr::from(0) });
        assert_eq!(columns[0].len(), columns[1].len(), "number of advice cells of one is_less_than");
        for (i, (x, y)) in columns[0].iter().zip(columns[1].iter()).enumerate() {
            assert_eq!(x, y, "advice cell {i} of is_less_than: the trace order differs from halo2-base's");
        }
        assert_eq!(copies[0], copies[1], "number of copy constraints");
        assert_eq!(lookups[0], lookups[1], "cells known to the copy manager (lookup registrations included)");
    }
}

#[cfg(feature = "reference-gadget")]
mod with_reference_gadget {
    use super::*;
    use indexed_merkle_tree_halo2::indexed_merkle_tree::{insert_leaf, IndexedMerkleTreeLeaf};

    fn run_insertions(depth: usize, k: usize, lookup_bits: usize, vals: &[Fr]) {
        let mut chip = IndexedMerkleTreeChip::new(depth, 1u64 << depth.min(20));
        for v in vals {
            // one insertion per circuit, as the reference's tests do (`:715-803`)
            let w = chip.insert(&[*v]).remove(0);
            let rows = chip.insert_traces(&[w.clone()]).remove(0);
            base_test().k(k as u32).lookup_bits(lookup_bits).expect_satisfied(true).run(|ctx, range| {
                let hasher = TracedPoseidonHasher::<Fr>::new(ctx);
                chip.prime_insert_traces(&hasher, &rows);
                let a = chip.assign_insert(ctx, &w);
                let low_leaf = IndexedMerkleTreeLeaf::new(a.low_leaf[0], a.low_leaf[1], a.low_leaf[2]);
                let new_leaf = IndexedMerkleTreeLeaf::new(a.new_leaf[0], a.new_leaf[1], a.new_leaf[2]);
                insert_leaf::<Fr, T, RATE>(
                    ctx,
                    range,
                    &hasher,
                    &a.old_root,
                    &low_leaf,
                    &a.low_leaf_proof,
                    &a.low_leaf_proof_helper,
                    &a.new_root,
                    &new_leaf,
                    &a.new_leaf_index,
                    &a.new_leaf_proof,
                    &a.new_leaf_proof_helper,
                    &a.is_new_leaf_largest,
                );
                assert_eq!(hasher.pending(), 0, "insert_leaf consumed {} traces fewer than were queued", hasher.pending());
            });
        }
    }

    /// the reference's multiple-round value sequence (`:699-713`) at its own circuit size (`:747-749`)
    #[test]
    fn insert_leaf_rounds_are_satisfied() {
        let vals: Vec<Fr> = [30u64, 10, 20, 5, 50, 35].iter().map(|v| Fr::from(*v)).collect();
        run_insertions(3, 19, 18, &vals);
    }

    /// BASELINE config 5: a depth-32 insertion (158 251 Poseidon witnesses) in a k = 17 circuit
    #[test]
    fn insert_leaf_at_depth_32_fits_k17() {
        let vals: Vec<Fr> = [7u64, 3, 1u64 << 40].iter().map(|v| Fr::from(*v)).collect();
        run_insertions(32, 17, 16, &vals);
    }

    /// non-membership alone (`verify_non_inclusion`, `:127-229`) through the chip's own priming call
    #[test]
    fn non_inclusion_is_satisfied() {
        use indexed_merkle_tree_halo2::indexed_merkle_tree::verify_non_inclusion;
        let mut chip = IndexedMerkleTreeChip::new(8, 256);
        chip.insert(&[Fr::from(30), Fr::from(10), Fr::from(20)]);
        let ws = chip.non_inclusion(&[Fr::from(15), Fr::from(99)]);
        base_test().k(16).lookup_bits(15).expect_satisfied(true).run(|ctx, range| {
            let hasher = TracedPoseidonHasher::<Fr>::new(ctx);
            chip.prime_non_inclusion_traces(&hasher, &ws);
            for w in &ws {
                let root = ctx.load_witness(w.root);
                let leaf = IndexedMerkleTreeLeaf::new(ctx.load_witness(w.low_leaf[0]), ctx.load_witness(w.low_leaf[1]), ctx.load_witness(w.low_leaf[2]));
                let proof: Vec<_> = w.low_leaf_proof.iter().map(|v| ctx.load_witness(*v)).collect();
                let helper: Vec<_> = w.low_leaf_proof_helper.iter().map(|v| ctx.load_witness(*v)).collect();
                let value = ctx.load_witness(w.new_leaf_value);
                let largest = ctx.load_witness(Fr::from(w.is_new_leaf_largest));
                verify_non_inclusion::<Fr, T, RATE>(ctx, range, &hasher, &root, &leaf, &proof, &helper, &value, &largest);
            }
            assert_eq!(hasher.pending(), 0);
        });
    }
}

/// The owner constraint of the SUBTREE layout (`chip::constrain_owner_subtree`) decomposes the value canonically: the
/// honest limbs of a small `v` are accepted, the limbs of `v + p` -- the same field element, another residue mod 2^k
/// since p is odd -- are refused.  (A bare `range.div_mod(v, 2^k, 254)` accepts both: halo2-base's div_mod is sound
/// only below the field's 253-bit capacity.)  Needs neither the GPU nor the reference crate.
#[test]
fn owner_constraint_refuses_the_limbs_of_v_plus_p() {
    use halo2_base::utils::{fe_to_biguint, modulus};
    use imt_hip::chip::constrain_owner_subtree_with;
    use num_bigint::BigUint;
    let k = 3usize;
    let v = Fr::from(10u64);                              // list 10 mod 8 = 2 owns it
    let helpers_for = |subtree: u64| -> Vec<Fr> { (0..8).map(|l| if l >= 5 { Fr::from(1 - ((subtree >> (l - 5)) & 1)) } else { Fr::from(1u64) }).collect() };
    let one = BigUint::from(1u64);
    let mask = (&one << 128) - &one;
    // honest: limbs of 10, subtree 2
    base_test().k(12).lookup_bits(8).expect_satisfied(true).run(|ctx, range| {
        let value = ctx.load_witness(v);
        let helper: Vec<_> = helpers_for(2).into_iter().map(|h| ctx.load_witness(h)).collect();
        constrain_owner_subtree_with(ctx, range, &value, &helper, k, None);
    });
    // forged: limbs of 10 + p (fits 254 bits), whose residue mod 8 is (10 + 1) mod 8 = 3: claims subtree 3
    let forged = fe_to_biguint(&v) + modulus::<Fr>();
    assert_eq!((&forged & BigUint::from(7u64)), BigUint::from(3u64));
    base_test().k(12).lookup_bits(8).expect_satisfied(false).run(|ctx, range| {
        let value = ctx.load_witness(v);
        let helper: Vec<_> = helpers_for(3).into_iter().map(|h| ctx.load_witness(h)).collect();
        constrain_owner_subtree_with(ctx, range, &value, &helper, k, Some((&forged >> 128, &forged & &mask)));
    });
}
