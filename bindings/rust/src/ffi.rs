//! `extern "C"` declarations of every export of `include/imt.h` (libimt_hip.so).
//!
//! One declaration per C prototype, same order as the header.  tests/test_rust_binding.py parses both files and
//! fails when a name, the number of arguments, an integer width or a constant differs.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_double, c_int, c_long, c_uint, c_void};

#[repr(C)]
pub struct imt_ctx {
    _opaque: [u8; 0],
}
#[repr(C)]
pub struct imt_tree {
    _opaque: [u8; 0],
}
#[repr(C)]
pub struct imt_itree {
    _opaque: [u8; 0],
}

/// `imt_insert_out`: outputs of a batch insertion; every pointer may be null.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct imt_insert_out {
    pub low_index: *mut u64,
    pub low_leaf: *mut c_void,
    pub is_largest: *mut u8,
    pub old_root: *mut c_void,
    pub interim_root: *mut c_void,
    pub new_root: *mut c_void,
    pub new_leaf: *mut c_void,
    pub low_sib: *mut c_void,
    pub new_sib: *mut c_void,
}

#[repr(C)]
pub struct imt_sliced {
    _opaque: [u8; 0],
}
#[repr(C)]
pub struct imt_transport {
    _opaque: [u8; 0],
}

/// `imt_transport_ops`: a caller-supplied collective for `imt_sliced_*` (NCCL semantics, stream-ordered).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct imt_transport_ops {
    pub self_: *mut c_void,
    pub all_gather: Option<unsafe extern "C" fn(self_: *mut c_void, channel: c_int, buffer: c_int, send: *const c_void, recv: *mut c_void, bytes: usize, hip_stream: *mut c_void) -> c_int>,
    pub destroy: Option<unsafe extern "C" fn(self_: *mut c_void)>,
}

/// `imt_sliced_info`
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct imt_sliced_info {
    pub world: c_int,
    pub n_local: c_int,
    pub lag: c_int,
    pub period: c_int,
    pub gathers_per_round: c_int,
    pub round_ticks: c_int,
    pub rounds_in_flight: c_int,
    pub payload_bytes: usize,
    pub rounds: u64,
    pub collectives: u64,
    pub bytes_gathered: u64,
    pub host_issue_ms: c_double,
    pub host_wait_ms: c_double,
    pub placement: c_int,
    pub hw_queues: c_int,
    pub comm_streams: c_int,
    pub streams_recreated: c_int,
    pub queue_map: [[c_int; 4]; 3],
    pub pools: c_int,
}

/// `imt_column_segment`: a stretch of insert_leaf's advice column (imt_insert_column_segments).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct imt_column_segment {
    pub kind: u32,
    pub arity: u32,
    pub first_row: u64,
    pub n_rows: u64,
}

/// `imt_trace_cell`: one cell of the advice column of a hash (imt_hash_trace_layout).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct imt_trace_cell {
    pub kind: u8,
    pub gate: u8,
    pub region: u16,
    pub index: u32,
}

pub const IMT_OK: c_int = 0;
pub const IMT_ERR_NO_LEAVES: c_int = -1;
pub const IMT_ERR_ODD_LEAVES: c_int = -2;
pub const IMT_ERR_NOT_POW2: c_int = -3;
pub const IMT_ERR_RANGE: c_int = -4;
pub const IMT_ERR_NONCANONICAL: c_int = -5;
pub const IMT_ERR_ALLOC: c_int = -6;
pub const IMT_ERR_NO_DEVICE: c_int = -7;
pub const IMT_ERR_HIP: c_int = -8;
pub const IMT_ERR_ARG: c_int = -9;
pub const IMT_ERR_VALUE: c_int = -10;
pub const IMT_ERR_FULL: c_int = -11;
pub const IMT_ERR_INTERNAL: c_int = -12;
pub const IMT_ERR_TIMEOUT: c_int = -13;

pub const IMT_FMT_CANONICAL: c_uint = 0;
pub const IMT_FMT_MONT256: c_uint = 1;
pub const IMT_FMT_DEVICE: c_uint = 2;
pub const IMT_FMT_MASK: c_uint = 3;
pub const IMT_DEVICE_PTRS: c_uint = 0x10;
pub const IMT_SIB_ITEM_MAJOR: c_uint = 0x20;
pub const IMT_TRACE_ITEM_MAJOR: c_uint = 0x20;
pub const IMT_ROOT_PER_ITEM: c_uint = 0x40;
pub const IMT_PIPELINE: c_uint = 0x80;
pub const IMT_HOST_PREP: c_uint = 0x100;
pub const IMT_INPUTS_READY: c_uint = 0x200;

pub const IMT_F_RANGE_PRED: u8 = 0x01;
pub const IMT_F_LOW_IN_ROOT: u8 = 0x02;
pub const IMT_F_LOW_LT_NEW: u8 = 0x04;
pub const IMT_F_ZERO_SLOT: u8 = 0x08;
pub const IMT_F_NEXT_VAL: u8 = 0x10;
pub const IMT_F_NEXT_IDX: u8 = 0x20;
pub const IMT_F_NEW_ROOT: u8 = 0x40;
pub const IMT_F_BAD_BIT: u8 = 0x80;

pub const IMT_PROF_LEAVES: usize = 0;
pub const IMT_PROF_INDEX: usize = 1;
pub const IMT_PROF_LEVEL: usize = 2;
pub const IMT_PROF_TOP: usize = 3;
pub const IMT_PROF_WRITEBACK: usize = 4;
pub const IMT_PROF_HOST: usize = 5;
pub const IMT_PROF_CLASSES: usize = 6;

pub const IMT_OPT_COOP_MAX_EVENTS: c_int = 1;
pub const IMT_SEG_GLUE: u32 = 0;
pub const IMT_SEG_HASH: u32 = 1;
pub const IMT_SLICED_ROUNDS: usize = 4;
pub const IMT_RCCL_UNIQUE_ID_BYTES: usize = 128;
pub const IMT_TRANSPORT_OPT_TIMEOUT_MS: c_int = 1;
pub const IMT_TRANSPORT_OPT_HOST_POLL: c_int = 2;
pub const IMT_SLICED_OPT_COMM_STREAMS: c_int = 1;
pub const IMT_SLICED_OPT_COMM_PRIORITY: c_int = 2;
pub const IMT_SLICED_OPT_ROUND_PRIORITIES: c_int = 3;
pub const IMT_SLICED_OPT_APPLY_STREAMS: c_int = 4;
pub const IMT_SLICED_OPT_PREP_STREAM: c_int = 5;
pub const IMT_SLICED_OPT_VERIFY_QUEUES: c_int = 6;
pub const IMT_SLICED_OPT_WATCHDOG_MS: c_int = 7;
pub const IMT_SLICED_OPT_TIMING: c_int = 8;
pub const IMT_SLICED_OPT_COMM_PLACEMENT: c_int = 9;
pub const IMT_SLICED_OPT_POOLS: c_int = 10;
pub const IMT_SLICED_OPT_RESET: c_int = 11;
pub const IMT_SLICED_PLACEMENT_UNVERIFIED: c_int = 0;
pub const IMT_SLICED_PLACEMENT_AS_CREATED: c_int = 1;
pub const IMT_SLICED_PLACEMENT_REPAIRED: c_int = 2;
pub const IMT_SLICED_PLACEMENT_DEGRADED: c_int = 3;

pub const IMT_CELL_CONST: u8 = 0;
pub const IMT_CELL_INPUT: u8 = 1;
pub const IMT_CELL_INIT: u8 = 2;
pub const IMT_CELL_WITNESS: u8 = 3;
pub const IMT_CELL_COPY: u8 = 4;

extern "C" {
    // ---- context
    pub fn imt_ctx_create(device: c_int, out: *mut *mut imt_ctx) -> c_int;
    pub fn imt_ctx_destroy(ctx: *mut imt_ctx);
    pub fn imt_last_error(ctx: *const imt_ctx) -> *const c_char;
    pub fn imt_ctx_set_stream(ctx: *mut imt_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn imt_ctx_sync(ctx: *mut imt_ctx) -> c_int;
    pub fn imt_host_alloc(ctx: *mut imt_ctx, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn imt_host_free(ctx: *mut imt_ctx, ptr: *mut c_void) -> c_int;
    pub fn imt_ctx_set_option(ctx: *mut imt_ctx, option: c_int, value: u64) -> c_int;
    pub fn imt_version() -> *const c_char;
    pub fn imt_measure_mad_peak(ctx: *mut imt_ctx, gmads: *mut c_double) -> c_int;
    pub fn imt_profile_enable(ctx: *mut imt_ctx, on: c_int) -> c_int;
    pub fn imt_profile_read(ctx: *mut imt_ctx, out: *mut c_double) -> c_int;

    // ---- a1 / a10: batched hashes
    pub fn imt_hash2_batch(ctx: *mut imt_ctx, input: *const c_void, out: *mut c_void, n: usize, flags: c_uint) -> c_int;
    pub fn imt_hash3_batch(ctx: *mut imt_ctx, input: *const c_void, out: *mut c_void, n: usize, flags: c_uint) -> c_int;
    pub fn imt_permute_batch(ctx: *mut imt_ctx, input: *const c_void, out: *mut c_void, n: usize, flags: c_uint) -> c_int;

    // ---- f1: witness trace of hash_fix_len_array
    pub fn imt_hash_trace_rows(arity: c_int) -> usize;
    pub fn imt_hash_trace_batch(ctx: *mut imt_ctx, input: *const c_void, arity: c_int, n: usize, trace: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_path_trace_batch(ctx: *mut imt_ctx, leaf: *const c_void, leaf3: *const c_void, index: *const u64, sib: *const c_void, depth: c_uint, n: usize, trace: *mut c_void, root_out: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_insert_trace_rows(depth: c_uint) -> usize;
    pub fn imt_insert_trace_batch(ctx: *mut imt_ctx, low_leaf: *const c_void, low_index: *const u64, low_sib: *const c_void, new_leaf: *const c_void, new_index: *const u64, new_path_index: *const u64, new_sib: *const c_void, depth: c_uint, n: usize, trace: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_hash_trace_layout(ctx: *mut imt_ctx, arity: c_int, cells: *mut imt_trace_cell, cells_cap: usize, n_cells: *mut usize, constants: *mut c_void, const_cap: usize, n_constants: *mut usize, out_row: *mut u32, flags: c_uint) -> c_int;

    // ---- a2 / a3 / a4: dense native tree
    // ---- f3: the rest of insert_leaf's advice column
    pub fn imt_less_than_trace_rows(lookup_bits: c_uint) -> usize;
    pub fn imt_less_than_trace_batch(ctx: *mut imt_ctx, a: *const c_void, b: *const c_void, n: usize, lookup_bits: c_uint, trace: *mut c_void, lt_out: *mut u8, flags: c_uint) -> c_int;
    pub fn imt_less_than_trace_layout(ctx: *mut imt_ctx, lookup_bits: c_uint, cells: *mut imt_trace_cell, cells_cap: usize, n_cells: *mut usize, constants: *mut c_void, const_cap: usize, n_constants: *mut usize, out_row: *mut u32, flags: c_uint) -> c_int;
    pub fn imt_less_than_lookup_rows(lookup_bits: c_uint, rows: *mut u32, cap: usize, n_rows: *mut usize) -> c_int;
    pub fn imt_insert_gadget_rows(depth: c_uint, lookup_bits: c_uint) -> usize;
    pub fn imt_insert_gadget_lookup_rows(depth: c_uint, lookup_bits: c_uint, rows: *mut u32, cap: usize, n_rows: *mut usize) -> c_int;
    pub fn imt_insert_gadget_trace_batch(ctx: *mut imt_ctx, low_leaf: *const c_void, low_index: *const u64, low_sib: *const c_void, new_leaf: *const c_void, new_index: *const u64, new_path_index: *const u64, new_sib: *const c_void, is_largest: *const u8, depth: c_uint, lookup_bits: c_uint, n: usize, trace: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_insert_column_segments(depth: c_uint, lookup_bits: c_uint, segs: *mut imt_column_segment, cap: usize, n_segs: *mut usize) -> c_int;
    pub fn imt_non_inclusion_gadget_rows(depth: c_uint, lookup_bits: c_uint) -> usize;
    pub fn imt_non_inclusion_gadget_trace_batch(ctx: *mut imt_ctx, low_leaf: *const c_void, low_index: *const u64, low_sib: *const c_void, new_val: *const c_void, is_largest: *const u8, depth: c_uint, lookup_bits: c_uint, n: usize, trace: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_non_inclusion_column_segments(depth: c_uint, lookup_bits: c_uint, segs: *mut imt_column_segment, cap: usize, n_segs: *mut usize) -> c_int;

    pub fn imt_tree_new(ctx: *mut imt_ctx, leaves: *const c_void, n_leaves: usize, flags: c_uint, out: *mut *mut imt_tree) -> c_int;
    pub fn imt_tree_free(t: *mut imt_tree);
    pub fn imt_tree_num_levels(t: *const imt_tree) -> usize;
    pub fn imt_tree_get_root(t: *mut imt_tree, root: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_tree_get_proof(t: *mut imt_tree, index: usize, proof: *mut c_void, helper: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_tree_get_proof_batch(t: *mut imt_tree, index: *const u64, n: usize, proof: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_tree_get_level(t: *mut imt_tree, level: usize, out: *mut c_void, n_out: *mut usize, flags: c_uint) -> c_int;
    pub fn imt_tree_build(ctx: *mut imt_ctx, leaves: *const c_void, n_leaves: usize, levels: *mut c_void, root: *mut c_void, flags: c_uint) -> c_int;

    // ---- a5 / a7 / a8 / a9: batched path recompute
    pub fn imt_path_root_batch(ctx: *mut imt_ctx, leaf: *const c_void, index: *const u64, sib: *const c_void, depth: c_uint, n: usize, root_out: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_compute_merkle_root_batch(ctx: *mut imt_ctx, leaf: *const c_void, helper_mask: *const u64, sib: *const c_void, depth: c_uint, n: usize, root_out: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_verify_proof_batch(ctx: *mut imt_ctx, leaf: *const c_void, index: *const u64, root: *const c_void, sib: *const c_void, depth: c_uint, n: usize, ok_out: *mut u8, flags: c_uint) -> c_int;

    // ---- a11 / a12 / a13: batched non-membership
    pub fn imt_non_membership_batch(ctx: *mut imt_ctx, root: *const c_void, low_leaf: *const c_void, low_index: *const u64, low_sib: *const c_void, depth: c_uint, new_val: *const c_void, is_largest: *const u8, n: usize, fail_out: *mut u8, root_out: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_split128_batch(ctx: *mut imt_ctx, vals: *const c_void, q: *mut c_void, r: *mut c_void, n: usize, flags: c_uint) -> c_int;

    // ---- a14: batched insert_leaf witness
    pub fn imt_insert_witness_batch(ctx: *mut imt_ctx, old_root: *const c_void, low_leaf: *const c_void, low_index: *const u64, low_sib: *const c_void, new_root: *const c_void, new_leaf: *const c_void, new_index: *const u64, new_path_index: *const u64, new_sib: *const c_void, is_largest: *const u8, depth: c_uint, n: usize, fail_out: *mut u8, trace_out: *mut c_void, flags: c_uint) -> c_int;

    // ---- a15: stateful indexed tree
    pub fn imt_itree_new(ctx: *mut imt_ctx, depth: c_uint, capacity: u64, out: *mut *mut imt_itree) -> c_int;
    pub fn imt_itree_free(t: *mut imt_itree);
    pub fn imt_itree_size(t: *const imt_itree) -> u64;
    pub fn imt_itree_root(t: *mut imt_itree, root: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_itree_root_lagged(t: *mut imt_itree, lag: c_uint, root: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_itree_insert_batch(t: *mut imt_itree, vals: *const c_void, n: usize, out: *const imt_insert_out, flags: c_uint) -> c_int;
    pub fn imt_itree_get_proof_batch(t: *mut imt_itree, index: *const u64, n: usize, sib: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_itree_get_leaves(t: *mut imt_itree, index: *const u64, n: usize, preimage: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_itree_load(t: *mut imt_itree, preimages: *const c_void, n: u64, flags: c_uint) -> c_int;
    pub fn imt_itree_find_low_batch(t: *mut imt_itree, vals: *const c_void, n: usize, low_index: *mut u64, flags: c_uint) -> c_int;
    pub fn imt_itree_non_membership_witness(t: *mut imt_itree, vals: *const c_void, n: usize, low_index: *mut u64, low_leaf: *mut c_void, is_largest: *mut u8, low_sib: *mut c_void, flags: c_uint) -> c_int;

    // ---- e: the tree as one subtree of a deeper tree
    pub fn imt_itree_set_placement(t: *mut imt_itree, global_depth: c_uint, subtree_index: u64) -> c_int;
    pub fn imt_itree_set_value_partition(t: *mut imt_itree, modulus: u32, residue: u32) -> c_int;
    pub fn imt_itree_lift_batch(t: *mut imt_itree, roots_before: *const c_void, roots_after: *const c_void, n_subtrees: usize, n: usize, out: *const imt_insert_out, flags: c_uint) -> c_int;

    // ---- e: one tree on several GPUs, single sorted list
    pub fn imt_itree_batch_begin(t: *mut imt_itree, vals: *const c_void, n: usize, flags: c_uint, events_out: *mut u32, l0_out: *mut u32) -> c_int;
    pub fn imt_itree_batch_leaves(t: *mut imt_itree, val0: *mut c_void, k_begin: u32, k_count: u32) -> c_int;
    pub fn imt_itree_batch_level(t: *mut imt_itree, level: c_uint, val_in: *const c_void, val_out: *mut c_void, k_begin: u32, k_count: u32) -> c_int;
    pub fn imt_itree_batch_top(t: *mut imt_itree, val_l0: *const c_void, e_begin: u32, e_count: u32, roots: *mut c_void, top_path: *mut c_void) -> c_int;
    pub fn imt_itree_batch_extract(t: *mut imt_itree, val_levels: *const *const c_void, roots: *const c_void, ins_begin: u32, ins_count: u32, out: *const imt_insert_out, flags: c_uint) -> c_int;
    pub fn imt_itree_batch_end(t: *mut imt_itree, val_levels: *const *const c_void, top_path: *const c_void) -> c_int;
    pub fn imt_itree_batch_abort(t: *mut imt_itree) -> c_int;

    // ---- e: one tree on several GPUs, single sorted list, time-sliced: schedule + collective inside the library
    pub fn imt_transport_custom_create(ops: *const imt_transport_ops, out: *mut *mut imt_transport) -> c_int;
    pub fn imt_transport_local_create(out: *mut *mut imt_transport) -> c_int;
    pub fn imt_rccl_get_unique_id(id: *mut c_void) -> c_int;
    pub fn imt_transport_rccl_create(ctx: *mut imt_ctx, unique_ids: *const c_void, n_comms: c_int, world: c_int, rank: c_int, out: *mut *mut imt_transport) -> c_int;
    pub fn imt_transport_rccl_adopt(nccl_comms: *const *mut c_void, n_comms: c_int, out: *mut *mut imt_transport) -> c_int;
    pub fn imt_rccl_library(version_out: *mut c_int) -> *const c_char;
    pub fn imt_transport_ipc_blob_bytes() -> usize;
    pub fn imt_transport_ipc_create(ctx: *mut imt_ctx, world: c_int, rank: c_int, depth: c_uint, max_slice: usize, lag: c_int, out: *mut *mut imt_transport, blob_out: *mut c_void) -> c_int;
    pub fn imt_transport_ipc_connect(tp: *mut imt_transport, all_blobs: *const c_void) -> c_int;
    pub fn imt_transport_destroy(tp: *mut imt_transport) -> c_int;
    pub fn imt_transport_set_option(tp: *mut imt_transport, option: c_int, value: c_long) -> c_int;
    pub fn imt_transport_all_gather(tp: *mut imt_transport, send: *const c_void, recv: *mut c_void, bytes: usize, hip_stream: *mut c_void) -> c_int;
    pub fn imt_transport_poll_error(tp: *mut imt_transport) -> c_int;
    pub fn imt_transport_last_error(tp: *const imt_transport) -> *const c_char;
    pub fn imt_sliced_create(trees: *const *mut imt_itree, n_local: c_int, world: c_int, first_rank: c_int, tp: *mut imt_transport, max_slice: usize, lag: c_int, out: *mut *mut imt_sliced) -> c_int;
    pub fn imt_sliced_step(w: *mut imt_sliced, vals: *const c_void, n: usize, outs: *const imt_insert_out, flags: c_uint, round_out: *mut u64) -> c_int;
    pub fn imt_sliced_wait(w: *mut imt_sliced, local_rank: c_int, round: u64) -> c_int;
    pub fn imt_sliced_flush(w: *mut imt_sliced) -> c_int;
    pub fn imt_sliced_set_option(w: *mut imt_sliced, option: c_int, value: c_long) -> c_int;
    pub fn imt_sliced_get_info(w: *const imt_sliced, out: *mut imt_sliced_info) -> c_int;
    pub fn imt_sliced_dump(w: *mut imt_sliced, out: *mut c_char, cap: usize) -> c_int;
    pub fn imt_sliced_last_error(w: *const imt_sliced) -> *const c_char;
    pub fn imt_sliced_destroy(w: *mut imt_sliced);
    // the building blocks (a host with its own scheduler)
    pub fn imt_itree_slice_payload_bytes(n: usize) -> usize;
    pub fn imt_itree_slice_unit_bytes(t: *const imt_itree, size_before: u64, n: usize, unit: c_uint) -> usize;
    pub fn imt_itree_slice_prepare(t: *mut imt_itree, vals: *const c_void, n_before: usize, n_own: usize, n_after: usize, out: *const imt_insert_out, flags: c_uint, slice_out: *mut c_int, l0_out: *mut u32) -> c_int;
    pub fn imt_itree_slice_unit(t: *mut imt_itree, slice: c_int, unit: c_uint, payload: *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn imt_itree_slice_apply(t: *mut imt_itree, size_before: u64, n: usize, unit: c_uint, payload: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn imt_itree_slice_apply_gathered(t: *mut imt_itree, gathered: *const c_void, stride: usize, count: usize, size_before: *const u64, n: *const u64, unit: *const i32, hip_stream: *mut c_void) -> c_int;

    // ---- e: multi-GPU helpers
    pub fn imt_combine_subtree_roots(ctx: *mut imt_ctx, sub_roots: *const c_void, n_roots: usize, sub_height: c_uint, depth: c_uint, root: *mut c_void, flags: c_uint) -> c_int;
    pub fn imt_zero_hashes(ctx: *mut imt_ctx, depth: c_uint, out: *mut c_void, flags: c_uint) -> c_int;
}
