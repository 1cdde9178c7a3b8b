/*
 * imt.h -- C ABI of libimt_hip.so: the MI355X (gfx950) indexed-Merkle-tree hot path.
 *
 * Drop-in boundary for the one data-parallel path of
 * aerius-labs/indexed-merkle-tree-halo2 (reference @ /root/reference): the Poseidon
 * (T=3, RATE=2, R_F=8, R_P=57 over bn256::Fr) hashes and the depth-d Merkle-path
 * recomputes behind
 *     src/utils.rs                  IndexedMerkleTree::{new,get_root,get_proof,verify_proof}
 *     src/indexed_merkle_tree.rs    verify_non_inclusion (:127), insert_leaf (:231),
 *                                   compute_merkle_root (:78), and the test module's
 *                                   update_idx_leaf / hash_nullifier_pre_images (:632-671)
 * Each entry point names the reference interface it replaces.  INTEGRATION.md shows the
 * Rust `extern "C"` binding a maintainer would add.
 *
 * Conventions
 *  - every function returns IMT_OK (0) or a negative IMT_ERR_* code; nothing throws or
 *    aborts across the ABI.  imt_last_error(ctx) gives a message for the last failure.
 *  - a field element is 32 bytes.  IMT_FMT_CANONICAL: little-endian integer < p
 *    (halo2curves Fr::to_repr()).  IMT_FMT_MONT256: the in-memory [u64;4] of a
 *    halo2curves bn256::Fr (Montgomery, R = 2^256), for zero-copy from Rust.
 *    IMT_FMT_DEVICE: the library's own resident format (Montgomery R = 2^261, reduced,
 *    packed 8 x u32); only meaningful for buffers produced by this library.
 *  - buffers are caller-allocated.  By default pointers are HOST pointers and the call
 *    is synchronous.  With IMT_DEVICE_PTRS all data pointers are device pointers on the
 *    context's device, the work is enqueued on the context's stream and the call returns
 *    without synchronising; input errors are then reported by imt_ctx_sync().
 *  - sibling arrays are LEVEL-MAJOR by default: sib[level][item] (coalesced on the
 *    device).  IMT_SIB_ITEM_MAJOR selects sib[item][level], the order of the reference's
 *    per-proof Vec<F> (src/utils.rs:63-85).
 *  - an imt_ctx belongs to one host thread at a time (the reference's objects are
 *    single-owner through &mut borrows: src/utils.rs:6-10).
 *  - there is NO CPU fallback: without a usable HIP device imt_ctx_create fails with
 *    IMT_ERR_NO_DEVICE.
 */
#ifndef IMT_H
#define IMT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMT_OK 0
#define IMT_ERR_NO_LEAVES (-1)    /* "Cannot create Merkle Tree with no leaves"  src/utils.rs:24-26 */
#define IMT_ERR_ODD_LEAVES (-2)   /* "Leaves must be even"                       src/utils.rs:34-36 */
#define IMT_ERR_NOT_POW2 (-3)     /* the index panic at src/utils.rs:45 for an even non-power-of-two */
#define IMT_ERR_RANGE (-4)        /* an index / depth / capacity out of range (reference: slice panic) */
#define IMT_ERR_NONCANONICAL (-5) /* a field element >= p (Fr::from_repr would return None) */
#define IMT_ERR_ALLOC (-6)
#define IMT_ERR_NO_DEVICE (-7)    /* no HIP device / runtime: this library has no CPU path */
#define IMT_ERR_HIP (-8)          /* a HIP call failed; see imt_last_error */
#define IMT_ERR_ARG (-9)          /* NULL pointer or inconsistent arguments */
#define IMT_ERR_VALUE (-10)       /* value 0 or already present: the reference's circuit panics at
                                     src/indexed_merkle_tree.rs:190 for such an insertion */
#define IMT_ERR_FULL (-11)        /* indexed tree capacity exhausted */
#define IMT_ERR_INTERNAL (-12)
#define IMT_ERR_TIMEOUT (-13)     /* a host-side wait inside imt_sliced_* ran into the world's watchdog (a peer died or
                                     hangs); the world's state has been written to stderr, see IMT_SLICED_OPT_WATCHDOG_MS.
                                     AFTER IT: something of the world may still run on the device and may never end (a
                                     collective whose peer is gone has no time limit of its own).  imt_sliced_destroy and
                                     imt_transport_destroy then return without waiting for the device -- the world's device
                                     buffers, streams and communicators are LEFT ALLOCATED -- but imt_itree_destroy /
                                     imt_ctx_destroy / hipFree would wait.  The process should report and EXIT non-zero
                                     (without those calls) and recovery (reload from a checkpoint) should happen in a
                                     freshly started process -- never by exec'ing over one that has touched the GPU. */

/* flags */
#define IMT_FMT_CANONICAL 0u
#define IMT_FMT_MONT256 1u
#define IMT_FMT_DEVICE 2u
#define IMT_FMT_MASK 3u
#define IMT_DEVICE_PTRS 0x10u
#define IMT_SIB_ITEM_MAJOR 0x20u
#define IMT_ROOT_PER_ITEM 0x40u   /* root argument is root[n][32] instead of one root[32] */
#define IMT_HOST_PREP 0x100u      /* imt_itree_insert_batch: do the low-leaf search, the event preimages and the
                                     event ordering on the host instead of on the GPU (the default keeps a
                                     device-resident sorted index and the host only launches kernels).
                                     Same results either way; the two can be mixed on one tree. */
#define IMT_PIPELINE 0x80u        /* imt_itree_insert_batch with IMT_DEVICE_PTRS only: consecutive batches run on
                                     internal streams (up to four in flight) one tree level apart, so their hash
                                     kernels share the GPU.
                                     The outputs of such a batch are ordered by imt_ctx_sync() (or by the next
                                     imt_itree_root / get_proof / non-pipelined call on the tree), not by
                                     the context's stream. */

#define IMT_INPUTS_READY 0x200u   /* imt_itree_insert_batch with IMT_DEVICE_PTRS: `vals` and the hash-free output buffers
                                     (low_index, is_largest, low_leaf, new_leaf) are idle -- nothing enqueued on the
                                     context's stream still writes `vals` or reads those outputs.  Without it the
                                     batch's preparation is ordered behind the context's stream first, which costs
                                     nothing unless that stream holds long-running work. */

/* failure bits written per item by the relation checkers; each is one constraint or
 * assert of the reference (src/indexed_merkle_tree.rs) */
#define IMT_F_RANGE_PRED 0x01   /* select(is_largest, next_val==0, new<next_val) == 1   :182-191 */
#define IMT_F_LOW_IN_ROOT 0x02  /* low leaf hashes up to the given root                  :196-204 */
#define IMT_F_LOW_LT_NEW 0x04   /* low_leaf.val < new value                             :206-228 */
#define IMT_F_ZERO_SLOT 0x08    /* the zero leaf sits at the new slot of the interim root :286-294 */
#define IMT_F_NEXT_VAL 0x10     /* new_leaf.next_val == low_leaf.next_val               :296 */
#define IMT_F_NEXT_IDX 0x20     /* new_leaf.next_idx == low_leaf.next_idx               :297 */
#define IMT_F_NEW_ROOT 0x40     /* new root recomputes                                  :305-313 */
#define IMT_F_BAD_BIT 0x80      /* a flag that must be 0/1 is not (gate.assert_bit      :41,54) */

typedef struct imt_ctx imt_ctx;
typedef struct imt_tree imt_tree;     /* dense tree: IndexedMerkleTree<'a,F,T,RATE> src/utils.rs:6-10 */
typedef struct imt_itree imt_itree;   /* depth-d append-only indexed tree (sparse storage) */

/* ---- context ------------------------------------------------------------------- */
/* Builds the Poseidon tables (Poseidon::<Fr,3,2>::new(8,57), src/indexed_merkle_tree.rs:370)
 * and uploads them to `device`.  device < 0 or no GPU -> IMT_ERR_NO_DEVICE. */
int imt_ctx_create(int device, imt_ctx **out);
void imt_ctx_destroy(imt_ctx *ctx);
const char *imt_last_error(const imt_ctx *ctx);
/* Use an existing hipStream_t (e.g. PyTorch's current stream); NULL = the context's own. */
int imt_ctx_set_stream(imt_ctx *ctx, void *hip_stream);
/* Wait for the stream and report deferred input errors (IMT_ERR_NONCANONICAL, ...). */
int imt_ctx_sync(imt_ctx *ctx);
/* Page-locked host memory that the device can address (hipHostMalloc).  A pointer from here may be
 * passed wherever IMT_DEVICE_PTRS expects a device pointer: the kernels then read the values and
 * write the roots and proofs straight into the caller's memory over PCIe, asynchronously and
 * pipelined like any device-pointer call, and the data is valid on the host after imt_ctx_sync().
 * This is how a host-language caller (the Rust shim of INTEGRATION.md) gets its witnesses without a
 * staging copy: measured at the same insertion rate as HBM-resident outputs.
 * Plain host pointers (no IMT_DEVICE_PTRS) stay supported and are synchronous. */
int imt_host_alloc(imt_ctx *ctx, size_t bytes, void **out);
int imt_host_free(imt_ctx *ctx, void *ptr);
/* Tuning knobs.  IMT_OPT_COOP_MAX_EVENTS: batch-insertion launches of at most this many events (2 per insertion) use
 * the latency form of the hash kernel -- four lanes per hash, 0.55x the time per launch, 2x the lane-instructions --
 * which pays while a launch leaves most of the chip idle; likewise plain hashes (imt_hash2/3_batch), path recomputes (imt_path_root_batch,
 * imt_compute_merkle_root_batch, imt_verify_proof_batch, imt_non_membership_batch, the path inputs of the trace calls)
 * of at most a quarter as many paths and imt_insert_witness_batch of at most a sixteenth as many items.  Default 16384
 * (one wave per SIMD); 0 = never.  Results are bit-identical either way. */
#define IMT_OPT_COOP_MAX_EVENTS 1
int imt_ctx_set_option(imt_ctx *ctx, int option, uint64_t value);
/* ABI / build identification, e.g. "imt-hip gfx950 r4" */
const char *imt_version(void);
/* Per-kernel timing with HIP events recorded on the context's stream around the launches of
 * imt_itree_insert_batch (used by bench.py for the roofline line; off by default).
 * imt_profile_read synchronises the stream, adds up the finished intervals, writes
 * out[2*c] = total milliseconds and out[2*c+1] = number of launches / calls for class c
 * (IMT_PROF_*), and resets the counters. */
#define IMT_PROF_LEAVES 0      /* k_sweep, leaf launches: the 3-input leaf hashes */
#define IMT_PROF_INDEX 1       /* k_merge_level + table copies: index phase, no hashing */
#define IMT_PROF_LEVEL 2       /* k_sweep, the levels below the highest meeting point: one hash per event per level */
#define IMT_PROF_TOP 3         /* k_sweep, the levels above it (every event against the empty subtree) */
#define IMT_PROF_WRITEBACK 4   /* k_writeback */
#define IMT_PROF_HOST 5        /* host side of imt_itree_insert_batch (wall time, waits excluded) */
#define IMT_PROF_CLASSES 6
/* Measures the device's v_mad_u64_u32 issue rate (8 independent chains per lane, 8 waves per SIMD): the
 * ceiling of the VALU roofline bench.py reports.  *gmads = 10^9 lane multiply-adds per second. */
int imt_measure_mad_peak(imt_ctx *ctx, double *gmads);
int imt_profile_enable(imt_ctx *ctx, int on);
int imt_profile_read(imt_ctx *ctx, double *out /*[2*IMT_PROF_CLASSES]*/);

/* ---- a1 / a10: batched hashes -------------------------------------------------- */
/* out[i] = Poseidon::update(&[in[i][0], in[i][1]]) ; squeeze_and_reset()
 * replaces src/utils.rs:46-47,96-100 and hash_fix_len_array at indexed_merkle_tree.rs:92 */
int imt_hash2_batch(imt_ctx *ctx, const void *in /*[n][2][32]*/, void *out /*[n][32]*/, size_t n,
                    unsigned flags);
/* 3-input leaf hash [val, next_val, next_idx]: indexed_merkle_tree.rs:193-194,271-275,299-303,663-668 */
int imt_hash3_batch(imt_ctx *ctx, const void *in /*[n][3][32]*/, void *out /*[n][32]*/, size_t n,
                    unsigned flags);
/* the bare permutation on [n][3] states (test hook for the round schedule) */
int imt_permute_batch(imt_ctx *ctx, const void *in /*[n][3][32]*/, void *out /*[n][3][32]*/, size_t n,
                      unsigned flags);

/* ---- f1: the witness trace of PoseidonHasher::hash_fix_len_array ---------------- */
/* The circuit recomputes every Poseidon on the CPU while it assigns (hasher.hash_fix_len_array at
 * src/indexed_merkle_tree.rs:92, :194, :271-275, :299-303).  These calls produce, on the GPU, every NEW advice value
 * the gadget assigns for a hash -- both permutations' absorb cells, S-box intermediates x^2, x^4, x^5 + c and the
 * running sums of every MDS row -- in assignment order ("trace rows"), so a chip can assign instead of recompute.
 * Order and cell structure follow the published halo2-lib v0.4.x gadget (poseidon/hasher/state.rs over GateChip's
 * vertical gate); halo2-base is not vendored in the reference, so this order is UNPINNED BY THE REFERENCE.  What pins
 * it: the output row is the hash (reference KAT), every gate of the reconstructed column holds, and the CPU oracle
 * restates it independently (oracle/trace.c).
 *   rows: 1208 for 2 inputs, 1209 for 3 (imt_hash_trace_rows); the hash itself is row rows - 4.
 *   trace layout: [rows][n][32] (row-major, coalesced on the device) or, with IMT_TRACE_ITEM_MAJOR, [n][rows][32]
 *   (one hash's rows contiguous).  With IMT_FMT_MONT256 a row is the in-memory [u64;4] of a halo2curves Fr:
 *   Witness(unsafe { transmute(row) }) needs no arithmetic on the host. */
#define IMT_TRACE_ITEM_MAJOR IMT_SIB_ITEM_MAJOR
size_t imt_hash_trace_rows(int arity);
int imt_hash_trace_batch(imt_ctx *ctx, const void *in /*[n][arity][32]*/, int arity, size_t n,
                         void *trace /*[rows][n][32]*/, unsigned flags);
/* The traces of ALL hashes of compute_merkle_root (src/indexed_merkle_tree.rs:78-96) for n paths: the leaf hash
 * (3 inputs; only when leaf3 is given instead of leaf) followed by the `depth` path hashes bottom-up, each with the
 * (left, right) inputs dual_mux selects.  trace = the blocks one after the other, [1209][n] (if leaf3) then depth x
 * [1208][n]; item-major: [n][1209 + depth * 1208] (IMT_TRACE_ITEM_MAJOR is the same bit as IMT_SIB_ITEM_MAJOR: the siblings
 * are then read item-major too, sib[item][level] -- the reference's per-proof Vec<F> -- exactly as in
 * imt_insert_trace_batch).  root_out (optional) = the recomputed roots. */
int imt_path_trace_batch(imt_ctx *ctx, const void *leaf /*[n][32] or NULL*/, const void *leaf3 /*[n][3][32] or NULL*/,
                         const uint64_t *index /*[n]*/, const void *sib, unsigned depth, size_t n, void *trace,
                         void *root_out /*[n][32] or NULL*/, unsigned flags);
/* The traces of ALL 3 + 4 * depth hashes of insert_leaf (src/indexed_merkle_tree.rs:231-314) for n insertions, in the
 * order the circuit reaches hash_fix_len_array: low leaf + its path (:193-204), rewritten low leaf {low.val, new.val,
 * new_index} + the same path (:271-284), the zero leaf's path at the new slot (:286-294; no leaf hash, the zero-leaf
 * hash is a constant), new leaf + its path (:299-312).  Inputs as imt_insert_witness_batch (what imt_itree_insert_batch
 * returned).  trace: the four blocks one after the other, each as imt_path_trace_batch lays it out; item-major
 * (IMT_TRACE_ITEM_MAJOR, which is the same bit as IMT_SIB_ITEM_MAJOR: siblings are then item-major too):
 * [n][imt_insert_trace_rows(depth)].  5.1 MB per insertion at depth 32: size the batch to the memory at hand. */
size_t imt_insert_trace_rows(unsigned depth);
int imt_insert_trace_batch(imt_ctx *ctx, const void *low_leaf /*[n][3][32]*/, const uint64_t *low_index,
                           const void *low_sib, const void *new_leaf /*[n][3][32]*/, const uint64_t *new_index,
                           const uint64_t *new_path_index /*[n] or NULL = new_index*/, const void *new_sib,
                           unsigned depth, size_t n, void *trace, unsigned flags);
/* The advice column of ONE hash, cell by cell, in assignment order: where each cell's value comes from and where
 * the vertical gates a + b*c = d start (gate = 1 on cell a).  Static per arity; host pointers only. */
#define IMT_CELL_CONST 0     /* constants[index] */
#define IMT_CELL_INPUT 1     /* copy (Existing) of hash input `index` */
#define IMT_CELL_INIT 2      /* copy of the hasher's initial-state cell `index`: 0 = 2^64, 1 and 2 = 0 */
#define IMT_CELL_WITNESS 3   /* NEW value: trace row `index` (rows appear in increasing order) */
#define IMT_CELL_COPY 4      /* copy (Existing) of trace row `index` */
typedef struct imt_trace_cell {
    uint8_t kind;            /* IMT_CELL_* */
    uint8_t gate;            /* 1: q_enable here, the gate covers this cell and the next three */
    uint16_t region;         /* 1: first cell of one ctx.assign_region call of the gadget (a gate.add / sum / mul /
                                mul_add / inner_product): copies inside a region only refer to EARLIER regions, so a
                                chip can assign region by region with Existing(..) handles it already holds */
    uint32_t index;
} imt_trace_cell;
/* cells / constants may be NULL (sizes only).  constants[n_constants][32] in the format of `flags`. */
int imt_hash_trace_layout(imt_ctx *ctx, int arity, imt_trace_cell *cells, size_t cells_cap, size_t *n_cells,
                          void *constants, size_t const_cap, size_t *n_constants, uint32_t *out_row, unsigned flags);

/* ---- f3: the rest of insert_leaf's advice column --------------------------------------------------
 * After f1 the only advice values of insert_leaf (src/indexed_merkle_tree.rs:231-314) a chip would still compute on the
 * CPU are those outside hash_fix_len_array: the two is_less_than calls of verify_non_inclusion (:180, :226 -> :98-125:
 * range.is_less_than(.,.,128) + gate.is_equal for the high and the low 128-bit limbs, then not x4, and x3, and, or),
 * is_equal(next_val, 0) with its inverse (:143), the limb loads and their mul_add checks (:169-178, :219-224), select
 * (:182-189 -> :33-45) and, per path, load_witness(leaf) + dual_mux's four values per level (:78-96 -> :47-63).  These
 * calls produce every such NEW advice value on the GPU, in assignment order.  Cell structure: the published halo2-lib
 * v0.4.x GateChip / RangeChip (sub [W a-b, b, 1, a]; mul [0, a, b, W]; mul_add [c, a, b, W]; or [W 1-b, 1, b, 1, b, a,
 * W 1-b, W out]; is_zero [W z, a, W 1/a, 1, 0, a, W z, 0]; range.is_less_than [W 2^p+a-b, b, 1, W 2^p+a, -2^p, 1, a] +
 * the limbs of the first cell as an inner product with 2^(lookup_bits i) + is_zero(top limb)).  UNPINNED BY THE
 * REFERENCE, exactly like f1 (halo2-base is not vendored): pinned by every gate of the column holding, by the results
 * equalling the reference's boolean formula, by the CPU oracle (oracle/gadget.c) and an independent big-integer model.
 * lookup_bits: the RangeChip's (the reference's tests: 18, :436); 1 .. 28.  Formats and layouts as for the f1 calls. */
size_t imt_less_than_trace_rows(unsigned lookup_bits);      /* 4 * (ceil(128 / lookup_bits) + 1) + 27; 63 for 18 */
/* rows of is_less_than(a_q, a_r, b_q, b_r) for the 256-bit values a[i], b[i] (a_q = a >> 128, ...: what
 * imt_split128_batch returns); trace [rows][n][32] or, with IMT_TRACE_ITEM_MAJOR, [n][rows][32]; lt_out[i] (optional)
 * = a[i] < b[i] */
int imt_less_than_trace_batch(imt_ctx *ctx, const void *a /*[n][32]*/, const void *b /*[n][32]*/, size_t n,
                              unsigned lookup_bits, void *trace, uint8_t *lt_out /*[n] or NULL*/, unsigned flags);
/* the column of one is_less_than, cell by cell, like imt_hash_trace_layout: IMT_CELL_INPUT index 0..3 = a_q, a_r, b_q,
 * b_r; *out_row = the row holding the result */
int imt_less_than_trace_layout(imt_ctx *ctx, unsigned lookup_bits, imt_trace_cell *cells, size_t cells_cap, size_t *n_cells,
                               void *constants, size_t const_cap, size_t *n_constants, uint32_t *out_row, unsigned flags);
/* Which of those rows the RangeChip ALSO constrains through its lookup table: range.is_less_than range-checks each
 * shifted difference by decomposing it into lookup_bits-wide limbs and adding every limb cell to the lookup
 * (add_cell_to_lookup); these are the 2 (ceil(128 / lookup_bits) + 1) limb rows, in column order.  A chip that assigns
 * the rows itself registers exactly these cells.  rows may be NULL (count only).  Arithmetic on sizes: no context. */
int imt_less_than_lookup_rows(unsigned lookup_bits, uint32_t *rows, size_t cap, size_t *n_rows);
/* ALL rows of insert_leaf outside its hashes ("glue rows"), in assignment order: is_equal(next_val, 0) [4 rows], the
 * limbs nl_q nl_r ll_q ll_r [4], their two mul_add [2], is_less_than(new, low.next_val) [K], select [3], then for the low
 * leaf's path load_witness [1] + dual_mux a-b, b-a, left, right per level [4 depth], the limbs of low.val + mul_add
 * [3], is_less_than(low.val, new) [K], and the same 1 + 4 depth rows for the rewritten low leaf's, the zero leaf's and
 * the new leaf's path: 20 + 2 K + 16 depth rows (658 at depth 32, lookup_bits 18).  Inputs as imt_insert_trace_batch
 * (what imt_itree_insert_batch returned) plus is_largest; depth >= 1.  Together with imt_insert_trace_batch this is
 * every new advice value of the call; imt_insert_column_segments tells how the two traces interleave. */
size_t imt_insert_gadget_rows(unsigned depth, unsigned lookup_bits);
/* the glue rows that are also lookup cells (imt_less_than_lookup_rows for both comparisons of the call, as glue-row
 * numbers): 4 (ceil(128 / lookup_bits) + 1) rows */
int imt_insert_gadget_lookup_rows(unsigned depth, unsigned lookup_bits, uint32_t *rows, size_t cap, size_t *n_rows);
int imt_insert_gadget_trace_batch(imt_ctx *ctx, const void *low_leaf /*[n][3][32]*/, const uint64_t *low_index,
                                  const void *low_sib, const void *new_leaf /*[n][3][32]*/, const uint64_t *new_index,
                                  const uint64_t *new_path_index /*[n] or NULL = new_index*/, const void *new_sib,
                                  const uint8_t *is_largest /*[n]*/, unsigned depth, unsigned lookup_bits, size_t n,
                                  void *trace, unsigned flags);
/* The same for ONE verify_non_inclusion call on its own (src/indexed_merkle_tree.rs:127-229 -- BASELINE config 3's
 * gadget): is_equal [4], limbs [4], mul_add [2], is_less_than(new, low.next_val) [K], select [3], the low leaf's path
 * [1 + 4 depth], the limbs of low.val + mul_add [3], is_less_than(low.val, new) [K] = 17 + 2 K + 4 depth rows per item
 * (the first rows of the insert_leaf call above, which begins with this gadget).  Inputs: what
 * imt_itree_non_membership_witness returned + the candidate values.  Its lookup cells: imt_insert_gadget_lookup_rows
 * (same row numbers).  Its hash blocks: imt_path_trace_batch(leaf3 = low_leaf, low_index, low_sib). */
size_t imt_non_inclusion_gadget_rows(unsigned depth, unsigned lookup_bits);
int imt_non_inclusion_gadget_trace_batch(imt_ctx *ctx, const void *low_leaf /*[n][3][32]*/, const uint64_t *low_index,
                                         const void *low_sib, const void *new_val /*[n][32]*/,
                                         const uint8_t *is_largest /*[n]*/, unsigned depth, unsigned lookup_bits, size_t n,
                                         void *trace, unsigned flags);
#define IMT_SEG_GLUE 0      /* n_rows rows of imt_insert_gadget_trace_batch starting at first_row */
#define IMT_SEG_HASH 1      /* one hash_fix_len_array call of `arity` inputs: rows [first_row, + n_rows) of imt_insert_trace_batch */
typedef struct imt_column_segment {
    uint32_t kind, arity;
    uint64_t first_row, n_rows;
} imt_column_segment;
/* the advice column of insert_leaf as alternating stretches of the two traces (3 + 4 depth hash segments); segs may be
 * NULL (count only) */
int imt_insert_column_segments(unsigned depth, unsigned lookup_bits, imt_column_segment *segs, size_t cap, size_t *n_segs);
/* ... and of verify_non_inclusion alone: 3 + 2 depth segments, 1 + depth of them hashes (rows of imt_path_trace_batch) */
int imt_non_inclusion_column_segments(unsigned depth, unsigned lookup_bits, imt_column_segment *segs, size_t cap,
                                      size_t *n_segs);

/* ---- a2 / a3 / a4: dense native tree ------------------------------------------- */
/* IndexedMerkleTree::new (src/utils.rs:20-57): level-by-level build on the device.
 * n_leaves == 0 -> IMT_ERR_NO_LEAVES; 1 -> root = leaf; odd -> IMT_ERR_ODD_LEAVES;
 * even but not a power of two -> IMT_ERR_NOT_POW2. */
int imt_tree_new(imt_ctx *ctx, const void *leaves /*[n][32]*/, size_t n_leaves, unsigned flags,
                 imt_tree **out);
void imt_tree_free(imt_tree *t);
size_t imt_tree_num_levels(const imt_tree *t);           /* tree.len() */
int imt_tree_get_root(imt_tree *t, void *root /*[32]*/, unsigned flags);          /* utils.rs:59-61 */
/* get_proof (src/utils.rs:63-85): siblings and helpers (helper = 1 iff the node is a left
 * child, :79) for one index; helper elements are written as field elements like the reference. */
int imt_tree_get_proof(imt_tree *t, size_t index, void *proof /*[levels-1][32]*/,
                       void *helper /*[levels-1][32] or NULL*/, unsigned flags);
int imt_tree_get_proof_batch(imt_tree *t, const uint64_t *index /*[n]*/, size_t n,
                             void *proof /*sib layout per flags*/, unsigned flags);
/* copy one level out (tree[level]); n_out receives its length */
int imt_tree_get_level(imt_tree *t, size_t level, void *out, size_t *n_out, unsigned flags);
/* one-shot: build and return root (+ all levels concatenated bottom-up if levels != NULL) */
int imt_tree_build(imt_ctx *ctx, const void *leaves, size_t n_leaves, void *levels /*[2n-1][32] or NULL*/,
                   void *root /*[32]*/, unsigned flags);

/* ---- a5 / a7 / a8 / a9: batched path recompute --------------------------------- */
/* WHEN A HOST CORE IS FASTER.  One call costs about 7 ms whatever it carries (a depth-32 path is 66 dependent
 * permutations; small calls use the latency form): one depth-32 verify_proof 7.0 ms here against 1.2 ms
 * on one host core, one insert_leaf witness check 8.5 against 2.6 ms (profiles/r05_latency_vs_cpu.txt).  The reference's
 * own call pattern -- one proof, one insertion per call (src/utils.rs:87, src/indexed_merkle_tree.rs:231) -- is therefore a
 * CPU job; these calls (and imt_non_membership_batch, imt_insert_witness_batch, imt_*_trace_batch below) win from about
 * 4 - 8 items per call and reach their throughput from a few thousand.
 *
 * root_out[i] = fold of hash2 over depth siblings, order from the parity of index>>level
 * (verify_proof, src/utils.rs:87-107; compute_merkle_root, indexed_merkle_tree.rs:78-96). */
int imt_path_root_batch(imt_ctx *ctx, const void *leaf /*[n][32]*/, const uint64_t *index /*[n]*/,
                        const void *sib, unsigned depth, size_t n, void *root_out /*[n][32]*/,
                        unsigned flags);
/* same with the circuit's helper bits: bit l of helper_mask[i] = proof_helper[l]
 * (1 = current node is the left input of the hash, dual_mux :47-63) */
int imt_compute_merkle_root_batch(imt_ctx *ctx, const void *leaf, const uint64_t *helper_mask,
                                  const void *sib, unsigned depth, size_t n, void *root_out,
                                  unsigned flags);
/* ok_out[i] = 1 iff the recomputed root equals root (verify_proof's bool, utils.rs:106) */
int imt_verify_proof_batch(imt_ctx *ctx, const void *leaf, const uint64_t *index, const void *root,
                           const void *sib, unsigned depth, size_t n, uint8_t *ok_out /*[n]*/,
                           unsigned flags);

/* ---- a11 / a12 / a13: batched non-membership ----------------------------------- */
/* verify_non_inclusion (indexed_merkle_tree.rs:127-229) on n items: fail_out[i] is a mask of
 * IMT_F_* (0 = every constraint holds); root_out (optional) is the recomputed root. */
int imt_non_membership_batch(imt_ctx *ctx, const void *root, const void *low_leaf /*[n][3][32]*/,
                             const uint64_t *low_index /*[n]*/, const void *low_sib, unsigned depth,
                             const void *new_val /*[n][32]*/, const uint8_t *is_largest /*[n]*/,
                             size_t n, uint8_t *fail_out /*[n]*/, void *root_out /*[n][32] or NULL*/,
                             unsigned flags);

/* The 128-bit limb witnesses verify_non_inclusion loads for its comparisons
 * (src/indexed_merkle_tree.rs:145-178, :206-224): q[i] = vals[i] >> 128, r[i] = vals[i] mod 2^128 as
 * integers, returned as field elements in the format of `flags` (vals[i] = q[i] * 2^128 + r[i]). */
int imt_split128_batch(imt_ctx *ctx, const void *vals /*[n][32]*/, void *q /*[n][32]*/, void *r /*[n][32]*/,
                       size_t n, unsigned flags);

/* ---- a14: batched insert_leaf witness ------------------------------------------ */
/* insert_leaf (indexed_merkle_tree.rs:231-314): recomputes the 3 leaf hashes and 4 paths per
 * item and checks every constraint.  trace_out (optional) receives, level-major
 * [7][n][32]: low_leaf_hash, root_from_low, new_low_leaf_hash, interim_root,
 * zero_slot_root, new_leaf_hash, new_root_recomputed.
 * new_index is the field element hashed into the rewritten low leaf (:265-269);
 * new_path_index positions the new slot's path (the circuit's new_leaf_proof_helper bits).
 * The reference never ties the two together; pass NULL to use new_index for both. */
int imt_insert_witness_batch(imt_ctx *ctx, const void *old_root /*[n][32]*/,
                             const void *low_leaf /*[n][3][32]*/, const uint64_t *low_index,
                             const void *low_sib, const void *new_root /*[n][32]*/,
                             const void *new_leaf /*[n][3][32]*/, const uint64_t *new_index,
                             const uint64_t *new_path_index /*[n] or NULL = new_index*/,
                             const void *new_sib, const uint8_t *is_largest, unsigned depth, size_t n,
                             uint8_t *fail_out /*[n]*/, void *trace_out /*[7][n][32] or NULL*/,
                             unsigned flags);

/* ---- a15 (+ a2 at depth 32): stateful indexed tree ----------------------------- */
/* A depth-`depth` tree whose leaf i is H(val,next_val,next_idx) of the i-th inserted value
 * (leaf 0 = the {0,0,0} sentinel, empty slot = H(0,0,0): indexed_merkle_tree.rs:373-376).
 * Only the filled prefix (up to `capacity` leaves, a power of two) is stored. */
int imt_itree_new(imt_ctx *ctx, unsigned depth, uint64_t capacity, imt_itree **out);
void imt_itree_free(imt_itree *t);
uint64_t imt_itree_size(const imt_itree *t);      /* leaves in use, sentinel included */
int imt_itree_root(imt_itree *t, void *root /*[32]*/, unsigned flags);
/* Root as it was after the batch inserted `lag` calls ago (0 = the latest batch, 1 = the one before;
 * lag <= 1).  Orders the context's stream behind THAT batch only, so with IMT_PIPELINE a root
 * exchange that lags one step does not stall the batches still in flight. */
int imt_itree_root_lagged(imt_itree *t, unsigned lag, void *root /*[32]*/, unsigned flags);

/* outputs of a batch insertion; every pointer may be NULL; host or device per flags */
typedef struct imt_insert_out {
    uint64_t *low_index;     /* [n]        low leaf of insertion i (update_idx_leaf's 2nd result) */
    void *low_leaf;          /* [n][3][32] the low leaf's preimage BEFORE insertion i */
    uint8_t *is_largest;     /* [n]        low_leaf.next_val == 0 (:737-742) */
    void *old_root;          /* [n][32]    root before insertion i */
    void *interim_root;      /* [n][32]    after the low leaf was rewritten */
    void *new_root;          /* [n][32]    after the new leaf was written */
    void *new_leaf;          /* [n][3][32] preimage written at the new slot */
    void *low_sib;           /* [depth][n][32] low-leaf proof against old_root (layout per flags) */
    void *new_sib;           /* [depth][n][32] new-slot proof against interim_root / new_root */
} imt_insert_out;
/* Every 32-byte row a device pointer refers to (values, preimages, roots, sibling rows) must be 16-byte
 * aligned: the kernels move field elements as two 16-byte words.  hipMalloc, imt_host_alloc and torch
 * allocations are; an offset into them must be a multiple of 16 bytes. */

/* n sequential insertions with the semantics of update_idx_leaf + rebuild (:632-660,
 * :715-735): insertion i finds the low leaf among everything inserted before it, rewrites
 * it, and writes the new leaf at index size+i.  All 2n path recomputes (2 + 2*depth hashes
 * per insertion) run on the device as a level sweep over time-versioned nodes.
 * vals[i] == 0, duplicates (within the batch or already present) -> IMT_ERR_VALUE, nothing
 * changes.  Exceeding capacity -> IMT_ERR_FULL.
 * HOW LONG THE CALL HOLDS THE CALLING THREAD.  IMT_ERR_VALUE must be the call's own return value with the tree
 * untouched, so the call returns when the batch's values have been checked on the GPU (the preparation: sort, low-leaf
 * search), and with IMT_PIPELINE that check is enqueued behind the hashing of the batches already in flight (at most four;
 * the back-pressure that keeps the queues short).  At 2^16 insertions per batch the thread is inside the call for 18.75 ms
 * of every 21.09 ms batch (BENCH_r05: host_call_ms_per_step), almost all of it waiting.  A host that needs its thread
 * calls from a thread of its own -- one caller at a time per tree is all the library asks.  Without IMT_DEVICE_PTRS the
 * call also copies the outputs back and returns when they are there. */
int imt_itree_insert_batch(imt_itree *t, const void *vals /*[n][32]*/, size_t n,
                           const imt_insert_out *out /*may be NULL*/, unsigned flags);
/* current siblings of leaf `index` for n indices (get_proof on the stored tree) */
int imt_itree_get_proof_batch(imt_itree *t, const uint64_t *index, size_t n, void *sib,
                              unsigned flags);
/* preimages {val, next_val, next_idx} of n leaves, all-zero for an empty slot.  index == NULL: the n leaves from the
 * tree's first one on (the whole tree: n = imt_itree_size).  Read from the device-resident index by one kernel (a leaf's
 * successor is the next value in value order); with IMT_DEVICE_PTRS `index` and `preimage` are device pointers and
 * nothing crosses PCIe.  An index outside the tree's capacity -> IMT_ERR_RANGE. */
int imt_itree_get_leaves(imt_itree *t, const uint64_t *index /*[n] or NULL*/, size_t n, void *preimage /*[n][3][32]*/,
                         unsigned flags);
/* Checkpoint / resume and bulk build.  The snapshot of a tree is its leaf preimages in index order
 * (imt_itree_get_leaves(t, NULL, size, ..)): the reference's serde leaf {val, next_val, next_idx}
 * (src/utils.rs:12-17).  imt_itree_load replaces the tree's contents with n such leaves (host memory, or device memory
 * with IMT_DEVICE_PTRS; any IMT_FMT_*).  ON THE GPU it checks that they form one sorted linked list starting at the
 * {0,..} sentinel -- leaves ordered by val (radix sort, 256-bit merge sort if top limbs tie), every leaf's next_val /
 * next_idx = its successor, the largest one's = {0, 0}, no duplicates, every element < p: IMT_ERR_VALUE /
 * IMT_ERR_NONCANONICAL otherwise, naming the first leaf with a broken link, and the tree is left as it was -- then
 * rebuilds every stored level (n leaf hashes + one pass of k_tree_level per level: about 2 hashes per leaf, the "final
 * root only" build of SURVEY.md 8d).  Host memory needed: none beyond the caller's own buffer. */
int imt_itree_load(imt_itree *t, const void *preimages /*[n][3][32]*/, uint64_t n, unsigned flags);
/* low leaf (greatest val < v) for n candidate values: a binary search per value in the device-resident index
 * (k_find_low; with IMT_DEVICE_PTRS `vals` and `low_index` are device pointers); IMT_ERR_VALUE if some v is 0, present,
 * or (with a value partition set) of another subtree's residue */
int imt_itree_find_low_batch(imt_itree *t, const void *vals /*[n][32]*/, size_t n,
                             uint64_t *low_index /*[n]*/, unsigned flags);

/* Witness of verify_non_inclusion for n candidate values against the current tree, produced on the
 * GPU from the device-resident index: the low leaf (greatest stored value below the candidate), its
 * preimage, the is_largest flag and its `depth` siblings.  Outputs feed imt_non_membership_batch
 * unchanged.  Any output pointer may be NULL.  IMT_ERR_VALUE if a candidate is 0, already stored, or -- with a value
 * partition set (imt_itree_set_value_partition) -- of another subtree's residue: this subtree's list says nothing about it.
 * On a placed tree low_sib holds the subtree's `depth` rows; imt_itree_lift_batch with out = {low_sib} and
 * roots_before = roots_after = the subtrees' current roots appends the rows above (a depth-global_depth witness). */
int imt_itree_non_membership_witness(imt_itree *t, const void *vals /*[n][32]*/, size_t n,
                                     uint64_t *low_index /*[n]*/, void *low_leaf /*[n][3][32]*/,
                                     uint8_t *is_largest /*[n]*/, void *low_sib /*[depth][n][32]*/,
                                     unsigned flags);

/* ---- e: the tree as ONE SUBTREE of a deeper tree (sharding by leaf-index range) -----------------
 * north_star's multi-GPU layout: GPU g owns leaf indices [g << depth, (g + 1) << depth) of a tree of depth
 * global_depth as an indexed tree of its own (own {0,0,0} sentinel at its first leaf, own sorted list; the
 * value space is partitioned between the subtrees by the caller).
 * After imt_itree_set_placement (on an empty tree) every LEAF INDEX that crosses the API is global,
 * (subtree_index << depth) + local: low_index outputs, index arguments, and the next_idx field of leaf
 * preimages -- hence what is hashed into the leaves (new_val_idx, src/indexed_merkle_tree.rs:655,:715).
 * Sibling arrays passed to imt_itree_insert_batch are dimensioned for global_depth levels ([global_depth][n]
 * or, item-major, [n][global_depth]); the batch writes levels [0, depth). */
int imt_itree_set_placement(imt_itree *t, unsigned global_depth, uint64_t subtree_index);
/* The value partition between subtrees: from now on imt_itree_insert_batch / imt_itree_batch_begin accept only
 * values with v mod modulus == residue (checked for every value, on the GPU with the default prepare) and
 * fail with IMT_ERR_VALUE otherwise, the tree unchanged.  modulus 0 or 1 = accept everything. */
int imt_itree_set_value_partition(imt_itree *t, uint32_t modulus, uint32_t residue);
/* Lifts the outputs of ONE imt_itree_insert_batch on a placed tree to witnesses of the enclosing tree, so
 * that they are what insert_leaf takes (src/indexed_merkle_tree.rs:231-245: depth-d proofs, depth-d roots):
 * old_root / interim_root / new_root (given as subtree roots, replaced in place) climb the global_depth - depth
 * upper levels, and rows [depth, global_depth) of low_sib / new_sib receive the siblings of that climb.
 * Order of a step across subtrees: subtree 0's insertions, then subtree 1's, ...; so the siblings left of
 * this subtree are taken from roots_after (every subtree's root after the step) and those right of it from
 * roots_before.  Both are [n_subtrees][32] in the format of `flags` (what an all-gather of imt_itree_root /
 * imt_itree_root_lagged returns); levels above depth + log2(n_subtrees) meet empty subtrees.
 * 2 * (global_depth - depth) hashes per insertion: with the batch's 2 + 2 * depth that is the 2 + 2 * 32 of an
 * unsharded depth-32 insertion.  Host or device pointers per flags; enqueued on the context's stream, which
 * the caller must have ordered behind the batch (imt_itree_root_lagged for that batch, or imt_ctx_sync). */
int imt_itree_lift_batch(imt_itree *t, const void *roots_before /*[n_subtrees][32]*/,
                         const void *roots_after /*[n_subtrees][32]*/, size_t n_subtrees, size_t n,
                         const imt_insert_out *out, unsigned flags);

/* ---- e: one tree on several GPUs, sequential semantics (single sorted list) ------------------
 * Every rank holds a replica of the tree and calls the same sequence with the same values; the hashing
 * of each step is split by slot range between the ranks, and the caller all-gathers the value arrays in
 * between (RCCL).  A batch of n insertions has E = 2n events; level l has E slots for l = 0..l0 (l0 =
 * ceil(log2(size + n))), val[l] is [E][32] in the library's device format.  All data pointers are
 * device pointers; work is enqueued on the context's stream.
 *   begin   : values (host or device per flags; identical on all ranks) -> plan; E and l0 returned.
 *             IMT_ERR_VALUE / IMT_ERR_NONCANONICAL / IMT_ERR_FULL exactly as imt_itree_insert_batch.
 *   leaves  : val0[k] for slots [k_begin, k_begin + k_count)
 *   level   : val_out[k] (level l+1) for a slot range, from the COMPLETE val_in (level l)
 *   top     : from the complete val[l0]: root after every event of [e_begin, +e_count) -> roots[e][32]
 *             (device format); the rank whose range contains the last event also fills top_path
 *             ([depth - l0 + 1][32], the stored nodes above l0), which the caller broadcasts.
 *   extract : witnesses of insertions [ins_begin, +ins_count) (any subset, any rank) from the complete
 *             val[0..l0] and roots: every field of imt_insert_out, rows indexed from ins_begin;
 *             formats / sibling layout per flags (level stride = ins_count).
 *   end     : write the batch into the replica (needs the complete val[0..l0] and top_path). */
int imt_itree_batch_begin(imt_itree *t, const void *vals /*[n][32]*/, size_t n, unsigned flags,
                          uint32_t *events_out, uint32_t *l0_out);
int imt_itree_batch_leaves(imt_itree *t, void *val0, uint32_t k_begin, uint32_t k_count);
int imt_itree_batch_level(imt_itree *t, unsigned level, const void *val_in, void *val_out, uint32_t k_begin,
                          uint32_t k_count);
int imt_itree_batch_top(imt_itree *t, const void *val_l0, uint32_t e_begin, uint32_t e_count, void *roots,
                        void *top_path);
int imt_itree_batch_extract(imt_itree *t, const void *const *val_levels /*host array of l0+1 device ptrs*/,
                            const void *roots, uint32_t ins_begin, uint32_t ins_count,
                            const imt_insert_out *out, unsigned flags);
int imt_itree_batch_end(imt_itree *t, const void *const *val_levels, const void *top_path);
/* gives up an open batch (after a failed collective, say): the tree is as it was before imt_itree_batch_begin */
int imt_itree_batch_abort(imt_itree *t);

/* ---- e: ONE tree on several GPUs, sequential semantics, TIME-SLICED ------------------------------
 * The reference's single sorted list (update_idx_leaf, src/indexed_merkle_tree.rs:632-660; insertion i at leaf
 * size + i, :715) at any number of GPUs, bit-exact with one GPU.  A step's world x n insertions are cut into `world`
 * consecutive slices in insertion order: GPU g hashes slice g -- low-leaf rewrites, new leaves, every node version up to
 * the root, 2 + 2 * depth hashes per insertion -- and writes the witnesses of its own insertions directly; every GPU
 * keeps a replica of the stored tree and of the sorted index (all ranks see all values of a step; the index work is
 * hash-free).  What crosses GPUs is what a slice WRITES BACK to the stored tree, level by level, as packed (node,
 * value) pairs: the ranks form a systolic chain `lag` levels apart, the pairs of a tick are all-gathered
 * asynchronously and applied `lag` ticks later, and up to IMT_SLICED_ROUNDS steps overlap.
 *
 * THE SCHEDULE, ITS STREAMS AND EVENTS AND THE COLLECTIVE LIVE IN THE LIBRARY: a host calls imt_sliced_step once per
 * step and imt_sliced_wait / imt_sliced_flush to read results.  Correctness (a slice's level l sees the level-l
 * write-backs of every earlier slice and of no later one) does not depend on anything the caller orders.
 *
 * One imt_sliced drives the ranks of THIS process: one (a distributed world, one process per GPU: the production form,
 * transport = RCCL) or all of them (all replicas in one process: tests and the one-GPU rehearsal, transport = local).
 * Every rank of a world must make the same sequence of imt_sliced_step / imt_sliced_wait / imt_sliced_flush calls, with
 * the same values and the same `round` arguments (imt_sliced_wait included: a wait may issue what is left of its round,
 * and the order in which the collectives of overlapping rounds are issued must be the same on every rank -- with it,
 * no placement of the library's streams on hardware queues can make two ranks wait for each other: tests/hwq_model.py). */
#define IMT_SLICED_ROUNDS 4
typedef struct imt_sliced imt_sliced;
typedef struct imt_transport imt_transport;    /* how the payloads of one tick meet: an all-gather */

/* A caller-supplied collective (MPI, a test double, ...).  all_gather is enqueued on hip_stream and must be complete, in
 * stream order, when the stream gets past it (NCCL semantics): recv[r * bytes, (r + 1) * bytes) = rank r's send[0, bytes)
 * for every rank r of the world.  Calls with different `channel` (0 .. IMT_SLICED_ROUNDS - 1) come from different streams
 * and may overlap; calls on one channel are issued in the same order on every rank.  `buffer` (0 .. lag) tells which of
 * the channel's buffer pairs is in use: the (send, recv) pointers of a (channel, buffer) never change. */
typedef struct imt_transport_ops {
    void *self;
    int (*all_gather)(void *self, int channel, int buffer, const void *send, void *recv, size_t bytes, void *hip_stream);
    void (*destroy)(void *self);       /* may be NULL */
} imt_transport_ops;
int imt_transport_custom_create(const imt_transport_ops *ops, imt_transport **out);
/* all ranks in this process (any devices that can copy to each other): device-to-device copies ordered by events */
int imt_transport_local_create(imt_transport **out);
/* RCCL over xGMI: ncclAllGather on communicators of this library's own (n_comms of them, 1 .. IMT_SLICED_ROUNDS: the
 * rounds in flight use different communicators, so their gathers do not queue behind each other).  Bootstrap like any
 * NCCL program: rank 0 calls imt_rccl_get_unique_id n_comms times, the host broadcasts the ids by whatever means it has,
 * every rank calls imt_transport_rccl_create (collective: it returns when all ranks have joined). */
#define IMT_RCCL_UNIQUE_ID_BYTES 128
int imt_rccl_get_unique_id(void *id /*[IMT_RCCL_UNIQUE_ID_BYTES]*/);
int imt_transport_rccl_create(imt_ctx *ctx, const void *unique_ids /*[n_comms][IMT_RCCL_UNIQUE_ID_BYTES]*/, int n_comms,
                              int world, int rank, imt_transport **out);
/* the same over communicators the host owns (ncclComm_t handles of this device; not destroyed by the library) */
int imt_transport_rccl_adopt(void *const *nccl_comms, int n_comms, imt_transport **out);
/* RCCL is bound at run time: the copy the process already holds (a host that has loaded one), else librccl.so.1 from
 * the library's RUNPATH.  Returns the path of the bound library (or why none was found); *version_out = its
 * NCCL_VERSION_CODE.  IMT_ERR_NO_DEVICE from the calls above when there is none. */
const char *imt_rccl_library(int *version_out);
/* Direct peer copies between processes of one node (HIP IPC memory + event handles; xGMI point-to-point reads when the
 * ranks sit on different GPUs, plain device copies when they share one: the one-GPU rehearsal of a multi-process run).
 * create writes this rank's handle blob (imt_transport_ipc_blob_bytes() bytes); the host all-gathers the blobs by
 * whatever means it has and every rank passes all of them, in rank order, to connect.  depth / max_slice / lag as given
 * to imt_sliced_create. */
size_t imt_transport_ipc_blob_bytes(void);
int imt_transport_ipc_create(imt_ctx *ctx, int world, int rank, unsigned depth, size_t max_slice, int lag,
                             imt_transport **out, void *blob_out);
int imt_transport_ipc_connect(imt_transport *tp, const void *all_blobs /*[world][blob_bytes]*/);
/* after every imt_sliced that uses it (IMT_ERR_ARG and nothing destroyed while one still does), and before the imt_ctx it
 * was created on */
int imt_transport_destroy(imt_transport *tp);
const char *imt_transport_last_error(const imt_transport *tp);
/* IMT_TRANSPORT_OPT_TIMEOUT_MS: how long a wait for a peer inside the IPC transport may last (GPU-side and host-side;
 * default 60 000, at least 10) before it gives up -- the transport's error word then turns non-zero and STAYS so: the
 * copies, acknowledgements and applies behind the wait are skipped on the device, the next imt_sliced_step / _wait /
 * _flush returns IMT_ERR_INTERNAL and the world refuses to go on.  IMT_TRANSPORT_OPT_HOST_POLL (before
 * imt_transport_ipc_connect): -1 decide from the PCI bus ids (ranks that share a GPU wait on the host), 0 / 1 forced. */
#define IMT_TRANSPORT_OPT_TIMEOUT_MS 1
#define IMT_TRANSPORT_OPT_HOST_POLL 2
int imt_transport_set_option(imt_transport *tp, int option, long value);
/* One small all-gather on the transport's own channel 0, enqueued on hip_stream (NULL = the stream of the context the
 * transport was created on): recv[r * bytes, (r + 1) * bytes) = rank r's send[0, bytes), device pointers, bytes a
 * multiple of 16 and at most 4096.  This is the ONE collective of the subtree layout (the all-gather of the per-GPU
 * subtree roots, 32 bytes per rank and step: examples/subtree_procs_demo.c, sharded.py), offered so that a host without a
 * collective library of its own can use the communicators this library already has (RCCL: ncclAllGather on communicator
 * 0; IPC: peer reads behind stream-ordered flags; custom: the vtable's all_gather with channel 0, buffer 0).  Every rank
 * calls it the same number of times in the same order; not while an imt_sliced uses the transport.  With the IPC
 * transport between ranks that share a GPU the call waits ON THE HOST for the peers (as that transport's fence does). */
int imt_transport_all_gather(imt_transport *tp, const void *send, void *recv, size_t bytes, void *hip_stream);
/* Has a GPU-side wait of the transport given up on a peer (IMT_TRANSPORT_OPT_TIMEOUT_MS)?  IMT_OK, or IMT_ERR_INTERNAL
 * (text in imt_transport_last_error): the error is STICKY -- every copy behind the wait was skipped on the device, `recv`
 * of that gather and of every later one holds whatever it held before, and later imt_transport_all_gather calls return
 * IMT_ERR_INTERNAL themselves.  The wait runs on the device: a caller of imt_transport_all_gather synchronises the stream
 * it named and then asks here BEFORE it reads `recv` (sharded.py does; an imt_sliced asks by itself in every
 * imt_sliced_step / _wait / _flush).  Transports without a GPU-side wait (RCCL, local, custom) always answer IMT_OK. */
int imt_transport_poll_error(imt_transport *tp);

/* trees[k] = the replica of rank first_rank + k (each on its own context; empty or with the same contents on every rank;
 * not placed, not partitioned); n_local = 1, or = world with the local transport.  max_slice = the largest n of a step.
 * lag = 0: the default max(2, ceil((depth + 1) / (3 * world))) (6, 3, 2 at 2, 4, 8 GPUs for depth 32).
 * IMT_ERR_RANGE if (world, depth, lag) would keep more than IMT_SLICED_ROUNDS steps in flight. */
int imt_sliced_create(imt_itree *const *trees, int n_local, int world, int first_rank, imt_transport *tp,
                      size_t max_slice, int lag, imt_sliced **out);
/* Options.  w == NULL: the defaults that imt_sliced_create calls made LATER in this process start from (every option);
 * w != NULL: an existing world (PREP_STREAM, WATCHDOG_MS, TIMING only -- the others decide which streams exist).
 * IMT_ERR_RANGE for a value outside the option's range, IMT_ERR_ARG for an unknown option or one that cannot be changed
 * any more.  (tools/ may override the defaults through environment variables of the same names, IMT_SLICED_COMM_STREAMS
 * ...; they are read once per imt_sliced_create.) */
#define IMT_SLICED_OPT_COMM_STREAMS 1      /* 0 .. 4 streams for the collectives (0: on the round's own stream); default 4 */
#define IMT_SLICED_OPT_COMM_PRIORITY 2     /* their HIP stream priority; default 0 = the pool of hardware queues the rounds use */
#define IMT_SLICED_OPT_ROUND_PRIORITIES 3  /* the four round streams: 0 (default) all at normal priority; 1 one normal + three high
                                              (not verified); 2 all LOW; 3 all HIGH -- a pool of hardware queues per priority */
#define IMT_SLICED_OPT_APPLY_STREAMS 4     /* 1: other ranks' write-backs are applied on a stream of their own per round slot; default 0 */
#define IMT_SLICED_OPT_PREP_STREAM 5       /* a step's preparation runs on 0 (default) the new round slot's collective stream, 1 its round stream, 2 the tree's side stream */
#define IMT_SLICED_OPT_VERIFY_QUEUES 6     /* 1 (default): imt_sliced_create measures which of its streams share a hardware queue
                                              and re-creates streams until the placement below holds; 0: take what the runtime gives.
                                              The measurement (a few milliseconds: a wave spinning 200 us on one stream, GPU-clock
                                              stamps on the others) wants the process's OTHER streams idle -- work queued elsewhere
                                              on a shared hardware queue delays a stamp and reads as "shares a queue"; the result is
                                              then a more cautious placement (imt_sliced_info.placement), never a wrong tree */
#define IMT_SLICED_OPT_WATCHDOG_MS 7       /* a host wait inside imt_sliced_step / _wait / _flush gives up after this long
                                              (default 120 000; 0 = never): IMT_ERR_TIMEOUT, the world's state on stderr and in
                                              imt_sliced_last_error, the world refuses to go on */
#define IMT_SLICED_OPT_TIMING 8            /* 1: host time per phase on stderr at imt_sliced_destroy */
#define IMT_SLICED_OPT_COMM_PLACEMENT 9    /* which hardware queues the collectives' streams are put on (by creating streams until one
                                              lands there): 0 (default) queues of their OWN -- none a round stream is on, none shared --
                                              if the runtime has any to give, else the queue of their round's stream; 1 their round's
                                              queue; 2 their own or IMT_SLICED_PLACEMENT_DEGRADED.  On its round's queue a collective
                                              holds up the round's next unit until the slowest rank has packed (every tick a barrier
                                              across ranks); on its own it overlaps the next `lag` units.  The runtime has four queues
                                              per priority level unless the host's environment says GPU_MAX_HW_QUEUES=8 (or more)
                                              BEFORE the first HIP call; queues in ANOTHER priority pool are what
                                              IMT_SLICED_OPT_POOLS (below, the default for one process per GPU) arranges */
#define IMT_SLICED_OPT_POOLS 10            /* 1: three priority pools -- round streams HIGH, collectives' streams LOW, the preparation
                                              on the round's stream (sets ROUND_PRIORITIES 3, COMM_PRIORITY lowest, PREP_STREAM 1); 2:
                                              round AND collectives' streams HIGH, the collectives' on their rounds' queues (4 % more per
                                              rank on one GPU, but every tick a barrier across ranks: COMM_PLACEMENT); 0: everything in
                                              the normal pool, as the options above say; -1 (default): 1 for a world of more than one
                                              rank with one rank in this process, else 0.
                                              Options the caller has set explicitly (imt_sliced_set_option(NULL, ...)) are NOT
                                              overwritten by a preset: the caller's value stands and imt_sliced_last_error of the new
                                              world says which preset was left out.  The runtime keeps a set of
                                              hardware queues per priority, and everybody else's streams are in the normal one -- the
                                              host's, and RCCL's: a communicator creates three of its own and brackets every
                                              collective with one (the user's stream waits for it, it waits for the kernel).  Sharing
                                              a round's queue, such a stream puts the round behind every collective and the collective
                                              behind the round's backlog, and this library can neither see nor move it; in the HIGH
                                              pool the rounds are alone.  Needs no GPU_MAX_HW_QUEUES.  One rank of 2 / 4 / 8 alone on
                                              a GPU: within 1 % of one pool */
#define IMT_SLICED_OPT_RESET 11            /* w == NULL, value 0: every default back to the library's own, and no option counts as
                                              "set explicitly by the caller" any more (see IMT_SLICED_OPT_POOLS) */
int imt_sliced_set_option(imt_sliced *w, int option, long value);
/* One step: vals = ALL world x n values of the step in insertion order (device pointer, identical contents on every
 * rank; format per flags), outs[k] = where local rank k's witnesses of ITS slice (insertions [rank * n, (rank + 1) * n)
 * of the step, n rows; sibling arrays [depth][n] or item-major per flags) are written; both stay untouched by the caller
 * until imt_sliced_wait for that round.  Enqueues the step's preparation and advances the schedule by one round
 * period; returns without waiting for the GPU (except for the values check: IMT_ERR_VALUE / IMT_ERR_NONCANONICAL /
 * IMT_ERR_FULL exactly as imt_itree_insert_batch, the same verdict on every rank, nothing changed).  *round_out = the
 * step's number.  flags: IMT_FMT_*, IMT_SIB_ITEM_MAJOR, IMT_INPUTS_READY. */
int imt_sliced_step(imt_sliced *w, const void *vals /*[world * n][32]*/, size_t n, const imt_insert_out *outs /*[n_local]*/,
                    unsigned flags, uint64_t *round_out);
/* HOW LONG imt_sliced_step BLOCKS, and why.  The call returns when the step's values have been checked on the GPU (the
 * preparation: sort, low-leaf search, merge into the index), because IMT_ERR_VALUE must be the call's own return value
 * and must leave every rank's tree untouched.  The preparation is short (1.5 - 2 ms at 2^16 values per rank) but is
 * enqueued behind the hashing already in flight, and one period of ticks (a whole step's worth of hashing per rank) is
 * issued per call and no more -- the back-pressure that keeps the hardware queues from filling up with a backlog other
 * streams would stand behind.  At 2^16 insertions per rank and step the calling thread is therefore inside the call for
 * about 16 - 17 ms of every 21.6 ms step: 1.4 - 3.8 ms issuing (imt_sliced_info.host_issue_ms), the rest waiting
 * (host_wait_ms).  A host that needs its thread runs the world from a thread of its own: the calls of one world need one
 * caller at a time, nothing else.  (Splitting the call into issue + verdict was measured and is slower at 8 ranks:
 * docs/LAB_NOTES.md.)
 *
 * host waits until local rank k's witnesses of `round` are complete (issues what is left of that ROUND first -- up to the
 * tick at which every rank's last unit of the round has been issued: the same tick on every rank) */
int imt_sliced_wait(imt_sliced *w, int local_rank, uint64_t round);
/* issues everything that is left of the steps in flight and waits for it: all replicas then hold the same tree, and the
 * trees may be used through the ordinary imt_itree_* calls again (roots, proofs, non-membership witnesses: every replica
 * holds the whole list, so such queries need no exchange).  Between imt_sliced_step and imt_sliced_flush only
 * imt_sliced_* calls may touch the trees. */
int imt_sliced_flush(imt_sliced *w);
typedef struct imt_sliced_info {
    int world, n_local, lag, period, gathers_per_round, round_ticks, rounds_in_flight;
    size_t payload_bytes;            /* size of one send buffer */
    uint64_t rounds, collectives, bytes_gathered;    /* since creation; collectives / bytes per local rank summed */
    double host_issue_ms, host_wait_ms;              /* wall time inside imt_sliced_step since creation: issuing work (every
                                                        launch, event and collective of the step) / waiting (for the GPU: the
                                                        values check of the step, back-pressure when the host runs ahead; for
                                                        peers: the host-polled IPC transport) */
    /* Where the world's streams sit (local rank 0's; measured at creation, IMT_SLICED_OPT_VERIFY_QUEUES).  The HIP runtime
     * multiplexes streams onto a few in-order hardware queues per priority level; wanted: the four round streams on four
     * different queues, slot i's collective stream on a queue of its own (IMT_SLICED_OPT_COMM_PLACEMENT; else on round stream
     * i's), its apply stream on round stream i's queue.  queue_map[k][slot] = hardware
     * queue class (0 .. hw_queues - 1, numbered by first use) of slot's round (k = 0), collective (1) and apply (2)
     * stream; 4 + slot = a queue of its own in the rounds' pool (shared with no round stream and no other helper), -2 = a
     * queue of its own in another priority pool, -1 = no such stream or not measured. */
    int placement;                   /* IMT_SLICED_PLACEMENT_* */
    int hw_queues;                   /* distinct hardware queues under the four round streams (4 wanted) */
    int comm_streams;                /* streams carrying collectives after placement (0: the round streams do) */
    int streams_recreated;           /* streams that had to be created again to get the placement */
    int queue_map[3][IMT_SLICED_ROUNDS];
    int pools;                       /* IMT_SLICED_OPT_POOLS as resolved: 1 = rounds HIGH / collectives LOW / RCCL and the host normal */
} imt_sliced_info;
#define IMT_SLICED_PLACEMENT_UNVERIFIED 0  /* not measured (option off, or unequal round priorities) */
#define IMT_SLICED_PLACEMENT_AS_CREATED 1  /* measured: as wanted, first try */
#define IMT_SLICED_PLACEMENT_REPAIRED 2    /* measured: as wanted after re-creating streams_recreated streams */
#define IMT_SLICED_PLACEMENT_DEGRADED 3    /* could not be had: collectives (applies) run on the round streams and / or the
                                              round streams share queues; imt_sliced_last_error says which */
int imt_sliced_get_info(const imt_sliced *w, imt_sliced_info *out);
/* Where the world stands, as text (what IMT_ERR_TIMEOUT writes to stderr): the global tick issued, per rank and round
 * slot the first tick whose unit / apply has not completed on the device, the collectives still pending with their
 * channel.  Writes at most cap - 1 characters + NUL; returns the length of the full text.  For a host's own watchdog. */
int imt_sliced_dump(imt_sliced *w, char *out, size_t cap);
const char *imt_sliced_last_error(const imt_sliced *w);
/* flushes first.  A world that failed (IMT_ERR_TIMEOUT, IMT_ERR_INTERNAL) is not flushed; if its streams do not drain
 * within the watchdog's limit the call returns anyway and leaves the world's device memory and streams allocated -- see
 * IMT_ERR_TIMEOUT above for what the process should do next. */
void imt_sliced_destroy(imt_sliced *w);

/* The building blocks imt_sliced_* is made of (kept exported for hosts that bring their own scheduler; a host that calls
 * them owns the ordering rule above).  A slice is hashed unit by unit (unit 0 = its leaf hashes, unit 1 + l = level
 * l -> l + 1) on a stream the caller names; each unit leaves a PAYLOAD -- the nodes it wrote back to level l of the
 * stored tree as packed (node, value) pairs.  Device pointers only.
 *   prepare : vals = the step's values for the slices before this GPU's, its own, and those after, in step order
 *             (identical on every GPU).  Index work for all of them on the tree's side stream (sort, low-leaf
 *             search, merge into the sorted index), events + level tables for the own slice only; the hash-free
 *             outputs (low_index, is_largest, low_leaf, new_leaf) are written now.  Blocks until the values are
 *             checked: IMT_ERR_VALUE / IMT_ERR_NONCANONICAL / IMT_ERR_FULL as imt_itree_insert_batch, the same
 *             verdict on every GPU, tree and index unchanged.  On success the index holds the whole step and
 *             *slice_out names the slice (up to 4 may be open).  `out` pointers are kept until the last unit.  The
 *             slice's units are ordered behind every ordinary batch (imt_itree_insert_batch, pipelined or not) issued
 *             on the tree before.
 *   unit    : enqueue unit `unit` (0 .. depth, in order) of an open slice on hip_stream; payload (NULL = none
 *             wanted) receives imt_itree_slice_payload_bytes(n_own) bytes, stream-ordered.
 *   apply   : enqueue another GPU's payload for (slice of n insertions into a tree of size_before leaves, unit) on
 *             hip_stream: writes that unit's nodes into this replica.
 * After the last apply the caller synchronises the streams it passed before any other call on the tree.
 * imt_itree_root_lagged does not see slices (a slice's own last root is a mid-step root on every rank but the last). */
size_t imt_itree_slice_payload_bytes(size_t n);   /* the largest payload of a slice of n insertions (buffer size) */
/* bytes the payload of `unit` of a slice of n insertions into a tree of size_before leaves actually uses (<= the above):
 * 128 B of header + one (node, value) pair per node the level can hold under the leaves in use, at most one per
 * event -- what an all-gather of that unit has to move.  The same on every GPU (it depends on sizes only). */
size_t imt_itree_slice_unit_bytes(const imt_itree *t, uint64_t size_before, size_t n, unsigned unit);
int imt_itree_slice_prepare(imt_itree *t, const void *vals /*[n_before + n_own + n_after][32]*/, size_t n_before,
                            size_t n_own, size_t n_after, const imt_insert_out *out, unsigned flags, int *slice_out,
                            uint32_t *l0_out);
int imt_itree_slice_unit(imt_itree *t, int slice, unsigned unit, void *payload, void *hip_stream);
int imt_itree_slice_apply(imt_itree *t, uint64_t size_before, size_t n, unsigned unit, const void *payload,
                          void *hip_stream);
/* the same for the `count` payloads of one all-gather (payload r at gathered + r * stride, stride >= the bytes
 * imt_itree_slice_unit_bytes gives for every r that is not skipped); unit[r] < 0 = skip */
int imt_itree_slice_apply_gathered(imt_itree *t, const void *gathered, size_t stride, size_t count,
                                   const uint64_t *size_before /*[count]*/, const uint64_t *n /*[count]*/,
                                   const int32_t *unit /*[count]*/, void *hip_stream);

/* ---- e: multi-GPU helpers ------------------------------------------------------ */
/* Root of a depth-`depth` tree whose 2^k subtrees of height `sub_height` have the given
 * roots (k = log2(n_roots)); the levels above sub_height + k are extended with the
 * all-empty subtree hashes.  Used after the all-gather of per-GPU subtree roots. */
int imt_combine_subtree_roots(imt_ctx *ctx, const void *sub_roots /*[n_roots][32]*/, size_t n_roots,
                              unsigned sub_height, unsigned depth, void *root /*[32]*/,
                              unsigned flags);
/* Z[0..depth]: hash of the empty subtree of each height, Z[0] = H(0,0,0) */
int imt_zero_hashes(imt_ctx *ctx, unsigned depth, void *out /*[depth+1][32]*/, unsigned flags);

#ifdef __cplusplus
}
#endif
#endif /* IMT_H */
