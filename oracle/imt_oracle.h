/*
 * imt_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the indexed-Merkle-tree hot path of
 * aerius-labs/indexed-merkle-tree-halo2 (reference mounted at /root/reference).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the shipped path (libimt_hip.so) never links or calls it.
 *
 * Parity pin: the reference holds exactly one absolute known-answer for this
 * path, Poseidon(T=3,RATE=2,R_F=8,R_P=57)([0,0,0]) at
 * src/indexed_merkle_tree.rs:247-250; oracle/selftest.c and
 * tests/test_oracle_golden.py check it.  The arithmetic itself lives in two
 * un-vendored crates (pse-poseidon @ aerius-labs branch feat/stateless-hash,
 * halo2-base @ aerius-labs/halo2-lib branch feat/secp256k1-hash2curve,
 * Cargo.toml:14-16, no lockfile); their published algorithm is restated here.
 * Everything except that one KAT (hash2 values, roots, depth > 3) is
 * "derived, KAT-anchored": unpinned by the reference itself.  The permutation alone
 * is additionally pinned by two public known answers of circomlib's Poseidon (same
 * t=3 constants): lane 0 of the permutation of [0,1,2] and of [0,0,0]
 * (selftest.c, tests/golden/vectors.json "public-circomlib").
 *
 * Field elements cross this API as 32-byte little-endian canonical integers
 * (what halo2curves' Fr::to_repr() yields).
 */
#ifndef IMT_ORACLE_H
#define IMT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } ofr_t; /* Montgomery form, R = 2^256 */

/* error codes: the reference's Err strings and panics (src/utils.rs:24-36,45) */
#define ORC_OK 0
#define ORC_ERR_NO_LEAVES (-1)   /* "Cannot create Merkle Tree with no leaves" utils.rs:25 */
#define ORC_ERR_ODD_LEAVES (-2)  /* "Leaves must be even" utils.rs:35 */
#define ORC_ERR_NOT_POW2 (-3)    /* index-out-of-bounds panic at utils.rs:45 */
#define ORC_ERR_RANGE (-4)       /* index-out-of-bounds panic in get_proof utils.rs:76 */
#define ORC_ERR_NONCANONICAL (-5)
#define ORC_ERR_ALLOC (-6)

/* ---- field (halo2curves bn256::Fr, named grumpkin::Fq at indexed_merkle_tree.rs:327) */
void ofr_init(void);
int ofr_from_bytes(ofr_t *out, const uint8_t in[32]); /* rejects >= p */
void ofr_to_bytes(uint8_t out[32], const ofr_t *a);
void ofr_from_u64(ofr_t *out, uint64_t v);
void ofr_add(ofr_t *o, const ofr_t *a, const ofr_t *b);
void ofr_sub(ofr_t *o, const ofr_t *a, const ofr_t *b);
void ofr_mul(ofr_t *o, const ofr_t *a, const ofr_t *b);
void ofr_inv(ofr_t *o, const ofr_t *a);
int ofr_eq(const ofr_t *a, const ofr_t *b);
int ofr_is_zero(const ofr_t *a);
int ofr_cmp_canonical(const ofr_t *a, const ofr_t *b); /* -1/0/1 by integer value */
void ofr_raw_constants(uint64_t p[4], uint64_t r[4], uint64_t r2[4], uint64_t *inv);

/* ---- Poseidon (pse-poseidon Spec::new(8,57) + Poseidon::<F,3,2>) */
#define ORC_T 3
#define ORC_RF 8
#define ORC_RP 57
#define ORC_ROUNDS 65
void orc_poseidon_init(void);                 /* Grain LFSR constants + Cauchy MDS */
void orc_poseidon_params(uint8_t rc[ORC_ROUNDS * 3][32], uint8_t mds[9][32]);
void orc_permute(ofr_t s[3]);                 /* plain 65-round form */
void orc_permute_bytes(uint8_t s[3][32]);
void orc_hash2_fr(ofr_t *out, const ofr_t *a, const ofr_t *b);
void orc_hash3_fr(ofr_t *out, const ofr_t *a, const ofr_t *b, const ofr_t *c);
/* update(&[..]) + squeeze_and_reset(): utils.rs:46-47, indexed_merkle_tree.rs:663-668 */
int orc_hash2(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]);
int orc_hash3(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], const uint8_t c[32]);
int orc_hash2_batch(uint8_t *out, const uint8_t *in, size_t n); /* in[n][2][32] */
int orc_hash3_batch(uint8_t *out, const uint8_t *in, size_t n); /* in[n][3][32] */
/* generic sponge over any input length (update + squeeze_and_reset) */
int orc_hash_var(uint8_t out[32], const uint8_t *in, size_t n_elems);

/* ---- dense native tree: IndexedMerkleTree::{new,get_root,get_proof,verify_proof}
 *      src/utils.rs:19-108 */
typedef struct orc_tree orc_tree;
int orc_tree_new(orc_tree **out, const uint8_t *leaves, size_t n_leaves); /* utils.rs:20 */
void orc_tree_free(orc_tree *t);
size_t orc_tree_num_levels(const orc_tree *t);
void orc_tree_get_root(const orc_tree *t, uint8_t root[32]);              /* utils.rs:59 */
int orc_tree_get_proof(const orc_tree *t, size_t index, uint8_t *proof, uint8_t *helper); /* :63 */
int orc_tree_level(const orc_tree *t, size_t level, uint8_t *out, size_t *n);
/* verify_proof ignores helpers, order from index parity: utils.rs:87-107 */
int orc_verify_proof(const uint8_t leaf[32], uint64_t index, const uint8_t root[32],
                     const uint8_t *proof, size_t depth);
int orc_path_root(uint8_t root_out[32], const uint8_t leaf[32], uint64_t index,
                  const uint8_t *proof, size_t depth);

/* ---- circuit witness values (src/indexed_merkle_tree.rs:33-314) */
/* compute_merkle_root :78-96 (dual_mux by helper bit :47-63). helper[i][32] is 0 or 1. */
int orc_compute_merkle_root(uint8_t root_out[32], const uint8_t leaf[32], const uint8_t *proof,
                            const uint8_t *helper, size_t depth);
/* 256-bit a<b from 128-bit limbs, the boolean formula at :98-125 */
int orc_is_less_than_limbs(const uint8_t a[32], const uint8_t b[32]);

/* failure bits of the relation checkers (a constraint / assert of the reference each) */
#define ORC_F_RANGE_PRED 0x001   /* select(...)==1 assert :190-191 */
#define ORC_F_LOW_IN_ROOT 0x002  /* verify_merkle_proof(root, low_leaf_hash) :196-204 */
#define ORC_F_LOW_LT_NEW 0x004   /* check_less_than == 1 :226-228 */
#define ORC_F_ZERO_SLOT 0x008    /* zero leaf in interim root :286-294 */
#define ORC_F_NEXT_VAL 0x010     /* new_leaf.next_val == low_leaf.next_val :296 */
#define ORC_F_NEXT_IDX 0x020     /* new_leaf.next_idx == low_leaf.next_idx :297 */
#define ORC_F_NEW_ROOT 0x040     /* new_root == recomputed :313 */
#define ORC_F_BAD_BIT 0x080      /* assert_bit on a helper / flag :41,54 */

typedef struct {
    uint8_t low_leaf_hash[32];
    uint8_t root_from_low[32];   /* recomputed old root */
    uint8_t new_low_leaf_hash[32];
    uint8_t interim_root[32];
    uint8_t zero_slot_root[32];  /* root recomputed from the zero leaf over new_leaf_proof */
    uint8_t new_leaf_hash[32];
    uint8_t new_root[32];        /* recomputed */
} orc_insert_trace;

/* verify_non_inclusion :127-229; returns failure bitmask (0 = satisfied) */
int orc_verify_non_inclusion(const uint8_t root[32], const uint8_t low_leaf[3][32],
                             const uint8_t *low_proof, const uint8_t *low_helper, size_t depth,
                             const uint8_t new_val[32], int is_new_leaf_largest,
                             uint8_t low_leaf_hash_out[32], uint8_t root_out[32]);
/* insert_leaf :231-314; returns failure bitmask (0 = all constraints satisfied) */
int orc_insert_leaf(const uint8_t old_root[32], const uint8_t low_leaf[3][32],
                    const uint8_t *low_proof, const uint8_t *low_helper,
                    const uint8_t new_root[32], const uint8_t new_leaf[3][32],
                    uint64_t new_leaf_index, const uint8_t *new_proof, const uint8_t *new_helper,
                    int is_new_leaf_largest, size_t depth, orc_insert_trace *trace);

/* ---- f1: cell-by-cell witness trace of halo2-base's PoseidonHasher::hash_fix_len_array (trace.c) ----
 * The advice column the gadget appends for one hash of `arity` inputs, in assignment order.  A cell is a
 * constant, a copy of a hash input / of the initial state [2^64, 0, 0] / of an earlier witness, or a NEW witness
 * (numbered in order: the "trace rows").  gate = 1 where a vertical gate a + b*c = d starts. */
#define ORC_CELL_CONST 0
#define ORC_CELL_INPUT 1
#define ORC_CELL_INIT 2
#define ORC_CELL_WITNESS 3
#define ORC_CELL_COPY 4
typedef struct { uint8_t kind, gate; uint16_t region; uint32_t index; } orc_trace_cell;   /* region = 1: first cell of an assign_region call */
/* cells [cap][32] / desc [cap] / witness [wcap][32]: any may be NULL; counts are always returned.
 * out_row = the trace row that is the hash (state[1] after the last permutation). */
int orc_hash_trace(const uint8_t *in /*[arity][32]*/, int arity, uint8_t *cells, orc_trace_cell *desc, size_t cap,
                   size_t *n_cells, uint8_t *witness, size_t wcap, size_t *n_witness, uint32_t *out_row);
/* the optimised spec this file derives (for comparison with the product's own derivation) */
void orc_trace_spec(uint8_t *start /*[5][3][32]*/, uint8_t *partial /*[57][32]*/, uint8_t *end /*[3][3][32]*/,
                    uint8_t *pre_sparse /*[9][32]*/, uint8_t *sp_row /*[57][3][32]*/, uint8_t *sp_col_hat /*[57][2][32]*/);

/* ---- f3: the advice cells of insert_leaf OUTSIDE hash_fix_len_array (gadget.c) ----
 * is_less_than (src/indexed_merkle_tree.rs:98-125) of two 256-bit values: its whole column (inputs 0..3 = a_q, a_r, b_q,
 * b_r), 4 * (ceil(128 / lookup_bits) + 1) + 27 new witnesses; out_row = the row of the result. */
size_t orc_less_than_trace_rows(unsigned lookup_bits);
int orc_less_than_trace(const uint8_t a[32], const uint8_t b[32], unsigned lookup_bits, uint8_t *cells, orc_trace_cell *desc,
                        size_t cap, size_t *n_cells, uint8_t *witness, size_t wcap, size_t *n_witness, uint32_t *out_row);
/* the witness rows of that column which are range-checked through the lookup table (the limbs), in column order */
int orc_less_than_lookup_rows(unsigned lookup_bits, uint32_t *rows, size_t cap, size_t *n_rows);
/* every new witness of insert_leaf (:231-314) that is not inside a hash ("glue rows": is_equal, the limb loads, both
 * is_less_than, select, load_witness + dual_mux of the four paths), in assignment order, and the segments -- glue rows
 * (kind 0) / one hash_fix_len_array call (kind 1, rows of orc_hash_trace in imt_insert_trace_batch's order) -- that make
 * up the column */
typedef struct { uint32_t kind, arity; uint64_t first_row, n_rows; } orc_column_segment;
size_t orc_insert_gadget_rows(size_t depth, unsigned lookup_bits);
int orc_insert_gadget_trace(const uint8_t low_leaf[3][32], uint64_t low_index, const uint8_t *low_proof,
                            const uint8_t new_leaf[3][32], uint64_t new_index, uint64_t new_path_index,
                            const uint8_t *new_proof, int is_new_leaf_largest, size_t depth, unsigned lookup_bits,
                            uint8_t *witness, size_t wcap, size_t *n_witness, orc_column_segment *segs, size_t seg_cap,
                            size_t *n_segs);
/* the same for one verify_non_inclusion call alone (:127-229): 17 + 2 K + 4 depth glue rows, 3 + 2 depth segments (the
 * hash blocks are H(low_leaf) and the path's hashes, imt_path_trace_batch's order) */
size_t orc_non_inclusion_gadget_rows(size_t depth, unsigned lookup_bits);
int orc_non_inclusion_gadget_trace(const uint8_t low_leaf[3][32], uint64_t low_index, const uint8_t *low_proof,
                                   const uint8_t new_val[32], int is_new_leaf_largest, size_t depth, unsigned lookup_bits,
                                   uint8_t *witness, size_t wcap, size_t *n_witness, orc_column_segment *segs,
                                   size_t seg_cap, size_t *n_segs);


/* ---- indexed-list insertion of the test module (:632-671) */
/* update_idx_leaf: linear scan, in place on preimages[n][3][32]; returns low idx in *low */
int orc_update_idx_leaf(uint8_t *preimages, size_t n, const uint8_t new_val[32],
                        uint64_t new_val_idx, uint64_t *low);
int orc_hash_preimages(uint8_t *leaves_out, const uint8_t *preimages, size_t n); /* :662-671 */

/* ---- sparse depth-d append-only indexed tree: the depth-32 CPU baseline.  The
 * reference's dense Vec<Vec<F>> needs 256 GiB at depth 32; this keeps
 * the same values per node and per insertion with only the filled prefix stored. */
typedef struct orc_sparse orc_sparse;
int orc_sparse_new(orc_sparse **out, unsigned depth, uint64_t capacity);
void orc_sparse_free(orc_sparse *t);
/* subtree of a deeper tree: next_idx fields hold base + local index (positions in this API stay local) */
void orc_sparse_set_index_base(orc_sparse *t, uint64_t base);
void orc_sparse_root(const orc_sparse *t, uint8_t root[32]);
uint64_t orc_sparse_size(const orc_sparse *t);
int orc_sparse_proof(const orc_sparse *t, uint64_t index, uint8_t *proof /*[d][32]*/);
int orc_sparse_preimage(const orc_sparse *t, uint64_t index, uint8_t out[3][32]);
void orc_zero_hashes(uint8_t *out /*[d+1][32]*/, unsigned depth);
/* One sequential insertion with the semantics of update_idx_leaf + rebuild
 * (:632-660, :715-735).  Outputs (any may be NULL): low idx, old low preimage,
 * is_largest, interim root (after the low-leaf rewrite), new root, low-leaf proof
 * against the old root and new-leaf proof against the new root ([d][32] each).
 * Returns 0, or -10 for a value that is 0 / already present / non-canonical (the
 * reference's circuit would panic at :190), -11 when full. */
int orc_sparse_insert(orc_sparse *t, const uint8_t val[32], uint64_t *low_idx,
                      uint8_t low_leaf[3][32], int *is_largest, uint8_t interim_root[32],
                      uint8_t new_root[32], uint8_t *low_proof, uint8_t *new_proof);
/* low-leaf search only (predecessor by canonical integer order) */
int orc_sparse_find_low(const orc_sparse *t, const uint8_t val[32], uint64_t *low_idx);
/* Rebuild a fresh tree from the preimages of its first n leaves: hash_nullifier_pre_images (:662-671) +
 * IndexedMerkleTree::new (src/utils.rs:38-51) with every other slot empty.  -10 unless the preimages are one sorted
 * list starting at the sentinel.  For starting a sequential run at a checkpoint. */
int orc_sparse_load(orc_sparse *t, const uint8_t *preimages /*[n][3][32]*/, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
