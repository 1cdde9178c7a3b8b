/* insert_demo.c -- plain C against include/imt.h: 6 insertions into a depth-3 tree (the values of the
 * reference's test_insert_leaf_multiple_round, src/indexed_merkle_tree.rs:683-690), then every
 * insert_leaf constraint re-checked on the GPU, then the witness trace of all 3 + 4*3 Poseidon gadget calls
 * of every insert_leaf (what a chip assigns instead of recomputing) and its cell map, then a checkpoint of the tree
 * reloaded into a second one.  Build:
 *   gcc -std=c11 -I include examples/insert_demo.c -L indexed-merkle-tree-halo2_amd/csrc -limt_hip -o insert_demo
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "imt.h"

#define N 6
#define DEPTH 3

int main(void) {
    imt_ctx *ctx = NULL;
    imt_itree *tree = NULL;
    int rc = imt_ctx_create(0, &ctx);
    if (rc) { fprintf(stderr, "imt_ctx_create: %d (no GPU?)\n", rc); return 1; }
    if ((rc = imt_itree_new(ctx, DEPTH, 8, &tree))) { fprintf(stderr, "%s\n", imt_last_error(ctx)); return 1; }

    static const unsigned vals_small[N] = {30, 10, 20, 5, 50, 35};
    unsigned char vals[N][32];
    memset(vals, 0, sizeof vals);
    for (int i = 0; i < N; i++) vals[i][0] = (unsigned char)vals_small[i];

    uint64_t low_index[N], new_index[N];
    unsigned char is_largest[N], low_leaf[N][3][32], new_leaf[N][3][32];
    unsigned char old_root[N][32], interim_root[N][32], new_root[N][32];
    unsigned char low_sib[DEPTH][N][32], new_sib[DEPTH][N][32];
    imt_insert_out out = {low_index, low_leaf, is_largest, old_root, interim_root, new_root, new_leaf, low_sib, new_sib};
    if ((rc = imt_itree_insert_batch(tree, vals, N, &out, IMT_FMT_CANONICAL))) {
        fprintf(stderr, "insert: %s\n", imt_last_error(ctx));
        return 1;
    }
    for (int i = 0; i < N; i++) {
        new_index[i] = (uint64_t)i + 1;
        printf("insert %2u: low leaf %llu, largest %d, new root ", vals_small[i], (unsigned long long)low_index[i],
               is_largest[i]);
        for (int k = 31; k >= 0; k--) printf("%02x", new_root[i][k]);
        printf("\n");
    }
    unsigned char fail[N];
    rc = imt_insert_witness_batch(ctx, old_root, low_leaf, low_index, low_sib, new_root, new_leaf, new_index, NULL,
                                  new_sib, is_largest, DEPTH, N, fail, NULL, IMT_FMT_CANONICAL);
    if (rc) { fprintf(stderr, "witness: %s\n", imt_last_error(ctx)); return 1; }
    int bad = 0;
    for (int i = 0; i < N; i++) bad |= fail[i];
    printf("insert_leaf constraints: %s\n", bad ? "VIOLATED" : "all satisfied");

    /* f1: every new advice value of the 15 hashes of each insert_leaf, one insertion's rows next to each other */
    const size_t rows = imt_insert_trace_rows(DEPTH);
    unsigned char(*trace)[32] = malloc(N * rows * 32);
    unsigned char low_sib_im[N][DEPTH][32], new_sib_im[N][DEPTH][32];      /* item-major proofs for this call */
    for (int i = 0; i < N; i++)
        for (int l = 0; l < DEPTH; l++) {
            memcpy(low_sib_im[i][l], low_sib[l][i], 32);
            memcpy(new_sib_im[i][l], new_sib[l][i], 32);
        }
    rc = imt_insert_trace_batch(ctx, low_leaf, low_index, low_sib_im, new_leaf, new_index, NULL, new_sib_im, DEPTH, N,
                                trace, IMT_FMT_CANONICAL | IMT_TRACE_ITEM_MAJOR);
    if (rc) { fprintf(stderr, "trace: %s\n", imt_last_error(ctx)); return 1; }
    size_t n_cells = 0, n_consts = 0;
    uint32_t out_row = 0;
    rc = imt_hash_trace_layout(ctx, 2, NULL, 0, &n_cells, NULL, 0, &n_consts, &out_row, IMT_FMT_CANONICAL);
    if (rc) { fprintf(stderr, "layout: %s\n", imt_last_error(ctx)); return 1; }
    int trace_ok = rows == 3 * imt_hash_trace_rows(3) + 4 * DEPTH * imt_hash_trace_rows(2);
    for (int i = 0; i < N; i++) {
        /* the last hash of the last path of insertion i is the new root: its output row is new_root[i] */
        const unsigned char *last = trace[(size_t)i * rows + rows - imt_hash_trace_rows(2) + out_row];
        trace_ok &= memcmp(last, new_root[i], 32) == 0;
        /* ... and the second block (rewritten low leaf) ends in the interim root */
        const size_t blk = imt_hash_trace_rows(3) + DEPTH * imt_hash_trace_rows(2);
        trace_ok &= memcmp(trace[(size_t)i * rows + 2 * blk - imt_hash_trace_rows(2) + out_row], interim_root[i], 32) == 0;
    }
    printf("witness trace: %zu rows per insert_leaf, %zu cells / %zu constants per 2-input hash, output row %u: %s\n", rows,
           n_cells, n_consts, out_row, trace_ok ? "trace rows ok" : "MISMATCH");
    bad |= !trace_ok;
    free(trace);
    /* f4: checkpoint / resume -- the leaves {val, next_val, next_idx} in index order (the reference's serde leaf,
     * src/utils.rs:12-17) read from the device index, loaded into a second tree (list check + rebuild on the GPU) */
    const uint64_t size = imt_itree_size(tree);
    unsigned char(*snap)[3][32] = malloc(size * 96);
    unsigned char root_a[32], root_b[32];
    imt_itree *resumed = NULL;
    rc = imt_itree_get_leaves(tree, NULL, size, snap, IMT_FMT_CANONICAL);
    if (!rc) rc = imt_itree_new(ctx, DEPTH, 8, &resumed);
    if (!rc) rc = imt_itree_load(resumed, snap, size, IMT_FMT_CANONICAL);
    if (!rc) rc = imt_itree_root(tree, root_a, IMT_FMT_CANONICAL);
    if (!rc) rc = imt_itree_root(resumed, root_b, IMT_FMT_CANONICAL);
    if (rc) { fprintf(stderr, "checkpoint: %s\n", imt_last_error(ctx)); return 1; }
    snap[2][1][0] ^= 1;                                  /* leaf 2 no longer points to its successor */
    const int refused = imt_itree_load(resumed, snap, size, IMT_FMT_CANONICAL) == IMT_ERR_VALUE;
    printf("checkpoint: %llu leaves reloaded, root %s; corrupted snapshot %s (%s)\n", (unsigned long long)size,
           memcmp(root_a, root_b, 32) ? "DIFFERS" : "equal", refused ? "refused" : "ACCEPTED", imt_last_error(ctx));
    bad |= memcmp(root_a, root_b, 32) != 0 || !refused;
    free(snap);
    imt_itree_free(resumed);
    imt_itree_free(tree);
    imt_ctx_destroy(ctx);
    return bad != 0;
}
