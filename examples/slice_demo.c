/* slice_demo.c -- plain C against include/imt.h: the reference's ONE sorted list (update_idx_leaf's sequential
 * semantics, src/indexed_merkle_tree.rs:632-660) kept by TWO replicas that each hash half of every step, through
 * imt_sliced_create / imt_sliced_step / imt_sliced_flush: the schedule, its streams and events and the exchange of each
 * slice's per-level write-backs are the library's business; the host makes one call per step.  Both replicas live in
 * this process on one GPU (the local transport); one process per GPU passes imt_transport_rccl_create's handle instead
 * and nothing else changes.  The replicas' roots must equal a third, ordinary tree's over the same values.  Build:
 *   gcc -std=c11 -I include examples/slice_demo.c -L indexed-merkle-tree-halo2_amd/csrc -limt_hip -o slice_demo
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "imt.h"

#define DEPTH 32
#define WORLD 2
#define SLICE 8            /* insertions per replica and step */
#define STEPS 3

#define CHECK(call)                                                                         \
    do {                                                                                    \
        int rc__ = (call);                                                                  \
        if (rc__) { fprintf(stderr, "%s: %d %s\n", #call, rc__, imt_last_error(ctx[0])); return 1; } \
    } while (0)

int main(void) {
    imt_ctx *ctx[WORLD + 1] = {0};
    imt_itree *tree[WORLD + 1] = {0};
    for (int g = 0; g <= WORLD; g++) {
        int rc = imt_ctx_create(0, &ctx[g]);
        if (rc) { fprintf(stderr, "imt_ctx_create: %d (no GPU?)\n", rc); return 1; }
        CHECK(imt_itree_new(ctx[g], DEPTH, 256, &tree[g]));
    }
    /* distinct non-zero values, canonical little-endian, in memory every context's kernels can read */
    unsigned char(*vals)[32];
    CHECK(imt_host_alloc(ctx[0], (size_t)STEPS * WORLD * SLICE * 32, (void **)&vals));
    memset(vals, 0, (size_t)STEPS * WORLD * SLICE * 32);
    unsigned state = 12345u;
    for (int i = 0; i < STEPS * WORLD * SLICE; i++) {
        state = state * 1103515245u + 12345u;
        vals[i][0] = (unsigned char)(i + 1);           /* distinct */
        memcpy(vals[i] + 4, &state, 4);
        vals[i][20] = (unsigned char)(state >> 9);
    }
    /* per replica and step: the new roots of its slice (the other outputs work the same way) */
    unsigned char(*new_root[WORLD])[32];
    for (int g = 0; g < WORLD; g++) CHECK(imt_host_alloc(ctx[g], (size_t)STEPS * SLICE * 32, (void **)&new_root[g]));
    imt_transport *tp = NULL;
    imt_sliced *world = NULL;
    CHECK(imt_transport_local_create(&tp));
    CHECK(imt_sliced_create(tree, WORLD, WORLD, 0, tp, SLICE, 0, &world));
    for (int s = 0; s < STEPS; s++) {
        imt_insert_out outs[WORLD];
        memset(outs, 0, sizeof outs);
        for (int g = 0; g < WORLD; g++) outs[g].new_root = new_root[g][s * SLICE];
        uint64_t round = 0;
        int rc = imt_sliced_step(world, vals[s * WORLD * SLICE], SLICE, outs, 0, &round);
        if (rc || round != (uint64_t)s) { fprintf(stderr, "imt_sliced_step: %d %s\n", rc, imt_sliced_last_error(world)); return 1; }
    }
    CHECK(imt_sliced_flush(world));
    imt_sliced_info info;
    CHECK(imt_sliced_get_info(world, &info));
    /* the ordinary tree over the same values */
    CHECK(imt_itree_insert_batch(tree[WORLD], vals, (size_t)STEPS * WORLD * SLICE, NULL, IMT_DEVICE_PTRS));
    CHECK(imt_ctx_sync(ctx[WORLD]));
    unsigned char root[WORLD + 1][32];
    for (int g = 0; g <= WORLD; g++) CHECK(imt_itree_root(tree[g], root[g], IMT_FMT_CANONICAL));
    printf("root ");
    for (int k = 31; k >= 0; k--) printf("%02x", root[WORLD][k]);
    printf("\n");
    int same = 1;
    for (int g = 0; g < WORLD; g++) same &= memcmp(root[g], root[WORLD], 32) == 0;
    same &= memcmp(new_root[WORLD - 1][STEPS * SLICE - 1], root[WORLD], 32) == 0;      /* the last slice's last new root */
    printf("replicas %s the one-GPU tree (%llu leaves; lag %d, %llu all-gathers)\n", same ? "equal" : "DIFFER FROM",
           (unsigned long long)imt_itree_size(tree[0]), info.lag, (unsigned long long)info.collectives);
    imt_sliced_destroy(world);
    imt_transport_destroy(tp);
    for (int g = 0; g < WORLD; g++) imt_host_free(ctx[g], new_root[g]);
    imt_host_free(ctx[0], vals);
    for (int g = 0; g <= WORLD; g++) {
        imt_itree_free(tree[g]);
        imt_ctx_destroy(ctx[g]);
    }
    return same ? 0 : 3;
}
