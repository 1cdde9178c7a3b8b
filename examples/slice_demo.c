/* slice_demo.c -- plain C against include/imt.h: the reference's ONE sorted list (update_idx_leaf's sequential
 * semantics, src/indexed_merkle_tree.rs:632-660) kept by TWO replicas that each hash half of every step
 * (imt_itree_slice_prepare / _unit / _apply), with the simplest legal schedule: the slices one after the other, every
 * unit's payload applied to the other replica before the next slice starts.  Buffers come from imt_host_alloc
 * (page-locked, device-addressable: valid wherever a device pointer is expected), so the "transport" between the
 * replicas is a pointer; a multi-GPU host puts ncclAllGather there and overlaps the slices as imt::SliceSchedule
 * (include/imt.hpp) says.  The replicas' roots must equal a third, ordinary tree's over the same values.  Build:
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
    /* per replica: the roots of its slice (the other outputs work the same way) and one payload buffer */
    unsigned char(*new_root[WORLD])[32];
    unsigned char *payload[WORLD];
    const size_t pay_bytes = imt_itree_slice_payload_bytes(SLICE);
    for (int g = 0; g < WORLD; g++) {
        CHECK(imt_host_alloc(ctx[g], SLICE * 32, (void **)&new_root[g]));
        CHECK(imt_host_alloc(ctx[g], pay_bytes, (void **)&payload[g]));
    }
    for (int s = 0; s < STEPS; s++) {
        const unsigned char(*step_vals)[32] = (const unsigned char(*)[32])vals[s * WORLD * SLICE];
        const uint64_t size_before = imt_itree_size(tree[0]);
        int slice[WORLD];
        /* every replica sees the whole step: index work for all of it, events for its own slice */
        for (int g = 0; g < WORLD; g++) {
            imt_insert_out out;
            memset(&out, 0, sizeof out);
            out.new_root = new_root[g];
            CHECK(imt_itree_slice_prepare(tree[g], step_vals, (size_t)g * SLICE, SLICE, (size_t)(WORLD - 1 - g) * SLICE, &out,
                                          IMT_DEVICE_PTRS, &slice[g], NULL));
        }
        /* the slices in insertion order; a unit's payload reaches the other replica before anything later runs */
        for (int g = 0; g < WORLD; g++)
            for (unsigned q = 0; q <= DEPTH; q++) {
                CHECK(imt_itree_slice_unit(tree[g], slice[g], q, payload[g], NULL));
                CHECK(imt_ctx_sync(ctx[g]));
                if (imt_itree_slice_unit_bytes(tree[g], size_before + (uint64_t)g * SLICE, SLICE, q) > pay_bytes) return 2;
                for (int h = 0; h < WORLD; h++)
                    if (h != g) {
                        CHECK(imt_itree_slice_apply(tree[h], size_before + (uint64_t)g * SLICE, SLICE, q, payload[g], NULL));
                        CHECK(imt_ctx_sync(ctx[h]));
                    }
            }
    }
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
    same &= memcmp(new_root[WORLD - 1][SLICE - 1], root[WORLD], 32) == 0;      /* the last slice's last new root */
    printf("replicas %s the one-GPU tree (%llu leaves)\n", same ? "equal" : "DIFFER FROM", (unsigned long long)imt_itree_size(tree[0]));
    for (int g = 0; g < WORLD; g++) {
        imt_host_free(ctx[g], new_root[g]);
        imt_host_free(ctx[g], payload[g]);
    }
    imt_host_free(ctx[0], vals);
    for (int g = 0; g <= WORLD; g++) {
        imt_itree_free(tree[g]);
        imt_ctx_destroy(ctx[g]);
    }
    return same ? 0 : 3;
}
