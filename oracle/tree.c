/*
 * tree.c -- CPU ORACLE (test infrastructure): the reference's dense native tree,
 * IndexedMerkleTree::{new,get_root,get_proof,verify_proof} (src/utils.rs:19-108),
 * restated in C.  Err strings / panics become the ORC_ERR_* codes.
 */
#include "imt_oracle.h"
#include <stdlib.h>
#include <string.h>

struct orc_tree {
    size_t n_levels;   /* tree.len() */
    size_t *len;       /* len[i] = tree[i].len() */
    ofr_t **lvl;
    ofr_t root;
};

void orc_tree_free(orc_tree *t) {
    if (!t) return;
    if (t->lvl)
        for (size_t i = 0; i < t->n_levels; i++) free(t->lvl[i]);
    free(t->lvl);
    free(t->len);
    free(t);
}

/* src/utils.rs:20-57 */
int orc_tree_new(orc_tree **out, const uint8_t *leaves, size_t n) {
    *out = NULL;
    if (n == 0) return ORC_ERR_NO_LEAVES;               /* :24-26 */
    if (n != 1 && (n % 2) == 1) return ORC_ERR_ODD_LEAVES; /* :34-36 (after the len==1 case) */
    /* an even, non-power-of-two length reaches an odd level > 1 and indexes
       current_level[i + 1] out of bounds (:45): a panic in the reference */
    if (n & (n - 1)) return ORC_ERR_NOT_POW2;
    size_t nl = 1;
    for (size_t m = n; m > 1; m >>= 1) nl++;
    orc_tree *t = calloc(1, sizeof *t);
    if (!t) return ORC_ERR_ALLOC;
    t->n_levels = nl;
    t->len = calloc(nl, sizeof *t->len);
    t->lvl = calloc(nl, sizeof *t->lvl);
    if (!t->len || !t->lvl) { orc_tree_free(t); return ORC_ERR_ALLOC; }
    t->len[0] = n;
    t->lvl[0] = malloc(n * sizeof(ofr_t));
    if (!t->lvl[0]) { orc_tree_free(t); return ORC_ERR_ALLOC; }
    for (size_t i = 0; i < n; i++)
        if (ofr_from_bytes(&t->lvl[0][i], leaves + 32 * i)) { orc_tree_free(t); return ORC_ERR_NONCANONICAL; }
    for (size_t l = 1; l < nl; l++) {                   /* while current_level.len() > 1 :41 */
        size_t m = t->len[l - 1] / 2;
        t->len[l] = m;
        t->lvl[l] = malloc(m * sizeof(ofr_t));
        if (!t->lvl[l]) { orc_tree_free(t); return ORC_ERR_ALLOC; }
        for (size_t i = 0; i < m; i++)                  /* :43-48 */
            orc_hash2_fr(&t->lvl[l][i], &t->lvl[l - 1][2 * i], &t->lvl[l - 1][2 * i + 1]);
    }
    t->root = t->lvl[nl - 1][0];                        /* :27-33 covers n == 1 */
    *out = t;
    return ORC_OK;
}

size_t orc_tree_num_levels(const orc_tree *t) { return t->n_levels; }
void orc_tree_get_root(const orc_tree *t, uint8_t root[32]) { ofr_to_bytes(root, &t->root); }

int orc_tree_level(const orc_tree *t, size_t level, uint8_t *out, size_t *n) {
    if (level >= t->n_levels) return ORC_ERR_RANGE;
    if (n) *n = t->len[level];
    if (out)
        for (size_t i = 0; i < t->len[level]; i++) ofr_to_bytes(out + 32 * i, &t->lvl[level][i]);
    return ORC_OK;
}

/* src/utils.rs:63-85: sibling = idx^1; helper = 1 iff the node is a LEFT child (:79) */
int orc_tree_get_proof(const orc_tree *t, size_t index, uint8_t *proof, uint8_t *helper) {
    size_t cur = index;
    for (size_t i = 0; i + 1 < t->n_levels; i++) {
        int is_left = (cur % 2) == 0;
        size_t sib = is_left ? cur + 1 : cur - 1;
        if (sib >= t->len[i]) return ORC_ERR_RANGE;
        ofr_to_bytes(proof + 32 * i, &t->lvl[i][sib]);
        memset(helper + 32 * i, 0, 32);
        helper[32 * i] = is_left ? 1 : 0;
        cur /= 2;
    }
    return ORC_OK;
}

/* src/utils.rs:87-107 */
int orc_path_root(uint8_t root_out[32], const uint8_t leaf[32], uint64_t index,
                  const uint8_t *proof, size_t depth) {
    ofr_t h, s;
    if (ofr_from_bytes(&h, leaf)) return ORC_ERR_NONCANONICAL;
    uint64_t cur = index;
    for (size_t i = 0; i < depth; i++) {
        if (ofr_from_bytes(&s, proof + 32 * i)) return ORC_ERR_NONCANONICAL;
        if ((cur % 2) == 0) orc_hash2_fr(&h, &h, &s);
        else orc_hash2_fr(&h, &s, &h);
        cur /= 2;
    }
    ofr_to_bytes(root_out, &h);
    return ORC_OK;
}

int orc_verify_proof(const uint8_t leaf[32], uint64_t index, const uint8_t root[32],
                     const uint8_t *proof, size_t depth) {
    uint8_t r[32];
    int rc = orc_path_root(r, leaf, index, proof, depth);
    if (rc) return rc;
    return memcmp(r, root, 32) == 0 ? 1 : 0;
}
