/*
 * sparse.c -- CPU ORACLE (test infrastructure): depth-d append-only indexed tree.
 *
 * Same node values as the reference's dense builder (src/utils.rs:38-51) applied to
 * 2^d leaves of which only a prefix is non-empty: an empty slot is H(0,0,0)
 * (src/indexed_merkle_tree.rs:373-376, :247-251) and an all-empty subtree of height
 * l is Z[l], Z[l+1] = H(Z[l],Z[l]).  One insertion follows the test module's
 * update_idx_leaf + rebuild + get_proof sequence (:632-660, :715-735) but touches
 * only the two changed paths (2 + 2d hashes).  This is the depth-32 CPU baseline:
 * the dense Vec<Vec<F>> of the reference needs 256 GiB there.
 */
#include "imt_oracle.h"
#include <stdlib.h>
#include <string.h>

struct orc_sparse {
    unsigned depth;
    uint64_t cap, size;
    uint64_t *nlen;       /* stored nodes per level */
    ofr_t **lvl;          /* [depth+1] */
    ofr_t *zero;          /* Z[0..depth] */
    uint8_t *pre;         /* [cap][3][32] canonical preimages */
    uint64_t *sorted;     /* leaf indices ordered by val */
    uint64_t index_base;  /* added to every next_idx that is hashed into a leaf (subtree of a deeper tree) */
};

static void zero_table(ofr_t *z, unsigned depth) {
    ofr_t o;
    memset(&o, 0, sizeof o);
    orc_hash3_fr(&z[0], &o, &o, &o);
    for (unsigned l = 0; l < depth; l++) orc_hash2_fr(&z[l + 1], &z[l], &z[l]);
}

void orc_zero_hashes(uint8_t *out, unsigned depth) {
    ofr_t *z = malloc((depth + 1) * sizeof *z);
    zero_table(z, depth);
    for (unsigned l = 0; l <= depth; l++) ofr_to_bytes(out + 32 * l, &z[l]);
    free(z);
}

void orc_sparse_free(orc_sparse *t) {
    if (!t) return;
    if (t->lvl)
        for (unsigned l = 0; l <= t->depth; l++) free(t->lvl[l]);
    free(t->lvl); free(t->nlen); free(t->zero); free(t->pre); free(t->sorted);
    free(t);
}

int orc_sparse_new(orc_sparse **out, unsigned depth, uint64_t cap) {
    *out = NULL;
    if (depth == 0 || depth > 63 || cap < 2 || (cap & (cap - 1))) return ORC_ERR_RANGE;
    if (depth < 63 && cap > ((uint64_t)1 << depth)) return ORC_ERR_RANGE;
    orc_sparse *t = calloc(1, sizeof *t);
    if (!t) return ORC_ERR_ALLOC;
    t->depth = depth; t->cap = cap;
    t->nlen = calloc(depth + 1, sizeof *t->nlen);
    t->lvl = calloc(depth + 1, sizeof *t->lvl);
    t->zero = calloc(depth + 1, sizeof *t->zero);
    t->pre = calloc(cap, 96);
    t->sorted = calloc(cap, sizeof *t->sorted);
    if (!t->nlen || !t->lvl || !t->zero || !t->pre || !t->sorted) { orc_sparse_free(t); return ORC_ERR_ALLOC; }
    zero_table(t->zero, depth);
    for (unsigned l = 0; l <= depth; l++) {
        uint64_t n = cap >> l;
        if (n == 0) n = 1;
        t->nlen[l] = n;
        t->lvl[l] = malloc(n * sizeof(ofr_t));
        if (!t->lvl[l]) { orc_sparse_free(t); return ORC_ERR_ALLOC; }
        for (uint64_t i = 0; i < n; i++) t->lvl[l][i] = t->zero[l];
    }
    t->size = 1;          /* leaf 0 = the {0,0,0} sentinel; its hash is Z[0] */
    t->sorted[0] = 0;
    *out = t;
    return ORC_OK;
}

/* The tree is subtree number (base >> depth) of a deeper tree: leaf j of it is leaf base + j there, and
 * that global index is what a leaf's next_idx field holds (new_val_idx at :655, :715).  Positions passed
 * to / returned by this API stay local. */
void orc_sparse_set_index_base(orc_sparse *t, uint64_t base) { t->index_base = base; }

void orc_sparse_root(const orc_sparse *t, uint8_t root[32]) { ofr_to_bytes(root, &t->lvl[t->depth][0]); }
uint64_t orc_sparse_size(const orc_sparse *t) { return t->size; }

static const ofr_t *sibling(const orc_sparse *t, unsigned l, uint64_t node) {
    uint64_t s = node ^ 1;
    return s < t->nlen[l] ? &t->lvl[l][s] : &t->zero[l];
}

int orc_sparse_proof(const orc_sparse *t, uint64_t index, uint8_t *proof) {
    if (index >= t->cap) return ORC_ERR_RANGE;
    for (unsigned l = 0; l < t->depth; l++) ofr_to_bytes(proof + 32 * l, sibling(t, l, index >> l));
    return ORC_OK;
}

int orc_sparse_preimage(const orc_sparse *t, uint64_t index, uint8_t out[3][32]) {
    if (index >= t->cap) return ORC_ERR_RANGE;
    memcpy(out, t->pre + 96 * index, 96);
    return ORC_OK;
}

static void set_leaf(orc_sparse *t, uint64_t index, const ofr_t *h) {
    ofr_t cur = *h;
    t->lvl[0][index] = cur;
    for (unsigned l = 0; l < t->depth; l++) {
        uint64_t node = index >> l;
        const ofr_t *s = sibling(t, l, node);
        if ((node & 1) == 0) orc_hash2_fr(&cur, &cur, s);   /* same order as utils.rs:95-101 */
        else orc_hash2_fr(&cur, s, &cur);
        t->lvl[l + 1][node >> 1] = cur;
    }
}

static int cmp_bytes(const uint8_t *a, const uint8_t *b) {
    for (int i = 31; i >= 0; i--) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}

/* position in sorted[] of the greatest val < v; -1 if v is 0; -2 if v already present */
static int64_t find_pred(const orc_sparse *t, const uint8_t v[32]) {
    uint64_t lo = 0, hi = t->size; /* first position with val >= v */
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (cmp_bytes(t->pre + 96 * t->sorted[mid], v) < 0) lo = mid + 1;
        else hi = mid;
    }
    if (lo < t->size && cmp_bytes(t->pre + 96 * t->sorted[lo], v) == 0) return -2;
    if (lo == 0) return -1;
    return (int64_t)lo - 1;
}

int orc_sparse_find_low(const orc_sparse *t, const uint8_t val[32], uint64_t *low_idx) {
    ofr_t chk;
    if (ofr_from_bytes(&chk, val)) return -10;
    int64_t p = find_pred(t, val);
    if (p < 0) return -10;
    *low_idx = t->sorted[p];
    return ORC_OK;
}

static void put_u64(uint8_t *o, uint64_t v) {
    memset(o, 0, 32);
    for (int k = 0; k < 8; k++) o[k] = (uint8_t)(v >> (8 * k));
}

int orc_sparse_insert(orc_sparse *t, const uint8_t val[32], uint64_t *low_idx,
                      uint8_t low_leaf[3][32], int *is_largest, uint8_t interim_root[32],
                      uint8_t new_root[32], uint8_t *low_proof, uint8_t *new_proof) {
    ofr_t chk, a, b, c, h;
    if (ofr_from_bytes(&chk, val)) return -10;
    if (t->size >= t->cap) return -11;
    int64_t p = find_pred(t, val);
    if (p < 0) return -10;
    uint64_t low = t->sorted[p], idx = t->size;   /* new_val_idx = insertion ordinal :715 */
    uint8_t *lp = t->pre + 96 * low, *np = t->pre + 96 * idx;
    if (low_idx) *low_idx = low;
    if (low_leaf) memcpy(low_leaf, lp, 96);
    int largest = 1;
    for (int i = 0; i < 32; i++) if (lp[32 + i]) largest = 0;   /* :737-742 */
    if (is_largest) *is_largest = largest;
    if (low_proof) orc_sparse_proof(t, low, low_proof);          /* get_proof before the change :722 */
    /* update_idx_leaf :647-657 */
    memcpy(np, val, 32);
    memcpy(np + 32, lp + 32, 64);
    memcpy(lp + 32, val, 32);
    put_u64(lp + 64, t->index_base + idx);
    ofr_from_bytes(&a, lp); ofr_from_bytes(&b, lp + 32); ofr_from_bytes(&c, lp + 64);
    orc_hash3_fr(&h, &a, &b, &c);
    set_leaf(t, low, &h);
    if (interim_root) orc_sparse_root(t, interim_root);
    if (new_proof) orc_sparse_proof(t, idx, new_proof);          /* siblings of the new slot :734 */
    ofr_from_bytes(&a, np); ofr_from_bytes(&b, np + 32); ofr_from_bytes(&c, np + 64);
    orc_hash3_fr(&h, &a, &b, &c);
    set_leaf(t, idx, &h);
    if (new_root) orc_sparse_root(t, new_root);
    memmove(&t->sorted[p + 2], &t->sorted[p + 1], (t->size - (uint64_t)p - 1) * sizeof(uint64_t));
    t->sorted[p + 1] = idx;
    t->size++;
    return ORC_OK;
}

/* The test module's "rebuild from the preimages": hash_nullifier_pre_images (:662-671) + IndexedMerkleTree::new
 * (src/utils.rs:38-51), on a tree whose first n leaves hold preimages[n][3][32] and whose other slots are empty
 * (H(0,0,0), :373-376).  The list is taken as given except for what orc_sparse_insert relies on: leaf 0 is the sentinel
 * (val 0), values are canonical and pairwise different, every next_val / next_idx names the next larger value's leaf
 * (0 / 0 at the largest).  Used to start a sequential run at a checkpoint (tests/golden/make_config4_digest.py, which
 * also checks that the root loaded here equals the root the sequential run before it ended with).  -10: not such a list. */
static const orc_sparse *g_sort_tree;
static int cmp_leaf_by_val(const void *x, const void *y) {
    uint64_t a = *(const uint64_t *)x, b = *(const uint64_t *)y;
    return cmp_bytes(g_sort_tree->pre + 96 * a, g_sort_tree->pre + 96 * b);
}

int orc_sparse_load(orc_sparse *t, const uint8_t *preimages, uint64_t n) {
    if (n < 1 || n > t->cap || t->size != 1) return ORC_ERR_RANGE;
    memcpy(t->pre, preimages, 96 * n);
    for (uint64_t i = 0; i < n; i++) t->sorted[i] = i;
    g_sort_tree = t;
    qsort(t->sorted, n, sizeof *t->sorted, cmp_leaf_by_val);
    if (t->sorted[0] != 0) return -10;
    for (int k = 0; k < 32; k++) if (t->pre[k]) return -10;
    for (uint64_t j = 0; j < n; j++) {
        const uint8_t *me = t->pre + 96 * t->sorted[j];
        uint8_t want_idx[32];
        ofr_t chk;
        if (ofr_from_bytes(&chk, me)) return -10;
        if (j + 1 < n) {
            const uint8_t *nx = t->pre + 96 * t->sorted[j + 1];
            if (cmp_bytes(me, nx) == 0) return -10;
            put_u64(want_idx, t->index_base + t->sorted[j + 1]);
            if (memcmp(me + 32, nx, 32) || memcmp(me + 64, want_idx, 32)) return -10;
        } else {
            memset(want_idx, 0, 32);
            if (memcmp(me + 32, want_idx, 32) || memcmp(me + 64, want_idx, 32)) return -10;
        }
    }
    for (uint64_t i = 0; i < n; i++) {                       /* :662-671 */
        ofr_t a, b, c;
        const uint8_t *q = t->pre + 96 * i;
        ofr_from_bytes(&a, q); ofr_from_bytes(&b, q + 32); ofr_from_bytes(&c, q + 64);
        orc_hash3_fr(&t->lvl[0][i], &a, &b, &c);
    }
    uint64_t filled = n;                                     /* utils.rs:38-51, only where a child is not all-empty */
    for (unsigned l = 0; l < t->depth; l++) {
        uint64_t parents = (filled + 1) / 2;
        for (uint64_t k = 0; k < parents; k++) {
            const ofr_t *left = &t->lvl[l][2 * k];
            const ofr_t *right = 2 * k + 1 < t->nlen[l] ? &t->lvl[l][2 * k + 1] : &t->zero[l];
            orc_hash2_fr(&t->lvl[l + 1][k], left, right);
        }
        filled = parents;
    }
    t->size = n;
    return ORC_OK;
}
