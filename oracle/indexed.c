/*
 * indexed.c -- CPU ORACLE (test infrastructure): witness VALUES and constraint
 * relations of the circuit gadget (src/indexed_merkle_tree.rs:33-314) and the
 * indexed-list insertion of its test module (:632-671), restated in C.
 * A reference constraint / assert that would fail is reported as a bit in the
 * returned mask instead of a panic or an unsatisfied MockProver.
 */
#include "imt_oracle.h"
#include <string.h>

static int is_bit(const ofr_t *x, const ofr_t *one) { return ofr_is_zero(x) || ofr_eq(x, one); }

/* dual_mux :47-63: left=(a-b)*s+b, right=(b-a)*s+a */
static void dual_mux(ofr_t *l, ofr_t *r, const ofr_t *a, const ofr_t *b, const ofr_t *s) {
    ofr_t d, t;
    ofr_sub(&d, a, b);
    ofr_mul(&t, &d, s);
    ofr_add(l, &t, b);
    ofr_sub(&d, b, a);
    ofr_mul(&t, &d, s);
    ofr_add(r, &t, a);
}

/* compute_merkle_root :78-96 */
static int merkle_root_fr(ofr_t *out, const ofr_t *leaf, const uint8_t *proof,
                          const uint8_t *helper, size_t depth, int *bad_bit) {
    ofr_t one, cur = *leaf, sib, s, l, r;
    ofr_from_u64(&one, 1);
    for (size_t i = 0; i < depth; i++) {
        if (ofr_from_bytes(&sib, proof + 32 * i)) return ORC_ERR_NONCANONICAL;
        if (ofr_from_bytes(&s, helper + 32 * i)) return ORC_ERR_NONCANONICAL;
        if (!is_bit(&s, &one)) *bad_bit = 1;           /* gate.assert_bit :54 */
        dual_mux(&l, &r, &cur, &sib, &s);
        orc_hash2_fr(&cur, &l, &r);                    /* hash_fix_len_array :92 */
    }
    *out = cur;
    return ORC_OK;
}

int orc_compute_merkle_root(uint8_t root_out[32], const uint8_t leaf[32], const uint8_t *proof,
                            const uint8_t *helper, size_t depth) {
    ofr_t l, o;
    int bad = 0;
    if (ofr_from_bytes(&l, leaf)) return ORC_ERR_NONCANONICAL;
    int rc = merkle_root_fr(&o, &l, proof, helper, depth, &bad);
    if (rc) return rc;
    ofr_to_bytes(root_out, &o);
    return bad ? ORC_F_BAD_BIT : ORC_OK;
}

/* split at 2^128 (:145-178) + is_less_than (:98-125) */
static int lt128(const uint8_t a[16], const uint8_t b[16]) {
    for (int i = 15; i >= 0; i--) {
        if (a[i] < b[i]) return 1;
        if (a[i] > b[i]) return 0;
    }
    return 0;
}
int orc_is_less_than_limbs(const uint8_t av[32], const uint8_t bv[32]) {
    const uint8_t *a_r = av, *a_q = av + 16, *b_r = bv, *b_q = bv + 16;
    int a = lt128(a_q, b_q);                    /* is_ll_msb_gr */
    int c_not = memcmp(a_q, b_q, 16) != 0;      /* !are_msb_eq */
    int a_not = !a;
    int b = lt128(a_r, b_r);                    /* is_ll_lsb_gr */
    int c = !c_not;
    int d_not = memcmp(a_r, b_r, 16) != 0;      /* !are_lsb_eq */
    int rhs = a_not & b & c & d_not;
    int lhs = a & c_not;
    return lhs | rhs;
}

static int load3(ofr_t o[3], const uint8_t in[3][32]) {
    for (int i = 0; i < 3; i++)
        if (ofr_from_bytes(&o[i], in[i])) return ORC_ERR_NONCANONICAL;
    return ORC_OK;
}

/* verify_non_inclusion :127-229 */
static int non_inclusion(const uint8_t root[32], const uint8_t low_leaf[3][32],
                         const uint8_t *low_proof, const uint8_t *low_helper, size_t depth,
                         const uint8_t new_val[32], int largest, ofr_t *low_hash, ofr_t *root_calc) {
    ofr_t low[3], nv, rt;
    int fail = 0, bad = 0;
    if (load3(low, low_leaf) || ofr_from_bytes(&nv, new_val) || ofr_from_bytes(&rt, root))
        return ORC_ERR_NONCANONICAL;
    int is_zero = ofr_is_zero(&low[1]);                         /* :143 */
    int next_gr = orc_is_less_than_limbs(new_val, low_leaf[1]); /* :180 */
    if (largest != 0 && largest != 1) fail |= ORC_F_BAD_BIT;    /* assert_bit in select :41 */
    int is_true = largest ? is_zero : next_gr;                  /* select :182-189 */
    if (!is_true) fail |= ORC_F_RANGE_PRED;                     /* :190-191 */
    orc_hash3_fr(low_hash, &low[0], &low[1], &low[2]);          /* :193-194 */
    int rc = merkle_root_fr(root_calc, low_hash, low_proof, low_helper, depth, &bad);
    if (rc) return rc;
    if (bad) fail |= ORC_F_BAD_BIT;
    if (!ofr_eq(root_calc, &rt)) fail |= ORC_F_LOW_IN_ROOT;     /* :196-204 */
    if (!orc_is_less_than_limbs(low_leaf[0], new_val)) fail |= ORC_F_LOW_LT_NEW; /* :226-228 */
    return fail;
}

int orc_verify_non_inclusion(const uint8_t root[32], const uint8_t low_leaf[3][32],
                             const uint8_t *low_proof, const uint8_t *low_helper, size_t depth,
                             const uint8_t new_val[32], int largest,
                             uint8_t low_leaf_hash_out[32], uint8_t root_out[32]) {
    ofr_t lh, rc_;
    int f = non_inclusion(root, low_leaf, low_proof, low_helper, depth, new_val, largest, &lh, &rc_);
    if (f < 0) return f;
    if (low_leaf_hash_out) ofr_to_bytes(low_leaf_hash_out, &lh);
    if (root_out) ofr_to_bytes(root_out, &rc_);
    return f;
}

/* insert_leaf :231-314 */
int orc_insert_leaf(const uint8_t old_root[32], const uint8_t low_leaf[3][32],
                    const uint8_t *low_proof, const uint8_t *low_helper,
                    const uint8_t new_root[32], const uint8_t new_leaf[3][32],
                    uint64_t new_leaf_index, const uint8_t *new_proof, const uint8_t *new_helper,
                    int largest, size_t depth, orc_insert_trace *tr) {
    ofr_t low[3], nl[3], lh, r0, nlh, interim, zroot, nh, nr, nroot_in, zero, zh, idx;
    int bad = 0;
    if (load3(low, low_leaf) || load3(nl, new_leaf) || ofr_from_bytes(&nroot_in, new_root))
        return ORC_ERR_NONCANONICAL;
    int fail = non_inclusion(old_root, low_leaf, low_proof, low_helper, depth, new_leaf[0],
                             largest, &lh, &r0);                /* :253-263 */
    if (fail < 0) return fail;
    /* newlowleaf {low.val, new.val, new_leaf_index} :265-275 */
    ofr_from_u64(&idx, new_leaf_index);
    orc_hash3_fr(&nlh, &low[0], &nl[0], &idx);
    int rc = merkle_root_fr(&interim, &nlh, low_proof, low_helper, depth, &bad); /* :277-284 */
    if (rc) return rc;
    /* zero leaf (the constant :247-251 == H(0,0,0)) must sit at the new slot :286-294 */
    memset(&zero, 0, sizeof zero);
    orc_hash3_fr(&zh, &zero, &zero, &zero);
    rc = merkle_root_fr(&zroot, &zh, new_proof, new_helper, depth, &bad);
    if (rc) return rc;
    if (!ofr_eq(&zroot, &interim)) fail |= ORC_F_ZERO_SLOT;
    if (!ofr_eq(&nl[1], &low[1])) fail |= ORC_F_NEXT_VAL;       /* :296 */
    if (!ofr_eq(&nl[2], &low[2])) fail |= ORC_F_NEXT_IDX;       /* :297 */
    orc_hash3_fr(&nh, &nl[0], &nl[1], &nl[2]);                  /* :299-303 */
    rc = merkle_root_fr(&nr, &nh, new_proof, new_helper, depth, &bad); /* :305-312 */
    if (rc) return rc;
    if (!ofr_eq(&nr, &nroot_in)) fail |= ORC_F_NEW_ROOT;        /* :313 */
    if (bad) fail |= ORC_F_BAD_BIT;
    if (tr) {
        ofr_to_bytes(tr->low_leaf_hash, &lh);
        ofr_to_bytes(tr->root_from_low, &r0);
        ofr_to_bytes(tr->new_low_leaf_hash, &nlh);
        ofr_to_bytes(tr->interim_root, &interim);
        ofr_to_bytes(tr->zero_slot_root, &zroot);
        ofr_to_bytes(tr->new_leaf_hash, &nh);
        ofr_to_bytes(tr->new_root, &nr);
    }
    return fail;
}

/* update_idx_leaf :632-660.  preimages[n][3][32] = {val,next_val,next_idx}. */
static int lt_bytes(const uint8_t *a, const uint8_t *b) {
    for (int i = 31; i >= 0; i--) {
        if (a[i] < b[i]) return 1;
        if (a[i] > b[i]) return 0;
    }
    return 0;
}
static int zero_bytes(const uint8_t *a) {
    for (int i = 0; i < 32; i++) if (a[i]) return 0;
    return 1;
}
static void put_u64(uint8_t *o, uint64_t v) {
    memset(o, 0, 32);
    for (int k = 0; k < 8; k++) o[k] = (uint8_t)(v >> (8 * k));
}
int orc_update_idx_leaf(uint8_t *pre, size_t n, const uint8_t new_val[32], uint64_t new_val_idx,
                        uint64_t *low) {
    *low = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t *node = pre + 96 * i;
        if (zero_bytes(node + 32) && i == 0) {                       /* :640-646 */
            if (n < 2) return ORC_ERR_RANGE;
            memcpy(pre + 96 * (i + 1), new_val, 32);
            memcpy(node + 32, new_val, 32);
            put_u64(node + 64, (uint64_t)i + 1);
            *low = i;
            break;
        }
        if (lt_bytes(node, new_val) && (lt_bytes(new_val, node + 32) || zero_bytes(node + 32))) { /* :647 */
            if (new_val_idx >= n) return ORC_ERR_RANGE;
            uint8_t *nw = pre + 96 * new_val_idx;
            memcpy(nw, new_val, 32);
            memmove(nw + 32, node + 32, 32);
            memmove(nw + 64, node + 64, 32);
            memcpy(node + 32, new_val, 32);
            put_u64(node + 64, new_val_idx);
            *low = i;
            break;
        }
    }
    return ORC_OK;
}

/* hash_nullifier_pre_images :662-671 */
int orc_hash_preimages(uint8_t *leaves_out, const uint8_t *pre, size_t n) {
    return orc_hash3_batch(leaves_out, pre, n);
}
