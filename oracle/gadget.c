/*
 * gadget.c -- CPU ORACLE (test infrastructure): every advice cell the reference's insert_leaf assigns OUTSIDE
 * hash_fix_len_array (those are trace.c), in assignment order.
 *
 * Reference: /root/reference/src/indexed_merkle_tree.rs
 *   is_less_than :98-125   range.is_less_than(a_q, b_q, 128), gate.is_equal(a_q, b_q), the same for the low limbs,
 *                          then not / not / not / not, a 3-fold and, an and, an or
 *   select :33-45, dual_mux :47-63, compute_merkle_root :78-96 (load_witness of the leaf, dual_mux per level)
 *   verify_non_inclusion :127-229, insert_leaf :231-314
 * The cells those calls expand to live in the un-vendored crate halo2-base (aerius-labs/halo2-lib, branch
 * feat/secp256k1-hash2curve, Cargo.toml:14, no pinned commit).  What is restated here is the PUBLISHED design of that
 * crate line (halo2-lib v0.4.x, halo2-base/src/gates/{flex_gate,range}.rs; vertical gate q * (a + b * c - d) = 0):
 *
 *   sub(a, b)            [W a-b, b, 1, a]                      not(a) = sub(1, a) = [W 1-a, a, 1, 1]
 *   mul(a, b) = and      [0, a, b, W ab]                       mul_add(a, b, c) = [c, a, b, W ab+c]
 *   or(a, b)             [W 1-b, 1, b, 1, b, a, W 1-b, W a+b-ab]                         gates at 0 and 4
 *   is_zero(a)           [W z, a, W inv, 1, 0, a, W z, 0], inv = 1/a (1 if a = 0)        gates at 0 and 4
 *   is_equal(a, b)       sub(a, b); is_zero(diff)
 *   assert_bit(x)        [0, x, x, x]                          load_witness(v) = [W v]       load_constant(c) = [c]
 *   range.is_less_than(a, b, num_bits) with k = ceil(num_bits / lookup_bits), padded = k * lookup_bits:
 *                        [W 2^padded+a-b, b, 1, W 2^padded+a, -2^padded, 1, a]            gates at 0 and 3
 *                        range check of the first cell on padded + lookup_bits bits = k + 1 limbs of lookup_bits bits:
 *                        inner_product(limbs, 2^(lookup_bits i)) = [W l0, W l1, 2^lb, W s1, W l2, 2^2lb, W s2, ...]
 *                        (the limb cells also go to the lookup columns, which are not part of this column);
 *                        is_zero(last limb)
 *
 * PARITY STATUS: UNPINNED BY THE REFERENCE, like trace.c (halo2-base cannot be built here; the reference holds no cell
 * vector).  What pins this file: every vertical gate of the emitted column holds (selftest / tests), the outputs are
 * the reference's boolean formula (indexed.c, orc_is_less_than_limbs), and a big-integer model in the tests restates the
 * rows independently.
 */
#include "column.h"
#include <stdlib.h>
#include <string.h>

static ofr_t G_ZERO, G_ONE;
static int g_ginit;
static void ginit(void) {
    if (g_ginit) return;
    ofr_init();
    ofr_from_u64(&G_ZERO, 0);
    ofr_from_u64(&G_ONE, 1);
    g_ginit = 1;
}

/* 2^e as a field element, e < 254 */
static void pow2(ofr_t *o, unsigned e) {
    uint8_t b[32] = {0};
    b[e / 8] = (uint8_t)(1u << (e % 8));
    ofr_from_bytes(o, b);
}

/* ---- GateChip ---- */
static aval g_sub(col_t *c, const aval *a, const aval *b) {              /* [W a-b, b, 1, a] */
    ofr_t o;
    ofr_sub(&o, &a->v, &b->v);
    c->region_open = 1;
    aval r = put_witness(c, &o);
    mark_gate_last(c);
    put_existing(c, b, 0); put_const(c, &G_ONE, 0); put_existing(c, a, 0);
    return r;
}
static aval g_not(col_t *c, const aval *a) {                             /* sub(Constant(1), a) = [W 1-a, a, 1, 1] */
    ofr_t o;
    ofr_sub(&o, &G_ONE, &a->v);
    c->region_open = 1;
    aval r = put_witness(c, &o);
    mark_gate_last(c);
    put_existing(c, a, 0); put_const(c, &G_ONE, 0); put_const(c, &G_ONE, 0);
    return r;
}
static aval g_mul(col_t *c, const aval *a, const aval *b) {              /* [0, a, b, W ab] */
    ofr_t o;
    ofr_mul(&o, &a->v, &b->v);
    c->region_open = 1;
    put_const(c, &G_ZERO, 1); put_existing(c, a, 0); put_existing(c, b, 0);
    return put_witness(c, &o);
}
static aval g_mul_add(col_t *c, const aval *a, const aval *b, const aval *cc) {   /* [c, a, b, W ab+c] */
    ofr_t o;
    ofr_mul(&o, &a->v, &b->v);
    ofr_add(&o, &o, &cc->v);
    c->region_open = 1;
    put_existing(c, cc, 1); put_existing(c, a, 0); put_existing(c, b, 0);
    return put_witness(c, &o);
}
static aval g_or(col_t *c, const aval *a, const aval *b) {
    ofr_t nb, ab, o;
    ofr_sub(&nb, &G_ONE, &b->v);
    ofr_mul(&ab, &a->v, &b->v);
    ofr_add(&o, &a->v, &b->v);
    ofr_sub(&o, &o, &ab);
    c->region_open = 1;
    put_witness(c, &nb);
    mark_gate_last(c);
    put_const(c, &G_ONE, 0); put_existing(c, b, 0); put_const(c, &G_ONE, 0);
    put_existing(c, b, 1); put_existing(c, a, 0);
    put_witness(c, &nb);                             /* a NEW cell, tied to the first by an equality constraint */
    return put_witness(c, &o);
}
static aval g_is_zero(col_t *c, const aval *a) {
    ofr_t z, inv;
    if (ofr_is_zero(&a->v)) { z = G_ONE; inv = G_ONE; }
    else { z = G_ZERO; ofr_inv(&inv, &a->v); }
    c->region_open = 1;
    put_witness(c, &z);
    mark_gate_last(c);
    put_existing(c, a, 0);
    put_witness(c, &inv);                            /* WitnessFraction(1 / a), resolved to this value */
    put_const(c, &G_ONE, 0);
    put_const(c, &G_ZERO, 1); put_existing(c, a, 0);
    aval r = put_witness(c, &z);                     /* the cell the gadget returns (ctx.get(-2)) */
    put_const(c, &G_ZERO, 0);
    return r;
}
static aval g_is_equal(col_t *c, const aval *a, const aval *b) {
    aval d = g_sub(c, a, b);
    return g_is_zero(c, &d);
}
static void g_assert_bit(col_t *c, const aval *x) {                      /* [0, x, x, x] */
    c->region_open = 1;
    put_const(c, &G_ZERO, 1); put_existing(c, x, 0); put_existing(c, x, 0); put_existing(c, x, 0);
}
static aval g_load_witness(col_t *c, const ofr_t *v) {
    c->region_open = 1;
    return put_witness(c, v);
}
static aval g_load_constant(col_t *c, const ofr_t *v) {
    c->region_open = 1;
    put_const(c, v, 0);
    aval r;
    r.v = *v; r.kind = ORC_CELL_CONST; r.index = 0;
    return r;
}

/* RangeChip::add_cell_to_lookup on the witness just put */
static void mark_lookup(col_t *c, const aval *a) {
    if (c->lookup && c->n_lookup < c->lookup_cap) c->lookup[c->n_lookup] = a->index;
    c->n_lookup++;
}

/* ---- RangeChip::is_less_than(a, b, 128) with the given lookup_bits; a, b < 2^128 as integers ---- */
static aval r_is_less_than(col_t *c, const aval *a, const aval *b, unsigned lb) {
    const unsigned k = (128 + lb - 1) / lb, padded = k * lb, L = k + 1;
    ofr_t pw, npw, sa, sab;
    pow2(&pw, padded);
    ofr_sub(&npw, &G_ZERO, &pw);
    ofr_add(&sa, &pw, &a->v);
    ofr_sub(&sab, &sa, &b->v);
    c->region_open = 1;
    aval shifted = put_witness(c, &sab);
    mark_gate_last(c);
    put_existing(c, b, 0); put_const(c, &G_ONE, 0);
    put_witness(c, &sa);
    mark_gate_last(c);
    put_const(c, &npw, 0); put_const(c, &G_ONE, 0); put_existing(c, a, 0);
    /* range_check(shifted, padded + lb): limbs of lb bits, little-endian, as an inner product with 2^(lb i) */
    uint8_t bytes[32];
    ofr_to_bytes(bytes, &shifted.v);
    ofr_t sum = G_ZERO;
    aval last = shifted;
    c->region_open = 1;
    for (unsigned i = 0; i < L; i++) {
        uint64_t limb = 0;
        for (unsigned bit = 0; bit < lb; bit++) {
            const unsigned pos = i * lb + bit;
            limb |= (uint64_t)((bytes[pos / 8] >> (pos % 8)) & 1) << bit;
        }
        ofr_t lf, base, t;
        ofr_from_u64(&lf, limb);
        if (i == 0) {
            sum = lf;
            last = put_witness(c, &lf);              /* limb_bases[0] = 1: the first limb opens the running sum */
            if (L > 1) mark_gate_last(c);
            mark_lookup(c, &last);
        } else {
            pow2(&base, i * lb);
            ofr_mul(&t, &lf, &base);
            ofr_add(&sum, &sum, &t);
            last = put_witness(c, &lf);
            mark_lookup(c, &last);
            put_const(c, &base, 0);
            put_witness(c, &sum);
            if (i + 1 < L) mark_gate_last(c);
        }
    }
    if (!ofr_eq(&sum, &shifted.v)) abort();          /* ctx.constrain_equal(&a, &acc): a, b were not < 2^128 */
    return g_is_zero(c, &last);
}

/* is_less_than :98-125 */
static aval ref_is_less_than(col_t *c, const aval *a_q, const aval *a_r, const aval *b_q, const aval *b_r, unsigned lb) {
    aval is_ll_msb_gr = r_is_less_than(c, a_q, b_q, lb);
    aval are_msb_eq = g_is_equal(c, a_q, b_q);
    aval is_ll_lsb_gr = r_is_less_than(c, a_r, b_r, lb);
    aval are_lsb_eq = g_is_equal(c, a_r, b_r);
    aval a = is_ll_msb_gr;
    aval c_not = g_not(c, &are_msb_eq);
    aval a_not = g_not(c, &a);
    aval b = is_ll_lsb_gr;
    aval cc = g_not(c, &c_not);
    aval d_not = g_not(c, &are_lsb_eq);
    aval rhs = g_mul(c, &a_not, &b);                 /* [b, c, d_not].fold(a_not, and) */
    rhs = g_mul(c, &rhs, &cc);
    rhs = g_mul(c, &rhs, &d_not);
    aval lhs = g_mul(c, &a, &c_not);
    return g_or(c, &lhs, &rhs);
}

static void split128(const uint8_t v[32], ofr_t *q, ofr_t *r) {
    uint8_t lo[32] = {0}, hi[32] = {0};
    memcpy(lo, v, 16);
    memcpy(hi, v + 16, 16);
    ofr_from_bytes(r, lo);
    ofr_from_bytes(q, hi);
}

size_t orc_less_than_trace_rows(unsigned lookup_bits) {
    const unsigned L = (128 + lookup_bits - 1) / lookup_bits + 1;
    return 4 * (size_t)L + 27;
}

/* The column of ONE is_less_than(a_q, a_r, b_q, b_r) for 256-bit a, b (inputs 0..3 = a_q, a_r, b_q, b_r). */
int orc_less_than_trace(const uint8_t a[32], const uint8_t b[32], unsigned lookup_bits, uint8_t *cells, orc_trace_cell *desc,
                        size_t cap, size_t *n_cells, uint8_t *witness, size_t wcap, size_t *n_witness, uint32_t *out_row) {
    ginit();
    if (lookup_bits < 1 || lookup_bits > 28) return ORC_ERR_RANGE;
    ofr_t chk;
    if (ofr_from_bytes(&chk, a) || ofr_from_bytes(&chk, b)) return ORC_ERR_NONCANONICAL;
    col_t c;
    memset(&c, 0, sizeof c);
    c.cells = cells; c.desc = desc; c.cap = cap; c.wit = witness; c.wcap = wcap;
    aval in[4];
    split128(a, &in[0].v, &in[1].v);
    split128(b, &in[2].v, &in[3].v);
    for (int i = 0; i < 4; i++) { in[i].kind = ORC_CELL_INPUT; in[i].index = (uint32_t)i; }
    aval out = ref_is_less_than(&c, &in[0], &in[1], &in[2], &in[3], lookup_bits);
    if (n_cells) *n_cells = c.n;
    if (n_witness) *n_witness = c.nw;
    if (out_row) *out_row = out.index;
    return c.overflow ? ORC_ERR_RANGE : ORC_OK;
}

/* The witness rows of that column the RangeChip adds to its lookup table (range_check: the lookup_bits-wide limbs of
 * both shifted differences; padded + lookup_bits is a multiple of lookup_bits, so no scaled last limb): 2 L of them. */
int orc_less_than_lookup_rows(unsigned lookup_bits, uint32_t *rows, size_t cap, size_t *n_rows) {
    ginit();
    if (lookup_bits < 1 || lookup_bits > 28) return ORC_ERR_RANGE;
    col_t c;
    memset(&c, 0, sizeof c);
    c.lookup = rows; c.lookup_cap = rows ? cap : 0;
    aval in[4];
    for (int i = 0; i < 4; i++) { in[i].v = G_ZERO; in[i].kind = ORC_CELL_INPUT; in[i].index = (uint32_t)i; }
    (void)ref_is_less_than(&c, &in[0], &in[1], &in[2], &in[3], lookup_bits);
    if (n_rows) *n_rows = c.n_lookup;
    return rows && c.n_lookup > cap ? ORC_ERR_RANGE : ORC_OK;
}

/* ---- the non-hash part of insert_leaf :231-314 ---- */
typedef struct {
    col_t c;
    orc_column_segment *segs;
    size_t seg_cap, n_segs;
    uint64_t glue_mark;          /* witness rows already attributed to a glue segment */
    uint64_t hash_rows;          /* rows of the hash trace (imt_insert_trace_batch order) so far */
} walk_t;

static void close_glue(walk_t *w) {
    if (w->c.nw > w->glue_mark) {
        if (w->segs && w->n_segs < w->seg_cap) {
            w->segs[w->n_segs].kind = 0; w->segs[w->n_segs].arity = 0;
            w->segs[w->n_segs].first_row = w->glue_mark; w->segs[w->n_segs].n_rows = w->c.nw - w->glue_mark;
        }
        w->n_segs++;
        w->glue_mark = w->c.nw;
    }
}
/* a hash_fix_len_array call: its cells are NOT in this column (trace.c); its output enters as an external value */
static aval hash_block(walk_t *w, const aval *in, int arity) {
    close_glue(w);
    const uint64_t rows = arity == 2 ? 1208 : 1209;
    if (w->segs && w->n_segs < w->seg_cap) {
        w->segs[w->n_segs].kind = 1; w->segs[w->n_segs].arity = (uint32_t)arity;
        w->segs[w->n_segs].first_row = w->hash_rows; w->segs[w->n_segs].n_rows = rows;
    }
    w->n_segs++;
    w->hash_rows += rows;
    aval r;
    if (arity == 2) orc_hash2_fr(&r.v, &in[0].v, &in[1].v);
    else orc_hash3_fr(&r.v, &in[0].v, &in[1].v, &in[2].v);
    r.kind = ORC_CELL_INPUT;         /* from outside this column */
    r.index = 0xffff;
    return r;
}
static void ref_dual_mux(col_t *c, const aval *a, const aval *b, const aval *sw, aval out[2]) {     /* :47-63 */
    g_assert_bit(c, sw);
    aval a_sub_b = g_sub(c, a, b);
    aval b_sub_a = g_sub(c, b, a);
    out[0] = g_mul_add(c, &a_sub_b, sw, b);
    out[1] = g_mul_add(c, &b_sub_a, sw, a);
}
static aval ref_compute_merkle_root(walk_t *w, const aval *leaf, const aval *proof, const aval *helper, size_t depth) {   /* :78-96 */
    aval cur = g_load_witness(&w->c, &leaf->v);
    for (size_t l = 0; l < depth; l++) {
        aval inp[2];
        ref_dual_mux(&w->c, &cur, &proof[l], &helper[l], inp);
        cur = hash_block(w, inp, 2);
    }
    return cur;
}
static aval ref_select(col_t *c, const aval *one, const aval *s, const aval *a, const aval *b) {    /* :33-45 */
    g_assert_bit(c, s);
    aval a_s = g_mul(c, a, s);
    aval oms = g_sub(c, one, s);
    return g_mul_add(c, &oms, b, &a_s);
}
static aval ext(const ofr_t *v, uint32_t index) {
    aval r;
    r.v = *v; r.kind = ORC_CELL_INPUT; r.index = index;
    return r;
}

/* verify_non_inclusion :127-229 -- the cells outside its hashes: is_equal(next_val, 0), the limb loads and their
 * mul_add checks, is_less_than(new, low.next_val), select, the low leaf's path (load_witness + dual_mux per level), the
 * limbs of low.val, is_less_than(low.val, new) */
static void walk_non_inclusion(walk_t *w, const aval low[3], const aval *new_val, const aval *largest, const aval *lp,
                               const aval *lh, size_t depth, unsigned lookup_bits) {
    col_t *c = &w->c;
    aval one = g_load_constant(c, &G_ONE);
    aval zero = g_load_constant(c, &G_ZERO);
    aval is_zero = g_is_equal(c, &low[1], &zero);
    uint8_t nlb[32], llb[32], llvb[32];
    ofr_to_bytes(nlb, &new_val->v);
    ofr_to_bytes(llb, &low[1].v);
    ofr_to_bytes(llvb, &low[0].v);
    ofr_t q, r, p128;
    split128(nlb, &q, &r);
    aval nl_q = g_load_witness(c, &q), nl_r = g_load_witness(c, &r);
    split128(llb, &q, &r);
    aval ll_q = g_load_witness(c, &q), ll_r = g_load_witness(c, &r);
    pow2(&p128, 128);
    aval pow_128 = g_load_constant(c, &p128);
    g_mul_add(c, &nl_q, &pow_128, &nl_r);
    g_mul_add(c, &ll_q, &pow_128, &ll_r);
    aval is_next_val_greater = ref_is_less_than(c, &nl_q, &nl_r, &ll_q, &ll_r, lookup_bits);
    ref_select(c, &one, largest, &is_zero, &is_next_val_greater);
    aval low_leaf_hash = hash_block(w, low, 3);
    ref_compute_merkle_root(w, &low_leaf_hash, lp, lh, depth);
    split128(llvb, &q, &r);
    aval llv_q = g_load_witness(c, &q), llv_r = g_load_witness(c, &r);
    g_mul_add(c, &llv_q, &pow_128, &llv_r);
    ref_is_less_than(c, &llv_q, &llv_r, &nl_q, &nl_r, lookup_bits);
    g_load_constant(c, &G_ONE);
}

size_t orc_non_inclusion_gadget_rows(size_t depth, unsigned lookup_bits) {
    return 17 + 2 * orc_less_than_trace_rows(lookup_bits) + 4 * depth;
}

/* The same for ONE verify_non_inclusion call on its own (BASELINE config 3's gadget): witness rows outside its 1 + depth
 * hashes and how they interleave with the hash blocks (the rows of orc_hash_trace for H(low_leaf) and the path, in the
 * order imt_path_trace_batch lays them out). */
int orc_non_inclusion_gadget_trace(const uint8_t low_leaf[3][32], uint64_t low_index, const uint8_t *low_proof,
                                   const uint8_t new_val[32], int is_new_leaf_largest, size_t depth, unsigned lookup_bits,
                                   uint8_t *witness, size_t wcap, size_t *n_witness, orc_column_segment *segs,
                                   size_t seg_cap, size_t *n_segs) {
    ginit();
    orc_poseidon_init();
    if (lookup_bits < 1 || lookup_bits > 28 || depth == 0 || depth > 64) return ORC_ERR_RANGE;
    walk_t w;
    memset(&w, 0, sizeof w);
    w.c.wit = witness; w.c.wcap = wcap;
    w.segs = segs; w.seg_cap = seg_cap;
    ofr_t v;
    aval low[3], *lp = malloc(depth * sizeof(aval)), *lh = malloc(depth * sizeof(aval));
    int rc = ORC_OK;
    for (int i = 0; i < 3; i++) {
        if (ofr_from_bytes(&v, low_leaf[i])) rc = ORC_ERR_NONCANONICAL;
        low[i] = ext(&v, (uint32_t)i);
    }
    if (ofr_from_bytes(&v, new_val)) rc = ORC_ERR_NONCANONICAL;
    aval nv = ext(&v, 8);
    for (size_t l = 0; l < depth && !rc; l++) {
        if (ofr_from_bytes(&v, low_proof + 32 * l)) rc = ORC_ERR_NONCANONICAL;
        lp[l] = ext(&v, 100);
        ofr_from_u64(&v, ((low_index >> l) & 1) ^ 1);
        lh[l] = ext(&v, 102);
    }
    if (!rc) {
        ofr_from_u64(&v, is_new_leaf_largest ? 1 : 0);
        aval largest = ext(&v, 12);
        walk_non_inclusion(&w, low, &nv, &largest, lp, lh, depth, lookup_bits);
        close_glue(&w);
    }
    free(lp); free(lh);
    if (rc) return rc;
    if (n_witness) *n_witness = w.c.nw;
    if (n_segs) *n_segs = w.n_segs;
    if (w.c.overflow || (segs && w.n_segs > seg_cap)) return ORC_ERR_RANGE;
    return w.c.nw == orc_non_inclusion_gadget_rows(depth, lookup_bits) ? ORC_OK : ORC_ERR_RANGE;
}

size_t orc_insert_gadget_rows(size_t depth, unsigned lookup_bits) {
    return 20 + 2 * orc_less_than_trace_rows(lookup_bits) + 16 * depth;
}

/* Witness rows ("glue rows") of insert_leaf outside its hashes, and how they interleave with the hash blocks of
 * orc_hash_trace in the order imt_insert_trace_batch lays them out.  Helper bit l = 1 iff bit l of the index is 0. */
int orc_insert_gadget_trace(const uint8_t low_leaf[3][32], uint64_t low_index, const uint8_t *low_proof,
                            const uint8_t new_leaf[3][32], uint64_t new_index, uint64_t new_path_index,
                            const uint8_t *new_proof, int is_new_leaf_largest, size_t depth, unsigned lookup_bits,
                            uint8_t *witness, size_t wcap, size_t *n_witness, orc_column_segment *segs, size_t seg_cap,
                            size_t *n_segs) {
    ginit();
    orc_poseidon_init();
    if (lookup_bits < 1 || lookup_bits > 28 || depth == 0 || depth > 64) return ORC_ERR_RANGE;
    walk_t w;
    memset(&w, 0, sizeof w);
    w.c.wit = witness; w.c.wcap = wcap;
    w.segs = segs; w.seg_cap = seg_cap;
    col_t *c = &w.c;
    ofr_t v;
    aval low[3], nw[3], *lp = malloc(depth * sizeof(aval)), *lh = malloc(depth * sizeof(aval)),
                        *np = malloc(depth * sizeof(aval)), *nh = malloc(depth * sizeof(aval));
    int rc = ORC_OK;
    for (int i = 0; i < 3; i++) {
        if (ofr_from_bytes(&v, low_leaf[i])) rc = ORC_ERR_NONCANONICAL;
        low[i] = ext(&v, (uint32_t)i);
        if (ofr_from_bytes(&v, new_leaf[i])) rc = ORC_ERR_NONCANONICAL;
        nw[i] = ext(&v, 8 + (uint32_t)i);
    }
    for (size_t l = 0; l < depth && !rc; l++) {
        if (ofr_from_bytes(&v, low_proof + 32 * l)) rc = ORC_ERR_NONCANONICAL;
        lp[l] = ext(&v, 100);
        if (ofr_from_bytes(&v, new_proof + 32 * l)) rc = ORC_ERR_NONCANONICAL;
        np[l] = ext(&v, 101);
        ofr_from_u64(&v, ((low_index >> l) & 1) ^ 1);
        lh[l] = ext(&v, 102);
        ofr_from_u64(&v, ((new_path_index >> l) & 1) ^ 1);
        nh[l] = ext(&v, 103);
    }
    if (rc) { free(lp); free(lh); free(np); free(nh); return rc; }
    ofr_from_u64(&v, new_index);
    aval new_idx_fe = ext(&v, 11);
    ofr_from_u64(&v, is_new_leaf_largest ? 1 : 0);
    aval largest = ext(&v, 12);

    /* insert_leaf :246-251 */
    static const uint8_t ZERO3[3][32] = {{0}};
    uint8_t zh[32];
    orc_hash3(zh, ZERO3[0], ZERO3[1], ZERO3[2]);
    ofr_from_bytes(&v, zh);
    aval zero_leaf_hash = g_load_constant(c, &v);
    /* verify_non_inclusion(old_root, low_leaf, low proof, new_leaf.val, is_largest) :253-257 -> :127-229 */
    walk_non_inclusion(&w, low, &nw[0], &largest, lp, lh, depth, lookup_bits);
    /* :259-312 */
    aval newlow[3] = {low[0], nw[0], new_idx_fe};
    aval new_low_leaf_hash = hash_block(&w, newlow, 3);
    ref_compute_merkle_root(&w, &new_low_leaf_hash, lp, lh, depth);
    ref_compute_merkle_root(&w, &zero_leaf_hash, np, nh, depth);
    aval new_leaf_hash = hash_block(&w, nw, 3);
    ref_compute_merkle_root(&w, &new_leaf_hash, np, nh, depth);
    close_glue(&w);
    free(lp); free(lh); free(np); free(nh);
    if (n_witness) *n_witness = c->nw;
    if (n_segs) *n_segs = w.n_segs;
    if (c->overflow || (segs && w.n_segs > seg_cap)) return ORC_ERR_RANGE;
    return c->nw == orc_insert_gadget_rows(depth, lookup_bits) ? ORC_OK : ORC_ERR_RANGE;
}
