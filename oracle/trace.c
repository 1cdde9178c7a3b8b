/*
 * trace.c -- CPU ORACLE (test infrastructure): the witness trace of halo2-base's
 * PoseidonHasher::hash_fix_len_array, cell by cell, for T = 3, RATE = 2, R_F = 8, R_P = 57.
 *
 * Reference call sites: src/indexed_merkle_tree.rs:92 (path loop), :194, :271-275, :299-303 (leaf
 * hashes); hasher set-up :440-442 (OptimizedPoseidonSpec::new::<8, 57, 0>(), initialize_consts).
 * The gadget itself lives in the un-vendored crate halo2-base (aerius-labs/halo2-lib, branch
 * feat/secp256k1-hash2curve, Cargo.toml:14, no pinned commit).  What is restated here is the PUBLISHED
 * design of that crate line (halo2-lib v0.4.x, halo2-base/src/poseidon/hasher/{mod,state,spec,mds}.rs and
 * gates/flex_gate: the "vertical" gate q * (a + b * c - d) = 0 over four consecutive advice cells):
 *
 *   spec    OptimizedPoseidonSpec::new = pse-poseidon's Spec::new: constants {start[R_F/2 + 1], partial[R_P],
 *           end[R_F/2 - 1]} from the Grain constants c_r and M^-1 (calculate_optimized_constants), and the MDS
 *           factored into pre_sparse_mds and R_P sparse matrices {row, col_hat} (calculate_sparse_matrices:
 *           transpose, factorise, accumulate, reverse).
 *   hash    state = [2^64, 0, 0] (existing constant cells); one permutation per RATE-chunk of the inputs, one
 *           more with no inputs if the length is a multiple of RATE; result = state[1].
 *   perm    absorb_with_pre_constants(inputs, start[0]); for start[1..R_F/2): sbox_full(c), apply_mds(mds);
 *           sbox_full(start.last), apply_mds(pre_sparse); R_P x { sbox_part(c), apply_sparse_mds };
 *           for end: sbox_full(c), apply_mds(mds); sbox_full(0), apply_mds(mds).
 *   cells   gate.add(a,b) = [a, b, 1, a+b]; gate.mul(a,b) = [0, a, b, ab]; gate.mul_add(a,b,c) = [c, a, b, ab+c];
 *           gate.sum([a,b,c]) = [a, b, 1, a+b, c, 1, a+b+c]; gate.inner_product(a[3], const b[3]) =
 *           [0, a0, b0, a0b0, a1, b1, a0b0+a1b1, a2, b2, total]; x^5 + c = mul(x,x), mul(x2,x2), mul_add(x,x4,c).
 *
 * PARITY STATUS: the intermediate values are UNPINNED BY THE REFERENCE (it holds no trace vector and cannot be
 * built here).  What pins this file: (1) its output cell equals the plain 65-round sponge of poseidon.c, which
 * the reference's zero-leaf KAT (src/indexed_merkle_tree.rs:247-250) pins; (2) every gate of the emitted column
 * holds (selftest).  The optimised constants are derived HERE from the plain Grain constants -- nothing is
 * shared with the product's generator (csrc/imt_params.cpp).
 */
#include "imt_oracle.h"
#include <stdlib.h>
#include <string.h>

#define T 3
#define RF 8
#define RP 57
#define RFH (RF / 2)

typedef struct { ofr_t m[T][T]; } mat3;

static ofr_t START[RFH + 1][T], PARTIAL[RP], END[RFH - 1][T];
static mat3 MDSM, PRE_SPARSE;
static ofr_t SP_ROW[RP][T], SP_COLHAT[RP][T - 1];
static ofr_t CAP, ZERO, ONE;
static int g_tinit;

static void mat_identity(mat3 *a) {
    for (int i = 0; i < T; i++)
        for (int j = 0; j < T; j++) a->m[i][j] = (i == j) ? ONE : ZERO;
}
static void mat_mul(mat3 *o, const mat3 *a, const mat3 *b) {
    mat3 r;
    for (int i = 0; i < T; i++)
        for (int j = 0; j < T; j++) {
            ofr_t acc = ZERO, t;
            for (int k = 0; k < T; k++) { ofr_mul(&t, &a->m[i][k], &b->m[k][j]); ofr_add(&acc, &acc, &t); }
            r.m[i][j] = acc;
        }
    *o = r;
}
static void mat_transpose(mat3 *o, const mat3 *a) {
    mat3 r;
    for (int i = 0; i < T; i++)
        for (int j = 0; j < T; j++) r.m[i][j] = a->m[j][i];
    *o = r;
}
static void mat_vec(ofr_t o[T], const mat3 *a, const ofr_t v[T]) {
    ofr_t r[T];
    for (int i = 0; i < T; i++) {
        r[i] = ZERO;
        for (int k = 0; k < T; k++) { ofr_t t; ofr_mul(&t, &a->m[i][k], &v[k]); ofr_add(&r[i], &r[i], &t); }
    }
    memcpy(o, r, sizeof r);
}
/* Gauss-Jordan inverse of an n x n matrix (n <= 3), row-major in a[n][n] */
static void inv_n(ofr_t *a, int n) {
    ofr_t w[3][6];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) { w[i][j] = a[i * n + j]; w[i][n + j] = (i == j) ? ONE : ZERO; }
    for (int c = 0; c < n; c++) {
        int piv = c;
        while (piv < n && ofr_is_zero(&w[piv][c])) piv++;
        if (piv == n) abort();                      /* an MDS matrix and its minors are invertible */
        if (piv != c)
            for (int j = 0; j < 2 * n; j++) { ofr_t t = w[c][j]; w[c][j] = w[piv][j]; w[piv][j] = t; }
        ofr_t iv;
        ofr_inv(&iv, &w[c][c]);
        for (int j = 0; j < 2 * n; j++) ofr_mul(&w[c][j], &w[c][j], &iv);
        for (int i = 0; i < n; i++) {
            if (i == c) continue;
            ofr_t f = w[i][c];
            for (int j = 0; j < 2 * n; j++) { ofr_t t; ofr_mul(&t, &f, &w[c][j]); ofr_sub(&w[i][j], &w[i][j], &t); }
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) a[i * n + j] = w[i][n + j];
}

/* MDSMatrix::factorise: self = prime * prime_prime; returns prime and the sparse form of prime_prime^T */
static void factorise(const mat3 *self, mat3 *prime, ofr_t row[T], ofr_t col_hat[T - 1]) {
    ofr_t w[T - 1], mh[(T - 1) * (T - 1)], w_hat[T - 1];
    for (int i = 1; i < T; i++) w[i - 1] = self->m[i][0];
    for (int i = 1; i < T; i++)
        for (int j = 1; j < T; j++) mh[(i - 1) * (T - 1) + (j - 1)] = self->m[i][j];
    mat_identity(prime);
    for (int i = 1; i < T; i++)
        for (int j = 1; j < T; j++) prime->m[i][j] = mh[(i - 1) * (T - 1) + (j - 1)];
    inv_n(mh, T - 1);
    for (int i = 0; i < T - 1; i++) {               /* w_hat = m_hat^-1 * w */
        w_hat[i] = ZERO;
        for (int k = 0; k < T - 1; k++) { ofr_t t; ofr_mul(&t, &mh[i * (T - 1) + k], &w[k]); ofr_add(&w_hat[i], &w_hat[i], &t); }
    }
    /* prime_prime = identity with row 0 = self row 0 and column 0 (below the top) = w_hat; its TRANSPOSE as a
     * SparseMDSMatrix: row = first row of the transpose, col_hat = first column of the transpose below the top */
    row[0] = self->m[0][0];
    for (int i = 1; i < T; i++) row[i] = w_hat[i - 1];
    for (int i = 1; i < T; i++) col_hat[i - 1] = self->m[0][i];
}

static void trace_init(void) {
    if (g_tinit) return;
    orc_poseidon_init();
    static uint8_t rcb[ORC_ROUNDS * 3][32], mdsb[9][32];
    ofr_t c[ORC_ROUNDS][T];
    orc_poseidon_params(rcb, mdsb);
    for (int r = 0; r < ORC_ROUNDS; r++)
        for (int i = 0; i < T; i++) ofr_from_bytes(&c[r][i], rcb[r * 3 + i]);
    for (int i = 0; i < T; i++)
        for (int j = 0; j < T; j++) ofr_from_bytes(&MDSM.m[i][j], mdsb[i * 3 + j]);
    ofr_from_u64(&ZERO, 0);
    ofr_from_u64(&ONE, 1);
    ofr_t two32;
    ofr_from_u64(&two32, (uint64_t)1 << 32);
    ofr_mul(&CAP, &two32, &two32);

    /* ---- calculate_optimized_constants ---- */
    mat3 minv = MDSM;
    inv_n(&minv.m[0][0], T);
    memcpy(START[0], c[0], sizeof c[0]);
    for (int r = 1; r < RFH; r++) mat_vec(START[r], &minv, c[r]);
    ofr_t acc[T];
    memcpy(acc, c[RFH + RP], sizeof acc);
    for (int p = RP - 1; p >= 0; p--) {             /* partial[p] <-> plain round RFH + p, last first */
        ofr_t tmp[T];
        mat_vec(tmp, &minv, acc);
        PARTIAL[p] = tmp[0];
        tmp[0] = ZERO;
        for (int i = 0; i < T; i++) ofr_add(&acc[i], &tmp[i], &c[RFH + p][i]);
    }
    mat_vec(START[RFH], &minv, acc);
    for (int r = 0; r < RFH - 1; r++) mat_vec(END[r], &minv, c[RFH + RP + 1 + r]);

    /* ---- calculate_sparse_matrices ---- */
    mat3 mt, accm, prime;
    mat_transpose(&mt, &MDSM);
    accm = mt;
    for (int k = 0; k < RP; k++) {                  /* generated last round first, then reversed */
        factorise(&accm, &prime, SP_ROW[RP - 1 - k], SP_COLHAT[RP - 1 - k]);
        mat_mul(&accm, &mt, &prime);
    }
    mat_transpose(&PRE_SPARSE, &accm);
    g_tinit = 1;
}

#include "column.h"

/* gate.add(a, Constant(k)) */
static aval g_add_const(col_t *c, const aval *a, const ofr_t *k) {
    ofr_t o;
    ofr_add(&o, &a->v, k);
    c->region_open = 1;
    put_existing(c, a, 1); put_const(c, k, 0); put_const(c, &ONE, 0);
    return put_witness(c, &o);
}
/* gate.sum([Existing(x), Existing(in), Constant(k)]) */
static aval g_sum3(col_t *c, const aval *x, const aval *in, const ofr_t *k) {
    ofr_t s1, s2;
    ofr_add(&s1, &x->v, &in->v);
    ofr_add(&s2, &s1, k);
    c->region_open = 1;
    put_existing(c, x, 1); put_existing(c, in, 0); put_const(c, &ONE, 0);
    put_witness(c, &s1);
    mark_gate_last(c);                               /* the running sum opens the second gate */
    put_const(c, k, 0); put_const(c, &ONE, 0);
    return put_witness(c, &s2);
}
static aval g_mul(col_t *c, const aval *a, const aval *b) {
    ofr_t o;
    ofr_mul(&o, &a->v, &b->v);
    c->region_open = 1;
    put_const(c, &ZERO, 1); put_existing(c, a, 0); put_existing(c, b, 0);
    return put_witness(c, &o);
}
/* gate.mul_add(a, b, Constant(k)) = a*b + k */
static aval g_mul_add_const(col_t *c, const aval *a, const aval *b, const ofr_t *k) {
    ofr_t o;
    ofr_mul(&o, &a->v, &b->v);
    ofr_add(&o, &o, k);
    c->region_open = 1;
    put_const(c, k, 1); put_existing(c, a, 0); put_existing(c, b, 0);
    return put_witness(c, &o);
}
/* gate.mul_add(a, Constant(k), cc) = a*k + cc */
static aval g_mul_const_add(col_t *c, const aval *a, const ofr_t *k, const aval *cc) {
    ofr_t o;
    ofr_mul(&o, &a->v, k);
    ofr_add(&o, &o, &cc->v);
    c->region_open = 1;
    put_existing(c, cc, 1); put_existing(c, a, 0); put_const(c, k, 0);
    return put_witness(c, &o);
}
/* gate.inner_product(s[0..3], Constant(row[0..3])) */
static aval g_inner(col_t *c, const aval *s, const ofr_t *row) {
    ofr_t sum = ZERO, t;
    aval last;
    c->region_open = 1;
    put_const(c, &ZERO, 1);
    for (int i = 0; i < T; i++) {
        ofr_mul(&t, &s[i].v, &row[i]);
        ofr_add(&sum, &sum, &t);
        put_existing(c, &s[i], 0); put_const(c, &row[i], 0);
        last = put_witness(c, &sum);
        if (i + 1 < T) mark_gate_last(c);            /* the running sum opens the next gate */
    }
    return last;
}
static aval x5_plus(col_t *c, const aval *x, const ofr_t *k) {
    aval x2 = g_mul(c, x, x);
    aval x4 = g_mul(c, &x2, &x2);
    return g_mul_add_const(c, x, &x4, k);
}
static void apply_mds(col_t *c, aval s[T], const mat3 *m) {
    aval r[T];
    for (int i = 0; i < T; i++) r[i] = g_inner(c, s, m->m[i]);
    memcpy(s, r, sizeof r);
}

static void permutation(col_t *c, aval s[T], const aval *inputs, int ni) {
    /* absorb_with_pre_constants */
    s[0] = g_add_const(c, &s[0], &START[0][0]);
    for (int i = 0; i < ni; i++) s[1 + i] = g_sum3(c, &s[1 + i], &inputs[i], &START[0][1 + i]);
    for (int i = 0, j = ni + 1; j < T; i++, j++) {
        ofr_t k = START[0][j];
        if (i == 0) ofr_add(&k, &k, &ONE);           /* the padding 1 rides on the first free lane */
        s[j] = g_add_const(c, &s[j], &k);
    }
    for (int r = 1; r < RFH; r++) {
        for (int i = 0; i < T; i++) s[i] = x5_plus(c, &s[i], &START[r][i]);
        apply_mds(c, s, &MDSM);
    }
    for (int i = 0; i < T; i++) s[i] = x5_plus(c, &s[i], &START[RFH][i]);
    apply_mds(c, s, &PRE_SPARSE);
    for (int p = 0; p < RP; p++) {
        s[0] = x5_plus(c, &s[0], &PARTIAL[p]);
        aval r[T];
        r[0] = g_inner(c, s, SP_ROW[p]);
        for (int i = 1; i < T; i++) r[i] = g_mul_const_add(c, &s[0], &SP_COLHAT[p][i - 1], &s[i]);
        memcpy(s, r, sizeof r);
    }
    for (int r = 0; r < RFH - 1; r++) {
        for (int i = 0; i < T; i++) s[i] = x5_plus(c, &s[i], &END[r][i]);
        apply_mds(c, s, &MDSM);
    }
    for (int i = 0; i < T; i++) s[i] = x5_plus(c, &s[i], &ZERO);
    apply_mds(c, s, &MDSM);
}

int orc_hash_trace(const uint8_t *in, int arity, uint8_t *cells, orc_trace_cell *desc, size_t cap, size_t *n_cells,
                   uint8_t *witness, size_t wcap, size_t *n_witness, uint32_t *out_row) {
    if (arity < 1 || arity > 16) return ORC_ERR_RANGE;
    trace_init();
    col_t c;
    memset(&c, 0, sizeof c);
    c.cells = cells; c.desc = desc; c.cap = cap; c.wit = witness; c.wcap = wcap;
    aval inputs[16], s[T];
    for (int i = 0; i < arity; i++) {
        if (ofr_from_bytes(&inputs[i].v, in + 32 * i)) return ORC_ERR_NONCANONICAL;
        inputs[i].kind = ORC_CELL_INPUT;
        inputs[i].index = (uint32_t)i;
    }
    s[0].v = CAP; s[1].v = ZERO; s[2].v = ZERO;      /* PoseidonState::default: existing constant cells */
    for (int i = 0; i < T; i++) { s[i].kind = ORC_CELL_INIT; s[i].index = (uint32_t)i; }
    int i = 0;
    for (; i + 2 <= arity; i += 2) permutation(&c, s, &inputs[i], 2);
    if (i < arity) permutation(&c, s, &inputs[i], 1);
    else permutation(&c, s, NULL, 0);
    if (n_cells) *n_cells = c.n;
    if (n_witness) *n_witness = c.nw;
    if (out_row) *out_row = s[1].index;
    return c.overflow ? ORC_ERR_RANGE : ORC_OK;
}

void orc_trace_spec(uint8_t *start /*[5][3][32]*/, uint8_t *partial /*[57][32]*/, uint8_t *end /*[3][3][32]*/,
                    uint8_t *pre_sparse /*[9][32]*/, uint8_t *sp_row /*[57][3][32]*/, uint8_t *sp_col_hat /*[57][2][32]*/) {
    trace_init();
    for (int r = 0; r <= RFH; r++)
        for (int i = 0; i < T; i++) ofr_to_bytes(start + 32 * (r * T + i), &START[r][i]);
    for (int p = 0; p < RP; p++) ofr_to_bytes(partial + 32 * p, &PARTIAL[p]);
    for (int r = 0; r < RFH - 1; r++)
        for (int i = 0; i < T; i++) ofr_to_bytes(end + 32 * (r * T + i), &END[r][i]);
    for (int i = 0; i < T; i++)
        for (int j = 0; j < T; j++) ofr_to_bytes(pre_sparse + 32 * (i * T + j), &PRE_SPARSE.m[i][j]);
    for (int p = 0; p < RP; p++) {
        for (int i = 0; i < T; i++) ofr_to_bytes(sp_row + 32 * (p * T + i), &SP_ROW[p][i]);
        for (int i = 0; i < T - 1; i++) ofr_to_bytes(sp_col_hat + 32 * (p * (T - 1) + i), &SP_COLHAT[p][i]);
    }
}
