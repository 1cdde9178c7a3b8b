/*
 * column.h -- CPU ORACLE (test infrastructure): the advice column of a halo2-base Context as the gadgets fill it, cell
 * by cell -- shared by trace.c (hash_fix_len_array) and gadget.c (is_less_than, dual_mux, select, ...).  A cell is a
 * constant, a copy of an input / of an earlier witness, or a NEW witness (numbered in order: the "trace rows");
 * gate = 1 where a vertical gate a + b*c = d over four consecutive cells starts.
 */
#ifndef ORC_COLUMN_H
#define ORC_COLUMN_H
#include "imt_oracle.h"

typedef struct { ofr_t v; uint8_t kind; uint32_t index; } aval;     /* an AssignedValue and where it came from */

typedef struct {
    uint8_t *cells;            /* [cap][32] canonical, may be NULL */
    orc_trace_cell *desc;      /* [cap], may be NULL */
    size_t cap, n;
    uint8_t *wit;              /* [wcap][32], may be NULL */
    size_t wcap, nw;
    int overflow;
    int region_open;           /* the next cell starts a new ctx.assign_region call */
    uint32_t *lookup;          /* [lookup_cap] witness rows the RangeChip adds to its lookup (range_check limbs), may be NULL */
    size_t lookup_cap, n_lookup;
} col_t;

static void put(col_t *c, const ofr_t *v, uint8_t kind, uint32_t index, int gate) {
    if (c->n >= c->cap && (c->cells || c->desc)) { c->overflow = 1; c->region_open = 0; c->n++; return; }
    if (c->cells) ofr_to_bytes(c->cells + 32 * c->n, v);
    if (c->desc) { c->desc[c->n].kind = kind; c->desc[c->n].gate = (uint8_t)gate; c->desc[c->n].region = (uint16_t)c->region_open; c->desc[c->n].index = index; }
    c->region_open = 0;
    c->n++;
}
static void put_const(col_t *c, const ofr_t *v, int gate) { put(c, v, ORC_CELL_CONST, 0, gate); }
static void put_existing(col_t *c, const aval *a, int gate) {
    put(c, &a->v, a->kind == ORC_CELL_WITNESS ? ORC_CELL_COPY : a->kind, a->index, gate);
}
static aval put_witness(col_t *c, const ofr_t *v) {
    aval r;
    r.v = *v;
    r.kind = ORC_CELL_WITNESS;
    r.index = (uint32_t)c->nw;
    if (c->wit) { if (c->nw < c->wcap) ofr_to_bytes(c->wit + 32 * c->nw, v); else c->overflow = 1; }
    put(c, v, ORC_CELL_WITNESS, r.index, 0);
    c->nw++;
    return r;
}

static void mark_gate_last(col_t *c) {
    if (c->desc && c->n >= 1 && c->n <= c->cap) c->desc[c->n - 1].gate = 1;
}


#endif
