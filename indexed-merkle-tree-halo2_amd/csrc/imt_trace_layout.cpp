// imt_trace_layout.cpp -- f1: the advice column halo2-base's PoseidonHasher::hash_fix_len_array appends for
// one hash, described cell by cell (imt_hash_trace_layout), so that a chip can turn the GPU's trace rows
// (imt_hash_trace_batch) into ctx.assign_region calls without recomputing anything:
//   hasher.hash_fix_len_array(ctx, gate, &inp)   /root/reference/src/indexed_merkle_tree.rs:92,194,271-275,299-303
// The cell order is the published halo2-lib v0.4.x gadget (poseidon/hasher/state.rs) over GateChip's vertical gate
// q (a + b c - d) = 0 (gates/flex_gate.rs); see imt_trace_device.hpp for the row order and the parity status.
// Host code only: no values are computed here, only where each one comes from.
#include "imt_ctx.hpp"
#include "imt_device.hpp"
#include <cstring>
#include <vector>

using namespace imt;

namespace {

struct Ref { uint8_t kind; uint32_t index; };     // where an assigned value lives

struct Walker {
    const HostPoseidon& hp;
    std::vector<imt_trace_cell> cells;
    std::vector<HFr> consts;
    uint32_t n_witness = 0;
    explicit Walker(const HostPoseidon& h) : hp(h) {}

    uint32_t cst(const HFr& v) {
        for (size_t i = 0; i < consts.size(); i++)
            if (consts[i] == v) return (uint32_t)i;
        consts.push_back(v);
        return (uint32_t)consts.size() - 1;
    }
    bool region_open = false;       // the next cell pushed starts a new assign_region call
    void begin_region() { region_open = true; }
    void push(uint8_t kind, uint32_t index, bool gate) {
        cells.push_back(imt_trace_cell{kind, (uint8_t)(gate ? 1 : 0), (uint16_t)(region_open ? 1 : 0), index});
        region_open = false;
    }
    void constant(const HFr& v, bool gate = false) { push(IMT_CELL_CONST, cst(v), gate); }
    void existing(Ref r, bool gate = false) { push(r.kind == IMT_CELL_WITNESS ? IMT_CELL_COPY : r.kind, r.index, gate); }
    Ref witness() {
        push(IMT_CELL_WITNESS, n_witness, false);
        return Ref{IMT_CELL_WITNESS, n_witness++};
    }
    // GateChip (gates/flex_gate.rs)
    Ref add_const(Ref a, const HFr& k) { begin_region(); existing(a, true); constant(k); constant(hp.F.one()); return witness(); }
    Ref sum3(Ref x, Ref in, const HFr& k) {          // gate.sum([x, in, Constant(k)]): two chained gates
        begin_region();
        existing(x, true); existing(in); constant(hp.F.one());
        witness();
        cells.back().gate = 1;
        constant(k); constant(hp.F.one());
        return witness();
    }
    Ref mul(Ref a, Ref b) { begin_region(); constant(hp.F.zero(), true); existing(a); existing(b); return witness(); }
    Ref mul_add_const(Ref a, Ref b, const HFr& k) { begin_region(); constant(k, true); existing(a); existing(b); return witness(); }
    Ref mul_const_add(Ref a, const HFr& k, Ref c) { begin_region(); existing(c, true); existing(a); constant(k); return witness(); }
    Ref inner(const Ref s[3], const HFr row[3]) {    // gate.inner_product(s, Constant(row)): three chained gates
        begin_region();
        constant(hp.F.zero(), true);
        Ref w{};
        for (int i = 0; i < 3; i++) {
            existing(s[i]); constant(row[i]);
            w = witness();
            if (i < 2) cells.back().gate = 1;
        }
        return w;
    }
    // PoseidonState (poseidon/hasher/state.rs)
    Ref x5c(Ref x, const HFr& k) {
        Ref x2 = mul(x, x), x4 = mul(x2, x2);
        return mul_add_const(x, x4, k);
    }
    void apply_mds(Ref s[3], const HFr m[3][3]) {
        Ref r[3];
        for (int i = 0; i < 3; i++) r[i] = inner(s, m[i]);
        s[0] = r[0]; s[1] = r[1]; s[2] = r[2];
    }
    void permutation(Ref s[3], const Ref* inputs, int n_in) {
        const auto& F = hp.F;
        s[0] = add_const(s[0], hp.tr_start[0][0]);                               // absorb_with_pre_constants
        for (int i = 0; i < n_in; i++) s[1 + i] = sum3(s[1 + i], inputs[i], hp.tr_start[0][1 + i]);
        for (int i = 0, j = n_in + 1; j < 3; i++, j++)
            s[j] = add_const(s[j], i == 0 ? F.add(hp.tr_start[0][j], F.one()) : hp.tr_start[0][j]);
        for (int r = 1; r <= 4; r++) {
            for (int i = 0; i < 3; i++) s[i] = x5c(s[i], hp.tr_start[r][i]);
            apply_mds(s, r == 4 ? hp.tr_pre : hp.mds);
        }
        for (int p = 0; p < 57; p++) {
            s[0] = x5c(s[0], hp.tr_partial[p]);
            Ref r0 = inner(s, hp.tr_row[p]);
            Ref r1 = mul_const_add(s[0], hp.tr_col_hat[p][0], s[1]);
            Ref r2 = mul_const_add(s[0], hp.tr_col_hat[p][1], s[2]);
            s[0] = r0; s[1] = r1; s[2] = r2;
        }
        for (int r = 0; r < 4; r++) {
            for (int i = 0; i < 3; i++) s[i] = x5c(s[i], r < 3 ? hp.tr_end[r][i] : F.zero());
            apply_mds(s, hp.mds);
        }
    }
    uint32_t hash(int arity) {                       // fix_len_array_squeeze; returns the output's trace row
        Ref in[3] = {{IMT_CELL_INPUT, 0}, {IMT_CELL_INPUT, 1}, {IMT_CELL_INPUT, 2}};
        Ref s[3] = {{IMT_CELL_INIT, 0}, {IMT_CELL_INIT, 1}, {IMT_CELL_INIT, 2}};
        permutation(s, in, 2);
        permutation(s, in + 2, arity == 3 ? 1 : 0);
        return s[1].index;
    }
};

}  // namespace

extern "C" size_t imt_hash_trace_rows(int arity) {
    return arity == 2 ? (size_t)dev::TRACE_ROWS_H2 : arity == 3 ? (size_t)dev::TRACE_ROWS_H3 : 0;
}

extern "C" int imt_hash_trace_layout(imt_ctx* c, int arity, imt_trace_cell* cells, size_t cells_cap, size_t* n_cells,
                                     void* constants, size_t const_cap, size_t* n_constants, uint32_t* out_row,
                                     unsigned flags) {
    if (!c) return IMT_ERR_ARG;
    if (arity != 2 && arity != 3) return c->fail(IMT_ERR_ARG, "arity must be 2 or 3");
    if (flags & IMT_DEVICE_PTRS) return c->fail(IMT_ERR_ARG, "imt_hash_trace_layout takes host pointers");
    const unsigned fmt = flags & IMT_FMT_MASK;
    if (fmt == 3) return c->fail(IMT_ERR_ARG, "unknown field-element format");
    Walker w(c->hp);
    const uint32_t row = w.hash(arity);
    if (w.n_witness != imt_hash_trace_rows(arity)) return c->fail(IMT_ERR_INTERNAL, "trace layout and kernel disagree");
    if (n_cells) *n_cells = w.cells.size();
    if (n_constants) *n_constants = w.consts.size();
    if (out_row) *out_row = row;
    if (cells) {
        if (cells_cap < w.cells.size()) return c->fail(IMT_ERR_RANGE, "cells: need %zu entries", w.cells.size());
        std::memcpy(cells, w.cells.data(), w.cells.size() * sizeof(imt_trace_cell));
    }
    if (constants) {
        if (const_cap < w.consts.size()) return c->fail(IMT_ERR_RANGE, "constants: need %zu entries", w.consts.size());
        uint8_t* o = (uint8_t*)constants;
        for (size_t i = 0; i < w.consts.size(); i++, o += 32) {
            if (fmt == IMT_FMT_CANONICAL) {
                c->hp.F.to_bytes(o, w.consts[i]);
            } else if (fmt == IMT_FMT_MONT256) {
                std::memcpy(o, w.consts[i].l, 32);            // HFr IS the [u64; 4] of a halo2curves Fr
            } else {
                uint32_t words[8];
                dev::pack(words, c->hp.to_dev(w.consts[i]));
                std::memcpy(o, words, 32);
            }
        }
    }
    return IMT_OK;
}
