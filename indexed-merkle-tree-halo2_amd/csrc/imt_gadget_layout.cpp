// imt_gadget_layout.cpp -- f3: the advice column of ONE is_less_than call of the reference
// (/root/reference/src/indexed_merkle_tree.rs:98-125), cell by cell (imt_less_than_trace_layout), so that a chip can
// turn the GPU's rows (imt_less_than_trace_batch, or the is_less_than stretches of imt_insert_gadget_trace_batch) into
// ctx.assign_region calls without any field arithmetic on the host.  Cell order: the published halo2-lib v0.4.x
// GateChip / RangeChip (gates/flex_gate.rs, gates/range.rs) over the vertical gate q (a + b c - d) = 0; halo2-base is
// not vendored in the reference, so the order is UNPINNED BY THE REFERENCE like the f1 layout (imt_trace_layout.cpp).
// Host code only: no values are computed here, only where each one comes from.
#include "imt_ctx.hpp"
#include "imt_device.hpp"
#include <cstring>
#include <vector>

using namespace imt;

namespace {

struct Ref { uint8_t kind; uint32_t index; };

struct GWalker {
    const HostPoseidon& hp;
    std::vector<imt_trace_cell> cells;
    std::vector<HFr> consts;
    uint32_t n_witness = 0;
    bool region_open = false;
    explicit GWalker(const HostPoseidon& h) : hp(h) {}

    uint32_t cst(const HFr& v) {
        for (size_t i = 0; i < consts.size(); i++)
            if (consts[i] == v) return (uint32_t)i;
        consts.push_back(v);
        return (uint32_t)consts.size() - 1;
    }
    void begin_region() { region_open = true; }
    void push(uint8_t kind, uint32_t index, bool gate) {
        cells.push_back(imt_trace_cell{kind, (uint8_t)(gate ? 1 : 0), (uint16_t)(region_open ? 1 : 0), index});
        region_open = false;
    }
    void constant(const HFr& v, bool gate = false) { push(IMT_CELL_CONST, cst(v), gate); }
    void existing(Ref r, bool gate = false) { push(r.kind == IMT_CELL_WITNESS ? IMT_CELL_COPY : r.kind, r.index, gate); }
    Ref witness(bool gate = false) {
        push(IMT_CELL_WITNESS, n_witness, gate);
        return Ref{IMT_CELL_WITNESS, n_witness++};
    }
    HFr pow2(unsigned e) const {
        HFr r = hp.F.one();
        const HFr two = hp.F.add(hp.F.one(), hp.F.one());
        for (unsigned i = 0; i < e; i++) r = hp.F.mul(r, two);
        return r;
    }
    // GateChip
    Ref sub(Ref a, Ref b) { begin_region(); Ref r = witness(true); existing(b); constant(hp.F.one()); existing(a); return r; }
    Ref not_(Ref a) { begin_region(); Ref r = witness(true); existing(a); constant(hp.F.one()); constant(hp.F.one()); return r; }
    Ref mul(Ref a, Ref b) { begin_region(); constant(hp.F.zero(), true); existing(a); existing(b); return witness(); }
    Ref or_(Ref a, Ref b) {
        begin_region();
        witness(true); constant(hp.F.one()); existing(b); constant(hp.F.one());
        existing(b, true); existing(a); witness();
        return witness();
    }
    Ref is_zero(Ref a) {
        begin_region();
        witness(true); existing(a); witness(); constant(hp.F.one());
        constant(hp.F.zero(), true); existing(a);
        Ref r = witness();
        constant(hp.F.zero());
        return r;
    }
    Ref is_equal(Ref a, Ref b) { return is_zero(sub(a, b)); }
    // RangeChip::is_less_than(a, b, 128)
    Ref range_lt(Ref a, Ref b, unsigned lb) {
        const unsigned k = (128 + lb - 1) / lb, padded = k * lb, L = k + 1;
        begin_region();
        witness(true); existing(b); constant(hp.F.one());
        witness(true); constant(hp.F.sub(hp.F.zero(), pow2(padded))); constant(hp.F.one()); existing(a);
        begin_region();
        Ref last = witness(L > 1);                      // limb 0 (limb_bases[0] = 1) opens the running sum
        for (unsigned i = 1; i < L; i++) {
            last = witness();
            constant(pow2(i * lb));
            witness(i + 1 < L);
        }
        return is_zero(last);
    }
    uint32_t less_than(unsigned lb) {
        const Ref a_q{IMT_CELL_INPUT, 0}, a_r{IMT_CELL_INPUT, 1}, b_q{IMT_CELL_INPUT, 2}, b_r{IMT_CELL_INPUT, 3};
        Ref msb_lt = range_lt(a_q, b_q, lb);
        Ref msb_eq = is_equal(a_q, b_q);
        Ref lsb_lt = range_lt(a_r, b_r, lb);
        Ref lsb_eq = is_equal(a_r, b_r);
        Ref c_not = not_(msb_eq), a_not = not_(msb_lt), c = not_(c_not), d_not = not_(lsb_eq);
        Ref rhs = mul(mul(mul(a_not, lsb_lt), c), d_not);
        Ref lhs = mul(msb_lt, c_not);
        return or_(lhs, rhs).index;
    }
};

}  // namespace

extern "C" size_t imt_less_than_trace_rows(unsigned lookup_bits);

extern "C" int imt_less_than_trace_layout(imt_ctx* c, unsigned lookup_bits, imt_trace_cell* cells, size_t cells_cap,
                                          size_t* n_cells, void* constants, size_t const_cap, size_t* n_constants,
                                          uint32_t* out_row, unsigned flags) {
    if (!c) return IMT_ERR_ARG;
    if (flags & IMT_DEVICE_PTRS) return c->fail(IMT_ERR_ARG, "imt_less_than_trace_layout takes host pointers");
    const unsigned fmt = flags & IMT_FMT_MASK;
    if (fmt == 3) return c->fail(IMT_ERR_ARG, "unknown field-element format");
    if (lookup_bits < 1 || lookup_bits > 28) return c->fail(IMT_ERR_RANGE, "lookup_bits %u out of [1, 28]", lookup_bits);
    GWalker w(c->hp);
    const uint32_t row = w.less_than(lookup_bits);
    if (w.n_witness != imt_less_than_trace_rows(lookup_bits)) return c->fail(IMT_ERR_INTERNAL, "layout and kernel disagree");
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
                std::memcpy(o, w.consts[i].l, 32);
            } else {
                uint32_t words[8];
                dev::pack(words, c->hp.to_dev(w.consts[i]));
                std::memcpy(o, words, 32);
            }
        }
    }
    return IMT_OK;
}
