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
    std::vector<uint32_t> lookup;          // witness rows RangeChip::range_check adds to the lookup table (the limbs)
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
        lookup.push_back(last.index);
        for (unsigned i = 1; i < L; i++) {
            last = witness();
            lookup.push_back(last.index);
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

// new advice values of one is_less_than: two range.is_less_than (2 L + 4 rows each with L = ceil(128 / lookup_bits) + 1
// limbs), two is_equal (4 each), not x4, mul x4, or (3): 4 L + 27; of a whole insert_leaf outside its hashes: 20 + 2 K +
// 16 depth (imt.h)
extern "C" size_t imt_less_than_trace_rows(unsigned lookup_bits) {
    return (lookup_bits < 1 || lookup_bits > 28) ? 0 : 4 * (size_t)((128 + lookup_bits - 1) / lookup_bits + 1) + 27;
}
extern "C" size_t imt_insert_gadget_rows(unsigned depth, unsigned lookup_bits) {
    const size_t k = imt_less_than_trace_rows(lookup_bits);
    return (k && depth >= 1 && depth <= IMT_MAX_DEPTH) ? 20 + 2 * k + 16 * (size_t)depth : 0;
}
// ... of one verify_non_inclusion alone (:127-229): insert_leaf's rows up to and including the second comparison
extern "C" size_t imt_non_inclusion_gadget_rows(unsigned depth, unsigned lookup_bits) {
    const size_t k = imt_less_than_trace_rows(lookup_bits);
    return (k && depth >= 1 && depth <= IMT_MAX_DEPTH) ? 17 + 2 * k + 4 * (size_t)depth : 0;
}
extern "C" int imt_less_than_lookup_rows(unsigned lookup_bits, uint32_t* rows, size_t cap, size_t* n_rows);

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
    {   // the closed form of imt_less_than_lookup_rows against the limb cells this walk has just laid down
        std::vector<uint32_t> lk(w.lookup.size() + 1);
        size_t nl = 0;
        if (imt_less_than_lookup_rows(lookup_bits, lk.data(), lk.size(), &nl) || nl != w.lookup.size() ||
            std::memcmp(lk.data(), w.lookup.data(), nl * sizeof(uint32_t)))
            return c->fail(IMT_ERR_INTERNAL, "lookup rows: closed form and layout disagree");
    }
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

// Which rows of that column the RangeChip also constrains through its LOOKUP table: range_check(shifted, padded +
// lookup_bits) decomposes each shifted difference into lookup_bits-wide limbs and calls add_cell_to_lookup on every limb
// cell (padded + lookup_bits is a multiple of lookup_bits: no scaled last limb).  Row numbers (trace rows of
// imt_less_than_trace_batch) in column order: the L limbs of the high-limb comparison, then the L of the low-limb one.
// Arithmetic on sizes only: no context, no GPU.
extern "C" int imt_less_than_lookup_rows(unsigned lookup_bits, uint32_t* rows, size_t cap, size_t* n_rows) {
    if (lookup_bits < 1 || lookup_bits > 28) return IMT_ERR_RANGE;
    const unsigned L = (128 + lookup_bits - 1) / lookup_bits + 1;
    // rows of one range.is_less_than: shifted, shift_a, limb 0, then (limb i, running sum i) for i = 1 .. L-1, then the 3
    // of is_zero = 2 L + 4; gate.is_equal adds 4 (difference + is_zero); the low-limb comparison follows both
    const uint32_t per_range = 2 * L + 4, per_half = per_range + 4;
    if (n_rows) *n_rows = 2 * (size_t)L;
    if (!rows) return IMT_OK;
    if (cap < 2 * (size_t)L) return IMT_ERR_RANGE;
    for (unsigned h = 0; h < 2; h++) {
        uint32_t* o = rows + h * L;
        const uint32_t base = h * per_half;
        o[0] = base + 2;
        for (unsigned i = 1; i < L; i++) o[i] = base + 3 + 2 * (i - 1);
    }
    return IMT_OK;
}

// The same for the glue rows of a whole insert_leaf (imt_insert_gadget_trace_batch): its two is_less_than calls start
// at glue row 10 (behind is_equal [4], the four limbs [4], two mul_add [2]) and at 17 + K + 4 depth (behind the first
// comparison [K], select [3], the low leaf's path [1 + 4 depth] and the limbs of low.val with their mul_add [3]).
extern "C" int imt_insert_gadget_lookup_rows(unsigned depth, unsigned lookup_bits, uint32_t* rows, size_t cap, size_t* n_rows) {
    if (!imt_insert_gadget_rows(depth, lookup_bits)) return IMT_ERR_RANGE;
    size_t per = 0;
    int rc = imt_less_than_lookup_rows(lookup_bits, nullptr, 0, &per);
    if (rc) return rc;
    if (n_rows) *n_rows = 2 * per;
    if (!rows) return IMT_OK;
    if (cap < 2 * per) return IMT_ERR_RANGE;
    if ((rc = imt_less_than_lookup_rows(lookup_bits, rows, per, nullptr))) return rc;
    const uint32_t K = (uint32_t)imt_less_than_trace_rows(lookup_bits);
    const uint32_t first = 10, second = 17 + K + 4 * depth;
    for (size_t i = 0; i < per; i++) {
        rows[per + i] = rows[i] + second;
        rows[i] += first;
    }
    return IMT_OK;
}
