// emul_device.cpp -- TEST-ONLY host build of the device arithmetic (imt_device.hpp) and
// of the host table generator, so that `-m "not gpu"` tests can compare the radix-2^29
// Montgomery code, the optimised Poseidon schedule and the index kernels' per-element
// logic with the oracle on a machine without a GPU.  Not part of the shipped library.
#include "imt_device.hpp"
#include "imt_trace_device.hpp"
#include "imt_params.hpp"
#include <cstring>
#include <string>

using namespace imt;
static HostPoseidon* g_hp;
static dev::PoseidonConsts g_consts;
static dev::TraceConsts g_tconsts;

extern "C" int emul_init(void) {
    if (g_hp) return 0;
    g_hp = new HostPoseidon();
    std::string err;
    if (!g_hp->init(err)) return -1;
    g_hp->fill_consts(g_consts);
    g_hp->fill_trace_consts(g_tconsts);
    return 0;
}
extern "C" int emul_consts_size(void) { return (int)sizeof(dev::PoseidonConsts); }

// out = hash2/hash3 of canonical little-endian inputs, through the device code path
extern "C" int emul_hash(const uint8_t* in, int arity, uint8_t* out, unsigned fmt_in, unsigned fmt_out) {
    dev::Fe a, b, c, o;
    bool ok = dev::load_fe(g_consts, a, in, fmt_in);
    ok &= dev::load_fe(g_consts, b, in + 32, fmt_in);
    c = a;
    if (arity == 3) ok &= dev::load_fe(g_consts, c, in + 64, fmt_in);
    dev::hash23(g_consts, o, a, b, c, arity == 3);
    {   // the variant k_sweep uses (third input parked in a stash until the second permutation) must agree
        dev::Fe o2;
        uint32_t stash[dev::NL * 4];
        for (int i = 0; i < dev::NL; i++) stash[i * 4] = c.v[i];
        dev::hash23_stashed(g_consts, o2, a, b, arity == 3, stash, 4);
        if (!dev::fe_eq(o, o2)) return -12;
    }
    dev::store_fe(g_consts, out, o, fmt_out);
    return ok ? 0 : -5;
}
extern "C" int emul_permute(const uint8_t* in, uint8_t* out) {
    dev::Fe s[3];
    bool ok = true;
    for (int i = 0; i < 3; i++) ok &= dev::load_fe(g_consts, s[i], in + 32 * i, dev::FMT_CANONICAL);
    dev::permute(g_consts, s, g_consts.rc_full[0]);
    for (int i = 0; i < 3; i++) { dev::canonicalize(s[i]); dev::store_fe(g_consts, out + 32 * i, s[i], dev::FMT_CANONICAL); }
    return ok ? 0 : -5;
}
// r = a*b (field product of canonical inputs) via mont_mul / mont_sqr
extern "C" void emul_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, int use_sqr) {
    dev::Fe x, y, r;
    dev::load_fe(g_consts, x, a, dev::FMT_CANONICAL);
    dev::load_fe(g_consts, y, b, dev::FMT_CANONICAL);
    if (use_sqr) dev::mont_sqr(r, x); else dev::mont_mul(r, x, y);
    dev::canonicalize(r);
    dev::store_fe(g_consts, out, r, dev::FMT_CANONICAL);
}
extern "C" void emul_convert(const uint8_t* in, uint8_t* out, unsigned fmt_in, unsigned fmt_out) {
    dev::Fe x;
    dev::load_fe(g_consts, x, in, fmt_in);
    dev::store_fe(g_consts, out, x, fmt_out);
}
// host-side (product) Poseidon used for table self-checks
extern "C" void emul_host_hash(const uint8_t* in, int arity, uint8_t* out) {
    HFr a, b, c;
    g_hp->F.from_bytes(a, in);
    g_hp->F.from_bytes(b, in + 32);
    HFr h;
    if (arity == 3) { g_hp->F.from_bytes(c, in + 64); h = g_hp->hash3(a, b, c); }
    else h = g_hp->hash2(a, b);
    g_hp->F.to_bytes(out, h);
}

// ---- index logic of the batch-insertion sweep (imt_sweep.hpp), one level on the host ----
#include "imt_sweep.hpp"
extern "C" void emul_merge_level(const uint32_t* node, const uint32_t* time, const uint32_t* rs, const uint32_t* re,
                                 uint32_t total, uint32_t* o_node, uint32_t* o_time, uint32_t* o_rs, uint32_t* o_re,
                                 uint32_t* o_from, int32_t* o_sibsrc, uint32_t* o_node_below) {
    sweep::LevelTable in{node, time, rs, re};
    sweep::LevelOut out{o_node, o_time, o_rs, o_re, o_from, o_sibsrc, o_node_below, nullptr};
    for (uint32_t k = 0; k < total; k++) sweep::merge_element(in, out, k, total);
}

// ---- GPU-prepare logic (imt_prep_logic.hpp) on the host ----
#include "imt_prep_logic.hpp"
extern "C" void emul_sparse_table(uint32_t* st, uint32_t n, int levels) {
    for (int k = 1; k < levels; k++)
        for (uint32_t j = 0; j + (1u << k) <= n; j++) {
            uint32_t a = st[(uint64_t)(k - 1) * n + j], b = st[(uint64_t)(k - 1) * n + j + (1u << (k - 1))];
            st[(uint64_t)k * n + j] = a < b ? a : b;
        }
}
extern "C" uint32_t emul_nsl(const uint32_t* st, uint32_t n, int levels, uint32_t j) {
    return imt::prep::nearest_smaller_left(st, n, levels, j);
}
extern "C" uint32_t emul_nsr(const uint32_t* st, uint32_t n, int levels, uint32_t j) {
    return imt::prep::nearest_smaller_right(st, n, levels, j);
}
// snapshot check of imt_itree_load: OR of load_check_rank over every rank
extern "C" int emul_load_check(const uint8_t* pre, uint32_t n, uint64_t base, const uint32_t* idx) {
    int e = 0;
    for (uint32_t r = 0; r < n; r++) e |= prep::load_check_rank(pre, n, base, idx, r);
    return e;
}
extern "C" uint32_t emul_count_below(const uint8_t* val, const uint32_t* sorted, uint32_t M, const uint8_t* x) {
    return imt::prep::count_below(val, sorted, M, x);
}

// ---- f1: the witness-trace device code (imt_trace_device.hpp) on the host: rows [n_rows][32] in fmt_out ----
extern "C" int emul_hash_trace(const uint8_t* in, int arity, uint8_t* rows, unsigned fmt_in, unsigned fmt_out) {
    dev::Fe a, b, c;
    bool ok = dev::load_fe(g_consts, a, in, fmt_in);
    ok &= dev::load_fe(g_consts, b, in + 32, fmt_in);
    c = a;
    if (arity == 3) ok &= dev::load_fe(g_consts, c, in + 64, fmt_in);
    dev::TraceSink o{rows, 32};
    if (fmt_out == dev::FMT_MONT256) dev::hash_trace<dev::FMT_MONT256>(g_consts, g_tconsts, o, a, b, c, arity == 3);
    else if (fmt_out == dev::FMT_DEVICE) dev::hash_trace<dev::FMT_DEVICE>(g_consts, g_tconsts, o, a, b, c, arity == 3);
    else dev::hash_trace<dev::FMT_CANONICAL>(g_consts, g_tconsts, o, a, b, c, arity == 3);
    return ok ? (int)((o.p - rows) / 32) : -5;
}
// store_mont256 alone: 9 normalised limbs of a value below 4p -> the 32 bytes it stores (value / 32 mod p)
extern "C" void emul_store_mont256(const uint32_t* limbs, uint8_t* out) {
    dev::Fe a;
    for (int i = 0; i < dev::NL; i++) a.v[i] = limbs[i];
    alignas(16) uint8_t row[32];
    dev::store_mont256(row, a);
    std::memcpy(out, row, 32);
}
// the product's cell layout (imt_trace_layout.cpp) without a GPU: a context that only carries the tables
#include "imt_ctx.hpp"
extern "C" int emul_trace_layout(int arity, imt_trace_cell* cells, size_t cells_cap, size_t* n_cells, void* constants,
                                 size_t const_cap, size_t* n_constants, uint32_t* out_row, unsigned flags) {
    static imt_ctx* fake = nullptr;
    if (!fake) {
        fake = new imt_ctx();
        std::string err;
        if (!fake->hp.init(err)) return -12;
    }
    return imt_hash_trace_layout(fake, arity, cells, cells_cap, n_cells, constants, const_cap, n_constants, out_row, flags);
}
// f3: the cell layout of one is_less_than (imt_gadget_layout.cpp), the same way
extern "C" int emul_less_than_layout(int lookup_bits, imt_trace_cell* cells, size_t cells_cap, size_t* n_cells, void* constants,
                                     size_t const_cap, size_t* n_constants, uint32_t* out_row, unsigned flags) {
    static imt_ctx* fake = nullptr;
    if (!fake) {
        fake = new imt_ctx();
        std::string err;
        if (!fake->hp.init(err)) return -12;
    }
    return imt_less_than_trace_layout(fake, (unsigned)lookup_bits, cells, cells_cap, n_cells, constants, const_cap, n_constants,
                                      out_row, flags);
}
