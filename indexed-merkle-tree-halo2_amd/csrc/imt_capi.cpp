// imt_capi.cpp -- the extern "C" boundary declared in include/imt.h (context, batched
// hashes, dense tree, path recompute, non-membership, insert witness, multi-GPU helpers).
// Host code only; the kernels are in imt_kernels.hip and imt_prep.hip.
#include "imt_ctx.hpp"
#include "imt_gadget.hpp"
#include <cstring>
#include <new>

using namespace imt;

// ------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------
void* imt_ctx::dev_scratch(size_t slot, size_t bytes) {
    if (slot >= scratch.size()) scratch.resize(slot + 1);
    Scratch& s = scratch[slot];
    if (s.cap >= bytes && s.p) return s.p;
    if (s.p) {
        hipStreamSynchronize(stream);
        hipFree(s.p);
        s.p = nullptr;
        s.cap = 0;
    }
    size_t cap = bytes < 4096 ? 4096 : bytes + bytes / 4;
    hipError_t e = hipMalloc(&s.p, cap);
    if (e != hipSuccess) {
        s.p = nullptr;
        hip_fail(e, "hipMalloc(scratch)");
        return nullptr;
    }
    s.cap = cap;
    return s.p;
}

void imt_ctx::trim_scratch(size_t slot, size_t keep_below) {
    if (slot >= scratch.size()) return;
    Scratch& s = scratch[slot];
    if (!s.p || s.cap <= keep_below) return;
    hipStreamSynchronize(stream);
    hipFree(s.p);
    s.p = nullptr;
    s.cap = 0;
}

int imt_ctx::set_device() {
    IMT_HIP(this, hipSetDevice(device));
    return IMT_OK;
}
int imt_ctx::clear_err() {
    IMT_HIP(this, hipMemsetAsync(d_err, 0, sizeof(int), stream));
    return IMT_OK;
}
int imt_ctx::sync_and_check() {
    int h = 0;
    IMT_HIP(this, hipMemcpyAsync(&h, d_err, sizeof(int), hipMemcpyDeviceToHost, stream));
    IMT_HIP(this, hipStreamSynchronize(stream));
    if (h) {
        IMT_HIP(this, hipMemsetAsync(d_err, 0, sizeof(int), stream));
        return fail(IMT_ERR_NONCANONICAL, "a field element in the input is not reduced (>= p)");
    }
    return IMT_OK;
}

hipEvent_t imt_ctx::prof_event() {
    if (!prof_pool.empty()) {
        hipEvent_t e = prof_pool.back();
        prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
int imt_ctx::prof_begin(int cls, hipStream_t on) {
    if (!profiling) return -1;
    ProfPair p{prof_event(), prof_event(), cls};
    if (!p.a || !p.b) return -1;
    (void)hipEventRecord(p.a, on ? on : stream);
    prof_pending.push_back(p);
    return (int)prof_pending.size() - 1;
}
void imt_ctx::prof_end(int idx, hipStream_t on) {
    if (idx >= 0) (void)hipEventRecord(prof_pending[(size_t)idx].b, on ? on : stream);
}

extern "C" int imt_host_alloc(imt_ctx* c, size_t bytes, void** out) {
    if (!c || !out || bytes == 0) return c ? c->fail(IMT_ERR_ARG, "null / empty host allocation") : IMT_ERR_ARG;
    int rc = c->set_device();
    if (rc) return rc;
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return c->fail(IMT_ERR_ALLOC, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    *out = p;
    return IMT_OK;
}
extern "C" int imt_host_free(imt_ctx* c, void* ptr) {
    if (!c) return IMT_ERR_ARG;
    if (!ptr) return IMT_OK;
    IMT_HIP(c, hipHostFree(ptr));
    return IMT_OK;
}

extern "C" int imt_measure_mad_peak(imt_ctx* c, double* gmads) {
    if (!c || !gmads) return IMT_ERR_ARG;
    int rc = c->set_device();
    if (rc) return rc;
    hipDeviceProp_t prop;
    IMT_HIP(c, hipGetDeviceProperties(&prop, c->device));
    const unsigned blocks = (unsigned)prop.multiProcessorCount * 8;    // 8 waves per SIMD
    const int iters = 4096;
    uint32_t* out = (uint32_t*)c->dev_scratch(0, (size_t)blocks * 256 * 4);
    if (!out) return IMT_ERR_HIP;
    hipEvent_t e0, e1;
    IMT_HIP(c, hipEventCreate(&e0));
    IMT_HIP(c, hipEventCreate(&e1));
    launch::mad_peak(c->stream, out, blocks, iters);                    // warm-up
    IMT_HIP(c, hipEventRecord(e0, c->stream));
    for (int r = 0; r < 3; r++) launch::mad_peak(c->stream, out, blocks, iters);
    IMT_HIP(c, hipEventRecord(e1, c->stream));
    IMT_HIP(c, hipEventSynchronize(e1));
    float ms = 0;
    IMT_HIP(c, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *gmads = 3.0 * blocks * 256.0 * 8.0 * iters / (ms * 1e-3) / 1e9;
    return IMT_OK;
}
extern "C" int imt_profile_enable(imt_ctx* c, int on) {
    if (!c) return IMT_ERR_ARG;
    c->profiling = on != 0;
    return IMT_OK;
}
extern "C" int imt_profile_read(imt_ctx* c, double* out) {
    if (!c || !out) return IMT_ERR_ARG;
    int rc = c->set_device();
    if (rc) return rc;
    IMT_HIP(c, hipStreamSynchronize(c->stream));
    for (auto st : c->side_streams) IMT_HIP(c, hipStreamSynchronize(st));
    for (auto& p : c->prof_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->prof_ms[p.cls] += ms;
            c->prof_n[p.cls] += 1;
        }
        c->prof_pool.push_back(p.a);
        c->prof_pool.push_back(p.b);
    }
    c->prof_pending.clear();
    for (int k = 0; k < IMT_PROF_CLASSES; k++) {
        out[2 * k] = c->prof_ms[k];
        out[2 * k + 1] = c->prof_n[k];
        c->prof_ms[k] = c->prof_n[k] = 0;
    }
    return IMT_OK;
}

extern "C" int imt_ctx_set_option(imt_ctx* c, int option, uint64_t value) {
    if (!c) return IMT_ERR_ARG;
    switch (option) {
        case IMT_OPT_COOP_MAX_EVENTS:
            if (value > 0xffffffffu) return c->fail(IMT_ERR_RANGE, "value out of range");
            c->coop_max_events = (uint32_t)value;
            return IMT_OK;
        default:
            return c->fail(IMT_ERR_ARG, "unknown option %d", option);
    }
}

extern "C" const char* imt_version(void) { return "imt-hip gfx950 r4"; }

extern "C" int imt_ctx_create(int device, imt_ctx** out) {
    if (!out) return IMT_ERR_ARG;
    *out = nullptr;
    if (device < 0) return IMT_ERR_NO_DEVICE;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device >= count) return IMT_ERR_NO_DEVICE;
    imt_ctx* c = new (std::nothrow) imt_ctx();
    if (!c) return IMT_ERR_ALLOC;
    c->device = device;
    std::string err;
    if (!c->hp.init(err)) { delete c; return IMT_ERR_INTERNAL; }
    dev::PoseidonConsts pc;
    c->hp.fill_consts(pc);
    dev::TraceConsts tc;
    c->hp.fill_trace_consts(tc);
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&c->own_stream)) != hipSuccess ||
        (e = launch::upload_consts(pc)) != hipSuccess || (e = launch::upload_trace_consts(tc)) != hipSuccess || (e = hipMalloc((void**)&c->d_err, sizeof(int))) != hipSuccess ||
        (e = hipMalloc((void**)&c->d_zero, (IMT_MAX_DEPTH + 1) * 32)) != hipSuccess) {
        if (c->own_stream) hipStreamDestroy(c->own_stream);
        if (c->d_err) hipFree(c->d_err);
        delete c;
        return e == hipErrorNoDevice ? IMT_ERR_NO_DEVICE : IMT_ERR_HIP;
    }
    c->stream = c->own_stream;
    e = hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream);
    launch::zero_chain(c->stream, c->d_zero, IMT_MAX_DEPTH);
    if (e != hipSuccess || (e = hipStreamSynchronize(c->stream)) != hipSuccess) {
        imt_ctx_destroy(c);
        return IMT_ERR_HIP;
    }
    *out = c;
    return IMT_OK;
}

extern "C" void imt_ctx_destroy(imt_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (auto& s : c->scratch)
        if (s.p) hipFree(s.p);
    for (auto& p : c->prof_pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto& e : c->prof_pool) hipEventDestroy(e);
    if (c->d_err) hipFree(c->d_err);
    if (c->d_zero) hipFree(c->d_zero);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" const char* imt_last_error(const imt_ctx* c) { return c ? c->last_error.c_str() : "null context"; }

extern "C" int imt_ctx_set_stream(imt_ctx* c, void* s) {
    if (!c) return IMT_ERR_ARG;
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return IMT_OK;
}

extern "C" int imt_ctx_sync(imt_ctx* c) {
    if (!c) return IMT_ERR_ARG;
    int rc = c->set_device();
    if (rc) return rc;
    for (auto st : c->side_streams) IMT_HIP(c, hipStreamSynchronize(st));
    return c->sync_and_check();
}

// ------------------------------------------------------------------------------------
// host/device pointer plumbing
// ------------------------------------------------------------------------------------
namespace {

struct Io {
    imt_ctx* c;
    unsigned flags;
    bool dev;
    int rc = IMT_OK;
    size_t next_slot = 0;
    struct Pending { void* user; const uint8_t* d; size_t bytes; };
    std::vector<Pending> outs;

    Io(imt_ctx* ctx, unsigned f) : c(ctx), flags(f), dev((f & IMT_DEVICE_PTRS) != 0) {}

    // device view of an input buffer
    const uint8_t* in(const void* p, size_t bytes) {
        if (rc || !bytes) return (const uint8_t*)p;
        if (!p) { rc = c->fail(IMT_ERR_ARG, "null input pointer"); return nullptr; }
        if (dev) return (const uint8_t*)p;
        void* d = c->dev_scratch(next_slot++, bytes);
        if (!d) { rc = IMT_ERR_HIP; return nullptr; }
        hipError_t e = hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { rc = c->hip_fail(e, "hipMemcpyAsync H2D"); return nullptr; }
        return (const uint8_t*)d;
    }
    // device buffer an output is produced into (NULL user pointer -> NULL)
    uint8_t* out(void* p, size_t bytes) {
        if (rc || !p || !bytes) return (uint8_t*)(rc ? nullptr : p);
        if (dev) return (uint8_t*)p;
        void* d = c->dev_scratch(next_slot++, bytes);
        if (!d) { rc = IMT_ERR_HIP; return nullptr; }
        outs.push_back({p, (const uint8_t*)d, bytes});
        return (uint8_t*)d;
    }
    // field-element buffers: the kernels move an element as two 16-byte words, so a DEVICE pointer the caller hands
    // over must be 16-byte aligned (include/imt.h); a misaligned one is an argument error here, not a GPU fault there
    bool aligned(const void* p) {
        if (!dev || !p || ((uintptr_t)p & 15u) == 0) return true;
        rc = c->fail(IMT_ERR_ARG, "device pointer %p to field elements is not 16-byte aligned", p);
        return false;
    }
    const uint8_t* in_fe(const void* p, size_t bytes) { return (rc || !bytes || aligned(p)) ? in(p, bytes) : nullptr; }
    uint8_t* out_fe(void* p, size_t bytes) { return (rc || !bytes || aligned(p)) ? out(p, bytes) : nullptr; }
    uint8_t* temp(size_t bytes) {
        if (rc) return nullptr;
        void* d = c->dev_scratch(next_slot++, bytes ? bytes : 1);
        if (!d) rc = IMT_ERR_HIP;
        return (uint8_t*)d;
    }
    // host-pointer mode: copy results back, synchronise and report input errors
    int finish() {
        if (rc) return rc;
        if (dev) return IMT_OK;
        for (auto& o : outs) {
            hipError_t e = hipMemcpyAsync(o.user, o.d, o.bytes, hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) return c->hip_fail(e, "hipMemcpyAsync D2H");
        }
        return c->sync_and_check();
    }
};

launch::SibLayout sib_layout(unsigned flags, unsigned depth, size_t n) {
    if (flags & IMT_SIB_ITEM_MAJOR) return {1, depth};
    return {n, 1};
}

int begin(imt_ctx* c, unsigned flags) {
    if (!c) return IMT_ERR_ARG;
    if ((flags & IMT_FMT_MASK) == 3) return c->fail(IMT_ERR_ARG, "unknown field-element format");
    int rc = c->set_device();
    if (rc) return rc;
    if (!(flags & IMT_DEVICE_PTRS)) return c->clear_err();
    return IMT_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------
// a1 / a10
// ------------------------------------------------------------------------------------
static int hash_n(imt_ctx* c, const void* in, void* out, size_t n, int arity, unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    if (n == 0) return IMT_OK;
    if (!in || !out) return c->fail(IMT_ERR_ARG, "null buffer");
    Io io(c, flags);
    const uint8_t* d_in = io.in_fe(in, n * 32 * (size_t)arity);
    uint8_t* d_out = io.out_fe(out, n * 32);
    if (io.rc) return io.rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    launch::hash_batch(c->stream, d_in, d_out, n, arity, fmt, fmt, c->d_err, c->coop_max_events);
    return io.finish();
}
extern "C" int imt_hash2_batch(imt_ctx* c, const void* in, void* out, size_t n, unsigned flags) {
    return hash_n(c, in, out, n, 2, flags);
}
extern "C" int imt_hash3_batch(imt_ctx* c, const void* in, void* out, size_t n, unsigned flags) {
    return hash_n(c, in, out, n, 3, flags);
}
extern "C" int imt_permute_batch(imt_ctx* c, const void* in, void* out, size_t n, unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    if (n == 0) return IMT_OK;
    if (!in || !out) return c->fail(IMT_ERR_ARG, "null buffer");
    Io io(c, flags);
    const uint8_t* d_in = io.in_fe(in, n * 96);
    uint8_t* d_out = io.out_fe(out, n * 96);
    if (io.rc) return io.rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    launch::permute_batch(c->stream, d_in, d_out, n, fmt, fmt, c->d_err);
    return io.finish();
}

// ------------------------------------------------------------------------------------
// f1: witness trace of hash_fix_len_array
// ------------------------------------------------------------------------------------
extern "C" int imt_hash_trace_batch(imt_ctx* c, const void* in, int arity, size_t n, void* trace, unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    const size_t rows = imt_hash_trace_rows(arity);
    if (!rows) return c->fail(IMT_ERR_ARG, "arity must be 2 or 3");
    if (n == 0) return IMT_OK;
    if (!in || !trace) return c->fail(IMT_ERR_ARG, "null buffer");
    Io io(c, flags);
    const uint8_t* d_in = io.in_fe(in, n * 32 * (size_t)arity);
    uint8_t* d_tr = io.out_fe(trace, n * rows * 32);
    if (io.rc) return io.rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    const bool item_major = flags & IMT_TRACE_ITEM_MAJOR;
    launch::hash_trace(c->stream, d_in, n, arity, d_tr, n, 0, rows, item_major, fmt, fmt, c->d_err);
    return io.finish();
}

namespace {
// The traces of one compute_merkle_root per item, given the chain's pairs (path_pairs ran before): the leaf hash
// (if leaf3) then `depth` path hashes, into `trace` from block row `row0` on (advanced).  ONE launch for the leaf
// hashes, ONE for all levels: every hash of every path is an independent trace.
// appends the trace jobs of one compute_merkle_root call: the leaf hash (if the chain starts from a preimage), then all
// levels of the path as one group of depth * n hashes
void path_trace_jobs(launch::TraceJobs& tj, const uint8_t* d_leaf3, const uint8_t* pairs, unsigned depth, size_t n,
                     size_t& row0, unsigned fmt) {
    if (d_leaf3) {
        tj.j[tj.n_jobs++] = {d_leaf3, n, 3, row0, fmt};
        row0 += dev::TRACE_ROWS_H3;
    }
    if (depth) tj.j[tj.n_jobs++] = {pairs, (size_t)depth * n, 2, row0, IMT_FMT_DEVICE};
    row0 += (size_t)depth * dev::TRACE_ROWS_H2;
}
}  // namespace

extern "C" int imt_path_trace_batch(imt_ctx* c, const void* leaf, const void* leaf3, const uint64_t* index,
                                    const void* sib, unsigned depth, size_t n, void* trace, void* root_out,
                                    unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    if (depth > IMT_MAX_DEPTH) return c->fail(IMT_ERR_RANGE, "depth %u > %d", depth, IMT_MAX_DEPTH);
    if (n == 0) return IMT_OK;
    if ((!leaf) == (!leaf3)) return c->fail(IMT_ERR_ARG, "exactly one of leaf / leaf3 must be given");
    if (!index || !trace || (depth && !sib)) return c->fail(IMT_ERR_ARG, "null buffer");
    const size_t rows_total = (leaf3 ? (size_t)dev::TRACE_ROWS_H3 : 0) + (size_t)depth * dev::TRACE_ROWS_H2;
    Io io(c, flags);
    const uint8_t* d_leaf = leaf ? io.in_fe(leaf, n * 32) : nullptr;
    const uint8_t* d_leaf3 = leaf3 ? io.in_fe(leaf3, n * 96) : nullptr;
    const uint64_t* d_idx = (const uint64_t*)io.in(index, n * 8);
    const uint8_t* d_sib = io.in_fe(sib, (size_t)depth * n * 32);
    uint8_t* d_tr = io.out_fe(trace, n * rows_total * 32);
    uint8_t* d_root = io.out_fe(root_out, n * 32);
    uint8_t* pairs = io.temp((size_t)depth * n * 64);
    if (io.rc) return io.rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    // the chain first (one fast hash per level), which yields the two inputs of every hash on the path
    launch::PathChains pc{};
    pc.c[0] = {d_leaf, d_leaf3, d_idx, d_sib, pairs, d_root};
    pc.n_chains = 1;
    pc.lay = sib_layout(flags, depth, n);     // IMT_TRACE_ITEM_MAJOR is IMT_SIB_ITEM_MAJOR: siblings item-major too, as in imt_insert_trace_batch
    pc.depth = depth; pc.n = n; pc.fmt_in = fmt; pc.fmt_out = fmt; pc.err = c->d_err;
    launch::path_pairs(c->stream, pc, c->coop_max_events);
    size_t row0 = 0;
    launch::TraceJobs tj{};
    tj.trace = d_tr; tj.n_per = n; tj.rows_total = rows_total; tj.item_major = (flags & IMT_TRACE_ITEM_MAJOR) ? 1 : 0;
    tj.err = c->d_err;
    path_trace_jobs(tj, d_leaf3, pairs, depth, n, row0, fmt);
    launch::hash_trace_jobs(c->stream, tj, fmt);
    return io.finish();
}

extern "C" size_t imt_insert_trace_rows(unsigned depth) {
    return 3 * (size_t)dev::TRACE_ROWS_H3 + 4 * (size_t)depth * dev::TRACE_ROWS_H2;
}

extern "C" int imt_insert_trace_batch(imt_ctx* c, const void* low_leaf, const uint64_t* low_index, const void* low_sib,
                                      const void* new_leaf, const uint64_t* new_index, const uint64_t* new_path_index,
                                      const void* new_sib, unsigned depth, size_t n, void* trace, unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    if (depth > IMT_MAX_DEPTH) return c->fail(IMT_ERR_RANGE, "depth %u > %d", depth, IMT_MAX_DEPTH);
    if (n == 0) return IMT_OK;
    if (!low_leaf || !low_index || !new_leaf || !new_index || !trace || (depth && (!low_sib || !new_sib)))
        return c->fail(IMT_ERR_ARG, "null buffer");
    const size_t rows_total = imt_insert_trace_rows(depth);
    Io io(c, flags);
    const uint8_t* d_ll = io.in_fe(low_leaf, n * 96);
    const uint64_t* d_li = (const uint64_t*)io.in(low_index, n * 8);
    const uint8_t* d_ls = io.in_fe(low_sib, (size_t)depth * n * 32);
    const uint8_t* d_nl = io.in_fe(new_leaf, n * 96);
    const uint64_t* d_ni = (const uint64_t*)io.in(new_index, n * 8);
    const uint64_t* d_np = new_path_index ? (const uint64_t*)io.in(new_path_index, n * 8) : d_ni;
    const uint8_t* d_ns = io.in_fe(new_sib, (size_t)depth * n * 32);
    uint8_t* d_tr = io.out_fe(trace, n * rows_total * 32);
    uint8_t* pairs = io.temp(4 * (size_t)depth * n * 64);
    uint8_t* tmp3 = io.temp(n * 96);      // the rewritten low leaf {low.val, new.val, new_index}   :265-269
    uint8_t* tmpz = io.temp(n * 32);      // the zero-leaf hash per item                            :247-251
    if (io.rc) return io.rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    const bool item_major = flags & IMT_TRACE_ITEM_MAJOR;
    launch::insert_trace_inputs(c->stream, d_ll, d_nl, d_ni, n, tmp3, tmpz, fmt, c->d_err);
    // the four compute_merkle_root chains of insert_leaf as ONE launch, in the order the circuit reaches them:
    // :193-204 low leaf, :271-284 rewritten low leaf, :286-294 the zero leaf at the new slot, :299-312 new leaf
    const size_t ps = (size_t)depth * n * 64;
    launch::PathChains pc{};
    pc.c[0] = {nullptr, d_ll, d_li, d_ls, pairs, nullptr};
    pc.c[1] = {nullptr, tmp3, d_li, d_ls, pairs + ps, nullptr};
    pc.c[2] = {tmpz, nullptr, d_np, d_ns, pairs + 2 * ps, nullptr};
    pc.c[3] = {nullptr, d_nl, d_np, d_ns, pairs + 3 * ps, nullptr};
    pc.n_chains = 4;
    pc.lay = sib_layout(flags, depth, n);
    pc.depth = depth; pc.n = n; pc.fmt_in = fmt; pc.fmt_out = fmt; pc.err = c->d_err;
    launch::path_pairs(c->stream, pc, c->coop_max_events);
    // every hash of the call -- 3 leaf hashes and 4 x depth path hashes per item -- traced by ONE launch
    size_t row0 = 0;
    launch::TraceJobs tj{};
    tj.trace = d_tr; tj.n_per = n; tj.rows_total = rows_total; tj.item_major = item_major ? 1 : 0; tj.err = c->d_err;
    for (int k = 0; k < 4; k++) path_trace_jobs(tj, pc.c[k].leaf3, pc.c[k].pairs, depth, n, row0, fmt);
    launch::hash_trace_jobs(c->stream, tj, fmt);
    return io.finish();
}

// ------------------------------------------------------------------------------------
// f3: the advice values of insert_leaf outside hash_fix_len_array (imt_gadget.hip)
// ------------------------------------------------------------------------------------
// row counts: imt_gadget_layout.cpp (host-only arithmetic, shared with the CPU test build)
extern "C" size_t imt_less_than_trace_rows(unsigned lookup_bits);
extern "C" size_t imt_insert_gadget_rows(unsigned depth, unsigned lookup_bits);
extern "C" size_t imt_non_inclusion_gadget_rows(unsigned depth, unsigned lookup_bits);

extern "C" int imt_less_than_trace_batch(imt_ctx* c, const void* a, const void* b, size_t n, unsigned lookup_bits, void* trace,
                                         uint8_t* lt_out, unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    const size_t rows = imt_less_than_trace_rows(lookup_bits);
    if (!rows) return c->fail(IMT_ERR_RANGE, "lookup_bits %u out of [1, 28]", lookup_bits);
    if (n == 0) return IMT_OK;
    if (!a || !b || !trace) return c->fail(IMT_ERR_ARG, "null buffer");
    Io io(c, flags);
    const uint8_t* d_a = io.in_fe(a, n * 32);
    const uint8_t* d_b = io.in_fe(b, n * 32);
    uint8_t* d_tr = io.out_fe(trace, n * rows * 32);
    uint8_t* d_lt = io.out(lt_out, n);
    const unsigned fmt = flags & IMT_FMT_MASK;
    if (fmt != IMT_FMT_CANONICAL && !io.rc) {          // the kernel works on canonical integers
        uint8_t* ca = io.temp(n * 32);
        uint8_t* cb = io.temp(n * 32);
        if (io.rc) return io.rc;
        launch::convert(c->stream, d_a, ca, n, fmt, IMT_FMT_CANONICAL, c->d_err);
        launch::convert(c->stream, d_b, cb, n, fmt, IMT_FMT_CANONICAL, c->d_err);
        d_a = ca;
        d_b = cb;
    } else if (!io.rc) {
        // canonical inputs are still range-checked (>= p sets the context's error word: IMT_ERR_NONCANONICAL at the sync)
        uint8_t* chk = io.temp(n * 32);
        if (io.rc) return io.rc;
        launch::convert(c->stream, d_a, chk, n, IMT_FMT_CANONICAL, IMT_FMT_DEVICE, c->d_err);
        launch::convert(c->stream, d_b, chk, n, IMT_FMT_CANONICAL, IMT_FMT_DEVICE, c->d_err);
    }
    if (io.rc) return io.rc;
    const bool item_major = flags & IMT_TRACE_ITEM_MAJOR;
    launch::less_than_trace(c->stream, d_a, d_b, n, lookup_bits, d_tr, item_major ? 32 : n * 32, item_major ? rows * 32 : 32, d_lt);
    if (fmt != IMT_FMT_CANONICAL) launch::convert(c->stream, d_tr, d_tr, n * rows, IMT_FMT_CANONICAL, fmt, c->d_err);
    return io.finish();
}

extern "C" int imt_insert_gadget_trace_batch(imt_ctx* c, const void* low_leaf, const uint64_t* low_index, const void* low_sib,
                                             const void* new_leaf, const uint64_t* new_index, const uint64_t* new_path_index,
                                             const void* new_sib, const uint8_t* is_largest, unsigned depth,
                                             unsigned lookup_bits, size_t n, void* trace, unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    const size_t rows = imt_insert_gadget_rows(depth, lookup_bits);
    if (!rows) return c->fail(IMT_ERR_RANGE, "depth %u out of [1, %d] or lookup_bits %u out of [1, 28]", depth, IMT_MAX_DEPTH, lookup_bits);
    if (n == 0) return IMT_OK;
    if (!low_leaf || !low_index || !new_leaf || !new_index || !trace || !low_sib || !new_sib || !is_largest)
        return c->fail(IMT_ERR_ARG, "null buffer");
    Io io(c, flags);
    const uint8_t* d_ll = io.in_fe(low_leaf, n * 96);
    const uint64_t* d_li = (const uint64_t*)io.in(low_index, n * 8);
    const uint8_t* d_ls = io.in_fe(low_sib, (size_t)depth * n * 32);
    const uint8_t* d_nl = io.in_fe(new_leaf, n * 96);
    const uint64_t* d_ni = (const uint64_t*)io.in(new_index, n * 8);
    const uint64_t* d_np = new_path_index ? (const uint64_t*)io.in(new_path_index, n * 8) : d_ni;
    const uint8_t* d_ns = io.in_fe(new_sib, (size_t)depth * n * 32);
    const uint8_t* d_lg = io.in(is_largest, n);
    uint8_t* d_tr = io.out_fe(trace, n * rows * 32);
    const size_t ps = (size_t)depth * n * 64;
    uint8_t* pairs = io.temp(4 * ps);
    uint8_t* tmp3 = io.temp(n * 96);      // the rewritten low leaf {low.val, new.val, new_index}   :265-269
    uint8_t* tmpz = io.temp(n * 32);      // the zero-leaf hash per item                            :247-251
    uint8_t* cll = io.temp(n * 96);
    uint8_t* cnl = io.temp(n * 96);
    if (io.rc) return io.rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    // the same path walk as imt_insert_trace_batch: the (left, right) inputs of every path hash of the four chains
    launch::insert_trace_inputs(c->stream, d_ll, d_nl, d_ni, n, tmp3, tmpz, fmt, c->d_err);
    launch::PathChains pc{};
    pc.c[0] = {nullptr, d_ll, d_li, d_ls, pairs, nullptr};
    pc.c[1] = {nullptr, tmp3, d_li, d_ls, pairs + ps, nullptr};
    pc.c[2] = {tmpz, nullptr, d_np, d_ns, pairs + 2 * ps, nullptr};
    pc.c[3] = {nullptr, d_nl, d_np, d_ns, pairs + 3 * ps, nullptr};
    pc.n_chains = 4;
    pc.lay = sib_layout(flags, depth, n);
    pc.depth = depth; pc.n = n; pc.fmt_in = fmt; pc.fmt_out = fmt; pc.err = c->d_err;
    launch::path_pairs(c->stream, pc, c->coop_max_events);
    // the glue kernel works on canonical integers
    launch::convert(c->stream, pairs, pairs, 4 * (size_t)depth * n * 2, IMT_FMT_DEVICE, IMT_FMT_CANONICAL, c->d_err);
    launch::convert(c->stream, d_ll, cll, n * 3, fmt, IMT_FMT_CANONICAL, c->d_err);
    launch::convert(c->stream, d_nl, cnl, n * 3, fmt, IMT_FMT_CANONICAL, c->d_err);
    const bool item_major = flags & IMT_TRACE_ITEM_MAJOR;
    launch::insert_gadget(c->stream, cll, d_li, cnl, d_np, d_lg, pairs, depth, lookup_bits, n, d_tr, item_major ? 32 : n * 32,
                          item_major ? rows * 32 : 32);
    if (fmt != IMT_FMT_CANONICAL) launch::convert(c->stream, d_tr, d_tr, n * rows, IMT_FMT_CANONICAL, fmt, c->d_err);
    return io.finish();
}

// verify_non_inclusion alone (:127-229; BASELINE config 3's gadget): the same kernel stopped behind the second comparison
extern "C" int imt_non_inclusion_gadget_trace_batch(imt_ctx* c, const void* low_leaf, const uint64_t* low_index, const void* low_sib,
                                                    const void* new_val, const uint8_t* is_largest, unsigned depth,
                                                    unsigned lookup_bits, size_t n, void* trace, unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    const size_t rows = imt_non_inclusion_gadget_rows(depth, lookup_bits);
    if (!rows) return c->fail(IMT_ERR_RANGE, "depth %u out of [1, %d] or lookup_bits %u out of [1, 28]", depth, IMT_MAX_DEPTH, lookup_bits);
    if (n == 0) return IMT_OK;
    if (!low_leaf || !low_index || !low_sib || !new_val || !is_largest || !trace) return c->fail(IMT_ERR_ARG, "null buffer");
    Io io(c, flags);
    const uint8_t* d_ll = io.in_fe(low_leaf, n * 96);
    const uint64_t* d_li = (const uint64_t*)io.in(low_index, n * 8);
    const uint8_t* d_ls = io.in_fe(low_sib, (size_t)depth * n * 32);
    const uint8_t* d_nv = io.in_fe(new_val, n * 32);
    const uint8_t* d_lg = io.in(is_largest, n);
    uint8_t* d_tr = io.out_fe(trace, n * rows * 32);
    const size_t ps = (size_t)depth * n * 64;
    uint8_t* pairs = io.temp(ps);
    uint8_t* cll = io.temp(n * 96);
    uint8_t* cnv = io.temp(n * 32);
    if (io.rc) return io.rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    launch::PathChains pc{};
    pc.c[0] = {nullptr, d_ll, d_li, d_ls, pairs, nullptr};
    pc.n_chains = 1;
    pc.lay = sib_layout(flags, depth, n);
    pc.depth = depth; pc.n = n; pc.fmt_in = fmt; pc.fmt_out = fmt; pc.err = c->d_err;
    launch::path_pairs(c->stream, pc, c->coop_max_events);
    launch::convert(c->stream, pairs, pairs, (size_t)depth * n * 2, IMT_FMT_DEVICE, IMT_FMT_CANONICAL, c->d_err);
    launch::convert(c->stream, d_ll, cll, n * 3, fmt, IMT_FMT_CANONICAL, c->d_err);
    launch::convert(c->stream, d_nv, cnv, n, fmt, IMT_FMT_CANONICAL, c->d_err);
    const bool item_major = flags & IMT_TRACE_ITEM_MAJOR;
    launch::non_inclusion_gadget(c->stream, cll, d_li, cnv, d_lg, pairs, depth, lookup_bits, n, d_tr, item_major ? 32 : n * 32,
                                 item_major ? rows * 32 : 32);
    if (fmt != IMT_FMT_CANONICAL) launch::convert(c->stream, d_tr, d_tr, n * rows, IMT_FMT_CANONICAL, fmt, c->d_err);
    return io.finish();
}

// how the glue rows and the hash blocks (imt_insert_trace_batch's order) interleave in insert_leaf's advice column;
// whole = false: verify_non_inclusion's part alone, whose hash blocks are imt_path_trace_batch's (leaf3 form)
static int column_segments(unsigned depth, unsigned lookup_bits, bool whole, imt_column_segment* segs, size_t cap, size_t* n_segs) {
    const size_t k = imt_less_than_trace_rows(lookup_bits);
    if (!k || depth < 1 || depth > IMT_MAX_DEPTH) return IMT_ERR_RANGE;
    std::vector<imt_column_segment> v;
    uint64_t glue = 0, hash = 0;
    auto G = [&](uint64_t rows) {
        if (!v.empty() && v.back().kind == IMT_SEG_GLUE) v.back().n_rows += rows;
        else v.push_back(imt_column_segment{IMT_SEG_GLUE, 0, glue, rows});
        glue += rows;
    };
    auto H = [&](unsigned arity) {
        const uint64_t rows = imt_hash_trace_rows((int)arity);
        v.push_back(imt_column_segment{IMT_SEG_HASH, arity, hash, rows});
        hash += rows;
    };
    auto chain = [&]() {
        G(1);                                  // load_witness(leaf) :88
        for (unsigned l = 0; l < depth; l++) { G(4); H(2); }      // dual_mux, hash :90-93
    };
    G(4 + 4 + 2 + k + 3);                      // verify_non_inclusion up to the select :143-191
    H(3); chain();                             // low leaf hash + verify_merkle_proof :193-204
    G(3 + k);                                  // :206-228
    if (whole) {
        H(3); chain();                         // rewritten low leaf + interim root :271-284
        chain();                               // zero leaf in the interim root :286-294
        H(3); chain();                         // new leaf + new root :299-312
    }
    if (n_segs) *n_segs = v.size();
    if (whole ? (glue != imt_insert_gadget_rows(depth, lookup_bits) || hash != imt_insert_trace_rows(depth))
              : (glue != imt_non_inclusion_gadget_rows(depth, lookup_bits) ||
                 hash != imt_hash_trace_rows(3) + (uint64_t)depth * imt_hash_trace_rows(2)))
        return IMT_ERR_INTERNAL;
    if (segs) {
        if (cap < v.size()) return IMT_ERR_RANGE;
        std::memcpy(segs, v.data(), v.size() * sizeof(imt_column_segment));
    }
    return IMT_OK;
}
extern "C" int imt_insert_column_segments(unsigned depth, unsigned lookup_bits, imt_column_segment* segs, size_t cap, size_t* n_segs) {
    return column_segments(depth, lookup_bits, true, segs, cap, n_segs);
}
extern "C" int imt_non_inclusion_column_segments(unsigned depth, unsigned lookup_bits, imt_column_segment* segs, size_t cap,
                                                 size_t* n_segs) {
    return column_segments(depth, lookup_bits, false, segs, cap, n_segs);
}

// ------------------------------------------------------------------------------------
// a2 / a3 / a4: dense tree
// ------------------------------------------------------------------------------------
extern "C" void imt_tree_free(imt_tree* t) {
    if (!t) return;
    hipSetDevice(t->ctx->device);
    hipStreamSynchronize(t->ctx->stream);
    if (t->d_nodes) hipFree(t->d_nodes);
    if (t->d_off) hipFree(t->d_off);
    if (t->d_len) hipFree(t->d_len);
    delete t;
}

extern "C" int imt_tree_new(imt_ctx* c, const void* leaves, size_t n, unsigned flags, imt_tree** out) {
    if (!c || !out) return IMT_ERR_ARG;
    *out = nullptr;
    // the reference's checks, in its order (src/utils.rs:24-36), then the :45 panic
    if (n == 0) return c->fail(IMT_ERR_NO_LEAVES, "Cannot create Merkle Tree with no leaves");
    if (n != 1 && (n % 2) == 1) return c->fail(IMT_ERR_ODD_LEAVES, "Leaves must be even");
    if (n & (n - 1)) return c->fail(IMT_ERR_NOT_POW2, "leaf count is even but not a power of two");
    int rc = begin(c, flags & ~IMT_DEVICE_PTRS);   // always synchronous: the err word is checked below
    if (rc) return rc;
    if (!leaves) return c->fail(IMT_ERR_ARG, "null leaves");
    imt_tree* t = new (std::nothrow) imt_tree();
    if (!t) return c->fail(IMT_ERR_ALLOC, "out of host memory");
    t->ctx = c;
    t->n_leaves = n;
    size_t nl = 1;
    for (size_t m = n; m > 1; m >>= 1) nl++;
    t->n_levels = nl;
    uint64_t off = 0;
    for (size_t l = 0; l < nl; l++) {
        t->h_off.push_back(off);
        t->h_len.push_back(n >> l);
        off += n >> l;
    }
    hipError_t e;
    if ((e = hipMalloc((void**)&t->d_nodes, off * 32)) != hipSuccess ||
        (e = hipMalloc((void**)&t->d_off, nl * 8)) != hipSuccess ||
        (e = hipMalloc((void**)&t->d_len, nl * 8)) != hipSuccess) {
        imt_tree_free(t);
        return c->hip_fail(e, "hipMalloc(tree)");
    }
    if ((e = hipMemcpyAsync(t->d_off, t->h_off.data(), nl * 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(t->d_len, t->h_len.data(), nl * 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess) {
        imt_tree_free(t);
        return c->hip_fail(e, "hipMemcpyAsync(tree offsets)");
    }
    Io io(c, flags);
    const uint8_t* d_in = io.in_fe(leaves, n * 32);
    if (io.rc) { imt_tree_free(t); return io.rc; }
    launch::convert(c->stream, d_in, t->d_nodes, n, flags & IMT_FMT_MASK, IMT_FMT_DEVICE, c->d_err);
    for (size_t l = 1; l < nl; l++)   // while current_level.len() > 1 (src/utils.rs:41-51)
        launch::tree_level(c->stream, t->d_nodes + t->h_off[l - 1] * 32, t->d_nodes + t->h_off[l] * 32, t->h_len[l],
                           c->coop_max_events);
    rc = c->sync_and_check();
    if (rc) { imt_tree_free(t); return rc; }
    *out = t;
    return IMT_OK;
}

extern "C" size_t imt_tree_num_levels(const imt_tree* t) { return t ? t->n_levels : 0; }

extern "C" int imt_tree_get_level(imt_tree* t, size_t level, void* out, size_t* n_out, unsigned flags) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    if (level >= t->n_levels) return c->fail(IMT_ERR_RANGE, "level %zu out of range", level);
    if (n_out) *n_out = t->h_len[level];
    if (!out) return IMT_OK;
    int rc = begin(c, flags);
    if (rc) return rc;
    Io io(c, flags);
    const size_t n = t->h_len[level];
    uint8_t* d = io.out_fe(out, n * 32);
    if (io.rc) return io.rc;
    launch::convert(c->stream, t->d_nodes + t->h_off[level] * 32, d, n, IMT_FMT_DEVICE, flags & IMT_FMT_MASK, c->d_err);
    return io.finish();
}

extern "C" int imt_tree_get_root(imt_tree* t, void* root, unsigned flags) {
    if (!t || !root) return IMT_ERR_ARG;
    return imt_tree_get_level(t, t->n_levels - 1, root, nullptr, flags);
}

extern "C" int imt_tree_get_proof_batch(imt_tree* t, const uint64_t* index, size_t n, void* proof, unsigned flags) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    int rc = begin(c, flags);
    if (rc) return rc;
    const unsigned depth = (unsigned)(t->n_levels - 1);
    if (n == 0 || depth == 0) return IMT_OK;
    if (!index || !proof) return c->fail(IMT_ERR_ARG, "null buffer");
    if (!(flags & IMT_DEVICE_PTRS))
        for (size_t i = 0; i < n; i++)
            if (index[i] >= t->n_leaves) return c->fail(IMT_ERR_RANGE, "leaf index %llu out of range", (unsigned long long)index[i]);
    Io io(c, flags);
    const uint64_t* d_idx = (const uint64_t*)io.in(index, n * 8);
    uint8_t* d_out = io.out_fe(proof, (size_t)depth * n * 32);
    if (io.rc) return io.rc;
    launch::TreeView tv{t->d_nodes, t->d_off, t->d_len, c->d_zero};
    launch::gather_proof(c->stream, tv, d_idx, n, depth, d_out, sib_layout(flags, depth, n), flags & IMT_FMT_MASK);
    return io.finish();
}

extern "C" int imt_tree_get_proof(imt_tree* t, size_t index, void* proof, void* helper, unsigned flags) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    if (index >= t->n_leaves) return c->fail(IMT_ERR_RANGE, "leaf index %zu out of range", index);
    if (flags & IMT_DEVICE_PTRS) return c->fail(IMT_ERR_ARG, "imt_tree_get_proof takes host pointers");
    uint64_t idx = index;
    int rc = imt_tree_get_proof_batch(t, &idx, 1, proof, flags);
    if (rc || !helper) return rc;
    const unsigned depth = (unsigned)(t->n_levels - 1);
    if (!depth) return IMT_OK;
    Io io(c, flags);
    uint8_t* d = io.out_fe(helper, (size_t)depth * 32);
    if (io.rc) return io.rc;
    launch::write_helpers(c->stream, index, depth, d, flags & IMT_FMT_MASK);
    return io.finish();
}

extern "C" int imt_tree_build(imt_ctx* c, const void* leaves, size_t n, void* levels, void* root, unsigned flags) {
    imt_tree* t = nullptr;
    int rc = imt_tree_new(c, leaves, n, flags, &t);
    if (rc) return rc;
    if (levels) {
        for (size_t l = 0; l < t->n_levels && !rc; l++)
            rc = imt_tree_get_level(t, l, (uint8_t*)levels + t->h_off[l] * 32, nullptr, flags);
    }
    if (!rc && root) rc = imt_tree_get_root(t, root, flags);
    if (!rc && (flags & IMT_DEVICE_PTRS)) rc = c->sync_and_check();
    imt_tree_free(t);
    return rc;
}

// ------------------------------------------------------------------------------------
// a5 / a7 / a8 / a9
// ------------------------------------------------------------------------------------
static int path_common(imt_ctx* c, const void* leaf, const uint64_t* index, bool is_helper, const void* root,
                       const void* sib, unsigned depth, size_t n, void* root_out, uint8_t* ok_out, unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    if (depth > IMT_MAX_DEPTH) return c->fail(IMT_ERR_RANGE, "depth %u > %d", depth, IMT_MAX_DEPTH);
    if (n == 0) return IMT_OK;
    if (!leaf || !index || (depth && !sib)) return c->fail(IMT_ERR_ARG, "null buffer");
    Io io(c, flags);
    const uint8_t* d_leaf = io.in_fe(leaf, n * 32);
    const uint64_t* d_idx = (const uint64_t*)io.in(index, n * 8);
    const uint8_t* d_sib = io.in_fe(sib, (size_t)depth * n * 32);
    const unsigned rstride = (flags & IMT_ROOT_PER_ITEM) ? 32 : 0;
    const uint8_t* d_root = ok_out ? io.in_fe(root, rstride ? n * 32 : 32) : nullptr;
    uint8_t* d_out = io.out_fe(root_out, n * 32);
    uint8_t* d_ok = io.out(ok_out, n);
    if (io.rc) return io.rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    launch::path_root(c->stream, d_leaf, nullptr, d_idx, is_helper, d_sib, sib_layout(flags, depth, n), depth, n, d_out,
                      d_root, rstride, d_ok, fmt, fmt, c->d_err, c->coop_max_events);
    return io.finish();
}
extern "C" int imt_path_root_batch(imt_ctx* c, const void* leaf, const uint64_t* index, const void* sib,
                                   unsigned depth, size_t n, void* root_out, unsigned flags) {
    if (c && n && !root_out) return c->fail(IMT_ERR_ARG, "null root_out");
    return path_common(c, leaf, index, false, nullptr, sib, depth, n, root_out, nullptr, flags);
}
extern "C" int imt_compute_merkle_root_batch(imt_ctx* c, const void* leaf, const uint64_t* helper_mask,
                                             const void* sib, unsigned depth, size_t n, void* root_out,
                                             unsigned flags) {
    if (c && n && !root_out) return c->fail(IMT_ERR_ARG, "null root_out");
    return path_common(c, leaf, helper_mask, true, nullptr, sib, depth, n, root_out, nullptr, flags);
}
extern "C" int imt_verify_proof_batch(imt_ctx* c, const void* leaf, const uint64_t* index, const void* root,
                                      const void* sib, unsigned depth, size_t n, uint8_t* ok_out, unsigned flags) {
    if (c && n && (!ok_out || !root)) return c->fail(IMT_ERR_ARG, "null root / ok_out");
    return path_common(c, leaf, index, false, root, sib, depth, n, nullptr, ok_out, flags);
}

// ------------------------------------------------------------------------------------
// a13
// ------------------------------------------------------------------------------------
extern "C" int imt_non_membership_batch(imt_ctx* c, const void* root, const void* low_leaf, const uint64_t* low_index,
                                        const void* low_sib, unsigned depth, const void* new_val,
                                        const uint8_t* is_largest, size_t n, uint8_t* fail_out, void* root_out,
                                        unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    if (depth > IMT_MAX_DEPTH) return c->fail(IMT_ERR_RANGE, "depth %u > %d", depth, IMT_MAX_DEPTH);
    if (n == 0) return IMT_OK;
    if (!root || !low_leaf || !low_index || (depth && !low_sib) || !new_val || !is_largest || !fail_out)
        return c->fail(IMT_ERR_ARG, "null buffer");
    Io io(c, flags);
    const unsigned rstride = (flags & IMT_ROOT_PER_ITEM) ? 32 : 0;
    const uint8_t* d_root = io.in_fe(root, rstride ? n * 32 : 32);
    const uint8_t* d_low = io.in_fe(low_leaf, n * 96);
    const uint64_t* d_idx = (const uint64_t*)io.in(low_index, n * 8);
    const uint8_t* d_sib = io.in_fe(low_sib, (size_t)depth * n * 32);
    const uint8_t* d_nv = io.in_fe(new_val, n * 32);
    const uint8_t* d_lg = io.in(is_largest, n);
    uint8_t* d_fail = io.out(fail_out, n);
    uint8_t* d_rout = io.out_fe(root_out, n * 32);
    if (io.rc) return io.rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    launch::non_membership(c->stream, d_root, rstride, d_low, d_idx, d_sib, sib_layout(flags, depth, n), depth, d_nv,
                           d_lg, n, d_fail, d_rout, fmt, fmt, c->d_err, c->coop_max_events);
    return io.finish();
}

extern "C" int imt_split128_batch(imt_ctx* c, const void* vals, void* q, void* r, size_t n, unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    if (n == 0) return IMT_OK;
    if (!vals || !q || !r) return c->fail(IMT_ERR_ARG, "null buffer");
    Io io(c, flags);
    const uint8_t* d_in = io.in_fe(vals, n * 32);
    uint8_t* d_q = io.out_fe(q, n * 32);
    uint8_t* d_r = io.out_fe(r, n * 32);
    if (io.rc) return io.rc;
    launch::split128(c->stream, d_in, d_q, d_r, n, flags & IMT_FMT_MASK, c->d_err);
    return io.finish();
}

// ------------------------------------------------------------------------------------
// a14
// ------------------------------------------------------------------------------------
extern "C" int imt_insert_witness_batch(imt_ctx* c, const void* old_root, const void* low_leaf,
                                        const uint64_t* low_index, const void* low_sib, const void* new_root,
                                        const void* new_leaf, const uint64_t* new_index,
                                        const uint64_t* new_path_index, const void* new_sib,
                                        const uint8_t* is_largest, unsigned depth, size_t n, uint8_t* fail_out,
                                        void* trace_out, unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    if (depth > IMT_MAX_DEPTH) return c->fail(IMT_ERR_RANGE, "depth %u > %d", depth, IMT_MAX_DEPTH);
    if (n == 0) return IMT_OK;
    if (!old_root || !low_leaf || !low_index || !new_root || !new_leaf || !new_index || !is_largest || !fail_out ||
        (depth && (!low_sib || !new_sib)))
        return c->fail(IMT_ERR_ARG, "null buffer");
    Io io(c, flags);
    const uint8_t* d_or = io.in_fe(old_root, n * 32);
    const uint8_t* d_ll = io.in_fe(low_leaf, n * 96);
    const uint64_t* d_li = (const uint64_t*)io.in(low_index, n * 8);
    const uint8_t* d_ls = io.in_fe(low_sib, (size_t)depth * n * 32);
    const uint8_t* d_nr = io.in_fe(new_root, n * 32);
    const uint8_t* d_nl = io.in_fe(new_leaf, n * 96);
    const uint64_t* d_ni = (const uint64_t*)io.in(new_index, n * 8);
    const uint64_t* d_np = new_path_index ? (const uint64_t*)io.in(new_path_index, n * 8) : d_ni;
    const uint8_t* d_ns = io.in_fe(new_sib, (size_t)depth * n * 32);
    const uint8_t* d_lg = io.in(is_largest, n);
    uint8_t* d_fail = io.out(fail_out, n);
    const unsigned fmt = flags & IMT_FMT_MASK;
    uint8_t* d_trace_user = io.out_fe(trace_out, 7 * n * 32);
    uint8_t* d_trace = (d_trace_user && fmt == IMT_FMT_DEVICE) ? d_trace_user : io.temp(7 * n * 32);
    if (io.rc) return io.rc;
    launch::insert_witness(c->stream, d_or, d_ll, d_li, d_ls, d_nr, d_nl, d_ni, d_np, d_ns, sib_layout(flags, depth, n),
                           d_lg, depth, n, d_fail, d_trace, fmt, fmt, c->d_err, c->coop_max_events);
    if (d_trace_user && d_trace_user != d_trace)
        launch::convert(c->stream, d_trace, d_trace_user, 7 * n, IMT_FMT_DEVICE, fmt, c->d_err);
    return io.finish();
}

// ------------------------------------------------------------------------------------
// e: multi-GPU helpers
// ------------------------------------------------------------------------------------
extern "C" int imt_zero_hashes(imt_ctx* c, unsigned depth, void* out, unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    if (depth > IMT_MAX_DEPTH) return c->fail(IMT_ERR_RANGE, "depth %u > %d", depth, IMT_MAX_DEPTH);
    if (!out) return c->fail(IMT_ERR_ARG, "null buffer");
    Io io(c, flags);
    uint8_t* d = io.out_fe(out, (size_t)(depth + 1) * 32);
    if (io.rc) return io.rc;
    launch::convert(c->stream, c->d_zero, d, depth + 1, IMT_FMT_DEVICE, flags & IMT_FMT_MASK, c->d_err);
    return io.finish();
}

extern "C" int imt_combine_subtree_roots(imt_ctx* c, const void* sub_roots, size_t n_roots, unsigned sub_height,
                                         unsigned depth, void* root, unsigned flags) {
    int rc = begin(c, flags);
    if (rc) return rc;
    if (!sub_roots || !root) return c->fail(IMT_ERR_ARG, "null buffer");
    if (n_roots == 0 || (n_roots & (n_roots - 1))) return c->fail(IMT_ERR_ARG, "n_roots must be a power of two");
    unsigned k = 0;
    while (((size_t)1 << k) < n_roots) k++;
    if (depth > IMT_MAX_DEPTH || sub_height + k > depth) return c->fail(IMT_ERR_RANGE, "sub_height + log2(n_roots) > depth");
    Io io(c, flags);
    const uint8_t* d_in = io.in_fe(sub_roots, n_roots * 32);
    uint8_t* d_out = io.out_fe(root, 32);
    uint8_t* a = io.temp(n_roots * 32);
    uint8_t* b = io.temp(n_roots * 32);
    if (io.rc) return io.rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    launch::convert(c->stream, d_in, a, n_roots, fmt, IMT_FMT_DEVICE, c->d_err);
    for (size_t m = n_roots; m > 1; m >>= 1) {
        launch::tree_level(c->stream, a, b, m / 2, c->coop_max_events);
        std::swap(a, b);
    }
    launch::extend_root(c->stream, a, c->d_zero, sub_height + k, depth);
    launch::convert(c->stream, a, d_out, 1, IMT_FMT_DEVICE, fmt, c->d_err);
    return io.finish();
}
