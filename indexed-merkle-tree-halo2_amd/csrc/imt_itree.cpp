// imt_itree.cpp -- the stateful depth-d indexed tree behind imt_itree_* (include/imt.h).
//
// A batch insertion has a hash-free part -- the low-leaf search of update_idx_leaf
// (/root/reference/src/indexed_merkle_tree.rs:632-660), the leaf preimages at every time
// step and the (position, time) order of the 2N leaf events -- and the hashing, which is the
// level sweep of imt_sweep.hpp on the GPU.  The hash-free part runs on the GPU too by default
// (imt_prep.hip, over a device-resident sorted index); IMT_HOST_PREP selects the equivalent host
// code in this file.  Plan buffers exist three times and are filled on a side stream, so with
// IMT_DEVICE_PTRS | IMT_PIPELINE consecutive batches overlap on two compute streams.
#include "imt_ctx.hpp"
#include "imt_prep.hpp"
#include "imt_prep_logic.hpp"
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <new>
#include <numeric>
#include <thread>

using namespace imt;

namespace {

typedef std::array<uint64_t, 4> U256;   // little-endian limbs of a canonical integer

inline bool lt256(const U256& a, const U256& b) {
    for (int i = 3; i >= 0; i--)
        if (a[i] != b[i]) return a[i] < b[i];
    return false;
}
inline bool is_zero256(const U256& a) { return (a[0] | a[1] | a[2] | a[3]) == 0; }

struct Pre {            // {val, next_val, next_idx}: src/utils.rs:12-17
    U256 val, next_val;
    uint64_t next_idx;
};
struct SortedEnt {      // 16 bytes: the value's top limb (ties resolved through pre[idx].val) + leaf index
    uint64_t top;
    uint64_t idx;
};

inline void put_pre(uint8_t* dst, const U256& val, const U256& next_val, uint64_t next_idx) {
    std::memcpy(dst, val.data(), 32);
    std::memcpy(dst + 32, next_val.data(), 32);
    std::memset(dst + 64, 0, 32);
    std::memcpy(dst + 64, &next_idx, 8);
}

// include/imt.h: every field-element row a DEVICE pointer refers to must be 16-byte aligned (the kernels move an
// element as two 16-byte words).  A misaligned one is refused here as an argument error instead of faulting there.
int check_fe_ptrs(imt_ctx* c, bool dev, std::initializer_list<const void*> ps) {
    if (!dev) return IMT_OK;
    for (const void* p : ps)
        if (p && ((uintptr_t)p & 15u)) return c->fail(IMT_ERR_ARG, "device pointer %p to field elements is not 16-byte aligned", p);
    return IMT_OK;
}
int check_out_ptrs(imt_ctx* c, bool dev, const imt_insert_out* o) {
    if (!o) return IMT_OK;
    return check_fe_ptrs(c, dev, {o->low_leaf, o->old_root, o->interim_root, o->new_root, o->new_leaf, o->low_sib, o->new_sib});
}

unsigned ceil_log2(uint64_t x) {
    unsigned l = 0;
    while (l < 63 && ((uint64_t)1 << l) < x) l++;
    return l;
}

// one set of plan buffers (device + pinned host staging)
// Batches in flight on the GPU under IMT_PIPELINE, each on its own stream, one tree level apart.  At 2^16
// insertions per batch two already saturate the SIMDs (2, 4: 3.03 M insertions/s; 8: 2.97).  Four is for smaller
// batches, whose 33 dependent launches cost 0.39 ms each whatever their size: 2^15 goes from 2.73 to 3.01 M/s,
// 2^10..2^13 double (profiles/r02_small_batch_rates.txt).  More needs GPU_MAX_HW_QUEUES > 4 in the environment.
#ifndef IMT_NPIPE
#define IMT_NPIPE 4
#endif

struct PlanSet {
    size_t cap_events = 0;       // capacity in events
    unsigned cap_levels = 0;
    uint8_t* h_pin = nullptr;    // pinned: [pre 96*E][node,time,rs,re 4*E each]
    uint8_t* d_pre = nullptr;
    uint32_t* d_tab[2][4] = {{nullptr}};   // ping-pong {node,time,rs,re}
    uint32_t* d_from = nullptr;  // [levels][E]
    int32_t* d_sibsrc = nullptr;
    uint32_t* d_nodeb = nullptr;
    uint32_t* d_timen = nullptr; // [levels][E] time table of level l+1
    uint8_t* d_val[2] = {nullptr, nullptr};
    prep::Workspace ws;                  // GPU-prepare scratch
    uint32_t* d_slot = nullptr;          // [levels + 1][E] slot of every event per level (sharded mode)
    const uint8_t** d_valptr = nullptr;  // [levels + 1] device array of level value pointers (sharded mode)
    uint8_t* d_root = nullptr;           // stored root right after this batch (device format)
    bool has_root = false;
    hipEvent_t done = nullptr;           // recorded after the batch's last kernel
    hipEvent_t wb_done[IMT_MAX_DEPTH + 1] = {nullptr};   // recorded after k_writeback of each level
    bool in_flight = false;
    bool pipelined = false;              // ran on a pipeline stream (IMT_PIPELINE)
    unsigned l0 = 0;                     // its L0
    uint8_t* d_canon = nullptr;          // [E/2][32] canonical copy of the batch's values when they arrive in another format
    // a time slice of a multi-GPU step (imt_itree_slice_*): prepared here, hashed unit by unit on the caller's streams
    hipEvent_t prep_done = nullptr;      // recorded on the side stream behind the slice's preparation and index phase
    bool open = false;                   // prepared, last unit not yet issued
    bool sliced = false;                 // the set's last user was a slice (imt_itree_slice_prepare), not an ordinary batch
    size_t slice_n = 0;
    imt_insert_out slice_out = {};
    unsigned slice_fmt = 0;
    unsigned slice_next_unit = 0;
    uint64_t slice_size_before = 0;      // leaves in the tree before this slice (its own first new leaf)
    launch::SibLayout slice_lay = {0, 0};
};

}  // namespace

struct imt_itree {
    imt_ctx* ctx = nullptr;
    unsigned depth = 0;
    uint64_t cap = 0, size = 0;
    // placement as a subtree of a deeper tree (imt_itree_set_placement); the defaults are "not placed"
    unsigned global_depth = 0;       // = depth when not placed
    uint64_t sub_index = 0;          // which subtree of height `depth`
    uint64_t index_base = 0;         // sub_index << depth: added to every leaf index that crosses the API
    uint32_t part_mod = 0, part_res = 0;   // value partition between the subtrees (imt_itree_set_value_partition)
    hipEvent_t in_mark = nullptr;    // position of the context's stream when a device-pointer call starts
    std::vector<uint64_t> h_off, h_len;
    uint8_t* d_nodes = nullptr;
    uint64_t* d_off = nullptr;
    uint64_t* d_len = nullptr;
    std::vector<Pre> pre;            // host mirror of the leaf preimages
    std::vector<SortedEnt> sorted;   // leaves ordered by val
    // device-resident index (default prepare path): values in leaf order + leaf indices in value order.
    // The host mirror and the device index are each refreshed from the other on demand.
    uint8_t* d_val = nullptr;        // [cap][32] canonical
    uint32_t* d_sorted[2] = {nullptr, nullptr};
    int sorted_cur = 0;
    bool mirror_valid = true, dev_index_valid = true;
    int* h_err_pin = nullptr;        // pinned word for the prepare kernels' error bits
    // a sharded batch between imt_itree_batch_begin and _end
    struct Pending { bool active = false; size_t n = 0; unsigned l0 = 0; int set = 0; } pending;
    static constexpr int NPIPE = IMT_NPIPE;     // batches in flight on the GPU under IMT_PIPELINE
    static constexpr int NSETS = NPIPE + 1;     // host work may run one batch further ahead
    PlanSet plan[NSETS];
    int cur = 0;
    uint64_t batch_no = 0;
    hipStream_t up_stream = nullptr;
    hipEvent_t up_done = nullptr;
    // IMT_PIPELINE: consecutive batches rotate over NPIPE internal streams and run one level
    // apart (batch k+1 sweeps level l once batch k has written level l back), so up to NPIPE hash
    // kernels share the GPU and the SIMDs see several times the waves of a single 2^16 batch.
    hipStream_t pipe_stream[NPIPE] = {};
    hipEvent_t user_mark = nullptr;  // position of the context's stream when a pipelined call starts
    bool pipe_pending = false;       // pipelined work the context's stream has not been ordered behind
    // reusable host work arrays of insert_batch
    std::vector<uint32_t> w_ord, w_rank;
    std::vector<std::pair<uint64_t, uint32_t>> w_sortkey;
    std::vector<int64_t> w_prv, w_nxt, w_pred, w_succ;
    std::vector<uint64_t> w_keys, w_keys2, w_low;
    std::vector<uint8_t> w_largest;
    std::vector<uint32_t> w_hist;
    std::vector<SortedEnt> w_merged;
    // imt_itree_slice_prepare: index workspace for the values other GPUs hash, canonical copy of a step's values
    prep::Workspace fws;
    uint8_t* d_canon_all = nullptr;
    size_t canon_all_cap = 0;
    uint32_t* d_sorted_extra = nullptr;      // third index buffer: a step's up to three merges never write the committed one
    hipStream_t slice_prep_stream = nullptr; // where the next imt_itree_slice_prepare runs (nullptr: the side stream)
    const uint32_t* slice_poison = nullptr;  // device-visible word of the world's transport: non-zero = skip applies
    double slice_wait_limit_ms = 0;          // > 0: host waits inside imt_itree_slice_prepare give up after this long
    hipEvent_t slice_tail_event = nullptr;   // one shot (imt_itree_set_slice_tail_event)
    bool slice_tail_attached = false;
    double slice_wait_ms = 0;                // host time spent waiting for the GPU inside imt_itree_slice_prepare
    double slice_backpressure_ms = 0;        // ... the part of it spent waiting for the plan set's previous slice (all-time total)
    bool sliced_busy = false;                // an imt_sliced world has steps in flight on this replica (until its flush)
    size_t reserved_events = 0;              // every plan set holds at least this many events (reserve_all_plans)
};

static void plan_free(PlanSet& p) {
    if (p.h_pin) hipHostFree(p.h_pin);
    if (p.d_pre) hipFree(p.d_pre);
    for (auto& pp : p.d_tab)
        for (auto& q : pp)
            if (q) hipFree(q);
    if (p.d_from) hipFree(p.d_from);
    if (p.d_sibsrc) hipFree(p.d_sibsrc);
    if (p.d_nodeb) hipFree(p.d_nodeb);
    if (p.d_timen) hipFree(p.d_timen);
    for (auto& q : p.d_val)
        if (q) hipFree(q);
    if (p.d_root) hipFree(p.d_root);
    if (p.d_slot) hipFree(p.d_slot);
    if (p.d_valptr) hipFree(p.d_valptr);
    if (p.d_canon) hipFree(p.d_canon);
    for (void* q : {(void*)p.ws.o_low, (void*)p.ws.o_largest, (void*)p.ws.o_lowleaf, (void*)p.ws.o_newleaf})
        if (q) hipFree(q);
    for (void* q : {(void*)p.ws.iota, (void*)p.ws.bsorted, (void*)p.ws.gap, (void*)p.ws.st, (void*)p.ws.low,
                    (void*)p.ws.succ, (void*)p.ws.keys, (void*)p.ws.keys_sorted, p.ws.tmp, (void*)p.ws.err})
        if (q) hipFree(q);
    PlanSet keep;
    keep.done = p.done;
    keep.prep_done = p.prep_done;
    for (int l = 0; l <= IMT_MAX_DEPTH; l++) keep.wb_done[l] = p.wb_done[l];
    p = keep;
}

static int plan_reserve(imt_ctx* c, PlanSet& p, size_t events, unsigned levels, size_t tree_cap) {
    if (p.cap_events >= events && p.cap_levels >= levels) return IMT_OK;
    plan_free(p);
    const size_t E = std::max(events + events / 4, (size_t)1024);
    const unsigned L = std::max(levels, 1u);
    hipError_t e = hipSuccess;
    auto A = [&](void** ptr, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(ptr, bytes);
    };
    if (hipHostMalloc((void**)&p.h_pin, E * (96 + 16), hipHostMallocDefault) != hipSuccess)
        return c->fail(IMT_ERR_ALLOC, "hipHostMalloc(plan staging) failed");
    A((void**)&p.d_pre, E * 96);
    for (int s = 0; s < 2; s++)
        for (int j = 0; j < 4; j++) A((void**)&p.d_tab[s][j], E * 4);
    A((void**)&p.d_from, (size_t)L * E * 4);
    A((void**)&p.d_sibsrc, (size_t)L * E * 4);
    A((void**)&p.d_nodeb, (size_t)L * E * 4);
    A((void**)&p.d_timen, (size_t)L * E * 4);
    A((void**)&p.d_val[0], E * 32);
    A((void**)&p.d_val[1], E * 32);
    A((void**)&p.d_root, 32);
    {   // GPU-prepare workspace for E/2 insertions
        const size_t N = E / 2;
        int lv = 1;
        while (((size_t)1 << lv) <= N) lv++;
        p.ws.cap_n = N;
        A((void**)&p.ws.iota, N * 4);
        A((void**)&p.ws.bsorted, N * 4);
        A((void**)&p.ws.gap, N * 4);
        A((void**)&p.ws.st, (size_t)lv * N * 4);
        A((void**)&p.ws.low, N * 4);
        A((void**)&p.ws.succ, N * 4);
        A((void**)&p.ws.keys, E * 8);
        A((void**)&p.ws.keys_sorted, E * 8);
        p.ws.tmp_bytes = prep::temp_bytes_needed(N, tree_cap);
        A((void**)&p.ws.tmp, p.ws.tmp_bytes);
        A((void**)&p.ws.err, sizeof(int));
        A((void**)&p.ws.o_low, N * 8);
        A((void**)&p.ws.o_largest, N);
        A((void**)&p.ws.o_lowleaf, N * 96);
        A((void**)&p.ws.o_newleaf, N * 96);
    }
    A((void**)&p.d_slot, (size_t)(L + 1) * E * 4);
    A((void**)&p.d_valptr, (size_t)(L + 1) * sizeof(void*));
    A((void**)&p.d_canon, (E / 2) * 32);
    if (e != hipSuccess) {
        plan_free(p);
        return c->hip_fail(e, "hipMalloc(plan)");
    }
    p.cap_events = E;
    p.cap_levels = L;
    return IMT_OK;
}

// All plan sets at once: a pipelined caller (IMT_PIPELINE, imt_sliced_*) rotates through every set within its first
// NSETS batches, and each first use is a dozen hipMalloc calls in the middle of somebody's pipeline.
static int reserve_all_plans(imt_itree* t, size_t events) {
    if (t->reserved_events >= events) return IMT_OK;
    for (auto& p : t->plan) {
        if (p.in_flight || p.open) continue;              // its buffers are in use: it grows when its turn comes
        int rc = plan_reserve(t->ctx, p, events, t->depth, t->cap);
        if (rc) return rc;
    }
    t->reserved_events = events;
    return IMT_OK;
}

extern "C" void imt_itree_free(imt_itree* t) {
    if (!t) return;
    hipSetDevice(t->ctx->device);
    hipStreamSynchronize(t->ctx->stream);
    if (t->up_stream) hipStreamSynchronize(t->up_stream);
    for (auto& p : t->plan) {
        plan_free(p);
        if (p.done) hipEventDestroy(p.done);
        if (p.prep_done) hipEventDestroy(p.prep_done);
        for (auto& e : p.wb_done)
            if (e) hipEventDestroy(e);
    }
    if (t->up_done) hipEventDestroy(t->up_done);
    if (t->up_stream) hipStreamDestroy(t->up_stream);
    for (auto& ps : t->pipe_stream) {
        if (!ps) continue;
        hipStreamSynchronize(ps);
        auto& ss = t->ctx->side_streams;
        ss.erase(std::remove(ss.begin(), ss.end(), ps), ss.end());
        hipStreamDestroy(ps);
    }
    if (t->user_mark) hipEventDestroy(t->user_mark);
    if (t->in_mark) hipEventDestroy(t->in_mark);
    if (t->d_nodes) hipFree(t->d_nodes);
    if (t->d_off) hipFree(t->d_off);
    if (t->d_len) hipFree(t->d_len);
    if (t->d_val) hipFree(t->d_val);
    for (auto q : t->d_sorted)
        if (q) hipFree(q);
    if (t->h_err_pin) hipHostFree(t->h_err_pin);
    for (void* q : {(void*)t->fws.iota, (void*)t->fws.bsorted, (void*)t->fws.gap, (void*)t->fws.st, t->fws.tmp,
                    (void*)t->d_canon_all, (void*)t->d_sorted_extra})
        if (q) hipFree(q);
    delete t;
}

extern "C" int imt_itree_new(imt_ctx* c, unsigned depth, uint64_t capacity, imt_itree** out) {
    if (!c || !out) return IMT_ERR_ARG;
    *out = nullptr;
    if (depth == 0 || depth > IMT_MAX_DEPTH) return c->fail(IMT_ERR_RANGE, "depth %u out of range", depth);
    if (capacity < 2 || (capacity & (capacity - 1)) || capacity > ((uint64_t)1 << 31))
        return c->fail(IMT_ERR_RANGE, "capacity must be a power of two in [2, 2^31]");
    if (depth < 63 && capacity > ((uint64_t)1 << depth)) return c->fail(IMT_ERR_RANGE, "capacity exceeds 2^depth");
    int rc = c->set_device();
    if (rc) return rc;
    imt_itree* t = new (std::nothrow) imt_itree();
    if (!t) return c->fail(IMT_ERR_ALLOC, "out of host memory");
    t->ctx = c;
    t->depth = depth;
    t->global_depth = depth;
    t->cap = capacity;
    uint64_t off = 0;
    for (unsigned l = 0; l <= depth; l++) {
        uint64_t n = l < 63 ? capacity >> l : 0;
        if (n == 0) n = 1;
        t->h_off.push_back(off);
        t->h_len.push_back(n);
        off += n;
    }
    hipError_t e;
    if ((e = hipMalloc((void**)&t->d_val, capacity * 32)) != hipSuccess ||
        (e = hipMalloc((void**)&t->d_sorted[0], capacity * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&t->d_sorted[1], capacity * 4)) != hipSuccess ||
        (e = hipHostMalloc((void**)&t->h_err_pin, sizeof(int), hipHostMallocDefault)) != hipSuccess ||
        (e = hipMalloc((void**)&t->d_nodes, off * 32)) != hipSuccess ||
        (e = hipMalloc((void**)&t->d_off, (depth + 1) * 8)) != hipSuccess ||
        (e = hipMalloc((void**)&t->d_len, (depth + 1) * 8)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&t->up_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&t->up_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&t->in_mark, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&t->user_mark, hipEventDisableTiming)) != hipSuccess) {
        imt_itree_free(t);
        return c->hip_fail(e, "imt_itree_new allocation");
    }
    {
        // different priorities: the runtime may otherwise map the streams to one hardware queue, which
        // serialises them (seen with rocprofv3: same Queue_Id, zero overlap)
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;   // numerically: least >= greatest
        for (int i = 0; i < imt_itree::NPIPE; i++) {
            const int prio = std::max(greatest, std::min(least, 0 - i));
            if ((e = hipStreamCreateWithPriority(&t->pipe_stream[i], hipStreamNonBlocking, prio)) != hipSuccess) {
                imt_itree_free(t);
                return c->hip_fail(e, "hipStreamCreateWithPriority");
            }
        }
        for (auto& pl : t->plan)
            if ((e = hipEventCreateWithFlags(&pl.done, hipEventDisableTiming)) != hipSuccess ||
                (e = hipEventCreateWithFlags(&pl.prep_done, hipEventDisableTiming)) != hipSuccess) {
                imt_itree_free(t);
                return c->hip_fail(e, "hipEventCreate");
            }
    }
    for (auto& pl : t->plan)
        for (unsigned l = 0; l <= depth; l++)
            if ((e = hipEventCreateWithFlags(&pl.wb_done[l], hipEventDisableTiming)) != hipSuccess) {
                imt_itree_free(t);
                return c->hip_fail(e, "hipEventCreate");
            }
    if ((e = hipMemsetAsync(t->d_val, 0, 32, c->stream)) != hipSuccess ||          // leaf 0: the sentinel value 0
        (e = hipMemsetAsync(t->d_sorted[0], 0, 4, c->stream)) != hipSuccess ||     // sorted index = [leaf 0]
        (e = hipMemcpyAsync(t->d_off, t->h_off.data(), (depth + 1) * 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(t->d_len, t->h_len.data(), (depth + 1) * 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess) {
        imt_itree_free(t);
        return c->hip_fail(e, "imt_itree_new init copies");
    }
    for (unsigned l = 0; l <= depth; l++)   // every stored node starts as the empty subtree of its height
        launch::fill_level(c->stream, t->d_nodes + t->h_off[l] * 32, t->h_len[l], c->d_zero + (size_t)l * 32);
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) {
        imt_itree_free(t);
        return c->hip_fail(e, "imt_itree_new init");
    }
    // leaf 0 is the {0,0,0} sentinel; its hash equals the empty-slot hash
    t->pre.push_back(Pre{{0, 0, 0, 0}, {0, 0, 0, 0}, 0});
    t->sorted.push_back(SortedEnt{0, 0});
    t->size = 1;
    for (auto ps : t->pipe_stream) c->side_streams.push_back(ps);
    *out = t;
    return IMT_OK;
}

// a slice between imt_itree_slice_prepare and its last unit owns its plan set and has moved the index ahead of the
// stored tree: the batch calls wait until it is finished
static bool slice_open(const imt_itree* t) {
    for (const auto& pl : t->plan)
        if (pl.open) return true;
    return false;
}

// Between imt_sliced_step and imt_sliced_flush the replica is mid-step: other ranks' write-backs are still arriving on
// the world's own streams, which nothing here is ordered behind.  The ordinary entry points refuse instead of reading
// or writing a half-applied tree (include/imt.h: "only imt_sliced_* calls may touch the trees").
#define IMT_NOT_SLICED(t)                                                                                              \
    do {                                                                                                               \
        if ((t)->sliced_busy) return (t)->ctx->fail(IMT_ERR_ARG, "the tree has sliced steps in flight: imt_sliced_flush first"); \
    } while (0)

// order the context's stream behind every pipelined batch still in flight
static int join_top(imt_itree* t) {
    if (!t->pipe_pending) return IMT_OK;
    for (auto& pl : t->plan)
        if (pl.in_flight && pl.pipelined) IMT_HIP(t->ctx, hipStreamWaitEvent(t->ctx->stream, pl.done, 0));
    t->pipe_pending = false;
    return IMT_OK;
}

// host mirror <- device index (after GPU-prepared batches)
static int ensure_mirror(imt_itree* t) {
    if (t->mirror_valid) return IMT_OK;
    imt_ctx* c = t->ctx;
    const size_t M = t->size;
    std::vector<U256> vals(M);
    std::vector<uint32_t> order(M);
    IMT_HIP(c, hipStreamSynchronize(t->up_stream));
    IMT_HIP(c, hipMemcpy(vals.data(), t->d_val, M * 32, hipMemcpyDeviceToHost));
    IMT_HIP(c, hipMemcpy(order.data(), t->d_sorted[t->sorted_cur], M * 4, hipMemcpyDeviceToHost));
    static const U256 ZERO = {0, 0, 0, 0};
    t->pre.resize(M);
    t->sorted.resize(M);
    for (size_t r = 0; r < M; r++) {
        const uint32_t i = order[r];
        t->sorted[r] = SortedEnt{vals[i][3], i};
        t->pre[i].val = vals[i];
        t->pre[i].next_val = r + 1 < M ? vals[order[r + 1]] : ZERO;
        t->pre[i].next_idx = r + 1 < M ? order[r + 1] : 0;
    }
    t->mirror_valid = true;
    return IMT_OK;
}
// device index <- host mirror (after host-prepared batches or a snapshot load)
static int ensure_device_index(imt_itree* t) {
    if (t->dev_index_valid) return IMT_OK;
    imt_ctx* c = t->ctx;
    int rc = ensure_mirror(t);
    if (rc) return rc;
    const size_t M = t->size;
    std::vector<U256> vals(M);
    std::vector<uint32_t> order(M);
    for (size_t i = 0; i < M; i++) vals[i] = t->pre[i].val;
    for (size_t r = 0; r < M; r++) order[r] = (uint32_t)t->sorted[r].idx;
    IMT_HIP(c, hipStreamSynchronize(t->up_stream));
    IMT_HIP(c, hipMemcpy(t->d_val, vals.data(), M * 32, hipMemcpyHostToDevice));
    IMT_HIP(c, hipMemcpy(t->d_sorted[t->sorted_cur], order.data(), M * 4, hipMemcpyHostToDevice));
    t->dev_index_valid = true;
    return IMT_OK;
}

extern "C" uint64_t imt_itree_size(const imt_itree* t) { return t ? t->size : 0; }
// internal (imt_itree_internal.hpp): what imt_sliced.cpp needs to know about a tree
imt_ctx* imt_itree_ctx(const imt_itree* t) { return t ? t->ctx : nullptr; }
unsigned imt_itree_depth(const imt_itree* t) { return t ? t->depth : 0; }
void imt_itree_mark_sliced(imt_itree* t, bool busy) {
    if (t) t->sliced_busy = busy;
}
double imt_itree_slice_backpressure_ms(const imt_itree* t) { return t ? t->slice_backpressure_ms : 0; }
void imt_itree_set_slice_tail_event(imt_itree* t, void* hip_event) {
    if (t) {
        t->slice_tail_event = (hipEvent_t)hip_event;
        t->slice_tail_attached = false;
    }
}
bool imt_itree_take_slice_tail_attached(imt_itree* t) {
    if (!t) return false;
    const bool a = t->slice_tail_attached;
    t->slice_tail_attached = false;
    t->slice_tail_event = nullptr;
    return a;
}
void imt_itree_set_slice_poison(imt_itree* t, const uint32_t* device_word) {
    if (t) t->slice_poison = device_word;
}
void imt_itree_set_slice_wait_limit(imt_itree* t, double ms) {
    if (t) t->slice_wait_limit_ms = ms;
}
void imt_itree_set_slice_prep_stream(imt_itree* t, void* hip_stream) {
    if (t) t->slice_prep_stream = (hipStream_t)hip_stream;
}
double imt_itree_take_wait_ms(imt_itree* t) {
    const double w = t ? t->slice_wait_ms : 0;
    if (t) t->slice_wait_ms = 0;
    return w;
}
bool imt_itree_is_plain(const imt_itree* t) { return t && !t->index_base && t->part_mod <= 1 && !t->pending.active; }

extern "C" int imt_itree_root(imt_itree* t, void* root, unsigned flags) {
    if (!t || !root) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    IMT_NOT_SLICED(t);
    int rc = c->set_device();
    if (rc) return rc;
    if ((rc = check_fe_ptrs(c, flags & IMT_DEVICE_PTRS, {root}))) return rc;
    if ((rc = join_top(t))) return rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    const uint8_t* src = t->d_nodes + t->h_off[t->depth] * 32;
    if (flags & IMT_DEVICE_PTRS) {
        launch::convert(c->stream, src, (uint8_t*)root, 1, IMT_FMT_DEVICE, fmt, c->d_err);
        return IMT_OK;
    }
    uint8_t* d = (uint8_t*)c->dev_scratch(0, 32);
    if (!d) return IMT_ERR_HIP;
    launch::convert(c->stream, src, d, 1, IMT_FMT_DEVICE, fmt, c->d_err);
    IMT_HIP(c, hipMemcpyAsync(root, d, 32, hipMemcpyDeviceToHost, c->stream));
    IMT_HIP(c, hipStreamSynchronize(c->stream));
    return IMT_OK;
}

extern "C" int imt_itree_root_lagged(imt_itree* t, unsigned lag, void* root, unsigned flags) {
    if (!t || !root) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    IMT_NOT_SLICED(t);
    if (lag > 1) return c->fail(IMT_ERR_RANGE, "lag must be 0 or 1");
    int rc = c->set_device();
    if (rc) return rc;
    if (t->batch_no <= lag) return c->fail(IMT_ERR_RANGE, "no batch %u calls ago", lag);
    if ((rc = check_fe_ptrs(c, flags & IMT_DEVICE_PTRS, {root}))) return rc;
    const PlanSet& P = t->plan[(t->cur + 2 * imt_itree::NSETS - 1 - (int)lag) % imt_itree::NSETS];
    if (!P.has_root)
        return c->fail(P.sliced ? IMT_ERR_ARG : IMT_ERR_INTERNAL,
                       P.sliced ? "imt_itree_root_lagged does not see slices: imt_sliced_flush, then imt_itree_root" : "batch root not recorded");
    IMT_HIP(c, hipStreamWaitEvent(c->stream, P.done, 0));
    const unsigned fmt = flags & IMT_FMT_MASK;
    if (flags & IMT_DEVICE_PTRS) {
        launch::convert(c->stream, P.d_root, (uint8_t*)root, 1, IMT_FMT_DEVICE, fmt, c->d_err);
        return IMT_OK;
    }
    uint8_t* d = (uint8_t*)c->dev_scratch(0, 32);
    if (!d) return IMT_ERR_HIP;
    launch::convert(c->stream, P.d_root, d, 1, IMT_FMT_DEVICE, fmt, c->d_err);
    IMT_HIP(c, hipMemcpyAsync(root, d, 32, hipMemcpyDeviceToHost, c->stream));
    IMT_HIP(c, hipStreamSynchronize(c->stream));
    return IMT_OK;
}

// canonical host copy of n field elements given in any format / location
static int fetch_canonical(imt_ctx* c, hipStream_t st, const void* vals, size_t n, unsigned flags, std::vector<U256>& out) {
    out.resize(n);
    const unsigned fmt = flags & IMT_FMT_MASK;
    const bool dev = flags & IMT_DEVICE_PTRS;
    if (!dev && fmt == IMT_FMT_CANONICAL) {
        std::memcpy(out.data(), vals, n * 32);
    } else if (dev && fmt == IMT_FMT_CANONICAL) {
        IMT_HIP(c, hipMemcpyAsync(out.data(), vals, n * 32, hipMemcpyDeviceToHost, st));
        IMT_HIP(c, hipStreamSynchronize(st));
    } else {
        // convert on the device, then read back
        uint8_t* d_in = (uint8_t*)vals;
        if (!dev) {
            d_in = (uint8_t*)c->dev_scratch(0, n * 32);
            if (!d_in) return IMT_ERR_HIP;
            IMT_HIP(c, hipMemcpyAsync(d_in, vals, n * 32, hipMemcpyHostToDevice, st));
        }
        uint8_t* d_out = (uint8_t*)c->dev_scratch(1, n * 32);
        if (!d_out) return IMT_ERR_HIP;
        launch::convert(st, d_in, d_out, n, fmt, IMT_FMT_CANONICAL, c->d_err);
        IMT_HIP(c, hipMemcpyAsync(out.data(), d_out, n * 32, hipMemcpyDeviceToHost, st));
        IMT_HIP(c, hipStreamSynchronize(st));
    }
    for (size_t i = 0; i < n; i++)
        if (HField::geq_p(out[i].data())) return c->fail(IMT_ERR_NONCANONICAL, "value %zu is not reduced (>= p)", i);
    return IMT_OK;
}

// stored entry e compared with value v: -1 / 0 / +1
static inline int cmp_ent(const imt_itree* t, const SortedEnt& e, const U256& v) {
    if (e.top != v[3]) return e.top < v[3] ? -1 : 1;
    const U256& x = t->pre[e.idx].val;
    for (int i = 2; i >= 0; i--)
        if (x[i] != v[i]) return x[i] < v[i] ? -1 : 1;
    return 0;
}

// position in t->sorted of the greatest val < v; IMT_ERR_VALUE if v == 0 or present
static int find_pred(const imt_itree* t, const U256& v, size_t& pos) {
    size_t lo = 0, hi = t->sorted.size();   // first entry with val >= v
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (cmp_ent(t, t->sorted[mid], v) < 0) lo = mid + 1; else hi = mid;
    }
    if (lo < t->sorted.size() && cmp_ent(t, t->sorted[lo], v) == 0) return IMT_ERR_VALUE;
    if (lo == 0) return IMT_ERR_VALUE;
    pos = lo - 1;
    return IMT_OK;
}

extern "C" int imt_itree_find_low_batch(imt_itree* t, const void* vals, size_t n, uint64_t* low_index, unsigned flags) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    IMT_NOT_SLICED(t);
    if (n == 0) return IMT_OK;
    if (!vals || !low_index) return c->fail(IMT_ERR_ARG, "null buffer");
    int rc = c->set_device();
    if (rc) return rc;
    if ((rc = check_fe_ptrs(c, flags & IMT_DEVICE_PTRS, {vals}))) return rc;
    if ((flags & IMT_FMT_MASK) == 3) return c->fail(IMT_ERR_ARG, "unknown field-element format");
    if ((flags & IMT_DEVICE_PTRS) || !t->mirror_valid) {
        // predecessor search in the device-resident index (k_find_low); the host mirror is not built for it
        const bool dev = flags & IMT_DEVICE_PTRS;
        const unsigned fmt = flags & IMT_FMT_MASK;
        if (n > ((size_t)1 << 31)) return c->fail(IMT_ERR_RANGE, "batch too large");
        if (dev && ((uintptr_t)low_index & 7u)) return c->fail(IMT_ERR_ARG, "device low_index array is not 8-byte aligned");
        if ((rc = ensure_device_index(t))) return rc;
        if ((rc = join_top(t))) return rc;
        hipStream_t s = c->stream;
        IMT_HIP(c, hipStreamSynchronize(t->up_stream));
        const uint8_t* d_vals = (const uint8_t*)vals;
        int* d_perr = (int*)c->dev_scratch(2, sizeof(int));
        if (!d_perr) return IMT_ERR_HIP;
        IMT_HIP(c, hipMemsetAsync(d_perr, 0, sizeof(int), s));
        if (!dev || fmt != IMT_FMT_CANONICAL) {
            uint8_t* buf = (uint8_t*)c->dev_scratch(0, n * 32);
            if (!buf) return IMT_ERR_HIP;
            if (!dev) IMT_HIP(c, hipMemcpyAsync(buf, vals, n * 32, hipMemcpyHostToDevice, s));
            if (fmt != IMT_FMT_CANONICAL) launch::convert(s, dev ? d_vals : buf, buf, n, fmt, IMT_FMT_CANONICAL, d_perr);
            d_vals = buf;
        }
        uint64_t* d_low = dev ? low_index : (uint64_t*)c->dev_scratch(1, n * 8);
        if (!d_low) return IMT_ERR_HIP;
        prep::find_low(s, d_vals, t->d_val, t->d_sorted[t->sorted_cur], (uint32_t)t->size, (uint32_t)n, t->index_base,
                       t->part_mod, t->part_res, d_low, d_perr);
        int perr = 0;
        IMT_HIP(c, hipMemcpyAsync(&perr, d_perr, sizeof(int), hipMemcpyDeviceToHost, s));
        if (!dev) IMT_HIP(c, hipMemcpyAsync(low_index, d_low, n * 8, hipMemcpyDeviceToHost, s));
        IMT_HIP(c, hipStreamSynchronize(s));
        if (perr & prep::ERR_NONCANONICAL) return c->fail(IMT_ERR_NONCANONICAL, "a value is not reduced (>= p)");
        if (perr & prep::ERR_FOREIGN)
            return c->fail(IMT_ERR_VALUE, "a value belongs to another subtree (v %% %u != %u)", t->part_mod, t->part_res);
        if (perr & (prep::ERR_ZERO | prep::ERR_DUPLICATE)) return c->fail(IMT_ERR_VALUE, "a value is zero or already in the tree");
        return IMT_OK;
    }
    std::vector<U256> v;
    rc = fetch_canonical(c, c->stream, vals, n, flags, v);
    if (rc) return rc;
    std::vector<uint64_t> res(n);
    for (size_t i = 0; i < n; i++) {
        size_t pos;
        if (t->part_mod > 1 && prep::mod_small(reinterpret_cast<const uint8_t*>(v[i].data()), t->part_mod) != t->part_res)
            return c->fail(IMT_ERR_VALUE, "value %zu belongs to another subtree (v %% %u != %u)", i, t->part_mod, t->part_res);
        if (find_pred(t, v[i], pos)) return c->fail(IMT_ERR_VALUE, "value %zu is zero or already in the tree", i);
        res[i] = t->index_base + t->sorted[pos].idx;
    }
    if (flags & IMT_DEVICE_PTRS) {
        IMT_HIP(c, hipMemcpy(low_index, res.data(), n * 8, hipMemcpyHostToDevice));
    } else {
        std::memcpy(low_index, res.data(), n * 8);
    }
    return IMT_OK;
}

extern "C" int imt_itree_get_leaves(imt_itree* t, const uint64_t* index, size_t n, void* preimage, unsigned flags) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    IMT_NOT_SLICED(t);
    if (n == 0) return IMT_OK;
    if (!preimage) return c->fail(IMT_ERR_ARG, "null buffer");
    if ((flags & IMT_FMT_MASK) == 3) return c->fail(IMT_ERR_ARG, "unknown field-element format");
    if (n > ((size_t)1 << 32) - 1) return c->fail(IMT_ERR_RANGE, "too many leaves in one call");
    int rc = c->set_device();
    if (rc) return rc;
    const bool dev = flags & IMT_DEVICE_PTRS;
    const unsigned fmt = flags & IMT_FMT_MASK;
    if ((rc = check_fe_ptrs(c, dev, {preimage}))) return rc;
    if (dev && index && ((uintptr_t)index & 7u)) return c->fail(IMT_ERR_ARG, "device index array is not 8-byte aligned");
    if (!dev && t->mirror_valid) {
        // the host mirror is current (host-prepared batches): answer from it
        std::vector<uint8_t> buf(n * 96);
        for (size_t i = 0; i < n; i++) {
            const uint64_t li = (index ? index[i] : t->index_base + i) - t->index_base;   // wraps below the base: caught below
            if (li >= t->cap) return c->fail(IMT_ERR_RANGE, "leaf index out of range");
            if (li < t->size) {
                const Pre& p = t->pre[li];                      // the mirror holds local indices; 0 = "no successor"
                put_pre(&buf[i * 96], p.val, p.next_val, is_zero256(p.next_val) ? 0 : t->index_base + p.next_idx);
            } else {
                std::memset(&buf[i * 96], 0, 96);
            }
        }
        if (fmt == IMT_FMT_CANONICAL) {
            std::memcpy(preimage, buf.data(), n * 96);
            return IMT_OK;
        }
        uint8_t* d_in = (uint8_t*)c->dev_scratch(0, n * 96);
        uint8_t* d_out = (uint8_t*)c->dev_scratch(1, n * 96);
        if (!d_in || !d_out) return IMT_ERR_HIP;
        IMT_HIP(c, hipMemcpyAsync(d_in, buf.data(), n * 96, hipMemcpyHostToDevice, c->stream));
        launch::convert(c->stream, d_in, d_out, n * 3, IMT_FMT_CANONICAL, fmt, c->d_err);
        IMT_HIP(c, hipMemcpyAsync(preimage, d_out, n * 96, hipMemcpyDeviceToHost, c->stream));
        IMT_HIP(c, hipStreamSynchronize(c->stream));
        return IMT_OK;
    }
    // from the device index: a leaf's successor is the next value in value order (k_leaves)
    if ((rc = ensure_device_index(t))) return rc;
    if ((rc = join_top(t))) return rc;
    hipStream_t s = c->stream;
    IMT_HIP(c, hipStreamSynchronize(t->up_stream));     // the index may have been written on the side stream
    const uint64_t* d_idx = index;
    if (index && !dev) {
        uint64_t* up = (uint64_t*)c->dev_scratch(0, n * 8);
        if (!up) return IMT_ERR_HIP;
        IMT_HIP(c, hipMemcpyAsync(up, index, n * 8, hipMemcpyHostToDevice, s));
        d_idx = up;
    }
    uint8_t* d_out = dev ? (uint8_t*)preimage : (uint8_t*)c->dev_scratch(1, n * 96);
    int* d_perr = (int*)c->dev_scratch(2, sizeof(int));
    if (!d_out || !d_perr) return IMT_ERR_HIP;
    IMT_HIP(c, hipMemsetAsync(d_perr, 0, sizeof(int), s));
    prep::leaves(s, d_idx, t->index_base, (uint32_t)n, t->d_val, t->d_sorted[t->sorted_cur], (uint32_t)t->size, t->cap,
                 t->index_base, d_out, d_perr);
    if (fmt != IMT_FMT_CANONICAL) launch::convert(s, d_out, d_out, n * 3, IMT_FMT_CANONICAL, fmt, c->d_err);
    int perr = 0;
    IMT_HIP(c, hipMemcpyAsync(&perr, d_perr, sizeof(int), hipMemcpyDeviceToHost, s));
    if (!dev) IMT_HIP(c, hipMemcpyAsync(preimage, d_out, n * 96, hipMemcpyDeviceToHost, s));
    IMT_HIP(c, hipStreamSynchronize(s));
    c->trim_scratch(0, (size_t)64 << 20);               // a whole-tree snapshot to the host staged 104 bytes per leaf here
    c->trim_scratch(1, (size_t)64 << 20);
    if (perr & prep::ERR_RANGE) return c->fail(IMT_ERR_RANGE, "leaf index out of range");
    return IMT_OK;
}

extern "C" int imt_itree_get_proof_batch(imt_itree* t, const uint64_t* index, size_t n, void* sib, unsigned flags) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    IMT_NOT_SLICED(t);
    if (n == 0) return IMT_OK;
    if (!index || !sib) return c->fail(IMT_ERR_ARG, "null buffer");
    int rc = c->set_device();
    if (rc) return rc;
    const bool dev = flags & IMT_DEVICE_PTRS;
    if ((rc = check_fe_ptrs(c, dev, {sib}))) return rc;
    if (!dev)
        for (size_t i = 0; i < n; i++)
            if (index[i] - t->index_base >= t->cap) return c->fail(IMT_ERR_RANGE, "leaf index out of range");
    if ((rc = join_top(t))) return rc;
    const unsigned depth = t->depth;
    const uint64_t* d_idx = index;
    uint8_t* d_out = (uint8_t*)sib;
    if (!dev) {
        d_idx = (const uint64_t*)c->dev_scratch(0, n * 8);
        d_out = (uint8_t*)c->dev_scratch(1, (size_t)depth * n * 32);
        if (!d_idx || !d_out) return IMT_ERR_HIP;
        IMT_HIP(c, hipMemcpyAsync((void*)d_idx, index, n * 8, hipMemcpyHostToDevice, c->stream));
    }
    launch::TreeView tv{t->d_nodes, t->d_off, t->d_len, c->d_zero, t->index_base};
    launch::SibLayout lay = (flags & IMT_SIB_ITEM_MAJOR) ? launch::SibLayout{1, depth} : launch::SibLayout{n, 1};
    launch::gather_proof(c->stream, tv, d_idx, n, depth, d_out, lay, flags & IMT_FMT_MASK);
    if (!dev) {
        IMT_HIP(c, hipMemcpyAsync(sib, d_out, (size_t)depth * n * 32, hipMemcpyDeviceToHost, c->stream));
        IMT_HIP(c, hipStreamSynchronize(c->stream));
    }
    return IMT_OK;
}

// ------------------------------------------------------------------------------------
// non-membership witness on the GPU
// ------------------------------------------------------------------------------------
extern "C" int imt_itree_non_membership_witness(imt_itree* t, const void* vals, size_t n, uint64_t* low_index,
                                                void* low_leaf, uint8_t* is_largest, void* low_sib, unsigned flags) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    IMT_NOT_SLICED(t);
    if (n == 0) return IMT_OK;
    if (!vals) return c->fail(IMT_ERR_ARG, "null vals");
    if ((flags & IMT_FMT_MASK) == 3) return c->fail(IMT_ERR_ARG, "unknown field-element format");
    if (n > ((size_t)1 << 31)) return c->fail(IMT_ERR_RANGE, "batch too large");
    int rc = c->set_device();
    if (rc) return rc;
    if ((rc = check_fe_ptrs(c, flags & IMT_DEVICE_PTRS, {vals, low_leaf, low_sib}))) return rc;
    if ((rc = ensure_device_index(t))) return rc;
    if ((rc = join_top(t))) return rc;
    const bool dev = flags & IMT_DEVICE_PTRS;
    const unsigned fmt = flags & IMT_FMT_MASK;
    hipStream_t s = c->stream;
    IMT_HIP(c, hipStreamSynchronize(t->up_stream));     // the index may have been written on the side stream
    size_t slot = 0;
    auto scratch = [&](size_t bytes) { return (uint8_t*)c->dev_scratch(slot++, bytes); };
    const uint8_t* d_vals = (const uint8_t*)vals;
    if (!dev) {
        uint8_t* up = scratch(n * 32);
        if (!up) return IMT_ERR_HIP;
        IMT_HIP(c, hipMemcpyAsync(up, vals, n * 32, hipMemcpyHostToDevice, s));
        d_vals = up;
    }
    int* d_perr = (int*)scratch(sizeof(int));
    if (!d_perr) return IMT_ERR_HIP;
    IMT_HIP(c, hipMemsetAsync(d_perr, 0, sizeof(int), s));
    if (fmt != IMT_FMT_CANONICAL) {
        uint8_t* can = scratch(n * 32);
        if (!can) return IMT_ERR_HIP;
        launch::convert(s, d_vals, can, n, fmt, IMT_FMT_CANONICAL, d_perr);
        d_vals = can;
    }
    auto outbuf = [&](void* user, size_t bytes) -> uint8_t* {
        if (!user) return nullptr;
        return dev ? (uint8_t*)user : scratch(bytes);
    };
    // the low index is needed on the device for the proof gather even if the caller does not want it
    uint64_t* g_low = (uint64_t*)(low_index && dev ? (uint8_t*)low_index : scratch(n * 8));
    uint8_t* g_leaf = outbuf(low_leaf, n * 96);
    uint8_t* g_lg = outbuf(is_largest, n);
    const size_t sib_bytes = (size_t)t->depth * n * 32;
    uint8_t* g_sib = outbuf(low_sib, sib_bytes);
    if (!g_low || (low_leaf && !g_leaf) || (is_largest && !g_lg) || (low_sib && !g_sib)) return IMT_ERR_HIP;
    prep::nm_witness(s, d_vals, t->d_val, t->d_sorted[t->sorted_cur], (uint32_t)t->size, (uint32_t)n, t->index_base,
                     t->part_mod, t->part_res, g_low, g_leaf, g_lg, d_perr);
    if (g_leaf && fmt != IMT_FMT_CANONICAL) launch::convert(s, g_leaf, g_leaf, n * 3, IMT_FMT_CANONICAL, fmt, c->d_err);
    if (g_sib) {
        launch::TreeView tv{t->d_nodes, t->d_off, t->d_len, c->d_zero, t->index_base};
        launch::SibLayout lay = (flags & IMT_SIB_ITEM_MAJOR) ? launch::SibLayout{1, t->depth} : launch::SibLayout{n, 1};
        launch::gather_proof(s, tv, g_low, n, t->depth, g_sib, lay, fmt);
    }
    int perr = 0;
    IMT_HIP(c, hipMemcpyAsync(&perr, d_perr, sizeof(int), hipMemcpyDeviceToHost, s));
    if (!dev) {
        if (low_index) IMT_HIP(c, hipMemcpyAsync(low_index, g_low, n * 8, hipMemcpyDeviceToHost, s));
        if (low_leaf) IMT_HIP(c, hipMemcpyAsync(low_leaf, g_leaf, n * 96, hipMemcpyDeviceToHost, s));
        if (is_largest) IMT_HIP(c, hipMemcpyAsync(is_largest, g_lg, n, hipMemcpyDeviceToHost, s));
        if (low_sib) IMT_HIP(c, hipMemcpyAsync(low_sib, g_sib, sib_bytes, hipMemcpyDeviceToHost, s));
    }
    IMT_HIP(c, hipStreamSynchronize(s));
    if (perr & prep::ERR_NONCANONICAL) return c->fail(IMT_ERR_NONCANONICAL, "a candidate is not reduced (>= p)");
    if (perr & prep::ERR_FOREIGN)
        return c->fail(IMT_ERR_VALUE, "a candidate belongs to another subtree (v %% %u != %u): this list says nothing about it",
                       t->part_mod, t->part_res);
    if (perr & (prep::ERR_ZERO | prep::ERR_DUPLICATE))
        return c->fail(IMT_ERR_VALUE, "a candidate is 0 or already in the tree (it has no non-membership witness)");
    return IMT_OK;
}

// ------------------------------------------------------------------------------------
// snapshot load / bulk build
// ------------------------------------------------------------------------------------
extern "C" int imt_itree_load(imt_itree* t, const void* preimages, uint64_t n, unsigned flags) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    IMT_NOT_SLICED(t);
    if (!preimages || n == 0) return c->fail(IMT_ERR_ARG, "null / empty snapshot");
    if ((flags & IMT_FMT_MASK) == 3) return c->fail(IMT_ERR_ARG, "unknown field-element format");
    if (n > t->cap) return c->fail(IMT_ERR_FULL, "snapshot has %llu leaves, capacity is %llu", (unsigned long long)n,
                                   (unsigned long long)t->cap);
    if (slice_open(t)) return c->fail(IMT_ERR_ARG, "a slice is open (issue its remaining units first)");
    int rc = c->set_device();
    if (rc) return rc;
    const bool dev = flags & IMT_DEVICE_PTRS;
    const unsigned fmt = flags & IMT_FMT_MASK;
    if ((rc = check_fe_ptrs(c, dev, {preimages}))) return rc;
    if ((rc = join_top(t))) return rc;
    for (auto& pl : t->plan)
        if (pl.in_flight) { IMT_HIP(c, hipEventSynchronize(pl.done)); pl.in_flight = false; pl.pipelined = false; }
    IMT_HIP(c, hipStreamSynchronize(t->up_stream));
    hipStream_t s = c->stream;
    // the snapshot copy and the sort workspace are as large as the tree: not something a context keeps after the call
    struct Trim {
        imt_ctx* c;
        ~Trim() { c->trim_scratch(2, (size_t)64 << 20); c->trim_scratch(3, (size_t)64 << 20); }
    } trim{c};
    // ---- the canonical preimages on the device (what is hashed; next_idx fields are global indices) ----
    const uint8_t* d_pre = (const uint8_t*)preimages;
    if (!dev || fmt != IMT_FMT_CANONICAL) {
        uint8_t* buf = (uint8_t*)c->dev_scratch(2, (size_t)n * 96);
        if (!buf) return IMT_ERR_HIP;
        if (!dev) IMT_HIP(c, hipMemcpyAsync(buf, preimages, (size_t)n * 96, hipMemcpyHostToDevice, s));
        if (fmt != IMT_FMT_CANONICAL) {
            if ((rc = c->clear_err())) return rc;
            launch::convert(s, dev ? d_pre : buf, buf, (size_t)n * 3, fmt, IMT_FMT_CANONICAL, c->d_err);
            if ((rc = c->sync_and_check())) return rc;
        }
        d_pre = buf;
    }
    // ---- list check on the device: ordered by val, every leaf points to its successor, the last one to {0, 0} ----
    const size_t ws_bytes = prep::load_ws_bytes((size_t)n);
    uint8_t* ws = (uint8_t*)c->dev_scratch(3, ws_bytes + 256);      // [error word | 256-byte aligned workspace]
    if (!ws) return IMT_ERR_HIP;
    int* d_perr = (int*)ws;
    const uint32_t* d_order = nullptr;
    int perr = 0;
    uint32_t bad = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        uint32_t* d_bad = nullptr;
        IMT_HIP(c, hipMemsetAsync(d_perr, 0, sizeof(int), s));
        IMT_HIP(c, prep::load_check(s, d_pre, (uint32_t)n, t->index_base, t->part_mod, t->part_res, ws + 256, ws_bytes,
                                    attempt == 1, d_perr, &d_bad, &d_order));
        IMT_HIP(c, hipMemcpyAsync(&perr, d_perr, sizeof(int), hipMemcpyDeviceToHost, s));
        IMT_HIP(c, hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, s));
        IMT_HIP(c, hipStreamSynchronize(s));
        // values that agree in their top 64 bits and came out of the radix sort in the wrong order: sort with the
        // 256-bit comparator and check again (the other bits of a first attempt that tied mean nothing)
        if (!(perr & prep::LOAD_TIE)) break;
    }
    if (perr & prep::ERR_NONCANONICAL) return c->fail(IMT_ERR_NONCANONICAL, "a field element in the snapshot is not reduced (>= p)");
    if (perr & prep::LOAD_SENTINEL) return c->fail(IMT_ERR_VALUE, "leaf 0 must be the {0,..} sentinel");
    if (perr & prep::LOAD_DUP) return c->fail(IMT_ERR_VALUE, "duplicate value in snapshot");
    if (perr & prep::LOAD_LINK)
        return c->fail(IMT_ERR_VALUE, "leaf %llu does not point to its successor", (unsigned long long)(t->index_base + bad));
    if (perr & prep::LOAD_LAST) return c->fail(IMT_ERR_VALUE, "the largest leaf must have next_val = next_idx = 0");
    if (perr & prep::ERR_FOREIGN)
        return c->fail(IMT_ERR_VALUE, "a value belongs to another subtree (v %% %u != %u)", t->part_mod, t->part_res);
    if (perr) return c->fail(IMT_ERR_INTERNAL, "snapshot check: unexpected error bits %d", perr);
    // ---- device rebuild: nothing of the tree has been written up to here ----
    if ((rc = c->clear_err())) return rc;
    prep::load_commit(s, d_pre, (uint32_t)n, t->d_val);
    IMT_HIP(c, hipMemcpyAsync(t->d_sorted[t->sorted_cur], d_order, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    for (unsigned l = 0; l <= t->depth; l++)
        launch::fill_level(s, t->d_nodes + t->h_off[l] * 32, t->h_len[l], c->d_zero + (size_t)l * 32);
    launch::hash_batch(s, d_pre, t->d_nodes, (size_t)n, 3, IMT_FMT_CANONICAL, IMT_FMT_DEVICE, c->d_err);
    uint64_t filled = n;
    for (unsigned l = 0; l < t->depth; l++) {
        const uint64_t parents = (filled + 1) / 2;
        if (t->h_len[l] >= 2) {
            launch::tree_level(s, t->d_nodes + t->h_off[l] * 32, t->d_nodes + t->h_off[l + 1] * 32, parents);
        } else {   // a single stored node: its sibling is the empty subtree of this height
            IMT_HIP(c, hipMemcpyAsync(t->d_nodes + t->h_off[l + 1] * 32, t->d_nodes + t->h_off[l] * 32, 32,
                                      hipMemcpyDeviceToDevice, s));
            launch::extend_root(s, t->d_nodes + t->h_off[l + 1] * 32, c->d_zero, l, l + 1);
        }
        filled = parents;
    }
    rc = c->sync_and_check();
    if (rc) return rc;
    // the device index is the tree's list now; the host mirror is rebuilt from it if a host-side call asks (ensure_mirror)
    t->pre.clear();
    t->pre.shrink_to_fit();
    t->sorted.clear();
    t->sorted.shrink_to_fit();
    t->size = n;
    t->mirror_valid = false;
    t->dev_index_valid = true;
    t->pending.active = false;
    for (auto& pl : t->plan) pl.has_root = false;
    return IMT_OK;
}

// ------------------------------------------------------------------------------------
// batch insertion
// ------------------------------------------------------------------------------------
namespace {

const int64_t REF_NONE = INT64_MIN;   // "no successor" in the host path's neighbour references
const U256 U256_ZERO = {0, 0, 0, 0};

// Host prepare (IMT_HOST_PREP).  Neighbour references: >= 0 -> rank of a batch element in value order;
// < 0 -> ~(leaf index of a stored leaf); REF_NONE.
struct HostPlan {
    std::vector<U256> v;                      // batch values in insertion order, canonical
    std::vector<uint8_t> lowleaf, newleaf;    // hash-free outputs, if requested
    uint64_t M = 0;
};

inline uint64_t ref_leaf(const imt_itree* t, const HostPlan& hp, int64_t ref) {
    return ref >= 0 ? hp.M + t->w_ord[(size_t)ref] : (uint64_t)~ref;
}
inline const U256& ref_val(const imt_itree* t, const HostPlan& hp, int64_t ref) {
    return ref >= 0 ? hp.v[t->w_ord[(size_t)ref]] : t->pre[(size_t)~ref].val;
}

// Sections of the host path: (1) values to the host, (2) low leaf of every insertion
// (update_idx_leaf :639-658 as a predecessor search: sort the batch, locate each value between two
// stored leaves, then unlink the batch from that list in reverse insertion order -- what is adjacent at
// unlink time is exactly what had been inserted earlier), (4) event preimages and their (position,
// time) order into the pinned staging of P, (5) upload on the side stream.
int host_prepare(imt_itree* t, PlanSet& P, const void* vals, size_t n, unsigned flags, bool want_lowleaf,
                 bool want_newleaf, HostPlan& hp) {
    imt_ctx* c = t->ctx;
    const bool dev = flags & IMT_DEVICE_PTRS;
    const uint64_t M = hp.M = t->size;
    const size_t E = 2 * n;
    int rc = ensure_mirror(t);
    if (rc) return rc;
    // With device pointers the values are read on the side stream, so the call does not wait for an
    // earlier batch that is still running.
    std::vector<U256>& v = hp.v;
    if ((rc = fetch_canonical(c, dev ? t->up_stream : c->stream, vals, n, flags, v))) return rc;

    std::vector<uint32_t>& ord = t->w_ord;
    {
        std::vector<std::pair<uint64_t, uint32_t>>& sk = t->w_sortkey;   // (top limb, index): cheap compares
        sk.resize(n);
        for (size_t i = 0; i < n; i++) sk[i] = {v[i][3], (uint32_t)i};
        std::sort(sk.begin(), sk.end(),
                  [&](const std::pair<uint64_t, uint32_t>& a, const std::pair<uint64_t, uint32_t>& b) {
                      if (a.first != b.first) return a.first < b.first;
                      return lt256(v[a.second], v[b.second]);
                  });
        ord.resize(n);
        for (size_t i = 0; i < n; i++) ord[i] = sk[i].second;
    }
    if (is_zero256(v[ord[0]])) return c->fail(IMT_ERR_VALUE, "value 0 cannot be inserted");
    if (t->part_mod > 1)
        for (size_t i = 0; i < n; i++)
            if (prep::mod_small(reinterpret_cast<const uint8_t*>(v[i].data()), t->part_mod) != t->part_res)
                return c->fail(IMT_ERR_VALUE, "a value belongs to another subtree (v %% %u != %u)", t->part_mod, t->part_res);
    for (size_t r = 1; r < n; r++)
        if (v[ord[r]] == v[ord[r - 1]]) return c->fail(IMT_ERR_VALUE, "duplicate value inside the batch");
    std::vector<int64_t>&prv = t->w_prv, &nxt = t->w_nxt;
    prv.resize(n);
    nxt.resize(n);
    {
        size_t q = 0;   // first stored entry with val >= current
        const size_t S = t->sorted.size();
        for (size_t r = 0; r < n; r++) {
            const U256& x = v[ord[r]];
            while (q < S && cmp_ent(t, t->sorted[q], x) < 0) q++;
            if (q < S && cmp_ent(t, t->sorted[q], x) == 0) return c->fail(IMT_ERR_VALUE, "value already in the tree");
            // q >= 1 because the sentinel 0 is stored and x > 0
            const bool same_gap_prev = r > 0 && cmp_ent(t, t->sorted[q - 1], v[ord[r - 1]]) < 0;
            prv[r] = same_gap_prev ? (int64_t)(r - 1) : ~(int64_t)t->sorted[q - 1].idx;
            const bool next_in_gap = r + 1 < n && (q >= S || cmp_ent(t, t->sorted[q], v[ord[r + 1]]) > 0);
            nxt[r] = next_in_gap ? (int64_t)(r + 1) : (q < S ? ~(int64_t)t->sorted[q].idx : REF_NONE);
        }
    }
    std::vector<uint32_t>& rank_of = t->w_rank;
    rank_of.resize(n);
    for (size_t r = 0; r < n; r++) rank_of[ord[r]] = (uint32_t)r;
    std::vector<int64_t>&pred = t->w_pred, &succ = t->w_succ;   // by insertion time
    pred.resize(n);
    succ.resize(n);
    for (size_t ii = n; ii-- > 0;) {
        const uint32_t r = rank_of[ii];
        pred[ii] = prv[r];
        succ[ii] = nxt[r];
        if (prv[r] >= 0) nxt[prv[r]] = nxt[r];
        if (nxt[r] >= 0) prv[nxt[r]] = prv[r];
    }

    // events: preimages at every time step + the hash-free outputs
    uint8_t* h_pre = P.h_pin;
    uint32_t* h_tab = reinterpret_cast<uint32_t*>(P.h_pin + P.cap_events * 96);
    uint32_t *h_node = h_tab, *h_time = h_tab + P.cap_events, *h_rs = h_tab + 2 * P.cap_events,
             *h_re = h_tab + 3 * P.cap_events;
    std::vector<uint64_t>& o_low = t->w_low;
    std::vector<uint8_t>& o_largest = t->w_largest;
    o_low.resize(n);
    o_largest.resize(n);
    hp.lowleaf.assign(want_lowleaf ? n * 96 : 0, 0);
    hp.newleaf.assign(want_newleaf ? n * 96 : 0, 0);
    std::vector<uint64_t>& keys = t->w_keys;   // (pos << 32) | event
    keys.resize(E);
    for (size_t i = 0; i < n; i++) {
        const uint64_t low = ref_leaf(t, hp, pred[i]);
        const U256& lowval = ref_val(t, hp, pred[i]);
        const bool has_succ = succ[i] != REF_NONE;
        const U256& sval = has_succ ? ref_val(t, hp, succ[i]) : U256_ZERO;
        const uint64_t sidx = has_succ ? t->index_base + ref_leaf(t, hp, succ[i]) : 0;
        o_low[i] = low;
        o_largest[i] = has_succ ? 0 : 1;                               // :737-742
        if (want_lowleaf) put_pre(&hp.lowleaf[i * 96], lowval, sval, sidx);
        put_pre(h_pre + (2 * i) * 96, lowval, v[i], t->index_base + M + i);   // low leaf rewritten :655-656
        put_pre(h_pre + (2 * i + 1) * 96, v[i], sval, sidx);           // new leaf inherits :650-654
        if (want_newleaf) std::memcpy(&hp.newleaf[i * 96], h_pre + (2 * i + 1) * 96, 96);
        keys[2 * i] = (low << 32) | (uint64_t)(2 * i);
        keys[2 * i + 1] = ((M + i) << 32) | (uint64_t)(2 * i + 1);
    }
    {   // stable LSD radix sort on the 32-bit position (two 16-bit passes); events are already in time order
        std::vector<uint64_t>& tmp = t->w_keys2;
        std::vector<uint32_t>& hist = t->w_hist;
        tmp.resize(E);
        hist.resize(1u << 16);
        for (int pass = 0; pass < 2; pass++) {
            const int sh = 32 + 16 * pass;
            std::vector<uint64_t>& src = pass ? tmp : keys;
            std::vector<uint64_t>& dst = pass ? keys : tmp;
            std::fill(hist.begin(), hist.end(), 0u);
            for (size_t x = 0; x < E; x++) hist[(src[x] >> sh) & 0xffff]++;
            uint32_t run = 0;
            for (auto& hcount : hist) { const uint32_t cnt = hcount; hcount = run; run += cnt; }
            for (size_t x = 0; x < E; x++) dst[hist[(src[x] >> sh) & 0xffff]++] = src[x];
        }
    }
    for (size_t k = 0; k < E;) {
        size_t j = k;
        const uint32_t pos = (uint32_t)(keys[k] >> 32);
        while (j < E && (uint32_t)(keys[j] >> 32) == pos) j++;
        for (size_t x = k; x < j; x++) {
            h_node[x] = pos;
            h_time[x] = (uint32_t)keys[x];
            h_rs[x] = (uint32_t)k;
            h_re[x] = (uint32_t)j;
        }
        k = j;
    }
    IMT_HIP(c, hipMemcpyAsync(P.d_pre, h_pre, E * 96, hipMemcpyHostToDevice, t->up_stream));
    IMT_HIP(c, hipMemcpyAsync(P.d_tab[0][0], h_node, E * 4, hipMemcpyHostToDevice, t->up_stream));
    IMT_HIP(c, hipMemcpyAsync(P.d_tab[0][1], h_time, E * 4, hipMemcpyHostToDevice, t->up_stream));
    IMT_HIP(c, hipMemcpyAsync(P.d_tab[0][2], h_rs, E * 4, hipMemcpyHostToDevice, t->up_stream));
    IMT_HIP(c, hipMemcpyAsync(P.d_tab[0][3], h_re, E * 4, hipMemcpyHostToDevice, t->up_stream));
    return IMT_OK;
}

// host mirror after a host-prepared batch: new preimages, rewritten low leaves, merged sorted index
void host_commit(imt_itree* t, const HostPlan& hp, size_t n) {
    const uint64_t M = hp.M;
    const std::vector<U256>& v = hp.v;
    t->dev_index_valid = false;
    t->pre.resize(M + n);
    for (size_t i = 0; i < n; i++) {
        const bool has_succ = t->w_succ[i] != REF_NONE;
        Pre& lowp = t->pre[t->w_low[i]];
        Pre& np = t->pre[M + i];
        np.val = v[i];
        np.next_val = has_succ ? ref_val(t, hp, t->w_succ[i]) : U256_ZERO;
        np.next_idx = has_succ ? ref_leaf(t, hp, t->w_succ[i]) : 0;
        lowp.next_val = v[i];
        lowp.next_idx = M + i;
    }
    std::vector<SortedEnt>& merged = t->w_merged;
    merged.clear();
    merged.reserve(t->sorted.size() + n);
    size_t q = 0;
    for (size_t r = 0; r < n; r++) {
        const U256& x = v[t->w_ord[r]];
        while (q < t->sorted.size() && cmp_ent(t, t->sorted[q], x) < 0) merged.push_back(t->sorted[q++]);
        merged.push_back(SortedEnt{x[3], M + t->w_ord[r]});
    }
    while (q < t->sorted.size()) merged.push_back(t->sorted[q++]);
    t->sorted.swap(merged);
    t->size = M + n;
}

// GPU prepare (default): the same on the device (imt_prep.hip), on the side stream.  The hash-free
// outputs go straight to device buffers (the user's, or scratch in host-pointer mode).  The batch is
// committed to the device index only if its values are acceptable.
struct GpuOuts {
    uint64_t* low = nullptr;
    uint8_t *largest = nullptr, *lowleaf = nullptr, *newleaf = nullptr;
};
int gpu_prepare(imt_itree* t, PlanSet& P, const void* vals, size_t n, unsigned flags, const imt_insert_out* out,
                size_t& slot, GpuOuts& go, double& wait_ms) {
    imt_ctx* c = t->ctx;
    const bool dev = flags & IMT_DEVICE_PTRS;
    const unsigned fmt = flags & IMT_FMT_MASK;
    int rc = ensure_device_index(t);
    if (rc) return rc;
    hipStream_t ps = t->up_stream;
    const uint8_t* d_vals = (const uint8_t*)vals;
    if (!dev) {
        uint8_t* up = (uint8_t*)c->dev_scratch(slot++, n * 32);
        if (!up) return IMT_ERR_HIP;
        IMT_HIP(c, hipMemcpyAsync(up, vals, n * 32, hipMemcpyHostToDevice, ps));
        d_vals = up;
    }
    IMT_HIP(c, hipMemsetAsync(P.ws.err, 0, sizeof(int), ps));
    if (fmt != IMT_FMT_CANONICAL) {
        // the plan's own buffer, not context scratch: this runs on the side stream, which IMT_INPUTS_READY leaves
        // unordered behind the context's stream, where an earlier asynchronous call (imt_itree_lift_batch,
        // imt_insert_trace_batch) may still be using the context's scratch slots
        launch::convert(ps, d_vals, P.d_canon, n, fmt, IMT_FMT_CANONICAL, P.ws.err);   // sets bit 0 = non-canonical
        d_vals = P.d_canon;
    }
    if (out) {
        auto dev_out = [&](void* user, size_t bytes) -> void* {
            if (!user) return nullptr;
            return dev ? user : c->dev_scratch(slot++, bytes);
        };
        go.low = (uint64_t*)dev_out(out->low_index, n * 8);
        go.largest = (uint8_t*)dev_out(out->is_largest, n);
        go.lowleaf = (uint8_t*)dev_out(out->low_leaf, n * 96);
        go.newleaf = (uint8_t*)dev_out(out->new_leaf, n * 96);
        if ((out->low_index && !go.low) || (out->is_largest && !go.largest) || (out->low_leaf && !go.lowleaf) ||
            (out->new_leaf && !go.newleaf))
            return IMT_ERR_HIP;
    }
    P.ws.part_mod = t->part_mod;
    P.ws.part_res = t->part_res;
    IMT_HIP(c, prep::run(ps, P.ws, d_vals, t->d_val, t->d_sorted[t->sorted_cur], t->d_sorted[t->sorted_cur ^ 1],
                         (uint32_t)t->size, (uint32_t)n, t->index_base, P.d_pre, P.d_tab[0][0], P.d_tab[0][1], P.d_tab[0][2],
                         P.d_tab[0][3], go.low, go.largest, go.lowleaf, go.newleaf));
    IMT_HIP(c, hipMemcpyAsync(t->h_err_pin, P.ws.err, sizeof(int), hipMemcpyDeviceToHost, ps));
    const auto w0 = std::chrono::steady_clock::now();
    IMT_HIP(c, hipStreamSynchronize(ps));
    wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
    const int perr = *t->h_err_pin;
    if (perr & prep::ERR_NONCANONICAL) return c->fail(IMT_ERR_NONCANONICAL, "a value is not reduced (>= p)");
    if (perr & prep::ERR_ZERO) return c->fail(IMT_ERR_VALUE, "value 0 cannot be inserted");
    if (perr & prep::ERR_FOREIGN)
        return c->fail(IMT_ERR_VALUE, "a value belongs to another subtree (v %% %u != %u)", t->part_mod, t->part_res);
    if (perr & prep::ERR_DUPLICATE)
        return c->fail(IMT_ERR_VALUE, "duplicate value (inside the batch or already in the tree)");
    return IMT_OK;     // the merged index sits in d_sorted[sorted_cur ^ 1]; the caller flips when it commits
}

}  // namespace

extern "C" int imt_itree_insert_batch(imt_itree* t, const void* vals, size_t n, const imt_insert_out* out,
                                      unsigned flags) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    IMT_NOT_SLICED(t);
    if (n == 0) return IMT_OK;
    if (!vals) return c->fail(IMT_ERR_ARG, "null vals");
    if ((flags & IMT_FMT_MASK) == 3) return c->fail(IMT_ERR_ARG, "unknown field-element format");
    if (n > ((size_t)1 << 30)) return c->fail(IMT_ERR_RANGE, "batch too large");
    if (t->pending.active) return c->fail(IMT_ERR_ARG, "a sharded batch is open (imt_itree_batch_end first)");
    if (slice_open(t)) return c->fail(IMT_ERR_ARG, "a slice is open (issue its remaining units first)");
    int rc = c->set_device();
    if (rc) return rc;
    const bool dev = flags & IMT_DEVICE_PTRS;
    if ((rc = check_fe_ptrs(c, dev, {vals})) || (rc = check_out_ptrs(c, dev, out))) return rc;
    const bool gpu_prep = (flags & IMT_HOST_PREP) == 0;
    const unsigned fmt = flags & IMT_FMT_MASK;
    const uint64_t M = t->size;
    if (M + n > t->cap) return c->fail(IMT_ERR_FULL, "tree capacity %llu exceeded", (unsigned long long)t->cap);
    const auto host_t0 = std::chrono::steady_clock::now();
    double host_wait_ms = 0;

    // ---- plan buffers ----
    const size_t E = 2 * n;
    const unsigned L0 = std::min(ceil_log2(M + n), t->depth);
    PlanSet& P = t->plan[t->cur];
    if (P.in_flight) {   // back-pressure: at most NSETS batches in flight
        const auto w0 = std::chrono::steady_clock::now();
        IMT_HIP(c, hipEventSynchronize(P.done));
        host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
        P.in_flight = false;
    }
    rc = plan_reserve(c, P, E, t->depth, t->cap);   // all levels up front: growing later would stall the pipeline
    if (rc) return rc;
    if (dev && (flags & IMT_PIPELINE) && (rc = reserve_all_plans(t, E))) return rc;

    // ---- hash-free part: low leaves, event preimages, event order (side stream) ----
    // The side stream reads the caller's `vals` and writes the caller's hash-free outputs (low_index,
    // is_largest, low_leaf, new_leaf), so with device pointers it is first ordered behind everything the
    // caller has enqueued on the context's stream (imt.h: "work is enqueued on the context's stream").
    // IMT_INPUTS_READY waives that: the caller guarantees those buffers are idle.
    // (Host-pointer calls are ordered too: their staging scratch may still be read by an asynchronous
    // device-pointer call, e.g. imt_itree_lift_batch, enqueued earlier on the context's stream.)
    if (!(dev && (flags & IMT_INPUTS_READY))) {
        IMT_HIP(c, hipEventRecord(t->in_mark, c->stream));
        IMT_HIP(c, hipStreamWaitEvent(t->up_stream, t->in_mark, 0));
    }
    size_t slot = 2;
    HostPlan hp;
    GpuOuts go;
    if (gpu_prep)
        rc = gpu_prepare(t, P, vals, n, flags, out, slot, go, host_wait_ms);
    else
        rc = host_prepare(t, P, vals, n, flags, out && out->low_leaf, out && out->new_leaf, hp);
    if (rc) return rc;
    IMT_HIP(c, hipEventRecord(t->up_done, t->up_stream));

    // ---- compute stream: the context's, or one of the NPIPE pipeline streams ----
    const bool pipelined = dev && (flags & IMT_PIPELINE);
    hipStream_t s = c->stream;
    const PlanSet* prev = nullptr;      // the batch before this one, if it is still on a pipeline stream
    if (pipelined) {
        s = t->pipe_stream[t->batch_no % imt_itree::NPIPE];
        IMT_HIP(c, hipEventRecord(t->user_mark, c->stream));     // buffers last used on the user's stream
        IMT_HIP(c, hipStreamWaitEvent(s, t->user_mark, 0));
        const PlanSet& o = t->plan[(t->cur + imt_itree::NSETS - 1) % imt_itree::NSETS];
        if (o.in_flight && o.pipelined) prev = &o;
        t->pipe_pending = true;
    } else {
        if ((rc = join_top(t))) return rc;
    }
    IMT_HIP(c, hipStreamWaitEvent(s, t->up_done, 0));

    // ---- GPU outputs ----
    uint8_t *g_old = nullptr, *g_int = nullptr, *g_new = nullptr, *g_ls = nullptr, *g_ns = nullptr;
    auto gpu_out = [&](void* user, size_t bytes) -> uint8_t* {
        if (!user) return nullptr;
        if (dev) return (uint8_t*)user;
        return (uint8_t*)c->dev_scratch(slot++, bytes);
    };
    const size_t sib_bytes = (size_t)t->global_depth * n * 32;
    if (out) {
        g_old = gpu_out(out->old_root, n * 32);
        g_int = gpu_out(out->interim_root, n * 32);
        g_new = gpu_out(out->new_root, n * 32);
        g_ls = gpu_out(out->low_sib, sib_bytes);
        g_ns = gpu_out(out->new_sib, sib_bytes);
        if ((out->old_root && !g_old) || (out->interim_root && !g_int) || (out->new_root && !g_new) ||
            (out->low_sib && !g_ls) || (out->new_sib && !g_ns))
            return IMT_ERR_HIP;
    }
    // a placed tree writes rows [0, depth) of sibling arrays dimensioned for global_depth levels
    launch::SibLayout lay = (flags & IMT_SIB_ITEM_MAJOR) ? launch::SibLayout{1, t->global_depth} : launch::SibLayout{n, 1};

    // ---- leaf hashes, index phase (no hashing), then the hash sweep ----
    int pf = c->prof_begin(IMT_PROF_LEAVES, s);
    launch::sweep_leaves(s, P.d_pre, P.d_tab[0][1], P.d_val[0], 0, (uint32_t)E, IMT_FMT_CANONICAL, c->d_err, c->coop_max_events);
    c->prof_end(pf, s);
    pf = c->prof_begin(IMT_PROF_INDEX, s);
    for (unsigned l = 0; l < L0; l++) {
        const int a = l & 1, b = a ^ 1;
        // time table of level l: the uploaded one for l = 0, otherwise slot l-1 of d_timen
        const uint32_t* time_in = l == 0 ? P.d_tab[0][1] : P.d_timen + (size_t)(l - 1) * P.cap_events;
        sweep::LevelTable in{P.d_tab[a][0], time_in, P.d_tab[a][2], P.d_tab[a][3]};
        sweep::LevelOut o{P.d_tab[b][0], P.d_timen + (size_t)l * P.cap_events, P.d_tab[b][2], P.d_tab[b][3],
                          P.d_from + (size_t)l * P.cap_events, P.d_sibsrc + (size_t)l * P.cap_events,
                          P.d_nodeb + (size_t)l * P.cap_events, nullptr};
        launch::merge_level(s, in, o, (uint32_t)E);
    }
    c->prof_end(pf, s);
    for (unsigned l = 0; l < L0; l++) {
        const uint8_t* vin = P.d_val[l & 1];
        uint8_t* vout = P.d_val[(l & 1) ^ 1];
        const size_t o = (size_t)l * P.cap_events;
        // stored level l must hold the previous batch's final versions: its write-back of that level,
        // or (at and above its L0) its top kernel
        if (prev) IMT_HIP(c, hipStreamWaitEvent(s, l < prev->l0 ? prev->wb_done[l] : prev->done, 0));
        pf = c->prof_begin(IMT_PROF_LEVEL, s);
        launch::sweep_level(s, vin, vout, P.d_from + o, P.d_sibsrc + o, P.d_nodeb + o, P.d_timen + o,
                            t->d_nodes + t->h_off[l] * 32, t->h_len[l], c->d_zero + (size_t)l * 32, 0, (uint32_t)E, g_ls,
                            g_ns, lay, l, fmt, c->coop_max_events);
        c->prof_end(pf, s);
        pf = c->prof_begin(IMT_PROF_WRITEBACK, s);
        launch::writeback(s, vin, P.d_from + o, P.d_nodeb + o, t->d_nodes + t->h_off[l] * 32, (uint32_t)E);
        c->prof_end(pf, s);
        if (pipelined) IMT_HIP(c, hipEventRecord(P.wb_done[l], s));
    }
    // ---- levels [L0, depth): every event alone in node 0 against the empty subtree of that height.  Ordinary
    // launches of the same kernel, so consecutive batches overlap here level by level as well; the last event's node
    // of every level goes back to the stored tree.  old_root[0] is the root the previous batch left: it is read
    // right before the launch that overwrites the stored root.
    auto read_old_root = [&]() -> int {
        if (prev) IMT_HIP(c, hipStreamWaitEvent(s, prev->done, 0));
        if (g_old) launch::convert(s, t->d_nodes + t->h_off[t->depth] * 32, g_old, 1, IMT_FMT_DEVICE, fmt, c->d_err);
        return IMT_OK;
    };
    for (unsigned l = L0; l < t->depth; l++) {
        if (l + 1 == t->depth && (rc = read_old_root())) return rc;
        // the previous batch stored node l + 1 (and node l, at its own L0) from its launch of this level
        if (prev) IMT_HIP(c, hipStreamWaitEvent(s, l >= prev->l0 ? prev->wb_done[l] : prev->done, 0));
        pf = c->prof_begin(IMT_PROF_TOP, s);
        launch::sweep_upper(s, P.d_val[l & 1], P.d_val[(l & 1) ^ 1], c->d_zero + (size_t)l * 32, 0, (uint32_t)E,
                            (uint32_t)E - 1, l == L0 ? t->d_nodes + t->h_off[l] * 32 : nullptr,
                            t->d_nodes + t->h_off[l + 1] * 32, g_ls, g_ns, lay, l, fmt, c->coop_max_events);
        c->prof_end(pf, s);
        if (pipelined) IMT_HIP(c, hipEventRecord(P.wb_done[l], s));
    }
    if (L0 == t->depth && (rc = read_old_root())) return rc;
    launch::emit_roots(s, P.d_val[t->depth & 1], 0, (uint32_t)E, (uint32_t)E, g_old, g_int, g_new, fmt, nullptr,
                       L0 == t->depth ? t->d_nodes + t->h_off[t->depth] * 32 : nullptr);
    IMT_HIP(c, hipMemcpyAsync(P.d_root, t->d_nodes + t->h_off[t->depth] * 32, 32, hipMemcpyDeviceToDevice, s));
    P.has_root = true;
    IMT_HIP(c, hipEventRecord(P.done, s));
    P.pipelined = pipelined;
    P.sliced = false;
    P.l0 = L0;
    P.in_flight = true;
    t->cur = (t->cur + 1) % imt_itree::NSETS;
    t->batch_no++;

    // ---- commit the hash-free state ----
    if (gpu_prep) {
        t->sorted_cur ^= 1;          // the merged index becomes current together with the size
        t->size = M + n;
        t->mirror_valid = false;     // rebuilt from the device index when a host-side call needs it
    } else {
        host_commit(t, hp, n);
    }

    // ---- hash-free outputs ----
    if (out && gpu_prep) {
        // written by k_events in canonical form; other formats are converted in place behind up_done
        if (fmt != IMT_FMT_CANONICAL) {
            if (go.lowleaf) launch::convert(s, go.lowleaf, go.lowleaf, n * 3, IMT_FMT_CANONICAL, fmt, c->d_err);
            if (go.newleaf) launch::convert(s, go.newleaf, go.newleaf, n * 3, IMT_FMT_CANONICAL, fmt, c->d_err);
        }
        if (!dev) {
            if (go.low) IMT_HIP(c, hipMemcpyAsync(out->low_index, go.low, n * 8, hipMemcpyDeviceToHost, s));
            if (go.largest) IMT_HIP(c, hipMemcpyAsync(out->is_largest, go.largest, n, hipMemcpyDeviceToHost, s));
            if (go.lowleaf) IMT_HIP(c, hipMemcpyAsync(out->low_leaf, go.lowleaf, n * 96, hipMemcpyDeviceToHost, s));
            if (go.newleaf) IMT_HIP(c, hipMemcpyAsync(out->new_leaf, go.newleaf, n * 96, hipMemcpyDeviceToHost, s));
        }
    } else if (out) {
        auto host_out = [&](void* user, const void* src, size_t bytes) -> int {
            if (!user) return IMT_OK;
            if (dev) {   // side stream: does not wait for the sweep; src is host memory of this call
                IMT_HIP(c, hipMemcpyAsync(user, src, bytes, hipMemcpyHostToDevice, t->up_stream));
                IMT_HIP(c, hipStreamSynchronize(t->up_stream));
            } else {
                std::memcpy(user, src, bytes);
            }
            return IMT_OK;
        };
        if (t->index_base && out->low_index) {
            std::vector<uint64_t> glob(t->w_low);
            for (auto& x : glob) x += t->index_base;
            if ((rc = host_out(out->low_index, glob.data(), n * 8))) return rc;
        } else if ((rc = host_out(out->low_index, t->w_low.data(), n * 8))) return rc;
        if ((rc = host_out(out->is_largest, t->w_largest.data(), n))) return rc;
        if (fmt == IMT_FMT_CANONICAL) {
            if ((rc = host_out(out->low_leaf, hp.lowleaf.data(), n * 96))) return rc;
            if ((rc = host_out(out->new_leaf, hp.newleaf.data(), n * 96))) return rc;
        } else {
            for (int w = 0; w < 2; w++) {
                void* user = w ? out->new_leaf : out->low_leaf;
                if (!user) continue;
                const std::vector<uint8_t>& src = w ? hp.newleaf : hp.lowleaf;
                uint8_t* d_in = (uint8_t*)c->dev_scratch(slot++, n * 96);
                if (!d_in) return IMT_ERR_HIP;
                IMT_HIP(c, hipMemcpyAsync(d_in, src.data(), n * 96, hipMemcpyHostToDevice, s));
                uint8_t* d_o = dev ? (uint8_t*)user : (uint8_t*)c->dev_scratch(slot++, n * 96);
                if (!d_o) return IMT_ERR_HIP;
                launch::convert(s, d_in, d_o, n * 3, IMT_FMT_CANONICAL, fmt, c->d_err);
                if (!dev) IMT_HIP(c, hipMemcpyAsync(user, d_o, n * 96, hipMemcpyDeviceToHost, s));
                IMT_HIP(c, hipStreamSynchronize(s));
            }
        }
    }
    if (out && !dev) {
        if (out->old_root) IMT_HIP(c, hipMemcpyAsync(out->old_root, g_old, n * 32, hipMemcpyDeviceToHost, s));
        if (out->interim_root) IMT_HIP(c, hipMemcpyAsync(out->interim_root, g_int, n * 32, hipMemcpyDeviceToHost, s));
        if (out->new_root) IMT_HIP(c, hipMemcpyAsync(out->new_root, g_new, n * 32, hipMemcpyDeviceToHost, s));
        // level-major: only rows [0, depth) were written (a placed tree's upper rows come from imt_itree_lift_batch)
        const size_t sib_copy = (flags & IMT_SIB_ITEM_MAJOR) ? sib_bytes : (size_t)t->depth * n * 32;
        if (out->low_sib) IMT_HIP(c, hipMemcpyAsync(out->low_sib, g_ls, sib_copy, hipMemcpyDeviceToHost, s));
        if (out->new_sib) IMT_HIP(c, hipMemcpyAsync(out->new_sib, g_ns, sib_copy, hipMemcpyDeviceToHost, s));
    }
    if (c->profiling) {
        c->prof_ms[IMT_PROF_HOST] +=
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count() - host_wait_ms;
        c->prof_n[IMT_PROF_HOST] += 1;
    }
    if (!dev) IMT_HIP(c, hipStreamSynchronize(s));
    return IMT_OK;
}

// ------------------------------------------------------------------------------------
// e: one tree on several GPUs, sequential semantics (imt_itree_batch_*)
// ------------------------------------------------------------------------------------
extern "C" int imt_itree_batch_begin(imt_itree* t, const void* vals, size_t n, unsigned flags, uint32_t* events_out,
                                     uint32_t* l0_out) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    IMT_NOT_SLICED(t);
    if (!vals || n == 0) return c->fail(IMT_ERR_ARG, "null / empty batch");
    if ((flags & IMT_FMT_MASK) == 3) return c->fail(IMT_ERR_ARG, "unknown field-element format");
    if (t->pending.active) return c->fail(IMT_ERR_ARG, "a sharded batch is already open");
    if (slice_open(t)) return c->fail(IMT_ERR_ARG, "a slice is open (issue its remaining units first)");
    if (n > ((size_t)1 << 30)) return c->fail(IMT_ERR_RANGE, "batch too large");
    int rc = c->set_device();
    if (rc) return rc;
    const uint64_t M = t->size;
    if (M + n > t->cap) return c->fail(IMT_ERR_FULL, "tree capacity %llu exceeded", (unsigned long long)t->cap);
    if ((rc = join_top(t))) return rc;
    if ((rc = ensure_device_index(t))) return rc;
    const bool dev = flags & IMT_DEVICE_PTRS;
    if ((rc = check_fe_ptrs(c, dev, {vals}))) return rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    const size_t E = 2 * n;
    const unsigned L0 = std::min(ceil_log2(M + n), t->depth);
    PlanSet& P = t->plan[t->cur];
    if (P.in_flight) {
        IMT_HIP(c, hipEventSynchronize(P.done));
        P.in_flight = false;
    }
    if ((rc = plan_reserve(c, P, E, t->depth, t->cap))) return rc;
    hipStream_t s = c->stream;
    size_t slot = 2;
    const uint8_t* d_vals = (const uint8_t*)vals;
    if (!dev) {
        uint8_t* up = (uint8_t*)c->dev_scratch(slot++, n * 32);
        if (!up) return IMT_ERR_HIP;
        IMT_HIP(c, hipMemcpyAsync(up, vals, n * 32, hipMemcpyHostToDevice, s));
        d_vals = up;
    }
    IMT_HIP(c, hipMemsetAsync(P.ws.err, 0, sizeof(int), s));
    if (fmt != IMT_FMT_CANONICAL) {
        uint8_t* can = (uint8_t*)c->dev_scratch(slot++, n * 32);
        if (!can) return IMT_ERR_HIP;
        launch::convert(s, d_vals, can, n, fmt, IMT_FMT_CANONICAL, P.ws.err);
        d_vals = can;
    }
    IMT_HIP(c, hipStreamSynchronize(t->up_stream));
    P.ws.part_mod = t->part_mod;
    P.ws.part_res = t->part_res;
    IMT_HIP(c, prep::run(s, P.ws, d_vals, t->d_val, t->d_sorted[t->sorted_cur], t->d_sorted[t->sorted_cur ^ 1], (uint32_t)M,
                         (uint32_t)n, t->index_base, P.d_pre, P.d_tab[0][0], P.d_tab[0][1], P.d_tab[0][2], P.d_tab[0][3],
                         P.ws.o_low, P.ws.o_largest, P.ws.o_lowleaf, P.ws.o_newleaf));
    IMT_HIP(c, hipMemcpyAsync(t->h_err_pin, P.ws.err, sizeof(int), hipMemcpyDeviceToHost, s));
    IMT_HIP(c, hipStreamSynchronize(s));
    const int perr = *t->h_err_pin;
    if (perr & prep::ERR_NONCANONICAL) return c->fail(IMT_ERR_NONCANONICAL, "a value is not reduced (>= p)");
    if (perr & prep::ERR_ZERO) return c->fail(IMT_ERR_VALUE, "value 0 cannot be inserted");
    if (perr & prep::ERR_FOREIGN)
        return c->fail(IMT_ERR_VALUE, "a value belongs to another subtree (v %% %u != %u)", t->part_mod, t->part_res);
    if (perr & prep::ERR_DUPLICATE) return c->fail(IMT_ERR_VALUE, "duplicate value (inside the batch or already in the tree)");
    // index phase for every level, with the slot of every event per level
    launch::slot0(s, P.d_tab[0][1], P.d_slot, (uint32_t)E);
    for (unsigned l = 0; l < L0; l++) {
        const int a = l & 1, b = a ^ 1;
        const uint32_t* time_in = l == 0 ? P.d_tab[0][1] : P.d_timen + (size_t)(l - 1) * P.cap_events;
        sweep::LevelTable in{P.d_tab[a][0], time_in, P.d_tab[a][2], P.d_tab[a][3]};
        sweep::LevelOut o{P.d_tab[b][0], P.d_timen + (size_t)l * P.cap_events, P.d_tab[b][2], P.d_tab[b][3],
                          P.d_from + (size_t)l * P.cap_events, P.d_sibsrc + (size_t)l * P.cap_events,
                          P.d_nodeb + (size_t)l * P.cap_events, P.d_slot + (size_t)(l + 1) * P.cap_events};
        launch::merge_level(s, in, o, (uint32_t)E);
    }
    t->pending.active = true;
    t->pending.n = n;
    t->pending.l0 = L0;
    t->pending.set = t->cur;
    if (events_out) *events_out = (uint32_t)E;
    if (l0_out) *l0_out = L0;
    return IMT_OK;
}

#define IMT_PENDING(t, c)                                                                   \
    if (!(t)) return IMT_ERR_ARG;                                                           \
    imt_ctx* c = (t)->ctx;                                                                  \
    if (!(t)->pending.active) return c->fail(IMT_ERR_ARG, "no sharded batch is open");      \
    { int rc__ = c->set_device(); if (rc__) return rc__; }                                  \
    PlanSet& P = (t)->plan[(t)->pending.set];                                               \
    const size_t E = 2 * (t)->pending.n;                                                    \
    const unsigned L0 = (t)->pending.l0;                                                    \
    (void)E; (void)L0; (void)P

extern "C" int imt_itree_batch_leaves(imt_itree* t, void* val0, uint32_t k_begin, uint32_t k_count) {
    IMT_PENDING(t, c);
    if (!val0 || (size_t)k_begin + k_count > E) return c->fail(IMT_ERR_RANGE, "slot range outside the batch");
    if (int rc = check_fe_ptrs(c, true, {val0})) return rc;
    launch::sweep_leaves(c->stream, P.d_pre, P.d_tab[0][1], (uint8_t*)val0, k_begin, k_count, IMT_FMT_CANONICAL, c->d_err,
                         c->coop_max_events);
    return IMT_OK;
}

extern "C" int imt_itree_batch_level(imt_itree* t, unsigned level, const void* val_in, void* val_out, uint32_t k_begin,
                                     uint32_t k_count) {
    IMT_PENDING(t, c);
    if (level >= L0) return c->fail(IMT_ERR_RANGE, "level %u >= l0 %u", level, L0);
    if (!val_in || !val_out || (size_t)k_begin + k_count > E) return c->fail(IMT_ERR_RANGE, "slot range outside the batch");
    if (int rc = check_fe_ptrs(c, true, {val_in, val_out})) return rc;
    const size_t o = (size_t)level * P.cap_events;
    launch::sweep_level(c->stream, (const uint8_t*)val_in, (uint8_t*)val_out, P.d_from + o, P.d_sibsrc + o, P.d_nodeb + o,
                        P.d_timen + o, t->d_nodes + t->h_off[level] * 32, t->h_len[level], c->d_zero + (size_t)level * 32,
                        k_begin, k_count, nullptr, nullptr, launch::SibLayout{0, 0}, level, IMT_FMT_DEVICE,
                        c->coop_max_events);
    return IMT_OK;
}

extern "C" int imt_itree_batch_top(imt_itree* t, const void* val_l0, uint32_t e_begin, uint32_t e_count, void* roots,
                                   void* top_path) {
    IMT_PENDING(t, c);
    if (!val_l0 || !roots || !top_path || (size_t)e_begin + e_count > E)
        return c->fail(IMT_ERR_RANGE, "event range outside the batch");
    if (int rc = check_fe_ptrs(c, true, {val_l0, roots, top_path})) return rc;
    // levels [L0, depth) for this rank's events, ping-ponging through the plan's value scratch; the rank whose range
    // holds the last event fills top_path with that event's node at every level >= L0
    const uint8_t* vin = (const uint8_t*)val_l0;
    uint8_t* tp = (uint8_t*)top_path;
    for (unsigned l = L0; l < t->depth; l++) {
        uint8_t* vout = P.d_val[l & 1];
        launch::sweep_upper(c->stream, vin, vout, c->d_zero + (size_t)l * 32, e_begin, e_count, (uint32_t)E - 1,
                            l == L0 ? tp : nullptr, tp + (size_t)(l + 1 - L0) * 32, nullptr, nullptr,
                            launch::SibLayout{0, 0}, l, IMT_FMT_DEVICE, c->coop_max_events);
        vin = vout;
    }
    launch::emit_roots(c->stream, vin, e_begin, e_count, (uint32_t)E, nullptr, nullptr, nullptr, IMT_FMT_DEVICE,
                       (uint8_t*)roots, L0 == t->depth ? tp : nullptr);
    return IMT_OK;
}

extern "C" int imt_itree_batch_extract(imt_itree* t, const void* const* val_levels, const void* roots, uint32_t ins_begin,
                                       uint32_t ins_count, const imt_insert_out* out, unsigned flags) {
    IMT_PENDING(t, c);
    if (!val_levels || !roots || !out) return c->fail(IMT_ERR_ARG, "null argument");
    if (!(flags & IMT_DEVICE_PTRS)) return c->fail(IMT_ERR_ARG, "imt_itree_batch_extract takes device pointers");
    if ((size_t)ins_begin + ins_count > t->pending.n) return c->fail(IMT_ERR_RANGE, "insertion range outside the batch");
    if (ins_count == 0) return IMT_OK;
    if (int rc = check_fe_ptrs(c, true, {roots})) return rc;
    if (int rc = check_out_ptrs(c, true, out)) return rc;
    for (unsigned l = 0; l <= L0; l++)
        if (int rc = check_fe_ptrs(c, true, {val_levels[l]})) return rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    hipStream_t s = c->stream;
    IMT_HIP(c, hipMemcpyAsync(P.d_valptr, val_levels, (size_t)(L0 + 1) * sizeof(void*), hipMemcpyHostToDevice, s));
    launch::ExtractParams p{};
    p.val = P.d_valptr;
    p.slot = P.d_slot;
    p.sibsrc = P.d_sibsrc;
    p.node_below = P.d_nodeb;
    p.stride = P.cap_events;
    p.tree_nodes = t->d_nodes;
    p.tree_off = t->d_off;
    p.tree_len = t->d_len;
    p.zero = c->d_zero;
    p.roots = (const uint8_t*)roots;
    p.l0 = L0;
    p.depth = t->depth;
    p.ins_begin = ins_begin;
    p.ins_count = ins_count;
    p.n_total = (uint32_t)t->pending.n;
    p.old_root = (uint8_t*)out->old_root;
    p.interim_root = (uint8_t*)out->interim_root;
    p.new_root = (uint8_t*)out->new_root;
    p.low_sib = (uint8_t*)out->low_sib;
    p.new_sib = (uint8_t*)out->new_sib;
    p.lay = (flags & IMT_SIB_ITEM_MAJOR) ? launch::SibLayout{1, t->global_depth} : launch::SibLayout{ins_count, 1};
    p.fmt_out = fmt;
    launch::extract(s, p);
    if (out->low_index)
        IMT_HIP(c, hipMemcpyAsync(out->low_index, P.ws.o_low + ins_begin, (size_t)ins_count * 8, hipMemcpyDeviceToDevice, s));
    if (out->is_largest)
        IMT_HIP(c, hipMemcpyAsync(out->is_largest, P.ws.o_largest + ins_begin, ins_count, hipMemcpyDeviceToDevice, s));
    if (out->low_leaf)
        launch::convert(s, P.ws.o_lowleaf + (size_t)ins_begin * 96, (uint8_t*)out->low_leaf, (size_t)ins_count * 3,
                        IMT_FMT_CANONICAL, fmt, c->d_err);
    if (out->new_leaf)
        launch::convert(s, P.ws.o_newleaf + (size_t)ins_begin * 96, (uint8_t*)out->new_leaf, (size_t)ins_count * 3,
                        IMT_FMT_CANONICAL, fmt, c->d_err);
    return IMT_OK;
}

extern "C" int imt_itree_batch_abort(imt_itree* t) {
    if (!t) return IMT_ERR_ARG;
    // nothing of an open batch has touched the stored tree, the committed index or the size: rows
    // [size, size + n) of the value array and the spare sorted index are simply overwritten next time
    t->pending.active = false;
    return IMT_OK;
}

extern "C" int imt_itree_batch_end(imt_itree* t, const void* const* val_levels, const void* top_path) {
    IMT_PENDING(t, c);
    if (!val_levels || !top_path) return c->fail(IMT_ERR_ARG, "null argument");
    if (int rc = check_fe_ptrs(c, true, {top_path})) return rc;
    for (unsigned l = 0; l < L0; l++)
        if (int rc = check_fe_ptrs(c, true, {val_levels[l]})) return rc;
    hipStream_t s = c->stream;
    for (unsigned l = 0; l < L0; l++) {
        const size_t o = (size_t)l * P.cap_events;
        launch::writeback(s, (const uint8_t*)val_levels[l], P.d_from + o, P.d_nodeb + o, t->d_nodes + t->h_off[l] * 32,
                          (uint32_t)E);
    }
    launch::store_top_path(s, (const uint8_t*)top_path, t->d_nodes, t->d_off, L0, t->depth);
    IMT_HIP(c, hipMemcpyAsync(P.d_root, t->d_nodes + t->h_off[t->depth] * 32, 32, hipMemcpyDeviceToDevice, s));
    P.has_root = true;
    IMT_HIP(c, hipEventRecord(P.done, s));
    P.in_flight = true;
    P.pipelined = false;
    P.l0 = L0;
    t->sorted_cur ^= 1;
    t->size += t->pending.n;
    t->mirror_valid = false;
    t->cur = (t->cur + 1) % imt_itree::NSETS;
    t->batch_no++;
    t->pending.active = false;
    return IMT_OK;
}

// ------------------------------------------------------------------------------------
// e: one tree on several GPUs, sequential semantics, time-sliced (imt_itree_slice_*)
// ------------------------------------------------------------------------------------
// A step's insertions are cut into consecutive slices, one per GPU.  A slice is an ordinary batch for the GPU that
// hashes it -- same plan, same kernels, same 2 + 2 * depth hashes per insertion, proofs and roots written directly --
// except that its level sweeps are issued one UNIT at a time on the caller's stream, so that between units the caller
// can exchange with the other GPUs what each unit wrote back to the stored tree (the "payload" of a unit) and apply
// theirs (imt_itree_slice_apply).  Unit 0 = the leaf hashes, unit 1 + l = level l -> l + 1.  A slice's level l may run
// once every earlier slice's level l has been applied here, and nothing later: the caller's schedule (sharded.py,
// SlicedIndexedTree) guarantees both.  The values other GPUs hash only enter this replica's index (sort + merge, no
// events, no hashing).
namespace {
constexpr size_t SLICE_HDR = 128;      // payload header: node at l0 (32 B), node above (32 B), root (32 B), count (4 B), pad
constexpr size_t SLICE_COUNT_AT = 96;

// (node, value) pairs a slice of n insertions into a tree of size_before leaves can write back at level l: one per
// event at most, and no more than the level has nodes under the leaves in use
inline size_t slice_pairs(uint64_t size_before, size_t n, unsigned l) {
    const uint64_t nodes = l < 63 ? ((size_before + n - 1) >> l) + 1 : 1;
    return (size_t)std::min<uint64_t>(2 * (uint64_t)n, nodes);
}
// bytes of the payload of unit 1 + l: header, values, node ids, rounded to 16
inline size_t slice_unit_bytes(uint64_t size_before, size_t n, unsigned unit, unsigned depth) {
    if (unit == 0) return SLICE_HDR;
    const unsigned l = unit - 1;
    const unsigned L0 = std::min(ceil_log2(size_before + n), depth);
    if (l >= L0) return SLICE_HDR;
    return (SLICE_HDR + 36 * slice_pairs(size_before, n, l) + 15) & ~(size_t)15;
}

int fws_reserve(imt_itree* t, size_t n) {
    imt_ctx* c = t->ctx;
    prep::Workspace& w = t->fws;
    const size_t tmp_need = prep::temp_bytes_needed(n, t->cap);
    if (w.cap_n >= n && w.tmp_bytes >= tmp_need) return IMT_OK;
    IMT_HIP(c, hipStreamSynchronize(t->up_stream));
    for (void* q : {(void*)w.iota, (void*)w.bsorted, (void*)w.gap, (void*)w.st, w.tmp})
        if (q) hipFree(q);
    w = prep::Workspace();
    const size_t N = n + n / 4;
    hipError_t e = hipSuccess;
    auto A = [&](void** ptr, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(ptr, bytes);
    };
    A((void**)&w.iota, N * 4);
    A((void**)&w.bsorted, N * 4);
    A((void**)&w.gap, N * 4);
    A((void**)&w.st, N * 4);
    w.tmp_bytes = prep::temp_bytes_needed(N, t->cap);
    A(&w.tmp, w.tmp_bytes);
    if (e != hipSuccess) {
        for (void* q : {(void*)w.iota, (void*)w.bsorted, (void*)w.gap, (void*)w.st, w.tmp})
            if (q) hipFree(q);
        w = prep::Workspace();
        return c->hip_fail(e, "hipMalloc(slice index workspace)");
    }
    w.cap_n = N;
    return IMT_OK;
}
}  // namespace

extern "C" size_t imt_itree_slice_payload_bytes(size_t n) { return (SLICE_HDR + 72 * n + 15) & ~(size_t)15; }

extern "C" size_t imt_itree_slice_unit_bytes(const imt_itree* t, uint64_t size_before, size_t n, unsigned unit) {
    if (!t || n == 0 || unit > t->depth) return 0;
    return slice_unit_bytes(size_before, n, unit, t->depth);
}

// A host wait with a time limit (limit_ms <= 0: wait as long as it takes): the sliced mode's waits stand behind work
// that stands behind collectives, i.e. behind other ranks -- a peer that died would otherwise hang the caller for good.
// IMT_ERR_TIMEOUT leaves the stream as it is (still busy); the caller gives the world up.
template <class Query, class Sync>
static int bounded_wait(imt_ctx* c, double limit_ms, Query query, Sync sync, const char* what) {
    if (limit_ms <= 0) {
        const hipError_t e = sync();
        return e == hipSuccess ? IMT_OK : c->hip_fail(e, what);
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = query();
        if (e == hipSuccess) return IMT_OK;
        if (e != hipErrorNotReady) return c->hip_fail(e, what);
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));     // ~0.1 ms of pure polling first
        if ((spins & 63) == 63 &&
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > limit_ms)
            return c->fail(IMT_ERR_TIMEOUT, "gave up after %.0f ms waiting for %s", limit_ms, what);
    }
}

extern "C" int imt_itree_slice_prepare(imt_itree* t, const void* vals, size_t n_before, size_t n_own, size_t n_after,
                                       const imt_insert_out* out, unsigned flags, int* slice_out, uint32_t* l0_out) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    if (!vals || n_own == 0 || !slice_out) return c->fail(IMT_ERR_ARG, "null / empty slice");
    if (!(flags & IMT_DEVICE_PTRS)) return c->fail(IMT_ERR_ARG, "imt_itree_slice_prepare takes device pointers");
    if ((flags & IMT_FMT_MASK) == 3) return c->fail(IMT_ERR_ARG, "unknown field-element format");
    if (t->index_base || t->part_mod > 1) return c->fail(IMT_ERR_ARG, "a placed / partitioned tree is not sliced");
    if (t->pending.active) return c->fail(IMT_ERR_ARG, "a sharded batch is open (imt_itree_batch_end first)");
    const size_t n_all = n_before + n_own + n_after;
    if (n_all > ((size_t)1 << 30)) return c->fail(IMT_ERR_RANGE, "step too large");
    const uint64_t M0 = t->size;
    if (M0 + n_all > t->cap) return c->fail(IMT_ERR_FULL, "tree capacity %llu exceeded", (unsigned long long)t->cap);
    int rc = c->set_device();
    if (rc) return rc;
    if ((rc = ensure_device_index(t))) return rc;
    if ((rc = check_fe_ptrs(c, true, {vals})) || (rc = check_out_ptrs(c, true, out))) return rc;
    const unsigned fmt = flags & IMT_FMT_MASK;
    const int set = t->cur;
    PlanSet& P = t->plan[set];
    if (P.open) return c->fail(IMT_ERR_ARG, "too many slices prepared ahead (%d plan sets)", imt_itree::NSETS);
    if (P.in_flight) {
        const auto w0 = std::chrono::steady_clock::now();
        rc = bounded_wait(c, t->slice_wait_limit_ms, [&] { return hipEventQuery(P.done); }, [&] { return hipEventSynchronize(P.done); },
                          "the plan set's previous slice");
        const double waited = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
        t->slice_wait_ms += waited;
        t->slice_backpressure_ms += waited;
        if (rc) return rc;
        P.in_flight = false;
    }
    if ((rc = plan_reserve(c, P, 2 * n_own, t->depth, t->cap))) return rc;
    if ((rc = reserve_all_plans(t, 2 * n_own))) return rc;
    if (n_before || n_after)
        if ((rc = fws_reserve(t, std::max(n_before, n_after)))) return rc;
    // The side stream, or a stream the sliced schedule names (imt_itree_set_slice_prep_stream: the NEW round's own stream,
    // whose hardware queue is idle -- the side stream shares one with some round's stream, and a hardware queue runs what
    // it holds in submission order: the preparation then stood behind a whole period of that round's hash kernels and the
    // host's wait for the verdict lasted until the GPU had run dry).  Preparations of consecutive steps are ordered by the
    // host: each is waited for (the verdict) before the call returns.
    hipStream_t ps = t->slice_prep_stream ? t->slice_prep_stream : t->up_stream;
    if (t->slice_prep_stream) IMT_HIP(c, hipStreamSynchronize(t->up_stream));      // whatever an earlier call left there
    if (!(flags & IMT_INPUTS_READY)) {
        IMT_HIP(c, hipEventRecord(t->in_mark, c->stream));
        IMT_HIP(c, hipStreamWaitEvent(ps, t->in_mark, 0));
    }
    IMT_HIP(c, hipMemsetAsync(P.ws.err, 0, sizeof(int), ps));
    const uint8_t* d_vals = (const uint8_t*)vals;
    if (fmt != IMT_FMT_CANONICAL) {
        if (t->canon_all_cap < n_all * 32) {
            IMT_HIP(c, hipStreamSynchronize(ps));
            if (t->d_canon_all) hipFree(t->d_canon_all);
            t->d_canon_all = nullptr;
            t->canon_all_cap = 0;
            IMT_HIP(c, hipMalloc((void**)&t->d_canon_all, n_all * 40));
            t->canon_all_cap = n_all * 40;
        }
        launch::convert(ps, d_vals, t->d_canon_all, n_all, fmt, IMT_FMT_CANONICAL, P.ws.err);
        d_vals = t->d_canon_all;
    }
    // ---- the index, in the order of the step: the slices before this one, this one (with its events), those after.
    // Up to three merges, none of which may write the committed index (a refused step leaves the tree as it was): they
    // go committed -> ... -> spare, through a third buffer when there are two or three of them.
    if (!t->d_sorted_extra) IMT_HIP(c, hipMalloc((void**)&t->d_sorted_extra, t->cap * 4));
    const int n_merges = 1 + (n_before ? 1 : 0) + (n_after ? 1 : 0);
    uint32_t* const committed = t->d_sorted[t->sorted_cur];
    uint32_t* const spare = t->d_sorted[t->sorted_cur ^ 1];
    uint32_t* chain[4] = {committed, spare, spare, spare};
    if (n_merges == 2) chain[1] = t->d_sorted_extra;
    if (n_merges == 3) chain[2] = t->d_sorted_extra;
    int link = 0;
    uint64_t M = M0;
    t->fws.err = P.ws.err;
    t->fws.part_mod = t->fws.part_res = 0;
    P.ws.part_mod = P.ws.part_res = 0;
    if (n_before) {
        IMT_HIP(c, prep::index_only(ps, t->fws, d_vals, t->d_val, chain[link], chain[link + 1], (uint32_t)M,
                                    (uint32_t)n_before));
        link++;
        M += n_before;
    }
    const uint64_t M_own = M;
    IMT_HIP(c, prep::run(ps, P.ws, d_vals + n_before * 32, t->d_val, chain[link], chain[link + 1], (uint32_t)M,
                         (uint32_t)n_own, 0, P.d_pre, P.d_tab[0][0], P.d_tab[0][1], P.d_tab[0][2], P.d_tab[0][3],
                         out ? out->low_index : nullptr, out ? out->is_largest : nullptr,
                         out ? (uint8_t*)out->low_leaf : nullptr, out ? (uint8_t*)out->new_leaf : nullptr));
    link++;
    M += n_own;
    if (n_after) {
        IMT_HIP(c, prep::index_only(ps, t->fws, d_vals + (n_before + n_own) * 32, t->d_val, chain[link], chain[link + 1],
                                    (uint32_t)M, (uint32_t)n_after));
        link++;
    }
    IMT_HIP(c, hipMemcpyAsync(t->h_err_pin, P.ws.err, sizeof(int), hipMemcpyDeviceToHost, ps));
    // ---- this slice's index phase (no hashing) on the side stream as well ----
    const size_t E = 2 * n_own;
    const unsigned L0 = std::min(ceil_log2(M_own + n_own), t->depth);
    for (unsigned l = 0; l < L0; l++) {
        const int a = l & 1, b = a ^ 1;
        const uint32_t* time_in = l == 0 ? P.d_tab[0][1] : P.d_timen + (size_t)(l - 1) * P.cap_events;
        sweep::LevelTable in{P.d_tab[a][0], time_in, P.d_tab[a][2], P.d_tab[a][3]};
        sweep::LevelOut o{P.d_tab[b][0], P.d_timen + (size_t)l * P.cap_events, P.d_tab[b][2], P.d_tab[b][3],
                          P.d_from + (size_t)l * P.cap_events, P.d_sibsrc + (size_t)l * P.cap_events,
                          P.d_nodeb + (size_t)l * P.cap_events, nullptr};
        launch::merge_level(ps, in, o, (uint32_t)E);
    }
    if (out && fmt != IMT_FMT_CANONICAL) {
        if (out->low_leaf) launch::convert(ps, (uint8_t*)out->low_leaf, (uint8_t*)out->low_leaf, n_own * 3, IMT_FMT_CANONICAL, fmt, c->d_err);
        if (out->new_leaf) launch::convert(ps, (uint8_t*)out->new_leaf, (uint8_t*)out->new_leaf, n_own * 3, IMT_FMT_CANONICAL, fmt, c->d_err);
    }
    // The slice's units run on streams the caller names and read / write the stored levels: they must come behind
    // every ORDINARY batch still in flight on this tree (pipelined or on the context's stream), whose sweeps write the
    // same levels.  Unit 0 waits for prep_done, so ordering the side stream behind those batches orders the units.
    // Earlier slices are not waited for here: the sliced schedule orders slices among themselves level by level.
    for (auto& pl : t->plan)
        if (&pl != &P && pl.in_flight && !pl.sliced) IMT_HIP(c, hipStreamWaitEvent(ps, pl.done, 0));
    IMT_HIP(c, hipEventRecord(P.prep_done, ps));
    {
        const auto w0 = std::chrono::steady_clock::now();
        rc = bounded_wait(c, t->slice_wait_limit_ms, [&] { return hipStreamQuery(ps); }, [&] { return hipStreamSynchronize(ps); },
                          "the step's preparation (its value check)");
        t->slice_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
        if (rc) return rc;
    }
    const int perr = *t->h_err_pin;      // the same verdict on every GPU: they all see all values of the step
    if (perr & prep::ERR_NONCANONICAL) return c->fail(IMT_ERR_NONCANONICAL, "a value is not reduced (>= p)");
    if (perr & prep::ERR_ZERO) return c->fail(IMT_ERR_VALUE, "value 0 cannot be inserted");
    if (perr & prep::ERR_DUPLICATE) return c->fail(IMT_ERR_VALUE, "duplicate value (inside the step or already in the tree)");
    // ---- commit the index; the hashing follows unit by unit ----
    t->sorted_cur ^= 1;          // chain[n_merges] == spare
    t->size = M0 + n_all;
    t->mirror_valid = false;
    P.open = true;
    P.sliced = true;
    P.slice_n = n_own;
    P.slice_out = out ? *out : imt_insert_out{};
    P.slice_fmt = fmt;
    P.slice_next_unit = 0;
    P.slice_size_before = M_own;
    P.slice_lay = (flags & IMT_SIB_ITEM_MAJOR) ? launch::SibLayout{1, t->depth} : launch::SibLayout{n_own, 1};
    P.l0 = L0;
    P.has_root = false;
    t->cur = (t->cur + 1) % imt_itree::NSETS;
    t->batch_no++;
    *slice_out = set;
    if (l0_out) *l0_out = L0;
    return IMT_OK;
}

extern "C" int imt_itree_slice_unit(imt_itree* t, int slice, unsigned unit, void* payload, void* hip_stream) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    if (slice < 0 || slice >= imt_itree::NSETS || !t->plan[slice].open) return c->fail(IMT_ERR_ARG, "no such open slice");
    PlanSet& P = t->plan[slice];
    if (unit != P.slice_next_unit) return c->fail(IMT_ERR_ARG, "slice unit %u out of order (next is %u)", unit, P.slice_next_unit);
    int rc = c->set_device();
    if (rc) return rc;
    if ((rc = check_fe_ptrs(c, true, {payload}))) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const size_t n = P.slice_n, E = 2 * n;
    const unsigned L0 = P.l0, depth = t->depth, fmt = P.slice_fmt;
    const imt_insert_out& o = P.slice_out;
    uint8_t* pl = (uint8_t*)payload;
    uint8_t* root_node = t->d_nodes + t->h_off[depth] * 32;
    P.slice_next_unit = unit + 1;
    if (unit == 0) {
        IMT_HIP(c, hipStreamWaitEvent(s, P.prep_done, 0));
        int pf = c->prof_begin(IMT_PROF_LEAVES, s);
        launch::sweep_leaves(s, P.d_pre, P.d_tab[0][1], P.d_val[0], 0, (uint32_t)E, IMT_FMT_CANONICAL, c->d_err, c->coop_max_events);
        c->prof_end(pf, s);
        return IMT_OK;
    }
    const unsigned l = unit - 1;
    uint8_t* g_old = (uint8_t*)o.old_root;
    if (l < L0) {
        const size_t off = (size_t)l * P.cap_events;
        const uint8_t* vin = P.d_val[l & 1];
        int pf = c->prof_begin(IMT_PROF_LEVEL, s);
        launch::sweep_level(s, vin, P.d_val[(l & 1) ^ 1], P.d_from + off, P.d_sibsrc + off, P.d_nodeb + off, P.d_timen + off,
                            t->d_nodes + t->h_off[l] * 32, t->h_len[l], c->d_zero + (size_t)l * 32, 0, (uint32_t)E,
                            (uint8_t*)o.low_sib, (uint8_t*)o.new_sib, P.slice_lay, l, fmt, c->coop_max_events);
        c->prof_end(pf, s);
        pf = c->prof_begin(IMT_PROF_WRITEBACK, s);
        if (pl) {       // the write-back, and what the other replicas need to repeat it: (node, value) pairs, packed
            const size_t cap = slice_pairs(P.slice_size_before, n, l);
            IMT_HIP(c, hipMemsetAsync(pl + SLICE_COUNT_AT, 0, 4, s));
            // the pack is the last kernel of this unit (unless it is the round's last, or a profile brackets it): it can
            // signal the tick's event itself
            hipEvent_t tail = (l + 1 != depth && pf < 0) ? t->slice_tail_event : nullptr;
            launch::pack_writeback(s, vin, P.d_from + off, P.d_nodeb + off, (uint32_t)E, t->d_nodes + t->h_off[l] * 32,
                                   pl + SLICE_HDR, (uint32_t*)(pl + SLICE_HDR + cap * 32), (uint32_t*)(pl + SLICE_COUNT_AT),
                                   (uint32_t)cap, tail);
            if (tail) t->slice_tail_attached = true;
        } else {
            launch::writeback(s, vin, P.d_from + off, P.d_nodeb + off, t->d_nodes + t->h_off[l] * 32, (uint32_t)E);
        }
        c->prof_end(pf, s);
    } else {
        if (l + 1 == depth && g_old)
            launch::convert(s, root_node, g_old, 1, IMT_FMT_DEVICE, fmt, c->d_err);
        uint8_t* node_in = l == L0 ? t->d_nodes + t->h_off[l] * 32 : nullptr;
        uint8_t* node_out = t->d_nodes + t->h_off[l + 1] * 32;
        int pf = c->prof_begin(IMT_PROF_TOP, s);
        launch::sweep_upper(s, P.d_val[l & 1], P.d_val[(l & 1) ^ 1], c->d_zero + (size_t)l * 32, 0, (uint32_t)E, (uint32_t)E - 1,
                            node_in, node_out, (uint8_t*)o.low_sib, (uint8_t*)o.new_sib, P.slice_lay, l, fmt, c->coop_max_events);
        c->prof_end(pf, s);
        if (pl) {
            if (node_in) IMT_HIP(c, hipMemcpyAsync(pl, node_in, 32, hipMemcpyDeviceToDevice, s));
            IMT_HIP(c, hipMemcpyAsync(pl + 32, node_out, 32, hipMemcpyDeviceToDevice, s));
        }
    }
    if (l + 1 == depth) {       // the last unit: roots of every event, the batch's root, done
        if (L0 == depth && g_old) launch::convert(s, root_node, g_old, 1, IMT_FMT_DEVICE, fmt, c->d_err);
        launch::emit_roots(s, P.d_val[depth & 1], 0, (uint32_t)E, (uint32_t)E, g_old, (uint8_t*)o.interim_root,
                           (uint8_t*)o.new_root, fmt, nullptr, L0 == depth ? root_node : nullptr);
        if (pl && L0 == depth) IMT_HIP(c, hipMemcpyAsync(pl + 64, root_node, 32, hipMemcpyDeviceToDevice, s));
        P.has_root = false;         // the root after THIS slice is a mid-step root on every rank but the last: not offered
        IMT_HIP(c, hipEventRecord(P.done, s));
        P.in_flight = true;
        P.pipelined = true;         // not on the context's stream: join_top orders that stream behind it
        P.l0 = 0;                   // no per-level events were recorded: a pipelined batch that follows waits for `done`
        t->pipe_pending = true;
        P.open = false;
    }
    return IMT_OK;
}

extern "C" int imt_itree_slice_apply(imt_itree* t, uint64_t size_before, size_t n, unsigned unit, const void* payload,
                                     void* hip_stream) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    if (!payload || n == 0) return c->fail(IMT_ERR_ARG, "null / empty payload");
    if (unit > t->depth) return c->fail(IMT_ERR_RANGE, "unit %u beyond depth %u", unit, t->depth);
    if (size_before + n > t->cap) return c->fail(IMT_ERR_RANGE, "slice outside the tree's capacity");
    if (unit == 0) return IMT_OK;          // leaf hashes: nothing is stored yet
    int rc = c->set_device();
    if (rc) return rc;
    if ((rc = check_fe_ptrs(c, true, {payload}))) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const unsigned depth = t->depth, l = unit - 1;
    const unsigned L0 = std::min(ceil_log2(size_before + n), depth);
    const uint8_t* pl = (const uint8_t*)payload;
    if (l < L0) {
        const size_t cap = slice_pairs(size_before, n, l);
        launch::apply_packed(s, pl + SLICE_HDR, (const uint32_t*)(pl + SLICE_HDR + cap * 32), (const uint32_t*)(pl + SLICE_COUNT_AT),
                             (uint32_t)cap, t->d_nodes + t->h_off[l] * 32, t->h_len[l]);
    } else {
        if (l == L0) IMT_HIP(c, hipMemcpyAsync(t->d_nodes + t->h_off[l] * 32, pl, 32, hipMemcpyDeviceToDevice, s));
        IMT_HIP(c, hipMemcpyAsync(t->d_nodes + t->h_off[l + 1] * 32, pl + 32, 32, hipMemcpyDeviceToDevice, s));
    }
    if (l + 1 == depth && L0 == depth)
        IMT_HIP(c, hipMemcpyAsync(t->d_nodes + t->h_off[depth] * 32, pl + 64, 32, hipMemcpyDeviceToDevice, s));
    return IMT_OK;
}

extern "C" int imt_itree_slice_apply_gathered(imt_itree* t, const void* gathered, size_t stride, size_t count,
                                              const uint64_t* size_before, const uint64_t* n, const int32_t* unit,
                                              void* hip_stream) {
    if (!t) return IMT_ERR_ARG;
    if (!gathered || !size_before || !n || !unit) return t->ctx->fail(IMT_ERR_ARG, "null argument");
    imt_ctx* c = t->ctx;
    int rc = c->set_device();
    if (rc) return rc;
    if ((rc = check_fe_ptrs(c, true, {gathered})) || (stride & 15u))
        return rc ? rc : c->fail(IMT_ERR_ARG, "payload stride must be a multiple of 16 bytes");
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const unsigned depth = t->depth;
    launch::ApplyJobs jobs{};
    jobs.poison = t->slice_poison;
    for (size_t r = 0; r < count; r++) {
        if (unit[r] < 0) continue;
        if ((unsigned)unit[r] > depth) return c->fail(IMT_ERR_RANGE, "unit %d beyond depth %u", unit[r], depth);
        if (n[r] == 0 || size_before[r] + n[r] > t->cap) return c->fail(IMT_ERR_RANGE, "slice outside the tree's capacity");
        if (unit[r] == 0) continue;
        // the kernel reads the counter and the node ids at offsets computed from (size_before, n, unit): a payload
        // slot shorter than that would make it read the neighbouring payload
        if (count > 1 && stride < slice_unit_bytes(size_before[r], (size_t)n[r], (unsigned)unit[r], depth))
            return c->fail(IMT_ERR_ARG, "payload stride %zu is smaller than the %zu bytes unit %d of slot %zu uses", stride,
                           slice_unit_bytes(size_before[r], (size_t)n[r], (unsigned)unit[r], depth), unit[r], r);
        const unsigned l = (unsigned)unit[r] - 1;
        const unsigned L0 = std::min(ceil_log2(size_before[r] + n[r]), depth);
        launch::ApplyJobs::Job& j = jobs.j[jobs.n_jobs++];
        j = launch::ApplyJobs::Job{};
        j.payload = (const uint8_t*)gathered + r * stride;
        if (l < L0) {
            j.pairs = 1;
            j.cap = (uint32_t)slice_pairs(size_before[r], (size_t)n[r], l);
            j.tree_l = t->d_nodes + t->h_off[l] * 32;
            j.len_l = t->h_len[l];
        } else {
            if (l == L0) j.node_in = t->d_nodes + t->h_off[l] * 32;
            j.node_out = t->d_nodes + t->h_off[l + 1] * 32;
        }
        if (l + 1 == depth && L0 == depth) {
            if (j.pairs) {      // pairs and the root from one payload: the root goes as a job of its own
                if (jobs.n_jobs == 16) { launch::apply_gathered(s, jobs); jobs.n_jobs = 0; }
                launch::ApplyJobs::Job& k = jobs.j[jobs.n_jobs++];
                k = launch::ApplyJobs::Job{};
                k.payload = (const uint8_t*)gathered + r * stride;
                k.root = t->d_nodes + t->h_off[depth] * 32;
            } else {
                j.root = t->d_nodes + t->h_off[depth] * 32;
            }
        }
        if (jobs.n_jobs >= 15) { launch::apply_gathered(s, jobs); jobs.n_jobs = 0; }
    }
    if (jobs.n_jobs && t->slice_tail_event) {
        launch::apply_gathered(s, jobs, t->slice_tail_event);
        t->slice_tail_attached = true;
    } else {
        launch::apply_gathered(s, jobs);
    }
    return IMT_OK;
}

// ------------------------------------------------------------------------------------
// e: the tree as one subtree of a deeper tree (leaf-index-range sharding)
// ------------------------------------------------------------------------------------
extern "C" int imt_itree_set_placement(imt_itree* t, unsigned global_depth, uint64_t subtree_index) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    if (t->size != 1 || t->batch_no != 0) return c->fail(IMT_ERR_ARG, "placement must be set on an empty tree");
    if (global_depth < t->depth || global_depth > IMT_MAX_DEPTH)
        return c->fail(IMT_ERR_RANGE, "global depth %u outside [%u, %d]", global_depth, t->depth, IMT_MAX_DEPTH);
    const unsigned up = global_depth - t->depth;
    if (up < 64 && (subtree_index >> up) != 0) return c->fail(IMT_ERR_RANGE, "subtree index does not fit the global tree");
    if (global_depth > 63 && subtree_index != 0) return c->fail(IMT_ERR_RANGE, "leaf indices must fit 64 bits");
    t->global_depth = global_depth;
    t->sub_index = subtree_index;
    t->index_base = t->depth < 64 ? subtree_index << t->depth : 0;
    return IMT_OK;
}

extern "C" int imt_itree_set_value_partition(imt_itree* t, uint32_t modulus, uint32_t residue) {
    if (!t) return IMT_ERR_ARG;
    if (modulus > 1 && residue >= modulus) return t->ctx->fail(IMT_ERR_ARG, "residue %u >= modulus %u", residue, modulus);
    t->part_mod = modulus;
    t->part_res = residue;
    return IMT_OK;
}

extern "C" int imt_itree_lift_batch(imt_itree* t, const void* roots_before, const void* roots_after, size_t n_subtrees,
                                    size_t n, const imt_insert_out* out, unsigned flags) {
    if (!t) return IMT_ERR_ARG;
    imt_ctx* c = t->ctx;
    if (!out) return c->fail(IMT_ERR_ARG, "null outputs");
    if ((flags & IMT_FMT_MASK) == 3) return c->fail(IMT_ERR_ARG, "unknown field-element format");
    if (n_subtrees == 0 || (n_subtrees & (n_subtrees - 1)) || n_subtrees > 65536)
        return c->fail(IMT_ERR_ARG, "n_subtrees must be a power of two (<= 65536)");
    unsigned k = 0;
    while (((size_t)1 << k) < n_subtrees) k++;
    const unsigned levels = t->global_depth - t->depth;
    if (k > levels) return c->fail(IMT_ERR_RANGE, "depth + log2(n_subtrees) exceeds the global depth");
    if (t->sub_index >= n_subtrees) return c->fail(IMT_ERR_RANGE, "this tree is subtree %llu of %zu",
                                                   (unsigned long long)t->sub_index, n_subtrees);
    if (n_subtrees > 1 && (!roots_before || !roots_after)) return c->fail(IMT_ERR_ARG, "null subtree roots");
    if (n == 0 || levels == 0) return IMT_OK;
    if (n > ((size_t)1 << 30)) return c->fail(IMT_ERR_RANGE, "batch too large");
    int rc = c->set_device();
    if (rc) return rc;
    const bool dev = flags & IMT_DEVICE_PTRS;
    const unsigned fmt = flags & IMT_FMT_MASK;
    hipStream_t s = c->stream;
    if ((rc = check_fe_ptrs(c, dev, {roots_before, roots_after})) || (rc = check_out_ptrs(c, dev, out))) return rc;
    if (!dev && (rc = c->clear_err())) return rc;
    size_t slot = 2;
    auto scratch = [&](size_t bytes) { return (uint8_t*)c->dev_scratch(slot++, bytes); };
    // ---- the `levels` siblings above this subtree's root: constant for the whole batch ----
    uint8_t* tree_buf = scratch(2 * n_subtrees * 32);       // dense tree over the mixed roots
    uint8_t* top = scratch((size_t)levels * 32);            // device format
    if (!tree_buf || !top) return IMT_ERR_HIP;
    if (n_subtrees > 1) {
        const uint8_t *d_b = (const uint8_t*)roots_before, *d_a = (const uint8_t*)roots_after;
        if (!dev) {
            uint8_t* up = scratch(2 * n_subtrees * 32);
            if (!up) return IMT_ERR_HIP;
            IMT_HIP(c, hipMemcpyAsync(up, roots_before, n_subtrees * 32, hipMemcpyHostToDevice, s));
            IMT_HIP(c, hipMemcpyAsync(up + n_subtrees * 32, roots_after, n_subtrees * 32, hipMemcpyHostToDevice, s));
            d_b = up;
            d_a = up + n_subtrees * 32;
        }
        // subtrees left of this one (lower leaf indices) are taken AFTER the step, those right of it BEFORE:
        // within a step the insertions of subtree 0 come first, then subtree 1's, ...
        launch::mix_roots(s, d_b, d_a, tree_buf, (uint32_t)n_subtrees, (uint32_t)t->sub_index, fmt, c->d_err);
        size_t off = 0;
        for (size_t m = n_subtrees; m > 1; m >>= 1) {
            launch::tree_level(s, tree_buf + off * 32, tree_buf + (off + m) * 32, m / 2, c->coop_max_events);
            off += m;
        }
    }
    launch::pick_top(s, tree_buf, (uint32_t)n_subtrees, (uint32_t)t->sub_index, k, levels, c->d_zero, t->depth, top);
    const uint64_t pos_bits = t->sub_index;                 // bit j: the ancestor at height depth + j is a right child
    // ---- roots, in place ----
    uint8_t* d_roots[3] = {(uint8_t*)out->old_root, (uint8_t*)out->interim_root, (uint8_t*)out->new_root};
    void* h_roots[3] = {out->old_root, out->interim_root, out->new_root};
    if (!dev)
        for (int j = 0; j < 3; j++)
            if (h_roots[j]) {
                d_roots[j] = scratch(n * 32);
                if (!d_roots[j]) return IMT_ERR_HIP;
                IMT_HIP(c, hipMemcpyAsync(d_roots[j], h_roots[j], n * 32, hipMemcpyHostToDevice, s));
            }
    launch::lift_roots(s, d_roots[0], d_roots[1], d_roots[2], (uint32_t)n, top, pos_bits, levels, fmt, c->d_err);
    // ---- sibling rows [depth, global_depth) ----
    const bool item_major = flags & IMT_SIB_ITEM_MAJOR;
    launch::SibLayout lay = item_major ? launch::SibLayout{1, t->global_depth} : launch::SibLayout{n, 1};
    if (dev) {
        launch::fill_sib_rows(s, (uint8_t*)out->low_sib, lay, t->depth, levels, (uint32_t)n, top, fmt);
        launch::fill_sib_rows(s, (uint8_t*)out->new_sib, lay, t->depth, levels, (uint32_t)n, top, fmt);
        return IMT_OK;
    }
    for (int j = 0; j < 3; j++)
        if (h_roots[j]) IMT_HIP(c, hipMemcpyAsync(h_roots[j], d_roots[j], n * 32, hipMemcpyDeviceToHost, s));
    std::vector<uint8_t> h_top((size_t)levels * 32);
    if (out->low_sib || out->new_sib) {
        uint8_t* top_fmt = scratch((size_t)levels * 32);
        if (!top_fmt) return IMT_ERR_HIP;
        launch::convert(s, top, top_fmt, levels, IMT_FMT_DEVICE, fmt, c->d_err);
        IMT_HIP(c, hipMemcpyAsync(h_top.data(), top_fmt, (size_t)levels * 32, hipMemcpyDeviceToHost, s));
    }
    if ((rc = c->sync_and_check())) return rc;
    for (void* sibv : {out->low_sib, out->new_sib}) {
        uint8_t* sib = (uint8_t*)sibv;
        if (!sib) continue;
        for (unsigned j = 0; j < levels; j++)
            for (size_t i = 0; i < n; i++)
                std::memcpy(sib + ((uint64_t)(t->depth + j) * lay.level_stride + i * lay.item_stride) * 32,
                            &h_top[(size_t)j * 32], 32);
    }
    return IMT_OK;
}
