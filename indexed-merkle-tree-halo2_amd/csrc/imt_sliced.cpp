// imt_sliced.cpp -- imt_sliced_* (include/imt.h): the multi-GPU single-list mode behind one call per step.
//
// The schedule is imt_sliced_sched.hpp (HIP-free, tested on the CPU over a symbolic backend); this file gives it HIP
// streams, events and buffers, the slice calls of one replica (imt_itree_slice_* in imt_itree.cpp), and the transports
// that do not need RCCL: in-process copies, a caller-supplied vtable, and direct peer copies between processes over HIP
// IPC handles.  The RCCL transport is imt_sliced_rccl.cpp.  The data structure is the reference's ONE sorted list
// (/root/reference/src/indexed_merkle_tree.rs:632-660, insertion i at leaf size + i: :715).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include "imt_flags.hpp"
#include "imt_itree_internal.hpp"
#include "imt_sliced_sched.hpp"
#include "imt_sliced_transport.hpp"

using namespace imt::sliced;

namespace {

// ------------------------------------------------------------------------------------------------------- options
// imt_sliced_set_option (include/imt.h).  The defaults are what the measurements of rounds 4 and 5 chose; the environment
// variables of the same names (IMT_SLICED_COMM_STREAMS, ...) are read ONCE per imt_sliced_create and override them -- they
// exist for tools/ (stream matrices, the rank emulation), not as an interface.
struct SlicedOptions {
    long comm_streams = ROUNDS;      // streams that carry the collectives (0 = the round's own stream)
    long comm_priority = 0;          // their HIP priority (0 = the rounds' pool of hardware queues)
    long round_priorities = 0;       // 0 all normal, 1 = the batch pipeline's one normal + three high, 2 all LOW, 3 all HIGH
    long apply_streams = 0;          // 1 = the other ranks' write-backs are applied on a stream of their own per round slot
    long prep_stream = 0;            // where a step's preparation is enqueued: 0 the slot's collective stream, 1 its round stream, 2 the tree's side stream
    long comm_placement = 0;         // 0: the collectives' streams on queues of their own if there are any, else on their rounds'; 1: on their rounds'; 2: own or fail
    long pools = -1;                 // 1: three priority pools (rounds HIGH, collectives LOW, preparation on the round stream); 2: rounds AND collectives HIGH; 0: one pool; -1: 1 for one process per GPU
    long verify_queues = 1;          // probe the stream -> hardware queue placement at creation and repair it
    long watchdog_ms = 120000;       // host waits inside imt_sliced_* give up after this long (0 = never)
    long timing = 0;                 // print the host's time per phase at destroy
    unsigned long set_mask = 0;      // bit `option`: named explicitly (imt_sliced_set_option(NULL, ...) or the tools' environment variable)
};
SlicedOptions g_defaults;
std::mutex g_defaults_mu;

int option_field(SlicedOptions& o, int option, long** field) {
    switch (option) {
        case IMT_SLICED_OPT_COMM_STREAMS: *field = &o.comm_streams; return IMT_OK;
        case IMT_SLICED_OPT_COMM_PRIORITY: *field = &o.comm_priority; return IMT_OK;
        case IMT_SLICED_OPT_ROUND_PRIORITIES: *field = &o.round_priorities; return IMT_OK;
        case IMT_SLICED_OPT_APPLY_STREAMS: *field = &o.apply_streams; return IMT_OK;
        case IMT_SLICED_OPT_PREP_STREAM: *field = &o.prep_stream; return IMT_OK;
        case IMT_SLICED_OPT_VERIFY_QUEUES: *field = &o.verify_queues; return IMT_OK;
        case IMT_SLICED_OPT_WATCHDOG_MS: *field = &o.watchdog_ms; return IMT_OK;
        case IMT_SLICED_OPT_TIMING: *field = &o.timing; return IMT_OK;
        case IMT_SLICED_OPT_COMM_PLACEMENT: *field = &o.comm_placement; return IMT_OK;
        case IMT_SLICED_OPT_POOLS: *field = &o.pools; return IMT_OK;
    }
    return IMT_ERR_ARG;
}
bool option_value_ok(int option, long v) {
    switch (option) {
        case IMT_SLICED_OPT_COMM_STREAMS: return v >= 0 && v <= ROUNDS;
        case IMT_SLICED_OPT_COMM_PRIORITY: return v >= -8 && v <= 8;
        case IMT_SLICED_OPT_PREP_STREAM: return v >= 0 && v <= 2;
        case IMT_SLICED_OPT_ROUND_PRIORITIES: return v >= 0 && v <= 3;
        case IMT_SLICED_OPT_COMM_PLACEMENT: return v >= 0 && v <= 2;
        case IMT_SLICED_OPT_POOLS: return v >= -1 && v <= 2;
        case IMT_SLICED_OPT_WATCHDOG_MS: return v >= 0;
        default: return v == 0 || v == 1;
    }
}
SlicedOptions effective_options() {
    SlicedOptions o;
    {
        std::lock_guard<std::mutex> lk(g_defaults_mu);
        o = g_defaults;
    }
    auto env = [](const char* name, int option, SlicedOptions& oo) {
        const char* e = getenv(name);
        if (!e || !*e) return;
        long v = atol(e);
        if (option == IMT_SLICED_OPT_PREP_STREAM) v = !strcmp(e, "side") ? 2 : !strcmp(e, "round") ? 1 : !strcmp(e, "comm") ? 0 : v;
        if (option == IMT_SLICED_OPT_ROUND_PRIORITIES) v = !strcmp(e, "pipe") ? 1 : !strcmp(e, "low") ? 2 : !strcmp(e, "high") ? 3 : v;
        long* f = nullptr;
        if (option_value_ok(option, v) && option_field(oo, option, &f) == IMT_OK) {
            *f = v;
            oo.set_mask |= 1ul << option;
        }
    };
    env("IMT_SLICED_COMM_STREAMS", IMT_SLICED_OPT_COMM_STREAMS, o);
    env("IMT_SLICED_COMM_PRIO", IMT_SLICED_OPT_COMM_PRIORITY, o);
    env("IMT_SLICED_ROUND_PRIO", IMT_SLICED_OPT_ROUND_PRIORITIES, o);
    env("IMT_SLICED_APPLY_STREAMS", IMT_SLICED_OPT_APPLY_STREAMS, o);
    env("IMT_SLICED_PREP_STREAM", IMT_SLICED_OPT_PREP_STREAM, o);
    env("IMT_SLICED_VERIFY_QUEUES", IMT_SLICED_OPT_VERIFY_QUEUES, o);
    env("IMT_SLICED_WATCHDOG_MS", IMT_SLICED_OPT_WATCHDOG_MS, o);
    env("IMT_SLICED_TIMING", IMT_SLICED_OPT_TIMING, o);
    env("IMT_SLICED_COMM_PLACEMENT", IMT_SLICED_OPT_COMM_PLACEMENT, o);
    env("IMT_SLICED_POOLS", IMT_SLICED_OPT_POOLS, o);
    return o;
}

// a host wait with a time limit: poll `query` (hipSuccess / hipErrorNotReady) until it is done or limit_ms have passed
template <class Query>
int bounded_wait(imt_ctx* ctx, long limit_ms, Query query, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = query();
        if (e == hipSuccess) return IMT_OK;
        if (e != hipErrorNotReady) return ctx->hip_fail(e, what);
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (limit_ms > 0 && (spins & 63) == 63 &&
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > (double)limit_ms)
            return ctx->fail(IMT_ERR_TIMEOUT, "gave up after %ld ms waiting for %s (IMT_SLICED_OPT_WATCHDOG_MS)", limit_ms, what);
    }
}

// ------------------------------------------------------------------------- which streams share a hardware queue
// The HIP runtime multiplexes streams onto a few in-order hardware queues per priority level (four by default: measured
// with tools/microbench/queue_map_probe.hip -- a new stream goes to the queue with the fewest streams of its priority,
// ties in a fixed order; high and low priority have four queues of their own each).  A queue runs what it holds in
// submission order, one packet after the other, and an event wait blocks the whole queue: a stream that shares a queue
// with a busy one stands behind that stream's backlog.  Where the sliced mode's streams land therefore decides how much
// of the schedule's overlap is real (tests/hwq_model.py is the CPU model of exactly this).  The probe: hold
// one stream with a one-wave kernel that spins for 200 us of the GPU's wall clock and notes when it ended, stamp the
// time on every other stream; a stamp not earlier than the end was taken behind the spin -- same queue.  GPU clock
// only; the streams are idle when it runs (creation time).
struct QueueProbe {
    imt_ctx* ctx = nullptr;
    uint64_t* d = nullptr;
    static constexpr int MAXS = 16;
    uint64_t spin_ticks = 20000;
    int probes = 0;

    int init(imt_ctx* c) {
        ctx = c;
        int khz = 100000;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
        spin_ticks = (uint64_t)khz / 5;      // 200 us
        IMT_HIP(c, hipMalloc((void**)&d, sizeof(uint64_t) * (MAXS + 1)));
        // both kernels once, so that no first-launch cost (code object load) sits inside a measurement
        imt::launch::spin(c->stream, 1, d + MAXS);
        imt::launch::stamp(c->stream, d);
        IMT_HIP(c, hipStreamSynchronize(c->stream));
        return IMT_OK;
    }
    ~QueueProbe() {
        if (d) hipFree(d);
    }
    // behind[j] = 1: others[j] shares a hardware queue with `busy`
    int run(hipStream_t busy, const hipStream_t* others, int n, char* behind) {
        if (n > MAXS) return ctx->fail(IMT_ERR_INTERNAL, "queue probe: too many streams");
        uint64_t h[MAXS + 1];
        IMT_HIP(ctx, hipStreamSynchronize(busy));
        for (int j = 0; j < n; j++) IMT_HIP(ctx, hipStreamSynchronize(others[j]));
        IMT_HIP(ctx, hipMemsetAsync(d, 0, sizeof(uint64_t) * (MAXS + 1), ctx->stream));
        IMT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        imt::launch::spin(busy, spin_ticks, d + MAXS);
        for (int j = 0; j < n; j++) imt::launch::stamp(others[j], d + j);
        IMT_HIP(ctx, hipGetLastError());
        IMT_HIP(ctx, hipStreamSynchronize(busy));
        for (int j = 0; j < n; j++) IMT_HIP(ctx, hipStreamSynchronize(others[j]));
        IMT_HIP(ctx, hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        for (int j = 0; j < n; j++) behind[j] = h[j] >= h[MAXS];
        probes++;
        return IMT_OK;
    }
};

// ---------------------------------------------------------------------------------------------- one replica over HIP
struct HipBackend : Backend {
    imt_itree* tree;
    imt_ctx* ctx;
    hipStream_t rs[ROUNDS] = {}, cs[ROUNDS] = {}, aps[ROUNDS] = {};     // round, collective and apply streams
    int n_comm = ROUNDS;
    int round_prio = 0;                       // HIP priority of the (equal-priority) round streams
    SlicedOptions opt;
    // what the probe found (imt_sliced_info): hardware queue class of every stream, -1 = no such stream / not probed
    int q_round[ROUNDS], q_comm[ROUNDS], q_apply[ROUNDS];
    int n_queues = 0, placement = IMT_SLICED_PLACEMENT_UNVERIFIED, streams_recreated = 0;
    bool comm_own_queues = false;
    bool hung = false;                        // sync() ran into the watchdog: something of this world still runs on the device
    std::string placement_note;

    explicit HipBackend(imt_itree* t) : tree(t), ctx(imt_itree_ctx(t)) {
        for (int i = 0; i < ROUNDS; i++) q_round[i] = q_comm[i] = q_apply[i] = -1;
    }
    int new_stream(hipStream_t* out, int prio) {
        IMT_HIP(ctx, hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio));
        return IMT_OK;
    }
    void adopt(hipStream_t s) { ctx->side_streams.push_back(s); }
    // channels: how many independent channels the transport has (0 = one per round slot)
    int init(const SlicedOptions& o, int channels) {
        opt = o;
        int rc = ctx->set_device();
        if (rc) return rc;
        // Measured on one MI355X with in-process replicas (profiles/r04_sliced_stream_matrix.txt): the four round streams
        // at EQUAL priority (2.93-2.95 M insertions/s at world 1, 2.86 at world 2) beat the batch pipeline's scheme of one
        // normal + three high (2.49-2.64 / 2.64-2.83) -- rounds are whole slices apart here, not one level, and a
        // high-priority round starves the others.  The collectives go to four normal-priority streams of their own; WHERE
        // those sit on the runtime's hardware queues is place()'s business (queues of their own if the runtime has any to
        // give, else their rounds' queues; never another round's queue, where a gather would stand in front of that
        // round's hash kernels: -5 ... -18 % in round 4's emulation).
        n_comm = (int)o.comm_streams;
        // Round slots that share a channel of the transport (an RCCL communicator) must enqueue on ONE stream, or the
        // communicator would see its collectives in an order the GPU decides
        if (channels > 0 && channels < ROUNDS) n_comm = channels;
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
        // The four round streams at ONE priority (equal): normal, or -- a pool of hardware queues per priority level -- all
        // LOW / all HIGH: a pool nobody else in the process creates streams in (torch, RCCL: normal priority), so that no
        // foreign stream can share a round's queue.  That matters with RCCL: a communicator owns internal streams
        // (ncclCommInitRank creates three, normal priority: tools/microbench/rccl_streams_probe.hip) and brackets every
        // collective with one of them -- the user's stream waits for it, it waits for the kernel -- so on a round's queue it
        // would put the round behind every collective of that communicator, and the collective behind the round's backlog
        const bool equal = o.round_priorities != 1;
        round_prio = o.round_priorities == 2 ? least : o.round_priorities == 3 ? greatest : 0;
        const int comm_prio = std::max(greatest, std::min(least, (int)o.comm_priority));
        for (int i = 0; i < ROUNDS; i++)
            if ((rc = new_stream(&rs[i], equal ? round_prio : std::max(greatest, std::min(least, 0 - i))))) return rc;
        for (int i = 0; i < n_comm; i++)
            if ((rc = new_stream(&cs[i], comm_prio))) return rc;
        // The schedule (imt_sliced_sched.hpp) lets the other ranks' write-backs be applied on a stream of their own per
        // round slot; here they stay on the round's stream unless asked for (four more streams on the four hardware
        // queues cost 2-4 %: tools/rank_emulation.py, profiles/r04_rank_emulation.txt).
        if (o.apply_streams)
            for (int i = 0; i < ROUNDS; i++)
                if ((rc = new_stream(&aps[i], equal ? round_prio : 0))) return rc;
        if (o.verify_queues && equal) {
            if ((rc = place(comm_prio))) return rc;
        } else {
            placement_note = o.verify_queues ? "not verified: unequal round priorities put the round streams into different pools"
                                             : "not verified (IMT_SLICED_OPT_VERIFY_QUEUES = 0)";
        }
        for (hipStream_t* arr : {rs, cs, aps})
            for (int i = 0; i < ROUNDS; i++)
                if (arr[i]) adopt(arr[i]);
        imt_itree_set_slice_wait_limit(tree, (double)o.watchdog_ms);
        return IMT_OK;
    }

    // Verify where the streams landed and repair what can be repaired by creating streams again:
    //   (1) the four round streams on four DIFFERENT hardware queues (two rounds on one queue take turns);
    //   (2) a helper stream of slot i (collectives, applies) on the queue of round stream i when it lives in the rounds'
    //       pool, on none of the rounds' queues when its priority puts it into another pool.
    // The runtime gives a new stream the queue with the fewest streams of its priority, so whether (1) and (2) hold depends
    // on how many streams the host (torch, RCCL, the application) created before: nothing this library controls.  Spare
    // streams are created until a wanted place comes up (a stream that lands elsewhere is kept until the end so that the
    // next one lands on another queue) and destroyed afterwards.  What cannot be had is reported (imt_sliced_info.placement,
    // imt_sliced_last_error) and degrades to the collectives on the rounds' own streams.
    int place(int comm_prio) {
        QueueProbe pr;
        int rc = pr.init(ctx);
        if (rc) return rc;
        struct Spares : std::vector<hipStream_t> {      // streams that landed on the wrong queue: alive until the end (they keep
            ~Spares() {                                  // the next creation off their queue), destroyed on every way out
                for (hipStream_t s : *this) hipStreamDestroy(s);
            }
        } spare;
        auto drop_spares = [&] {
            for (hipStream_t s : spare) hipStreamDestroy(s);
            spare.clear();
        };
        char behind[QueueProbe::MAXS];
        // ---- (1) classes of the round streams
        int cls[ROUNDS];
        auto classify_rounds = [&]() -> int {
            for (int i = 0; i < ROUNDS; i++) cls[i] = -1;
            int next = 0;
            for (int i = 0; i < ROUNDS; i++) {
                if (cls[i] >= 0) continue;
                cls[i] = next++;
                if (i + 1 == ROUNDS) break;
                int r = pr.run(rs[i], rs + i + 1, ROUNDS - i - 1, behind);
                if (r) return r;
                for (int j = i + 1; j < ROUNDS; j++)
                    if (behind[j - i - 1] && cls[j] < 0) cls[j] = cls[i];
            }
            n_queues = next;
            return IMT_OK;
        };
        if ((rc = classify_rounds())) return rc;
        for (int attempt = 0; n_queues < ROUNDS && attempt < 2 * ROUNDS; attempt++) {
            // a round stream that shares its queue with an earlier one: create another stream and see where it lands
            int dup = -1;
            for (int i = 1; i < ROUNDS && dup < 0; i++)
                for (int j = 0; j < i; j++)
                    if (cls[i] == cls[j]) dup = i;
            hipStream_t x;
            if ((rc = new_stream(&x, round_prio))) return rc;
            if ((rc = pr.run(x, rs, ROUNDS, behind))) {
                hipStreamDestroy(x);
                return rc;
            }
            bool shares = false;
            for (int j = 0; j < ROUNDS; j++) shares = shares || (behind[j] && j != dup);
            if (!shares) {          // a queue no other round stream is on
                spare.push_back(rs[dup]);
                rs[dup] = x;
                streams_recreated++;
                if ((rc = classify_rounds())) return rc;
            } else {
                spare.push_back(x);
            }
        }
        for (int i = 0; i < ROUNDS; i++) q_round[i] = cls[i];
        // ---- (2) the helper streams
        const bool same_pool = comm_prio == round_prio;
        bool ok = n_queues == ROUNDS;
        auto partner = [&](hipStream_t s, int* out) -> int {       // which round stream's queue s is on (-1: none of them)
            int r = pr.run(s, rs, ROUNDS, behind);
            if (r) return r;
            *out = -1;
            for (int j = 0; j < ROUNDS; j++)
                if (behind[j]) { *out = j; break; }
            return IMT_OK;
        };
        auto settle = [&](hipStream_t* arr, int count, int prio, bool want_partner, int* qmap, int max_new) -> int {
            // want_partner: arr[i] on round stream i's queue; else: each on a queue of its own that no round stream is on
            std::vector<hipStream_t> pool(arr, arr + count);
            std::vector<hipStream_t> placed(count, nullptr);
            int created = 0;
            // An error return leaves arr[] as the caller gave it (~HipBackend destroys those): none of them may stay among
            // the spares (~Spares would destroy them a second time), and every stream created here goes there.
            auto bail = [&](int r, hipStream_t cur) -> int {
                auto callers = [&](hipStream_t x) { return std::find(arr, arr + count, x) != arr + count; };
                spare.erase(std::remove_if(spare.begin(), spare.end(), callers), spare.end());
                std::vector<hipStream_t> mine(placed);
                mine.insert(mine.end(), pool.begin(), pool.end());
                mine.push_back(cur);
                for (hipStream_t x : mine)
                    if (x && !callers(x) && std::find(spare.begin(), spare.end(), x) == spare.end()) spare.push_back(x);
                return r;
            };
            while (true) {
                bool full = true;
                for (int i = 0; i < count; i++) full = full && placed[i];
                if (full) break;
                hipStream_t s;
                if (!pool.empty()) {
                    s = pool.back();
                    pool.pop_back();
                } else {
                    if (created >= max_new) break;
                    int r = new_stream(&s, prio);
                    if (r) return bail(r, nullptr);
                    created++;
                }
                int p = -1;
                int r = partner(s, &p);
                if (r) return bail(r, s);
                int slot = -1;
                if (want_partner) {
                    // with fewer than four queues under the rounds, the partner is the first round stream of the class
                    for (int i = 0; i < count && slot < 0; i++)
                        if (!placed[i] && p >= 0 && cls[i] == cls[p]) slot = i;
                } else if (p < 0) {
                    // a queue none of the round streams is on -- and none of the helpers placed so far either: two slots'
                    // collectives on one queue would wait for each other's peers
                    std::vector<hipStream_t> others;
                    for (int i = 0; i < count; i++)
                        if (placed[i]) others.push_back(placed[i]);
                    bool shared = false;
                    if (!others.empty()) {
                        if ((r = pr.run(s, others.data(), (int)others.size(), behind))) return bail(r, s);
                        for (size_t j = 0; j < others.size(); j++) shared = shared || behind[j];
                    }
                    for (int i = 0; i < count && slot < 0 && !shared; i++)
                        if (!placed[i]) slot = i;
                }
                if (slot >= 0) placed[slot] = s;
                else spare.push_back(s);
            }
            bool all = true;
            for (int i = 0; i < count; i++) all = all && placed[i];
            if (!all) {         // could not be had: give the caller `count` live streams back (whichever), the rest are spares
                std::vector<hipStream_t> have;
                for (int i = 0; i < count; i++)
                    if (placed[i]) have.push_back(placed[i]);
                while ((int)have.size() < count && !pool.empty()) { have.push_back(pool.back()); pool.pop_back(); }
                while ((int)have.size() < count && !spare.empty()) { have.push_back(spare.back()); spare.pop_back(); }
                for (int i = 0; i < count; i++) arr[i] = i < (int)have.size() ? have[i] : nullptr;
                return 1;
            }
            for (int i = 0; i < count; i++) {
                if (placed[i] != arr[i]) streams_recreated++;
                arr[i] = placed[i];
                qmap[i] = want_partner ? cls[i] : (prio == round_prio ? ROUNDS + i : -2);
            }
            return IMT_OK;
        };
        if (n_comm == ROUNDS) {
            // Where the collectives' streams should sit.  A collective holds its hardware queue until every rank's has
            // started; on its round's queue it therefore holds up the round's NEXT unit until the slowest rank has packed
            // this tick -- every tick of every round becomes a barrier across ranks, although the schedule consumes a
            // gather only `lag` ticks later.  On a queue of its own the gather overlaps the next units and a rank may run
            // up to `lag` ticks ahead of its peers.  The queue model (tests/hwq_model.py, tools/hwq_calibrate.py) prices the
            // difference at 2 / 4 / 8 GPUs: 5.2 -> 5.7, 9.9 -> 10.9, 20.5 -> 21.1 M insertions/s.  Queues of their own exist
            // when the pool the rounds use has more than four queues (GPU_MAX_HW_QUEUES=8 in the host's environment)
            // or when the collectives' streams have another priority (IMT_SLICED_OPT_POOLS, or IMT_SLICED_OPT_COMM_PRIORITY
            // = 1: the low-priority pool; measured 2 - 6 % slower per rank than normal priority).  IMT_SLICED_OPT_COMM_PLACEMENT:
            // 0 = queues of their own if they can be had, else their rounds' queues; 1 = their rounds' queues; 2 = own or fail.
            rc = 1;
            bool own = false;
            if (!same_pool || opt.comm_placement != 1) {
                rc = settle(cs, ROUNDS, comm_prio, false, q_comm, same_pool ? 2 * ROUNDS : 3 * ROUNDS);
                if (rc < 0) return rc;
                own = rc == IMT_OK;
            }
            if (rc == 1 && same_pool && opt.comm_placement != 2) {
                rc = settle(cs, ROUNDS, comm_prio, true, q_comm, 3 * ROUNDS);
                if (rc < 0) return rc;
            }
            comm_own_queues = own;
            if (rc == 1) {
                // the collectives go to the rounds' own streams: the same queue order, no second stream to misplace
                for (int i = 0; i < ROUNDS; i++)
                    if (cs[i]) { spare.push_back(cs[i]); cs[i] = nullptr; }
                n_comm = 0;
                ok = false;
                placement_note = "the collectives' streams could be placed neither on queues of their own nor on their rounds' hardware queues: collectives are enqueued on the round streams";
            }
        } else if (n_comm > 0) {
            // fewer streams than round slots (a transport with fewer channels): every stream serves several slots, there is
            // no place that suits all of them; report where they are
            for (int i = 0; i < n_comm; i++) {
                int p = -1;
                if ((rc = partner(cs[i], &p))) return rc;
                q_comm[i] = p >= 0 ? cls[p] : -2;
            }
        }
        if (aps[0]) {
            rc = settle(aps, ROUNDS, round_prio, true, q_apply, 3 * ROUNDS);
            if (rc < 0) return rc;
            if (rc == 1) {
                for (int i = 0; i < ROUNDS; i++)
                    if (aps[i]) { spare.push_back(aps[i]); aps[i] = nullptr; }
                ok = false;
                placement_note += (placement_note.empty() ? "" : "; ");
                placement_note += "the apply streams could not be placed: applies run on the round streams";
            }
        }
        drop_spares();
        if (n_queues < ROUNDS) {
            char buf[160];
            snprintf(buf, sizeof buf, "%sthe %d round streams share %d hardware queues (GPU_MAX_HW_QUEUES?)", placement_note.empty() ? "" : "; ", ROUNDS, n_queues);
            placement_note += buf;
        }
        placement = ok ? (streams_recreated ? IMT_SLICED_PLACEMENT_REPAIRED : IMT_SLICED_PLACEMENT_AS_CREATED) : IMT_SLICED_PLACEMENT_DEGRADED;
        return IMT_OK;
    }
    ~HipBackend() override {
        if (ctx->set_device()) return;
        auto& ss = ctx->side_streams;
        for (hipStream_t* arr : {rs, cs, aps})
            for (int i = 0; i < ROUNDS; i++)
                if (arr[i]) {
                    // a hung world's streams are LEFT (not waited for, not destroyed): one of them holds a collective or a
                    // wait that may never end; the context forgets them so that imt_ctx_sync does not wait for them either
                    if (!hung) hipStreamSynchronize(arr[i]);
                    ss.erase(std::remove(ss.begin(), ss.end(), arr[i]), ss.end());
                    if (!hung) hipStreamDestroy(arr[i]);
                }
        imt_itree_set_slice_poison(tree, nullptr);
        imt_itree_set_slice_wait_limit(tree, 0);
    }
    Stream round_stream(int slot) override { return rs[slot]; }
    Stream comm_stream(int slot) override { return n_comm ? cs[slot % n_comm] : rs[slot]; }
    Stream apply_stream(int slot) override { return aps[slot] ? aps[slot] : rs[slot]; }
    int new_event(Event* out) override {
        hipEvent_t e;
        IMT_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        *out = e;
        return IMT_OK;
    }
    void free_event(Event e) override { hipEventDestroy((hipEvent_t)e); }
    int record(Event e, Stream s) override {
        IMT_HIP(ctx, hipEventRecord((hipEvent_t)e, (hipStream_t)s));
        return IMT_OK;
    }
    int wait(Stream s, Event e) override {
        IMT_HIP(ctx, hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0));
        return IMT_OK;
    }
    int event_sync(Event e) override {
        if (opt.watchdog_ms <= 0) {
            IMT_HIP(ctx, hipEventSynchronize((hipEvent_t)e));
            return IMT_OK;
        }
        return bounded_wait(ctx, opt.watchdog_ms, [&] { return hipEventQuery((hipEvent_t)e); }, "a round's last unit");
    }
    int event_query(Event e) override {
        const hipError_t r = hipEventQuery((hipEvent_t)e);
        return r == hipSuccess ? 1 : r == hipErrorNotReady ? 0 : -1;
    }
    int alloc(size_t bytes, Buffer* out) override {
        void* p = nullptr;
        IMT_HIP(ctx, hipMalloc(&p, bytes));
        IMT_HIP(ctx, hipMemset(p, 0, bytes));
        *out = p;
        return IMT_OK;
    }
    // hipFree waits for the whole device to go idle: never after a timeout (the buffers of a hung world are leaked; the
    // process is expected to exit, include/imt.h IMT_ERR_TIMEOUT)
    void free_buffer(Buffer b) override {
        if (!hung) hipFree(b);
    }
    int copy(Buffer dst, size_t doff, Buffer src, size_t soff, size_t bytes, Stream s) override {
        IMT_HIP(ctx, hipMemcpyAsync((uint8_t*)dst + doff, (const uint8_t*)src + soff, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
        return IMT_OK;
    }
    uint64_t tree_size() override { return imt_itree_size(tree); }
    size_t payload_bytes(size_t n) override { return imt_itree_slice_payload_bytes(n); }
    size_t unit_bytes(uint64_t sb, size_t n, unsigned q) override { return imt_itree_slice_unit_bytes(tree, sb, n, q); }
    int prepare(const void* vals, size_t nb, size_t no, size_t na, const imt_insert_out* out, unsigned flags, int slot,
                int* slice) override {
        // Where the step's preparation (some sixty small dependent kernels: sort, merges, low-leaf search, level tables)
        // is enqueued.  The tree's side stream shares a hardware queue with some round's stream, and a queue runs what it
        // holds in submission order: there the preparation stood behind that round's hash kernels.  The NEW round's slot has
        // been idle since its previous round ended, so its streams' queue is free; on the slot's collective stream the
        // preparation is also not in front of the round's own first units in stream order.  One rank of N = 2 / 4 / 8
        // alone on the GPU (tools/rank_emulation.py, first rank, modelled links): side stream 2.93 / 2.89 / 2.81, the round's
        // stream 3.02 / 2.99 / 2.77, the slot's collective stream 3.03 / 2.97 / 2.80 M insertions/s.
        // IMT_SLICED_OPT_PREP_STREAM: 0 the slot's collective stream (default), 1 its round stream, 2 the side stream.
        void* st = n_comm ? (void*)cs[slot % n_comm] : (void*)rs[slot];
        if (opt.prep_stream == 2) st = nullptr;
        if (opt.prep_stream == 1) st = rs[slot];
        imt_itree_set_slice_prep_stream(tree, st);
        const int rc = imt_itree_slice_prepare(tree, vals, nb, no, na, out, flags | IMT_DEVICE_PTRS, slice, nullptr);
        imt_itree_set_slice_prep_stream(tree, nullptr);
        return rc;
    }
    int unit(int slice, unsigned q, Buffer payload, Stream s) override { return imt_itree_slice_unit(tree, slice, q, payload, s); }
    // the unit's last kernel (the pack) / the apply kernel signals the event itself where it can: no marker packet behind it
    int unit_record(int slice, unsigned q, Buffer payload, Stream s, Event ev) override {
        imt_itree_set_slice_tail_event(tree, ev);
        const int rc = imt_itree_slice_unit(tree, slice, q, payload, s);
        const bool attached = imt_itree_take_slice_tail_attached(tree);
        return rc ? rc : attached ? IMT_OK : record(ev, s);
    }
    int apply_record(Buffer g, size_t stride, int count, const uint64_t* sb, const uint64_t* n, const int32_t* units, Stream s,
                     Event ev) override {
        imt_itree_set_slice_tail_event(tree, ev);
        const int rc = imt_itree_slice_apply_gathered(tree, g, stride, (size_t)count, sb, n, units, s);
        const bool attached = imt_itree_take_slice_tail_attached(tree);
        return rc ? rc : attached ? IMT_OK : record(ev, s);
    }
    int apply_gathered(Buffer g, size_t stride, int count, const uint64_t* sb, const uint64_t* n, const int32_t* units,
                       Stream s) override {
        return imt_itree_slice_apply_gathered(tree, g, stride, (size_t)count, sb, n, units, s);
    }
    int sync() override {
        int rc = ctx->set_device();
        if (rc) return rc;
        for (hipStream_t* arr : {rs, cs, aps})
            for (int i = 0; i < ROUNDS; i++)
                if (arr[i]) {
                    if (opt.watchdog_ms <= 0) IMT_HIP(ctx, hipStreamSynchronize(arr[i]));
                    else if ((rc = bounded_wait(ctx, opt.watchdog_ms, [&] { return hipStreamQuery(arr[i]); }, "the world's streams to drain"))) {
                        hung = hung || rc == IMT_ERR_TIMEOUT;
                        return rc;
                    }
                }
        if (opt.watchdog_ms > 0) {          // the context's own streams under the same limit, then the unbounded call finds them idle
            std::vector<hipStream_t> own(ctx->side_streams);
            own.push_back(ctx->stream);
            for (hipStream_t st : own)
                if ((rc = bounded_wait(ctx, opt.watchdog_ms, [&] { return hipStreamQuery(st); }, "the context's streams to drain"))) {
                    hung = hung || rc == IMT_ERR_TIMEOUT;
                    return rc;
                }
        }
        return imt_ctx_sync(ctx);
    }
};

// ------------------------------------------------------------------------------------- a caller-supplied collective
struct CustomTransport : Transport {
    imt_transport_ops ops;
    explicit CustomTransport(const imt_transport_ops& o) : ops(o) {}
    ~CustomTransport() override {
        if (ops.destroy) ops.destroy(ops.self);
    }
    int all_gather(Rank& rk, int slot, int r, size_t bytes, Stream st) override {
        const int i = rk.at(slot, r);
        return ops.all_gather(ops.self, slot, r, rk.send[i], rk.recv[i], bytes, st);
    }
    int small_gather(const void* send, void* recv, size_t bytes, Stream st) override {
        return ops.all_gather(ops.self, 0, 0, send, recv, bytes, st);
    }
};

// ------------------------------------------------------------------------------ direct peer copies between processes
// Every rank exports ONE device allocation holding all its send buffers (hipIpcMemHandle) and one page of POSIX shared
// memory with two 64-bit counters per (slot, ring): `packed` (my payload number k is in my send buffer) and `copied` (I
// have copied every peer's payload number k).  Both pages are registered with HIP in every process, so the counters are
// written and polled BY THE GPUS, in stream order (imt_flags.hip): a gather = publish my `packed`, wait until every
// peer's `packed` has reached k, copy each payload out of its owner's memory (a device-to-device copy: an xGMI
// point-to-point read when the ranks sit on different GPUs), publish my `copied`; the fence = wait until every peer's
// `copied` has reached k, then the send buffer may be rewritten.  Counters only grow, so nothing depends on the order in
// which the hosts issue their work, and no host ever waits for another.  (A first version used HIP's interprocess
// events; they failed after a few dozen records per event -- profiles/r04 notes -- and need a host-side handshake for
// HIP's "latest record issued before the wait" rule.)  A peer that never arrives: the waiting kernel gives up after
// IMT_IPC_TIMEOUT_S (default 60) and sets an error bit, reported by the next imt_sliced_wait / _flush.
constexpr int IPC_RING_MAX = 12;
constexpr int IPC_NEV = ROUNDS * IPC_RING_MAX;
// imt_transport_all_gather over IPC: a ring of small staging slots behind the send buffers, with counters of their own
constexpr int IPC_AUX_RING = 4;
constexpr size_t IPC_AUX_BYTES = 4096;

struct IpcBlob {
    hipIpcMemHandle_t mem;
    char shm_name[64];
    char bus_id[32];                 // PCI bus id of the rank's GPU: ranks that SHARE a device wait on the host (below)
    uint64_t arena_bytes;
    int32_t rank, world, ring, pid;
};
struct IpcShm {
    uint64_t packed[IPC_NEV], copied[IPC_NEV];
    uint64_t aux_packed[IPC_AUX_RING], aux_copied[IPC_AUX_RING];
    uint32_t err;
};
constexpr size_t IPC_SHM_BYTES = (sizeof(IpcShm) + 4095) & ~(size_t)4095;

struct HostTimer {          // IMT_SLICED_TIMING=1: where the host's time inside the transport goes (printed at destroy)
    double ms[6] = {0};
    uint64_t n[6] = {0};
    bool on = effective_options().timing != 0;
    struct Scope {
        HostTimer& t;
        int k;
        std::chrono::steady_clock::time_point t0;
        Scope(HostTimer& t_, int k_) : t(t_), k(k_), t0(std::chrono::steady_clock::now()) {}
        ~Scope() {
            if (!t.on) return;
            t.ms[k] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            t.n[k]++;
        }
    };
};

struct IpcTransport : Transport {
    imt_ctx* ctx;
    HostTimer ht;
    int world, rank, ring = 0;
    size_t payload_cap = 0;
    uint8_t* arena = nullptr;                    // [ROUNDS][ring][payload_cap], then [IPC_AUX_RING][IPC_AUX_BYTES]
    const char* arena_kind = "";                 // "uncached" / "fine-grained" / "default": the memory the arena got (create)
    size_t aux_off = 0;
    uint64_t aux_seq = 0;
    IpcShm *my_shm = nullptr, *my_shm_dev = nullptr;
    std::string shm_name;
    uint64_t seq[IPC_NEV] = {};                  // gathers issued per (slot, ring): the same on every rank
    struct Peer {
        uint8_t* arena = nullptr;
        IpcShm *shm = nullptr, *shm_dev = nullptr;
    };
    std::vector<Peer> peers;
    bool connected = false;
    uint64_t timeout_ticks = 0;
    double timeout_s = 60.0;
    // ---- host-polled form (ranks that share ONE GPU: the one-GPU rehearsal) ----
    // A kernel that waits for a peer holds one of its process's few hardware queues until the peer has got there; with
    // several processes time-sharing one device that is most of the time, and whatever sits behind it -- other rounds'
    // hash kernels, the side stream of the next step's value check -- waits too (profiles/r04_ipc_rehearsal_notes.txt).
    // So when a peer shares this rank's device the waiting moves to the host: a worker thread watches the peers'
    // `packed` counters in the shared pages and enqueues each payload's copy when it is there; the fence waits (on the
    // host) until the worker has enqueued everything of that gather and the peers' `copied` counters have arrived.
    // Nothing on the GPU ever waits for another process.  Ranks on different GPUs keep the GPU-polled form: no host in
    // the data path.  IMT_IPC_HOST_POLL=0 / 1 overrides.
    bool host_poll = false;
    struct Job {
        int slot, i;
        size_t bytes, off;
        uint64_t k;
        uint8_t* recv;
        uint32_t copied_mask;
        bool ordered;            // the worker's stream has been put behind ready_ev
    };
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> jobs;
    bool stop = false;
    std::atomic<uint64_t> issued[IPC_NEV];
    std::atomic<int> worker_error{0};
    hipStream_t ws[ROUNDS] = {};
    hipEvent_t done_ev[IPC_NEV] = {}, ready_ev[IPC_NEV] = {};
    char my_bus[32] = {0};

    IpcTransport(imt_ctx* c, int w, int r) : ctx(c), world(w), rank(r) {
        for (auto& x : issued) x.store(0);
    }
    int clock_khz = 100000;
    // at least 10 ms: a limit of zero (or a negative one) would make every wait give up at once
    void set_timeout(double seconds, int khz = 0) {
        if (khz > 0) clock_khz = khz;
        timeout_s = !(seconds >= 0.01) ? 0.01 : seconds > 86400.0 ? 86400.0 : seconds;
        timeout_ticks = (uint64_t)(timeout_s * 1e3 * clock_khz);
    }
    int host_poll_option = -1;                   // imt_transport_set_option: -1 decide from the bus ids, 0 / 1 forced
    const uint32_t* poison_word() override { return my_shm_dev ? &my_shm_dev->err : nullptr; }
    static int ei(int slot, int r) { return slot * IPC_RING_MAX + r; }

    int map_page(const char* name, bool create, IpcShm** host, IpcShm** dev) {
        const int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
        if (fd < 0 || (create && ftruncate(fd, (off_t)IPC_SHM_BYTES) != 0)) {
            if (fd >= 0) close(fd);
            return ctx->fail(IMT_ERR_ALLOC, "shm_open(%s) failed", name);
        }
        void* p = mmap(nullptr, IPC_SHM_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return ctx->fail(IMT_ERR_ALLOC, "mmap of the flag page %s failed", name);
        if (create) std::memset(p, 0, IPC_SHM_BYTES);
        hipError_t e = hipHostRegister(p, IPC_SHM_BYTES, hipHostRegisterMapped | hipHostRegisterPortable);
        if (e != hipSuccess) {
            munmap(p, IPC_SHM_BYTES);
            return ctx->hip_fail(e, "hipHostRegister(flag page)");
        }
        if ((e = hipHostGetDevicePointer((void**)dev, p, 0)) != hipSuccess) {
            hipHostUnregister(p);
            munmap(p, IPC_SHM_BYTES);
            return ctx->hip_fail(e, "hipHostGetDevicePointer(flag page)");
        }
        *host = (IpcShm*)p;
        return IMT_OK;
    }
    void unmap_page(IpcShm* host) {
        if (!host) return;
        hipHostUnregister(host);
        munmap(host, IPC_SHM_BYTES);
    }
    int create(unsigned depth, size_t max_slice, int lag, IpcBlob* blob) {
        Schedule sc;
        if (!sc.init(world, (int)depth + 1, lag)) return ctx->fail(IMT_ERR_RANGE, "not a schedule: world %d depth %u lag %d", world, depth, lag);
        ring = sc.lag + 1;
        if (ring > IPC_RING_MAX) return ctx->fail(IMT_ERR_RANGE, "lag %d too large for the IPC transport (max %d)", sc.lag, IPC_RING_MAX - 1);
        if (world - 1 > imt::launch::FLAG_WAIT_MAX) return ctx->fail(IMT_ERR_RANGE, "the IPC transport joins at most %d ranks", imt::launch::FLAG_WAIT_MAX + 1);
        int rc = ctx->set_device();
        if (rc) return rc;
        int khz = 100000;                        // wall_clock64 ticks per millisecond (100 MHz on gfx9)
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || khz <= 0) khz = 100000;
        if (const char* e = getenv("IMT_IPC_TIMEOUT_S")) timeout_s = atof(e);       // tools/ only; imt_transport_set_option otherwise
        set_timeout(timeout_s, khz);
        if (hipDeviceGetPCIBusId(my_bus, (int)sizeof my_bus, ctx->device) != hipSuccess) my_bus[0] = 0;
        payload_cap = imt_itree_slice_payload_bytes(max_slice);
        aux_off = (size_t)ROUNDS * ring * payload_cap;
        const size_t bytes = aux_off + IPC_AUX_RING * IPC_AUX_BYTES;
        // The arena is what OTHER GPUs read, while this GPU keeps rewriting it, ordered by nothing but the counters in the
        // flag pages: it lives in UNCACHED device memory (as RCCL's own buffers do), so that neither this GPU's per-XCD L2s
        // (dirty lines of a pack kernel) nor a reader's L2 (lines of the slot's previous contents) can stand between a
        // payload and its reader -- no reliance on what a kernel boundary writes back or invalidates for a peer.  The cost
        // is the pack kernel's 36 B per written node going past the L2: nothing beside the hashing.  Fine-grained, then
        // ordinary memory if the runtime cannot give it (or cannot export it); IMT_IPC_ARENA=default for tools' A/B runs.
        std::memset(blob, 0, sizeof *blob);
        const char* want = getenv("IMT_IPC_ARENA");
        const unsigned kinds[3] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained, hipDeviceMallocDefault};
        const char* names[3] = {"uncached", "fine-grained", "default"};
        hipError_t e = hipErrorUnknown;
        for (int k = (want && !strcmp(want, "default")) ? 2 : (want && !strcmp(want, "finegrained")) ? 1 : 0; k < 3; k++) {
            void* p = nullptr;
            e = k < 2 ? hipExtMallocWithFlags(&p, bytes, kinds[k]) : hipMalloc(&p, bytes);
            if (e == hipSuccess && (e = hipIpcGetMemHandle(&blob->mem, p)) != hipSuccess) hipFree(p);
            if (e == hipSuccess) {
                arena = (uint8_t*)p;
                arena_kind = names[k];
                break;
            }
            (void)hipGetLastError();
        }
        if (e != hipSuccess) return ctx->hip_fail(e, "IPC transport: allocating / exporting the send arena");
        IMT_HIP(ctx, hipMemset(arena, 0, bytes));
        char name[64];
        snprintf(name, sizeof name, "/imt_ipc_%d_%d_%llx", (int)getpid(), rank,
                 (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
        shm_name = name;                         // from here on the destructor unlinks it
        if ((rc = map_page(name, true, &my_shm, &my_shm_dev))) return rc;
        snprintf(blob->shm_name, sizeof blob->shm_name, "%s", name);
        snprintf(blob->bus_id, sizeof blob->bus_id, "%s", my_bus);
        blob->arena_bytes = bytes;
        blob->rank = rank;
        blob->world = world;
        blob->ring = ring;
        blob->pid = (int)getpid();
        return IMT_OK;
    }
    int connect(const IpcBlob* all) {
        int rc = ctx->set_device();
        if (rc) return rc;
        peers.assign(world, Peer());
        for (int h = 0; h < world; h++) {
            if (h == rank) continue;
            const IpcBlob& b = all[h];
            if (b.rank != h || b.world != world || b.ring != ring || b.arena_bytes != aux_off + IPC_AUX_RING * IPC_AUX_BYTES)
                return ctx->fail(IMT_ERR_ARG, "IPC blob %d does not describe rank %d of this world (same depth, max_slice and lag on every rank)", h, h);
            if (b.pid == (int)getpid()) return ctx->fail(IMT_ERR_ARG, "the IPC transport joins PROCESSES; ranks of one process use the local transport");
            Peer& p = peers[h];
            IMT_HIP(ctx, hipIpcOpenMemHandle((void**)&p.arena, b.mem, hipIpcMemLazyEnablePeerAccess));
            if ((rc = map_page(b.shm_name, false, &p.shm, &p.shm_dev))) return rc;
            if (my_bus[0] && !strncmp(my_bus, b.bus_id, sizeof my_bus)) host_poll = true;      // a peer on MY device
        }
        if (host_poll_option >= 0) host_poll = host_poll_option != 0;
        if (const char* e = getenv("IMT_IPC_HOST_POLL")) host_poll = atoi(e) != 0;                // tools/ only
        if (host_poll) {
            for (int slot = 0; slot < ROUNDS; slot++) IMT_HIP(ctx, hipStreamCreateWithFlags(&ws[slot], hipStreamNonBlocking));
            for (int slot = 0; slot < ROUNDS; slot++)
                for (int r = 0; r < ring; r++) {
                    IMT_HIP(ctx, hipEventCreateWithFlags(&done_ev[ei(slot, r)], hipEventDisableTiming));
                    IMT_HIP(ctx, hipEventCreateWithFlags(&ready_ev[ei(slot, r)], hipEventDisableTiming));
                }
            worker = std::thread([this] { this->work(); });
        }
        connected = true;
        return IMT_OK;
    }
    // the worker of the host-polled form: enqueue each peer's payload copy as soon as that peer says it is packed
    void work() {
        if (hipSetDevice(ctx->device) != hipSuccess) { worker_error.store(1); return; }
        std::vector<Job> pend;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                if (pend.empty()) cv.wait(lk, [this] { return stop || !jobs.empty(); });
                while (!jobs.empty()) { pend.push_back(jobs.front()); jobs.pop_front(); }
                if (stop) {
                    // jobs still waiting for a peer are given up: whoever waits for them (a fence on the host) sees the
                    // error instead of the time limit
                    if (!pend.empty()) worker_error.store(4);
                    return;
                }
            }
            bool progress = false;
            for (size_t q = 0; q < pend.size();) {
                Job& j = pend[q];
                const uint32_t all = ((1u << world) - 1u) & ~(1u << rank);
                for (int d = 1; d < world; d++) {
                    const int h = (rank + d) % world;
                    if (j.copied_mask & (1u << h)) continue;
                    if (__atomic_load_n(&peers[h].shm->packed[j.i], __ATOMIC_ACQUIRE) < j.k) continue;
                    if (!j.ordered) {
                        if (hipStreamWaitEvent(ws[j.slot], ready_ev[j.i], 0) != hipSuccess) worker_error.store(3);
                        j.ordered = true;
                    }
                    imt::launch::copy16(ws[j.slot], j.recv + (size_t)h * j.bytes, peers[h].arena + j.off, j.bytes, &my_shm_dev->err);
                    j.copied_mask |= 1u << h;
                    progress = true;
                }
                if (j.copied_mask == all) {
                    imt::launch::flag_set_checked(ws[j.slot], &my_shm_dev->copied[j.i], j.k, &my_shm_dev->err);
                    if (hipEventRecord(done_ev[j.i], ws[j.slot]) != hipSuccess || hipGetLastError() != hipSuccess) worker_error.store(2);
                    issued[j.i].store(j.k, std::memory_order_release);
                    pend.erase(pend.begin() + (long)q);
                    progress = true;
                } else {
                    q++;
                }
            }
            if (!progress) std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    // host: wait until `cond` holds or the time limit passes
    double waited_ms = 0;
    double take_wait_ms() override {
        const double w = waited_ms;
        waited_ms = 0;
        return w;
    }
    template <class F>
    bool host_wait(F cond) {
        if (cond()) return true;
        const auto t0 = std::chrono::steady_clock::now();
        bool ok = false;
        for (unsigned spins = 0;; spins++) {
            if (cond()) { ok = true; break; }
            if ((spins & 255) == 255) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) break;
                std::this_thread::yield();
            }
        }
        waited_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return ok;
    }
    void stop_worker() {
        if (worker.joinable()) {
            {
                std::lock_guard<std::mutex> lk(mu);
                stop = true;
            }
            cv.notify_all();
            worker.join();
        }
    }
    // host side only (imt_transport_destroy after a world that never drained): no call that could wait for the device
    void abandon() override {
        stop_worker();
        if (!shm_name.empty()) shm_unlink(shm_name.c_str());
        shm_name.clear();
    }
    ~IpcTransport() override {
        if (ht.on)
            fprintf(stderr, "[imt ipc rank %d] arena in %s memory; host ms (calls): flag_set %.1f (%llu)  wait packed %.1f (%llu)  memcpy %.1f (%llu)  wait copied %.1f (%llu)\n",
                    rank, arena_kind, ht.ms[0], (unsigned long long)ht.n[0], ht.ms[1], (unsigned long long)ht.n[1], ht.ms[2], (unsigned long long)ht.n[2], ht.ms[3],
                    (unsigned long long)ht.n[3]);
        stop_worker();
        if (!shm_name.empty()) shm_unlink(shm_name.c_str());      // whatever else fails: the name does not stay behind
        if (ctx->set_device()) {                                   // no device: give the host mappings back at least
            for (auto& p : peers)
                if (p.shm) munmap(p.shm, IPC_SHM_BYTES);
            if (my_shm) munmap(my_shm, IPC_SHM_BYTES);
            return;
        }
        for (auto& w_ : ws)
            if (w_) { hipStreamSynchronize(w_); hipStreamDestroy(w_); }
        for (auto& e : done_ev)
            if (e) hipEventDestroy(e);
        for (auto& e : ready_ev)
            if (e) hipEventDestroy(e);
        for (auto& p : peers) {
            if (p.arena) hipIpcCloseMemHandle(p.arena);
            unmap_page(p.shm);
        }
        if (arena) hipFree(arena);
        unmap_page(my_shm);
    }
    int attach(Rank& rk) override {
        if (!connected) return ctx->fail(IMT_ERR_ARG, "imt_transport_ipc_connect first");
        if (rk.world != world || rk.rank != rank || rk.ring != ring || rk.payload_cap != payload_cap)
            return ctx->fail(IMT_ERR_ARG, "the IPC transport was created for another world / depth / max_slice / lag");
        return IMT_OK;
    }
    Buffer provide_send(Rank&, int slot, int r, size_t) override { return arena + ((size_t)slot * ring + r) * payload_cap; }
    // The subtree layout's exchange (imt_transport_all_gather): gather number k uses staging slot (k - 1) % IPC_AUX_RING of
    // every rank's arena.  My payload goes into my slot (once every peer has copied the slot's previous contents), my
    // `aux_packed` says so; each peer's payload is read out of its slot once its `aux_packed` has reached k; my
    // `aux_copied` tells the peers.  GPU-polled between different GPUs, host-polled between ranks that share one.
    int small_gather(const void* send, void* recv, size_t bytes, Stream st_) override {
        if (!connected) return ctx->fail(IMT_ERR_ARG, "imt_transport_ipc_connect first");
        // A GPU-side wait of an EARLIER gather that gave up set the sticky error word: every copy and acknowledgement
        // enqueued since has been skipped on the device and would be again -- say so instead of returning IMT_OK over a
        // receive buffer nobody wrote.  A wait of THIS gather that gives up is seen by the next call, or by
        // imt_transport_poll_error after the caller has synchronised the stream.
        int rc0 = poll_error();
        if (rc0) return rc0;
        hipStream_t st = (hipStream_t)st_;
        const uint64_t k = ++aux_seq;
        const int r = (int)((k - 1) % IPC_AUX_RING);
        uint8_t* mine = arena + aux_off + (size_t)r * IPC_AUX_BYTES;
        auto wait_all = [&](bool copied, uint64_t value) -> int {
            if (host_poll) {
                for (int h = 0; h < world; h++) {
                    if (h == rank) continue;
                    const uint64_t* f = copied ? &peers[h].shm->aux_copied[r] : &peers[h].shm->aux_packed[r];
                    if (!host_wait([&] { return __atomic_load_n(f, __ATOMIC_ACQUIRE) >= value; }))
                        return ctx->fail(IMT_ERR_TIMEOUT, "IPC transport: rank %d did not reach all-gather %llu within %.0f s", h, (unsigned long long)value, timeout_s);
                }
                return IMT_OK;
            }
            imt::launch::FlagWait w{};
            for (int d = 1; d < world; d++) {
                const int h = (rank + d) % world;
                w.flag[w.n++] = copied ? &peers[h].shm_dev->aux_copied[r] : &peers[h].shm_dev->aux_packed[r];
            }
            w.value = value;
            w.timeout_ticks = timeout_ticks;
            w.err = &my_shm_dev->err;
            imt::launch::flag_wait(st, w);
            IMT_HIP(ctx, hipGetLastError());
            return IMT_OK;
        };
        int rc;
        if (k > IPC_AUX_RING && (rc = wait_all(true, k - IPC_AUX_RING))) return rc;
        imt::launch::copy16(st, mine, send, bytes, &my_shm_dev->err);
        imt::launch::copy16(st, (uint8_t*)recv + (size_t)rank * bytes, send, bytes, &my_shm_dev->err);
        imt::launch::flag_set_checked(st, &my_shm_dev->aux_packed[r], k, &my_shm_dev->err);
        IMT_HIP(ctx, hipGetLastError());
        if ((rc = wait_all(false, k))) return rc;
        for (int d = 1; d < world; d++) {
            const int h = (rank + d) % world;
            imt::launch::copy16(st, (uint8_t*)recv + (size_t)h * bytes, peers[h].arena + aux_off + (size_t)r * IPC_AUX_BYTES, bytes, &my_shm_dev->err);
        }
        imt::launch::flag_set_checked(st, &my_shm_dev->aux_copied[r], k, &my_shm_dev->err);
        IMT_HIP(ctx, hipGetLastError());
        return IMT_OK;
    }
    int wait_peers(hipStream_t st, int i, uint64_t k, bool copied) {
        imt::launch::FlagWait w{};
        for (int d = 1; d < world; d++) {
            const int h = (rank + d) % world;
            w.flag[w.n++] = copied ? &peers[h].shm_dev->copied[i] : &peers[h].shm_dev->packed[i];
        }
        w.value = k;
        w.timeout_ticks = timeout_ticks;
        w.err = &my_shm_dev->err;
        HostTimer::Scope sc(ht, copied ? 3 : 1);
        imt::launch::flag_wait(st, w);
        IMT_HIP(ctx, hipGetLastError());
        return IMT_OK;
    }
    int all_gather(Rank& rk, int slot, int r, size_t bytes, Stream st_) override {
        hipStream_t st = (hipStream_t)st_;
        const int i = ei(slot, r);
        const uint64_t k = ++seq[i];
        // st is already behind the unit that packed my send buffer
        {
            HostTimer::Scope sc(ht, 0);
            imt::launch::flag_set(st, &my_shm_dev->packed[i], k);
        }
        uint8_t* recv = (uint8_t*)rk.recv[rk.at(slot, r)];
        const size_t off = ((size_t)slot * ring + r) * payload_cap;
        if (host_poll) {                               // the worker copies each payload when its owner says it is there
            IMT_HIP(ctx, hipGetLastError());
            // the receive buffer is free again once st gets here (st is behind this tick's unit, hence behind the apply
            // that read the buffer's previous contents): the worker's copies wait for that
            IMT_HIP(ctx, hipEventRecord(ready_ev[i], st));
            {
                std::lock_guard<std::mutex> lk(mu);
                jobs.push_back(Job{slot, i, bytes, off, k, recv, 0u, false});
            }
            cv.notify_one();
            return IMT_OK;
        }
        int rc = wait_peers(st, i, k, false);
        if (rc) return rc;
        {                                              // every peer's payload in ONE launch: the peers' links side by side
            imt::launch::CopyJobs jobs{};
            for (int d = 1; d < world; d++) {
                const int h = (rank + d) % world;
                jobs.dst[jobs.n] = recv + (size_t)h * bytes;
                jobs.src[jobs.n++] = peers[h].arena + off;
            }
            HostTimer::Scope sc(ht, 2);
            imt::launch::copy16_multi(st, jobs, bytes, &my_shm_dev->err);
        }
        HostTimer::Scope sc(ht, 0);
        // not published once a wait has given up: the peers must not overwrite send buffers this rank has not read
        imt::launch::flag_set_checked(st, &my_shm_dev->copied[i], k, &my_shm_dev->err);
        IMT_HIP(ctx, hipGetLastError());
        return IMT_OK;
    }
    int fence(Rank&, int slot, int r, Stream st_) override {
        const int i = ei(slot, r);
        if (!host_poll) return wait_peers((hipStream_t)st_, i, seq[i], true);
        const uint64_t k = seq[i];
        HostTimer::Scope sc(ht, 3);
        // my gather: every payload's copy is enqueued (the stream then waits for them, not for a peer) ...
        if (!host_wait([&] { return issued[i].load(std::memory_order_acquire) >= k || worker_error.load(); }) || worker_error.load())
            return ctx->fail(IMT_ERR_INTERNAL, "IPC transport: a peer's payload did not arrive within %.0f s (a rank died or hangs)", timeout_s);
        IMT_HIP(ctx, hipStreamWaitEvent((hipStream_t)st_, done_ev[i], 0));
        // ... and my send buffer: every peer has copied it (their GPUs say so in their pages)
        for (int h = 0; h < world; h++) {
            if (h == rank) continue;
            if (!host_wait([&] { return __atomic_load_n(&peers[h].shm->copied[i], __ATOMIC_ACQUIRE) >= k; }))
                return ctx->fail(IMT_ERR_INTERNAL, "IPC transport: rank %d did not copy gather %llu within %.0f s", h, (unsigned long long)k, timeout_s);
        }
        return IMT_OK;
    }
    int poll_error() override {
        const uint32_t e = my_shm ? *(volatile uint32_t*)&my_shm->err : 0;
        if (e) return ctx->fail(IMT_ERR_INTERNAL, "IPC transport: a peer's payload did not arrive in time (flag mask 0x%x): a rank died or hangs", e);
        return IMT_OK;
    }
};

}  // namespace

// ---------------------------------------------------------------------------------------------------- the handles
imt_transport* imt_transport_wrap(Transport* t, imt_ctx* ctx) {
    imt_transport* h = new (std::nothrow) imt_transport();
    if (!h) {
        delete t;
        return nullptr;
    }
    h->impl.reset(t);
    h->ctx = ctx;
    return h;
}

struct imt_sliced {
    World w;
    std::vector<std::unique_ptr<HipBackend>> bes;
    std::vector<std::unique_ptr<Rank>> ranks;
    imt_transport* tp = nullptr;
    size_t max_slice = 0;
    SlicedOptions opt;
    std::string error, first_failure, presets_kept;
    double host_issue_ms = 0, host_wait_ms = 0;      // wall time inside imt_sliced_step: issuing / waiting for the GPU
    bool trees_marked = false;

    int refuse(const char* call) {
        error = std::string(call) + ": this world failed earlier and cannot go on -- destroy it and reload the trees from a checkpoint.  The first failure:\n" + first_failure;
        return IMT_ERR_INTERNAL;
    }

    void mark(bool busy) {
        for (auto& be : bes)
            if (be) imt_itree_mark_sliced(be->tree, busy);
        trees_marked = busy;
    }
    // A host wait ran into the watchdog (IMT_ERR_TIMEOUT) or the transport gave up on a peer: write where the world stands
    // to stderr, once per failure, and keep it for imt_sliced_last_error.  One rank's log is then enough to tell which
    // collective of which round never completed.
    int failed(int rc, const char* call) {
        if (rc == IMT_OK || bes.empty()) return rc;
        if (rc == IMT_ERR_TIMEOUT || w.poisoned) {
            const int rank = ranks.empty() || !ranks[0] ? -1 : ranks[0]->rank;
            imt_ctx* c0 = bes[0]->ctx;
            std::string why = c0->last_error.empty() ? imt_transport_last_error(tp) : c0->last_error;
            char head[256];
            snprintf(head, sizeof head, "[imt sliced rank %d] %s failed with %d: ", rank, call, rc);
            error = std::string(head) + why + "\n" + describe(w);
            char tail[160];
            snprintf(tail, sizeof tail, "  host inside imt_sliced_step so far: %.1f ms issuing, %.1f ms waiting\n", host_issue_ms, host_wait_ms);
            error += tail;
            if (first_failure.empty()) first_failure = error;
            fputs(error.c_str(), stderr);
            fflush(stderr);
        }
        return rc;
    }
};

extern "C" {

int imt_transport_custom_create(const imt_transport_ops* ops, imt_transport** out) {
    if (!ops || !ops->all_gather || !out) return IMT_ERR_ARG;
    *out = imt_transport_wrap(new (std::nothrow) CustomTransport(*ops), nullptr);
    return *out ? IMT_OK : IMT_ERR_ALLOC;
}

int imt_transport_local_create(imt_transport** out) {
    if (!out) return IMT_ERR_ARG;
    *out = imt_transport_wrap(new (std::nothrow) LocalTransport(), nullptr);
    return *out ? IMT_OK : IMT_ERR_ALLOC;
}

size_t imt_transport_ipc_blob_bytes(void) { return sizeof(IpcBlob); }

int imt_transport_ipc_create(imt_ctx* ctx, int world, int rank, unsigned depth, size_t max_slice, int lag, imt_transport** out,
                             void* blob_out) {
    if (!ctx || !out || !blob_out) return IMT_ERR_ARG;
    *out = nullptr;
    if (world < 2 || rank < 0 || rank >= world || max_slice == 0) return ctx->fail(IMT_ERR_ARG, "IPC transport: world >= 2, 0 <= rank < world");
    IpcTransport* t = new (std::nothrow) IpcTransport(ctx, world, rank);
    if (!t) return ctx->fail(IMT_ERR_ALLOC, "out of host memory");
    int rc = t->create(depth, max_slice, lag, (IpcBlob*)blob_out);
    if (rc) {
        delete t;
        return rc;
    }
    *out = imt_transport_wrap(t, ctx);
    return *out ? IMT_OK : IMT_ERR_ALLOC;
}

int imt_transport_ipc_connect(imt_transport* tp, const void* all_blobs) {
    if (!tp || !all_blobs) return IMT_ERR_ARG;
    IpcTransport* t = dynamic_cast<IpcTransport*>(tp->impl.get());
    if (!t) return IMT_ERR_ARG;
    return t->connect((const IpcBlob*)all_blobs);
}

int imt_transport_set_option(imt_transport* tp, int option, long value) {
    if (!tp) return IMT_ERR_ARG;
    IpcTransport* t = dynamic_cast<IpcTransport*>(tp->impl.get());
    switch (option) {
        case IMT_TRANSPORT_OPT_TIMEOUT_MS:
            if (value <= 0) return IMT_ERR_RANGE;
            if (t) t->set_timeout((double)value * 1e-3);
            return IMT_OK;                              // the other transports have no wait of their own to bound
        case IMT_TRANSPORT_OPT_HOST_POLL:
            if (!t || t->connected || value < -1 || value > 1) return IMT_ERR_ARG;      // before imt_transport_ipc_connect
            t->host_poll_option = (int)value;
            return IMT_OK;
    }
    return IMT_ERR_ARG;
}

int imt_transport_all_gather(imt_transport* tp, const void* send, void* recv, size_t bytes, void* hip_stream) {
    if (!tp || !send || !recv) return IMT_ERR_ARG;
    auto fail = [&](int code, const char* msg) {
        tp->error = msg;
        return code;
    };
    if (bytes == 0 || (bytes & 15u) || bytes > IPC_AUX_BYTES) return fail(IMT_ERR_RANGE, "imt_transport_all_gather: bytes is a multiple of 16, at most 4096");
    if (tp->users > 0) return fail(IMT_ERR_ARG, "imt_transport_all_gather: an imt_sliced is using the transport");
    if (dynamic_cast<LocalTransport*>(tp->impl.get())) return fail(IMT_ERR_ARG, "imt_transport_all_gather: the local transport has every rank in this process (copy instead)");
    if (!hip_stream && !tp->ctx) return fail(IMT_ERR_ARG, "imt_transport_all_gather: this transport has no context of its own, name a stream");
    if (tp->ctx) {
        int rc = tp->ctx->set_device();
        if (rc) return rc;
    }
    tp->error.clear();
    return tp->impl->small_gather(send, recv, bytes, hip_stream ? hip_stream : (void*)tp->ctx->stream);
}

int imt_transport_poll_error(imt_transport* tp) {
    if (!tp) return IMT_ERR_ARG;
    if (tp->ctx) {
        int rc = tp->ctx->set_device();
        if (rc) return rc;
    }
    const int rc = tp->impl->poll_error();
    if (rc && tp->ctx) tp->error = tp->ctx->last_error;
    return rc;
}

int imt_transport_destroy(imt_transport* tp) {
    if (!tp) return IMT_OK;
    if (tp->users > 0) {            // an imt_sliced still holds it: destroying it now would leave that world with a dangling pointer
        tp->error = "imt_transport_destroy: an imt_sliced still uses this transport (imt_sliced_destroy first)";
        return IMT_ERR_ARG;
    }
    if (tp->abandoned) {            // its last world never drained: host-side cleanup only, the device side is leaked
        tp->impl->abandon();
        (void)tp->impl.release();
    }
    delete tp;
    return IMT_OK;
}

const char* imt_transport_last_error(const imt_transport* tp) {
    if (!tp) return "";
    if (!tp->error.empty()) return tp->error.c_str();
    return tp->ctx ? tp->ctx->last_error.c_str() : "";
}

int imt_sliced_set_option(imt_sliced* s, int option, long value) {
    if (option == IMT_SLICED_OPT_RESET) {
        if (s || value != 0) return s ? IMT_ERR_ARG : IMT_ERR_RANGE;
        std::lock_guard<std::mutex> lk(g_defaults_mu);
        g_defaults = SlicedOptions();
        return IMT_OK;
    }
    if (!option_value_ok(option, value)) return IMT_ERR_RANGE;
    long* f = nullptr;
    if (!s) {                       // the defaults later imt_sliced_create calls start from
        std::lock_guard<std::mutex> lk(g_defaults_mu);
        int rc = option_field(g_defaults, option, &f);
        if (rc) return rc;
        *f = value;
        g_defaults.set_mask |= 1ul << option;
        return IMT_OK;
    }
    // an existing world: only what does not change its streams
    if (option != IMT_SLICED_OPT_PREP_STREAM && option != IMT_SLICED_OPT_WATCHDOG_MS && option != IMT_SLICED_OPT_TIMING) return IMT_ERR_ARG;
    int rc = option_field(s->opt, option, &f);
    if (rc) return rc;
    *f = value;
    for (auto& be : s->bes) {
        if (!be) continue;
        option_field(be->opt, option, &f);
        *f = value;
        if (option == IMT_SLICED_OPT_WATCHDOG_MS) imt_itree_set_slice_wait_limit(be->tree, (double)value);
    }
    return IMT_OK;
}

void imt_sliced_destroy(imt_sliced* s) {
    if (!s) return;
    if (s->opt.timing)
        fprintf(stderr, "[imt sliced rank %d] host ms over %llu rounds: apply %.1f  compute %.1f  send %.1f  (issue %.1f, waiting in prepare %.1f of which "
                        "%.1f for a plan set's previous slice)\n",
                s->ranks.empty() || !s->ranks[0] ? -1 : s->ranks[0]->rank, (unsigned long long)s->w.n_rounds, s->w.phase_ms[0], s->w.phase_ms[1],
                s->w.phase_ms[2], s->host_issue_ms, s->host_wait_ms, s->bes.empty() || !s->bes[0] ? 0.0 : imt_itree_slice_backpressure_ms(s->bes[0]->tree));
    // a poisoned world is not flushed (nothing can be issued any more); its replicas stay marked: they hold half a step
    const bool clean = !s->w.poisoned;
    if (clean && !s->w.ranks.empty() && s->w.n_rounds) s->w.flush();
    bool hung = false;
    for (auto& be : s->bes)
        if (be) {
            // bounded by the watchdog.  If it runs out, something of this world still runs on the device (a collective
            // whose peer never came, a wait): nothing below may then WAIT for the device -- hipFree does -- so the world's
            // buffers and streams are leaked, its transport is marked (imt_transport_destroy leaks its device side too) and
            // the replicas stay marked busy.  The process is expected to exit (imt.h, IMT_ERR_TIMEOUT).
            if (be->sync() == IMT_ERR_TIMEOUT) hung = true;
            if (clean && !s->w.poisoned && !be->hung) imt_itree_mark_sliced(be->tree, false);
        }
    if (hung) {
        for (auto& be : s->bes)
            if (be) be->hung = true;
        if (s->tp) s->tp->abandoned = true;
        fputs("[imt sliced] imt_sliced_destroy: the world's streams did not drain within the watchdog's limit; its device buffers and streams "
              "are left allocated (hipFree would wait for the device).  This process should exit; recover in a fresh one.\n", stderr);
    }
    for (auto& r : s->ranks)
        if (r) r->destroy();
    if (s->tp) s->tp->users--;
    delete s;
}

int imt_sliced_create(imt_itree* const* trees, int n_local, int world, int first_rank, imt_transport* tp, size_t max_slice,
                      int lag, imt_sliced** out) {
    if (!out) return IMT_ERR_ARG;
    *out = nullptr;
    if (!trees || !tp || n_local < 1 || !trees[0]) return IMT_ERR_ARG;
    imt_ctx* c0 = imt_itree_ctx(trees[0]);
    if (world < 1 || first_rank < 0 || first_rank + n_local > world || max_slice == 0 || (n_local != 1 && n_local != world))
        return c0->fail(IMT_ERR_ARG, "imt_sliced_create: n_local is 1 or world, 0 <= first_rank, first_rank + n_local <= world");
    const unsigned depth = imt_itree_depth(trees[0]);
    for (int k = 0; k < n_local; k++) {
        if (!trees[k]) return c0->fail(IMT_ERR_ARG, "null tree");
        if (imt_itree_depth(trees[k]) != depth || !imt_itree_is_plain(trees[k]))
            return c0->fail(IMT_ERR_ARG, "replicas have one depth and are not placed / partitioned");
        for (int j = 0; j < k; j++)
            if (trees[j] == trees[k] || imt_itree_ctx(trees[j]) == imt_itree_ctx(trees[k]))
                return c0->fail(IMT_ERR_ARG, "every replica needs its own tree on its own context");
    }
    if (n_local == world && world > 1 && !dynamic_cast<LocalTransport*>(tp->impl.get()))
        return c0->fail(IMT_ERR_ARG, "all ranks in one process use the local transport");
    if (n_local == 1 && world > 1 && dynamic_cast<LocalTransport*>(tp->impl.get()))
        return c0->fail(IMT_ERR_ARG, "the local transport needs every rank in this process (n_local = world)");
    // one world per transport at a time: the transports keep per-world state (peer tables, sequence counters, ring sizes)
    if (tp->users > 0) return c0->fail(IMT_ERR_ARG, "the transport is in use by another imt_sliced (one world per transport)");
    std::unique_ptr<imt_sliced> s(new (std::nothrow) imt_sliced());
    if (!s) return c0->fail(IMT_ERR_ALLOC, "out of host memory");
    if (!s->w.sc.init(world, (int)depth + 1, lag))
        return c0->fail(IMT_ERR_RANGE, "world %d, depth %u, lag %d would keep more than %d steps in flight (or is no schedule)", world, depth,
                        lag, ROUNDS);
    s->tp = tp;
    s->max_slice = max_slice;
    s->opt = effective_options();
    // Priority pools (IMT_SLICED_OPT_POOLS).  The runtime keeps a set of hardware queues PER stream priority, and everything
    // else in a process lives in the normal one: the host's streams, and RCCL's -- a communicator creates three streams of
    // its own (tools/microbench/rccl_streams_probe.hip) and brackets every collective with one of them (the user's stream
    // waits for it, it waits for the kernel).  On a round's queue such a stream puts the round's hash kernels behind every
    // collective of that communicator and the collective behind the round's backlog: 15 - 30 % in the queue model
    // (profiles/r05_hwq_model.txt), and nothing this library could see or repair.  So for one process per GPU the round
    // streams go to the HIGH-priority pool, where no foreign stream is, the collectives' streams to the LOW one (a gather
    // overlaps its round's next units: IMT_SLICED_OPT_COMM_PLACEMENT's reason), and the preparation to the round's stream
    // (it would crawl at low priority).  One rank of 2 / 4 / 8 alone on the GPU: within 1 % of everything in one pool
    // (profiles/r05_emu_priority_pools.txt).  Replicas of one process have no peers to wait for: one pool.
    if (s->opt.pools < 0) s->opt.pools = (n_local == 1 && world > 1) ? 1 : 0;
    // The presets fill in what the caller has NOT named: an option set explicitly (imt_sliced_set_option(NULL, ...)) keeps
    // the caller's value, and imt_sliced_last_error / the placement note say which preset was left out.
    std::string kept;
    auto preset = [&](int option, long* field, long value, const char* name) {
        if (s->opt.set_mask & (1ul << option)) {
            if (*field != value) {
                char b[160];
                snprintf(b, sizeof b, "%sIMT_SLICED_OPT_POOLS %ld: %s stays at the caller's %ld (the preset is %ld)", kept.empty() ? "" : "; ", s->opt.pools, name, *field, value);
                kept += b;
            }
            return;
        }
        *field = value;
    };
    if (s->opt.pools == 1) {
        preset(IMT_SLICED_OPT_ROUND_PRIORITIES, &s->opt.round_priorities, 3, "ROUND_PRIORITIES");
        preset(IMT_SLICED_OPT_COMM_PRIORITY, &s->opt.comm_priority, 8, "COMM_PRIORITY");       // clamped to the lowest the device has
        preset(IMT_SLICED_OPT_PREP_STREAM, &s->opt.prep_stream, 1, "PREP_STREAM");
    } else if (s->opt.pools == 2) {
        // the one-pool layout moved into the HIGH pool: the collectives' streams on their rounds' queues -- no foreign stream
        // on a round's queue either, no barrier packet between a round and its gathers (same queue: in order), 4 % more
        // per rank on one GPU than the three pools, but every tick a barrier across ranks again (IMT_SLICED_OPT_COMM_PLACEMENT)
        preset(IMT_SLICED_OPT_ROUND_PRIORITIES, &s->opt.round_priorities, 3, "ROUND_PRIORITIES");
        preset(IMT_SLICED_OPT_COMM_PRIORITY, &s->opt.comm_priority, -8, "COMM_PRIORITY");      // clamped to the highest
        preset(IMT_SLICED_OPT_COMM_PLACEMENT, &s->opt.comm_placement, 1, "COMM_PLACEMENT");
    }
    s->presets_kept = kept;
    tp->users++;
    for (int k = 0; k < n_local; k++) {
        s->bes.emplace_back(new (std::nothrow) HipBackend(trees[k]));
        s->ranks.emplace_back(new (std::nothrow) Rank());
        HipBackend* be = s->bes.back().get();
        Rank* rk = s->ranks.back().get();
        int rc;
        if (!be || !rk) rc = c0->fail(IMT_ERR_ALLOC, "out of host memory");
        else if (!(rc = be->init(s->opt, tp->impl->channels())) && !(rc = rk->init(be, tp->impl.get(), world, first_rank + k, max_slice)))
            rc = rk->build(s->w.sc);
        if (rc) {
            imt_sliced_destroy(s.release());
            return rc;
        }
        imt_itree_set_slice_poison(be->tree, tp->impl->poison_word());
        s->w.ranks.push_back(rk);
    }
    *out = s.release();
    return IMT_OK;
}

int imt_sliced_step(imt_sliced* s, const void* vals, size_t n, const imt_insert_out* outs, unsigned flags, uint64_t* round_out) {
    if (!s) return IMT_ERR_ARG;
    imt_ctx* c0 = s->bes[0]->ctx;
    if (!vals) return c0->fail(IMT_ERR_ARG, "null vals");
    if (n == 0 || n > s->max_slice) return c0->fail(IMT_ERR_RANGE, "a step is world x n values with 0 < n <= max_slice = %zu", s->max_slice);
    if (flags & ~(IMT_FMT_MASK | IMT_SIB_ITEM_MAJOR | IMT_INPUTS_READY | IMT_DEVICE_PTRS))
        return c0->fail(IMT_ERR_ARG, "imt_sliced_step takes IMT_FMT_*, IMT_SIB_ITEM_MAJOR, IMT_INPUTS_READY");
    if (s->w.poisoned) return s->refuse("imt_sliced_step");
    (void)s->tp->impl->take_wait_ms();                     // waits of imt_sliced_wait / _flush are not this call's
    for (auto& be : s->bes) (void)imt_itree_take_wait_ms(be->tree);
    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t rounds_before = s->w.n_rounds;
    const int rc = s->w.step(vals, n, outs, flags, round_out);
    if (rc == IMT_ERR_TIMEOUT) s->w.poisoned = true;       // the preparation's own wait gave up: its kernels are still in flight
    // the replicas are mid-step as soon as the step has been opened (the index has moved ahead of the stored tree), whether
    // or not every tick of the period could then be issued -- and stay closed to ordinary calls once the world has failed
    if (s->w.n_rounds != rounds_before || s->w.poisoned) s->mark(true);
    const double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    double waited = s->tp->impl->take_wait_ms();           // the host-polled transport: waiting for peers' payloads
    for (auto& be : s->bes) waited += imt_itree_take_wait_ms(be->tree);
    s->host_wait_ms += waited;
    s->host_issue_ms += total - waited;
    return s->failed(rc, "imt_sliced_step");
}

int imt_sliced_wait(imt_sliced* s, int local_rank, uint64_t round) {
    if (!s || local_rank < 0) return IMT_ERR_ARG;
    if (s->w.poisoned) return s->refuse("imt_sliced_wait");
    int rc = s->w.wait_round((size_t)local_rank, round);
    if (rc == IMT_ERR_TIMEOUT) s->w.poisoned = true;
    return s->failed(rc, "imt_sliced_wait");
}

int imt_sliced_flush(imt_sliced* s) {
    if (!s) return IMT_ERR_ARG;
    if (s->w.poisoned) return s->refuse("imt_sliced_flush");
    int rc = s->w.flush();
    if (rc == IMT_ERR_TIMEOUT) s->w.poisoned = true;
    if (rc == IMT_OK) s->mark(false);      // every replica holds the whole step now
    return s->failed(rc, "imt_sliced_flush");
}

int imt_sliced_dump(imt_sliced* s, char* out, size_t cap) {
    if (!s || (!out && cap)) return IMT_ERR_ARG;
    const std::string d = describe(s->w);
    if (cap) {
        const size_t n = std::min(cap - 1, d.size());
        memcpy(out, d.data(), n);
        out[n] = 0;
    }
    return (int)std::min<size_t>(d.size(), 0x7fffffff);
}

int imt_sliced_get_info(const imt_sliced* s, imt_sliced_info* o) {
    if (!s || !o) return IMT_ERR_ARG;
    const Schedule& sc = s->w.sc;
    o->world = sc.world;
    o->n_local = (int)s->ranks.size();
    o->lag = sc.lag;
    o->period = sc.period;
    o->gathers_per_round = sc.gathers;
    o->round_ticks = sc.round_ticks;
    o->rounds_in_flight = (sc.round_ticks + sc.period - 1) / sc.period;
    o->payload_bytes = s->ranks[0]->payload_cap;
    o->rounds = s->w.n_rounds;
    o->collectives = s->tp->impl->collectives;
    o->bytes_gathered = s->tp->impl->bytes_moved;
    o->host_issue_ms = s->host_issue_ms;
    o->host_wait_ms = s->host_wait_ms;
    const HipBackend& be = *s->bes[0];
    o->placement = be.placement;
    o->hw_queues = be.n_queues;
    o->comm_streams = be.n_comm;
    o->streams_recreated = be.streams_recreated;
    o->pools = (int)s->opt.pools;
    for (auto& b : s->bes) {                 // the worst of the local replicas
        if (b->placement > o->placement) o->placement = b->placement;
        o->streams_recreated = std::max(o->streams_recreated, b->streams_recreated);
    }
    for (int i = 0; i < ROUNDS && i < IMT_SLICED_ROUNDS; i++) {
        o->queue_map[0][i] = be.q_round[i];
        o->queue_map[1][i] = be.q_comm[i];
        o->queue_map[2][i] = be.q_apply[i];
    }
    return IMT_OK;
}

const char* imt_sliced_last_error(const imt_sliced* s) {
    if (!s) return "";
    if (!s->error.empty()) return s->error.c_str();
    for (auto& be : s->bes)
        if (be && be->placement == IMT_SLICED_PLACEMENT_DEGRADED && be->ctx->last_error.empty()) return be->placement_note.c_str();
    if (!s->presets_kept.empty() && !s->bes.empty() && s->bes[0] && s->bes[0]->ctx->last_error.empty()) return s->presets_kept.c_str();
    for (auto& be : s->bes)
        if (be && !be->ctx->last_error.empty()) return be->ctx->last_error.c_str();
    return imt_transport_last_error(s->tp);
}

}  // extern "C"
