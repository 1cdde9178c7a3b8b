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
#include "imt_flags.hpp"
#include "imt_itree_internal.hpp"
#include "imt_sliced_sched.hpp"
#include "imt_sliced_transport.hpp"

using namespace imt::sliced;

namespace {

// ---------------------------------------------------------------------------------------------- one replica over HIP
struct HipBackend : Backend {
    imt_itree* tree;
    imt_ctx* ctx;
    hipStream_t rs[ROUNDS] = {}, cs[ROUNDS] = {}, aps[ROUNDS] = {};     // round, collective and apply streams
    int n_comm = ROUNDS;

    explicit HipBackend(imt_itree* t) : tree(t), ctx(imt_itree_ctx(t)) {}
    int init() {
        int rc = ctx->set_device();
        if (rc) return rc;
        // Measured on one MI355X with in-process replicas (profiles/r04_sliced_stream_matrix.txt): the four round streams
        // at EQUAL priority (2.93-2.95 M insertions/s at world 1, 2.86 at world 2) beat the batch pipeline's scheme of one
        // normal + three high (2.49-2.64 / 2.64-2.83) -- rounds are whole slices apart here, not one level, and a
        // high-priority round starves the others; where the collectives are enqueued (the round's own stream, one or four
        // extra streams, normal or high priority) moves the rate by < 2 % with equal round priorities.  Default: four
        // normal-priority streams for the collectives, so that a gather never sits in front of a hash kernel.
        // Knobs: IMT_SLICED_COMM_STREAMS (0 = enqueue a round's gathers on the round's own stream), IMT_SLICED_COMM_PRIO,
        // IMT_SLICED_ROUND_PRIO=pipe.
        if (const char* e = getenv("IMT_SLICED_COMM_STREAMS")) n_comm = std::max(0, std::min(ROUNDS, atoi(e)));
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
        const char* rp = getenv("IMT_SLICED_ROUND_PRIO");
        const bool equal = !(rp && !strcmp(rp, "pipe"));
        int comm_prio = std::max(greatest, std::min(least, 0));
        if (const char* e = getenv("IMT_SLICED_COMM_PRIO")) comm_prio = std::max(greatest, std::min(least, atoi(e)));
        for (int i = 0; i < ROUNDS; i++) {
            // different priorities for the round streams, like the batch pipeline's: the runtime may otherwise map them to
            // one hardware queue, which serialises them
            const int prio = equal ? 0 : std::max(greatest, std::min(least, 0 - i));
            IMT_HIP(ctx, hipStreamCreateWithPriority(&rs[i], hipStreamNonBlocking, prio));
            ctx->side_streams.push_back(rs[i]);
        }
        for (int i = 0; i < n_comm; i++) {
            IMT_HIP(ctx, hipStreamCreateWithPriority(&cs[i], hipStreamNonBlocking, comm_prio));
            ctx->side_streams.push_back(cs[i]);
        }
        // The schedule (imt_sliced_sched.hpp) lets the other ranks' write-backs be applied on a stream of their own per round
        // slot; here they stay on the round's stream unless IMT_SLICED_APPLY_STREAMS=1.  Measured with one rank of an
        // 8-rank run alone on the GPU (tools/rank_emulation.py, profiles/r04_rank_emulation.txt): what moved the rate was
        // the PRECISE cross-round waits (round R + 1's unit q behind round R's apply of tick q + world * lag and its own
        // unit q + 1, instead of behind round R's whole tick q + world * lag, which made two rounds march in lockstep):
        // 2.66 -> 2.85 M insertions/s per rank.  Four more streams on the runtime's four hardware queues cost some of it
        // back (2.77 / 2.67 first / last rank), helper streams placed on other rounds' queues or at high priority more.
        const char* ae = getenv("IMT_SLICED_APPLY_STREAMS");
        if (ae && atoi(ae) != 0)
            for (int i = 0; i < ROUNDS; i++) {
                IMT_HIP(ctx, hipStreamCreateWithPriority(&aps[i], hipStreamNonBlocking, 0));
                ctx->side_streams.push_back(aps[i]);
            }
        return IMT_OK;
    }
    ~HipBackend() override {
        if (ctx->set_device()) return;
        auto& ss = ctx->side_streams;
        for (hipStream_t* arr : {rs, cs, aps})
            for (int i = 0; i < ROUNDS; i++)
                if (arr[i]) {
                    hipStreamSynchronize(arr[i]);
                    ss.erase(std::remove(ss.begin(), ss.end(), arr[i]), ss.end());
                    hipStreamDestroy(arr[i]);
                }
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
        IMT_HIP(ctx, hipEventSynchronize((hipEvent_t)e));
        return IMT_OK;
    }
    int alloc(size_t bytes, Buffer* out) override {
        void* p = nullptr;
        IMT_HIP(ctx, hipMalloc(&p, bytes));
        IMT_HIP(ctx, hipMemset(p, 0, bytes));
        *out = p;
        return IMT_OK;
    }
    void free_buffer(Buffer b) override { hipFree(b); }
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
        // IMT_SLICED_PREP_STREAM = comm (default) / round / side.
        static const char* where = getenv("IMT_SLICED_PREP_STREAM");
        void* st = n_comm ? (void*)cs[slot % n_comm] : (void*)rs[slot];
        if (where && !strcmp(where, "side")) st = nullptr;
        if (where && !strcmp(where, "round")) st = rs[slot];
        imt_itree_set_slice_prep_stream(tree, st);
        const int rc = imt_itree_slice_prepare(tree, vals, nb, no, na, out, flags | IMT_DEVICE_PTRS, slice, nullptr);
        imt_itree_set_slice_prep_stream(tree, nullptr);
        return rc;
    }
    int unit(int slice, unsigned q, Buffer payload, Stream s) override { return imt_itree_slice_unit(tree, slice, q, payload, s); }
    int apply_gathered(Buffer g, size_t stride, int count, const uint64_t* sb, const uint64_t* n, const int32_t* units,
                       Stream s) override {
        return imt_itree_slice_apply_gathered(tree, g, stride, (size_t)count, sb, n, units, s);
    }
    int sync() override {
        int rc = ctx->set_device();
        if (rc) return rc;
        for (hipStream_t* arr : {rs, cs, aps})
            for (int i = 0; i < ROUNDS; i++)
                if (arr[i]) IMT_HIP(ctx, hipStreamSynchronize(arr[i]));
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

struct IpcBlob {
    hipIpcMemHandle_t mem;
    char shm_name[64];
    char bus_id[32];                 // PCI bus id of the rank's GPU: ranks that SHARE a device wait on the host (below)
    uint64_t arena_bytes;
    int32_t rank, world, ring, pid;
};
struct IpcShm {
    uint64_t packed[IPC_NEV], copied[IPC_NEV];
    uint32_t err;
};
constexpr size_t IPC_SHM_BYTES = (sizeof(IpcShm) + 4095) & ~(size_t)4095;

struct HostTimer {          // IMT_SLICED_TIMING=1: where the host's time inside the transport goes (printed at destroy)
    double ms[6] = {0};
    uint64_t n[6] = {0};
    bool on = getenv("IMT_SLICED_TIMING") != nullptr;
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
    uint8_t* arena = nullptr;                    // [ROUNDS][ring][payload_cap]
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
        *host = (IpcShm*)p;
        hipError_t e = hipHostRegister(p, IPC_SHM_BYTES, hipHostRegisterMapped | hipHostRegisterPortable);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void**)dev, p, 0);
        if (e != hipSuccess) return ctx->hip_fail(e, "hipHostRegister(flag page)");
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
        if (const char* e = getenv("IMT_IPC_TIMEOUT_S")) timeout_s = atof(e);
        timeout_ticks = (uint64_t)(timeout_s * 1e3 * khz);
        if (hipDeviceGetPCIBusId(my_bus, (int)sizeof my_bus, ctx->device) != hipSuccess) my_bus[0] = 0;
        payload_cap = imt_itree_slice_payload_bytes(max_slice);
        const size_t bytes = (size_t)ROUNDS * ring * payload_cap;
        IMT_HIP(ctx, hipMalloc((void**)&arena, bytes));
        IMT_HIP(ctx, hipMemset(arena, 0, bytes));
        std::memset(blob, 0, sizeof *blob);
        IMT_HIP(ctx, hipIpcGetMemHandle(&blob->mem, arena));
        char name[64];
        snprintf(name, sizeof name, "/imt_ipc_%d_%d_%llx", (int)getpid(), rank,
                 (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
        if ((rc = map_page(name, true, &my_shm, &my_shm_dev))) return rc;
        shm_name = name;
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
            if (b.rank != h || b.world != world || b.ring != ring)
                return ctx->fail(IMT_ERR_ARG, "IPC blob %d does not describe rank %d of this world", h, h);
            if (b.pid == (int)getpid()) return ctx->fail(IMT_ERR_ARG, "the IPC transport joins PROCESSES; ranks of one process use the local transport");
            Peer& p = peers[h];
            IMT_HIP(ctx, hipIpcOpenMemHandle((void**)&p.arena, b.mem, hipIpcMemLazyEnablePeerAccess));
            if ((rc = map_page(b.shm_name, false, &p.shm, &p.shm_dev))) return rc;
            if (my_bus[0] && !strncmp(my_bus, b.bus_id, sizeof my_bus)) host_poll = true;      // a peer on MY device
        }
        if (const char* e = getenv("IMT_IPC_HOST_POLL")) host_poll = atoi(e) != 0;
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
                if (stop && pend.empty()) return;
                if (stop) return;
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
                    imt::launch::copy16(ws[j.slot], j.recv + (size_t)h * j.bytes, peers[h].arena + j.off, j.bytes);
                    j.copied_mask |= 1u << h;
                    progress = true;
                }
                if (j.copied_mask == all) {
                    imt::launch::flag_set(ws[j.slot], &my_shm_dev->copied[j.i], j.k);
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
    ~IpcTransport() override {
        if (ht.on)
            fprintf(stderr, "[imt ipc rank %d] host ms (calls): flag_set %.1f (%llu)  wait packed %.1f (%llu)  memcpy %.1f (%llu)  wait copied %.1f (%llu)\n",
                    rank, ht.ms[0], (unsigned long long)ht.n[0], ht.ms[1], (unsigned long long)ht.n[1], ht.ms[2], (unsigned long long)ht.n[2], ht.ms[3],
                    (unsigned long long)ht.n[3]);
        if (worker.joinable()) {
            {
                std::lock_guard<std::mutex> lk(mu);
                stop = true;
            }
            cv.notify_all();
            worker.join();
        }
        if (ctx->set_device()) return;
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
        if (!shm_name.empty()) shm_unlink(shm_name.c_str());
    }
    int attach(Rank& rk) override {
        if (!connected) return ctx->fail(IMT_ERR_ARG, "imt_transport_ipc_connect first");
        if (rk.world != world || rk.rank != rank || rk.ring != ring || rk.payload_cap != payload_cap)
            return ctx->fail(IMT_ERR_ARG, "the IPC transport was created for another world / depth / max_slice / lag");
        return IMT_OK;
    }
    Buffer provide_send(Rank&, int slot, int r, size_t) override { return arena + ((size_t)slot * ring + r) * payload_cap; }
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
        for (int d = 1; d < world; d++) {              // start with the next rank: spread the reads over the peers
            const int h = (rank + d) % world;
            HostTimer::Scope sc(ht, 2);
            imt::launch::copy16(st, recv + (size_t)h * bytes, peers[h].arena + off, bytes);
        }
        HostTimer::Scope sc(ht, 0);
        imt::launch::flag_set(st, &my_shm_dev->copied[i], k);
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
    std::string error;
    double host_issue_ms = 0, host_wait_ms = 0;      // wall time inside imt_sliced_step: issuing / waiting for the GPU
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

void imt_transport_destroy(imt_transport* tp) { delete tp; }

const char* imt_transport_last_error(const imt_transport* tp) {
    if (!tp) return "";
    if (!tp->error.empty()) return tp->error.c_str();
    return tp->ctx ? tp->ctx->last_error.c_str() : "";
}

void imt_sliced_destroy(imt_sliced* s) {
    if (!s) return;
    if (getenv("IMT_SLICED_TIMING"))
        fprintf(stderr, "[imt sliced rank %d] host ms over %llu rounds: apply %.1f  compute %.1f  send %.1f  (issue %.1f, waiting in prepare %.1f)\n",
                s->ranks.empty() ? -1 : s->ranks[0]->rank, (unsigned long long)s->w.n_rounds, s->w.phase_ms[0], s->w.phase_ms[1], s->w.phase_ms[2],
                s->host_issue_ms, s->host_wait_ms);
    if (!s->ranks.empty() && s->w.n_rounds) s->w.flush();
    for (auto& be : s->bes) {
        be->sync();
        imt_itree_mark_sliced(be->tree, false);
    }
    for (auto& r : s->ranks) r->destroy();
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
    std::unique_ptr<imt_sliced> s(new (std::nothrow) imt_sliced());
    if (!s) return c0->fail(IMT_ERR_ALLOC, "out of host memory");
    if (!s->w.sc.init(world, (int)depth + 1, lag))
        return c0->fail(IMT_ERR_RANGE, "world %d, depth %u, lag %d would keep more than %d steps in flight (or is no schedule)", world, depth,
                        lag, ROUNDS);
    s->tp = tp;
    s->max_slice = max_slice;
    tp->users++;
    for (int k = 0; k < n_local; k++) {
        s->bes.emplace_back(new (std::nothrow) HipBackend(trees[k]));
        s->ranks.emplace_back(new (std::nothrow) Rank());
        HipBackend* be = s->bes.back().get();
        Rank* rk = s->ranks.back().get();
        int rc;
        if (!be || !rk) rc = IMT_ERR_ALLOC;
        else if (!(rc = be->init()) && !(rc = rk->init(be, tp->impl.get(), world, first_rank + k, max_slice)))
            rc = rk->build(s->w.sc);
        if (rc) {
            imt_sliced_destroy(s.release());
            return rc;
        }
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
    (void)s->tp->impl->take_wait_ms();                     // waits of imt_sliced_wait / _flush are not this call's
    for (auto& be : s->bes) (void)imt_itree_take_wait_ms(be->tree);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = s->w.step(vals, n, outs, flags, round_out);
    if (rc == IMT_OK)
        for (auto& be : s->bes) imt_itree_mark_sliced(be->tree, true);
    const double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    double waited = s->tp->impl->take_wait_ms();           // the host-polled transport: waiting for peers' payloads
    for (auto& be : s->bes) waited += imt_itree_take_wait_ms(be->tree);
    s->host_wait_ms += waited;
    s->host_issue_ms += total - waited;
    return rc;
}

int imt_sliced_wait(imt_sliced* s, int local_rank, uint64_t round) {
    if (!s || local_rank < 0) return IMT_ERR_ARG;
    return s->w.wait_round((size_t)local_rank, round);
}

int imt_sliced_flush(imt_sliced* s) {
    if (!s) return IMT_ERR_ARG;
    const int rc = s->w.flush();
    if (rc == IMT_OK)
        for (auto& be : s->bes) imt_itree_mark_sliced(be->tree, false);     // every replica holds the whole step now
    return rc;
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
    return IMT_OK;
}

const char* imt_sliced_last_error(const imt_sliced* s) {
    if (!s) return "";
    for (auto& be : s->bes)
        if (!be->ctx->last_error.empty()) return be->ctx->last_error.c_str();
    return imt_transport_last_error(s->tp);
}

}  // extern "C"
