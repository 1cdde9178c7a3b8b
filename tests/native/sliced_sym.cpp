// sliced_sym.cpp -- TEST-ONLY build (g++, no HIP) of the product's schedule code, csrc/imt_sliced_sched.hpp, over a
// symbolic backend: every stream / event / buffer / slice operation the schedule issues is forwarded to callbacks
// (Python, tests/sliced_sim.py), where streams are FIFO queues drained by an adversarial scheduler and a replica is, per
// tree level, the list of slices whose write-backs it has seen.  What runs here is the same Rank / World / LocalTransport
// code libimt_hip.so runs over HIP (imt_sliced.cpp); nothing is hashed.
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#include "../../indexed-merkle-tree-halo2_amd/csrc/imt_sliced_sched.hpp"

using namespace imt::sliced;

extern "C" {
typedef struct sym_callbacks {
    int (*record)(int rank, int event, int stream);
    int (*wait)(int rank, int stream, int event);
    int (*event_sync)(int rank, int event);
    int (*alloc)(int rank, int buffer, size_t bytes);
    int (*copy)(int rank, int dst, size_t dst_off, int src_rank, int src, size_t src_off, size_t bytes, int stream);
    uint64_t (*tree_size)(int rank);
    size_t (*unit_bytes)(int rank, uint64_t size_before, size_t n, unsigned unit);
    int (*prepare)(int rank, size_t n_before, size_t n_own, size_t n_after, int slot, int* slice);
    int (*unit)(int rank, int slice, unsigned q, int payload, int stream);
    int (*apply_gathered)(int rank, int gathered, size_t stride, int count, const uint64_t* size_before, const uint64_t* n,
                          const int32_t* units, int stream);
    int (*sync)(int rank);
    // the collective of a distributed world (one rank per process); unused by the in-process transport
    int (*all_gather)(int rank, int slot, int ring, int send, int recv, size_t bytes, int stream);
    // "nobody still reads my send buffer (slot, ring)" enqueued on a stream: a transport without NCCL semantics (IPC)
    int (*fence)(int rank, int slot, int ring, int stream);
} sym_callbacks;
}

namespace {

inline void* handle(int rank, int id) { return (void*)(uintptr_t)((((uint64_t)rank + 1) << 32) | (uint32_t)(id + 1)); }
inline int h_rank(void* h) { return (int)((uintptr_t)h >> 32) - 1; }
inline int h_id(void* h) { return (int)((uintptr_t)h & 0xffffffffu) - 1; }

// which streams a replica has (sym_set_layout): the product's options IMT_SLICED_OPT_COMM_STREAMS / _APPLY_STREAMS.  The
// default here is the widest layout (a stream of its own for everything: the most interleavings for the adversary);
// the product's default is comm_streams = 4, apply_streams = 0.
int g_comm_streams = ROUNDS, g_apply_streams = 1;

struct SymBackend : Backend {
    sym_callbacks cb;
    int rank;
    size_t payload;
    int next_event = 0, next_buffer = 0;
    int n_comm = g_comm_streams, apply = g_apply_streams;
    SymBackend(const sym_callbacks& c, int r, size_t p) : cb(c), rank(r), payload(p) {}
    Stream round_stream(int slot) override { return handle(rank, slot); }
    Stream comm_stream(int slot) override { return n_comm ? handle(rank, ROUNDS + slot % n_comm) : handle(rank, slot); }
    Stream apply_stream(int slot) override { return apply ? handle(rank, 2 * ROUNDS + slot) : handle(rank, slot); }
    int new_event(Event* out) override { *out = handle(rank, next_event++); return IMT_OK; }
    void free_event(Event) override {}
    int record(Event e, Stream s) override { return cb.record(rank, h_id(e), h_id(s)); }
    int wait(Stream s, Event e) override { return cb.wait(rank, h_id(s), h_rank(e) * 100000 + h_id(e)); }
    int event_sync(Event e) override { return cb.event_sync(rank, h_id(e)); }
    int alloc(size_t bytes, Buffer* out) override {
        *out = handle(rank, next_buffer);
        return cb.alloc(rank, next_buffer++, bytes);
    }
    void free_buffer(Buffer) override {}
    int copy(Buffer dst, size_t doff, Buffer src, size_t soff, size_t bytes, Stream s) override {
        return cb.copy(rank, h_id(dst), doff, h_rank(src), h_id(src), soff, bytes, h_id(s));
    }
    uint64_t tree_size() override { return cb.tree_size(rank); }
    size_t payload_bytes(size_t) override { return payload; }
    size_t unit_bytes(uint64_t sb, size_t n, unsigned q) override { return cb.unit_bytes(rank, sb, n, q); }
    int prepare(const void*, size_t nb, size_t no, size_t na, const imt_insert_out*, unsigned, int slot, int* slice) override {
        return cb.prepare(rank, nb, no, na, slot, slice);
    }
    int unit(int slice, unsigned q, Buffer payload_buf, Stream s) override { return cb.unit(rank, slice, q, h_id(payload_buf), h_id(s)); }
    int apply_gathered(Buffer g, size_t stride, int count, const uint64_t* sb, const uint64_t* n, const int32_t* units,
                       Stream s) override {
        return cb.apply_gathered(rank, h_id(g), stride, count, sb, n, units, h_id(s));
    }
    int sync() override { return cb.sync(rank); }
};

struct CallbackTransport : Transport {
    sym_callbacks cb;
    explicit CallbackTransport(const sym_callbacks& c) : cb(c) {}
    int all_gather(Rank& rk, int slot, int r, size_t bytes, Stream st) override {
        const int i = rk.at(slot, r);
        return cb.all_gather(rk.rank, slot, r, h_id(rk.send[i]), h_id(rk.recv[i]), bytes, h_id(st));
    }
    int fence(Rank& rk, int slot, int r, Stream st) override { return cb.fence ? cb.fence(rk.rank, slot, r, h_id(st)) : IMT_OK; }
    int n_channels = 0;
    int channels() const override { return n_channels; }
};

struct SymWorld {
    World w;
    std::vector<std::unique_ptr<SymBackend>> bes;
    std::vector<std::unique_ptr<Rank>> ranks;
    std::unique_ptr<Transport> tp;
};

}  // namespace

extern "C" {

// the stream layout of worlds created from now on: comm_streams 0 .. ROUNDS (0: collectives on the round's stream),
// apply_streams 0 / 1
void sym_set_layout(int comm_streams, int apply_streams) {
    g_comm_streams = comm_streams < 0 ? 0 : comm_streams > ROUNDS ? ROUNDS : comm_streams;
    g_apply_streams = apply_streams != 0;
}

// out = {lag, period, gathers, round_ticks}; 0 on success, -1 if (world, units, lag) is not a schedule
int sym_schedule(int world, int units, int lag, int* out) {
    Schedule s;
    if (!s.init(world, units, lag)) return -1;
    out[0] = s.lag;
    out[1] = s.period;
    out[2] = s.gathers;
    out[3] = s.round_ticks;
    return 0;
}
int sym_unit_of(int world, int units, int lag, int rank, int rt) {
    Schedule s;
    if (!s.init(world, units, lag)) return -2;
    return s.unit_of(rank, rt);
}
int sym_payload_units(int world, int units, int lag, int rt, int32_t* out) {
    Schedule s;
    if (!s.init(world, units, lag)) return -2;
    const bool any = s.payload_units(rt, out);
    return (any ? 1 : 0) | (s.has_gather(rt) ? 2 : 0);
}

// n_local == world: every rank in this process, the in-process transport (LocalTransport: the product's code).
// n_local == 1: rank first_rank of a distributed world; the collective goes through cb.all_gather.
void* sym_world_create(const sym_callbacks* cb, int world, int first_rank, int n_local, size_t max_n, int depth, int lag,
                       size_t payload_bytes) {
    auto sw = std::make_unique<SymWorld>();
    if (!sw->w.sc.init(world, depth + 1, lag)) return nullptr;
    if (n_local == world)
        sw->tp.reset(new LocalTransport());
    else
        sw->tp.reset(new CallbackTransport(*cb));
    for (int k = 0; k < n_local; k++) {
        sw->bes.emplace_back(new SymBackend(*cb, first_rank + k, payload_bytes));
        // like imt_sliced.cpp: round slots that share a channel of the transport share ONE stream
        if (int ch = sw->tp->channels()) {
            if (ch < ROUNDS) sw->bes.back()->n_comm = ch;
        }
        sw->ranks.emplace_back(new Rank());
        Rank* rk = sw->ranks.back().get();
        if (rk->init(sw->bes.back().get(), sw->tp.get(), world, first_rank + k, max_n) || rk->build(sw->w.sc)) return nullptr;
        sw->w.ranks.push_back(rk);
    }
    return sw.release();
}
// the same with a transport of `channels` channels (an RCCL transport with fewer communicators than round slots)
void* sym_world_create_channels(const sym_callbacks* cb, int world, int first_rank, size_t max_n, int depth, int lag,
                                size_t payload_bytes, int channels) {
    auto sw = std::make_unique<SymWorld>();
    if (!sw->w.sc.init(world, depth + 1, lag)) return nullptr;
    auto* tp = new CallbackTransport(*cb);
    tp->n_channels = channels;
    sw->tp.reset(tp);
    sw->bes.emplace_back(new SymBackend(*cb, first_rank, payload_bytes));
    if (channels > 0 && channels < ROUNDS) sw->bes.back()->n_comm = channels;
    sw->ranks.emplace_back(new Rank());
    Rank* rk = sw->ranks.back().get();
    if (rk->init(sw->bes.back().get(), sw->tp.get(), world, first_rank, max_n) || rk->build(sw->w.sc)) return nullptr;
    sw->w.ranks.push_back(rk);
    return sw.release();
}
int sym_world_step(void* h, size_t n, uint64_t* round_out) {
    static const char dummy = 0;
    return ((SymWorld*)h)->w.step(&dummy, n, nullptr, 0, round_out);
}
int sym_world_flush(void* h) { return ((SymWorld*)h)->w.flush(); }
int sym_world_run_all(void* h) {       // issue everything, wait for nothing
    World& w = ((SymWorld*)h)->w;
    return w.n_rounds ? w.run_ticks(w.start_of(w.n_rounds - 1) + w.sc.round_ticks) : 0;
}
int sym_world_wait(void* h, int k, uint64_t R) { return ((SymWorld*)h)->w.wait_round((size_t)k, R); }
uint64_t sym_world_collectives(void* h) { return ((SymWorld*)h)->tp->collectives; }
uint64_t sym_world_tick(void* h) { return ((SymWorld*)h)->w.T; }
void sym_world_destroy(void* h) {
    SymWorld* sw = (SymWorld*)h;
    for (auto& r : sw->ranks) r->destroy();
    delete sw;
}
}
