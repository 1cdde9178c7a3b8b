// imt_sliced_sched.hpp -- the systolic schedule of the multi-GPU single-list mode (imt_sliced_*, include/imt.h), free of
// HIP: everything that is not hashing.  The reference's data structure is ONE sorted list with update_idx_leaf's
// sequential semantics (/root/reference/src/indexed_merkle_tree.rs:632-660), insertion i at leaf size + i (:715); a step
// of world x n insertions is cut into `world` consecutive SLICES in insertion order, rank g hashes slice g (its 2 + 2 *
// depth hashes per insertion, its witnesses), every rank keeps a replica, and what a slice WRITES BACK to the stored tree
// crosses ranks level by level:
//
//     round R = the world slices of step R.         unit q of a slice: q = 0 leaf hashes, q = 1 + l level l -> l + 1
//     round tick rt = 0, 1, ...:  rank g runs unit q = rt - g * lag of its slice      (units = depth + 1)
//                                 all ranks all-gather the payloads of that tick      (the collective)
//                                 payloads gathered at tick rt are applied at tick rt + lag
//     global tick T: round R is at round tick T - R * world * lag, so consecutive rounds overlap (up to ROUNDS in flight,
//     each on its own stream).
//
// Why it is right: rank g computes (R, q) at round tick q + g lag; the payload of an earlier slice (R, g' < g, q) was
// gathered at q + g' lag and applied by q + g' lag + lag <= q + g lag; round R - 1's last payload for unit q (rank world - 1)
// is applied at its round tick q + world * lag = the global tick at which (R, 0, q) runs, older rounds first.  A later
// slice's level l does not exist yet when an earlier one reads it.  Two ordering rules between rounds (different streams):
// before round R touches level q - 1 -- its own unit q, or its APPLY of other ranks' unit-q payloads (two rounds'
// write-backs to one node must land in slice order) -- round R - 1 has finished with that level: the other ranks' unit-q
// write-backs are in (round R - 1's apply of tick q + world * lag, an event on that round's apply stream), this rank's own
// is in and its next unit, which reads the level, has run (round R - 1's own unit q + 1, an event on its round stream).
// (Until late in round 4 both rules waited for round R - 1's whole TICK q + world * lag, i.e. also for this rank's own
// unit of that tick, 15 levels further up: nothing needs that, and it made the two rounds a rank has in flight march in
// lockstep -- tools/rank_emulation.py.)  Within a round a unit waits for its tick's apply; an apply needs no wait for the
// round's own units: a later slice's payload arrives through a collective that carries this rank's own, later, unit.
// Applies may run on a stream of their own per round slot (Backend::apply_stream) or on the round's.
//
// The classes here drive an abstract Backend (streams, events, buffers, the slice calls of one replica) and an abstract
// Transport (the all-gather).  The product instantiates them over HIP (imt_sliced.cpp: imt_itree_slice_*, RCCL / IPC /
// in-process transports); tests/native/sliced_sym.cpp instantiates the SAME code over a symbolic backend whose streams
// are FIFO queues drained by an adversarial scheduler (tests/test_sliced_schedule.py), so a missing event or an early
// buffer reuse fails on the CPU.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "../../include/imt.h"

namespace imt {
namespace sliced {

// The three switches below change the schedule this header compiles to.  They exist for tests (planted bugs the simulators
// must catch) and A/B experiment builds, which say so with -DIMT_TEST_BUILD; a product build into which one of them
// strays (a CXXFLAGS left over from an experiment) does not compile.  tests/test_sliced_schedule.py checks the guard.
#if !defined(IMT_TEST_BUILD) && ((defined(IMT_SCHED_MUTATION) && IMT_SCHED_MUTATION != 0) || defined(IMT_SCHED_ALL_WAITS) || \
                                 (defined(IMT_SLICED_ROUNDS_BUILD) && IMT_SLICED_ROUNDS_BUILD != 4))
#error "IMT_SCHED_MUTATION / IMT_SCHED_ALL_WAITS / IMT_SLICED_ROUNDS_BUILD are test switches: define IMT_TEST_BUILD with them, never in a product build"
#endif
#ifndef IMT_SLICED_ROUNDS_BUILD       // (experiment builds only: more steps in flight need -DIMT_NPIPE to match and break include/imt.h's layout)
#define IMT_SLICED_ROUNDS_BUILD 4
#endif
constexpr int ROUNDS = IMT_SLICED_ROUNDS_BUILD;   // rounds in flight = slices a tree keeps open (its plan sets minus one)

// Test hook (tests/test_sliced_schedule.py::test_the_simulator_catches_planted_mutations): a build with
// -DIMT_SCHED_MUTATION=k drops one ordering rule, and the adversarial simulator must notice.  0 in every product build.
#ifndef IMT_SCHED_MUTATION
#define IMT_SCHED_MUTATION 0
#endif
// (mutation 6: round R writes a level once round R - 1 has WRITTEN it, without waiting for that round to have read it)
#if IMT_SCHED_MUTATION == 6
#define IMT_SCHED_OWN_READ_SLACK (-1)
#else
#define IMT_SCHED_OWN_READ_SLACK 0
#endif
// -DIMT_SCHED_ALL_WAITS: issue every wait, also those another wait on the same stream implies (A/B builds; see
// wait_previous_round)
#ifdef IMT_SCHED_ALL_WAITS
#define IMT_SCHED_SKIP_IMPLIED 0
#else
#define IMT_SCHED_SKIP_IMPLIED 1
#endif

struct Schedule {
    int world = 0, units = 0, lag = 0;
    int period = 0;                  // global ticks between two rounds' starts
    int gathers = 0;                 // round ticks with a compute phase / a collective
    int round_ticks = 0;             // + the ticks that only apply

    // the smallest lag that keeps ROUNDS rounds in flight; >= 2 so that a gather overlaps the next unit
    static int default_lag(int world, int units) { return std::max(2, (units + (ROUNDS - 1) * world - 1) / ((ROUNDS - 1) * world)); }
    // false: not a schedule (world < 1, units < 2, lag < 1, or more than ROUNDS rounds in flight)
    bool init(int world_, int units_, int lag_) {
        if (world_ < 1 || units_ < 2) return false;
        world = world_;
        units = units_;
        lag = lag_ > 0 ? lag_ : default_lag(world_, units_);
        if (lag < 1) return false;
        period = world * lag;
        gathers = units + (world - 1) * lag;
        round_ticks = gathers + lag;
        return (round_ticks + period - 1) / period <= ROUNDS;
    }
    int unit_of(int rank, int rt) const {            // unit `rank` computes at round tick rt, or -1
        const int q = rt - rank * lag;
        return (q >= 0 && q < units) ? q : -1;
    }
    // [unit or -1 per rank] carried by the collective of round tick rt (unit 0 carries nothing); true if any
    bool payload_units(int rt, int32_t* out) const {
        bool any = false;
        for (int g = 0; g < world; g++) {
            const int q = unit_of(g, rt);
            out[g] = q >= 1 ? q : -1;
            any = any || q >= 1;
        }
        return any;
    }
    bool has_gather(int rt) const {
        if (rt < 0 || rt >= gathers) return false;
        for (int g = 0; g < world; g++)
            if (unit_of(g, rt) >= 1) return true;
        return false;
    }
};

typedef void* Stream;
typedef void* Event;
typedef void* Buffer;

// One rank's replica and its device objects.  Every call returns IMT_OK or an IMT_ERR_* code.
struct Backend {
    virtual ~Backend() {}
    virtual Stream round_stream(int slot) = 0;      // the stream round R runs on, slot = R % ROUNDS
    virtual Stream comm_stream(int slot) = 0;       // the stream its collectives are enqueued on
    virtual Stream apply_stream(int slot) = 0;      // the stream the other ranks' gathered write-backs are applied on
    virtual int new_event(Event* out) = 0;
    virtual void free_event(Event e) = 0;
    virtual int record(Event e, Stream s) = 0;
    virtual int wait(Stream s, Event e) = 0;        // HIP semantics: the latest record ISSUED before this call
    virtual int event_sync(Event e) = 0;            // host waits (a backend may give up: IMT_ERR_TIMEOUT)
    virtual int event_query(Event e) { (void)e; return 1; }   // 1 complete (or never recorded), 0 not yet, < 0 an error
    virtual int alloc(size_t bytes, Buffer* out) = 0;
    virtual void free_buffer(Buffer b) = 0;
    // src may belong to another rank of the same process (the in-process transport)
    virtual int copy(Buffer dst, size_t dst_off, Buffer src, size_t src_off, size_t bytes, Stream s) = 0;
    virtual uint64_t tree_size() = 0;
    virtual size_t payload_bytes(size_t n) = 0;     // imt_itree_slice_payload_bytes
    virtual size_t unit_bytes(uint64_t size_before, size_t n, unsigned unit) = 0;
    // slot = the round slot the step will run in (R % ROUNDS): a backend may prepare on that round's stream
    virtual int prepare(const void* vals, size_t n_before, size_t n_own, size_t n_after, const imt_insert_out* out,
                        unsigned flags, int slot, int* slice) = 0;
    virtual int unit(int slice, unsigned q, Buffer payload, Stream s) = 0;
    virtual int apply_gathered(Buffer gathered, size_t stride, int count, const uint64_t* size_before, const uint64_t* n,
                               const int32_t* units, Stream s) = 0;
    virtual int sync() = 0;                         // host waits for everything this rank has enqueued
    // unit / apply followed by "record `ev` on s".  A backend that can make the LAST kernel of the call signal the event
    // itself saves the marker packet a record costs on the hardware queue between two hash kernels (the HIP backend:
    // hipExtLaunchKernelGGL's stop event; tools/microbench/ext_launch_event.hip).
    virtual int unit_record(int slice, unsigned q, Buffer payload, Stream s, Event ev) {
        int rc = unit(slice, q, payload, s);
        return rc ? rc : record(ev, s);
    }
    virtual int apply_record(Buffer gathered, size_t stride, int count, const uint64_t* size_before, const uint64_t* n,
                             const int32_t* units, Stream s, Event ev) {
        int rc = apply_gathered(gathered, stride, count, size_before, n, units, s);
        return rc ? rc : record(ev, s);
    }
};

struct Rank;

// How the payloads of one tick meet.  all_gather is stream-ordered on `st` (the rank's comm stream, already behind the
// unit that packed the send buffer); fence makes `st` wait until the gather last issued on (slot, ring) is complete for
// everyone who reads this rank's send buffer (nothing to do for a collective with NCCL semantics).
struct Transport {
    virtual ~Transport() {}
    virtual int attach(Rank& rk) { (void)rk; return IMT_OK; }
    virtual void detach(Rank& rk) { (void)rk; }
    // a transport that exports its send buffers to other processes owns them; nullptr = the rank allocates
    virtual Buffer provide_send(Rank& rk, int slot, int ring, size_t bytes) { (void)rk; (void)slot; (void)ring; (void)bytes; return nullptr; }
    virtual int all_gather(Rank& rk, int slot, int ring, size_t bytes, Stream st) = 0;
    virtual int fence(Rank& rk, int slot, int ring, Stream st) { (void)rk; (void)slot; (void)ring; (void)st; return IMT_OK; }
    virtual int poll_error() { return IMT_OK; }    // after a host-side wait: did the transport give up on a peer?
    // How many independent channels the transport has (RCCL: its communicators; 0 = one per round slot).  Collectives of
    // round slots that share a channel are enqueued on ONE stream (Backend::comm_stream), so a channel sees its calls in
    // one order -- the order the hosts issue them in, which is the same on every rank (World: every tick is a function
    // of the call sequence, never of the rank).
    virtual int channels() const { return 0; }
    // a device-visible word that turns non-zero when a GPU-side wait of the transport has given up (nullptr: there is no
    // such wait); the replica's apply kernel skips its payloads then (Backend: imt_itree_set_slice_poison)
    virtual const uint32_t* poison_word() { return nullptr; }
    // one small all-gather outside any world (imt_transport_all_gather: the subtree layout's exchange of subtree roots)
    virtual int small_gather(const void* send, void* recv, size_t bytes, Stream st) {
        (void)send; (void)recv; (void)bytes; (void)st;
        return IMT_ERR_ARG;
    }
    virtual double take_wait_ms() { return 0; }    // host time spent WAITING for peers since the last call (not issuing)
    // imt_transport_destroy after a world that never drained: host-side cleanup only (threads, names in /dev/shm); the
    // object is then leaked instead of deleted -- its destructor would wait for the device
    virtual void abandon() {}
    uint64_t collectives = 0, bytes_moved = 0;
};

struct Round {
    int slice = -1;
    uint64_t size_before = 0;        // leaves in the tree before the round (the same on every rank)
    size_t n = 0;                    // insertions per slice
    uint64_t start = 0;              // global tick of the round's tick 0
};

// One rank of the sliced tree: the three phases of (round R, round tick rt).
struct Rank {
    Backend* be = nullptr;
    Transport* tp = nullptr;
    Schedule sc;
    int world = 0, rank = 0, ring = 0;
    size_t max_n = 0, payload_cap = 0;
    std::vector<Buffer> send, recv;             // [ROUNDS][ring]
    std::vector<char> send_owned;
    std::vector<size_t> gather_bytes;           // [ROUNDS][ring] bytes per rank of the collective in flight
    std::vector<char> issued;                   // [ROUNDS][ring] a collective has ever been issued on this buffer pair (diagnostics)
    std::vector<char> pending;                  // [ROUNDS][ring] a collective has been issued and not yet consumed
    std::vector<char> send_busy;                // [ROUNDS][ring] ... and the send buffer not yet known to be free again
    std::vector<Event> tick_ev;                 // [ROUNDS][round_ticks] round stream: the tick's unit is computed and packed
    std::vector<Event> applied_ev;              // [ROUNDS][round_ticks] apply stream: the tick's (and every earlier) apply is done
    std::vector<Event> gathered_ev;             // [ROUNDS][ring] the collective's stream: the gather into recv[i] is complete
    // [ROUNDS][ring] "the send buffer is packed" = the tick event of the tick that packed it: an ALIAS into tick_ev (not
    // owned), set when the tick is issued.  (Until round 5 a separate event, recorded right in front of the tick event on
    // the same stream -- one barrier packet more between two hash kernels, for nothing.)  Likewise "the receive buffer's
    // payloads have been applied" is the apply event of the tick that applied them (phase_send).
    std::vector<Event> packed_ev;
    Event done_ev[ROUNDS] = {};
    Round rounds[ROUNDS + 1];                   // round R at R % (ROUNDS + 1)
    uint64_t n_rounds = 0;
    std::vector<int32_t> w_units;
    std::vector<uint64_t> w_sb, w_n;

    Round& round(uint64_t R) { return rounds[R % (ROUNDS + 1)]; }
    int at(int slot, int r) const { return slot * ring + r; }

    int init(Backend* be_, Transport* tp_, int world_, int rank_, size_t max_n_) {
        be = be_;
        tp = tp_;
        world = world_;
        rank = rank_;
        max_n = max_n_;
        return (rank < 0 || rank >= world || max_n == 0) ? IMT_ERR_ARG : IMT_OK;
    }
    // buffers and events; after the schedule is known
    int build(const Schedule& s) {
        sc = s;
        ring = sc.lag + 1;
        payload_cap = be->payload_bytes(max_n);
        const int nb = ROUNDS * ring;
        send.assign(nb, nullptr);
        recv.assign(nb, nullptr);
        send_owned.assign(nb, 0);
        gather_bytes.assign(nb, payload_cap);
        pending.assign(nb, 0);
        issued.assign(nb, 0);
        send_busy.assign(nb, 0);
        packed_ev.assign(nb, nullptr);
        gathered_ev.assign(nb, nullptr);
        tick_ev.assign((size_t)ROUNDS * sc.round_ticks, nullptr);
        applied_ev.assign((size_t)ROUNDS * sc.round_ticks, nullptr);
        w_units.resize(world);
        w_sb.resize(world);
        w_n.resize(world);
        int rc = tp->attach(*this);
        if (rc) return rc;
        for (int slot = 0; slot < ROUNDS; slot++)
            for (int r = 0; r < ring; r++) {
                const int i = at(slot, r);
                send[i] = tp->provide_send(*this, slot, r, payload_cap);
                if (!send[i]) {
                    if ((rc = be->alloc(payload_cap, &send[i]))) return rc;
                    send_owned[i] = 1;
                }
                if ((rc = be->alloc(payload_cap * world, &recv[i]))) return rc;
                if ((rc = be->new_event(&gathered_ev[i]))) return rc;
            }
        for (auto& e : tick_ev)
            if ((rc = be->new_event(&e))) return rc;
        for (auto& e : applied_ev)
            if ((rc = be->new_event(&e))) return rc;
        for (auto& e : done_ev)
            if ((rc = be->new_event(&e))) return rc;
        return IMT_OK;
    }
    void destroy() {
        if (!be) return;
        if (tp) tp->detach(*this);
        for (size_t i = 0; i < send.size(); i++) {
            if (send[i] && send_owned[i]) be->free_buffer(send[i]);
            if (recv[i]) be->free_buffer(recv[i]);
            if (gathered_ev[i]) be->free_event(gathered_ev[i]);
        }
        for (auto e : tick_ev)
            if (e) be->free_event(e);
        for (auto e : applied_ev)
            if (e) be->free_event(e);
        for (auto e : done_ev)
            if (e) be->free_event(e);
        send.clear();
        recv.clear();
        tick_ev.clear();
        applied_ev.clear();
        be = nullptr;
    }

    // What round R needs of round R - 1 before it touches level u - 1 of this replica (its own unit u, or other ranks'
    // payloads of unit u): every write-back round R - 1 made to that level is here -- the other ranks' by that round's
    // apply of tick u + world * lag (the last slice's unit u is gathered at tick u + (world - 1) lag and applied `lag`
    // later), this rank's own by its unit u -- and this rank's round R - 1 has READ the level (its unit u + 1, which
    // computes level u from it).  Both streams of round R - 1 are in order, so one event of each covers everything before.
    //
    // Every wait is a barrier packet on the waiting stream's hardware queue, and a round's stream carries them BETWEEN two
    // of its hash kernels: ten packets a tick cost 150 us in which the round does not hash (rocprofv3 traces of one rank of
    // eight, docs/LAB_NOTES.md), so a wait that another one implies is not issued: when round R - 1 applies on its round
    // stream (the default), its tick event of tick `own` lies BEFORE its apply event of the later tick u + period on that
    // one stream, and waiting for the latter is enough.
    int wait_previous_round(Stream st, uint64_t R, int u) {
        const int pslot = (int)((R - 1) % ROUNDS);
        const size_t base = (size_t)pslot * sc.round_ticks;
        int rc;
        if ((rc = be->wait(st, applied_ev[base + u + sc.period]))) return rc;
        const int own = std::min(u + 1 + IMT_SCHED_OWN_READ_SLACK, sc.units - 1) + rank * sc.lag;
        // (mutation 9: ... taken for implied also when that round applies on ANOTHER stream)
        if (IMT_SCHED_SKIP_IMPLIED && (IMT_SCHED_MUTATION == 9 || be->apply_stream(pslot) == be->round_stream(pslot)) && own < u + sc.period) return IMT_OK;
        return be->wait(st, tick_ev[base + own]);
    }
    // the previous-round wait phase_apply issued on the round's own stream at (R, rt), and up to which unit it reaches:
    // phase_compute of the same tick, behind it on the same stream, needs no second one
    uint64_t memo_R = ~(uint64_t)0;
    int memo_rt = -1, memo_u = -1;

    // gathered payloads of tick rt - lag -> this replica, on the round's APPLY stream: an apply waits for a collective and
    // for the previous round's applies, not for this rank's own hashing, so the round's units are never queued behind it
    int phase_apply(uint64_t R, int rt) {
        const int src = rt - sc.lag;
        const int slot = (int)(R % ROUNDS);
        Stream ast = be->apply_stream(slot);
        Event done = applied_ev[(size_t)slot * sc.round_ticks + rt];
        if (src < 0 || !sc.has_gather(src)) return be->record(done, ast);
        const int i = at(slot, src % ring);
        const Round& rd = round(R);
        int rc;
        if (pending[i]) {           // the apply stream waits for the collective; the host does not
            if ((rc = be->wait(ast, gathered_ev[i])) || (IMT_SCHED_MUTATION != 3 && (rc = tp->fence(*this, slot, src % ring, ast)))) return rc;
            pending[i] = 0;
            if (ast == be->round_stream(slot)) send_busy[i] = 0;    // the stream that will pack into it has just waited
        }
        sc.payload_units(src, w_units.data());
        w_units[rank] = -1;         // own write-backs are already in this replica
        int top = -1;
        for (int g = 0; g < world; g++) top = std::max(top, (int)w_units[g]);
        if (top < 0) return be->record(done, ast);
        if (R >= 1 && IMT_SCHED_MUTATION != 2) {
            // a write-back of round R lands on a node after every write-back round R - 1 made to that level and after
            // that round's last look at it.  On the round's own stream the unit of this tick comes right behind and
            // needs the same of round R - 1 for ITS level (a later tick of that round: it implies this one): one wait
            // for both.
            int u = top;
            const bool same = IMT_SCHED_SKIP_IMPLIED && ast == be->round_stream(slot);
            if (same && IMT_SCHED_MUTATION != 1 && IMT_SCHED_MUTATION != 8) u = std::max(u, sc.unit_of(rank, rt));
            if ((rc = wait_previous_round(ast, R, u))) return rc;
            if (same) {
                memo_R = R;
                memo_rt = rt;
                // (mutation 8: the note claims the unit's level although only the payloads' level was waited for)
                memo_u = IMT_SCHED_MUTATION == 8 ? std::max(u, sc.unit_of(rank, rt)) : u;
            }
        }
        for (int g = 0; g < world; g++) {
            w_sb[g] = rd.size_before + (uint64_t)g * rd.n;
            w_n[g] = rd.n;
        }
        // Within the round no further wait is needed: a LATER slice's payload of unit u reaches this replica through a
        // collective that also carries this rank's own unit of that tick, u + (g - rank) lag >= u + 1 -- the unit that read
        // level u - 1 -- so the collective cannot complete before that read; an EARLIER slice's payload writes levels this
        // rank's units have not reached, and those wait for this apply (phase_compute).
        // (`done` is also "recv[i] has been read" for the collective that will refill it: phase_send)
        return be->apply_record(recv[i], gather_bytes[i], world, w_sb.data(), w_n.data(), w_units.data(), ast, done);
    }

    // this rank's unit of the tick, packed into the tick's send buffer
    int phase_compute(uint64_t R, int rt) {
        const int q = sc.unit_of(rank, rt);
        const int slot = (int)(R % ROUNDS);
        Stream st = be->round_stream(slot);
        const int i = at(slot, rt % ring);
        int rc;
        if (q >= 0) {
            // the earlier slices' write-backs of this tick (the previous slice's level q - 1 arrives exactly now); applies
            // on the round's own stream are in front of the unit anyway
            const bool same = IMT_SCHED_SKIP_IMPLIED && be->apply_stream(slot) == st;
            if (!same && IMT_SCHED_MUTATION != 4 && (rc = be->wait(st, applied_ev[(size_t)slot * sc.round_ticks + rt]))) return rc;
            if (q >= 1 && R >= 1 && IMT_SCHED_MUTATION != 1 && !(same && memo_R == R && memo_rt == rt && memo_u >= q))
                // level q - 1 of every slice of round R - 1 must be in this replica, and that round done reading it
                if ((rc = wait_previous_round(st, R, q))) return rc;
            if (send_busy[i]) {     // the unit packs into a send buffer an earlier collective may still be reading
                if ((rc = be->wait(st, gathered_ev[i])) || (IMT_SCHED_MUTATION != 3 && (rc = tp->fence(*this, slot, rt % ring, st)))) return rc;
                send_busy[i] = 0;
            }
        }
        // the tick's event: unit computed and packed (and what the tick's collective waits for)
        Event tick = tick_ev[(size_t)slot * sc.round_ticks + rt];
        if (sc.has_gather(rt)) packed_ev[i] = tick;
        if (q < 0) return be->record(tick, st);
        if ((rc = be->unit_record(round(R).slice, (unsigned)q, send[i], st, tick))) return rc;
        return q == sc.units - 1 ? be->record(done_ev[slot], st) : IMT_OK;
    }

    // the tick's collective (asynchronous, consumed `lag` ticks later)
    int phase_send(uint64_t R, int rt) {
        const int slot = (int)(R % ROUNDS);
        int rc;
        if (sc.has_gather(rt)) {
            const int r = rt % ring, i = at(slot, r);
            const Round& rd = round(R);
            // every rank contributes as many bytes as the largest payload of this tick needs (sizes only: the same
            // arithmetic on every rank)
            sc.payload_units(rt, w_units.data());
            size_t mx = 0;
            for (int g = 0; g < world; g++)
                if (w_units[g] >= 0) mx = std::max(mx, be->unit_bytes(rd.size_before + (uint64_t)g * rd.n, rd.n, (unsigned)w_units[g]));
            gather_bytes[i] = mx;
            Stream cs = be->comm_stream(slot);
            // the receive buffer is free once its previous payloads have been applied (another stream's business now):
            // those of this round's tick rt - ring, applied at tick rt - ring + lag = rt - 1; for the round's first gathers
            // those of the slot's previous round, the last of which its last tick applied (applies are in order)
            Event freed = nullptr;
            if (rt >= ring) freed = applied_ev[(size_t)slot * sc.round_ticks + rt - 1];
            else if (R >= (uint64_t)ROUNDS) freed = applied_ev[(size_t)slot * sc.round_ticks + sc.round_ticks - 1];
            // (applies on the round's own stream: both of those events lie in front of this tick's event on that stream,
            // which the collective waits for anyway)
            const bool implied = IMT_SCHED_SKIP_IMPLIED && be->apply_stream(slot) == be->round_stream(slot);
            if ((IMT_SCHED_MUTATION != 5 && freed && !implied && (rc = be->wait(cs, freed))) || (rc = be->wait(cs, packed_ev[i])) ||
                (rc = tp->all_gather(*this, slot, r, mx, cs)) || (rc = be->record(gathered_ev[i], cs)))
                return rc;
            pending[i] = 1;
            issued[i] = 1;
            send_busy[i] = 1;
            tp->collectives++;
            tp->bytes_moved += mx * (uint64_t)world;
        }
        return IMT_OK;
    }
};

// The ranks this process drives, in lockstep: one for a distributed world, all of them for an in-process one.
struct World;
std::string describe(World& w);

struct World {
    std::vector<Rank*> ranks;
    Schedule sc;
    double phase_ms[4] = {0, 0, 0, 0};      // host wall time in apply / compute / send / prepare (diagnostics)
    uint64_t T = 0;                  // next global tick to issue
    bool opened = false;             // step(): some replica has already opened the step (prepare succeeded)
    uint64_t n_rounds = 0;
    uint64_t starts[ROUNDS + 1] = {};
    // A failure while ticks were being issued leaves the replicas mid-step with part of a tick enqueued: nothing can be
    // resumed from there.  The world refuses every further step / wait; the trees stay marked as sliced (imt_sliced.cpp)
    // and are reloaded from a checkpoint by the caller.
    bool poisoned = false;
    int poison(int rc) {
        if (rc) poisoned = true;
        return rc;
    }

    uint64_t& start_of(uint64_t R) { return starts[R % (ROUNDS + 1)]; }

    int run_ticks(uint64_t upto) {
        int rc;
        for (; T < upto; T++) {
            const uint64_t first = n_rounds > (uint64_t)ROUNDS ? n_rounds - ROUNDS : 0;
            for (uint64_t R = first; R < n_rounds; R++) {        // oldest first
                if (T < start_of(R) || T - start_of(R) >= (uint64_t)sc.round_ticks) continue;
                const int rt = (int)(T - start_of(R));
                const auto t0 = std::chrono::steady_clock::now();
                for (Rank* rk : ranks)
                    if ((rc = rk->phase_apply(R, rt))) return rc;
                const auto t1 = std::chrono::steady_clock::now();
                for (Rank* rk : ranks)
                    if ((rc = rk->phase_compute(R, rt))) return rc;
                const auto t2 = std::chrono::steady_clock::now();
                for (Rank* rk : ranks)
                    if ((rc = rk->phase_send(R, rt))) return rc;
                const auto t3 = std::chrono::steady_clock::now();
                phase_ms[0] += std::chrono::duration<double, std::milli>(t1 - t0).count();
                phase_ms[1] += std::chrono::duration<double, std::milli>(t2 - t1).count();
                phase_ms[2] += std::chrono::duration<double, std::milli>(t3 - t2).count();
            }
        }
        return IMT_OK;
    }

    // Starts round R = n_rounds with vals = the WHOLE step (world x n values, identical on every rank) and advances the
    // global schedule by one round period.  outs[k] = witness buffers of local rank k (kept until its last unit).
    // A refused step (IMT_ERR_VALUE / NONCANONICAL / FULL: the same verdict on every rank) changes nothing.
    int step(const void* vals, size_t n, const imt_insert_out* outs, unsigned flags, uint64_t* round_out) {
        if (ranks.empty() || !vals) return IMT_ERR_ARG;
        if (n == 0 || n > ranks[0]->max_n) return IMT_ERR_RANGE;
        if (poisoned) return IMT_ERR_INTERNAL;
        // a transport that gave up on a peer (a GPU-side wait that timed out) has skipped payloads: no further step
        if (int rc = ranks[0]->tp->poll_error()) return poison(rc);
        const uint64_t R = n_rounds;
        const uint64_t size_before = ranks[0]->be->tree_size();
        // the plan set of round R - ROUNDS - 1 is reused by round R: its slice closed long ago (its last unit was issued
        // ROUNDS rounds back); nothing to wait for on the host
        std::vector<int> slices(ranks.size(), -1);
        opened = false;
        for (size_t k = 0; k < ranks.size(); k++) {
            Rank* rk = ranks[k];
            if (rk->be->tree_size() != size_before) return opened ? poison(IMT_ERR_INTERNAL) : IMT_ERR_INTERNAL;
            int rc = rk->be->prepare(vals, (size_t)rk->rank * n, n, (size_t)(rk->world - 1 - rk->rank) * n,
                                     outs ? &outs[k] : nullptr, flags, (int)(R % ROUNDS), &slices[k]);
            // a refusal by the first replica changes nothing; replicas that disagree about a step are broken, and so is
            // a world whose earlier replicas have already opened the step
            if (rc) return k == 0 ? rc : poison(IMT_ERR_INTERNAL);
            opened = true;
        }
        const uint64_t start = n_rounds == 0 ? T : std::max<uint64_t>(T, start_of(R - 1) + sc.period);
        start_of(R) = start;
        for (size_t k = 0; k < ranks.size(); k++) {
            Round& rd = ranks[k]->round(R);
            rd.slice = slices[k];
            rd.size_before = size_before;
            rd.n = n;
            rd.start = start;
            ranks[k]->n_rounds = R + 1;
        }
        n_rounds = R + 1;
        if (round_out) *round_out = R;
        return poison(run_ticks(start + sc.period));
    }

    // issue everything that is left of the rounds in flight and wait for it
    int flush() {
        int rc;
        if (poisoned) return IMT_ERR_INTERNAL;
        if (n_rounds && (rc = run_ticks(start_of(n_rounds - 1) + sc.round_ticks))) return poison(rc);
        for (Rank* rk : ranks)
            if ((rc = rk->be->sync())) return poison(rc);
        return ranks.empty() ? IMT_OK : poison(ranks[0]->tp->poll_error());
    }

    // host waits for local rank k's witnesses of round R (its slice's last unit)
    int wait_round(size_t k, uint64_t R) {
        if (k >= ranks.size() || R >= n_rounds) return IMT_ERR_RANGE;
        if (poisoned) return IMT_ERR_INTERNAL;
        Rank* rk = ranks[k];
        // its stream slot has been taken over by a later round, whose tick 0 was recorded on the same stream behind it
        int rc;
        if (R + ROUNDS < n_rounds) {
            rc = rk->be->event_sync(rk->tick_ev[(size_t)(R % ROUNDS) * sc.round_ticks]);
        } else {
            // Rank g's last unit is issued at round tick units - 1 + g * lag.  The schedule advances to the tick at which
            // EVERY rank's last unit of the round has been issued (its last compute tick, `gathers`), not to this rank's:
            // the global tick, and with it the start of the next round and the order in which collectives of
            // different rounds are issued, must be the same function of the call sequence on every rank (a world of
            // one process per rank calls wait with the same round everywhere, never with the same rank).
#if IMT_SCHED_MUTATION == 7      // (the rule until round 5: this rank's own last compute tick -- the global tick then depends on the rank)
            if ((rc = run_ticks(std::max<uint64_t>(T, start_of(R) + sc.units + (uint64_t)rk->rank * sc.lag)))) return poison(rc);
#else
            if ((rc = run_ticks(std::max<uint64_t>(T, start_of(R) + (uint64_t)sc.gathers)))) return poison(rc);
#endif
            rc = rk->be->event_sync(rk->done_ev[R % ROUNDS]);
        }
        return rc ? rc : rk->tp->poll_error();
    }
};

// All ranks in ONE process: the gather is a set of device-to-device copies between the replicas' buffers, ordered by
// events (the one-GPU rehearsal and test form, and the reference point for what a collective has to guarantee).
struct LocalTransport : Transport {
    std::vector<Rank*> peers;
    std::vector<Event> copied;       // [world][ROUNDS][ring]: rank g has copied everything of the gather (slot, ring)
    int ring = 0;

    int attach(Rank& rk) override {
        if ((int)peers.size() < rk.world) peers.resize(rk.world, nullptr);
        peers[rk.rank] = &rk;
        ring = rk.ring;
        if (copied.empty()) copied.assign((size_t)rk.world * ROUNDS * ring, nullptr);
        for (int i = 0; i < ROUNDS * ring; i++) {
            int rc = rk.be->new_event(&copied[(size_t)rk.rank * ROUNDS * ring + i]);
            if (rc) return rc;
        }
        return IMT_OK;
    }
    void detach(Rank& rk) override {
        if (copied.empty()) return;
        for (int i = 0; i < ROUNDS * ring; i++) {
            Event& e = copied[(size_t)rk.rank * ROUNDS * ring + i];
            if (e) rk.be->free_event(e);
            e = nullptr;
        }
        if (rk.rank < (int)peers.size()) peers[rk.rank] = nullptr;
    }
    int all_gather(Rank& rk, int slot, int r, size_t bytes, Stream st) override {
        // every rank has recorded packed_ev[slot][r] by now: the world drives the phases in lockstep
        int rc;
        const int i = rk.at(slot, r);
        for (int h = 0; h < rk.world; h++) {
            if (h == rk.rank) continue;
            Rank* p = peers[h];
            if ((rc = rk.be->wait(st, p->packed_ev[i])) ||
                (rc = rk.be->copy(rk.recv[i], (size_t)h * bytes, p->send[i], 0, bytes, st)))
                return rc;
        }
        return rk.be->record(copied[(size_t)rk.rank * ROUNDS * ring + i], st);
    }
    int fence(Rank& rk, int slot, int r, Stream st) override {      // nobody still reads my send buffer
        const int i = rk.at(slot, r);
        for (int h = 0; h < rk.world; h++) {
            if (h == rk.rank) continue;
            int rc = rk.be->wait(st, copied[(size_t)h * ROUNDS * ring + i]);
            if (rc) return rc;
        }
        return IMT_OK;
    }
};

// Where a world stands, for the log of a run that hangs (imt_sliced.cpp prints it when a host wait runs into the
// watchdog; bench.py when its own gives up): the global tick the host has issued up to, and per rank and round slot the
// round in it, how many of its ticks have been issued, and the first issued tick whose unit (tick_ev, round stream) /
// apply (applied_ev) has NOT completed on the device, the collectives issued and not yet complete (gathered_ev) with
// their channel, ring buffer and byte count.  A tick that stands still with its collective pending names the peer-side
// suspect; one that stands still behind a complete collective names a local one.
inline std::string describe(World& w) {
    char buf[256];
    std::string out;
    const Schedule& sc = w.sc;
    snprintf(buf, sizeof buf, "sliced world: %d ranks, lag %d, period %d, %d gathers / %d ticks per round; %llu rounds started, global tick %llu issued%s\n",
             sc.world, sc.lag, sc.period, sc.gathers, sc.round_ticks, (unsigned long long)w.n_rounds, (unsigned long long)w.T,
             w.poisoned ? " [POISONED]" : "");
    out += buf;
    const uint64_t first = w.n_rounds > (uint64_t)ROUNDS ? w.n_rounds - ROUNDS : 0;
    for (Rank* rk : w.ranks) {
        for (uint64_t R = first; R < w.n_rounds; R++) {
            const int slot = (int)(R % ROUNDS);
            const uint64_t st = w.start_of(R);
            const int issued = w.T <= st ? 0 : (int)std::min<uint64_t>(w.T - st, (uint64_t)sc.round_ticks);
            int unit_stuck = -1, apply_stuck = -1;
            for (int rt = 0; rt < issued && (unit_stuck < 0 || apply_stuck < 0); rt++) {
                const size_t i = (size_t)slot * sc.round_ticks + rt;
                if (unit_stuck < 0 && rk->be->event_query(rk->tick_ev[i]) == 0) unit_stuck = rt;
                if (apply_stuck < 0 && rk->be->event_query(rk->applied_ev[i]) == 0) apply_stuck = rt;
            }
            snprintf(buf, sizeof buf, "  rank %d slot %d: round %llu (start tick %llu), %d of %d ticks issued; ", rk->rank, slot,
                     (unsigned long long)R, (unsigned long long)st, issued, sc.round_ticks);
            out += buf;
            if (unit_stuck < 0 && apply_stuck < 0) {
                out += "everything issued is complete\n";
            } else {
                const int q = unit_stuck >= 0 ? sc.unit_of(rk->rank, unit_stuck) : -1;
                snprintf(buf, sizeof buf, "first incomplete: unit tick %d (unit %d), apply tick %d\n", unit_stuck, q, apply_stuck);
                out += buf;
            }
            for (int r = 0; r < rk->ring; r++) {
                const int i = rk->at(slot, r);
                // (the latest collective issued on this buffer pair: complete for everyone, or still waiting for a peer)
                if (rk->issued[i] && rk->be->event_query(rk->gathered_ev[i]) == 0) {
                    const int ch = rk->tp->channels() > 0 ? slot % rk->tp->channels() : slot;
                    snprintf(buf, sizeof buf, "    collective NOT COMPLETE on channel %d, ring buffer %d, %zu bytes per rank%s\n", ch, r, rk->gather_bytes[i],
                             rk->pending[i] ? " (its apply not yet issued)" : "");
                    out += buf;
                }
            }
        }
    }
    snprintf(buf, sizeof buf, "  host ms in apply / compute / send phases: %.1f / %.1f / %.1f; collectives issued %llu\n", w.phase_ms[0], w.phase_ms[1],
             w.phase_ms[2], w.ranks.empty() ? 0ull : (unsigned long long)w.ranks[0]->tp->collectives);
    out += buf;
    return out;
}

}  // namespace sliced
}  // namespace imt
