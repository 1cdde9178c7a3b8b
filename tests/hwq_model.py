"""Test infrastructure: the multi-GPU single-list schedule of libimt_hip.so on a CPU model of the HARDWARE QUEUES.

Why.  tests/sliced_sim.py treats every stream as a queue of its own.  The HIP runtime does not: streams are multiplexed
onto a few in-order hardware queues (four per priority level by default; tools/microbench/queue_map_probe.hip measures
the rule: a new stream goes to the queue with the fewest streams of its priority), a queue runs its packets one after the
other, and an event wait is a barrier packet that blocks the WHOLE queue -- every stream on it -- until the event has
happened.  A collective (an RCCL kernel, or the wait kernel of the GPU-polled IPC transport) holds its queue until every
peer's matching collective runs.  So whether a schedule makes progress, and how fast, depends on where its streams sit
and on what the OTHER ranks' hosts have issued -- neither of which the per-stream simulator sees.  And on other people's
streams: RCCL brackets every collective with a stream of the communicator's own (`rccl_internal`: the user's stream waits
for it, it waits for the kernel), which sits on SOME hardware queue of the normal-priority pool
(tools/microbench/rccl_streams_probe.hip) -- the reason the library keeps its round streams in another pool when there is
one process per GPU (IMT_SLICED_OPT_POOLS; here: QueueMap(comm_own_queues=True, rccl_dev=[queues >= 2 K])).

What.  Two steps.
  1. RECORD.  The product's schedule code (csrc/imt_sliced_sched.hpp, compiled by tests/native/sliced_sym.cpp) is driven
     exactly as the library drives it -- imt_sliced_step / _wait / _flush per rank -- and everything it issues is written
     down as one PROGRAM per host: "enqueue packet P on stream S", "host waits for event E" (the step's value check, a
     round's last unit), "host waits for everything".  What the schedule issues never depends on what the device does,
     so a program is a fixed list; one host per rank (a distributed world: the production form) or one host for all
     ranks (the in-process rehearsal).
  2. REPLAY the programs on devices with K hardware queues each (streams mapped by a queue map), either
       * adversarially (random choice among everything that may happen next, hosts included, some actors starved for a
         while): every run must DRAIN -- or the cycle of who waits for whom is reported -- and every unit must still see
         exactly the earlier slices' write-backs (src/indexed_merkle_tree.rs:632-660: one list, slices in insertion
         order); or
       * in TIME (a discrete-event simulation with measured kernel durations, processor sharing between the kernels at
         the heads of a device's queues, barrier-packet and launch costs, a link model for the collectives, per-rank
         speed skew): what N ranks deliver.  tools/hwq_calibrate.py fits the few free parameters to rocprofv3 traces of
         one GPU and checks the model against the one-GPU figures before it is asked about eight.

The oracle rule applies: this module is imported by tests/ and tools/ only."""
import ctypes
import heapq
import math
import random

import numpy as np

import sliced_sim
from sliced_sim import ROUNDS, Callbacks, CB_TYPES

c_u64 = ctypes.c_uint64

# packet kinds
REC, WAIT, HASH, SMALL, COLL, SLEEP, FSET, FWAIT = range(8)
KIND_NAME = {REC: "record", WAIT: "wait", HASH: "hash", SMALL: "small", COLL: "collective", SLEEP: "sleep", FSET: "flag_set", FWAIT: "flag_wait"}


class Pkt:
    """one packet of a hardware queue"""
    __slots__ = ("kind", "rank", "stream", "tok", "work", "fx", "key", "what", "left", "started", "t0", "heavy")

    def __init__(self, kind, rank, stream, tok=None, work=0.0, fx=None, key=None, what=""):
        self.kind, self.rank, self.stream, self.tok, self.work, self.fx, self.key, self.what = kind, rank, stream, tok, work, fx, key, what
        self.left = work
        self.started = False
        self.t0 = 0.0
        self.heavy = False          # a SMALL kernel that takes real capacity from the hash kernels while it runs (a preparation's sorts)

    def __repr__(self):
        return f"<{KIND_NAME[self.kind]} {self.what} rank {self.rank} stream {self.stream}>"


class Costs:
    """kernel durations ALONE on the device, in microseconds, for n = 2^16 insertions per slice (scaled linearly with n);
    tools/hwq_calibrate.py prints where each comes from (profiles/r04_alone_kernel_stats.csv, profiles/r05_*_trace.csv.gz)"""
    sweep_us = 690.0            # k_sweep over 2 n events (one level, or the leaf hashes)
    emit_us = 26.0              # k_emit_roots over 2 n events
    pack_base_us, pack_us_per_mb = 3.5, 4.0        # k_pack_writeback: launch + pairs written (36 B each)
    apply_base_us, apply_us_per_mb = 3.5, 0.3      # k_apply_gathered per launch + per MB of pairs applied (33 MB in 11 us at N = 8)
    fill_us = 2.3               # the 4-byte counter reset (a fill kernel)
    copy_base_us, copy_us_per_mb = 2.4, 1.0        # device-to-device copies (blit kernel): launch + bytes
    prep_us = (1500.0, 70.0)    # a step's preparation: base + per further rank's slice (index work for N x n values)
    prep_kernels = 60           # ... issued as this many launches by the host
    bar_us = 2.5                # a barrier packet (event record / stream wait) once it is at the head of its queue (3.6 measured alone; the fit)
    gap_us = 0.0                # between two kernels of one queue (the trace shows them back to back)
    issue_kernel_us, issue_event_us = 3.2, 1.6     # host time per launch / per event call
    # The device: `resident_max` hash kernels fit at a time (k_sweep: 2 048 waves of 64 lanes = 2 per SIMD, and the traces
    # show the regime change at two: with two sweeps running a 4-us kernel takes 80 - 300 us, i.e. it waits for wave slots
    # until a sweep retires; a third sweep "runs" for 1.6 - 1.9 ms of which it waits about half).  Resident hash kernels
    # share the device (rate rho[r] each, r of them resident; rho_busy while a kernel of a step's preparation -- its sorts
    # and merges -- runs as well); a SMALL kernel runs at sigma[r].  Whatever reaches the
    # head of its queue while the device is full waits, first come first served, until a resident kernel has retired --
    # a SMALL kernel only until one has reached the last `tail` of its work (a retiring kernel's waves end one by one).
    resident_max = 2
    rho = (1.0, 0.96, 0.56)
    rho_busy = (1.0, 0.62, 0.49)
    sigma = (1.0, 0.55, 0.25)
    tail = 0.15
    # more than four hardware queues in use on a device (GPU_MAX_HW_QUEUES=8: the collectives' streams on queues of their
    # own): the hash kernels run this much slower -- measured, not explained (one emulated rank, eight queues against four:
    # profiles/r05_emu_own_queues.txt; round 4 saw the same with the streams merely spread over eight queues)
    many_queues = 0.97
    # the link model of a collective between different GPUs: latency + bytes per peer / rate (one xGMI link per peer)
    link_latency_us, link_gbps = 40.0, 48.0

    def scaled(self, n):
        return n / 65536.0


def ceil_log2(x):
    return 0 if x <= 1 else (x - 1).bit_length()


def unit_bytes(size_before, n, unit, depth):
    """csrc/imt_itree.cpp: slice_unit_bytes"""
    if unit == 0:
        return 128
    l = unit - 1
    l0 = min(ceil_log2(size_before + n), depth)
    if l >= l0:
        return 128
    pairs = min(2 * n, ((size_before + n - 1) >> l) + 1)
    return (128 + 36 * pairs + 15) & ~15


class Recorder:
    """the callbacks of ONE host: whatever the schedule code issues for this host's ranks is appended to self.prog"""

    WORDS = 4

    def __init__(self, shared, lib, world, depth, batch, lag, first_rank, n_local, transport, costs, channels=0, real_sizes=False):
        self.sh, self.lib, self.world, self.depth, self.batch = shared, lib, world, depth, batch
        self.units = depth + 1
        self.transport, self.costs, self.real_sizes = transport, costs, real_sizes
        self.prog = []                      # ("op", Pkt) | ("wait", token) | ("sync", rank) | ("host", microseconds)
        self.errors = []
        self.ranks = list(range(first_rank, first_rank + n_local))
        for r in self.ranks:
            shared.new_rank(r, depth, batch, world)
        self._cb = Callbacks(**{name: typ(self._guard(getattr(self, "_" + name), name)) for name, typ in CB_TYPES})
        if n_local == world or channels == 0:
            self.h = lib.sym_world_create(ctypes.byref(self._cb), world, first_rank, n_local, batch, depth, lag or 0, self.WORDS * 8)
        else:
            self.h = lib.sym_world_create_channels(ctypes.byref(self._cb), world, first_rank, batch, depth, lag or 0, self.WORDS * 8, channels)
        assert self.h, "sym_world_create failed"
        self.channels = channels
        self.sched = sliced_sim.Schedule(lib, world, depth + 1, lag)
        self.worker = []                    # "ipc-host": the program of this rank's worker thread (a second host)
        self.seq = {}                       # (slot, ring) -> collectives issued (IPC counters)
        self.chan_seq = {}                  # channel -> collectives issued

    def _guard(self, fn, name):
        def wrapped(*a):
            try:
                r = fn(*a)
                return 0 if r is None else r
            except BaseException as e:
                self.errors.append(f"{name}{a}: {type(e).__name__}: {e}")
                return -12
        return wrapped

    def check(self, rc):
        assert not self.errors, self.errors[0]
        assert rc == 0, f"schedule call failed: {rc}"

    # ---- the script of a host
    def step(self):
        R = c_u64()
        self.check(self.lib.sym_world_step(self.h, self.batch, ctypes.byref(R)))
        return int(R.value)

    def wait(self, k, R):
        self.check(self.lib.sym_world_wait(self.h, k, R))

    def flush(self):
        self.check(self.lib.sym_world_flush(self.h))
        self.prog.append(("mark", "flushed"))

    def tick(self):
        return int(self.lib.sym_world_tick(self.h))

    def close(self):
        if self.h:
            self.lib.sym_world_destroy(self.h)
            self.h = None

    # ---- helpers
    def op(self, pkt, host_us):
        self.prog.append(("op", pkt, host_us))

    def kernel(self, kind, rank, stream, work, fx=None, what=""):
        self.op(Pkt(kind, rank, stream, work=work, fx=fx, what=what), self.costs.issue_kernel_us)

    # ---- callbacks
    def _record(self, rank, ev, stream):
        tok = self.sh.new_token()
        self.sh.ranks[rank].events[ev] = tok
        self.op(Pkt(REC, rank, stream, tok=tok, work=self.costs.bar_us, what=f"event {ev}"), self.costs.issue_event_us)

    def _wait(self, rank, stream, ev):
        owner, ev = divmod(ev, 100000)
        tok = self.sh.ranks[owner].events.get(ev)
        if tok is not None:                    # HIP: a wait on an event never recorded is a no-op
            self.op(Pkt(WAIT, rank, stream, tok=tok, work=self.costs.bar_us, what=f"event {ev} of rank {owner}"), self.costs.issue_event_us)

    def _event_sync(self, rank, ev):
        tok = self.sh.ranks[rank].events.get(ev)
        if tok is not None:
            self.prog.append(("wait", tok))

    def _alloc(self, rank, buf, nbytes):
        self.sh.ranks[rank].buffers[buf] = np.zeros(max(nbytes // 8, 4), dtype=np.int64)

    def _copy(self, rank, dst, doff, src_rank, src, soff, nbytes, stream):
        d, s = self.sh.ranks[rank].buffers[dst], self.sh.ranks[src_rank].buffers[src]
        c = self.costs

        def fx():
            w = min(nbytes // 8, self.WORDS)
            d[doff // 8:doff // 8 + w] = s[soff // 8:soff // 8 + w]
        rp = self.sh.ranks[rank]
        if rp.collecting:                      # (the in-process transport: no all_gather callback of its own)
            rp.last_gather_bytes, rp.collecting = rp.last_unit_bytes, False
        real = rp.last_gather_bytes if self.real_sizes else nbytes
        self.kernel(SMALL, rank, stream, c.copy_base_us + real / 1e6 * c.copy_us_per_mb, fx, f"copy from rank {src_rank}")

    def _tree_size(self, rank):
        return self.sh.ranks[rank].size

    def _unit_bytes(self, rank, size_before, n, q):
        rp = self.sh.ranks[rank]
        real = unit_bytes(size_before, n, q, self.depth)
        rp.last_unit_bytes = max(rp.last_unit_bytes, real) if rp.collecting else real
        rp.collecting = True
        # the symbolic payload is 32 bytes whatever the real one would be (smaller above, like the product's)
        return self.WORDS * 8 if q <= (self.depth + 1) // 2 else 24

    def _prepare(self, rank, n_before, n_own, n_after, slot, slice_out):
        rp = self.sh.ranks[rank]
        c = self.costs
        assert n_before == rank * n_own and n_own == self.batch and n_after == (self.world - 1 - rank) * n_own
        k = (rp.size - 1 + n_before) // self.batch
        sid = rp.next_id % (ROUNDS + 1)
        rp.next_id += 1
        if sid in rp.slices:
            assert rp.slices[sid]["next"] == rp.units, "plan set still open"
            self.prog.append(("wait", rp.slices[sid]["done"]))     # plan-set back-pressure (imt_itree_slice_prepare: P.done)
        rp.slices[sid] = dict(k=k, next=0, size_before=rp.size + n_before, done=None)
        rp.size += n_before + n_own + n_after
        slice_out[0] = sid
        # the step's preparation on the new round slot's collective stream (the product's default), then the host waits for
        # its verdict
        sc = self.costs.scaled(n_own)
        stream = 4 * ROUNDS if self.sh.prep_own_stream else slot if self.sh.prep_on_round else self.sh.comm_stream(slot)
        tok = self.sh.new_token()
        work = (c.prep_us[0] + c.prep_us[1] * (self.world - 1)) * sc
        # some sixty dependent small kernels: each has to get onto the device on its own
        for j in range(c.prep_kernels):
            pk = Pkt(SMALL, rank, stream, work=work / c.prep_kernels, what="preparation" if j == 0 else "preparation (cont.)")
            pk.heavy = True
            self.op(pk, c.issue_kernel_us)
        self.op(Pkt(REC, rank, stream, tok=tok, work=c.bar_us, what="prepared"), c.issue_event_us)
        rp.slices[sid]["prep"] = tok
        self.prog.append(("wait", tok))

    def _unit(self, rank, slice_id, q, payload, stream):
        rp = self.sh.ranks[rank]
        sl = rp.slices[slice_id]
        assert sl["next"] == q, "units out of order"
        sl["next"] = q + 1
        k = sl["k"]
        w = rp.buffers[payload]
        c = self.costs
        sc = c.scaled(self.batch)
        levels = rp.levels

        def fx():
            if q >= 2:
                rd = levels[q - 2]
                assert len(rd) == k + 1 and rd[-1] == k, f"rank {rank}: slice {k} unit {q} reads level {q - 2} holding {rd[-6:]} (wants 0..{k})"
            if q >= 1:
                lvl = levels[q - 1]
                assert len(lvl) == k and (k == 0 or lvl[-1] == k - 1), f"rank {rank}: slice {k} level {q - 1} sees {lvl[-6:]} (wants 0..{k - 1})"
                lvl.append(k)
            w[0], w[1], w[2] = k, q, k * 1000003 + q
            rp.computed.append((k, q))
        if q == 0:
            self.op(Pkt(WAIT, rank, stream, tok=sl["prep"], work=c.bar_us, what="prepared"), c.issue_event_us)
        self.kernel(HASH, rank, stream, c.sweep_us * sc, fx, f"slice {k} unit {q}")
        if q >= 1:
            l = q - 1
            l0 = min(ceil_log2(sl["size_before"] + self.batch), self.depth)
            if l < l0:
                pairs = min(2 * self.batch, ((sl["size_before"] + self.batch - 1) >> l) + 1)
                self.kernel(SMALL, rank, stream, c.fill_us, None, "counter reset")
                self.kernel(SMALL, rank, stream, c.pack_base_us + pairs * 36e-6 * c.pack_us_per_mb, None, "pack write-back")
            else:
                for _ in range(2 if l == l0 else 1):
                    self.kernel(SMALL, rank, stream, c.copy_base_us, None, "node copy")
        if q == self.depth:
            self.kernel(SMALL, rank, stream, c.emit_us * sc, None, "emit roots")
            tok = self.sh.new_token()
            sl["done"] = tok
            self.op(Pkt(REC, rank, stream, tok=tok, work=c.bar_us, what="slice done"), c.issue_event_us)

    def _apply_gathered(self, rank, gathered, stride, count, size_before, n, units, stream):
        rp = self.sh.ranks[rank]
        exp = [((size_before[r] - 1) // self.batch) for r in range(count)]
        qs = [units[r] for r in range(count)]
        sbs = [int(size_before[r]) for r in range(count)]
        w = rp.buffers[gathered]
        levels = rp.levels
        c = self.costs

        def fx():
            for r, q in enumerate(qs):
                if q < 0:
                    continue
                k, uq, chk = (int(x) for x in w[r * stride // 8: r * stride // 8 + 3])
                assert (k, uq) == (exp[r], q) and chk == k * 1000003 + q, \
                    f"rank {rank}: payload slot {r} holds slice {k} unit {uq}, expected slice {exp[r]} unit {q}"
                lvl = levels[q - 1]
                assert not lvl or lvl[-1] < k, f"rank {rank}: level {q - 1} gets slice {k} after {lvl[-1]}"
                lvl.append(k)
        mb = sum(max(unit_bytes(sbs[r], self.batch, q, self.depth) - 128, 0) for r, q in enumerate(qs) if q >= 1) / 1e6
        self.kernel(SMALL, rank, stream, c.apply_base_us + mb * c.apply_us_per_mb, fx, "apply gathered")

    def _sync(self, rank):
        self.prog.append(("sync", rank))

    def _all_gather(self, rank, slot, ring, send, recv, nbytes, stream):
        rp = self.sh.ranks[rank]
        c = self.costs
        real = rp.last_unit_bytes
        rp.collecting = False
        rp.last_gather_bytes = real
        sbuf, rbuf = rp.buffers[send], rp.buffers[recv]
        words = nbytes // 8
        if self.transport == "rccl":
            ch = slot % self.channels if self.channels else slot
            n = self.chan_seq.get(ch, 0)
            self.chan_seq[ch] = n + 1
            dev = 5 * ROUNDS + ch
            if self.sh.rccl_internal:              # ncclLaunchPrepare: the user's stream waits for deviceStream
                t0 = self.sh.new_token()
                self.op(Pkt(REC, rank, dev, tok=t0, work=c.bar_us, what=f"RCCL deviceStream {ch}: record"), c.issue_event_us)
                self.op(Pkt(WAIT, rank, stream, tok=t0, work=c.bar_us, what=f"RCCL: wait for deviceStream {ch}"), c.issue_event_us)
            self.op(Pkt(COLL, rank, stream, key=(ch, n), work=(real, words, sbuf, rbuf), what=f"all-gather {n} on channel {ch} ({nbytes} B)"),
                    c.issue_kernel_us)
            if self.sh.rccl_internal:              # ncclLaunchFinish: deviceStream waits for the kernel
                t1 = self.sh.new_token()
                self.op(Pkt(REC, rank, stream, tok=t1, work=c.bar_us, what=f"RCCL: all-gather {n} of channel {ch} launched"), c.issue_event_us)
                self.op(Pkt(WAIT, rank, dev, tok=t1, work=c.bar_us, what=f"RCCL deviceStream {ch}: wait for all-gather {n}"), c.issue_event_us)
        elif self.transport == "ipc":          # GPU-polled: flags in shared host memory, peer reads
            n = self.seq.get((slot, ring), 0) + 1
            self.seq[(slot, ring)] = n
            self.op(Pkt(FSET, rank, stream, key=("packed", rank, slot, ring, n), work=c.fill_us, what="packed"), c.issue_kernel_us)
            self.op(Pkt(FWAIT, rank, stream, key=[("packed", h, slot, ring, n) for h in range(self.world) if h != rank], work=c.fill_us,
                        what=f"wait packed {n} of ({slot},{ring})"), c.issue_kernel_us)
            def fx():                              # k_copy16_multi: every peer's payload in one launch, the links side by side
                for h in range(self.world):
                    if h != rank:
                        sb = self.sh.ranks[h].send_of[(slot, ring)]
                        rbuf[h * words:h * words + min(words, self.WORDS)] = sb[:min(words, self.WORDS)]
            self.kernel(SMALL, rank, stream, c.copy_base_us + (self.world - 1) * real / 1e6 * c.copy_us_per_mb + real / (c.link_gbps * 1e3), fx,
                        "peer reads")
            self.op(Pkt(FSET, rank, stream, key=("copied", rank, slot, ring, n), work=c.fill_us, what="copied"), c.issue_kernel_us)
            rp.send_of[(slot, ring)] = sbuf
        elif self.transport == "ipc-host":     # host-polled: ranks that share a GPU; nothing on the device waits for a peer
            n = self.seq.get((slot, ring), 0) + 1
            self.seq[(slot, ring)] = n
            self.op(Pkt(FSET, rank, stream, key=("packed", rank, slot, ring, n), work=c.fill_us, what="packed"), c.issue_kernel_us)
            ready = self.sh.new_token()         # the receive buffer is free once the stream gets here
            self.op(Pkt(REC, rank, stream, tok=ready, work=c.bar_us, what="receive buffer free"), c.issue_event_us)
            ws = 3 * ROUNDS + slot              # the worker's stream of this round slot
            w = self.worker
            self.prog.append(("setflag", ("job", rank, slot, ring, n)))      # the main thread hands the job to the worker
            w.append(("waitflags", [("job", rank, slot, ring, n)]))
            w.append(("waitflags", [("packed", h, slot, ring, n) for h in range(self.world) if h != rank]))
            w.append(("op", Pkt(WAIT, rank, ws, tok=ready, work=c.bar_us, what="receive buffer free"), c.issue_event_us))
            for d in range(1, self.world):
                h = (rank + d) % self.world

                def fx(h=h):
                    sb = self.sh.ranks[h].send_of[(slot, ring)]
                    rbuf[h * words:h * words + min(words, self.WORDS)] = sb[:min(words, self.WORDS)]
                w.append(("op", Pkt(SMALL, rank, ws, work=c.copy_base_us + real / 1e6 * c.copy_us_per_mb, fx=fx, what=f"peer read from rank {h}"), c.issue_kernel_us))
            w.append(("op", Pkt(FSET, rank, ws, key=("copied", rank, slot, ring, n), work=c.fill_us, what="copied"), c.issue_kernel_us))
            done = self.sh.new_token()
            w.append(("op", Pkt(REC, rank, ws, tok=done, work=c.bar_us, what="gather done"), c.issue_event_us))
            w.append(("setflag", ("issued", rank, slot, ring, n)))
            rp.send_of[(slot, ring)] = sbuf
            rp.done_of[(slot, ring)] = done
        else:                                   # "emu": tools/rank_emulation.py -- a modelled wait, then own payload into every slot
            if c.link_gbps > 0:
                self.op(Pkt(SLEEP, rank, stream, work=c.link_latency_us + real / (c.link_gbps * 1e3), what="modelled collective"), c.issue_kernel_us + 12.0)
            for h in range(self.world):
                self.kernel(SMALL, rank, stream, c.copy_base_us + real / 1e6 * c.copy_us_per_mb, None, "slot fill")

    def _fence(self, rank, slot, ring, stream):
        if self.transport == "ipc-host":        # the HOST waits: the worker has enqueued my gather's copies, every peer has copied my payload
            n = self.seq.get((slot, ring), 0)
            if n:
                self.prog.append(("waitflags", [("issued", rank, slot, ring, n)]))
                self.op(Pkt(WAIT, rank, stream, tok=self.sh.ranks[rank].done_of[(slot, ring)], work=self.costs.bar_us, what="gather done"),
                        self.costs.issue_event_us)
                self.prog.append(("waitflags", [("copied", h, slot, ring, n) for h in range(self.world) if h != rank]))
            return 0
        if self.transport != "ipc":
            return 0
        n = self.seq.get((slot, ring), 0)
        if n:
            self.op(Pkt(FWAIT, rank, stream, key=[("copied", h, slot, ring, n) for h in range(self.world) if h != rank], work=self.costs.fill_us,
                        what=f"wait copied {n} of ({slot},{ring})"), self.costs.issue_kernel_us)
        return 0


class RankState:
    def __init__(self, depth, batch, world):
        self.units = depth + 1
        self.events, self.buffers, self.slices = {}, {}, {}
        self.send_of, self.done_of = {}, {}
        self.levels = [[] for _ in range(depth)]
        self.computed = []
        self.size, self.next_id = 1, 0
        self.last_unit_bytes = self.last_gather_bytes = 128
        self.collecting = False


class Shared:
    """what the hosts of one world share: the ranks' symbolic replicas, the token counter, the stream layout"""

    def __init__(self, comm_streams=ROUNDS, apply_streams=False, prep_own_stream=False, rccl_internal=False, prep_on_round=False):
        self.ranks = {}
        self.tokens = 0
        self.comm_streams, self.apply_streams = comm_streams, apply_streams
        self.prep_own_stream = prep_own_stream          # the preparation on a stream (and hardware queue) of its own
        self.prep_on_round = prep_on_round              # ... on the new round's own stream (IMT_SLICED_OPT_PREP_STREAM 1: what the priority pools use)
        # RCCL's own stream per communicator (`deviceStream`; ncclCommInitRank creates three streams per communicator in the
        # normal-priority pool: tools/microbench/rccl_streams_probe.hip).  NCCL's eager launch brackets every collective
        # with it: the user's stream waits for an event recorded on deviceStream, and deviceStream then waits for an event
        # recorded on the user's stream behind the kernel -- a barrier packet on whatever hardware queue deviceStream shares
        self.rccl_internal = rccl_internal

    def new_rank(self, r, depth, batch, world):
        self.ranks[r] = RankState(depth, batch, world)

    def new_token(self):
        self.tokens += 1
        return self.tokens

    def comm_stream(self, slot):
        return ROUNDS + slot % self.comm_streams if self.comm_streams else slot


def record(lib, world, depth, batch, script, lag=None, hosts="per-rank", transport="rccl", comm_streams=ROUNDS, apply_streams=False,
           channels=0, costs=None, real_sizes=True, rank_scripts=None, only_ranks=None, prep_own_stream=False, rccl_internal=False,
           prep_on_round=False):
    """run `script` (a list of ("step",) / ("wait", R) / ("flush",)) on every host and return (programs, shared).
    hosts = "per-rank": one host per rank, collectives through `transport` ("rccl", "ipc", "emu"); "one": all ranks in one
    process, the product's in-process transport (copies ordered by events).  rank_scripts: per-rank scripts instead
    (unequal call sequences: what the contract forbids).  only_ranks: record these ranks only ("emu")."""
    costs = costs or Costs()
    lib.sym_set_layout(comm_streams, 1 if apply_streams else 0)
    sh = Shared(comm_streams, apply_streams, prep_own_stream, rccl_internal, prep_on_round)
    try:
        if hosts == "one":
            recs = [Recorder(sh, lib, world, depth, batch, lag, 0, world, "local", costs, real_sizes=real_sizes)]
        else:
            recs = [Recorder(sh, lib, world, depth, batch, lag, g, 1, transport, costs, channels, real_sizes=real_sizes)
                    for g in (only_ranks if only_ranks is not None else range(world))]
        for i, rec in enumerate(recs):
            for call in (rank_scripts[rec.ranks[0]] if rank_scripts else script):
                if call[0] == "step":
                    rec.step()
                elif call[0] == "wait":
                    rec.wait(0 if hosts != "one" else call[1] % world, call[-1])
                elif call[0] == "flush":
                    rec.flush()
        progs = [rec.prog for rec in recs] + [rec.worker for rec in recs if rec.worker]
    finally:
        lib.sym_set_layout(ROUNDS, 1)
    for rec in recs:
        rec.close()
    return progs, sh, recs


class QueueMap:
    """stream -> (device, hardware queue).  K queues per device.  The product's placement (imt_sliced_create measures and
    repairs it): round stream i on queue i, slot i's collective and apply streams on the same queue.  `rot[rank]` rotates a
    rank's whole map (which physical queue is "0" is the runtime's business), `comm_shift` / `apply_shift` move the helper
    streams onto ANOTHER round's queue (what an unverified creation order may give), `one_device` puts every rank on
    device 0 (the in-process rehearsal: all replicas of a process share the process's K queues)."""

    def __init__(self, K=4, rot=None, comm_shift=0, apply_shift=0, one_device=False, comm_own_queues=False, shared_gpu=False, rccl_dev=None):
        self.K, self.rot, self.comm_shift, self.apply_shift, self.one_device = K, rot or {}, comm_shift, apply_shift, one_device
        self.comm_own_queues = comm_own_queues          # the collectives' streams in another priority pool: K more queues
        self.shared_gpu = shared_gpu                    # one PROCESS per rank, all on device 0 (the rehearsal): queues per process
        # hardware queue of RCCL's deviceStream per channel: a list (the same on every rank) or {rank: list}; numbers >= 2 K
        # are queues that nothing of the world is on
        self.rccl_dev = rccl_dev

    def __call__(self, rank, stream):
        kind, slot = divmod(stream, ROUNDS)           # 0 round, 1 collective, 2 apply, 3 the host-polled transport's worker
        if kind == 5:                                 # RCCL's deviceStream of channel `slot`
            lst = self.rccl_dev[rank] if isinstance(self.rccl_dev, dict) else self.rccl_dev
            return (0 if (self.one_device or self.shared_gpu) else rank), lst[slot]
        if kind == 4:                                 # the preparation's own stream: a queue nothing else of the world is on
            return (0 if (self.one_device or self.shared_gpu) else rank), 3 * self.K + (rank if (self.one_device or self.shared_gpu) else 0)
        shift = (0, self.comm_shift, self.apply_shift, self.comm_shift)[kind]
        q = (slot + shift + self.rot.get(rank, 0)) % self.K
        if kind in (1, 3) and self.comm_own_queues:
            q += self.K
        if self.shared_gpu:
            return 0, q + 2 * self.K * rank
        return (0 if self.one_device else rank), q


# ------------------------------------------------------------------------------------------------ adversarial replay
class Deadlock(AssertionError):
    pass


def replay_adversarial(progs, sh, qmap, seed=0, world=None, max_steps=50_000_000):
    """every enabled action (a host's next entry, a queue's head packet) in random order, some actors starved for random
    stretches; returns the number of packets executed.  Raises Deadlock with the cycle when nothing can move."""
    rng = random.Random(seed)
    queues = {}                                 # (dev, q) -> list (FIFO, index = head)
    heads = {}
    done = set()                                # tokens
    flags = set()
    pcs = [0] * len(progs)
    pending = {}                                # rank -> packets enqueued and not complete
    coll_at_head = {}                           # key -> {rank: (qid)}
    executed = 0
    world = world or len(sh.ranks)

    def qid_of(p):
        return qmap(p.rank, p.stream)

    def host_enabled(h):
        if pcs[h] >= len(progs[h]):
            return False
        e = progs[h][pcs[h]]
        if e[0] == "wait":
            return e[1] in done
        if e[0] == "sync":
            return pending.get(e[1], 0) == 0
        if e[0] == "waitflags":
            return all(k in flags for k in e[1])
        return True                     # "op", "mark", "setflag"

    def head(qid):
        lst = queues[qid]
        i = heads[qid]
        return lst[i] if i < len(lst) else None

    def pkt_enabled(p, qid):
        if p.kind == WAIT:
            return p.tok in done
        if p.kind == FWAIT:
            return all(k in flags for k in p.key)
        if p.kind == COLL:
            # every rank's matching collective is at the head of its queue
            arrived = coll_at_head.setdefault(p.key, {})
            arrived[p.rank] = qid
            return len(arrived) == world
        return True

    def run_pkt(p, qid):
        nonlocal executed
        if p.kind == REC:
            done.add(p.tok)
        elif p.kind == FSET:
            flags.add(p.key)
        elif p.fx is not None:
            p.fx()
        heads[qid] += 1
        pending[p.rank] -= 1
        executed += 1

    def run_coll(key):
        nonlocal executed
        arrived = coll_at_head.pop(key)
        pk = {r: head(q) for r, q in arrived.items()}
        sizes = {p.work[1] for p in pk.values()}
        assert len(sizes) == 1, f"collective {key}: the ranks contribute different sizes {sorted(sizes)} -- not the same call on every rank"
        words = sizes.pop()
        w = min(words, Recorder.WORDS)
        for r, p in pk.items():
            for r2, p2 in pk.items():
                p.work[3][r2 * words:r2 * words + w] = p2.work[2][:w]
        for r, q in arrived.items():
            heads[q] += 1
            pending[r] -= 1
            executed += 1

    starved, budget = set(), 0
    steps = 0
    while True:
        steps += 1
        assert steps < max_steps, "replay does not end"
        acts = [("h", h) for h in range(len(progs)) if host_enabled(h)]
        colls_ready = set()
        for qid in queues:
            p = head(qid)
            if p is None:
                continue
            if pkt_enabled(p, qid):
                if p.kind == COLL:
                    colls_ready.add(p.key)
                else:
                    acts.append(("q", qid))
        acts += [("c", k) for k in colls_ready]
        if not acts:
            if all(pcs[h] >= len(progs[h]) for h in range(len(progs))) and all(head(q) is None for q in queues):
                return executed
            raise Deadlock(describe_deadlock(progs, pcs, queues, heads, done, flags, coll_at_head, world))
        if budget <= 0:
            names = [("h", h) for h in range(len(progs))] + [("q", q) for q in queues]
            starved = set(rng.sample(names, rng.randrange(0, max(1, len(names) * 2 // 3) + 1))) if names else set()
            budget = rng.randrange(1, 400)
        budget -= 1
        pool = [a for a in acts if a not in starved] or acts
        kind, x = rng.choice(pool)
        if kind == "h":
            e = progs[x][pcs[x]]
            pcs[x] += 1
            if e[0] == "op":
                p = e[1]
                qid = qid_of(p)
                if qid not in queues:
                    queues[qid] = []
                    heads[qid] = 0
                queues[qid].append(p)
                pending[p.rank] = pending.get(p.rank, 0) + 1
            elif e[0] == "setflag":
                flags.add(e[1])
        elif kind == "q":
            run_pkt(head(x), x)
        else:
            run_coll(x)


def describe_deadlock(progs, pcs, queues, heads, done, flags, coll_at_head, world):
    lines = ["deadlock: nothing can move"]
    for h, prog in enumerate(progs):
        if pcs[h] < len(prog):
            e = prog[pcs[h]]
            lines.append(f"  host {h} at entry {pcs[h]} of {len(prog)}: {e[0]} {e[1] if e[0] != 'op' else ''}")
    for qid, lst in sorted(queues.items()):
        i = heads[qid]
        if i < len(lst):
            p = lst[i]
            why = ""
            if p.kind == WAIT:
                why = "waits for an event that has not happened"
            elif p.kind == FWAIT:
                why = f"waits for flags {[k for k in p.key if k not in flags]}"
            elif p.kind == COLL:
                why = f"holds the queue until every rank runs collective {p.key}: here so far ranks {sorted(coll_at_head.get(p.key, {}))} of {world}"
            behind = [q for q in lst[i + 1:i + 6]]
            lines.append(f"  device {qid[0]} queue {qid[1]}: head {p!r} {why}; behind it: {behind}")
    return "\n".join(lines)


# ------------------------------------------------------------------------------------------------------ timed replay
class Timed:
    """discrete-event replay.  Every device runs the head packets of its K queues concurrently; HASH kernels share the
    device (rate rho[h] each when h of them run), SMALL kernels run at sigma[h]; barrier packets take bar_us at the head
    of their queue once their event has happened; a COLL completes link_latency + bytes / link after the last rank's
    has reached the head of its queue; hosts pay issue costs and block where the library blocks."""

    def __init__(self, progs, sh, qmap, costs, world, speed=None, host_speed=None):
        self.progs, self.sh, self.qmap, self.c, self.world = progs, sh, qmap, costs, world
        self.speed = speed or {}                # device -> relative speed (1.0 = the calibrated GPU)
        self.host_speed = host_speed or {}
        self.now = 0.0
        self.queues, self.heads = {}, {}        # qid -> list / head index
        self.running = {}                       # qid -> Pkt running at the head
        self.dev_q = {}                         # dev -> [qid]
        self.dev_t = {}                         # dev -> time its rates were last applied
        self.dev_ver = {}
        self.dev_wait = {}                      # dev -> [qid] whose head kernel waits for room on the device
        self.done_at = {}                       # token -> time
        self.waiters = {}                       # token -> [("q", qid) | ("h", host)]
        self.flags, self.flag_waiters, self.host_flag_waiters = set(), {}, {}
        self.pending, self.sync_waiters = {}, {}
        self.coll = {}                          # key -> {rank: qid}
        self.heap, self.n = [], 0
        self.pcs = [0] * len(progs)
        self.host_t = [0.0] * len(progs)
        self.host_blocked = [None] * len(progs)
        self.trace = None                       # list of (dev, queue, what, start, end) when set to []
        self.hash_busy = {}                     # dev -> [time with h hash kernels resident]
        self.stats = dict(packets=0)
        self.sync_times = [[] for _ in progs]   # per host: when each "sync" (flush) returned
        self.host_waited = [0.0] * len(progs)   # per host: time blocked in waits

    # ---- event heap
    def at(self, t, what, arg, ver=0):
        self.n += 1
        heapq.heappush(self.heap, (t, self.n, what, arg, ver))

    # ---- device processor sharing
    def rates(self, dev):
        h = s = 0
        for q in self.dev_q.get(dev, ()):
            p = self.running.get(q)
            if p is not None and p.started:
                if p.kind == HASH:
                    h += 1
                elif p.kind == SMALL and p.heavy:
                    s += 1
        sp = self.speed.get(dev, 1.0)
        c = self.c
        if len(self.dev_q.get(dev, ())) > 4 and not self.qmap.shared_gpu and not self.qmap.one_device:
            sp *= c.many_queues
        rho = (c.rho_busy if s else c.rho)
        return rho[min(h, len(rho) - 1)] * sp, c.sigma[min(h, len(c.sigma) - 1)] * sp, h

    def room(self, dev, p):
        """may p start now? (the device holds resident_max hash kernels; a SMALL kernel also gets in while one retires)"""
        res = [x for q in self.dev_q.get(dev, ()) if (x := self.running.get(q)) is not None and x.started and x.kind == HASH]
        if len(res) < self.c.resident_max:
            return True
        if p.kind in (SMALL, COLL, SLEEP):      # (a collective's kernel, or the emulation's one-wave sleep, gets in like a small kernel)
            return any(x.left <= self.c.tail * x.work for x in res)
        return False

    def serve_waiting(self, dev):
        """first come first served among the packets that wait for room on the device"""
        wl = self.dev_wait.get(dev)
        if not wl:
            return
        keep = []
        for qid in wl:
            p = self.running.get(qid)
            if p is None or p.started:
                continue
            if self.room(dev, p):
                p.started = True
                p.t0 = self.now
                if p.kind == COLL:
                    self.coll_arrive(p, qid)
            else:
                keep.append(qid)
        self.dev_wait[dev] = keep

    def advance(self, dev):
        """apply the elapsed time to the device's running kernels"""
        t0 = self.dev_t.get(dev, 0.0)
        dt = self.now - t0
        if dt > 0:
            rh, rs, h = self.rates(dev)
            for q in self.dev_q.get(dev, ()):
                p = self.running.get(q)
                if p is None or not p.started:
                    continue
                if p.kind == HASH:
                    p.left -= dt * rh
                elif p.kind == SMALL:
                    p.left -= dt * rs
                elif p.kind in (REC, WAIT, SLEEP, FSET, FWAIT):
                    p.left -= dt
            hb = self.hash_busy.setdefault(dev, [0.0] * 8)
            hb[min(h, 7)] += dt
        self.dev_t[dev] = self.now

    def reschedule(self, dev):
        """after any change on the device: the next completion under the new rates"""
        self.dev_ver[dev] = self.dev_ver.get(dev, 0) + 1
        rh, rs, _ = self.rates(dev)
        best = None
        waiting_small = any((x := self.running.get(q)) is not None and not x.started and x.kind in (SMALL, COLL, SLEEP) for q in self.dev_wait.get(dev, ()))
        for q in self.dev_q.get(dev, ()):
            p = self.running.get(q)
            if p is None or not p.started:
                continue
            r = rh if p.kind == HASH else rs if p.kind == SMALL else 1.0
            if p.kind == COLL:
                continue
            t = self.now + max(p.left, 0.0) / r
            if best is None or t < best[0]:
                best = (t, q)
            if waiting_small and p.kind == HASH and p.left > self.c.tail * p.work:
                tt = self.now + (p.left - self.c.tail * p.work) / r + 1e-3      # when it starts to retire
                if tt < best[0]:
                    best = (tt, q)
        if best is not None:
            self.at(best[0], "dev", (dev, best[1]), self.dev_ver[dev])

    # ---- queues
    def enqueue(self, p):
        qid = self.qmap(p.rank, p.stream)
        if qid not in self.queues:
            self.queues[qid] = []
            self.heads[qid] = 0
            self.dev_q.setdefault(qid[0], []).append(qid)
        self.queues[qid].append(p)
        self.pending[p.rank] = self.pending.get(p.rank, 0) + 1
        if qid not in self.running and self.heads[qid] == len(self.queues[qid]) - 1:
            self.at(self.now, "head", qid)

    def try_start(self, qid):
        """the packet at the head of qid, if any, starts when what it waits for is there"""
        if qid in self.running:
            return
        lst, i = self.queues[qid], self.heads[qid]
        if i >= len(lst):
            return
        p = lst[i]
        dev = qid[0]
        self.advance(dev)
        self.running[qid] = p
        p.left = p.work if p.kind != COLL else 0.0
        if p.kind == WAIT and p.tok not in self.done_at:
            self.waiters.setdefault(p.tok, []).append(("q", qid))
            return                                  # holds the queue; not started
        if p.kind == FWAIT:
            missing = [k for k in p.key if k not in self.flags]
            if missing:
                for k in missing:
                    self.flag_waiters.setdefault(k, []).append(qid)
                return
        if p.kind in (HASH, SMALL, COLL) and not self.room(dev, p):
            self.dev_wait.setdefault(dev, []).append(qid)
            self.reschedule(dev)
            return
        p.started = True
        p.t0 = self.now
        if p.kind == COLL:
            self.coll_arrive(p, qid)
            return
        self.reschedule(dev)

    def coll_arrive(self, p, qid):
        """an RCCL kernel has got onto its device: it spins there until every rank's has; then the payloads move"""
        arrived = self.coll.setdefault(p.key, {})
        arrived[p.rank] = qid
        if len(arrived) == self.world:
            nbytes = max(self.queues[q][self.heads[q]].work[0] for q in arrived.values())
            t = self.now + self.c.link_latency_us + (nbytes / (self.c.link_gbps * 1e3) if self.c.link_gbps > 0 else 0.0)
            self.at(t, "coll", p.key)

    def complete(self, qid):
        p = self.running.pop(qid)
        dev = qid[0]
        self.heads[qid] += 1
        self.pending[p.rank] -= 1
        self.stats["packets"] += 1
        if self.trace is not None and p.kind in (HASH, SMALL, SLEEP, COLL):
            self.trace.append((qid[0], qid[1], p.stream, p.what, p.kind, p.t0 if p.kind != COLL else self.now, self.now))
        if p.kind == REC:
            self.done_at[p.tok] = self.now
            for kind, x in self.waiters.pop(p.tok, ()):
                if kind == "q":
                    self.at(self.now, "unblock", x)
                else:
                    self.at(self.now, "host", x)
        elif p.kind == FSET:
            self.set_flag(p.key)
        if self.pending[p.rank] == 0:
            for h in self.sync_waiters.pop(p.rank, ()):
                self.at(self.now, "host", h)
        if p.kind == HASH:
            self.serve_waiting(dev)
        # the next packet of the queue: a kernel behind a kernel starts after the launch gap
        self.at(self.now + self.c.gap_us, "head", qid)
        self.reschedule(dev)

    def set_flag(self, key):
        self.flags.add(key)
        for q in self.flag_waiters.pop(key, ()):
            self.at(self.now, "unblock", q)
        for h in self.host_flag_waiters.pop(key, ()):
            self.at(self.now + 20.0, "host", h)          # a polling host thread notices within its polling interval

    # ---- hosts
    def run_host(self, h):
        prog = self.progs[h]
        hs = self.host_speed.get(h, 1.0)
        while self.pcs[h] < len(prog):
            e = prog[self.pcs[h]]
            if self.host_t[h] > self.now + 1e-9:
                self.at(self.host_t[h], "host", h)
                return
            if e[0] == "op":
                self.pcs[h] += 1
                self.host_t[h] = max(self.host_t[h], self.now) + e[2] / hs
                # the packet is visible to the device when the call returns
                self.at(self.host_t[h], "enq", e[1])
                continue
            if e[0] == "wait":
                if e[1] in self.done_at:
                    self.pcs[h] += 1
                    self.host_t[h] = max(self.host_t[h], self.now) + 4.0 / hs      # the wake-up
                    continue
                self.waiters.setdefault(e[1], []).append(("h", h))
                return
            if e[0] == "mark":
                self.pcs[h] += 1
                self.sync_times[h].append(max(self.now, self.host_t[h]))
                continue
            if e[0] == "setflag":
                self.pcs[h] += 1
                self.set_flag(e[1])
                continue
            if e[0] == "waitflags":
                missing = [k for k in e[1] if k not in self.flags]
                if not missing:
                    self.pcs[h] += 1
                    continue
                self.host_flag_waiters.setdefault(missing[0], []).append(h)
                return
            if e[0] == "sync":
                if self.pending.get(e[1], 0) == 0:
                    self.pcs[h] += 1
                    continue
                self.sync_waiters.setdefault(e[1], []).append(h)
                return

    def run(self):
        for h in range(len(self.progs)):
            self.at(0.0, "host", h)
        while self.heap:
            t, _, what, arg, ver = heapq.heappop(self.heap)
            self.now = max(self.now, t)
            if what == "host":
                self.run_host(arg)
            elif what == "enq":
                self.enqueue(arg)
            elif what == "head":
                self.try_start(arg)
            elif what == "unblock":
                p = self.running.get(arg)
                if p is None or p.started:
                    continue
                ok = (p.kind == WAIT and p.tok in self.done_at) or (p.kind == FWAIT and all(k in self.flags for k in p.key))
                if ok:
                    self.advance(arg[0])
                    p.started = True
                    p.t0 = self.now
                    self.reschedule(arg[0])
            elif what == "dev":
                dev, qid = arg
                if ver != self.dev_ver.get(dev):
                    continue
                self.advance(dev)
                p = self.running.get(qid)
                if p is not None and p.started and p.left <= 1e-6:
                    self.complete(qid)
                else:
                    self.serve_waiting(dev)         # a resident kernel has begun to retire
                    self.reschedule(dev)
            elif what == "coll":
                arrived = self.coll.pop(arg)
                for r, q in arrived.items():
                    self.advance(q[0])
                    self.complete(q)
        # a sync that waited for packets still in flight when it was reached
        left = [h for h in range(len(self.progs)) if self.pcs[h] < len(self.progs[h])]
        assert not left, f"timed replay stalled: hosts {left} did not finish (a deadlock: run replay_adversarial for the cycle)"
        return self.now
