"""Multi-GPU, the reference's data structure: ONE indexed tree -- one sorted list, update_idx_leaf's sequential
semantics (src/indexed_merkle_tree.rs:632-660), insertion i at leaf `size + i` (:715) -- on `world` GPUs, bit-exact
with one GPU at any world size.

A step inserts world x batch values.  The step is cut into `world` consecutive SLICES in insertion order; rank g
hashes slice g: its 2 + 2 * depth hashes per insertion, its witnesses (old / interim / new root, both proofs) written
by its own kernels.  Every rank keeps a replica of the stored tree and of the sorted index (all ranks see all values;
the index work -- sort, low-leaf search, merge -- is hash-free).  What crosses ranks is what a slice WRITES BACK to
the stored tree, level by level: slice k's level l must see the level-l nodes of every earlier slice, and of no later
one.  That makes the ranks a systolic chain, `lag` levels apart:

    round R = the world slices of step R.         unit q of a slice: q = 0 leaf hashes, q = 1 + l level l -> l + 1
    round tick rt = 0, 1, ...:  rank g runs unit q = rt - g * lag of its slice      (units = depth + 1)
                                all ranks all-gather the payloads of that tick      (RCCL over xGMI: the collective)
                                payloads gathered at tick rt are applied at tick rt + lag
    global tick T: round R is at round tick T - R * world * lag, so consecutive rounds overlap (up to four in
    flight, each on its own stream) and a rank always has about units / (world * lag) of its slices in the air.

Why this is right (checked symbolically on CPU by tests/test_sliced_schedule.py with a backend that tracks which slices'
levels a replica has seen): rank g computes (R, q) at round tick q + g lag; the payload of an earlier slice (R, g' < g,
q) was gathered at q + g' lag and applied by q + g' lag + lag <= q + g lag; round R - 1's last payload for unit q (rank
world - 1) is applied at its round tick q + world * lag = the global tick at which (R, 0, q) runs, older rounds first.
A later slice's level l does not exist yet when an earlier one reads it.  Streams: the round's; an event per round
tick orders round R's units AND its applies behind round R - 1's writes to the same level (two rounds' write-backs to
one node must land in slice order).

Per step and rank: depth + 1 + (world - 1) lag all-gathers, issued asynchronously and consumed `lag` ticks later.  A
payload is the packed (node, value) pairs of a level's write-back: one per event at the bottom of the tree (36 B x 2^17 =
4.7 MB at batch 2^16), half as many per level once a level has fewer nodes than the slice has events, 128 bytes above
l0; every all-gather moves world x the largest payload of its tick (imt_itree_slice_unit_bytes: the same arithmetic on
every rank).

The compute backend is pluggable like sharded.py's: `SliceGpuBackend` (libimt_hip.so) or, in CPU tests, a symbolic one.
Transports: `DistTransport` (torch.distributed: device tensors with "nccl" = RCCL, pinned host staging with gloo) and
`LocalWorld` (all ranks in one process on one GPU: the single-GPU rehearsal and test form).
"""
import ctypes

import torch


class SliceSchedule:
    """Pure arithmetic of the systolic schedule (no GPU, no collectives)."""

    STREAMS = 4                      # rounds in flight (= plan sets the library keeps open per tree)

    def __init__(self, world, units, lag=None):
        if world < 1 or units < 2:
            raise ValueError("world >= 1 and units >= 2")
        self.world, self.units = world, units
        # rounds in flight = ceil(units / (world * lag)) + 1 <= STREAMS; nccl wants lag >= 2 so that a gather overlaps
        # the next unit instead of stalling it
        self.lag = lag if lag is not None else max(2, -(-units // ((self.STREAMS - 1) * world)))
        if self.lag < 1:
            raise ValueError("lag >= 1")
        self.period = world * self.lag                              # global ticks between two rounds' starts
        self.gathers = units + (world - 1) * self.lag               # round ticks with a compute phase / a collective
        self.round_ticks = self.gathers + self.lag                  # + the ticks that only apply
        if -(-self.round_ticks // self.period) > self.STREAMS:
            raise ValueError(f"lag {self.lag} keeps more than {self.STREAMS} rounds in flight at world {world}")

    def unit_of(self, rank, rt):
        """unit rank `rank` computes at round tick rt, or None"""
        q = rt - rank * self.lag
        return q if 0 <= q < self.units else None

    def payload_units(self, rt):
        """[unit or -1 per rank] carried by the collective of round tick rt (unit 0 carries nothing)"""
        out = []
        for g in range(self.world):
            q = self.unit_of(g, rt)
            out.append(q if q is not None and q >= 1 else -1)
        return out

    def has_gather(self, rt):
        return rt < self.gathers and any(q >= 0 for q in self.payload_units(rt))

    def next_start(self, starts, T):
        """global tick at which the next round starts: one period after the previous one, or now if the schedule has
        run dry in between (flush)"""
        return T if not starts else max(T, starts[-1] + self.period)

    def active_rounds(self, T, starts):
        """[(round, round tick)] with work at global tick T, oldest first; starts[R] = global tick of round R's tick 0"""
        out = []
        for R in range(max(0, len(starts) - self.STREAMS), len(starts)):
            if 0 <= T - starts[R] < self.round_ticks:
                out.append((R, T - starts[R]))
        return out


class SliceGpuBackend:
    """One rank's replica: imt context + indexed tree on its GPU, rotating witness buffers, round streams."""

    FIELDS = ("low_index", "low_leaf", "is_largest", "old_root", "interim_root", "new_root", "new_leaf", "low_sib",
              "new_sib")

    def __init__(self, imt, device_index, depth, capacity, batch, nbuf=SliceSchedule.STREAMS + 1, fmt=0):
        self.imt, self.F, self.lib = imt, imt._ffi, imt.lib
        self.depth, self.batch, self.fmt = depth, batch, fmt
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(device_index)
        self.ctx = imt.Context(device_index)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.tree = imt.IndexedTree(self.ctx, depth, capacity)
        u8 = dict(dtype=torch.uint8, device=self.device)
        self.sets = [dict(low_index=torch.empty(batch, dtype=torch.int64, device=self.device),
                          low_leaf=torch.empty((batch, 3, 32), **u8), is_largest=torch.empty(batch, **u8),
                          old_root=torch.empty((batch, 32), **u8), interim_root=torch.empty((batch, 32), **u8),
                          new_root=torch.empty((batch, 32), **u8), new_leaf=torch.empty((batch, 3, 32), **u8),
                          low_sib=torch.empty((depth, batch, 32), **u8), new_sib=torch.empty((depth, batch, 32), **u8))
                     for _ in range(nbuf)]
        self.structs = [self.F.InsertOut(**{k: t.data_ptr() for k, t in b.items()}) for b in self.sets]
        self.flags = self.F.DEVICE_PTRS | fmt
        self.payload_bytes = int(self.lib.imt_itree_slice_payload_bytes(batch))
        self.units = depth + 1
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(SliceSchedule.STREAMS)]
        self.size_before_round = []          # tree size when round R started (the same on every rank)

    def size(self):
        return int(self.lib.imt_itree_size(self.tree.h))

    def make_buffer(self, nbytes):
        return torch.zeros(nbytes, dtype=torch.uint8, device=self.device)

    def prepare(self, vals, n_before, n_own, n_after, out_slot):
        """vals: uint8 [n_before + n_own + n_after, 32] on this device.  Returns the library's slice id."""
        sl = ctypes.c_int(-1)
        rc = self.lib.imt_itree_slice_prepare(self.tree.h, ctypes.c_void_p(vals.data_ptr()), n_before, n_own, n_after,
                                              ctypes.byref(self.structs[out_slot]), self.flags, ctypes.byref(sl), None)
        if rc == self.F.ERR["VALUE"]:
            raise ValueError(self.lib.imt_last_error(self.ctx.h).decode())
        self.ctx._check(rc)
        return sl.value

    def unit_bytes(self, size_before, n, q):
        """bytes the payload of unit q of a slice (n insertions into a tree of size_before leaves) uses"""
        return int(self.lib.imt_itree_slice_unit_bytes(self.tree.h, size_before, n, q))

    def unit(self, slice_id, q, payload, stream):
        self.ctx._check(self.lib.imt_itree_slice_unit(self.tree.h, slice_id, q, ctypes.c_void_p(payload.data_ptr()),
                                                      ctypes.c_void_p(stream.cuda_stream)))

    def apply_gathered(self, gathered, stride, size_before, n, units, stream):
        cnt = len(units)
        a = (ctypes.c_uint64 * cnt)(*size_before)
        b = (ctypes.c_uint64 * cnt)(*n)
        u = (ctypes.c_int32 * cnt)(*units)
        self.ctx._check(self.lib.imt_itree_slice_apply_gathered(self.tree.h, ctypes.c_void_p(gathered.data_ptr()), stride,
                                                                cnt, a, b, u, ctypes.c_void_p(stream.cuda_stream)))

    def stream_ctx(self, stream):
        return torch.cuda.stream(stream)

    def new_event(self):
        return torch.cuda.Event()

    def outputs(self, slot, n=None):
        """the slot's tensors; a slice shorter than `batch` fills a prefix (sibling rows are packed [depth][n])"""
        d = dict(self.sets[slot])
        if n is not None and n != self.batch:
            for k, t in d.items():
                d[k] = t.view(-1)[:self.depth * n * 32].view(self.depth, n, 32) if k.endswith("_sib") else t[:n]
        return d

    def sync(self):
        for s in self.streams:
            s.synchronize()
        self.ctx.sync()
        torch.cuda.synchronize(self.device)


class DistTransport:
    """all-gather of one payload per rank through torch.distributed.  backend "nccl" (= RCCL over xGMI): device
    tensors, asynchronous, waited for `lag` ticks later on the round's stream.  Anything else (gloo: rehearsal on a
    box without one GPU per rank, CPU tests): through host memory -- for device buffers on a helper thread with its
    own gloo group and copy stream, so that the host-staged gather overlaps the hashing like the RCCL one does."""

    class _HostWork:
        def __init__(self, tp, issued, done_event, stream):
            self.tp, self.issued, self.done_event, self.stream = tp, issued, done_event, stream

        def wait(self):
            if not self.issued.wait(timeout=600) or self.tp._failure is not None:   # the helper has enqueued the copy back ...
                raise RuntimeError(f"host-staged all-gather failed: {self.tp._failure or 'timed out'}")
            self.stream.wait_event(self.done_event)  # ... and the round's stream runs behind it

    def __init__(self, dist, via_host):
        self.dist, self.via_host = dist, via_host
        self.bytes_moved, self.collectives = 0, 0
        self._jobs = self._thread = self._group = self._failure = None
        self._staging = {}

    def _helper(self, device):
        torch.cuda.set_device(device)
        copy_stream = torch.cuda.Stream(device=device)
        while True:
            job = self._jobs.get()
            if job is None:
                return
            packed, inp, out, h, o, sizes, me, issued, done = job
            try:
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(packed)
                    h.copy_(inp, non_blocking=True)
                    copy_stream.synchronize()
                    # through the host every byte costs (gloo on the loopback: 1-2 GB/s): each rank broadcasts exactly
                    # what its payload uses instead of all of them padding to the largest of the tick
                    S = h.numel()
                    for r, nbytes in enumerate(sizes):
                        if nbytes:
                            self.dist.broadcast(h[:nbytes] if r == me else o[r * S:r * S + nbytes], src=self._ranks[r],
                                                group=self._group)
                    out.copy_(o, non_blocking=True)
                    done.record(copy_stream)
            except Exception as e:                   # reported by the next wait() on the main thread
                self._failure = f"{type(e).__name__}: {e}"
            issued.set()

    def all_gather(self, rk, slot, ring, stream):
        S = rk.gather_bytes[slot][ring]
        out, inp = rk.recv[slot][ring][:S * rk.world], rk.send[slot][ring][:S]
        self.collectives += 1
        if not self.via_host:
            self.bytes_moved += out.numel()
            return self.dist.all_gather_into_tensor(out, inp, async_op=True)
        self.bytes_moved += sum(rk.gather_sizes[slot][ring])
        if not inp.is_cuda:                          # CPU tests: nothing to overlap
            o = torch.empty(out.numel(), dtype=torch.uint8)
            self.dist.all_gather_into_tensor(o, inp)
            out.copy_(o)
            return None
        import queue
        import threading
        if self._thread is None:                     # every rank reaches this at its first gather: new_group is collective
            self._group = self.dist.new_group(backend="gloo")
            self._ranks = list(range(self.dist.get_world_size()))
            self._jobs = queue.Queue()
            self._thread = threading.Thread(target=self._helper, args=(inp.device,), daemon=True)
            self._thread.start()
        key = (slot, ring)
        if key not in self._staging:
            cap = rk.send[slot][ring].numel()
            self._staging[key] = (torch.empty(cap, dtype=torch.uint8, pin_memory=True),
                                  torch.empty(cap * rk.world, dtype=torch.uint8, pin_memory=True),
                                  torch.cuda.Event(), torch.cuda.Event())
        h, o, packed, done = self._staging[key]
        packed.record(stream)
        issued = threading.Event()
        self._jobs.put((packed, inp, out, h[:S], o[:S * rk.world], list(rk.gather_sizes[slot][ring]), rk.rank, issued, done))
        return self._HostWork(self, issued, done, stream)

    def close(self):
        if self._thread is not None:
            self._jobs.put(None)
            self._thread.join()
            self._thread = None


class SlicedIndexedTree:
    """One rank of the sliced single-list tree.  step(vals) starts a round (vals = the WHOLE step, world x batch
    values, identical on every rank) and advances the global schedule by one round period; flush() runs it dry.
    outputs(R) are rank's witnesses of round R (its slice), valid after flush() / sync or after done_event(R)."""

    def __init__(self, backend, world, rank, transport, lag=None):
        self.be, self.world, self.rank, self.tp = backend, world, rank, transport
        self.sched = SliceSchedule(world, backend.units, lag)
        S, D = SliceSchedule.STREAMS, self.sched.lag
        self.ring = D + 1
        pb = backend.payload_bytes
        self.gather_bytes = [[pb] * self.ring for _ in range(S)]     # per collective in flight: bytes per rank
        self.gather_sizes = [[[0] * world for _ in range(self.ring)] for _ in range(S)]    # ... and what each rank's payload uses
        self.send = [[backend.make_buffer(pb) for _ in range(self.ring)] for _ in range(S)]
        self.recv = [[backend.make_buffer(pb * world) for _ in range(self.ring)] for _ in range(S)]
        self.work = [[None] * self.ring for _ in range(S)]
        self.tick_ev = [[backend.new_event() for _ in range(self.sched.round_ticks)] for _ in range(S)]
        self.done_ev = [backend.new_event() for _ in range(S)]
        self.rounds = []                 # per round: dict(slice=.., out_slot=.., size_before=.., n=..)
        self.starts = []                 # global tick of every round's tick 0 (the same on every rank)
        self.T = 0                       # next global tick to issue

    # ---- the three phases of (round R, round tick rt) ----
    def phase_apply(self, R, rt):
        sc, D = self.sched, self.sched.lag
        src = rt - D
        if src < 0 or not sc.has_gather(src):
            return
        slot, ring = R % sc.STREAMS, src % self.ring
        st = self.be.streams[slot]
        rd = self.rounds[R]
        with self.be.stream_ctx(st):
            w = self.work[slot][ring]
            if w is not None:
                w.wait()                 # the round's stream waits for the collective; the host does not
                self.work[slot][ring] = None
            units = sc.payload_units(src)
            units[self.rank] = -1        # own write-backs are already in this replica
            if any(q >= 0 for q in units):
                if R >= 1:
                    # a write-back of round R lands on a node after every write-back round R - 1 made to that level
                    # (they run on different streams): behind that round's tick max(unit) + world * lag
                    st.wait_event(self.tick_ev[(R - 1) % sc.STREAMS][max(units) + sc.period])
                b = rd["n"]
                self.be.apply_gathered(self.recv[slot][ring], self.gather_bytes[slot][ring],
                                       [rd["size_before"] + g * b for g in range(self.world)], [b] * self.world, units, st)

    def phase_compute(self, R, rt):
        sc = self.sched
        q = sc.unit_of(self.rank, rt)
        slot = R % sc.STREAMS
        st = self.be.streams[slot]
        rd = self.rounds[R]
        if q is not None:
            with self.be.stream_ctx(st):
                if q >= 1 and R >= 1:
                    # level q - 1 of every slice of round R - 1 must be in this replica: applied (others) or written
                    # back (own) by the end of that round's tick q + world * lag
                    st.wait_event(self.tick_ev[(R - 1) % sc.STREAMS][q + sc.period])
                self.be.unit(rd["slice"], q, self.send[slot][rt % self.ring], st)
                if q == sc.units - 1:
                    self.done_ev[slot].record(st)

    def phase_send(self, R, rt):
        sc = self.sched
        slot = R % sc.STREAMS
        st = self.be.streams[slot]
        with self.be.stream_ctx(st):
            if sc.has_gather(rt):
                ring = rt % self.ring
                rd = self.rounds[R]
                # every rank contributes as many bytes as the largest payload of this tick needs
                sizes = [self.be.unit_bytes(rd["size_before"] + g * rd["n"], rd["n"], q) if q >= 0 else 0
                         for g, q in enumerate(sc.payload_units(rt))]
                self.gather_sizes[slot][ring] = sizes
                self.gather_bytes[slot][ring] = max(sizes)
                self.work[slot][ring] = self.tp.all_gather(self, slot, ring, st)
            self.tick_ev[slot][rt].record(st)

    # ---- driving ----
    def _start_round(self, vals):
        R = len(self.rounds)
        b = vals.shape[0] // self.world
        if vals.shape[0] != b * self.world or not 0 < b <= self.be.batch:
            raise ValueError(f"a step is world x n values, 0 < n <= batch = {self.be.batch} (a shorter step has shorter slices)")
        size_before = self.be.size()
        out_slot = R % len(self.be.sets)
        sl = self.be.prepare(vals, self.rank * b, b, (self.world - 1 - self.rank) * b, out_slot)
        self.rounds.append(dict(slice=sl, out_slot=out_slot, size_before=size_before, n=b))
        self.starts.append(self.sched.next_start(self.starts, self.T))
        return R

    def _run_ticks(self, upto):
        sc = self.sched
        while self.T < upto:
            for R, rt in sc.active_rounds(self.T, self.starts):
                self.phase_apply(R, rt)
                self.phase_compute(R, rt)
                self.phase_send(R, rt)
            self.T += 1

    def step(self, vals):
        R = self._start_round(vals)
        self._run_ticks(self.starts[R] + self.sched.period)
        return R

    def flush(self):
        """issue everything that is left of the rounds in flight and wait for it"""
        if self.rounds:
            self._run_ticks(self.starts[-1] + self.sched.round_ticks)
        self.be.sync()

    def close(self):
        if hasattr(self.tp, "close"):
            self.tp.close()

    def outputs(self, R):
        """this rank's witnesses of round R: rows [0, n) of every field (n = the round's slice length); sibling arrays
        are [depth, n, 32]"""
        rd = self.rounds[R]
        d = self.be.outputs(rd["out_slot"], rd["n"])
        d["first_insertion"] = rd["size_before"] + self.rank * rd["n"]      # = first new leaf index
        return d

    def done_event(self, R):
        return self.done_ev[R % self.sched.STREAMS]


class LocalWorld:
    """All `world` ranks in ONE process, one replica each on the same GPU: the schedule in lockstep with device-to-device
    copies as the all-gather.  This is how a one-GPU box runs (and tests) the multi-GPU path at world 2, 4, 8."""

    class _Work:
        def __init__(self, events, stream):
            self.events, self.stream = events, stream

        def wait(self):                  # like a collective's: nobody still reads my send buffer, my gather is complete
            for e in self.events:
                self.stream.wait_event(e)

    class _Transport:
        def __init__(self, world):
            self.world = world
            self.peers = None            # list of SlicedIndexedTree, set by LocalWorld
            self.packed = self.copied = None
            self.bytes_moved, self.collectives = 0, 0

        def all_gather(self, rk, slot, ring, stream):
            # every rank has recorded packed[slot][ring][rank] by now (LocalWorld drives the phases in lockstep)
            S = rk.gather_bytes[slot][ring]
            for h, peer in enumerate(self.peers):
                if h == rk.rank:
                    continue
                stream.wait_event(self.packed[slot][ring][h])
                rk.recv[slot][ring][h * S:(h + 1) * S].copy_(peer.send[slot][ring][0:S], non_blocking=True)
            self.copied[slot][ring][rk.rank].record(stream)
            self.collectives += 1
            self.bytes_moved += S * self.world
            return LocalWorld._Work(self.copied[slot][ring], stream)

    def __init__(self, backends, lag=None):
        self.world = len(backends)
        self.tp = self._Transport(self.world)
        self.ranks = [SlicedIndexedTree(be, self.world, g, self.tp, lag) for g, be in enumerate(backends)]
        self.tp.peers = self.ranks
        self.sched = self.ranks[0].sched
        mk = lambda: [[[be.new_event() for be in backends] for _ in range(self.ranks[0].ring)]
                      for _ in range(SliceSchedule.STREAMS)]
        self.tp.packed, self.tp.copied = mk(), mk()
        self.T = 0

    def _run_ticks(self, upto):
        sc = self.sched
        while self.T < upto:
            for R, rt in sc.active_rounds(self.T, self.ranks[0].starts):
                slot, ring = R % sc.STREAMS, rt % self.ranks[0].ring
                for rk in self.ranks:
                    rk.phase_apply(R, rt)
                for rk in self.ranks:
                    rk.phase_compute(R, rt)
                    if sc.has_gather(rt):
                        self.tp.packed[slot][ring][rk.rank].record(rk.be.streams[slot])
                for rk in self.ranks:
                    rk.phase_send(R, rt)
            self.T += 1

    def step(self, vals_per_rank):
        """vals_per_rank: the step's values, one copy per rank (each on that rank's device)"""
        R = None
        for rk, v in zip(self.ranks, vals_per_rank):
            rk.T = self.T
            R = rk._start_round(v)
        self._run_ticks(self.ranks[0].starts[R] + self.sched.period)
        return R

    def flush(self):
        if self.ranks[0].rounds:
            self._run_ticks(self.ranks[0].starts[-1] + self.sched.round_ticks)
        for rk in self.ranks:
            rk.be.sync()
