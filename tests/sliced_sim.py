"""Test infrastructure: a symbolic backend for the product's multi-GPU schedule code.

tests/native/sliced_sym.cpp compiles indexed-merkle-tree-halo2_amd/csrc/imt_sliced_sched.hpp -- the SAME Rank / World /
LocalTransport classes libimt_hip.so runs over HIP -- with g++ and forwards every stream, event, buffer and slice
operation to the callbacks below.  Nothing is hashed.  A replica is, per tree level, the ordered list of slices (global
slice number k = round * world + rank) whose write-backs it holds; a unit asserts that its level holds exactly the
slices before it, a payload names (k, unit), an apply asserts order.  Two execution modes:

  deferred   streams are FIFO queues, events follow HIP semantics (a wait captures the latest record issued before
             it), buffers copy when their op runs; run() executes the queues in a RANDOM interleaving that respects
             only stream order and event waits -- so a missing event or a too-early buffer reuse shows up as a failed
             assertion for some seed.  All ranks in one process (the in-process transport).
  immediate  ops run when issued: one rank per process, the collective through torch.distributed (gloo).
"""
import ctypes
import os
import random
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUNDS = 4
LIB = os.path.join(ROOT, "tests", "native", "libslicedsym.so")


def build_lib():
    src = os.path.join(ROOT, "tests", "native", "sliced_sym.cpp")
    hdr = os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "csrc", "imt_sliced_sched.hpp")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        tmp = LIB + f".{os.getpid()}.tmp"
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", tmp, src], check=True)
        os.replace(tmp, LIB)
    return LIB


c_int, c_size_t, c_u64, c_uint = ctypes.c_int, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint
P = ctypes.POINTER
CB_TYPES = [
    ("record", ctypes.CFUNCTYPE(c_int, c_int, c_int, c_int)),
    ("wait", ctypes.CFUNCTYPE(c_int, c_int, c_int, c_int)),
    ("event_sync", ctypes.CFUNCTYPE(c_int, c_int, c_int)),
    ("alloc", ctypes.CFUNCTYPE(c_int, c_int, c_int, c_size_t)),
    ("copy", ctypes.CFUNCTYPE(c_int, c_int, c_int, c_size_t, c_int, c_int, c_size_t, c_size_t, c_int)),
    ("tree_size", ctypes.CFUNCTYPE(c_u64, c_int)),
    ("unit_bytes", ctypes.CFUNCTYPE(c_size_t, c_int, c_u64, c_size_t, c_uint)),
    ("prepare", ctypes.CFUNCTYPE(c_int, c_int, c_size_t, c_size_t, c_size_t, c_int, P(c_int))),
    ("unit", ctypes.CFUNCTYPE(c_int, c_int, c_int, c_uint, c_int, c_int)),
    ("apply_gathered", ctypes.CFUNCTYPE(c_int, c_int, c_int, c_size_t, c_int, P(c_u64), P(c_u64), P(ctypes.c_int32), c_int)),
    ("sync", ctypes.CFUNCTYPE(c_int, c_int)),
    ("all_gather", ctypes.CFUNCTYPE(c_int, c_int, c_int, c_int, c_int, c_int, c_size_t, c_int)),
    ("fence", ctypes.CFUNCTYPE(c_int, c_int, c_int, c_int, c_int)),
]


class Callbacks(ctypes.Structure):
    _fields_ = CB_TYPES


def load():
    lib = ctypes.CDLL(build_lib())
    lib.sym_schedule.argtypes = [c_int, c_int, c_int, P(c_int)]
    lib.sym_unit_of.argtypes = [c_int] * 5
    lib.sym_payload_units.argtypes = [c_int, c_int, c_int, c_int, P(ctypes.c_int32)]
    lib.sym_world_create.restype = ctypes.c_void_p
    lib.sym_world_create.argtypes = [P(Callbacks), c_int, c_int, c_int, c_size_t, c_int, c_int, c_size_t]
    lib.sym_world_step.argtypes = [ctypes.c_void_p, c_size_t, P(c_u64)]
    for f in ("sym_world_flush", "sym_world_run_all"):
        getattr(lib, f).argtypes = [ctypes.c_void_p]
    lib.sym_world_wait.argtypes = [ctypes.c_void_p, c_int, c_u64]
    lib.sym_world_collectives.restype = c_u64
    lib.sym_world_collectives.argtypes = [ctypes.c_void_p]
    lib.sym_world_destroy.argtypes = [ctypes.c_void_p]
    lib.sym_world_create_channels.restype = ctypes.c_void_p
    lib.sym_world_create_channels.argtypes = [P(Callbacks), c_int, c_int, c_size_t, c_int, c_int, c_size_t, c_int]
    lib.sym_set_layout.argtypes = [c_int, c_int]
    lib.sym_set_layout.restype = None
    lib.sym_world_tick.restype = c_u64
    lib.sym_world_tick.argtypes = [ctypes.c_void_p]
    return lib


class Schedule:
    """the product's schedule arithmetic (imt::sliced::Schedule) through the test library"""

    def __init__(self, lib, world, units, lag=None):
        out = (c_int * 4)()
        if lib.sym_schedule(world, units, lag or 0, out) != 0:
            raise ValueError("not a schedule")
        self.lib, self.world, self.units = lib, world, units
        self.lag, self.period, self.gathers, self.round_ticks = (int(x) for x in out)

    def unit_of(self, rank, rt):
        q = self.lib.sym_unit_of(self.world, self.units, self.lag, rank, rt)
        return None if q < 0 else q

    def payload_units(self, rt):
        out = (ctypes.c_int32 * self.world)()
        self.lib.sym_payload_units(self.world, self.units, self.lag, rt, out)
        return list(out)

    def has_gather(self, rt):
        out = (ctypes.c_int32 * self.world)()
        return bool(self.lib.sym_payload_units(self.world, self.units, self.lag, rt, out) & 2)


class Sim:
    """streams as FIFO queues + HIP event semantics"""

    def __init__(self, immediate=False, seed=0):
        self.immediate, self.rng, self.streams = immediate, random.Random(seed), []
        self.executed = 0

    def stream(self):
        s = []
        self.streams.append(s)
        return s

    def push(self, stream, op):
        if self.immediate:
            self.execute(op)
        else:
            stream.append(op)

    def execute(self, op):
        kind, arg = op
        if kind == "record":
            arg[0] = True
        elif kind == "wait":
            assert arg[0], "immediate mode: waiting for an event that has not happened"
        else:
            arg()
        self.executed += 1

    def run(self):
        """drain every queue in a random order that respects stream FIFO and event waits"""
        starved, budget = set(), 0
        ids = list(range(len(self.streams)))
        while True:
            ready = [i for i in ids if self.streams[i] and not (self.streams[i][0][0] == "wait" and not self.streams[i][0][1][0])]
            if not ready:
                assert not any(self.streams), "deadlock: every stream waits for an event nobody records"
                return
            if budget <= 0:       # adversary: some streams get no time for a while (a slow GPU, a descheduled queue)
                starved = set(self.rng.sample(ids, self.rng.randrange(0, max(1, len(ids) * 2 // 3) + 1)))
                budget = self.rng.randrange(1, 400)
            budget -= 1
            pool = [i for i in ready if i not in starved] or ready
            self.execute(self.streams[self.rng.choice(pool)].pop(0))


class Replica:
    """what one rank's callbacks act on"""

    def __init__(self, sim, depth, batch, world, rank):
        self.sim, self.depth, self.batch, self.world, self.rank = sim, depth, batch, world, rank
        self.units = depth + 1
        self.streams = [sim.stream() for _ in range(3 * ROUNDS)]       # round streams, comm streams, apply streams
        self.events = {}                                   # id -> token of the latest record issued (or None)
        self.buffers = {}                                  # id -> int64 words
        self.levels = [[] for _ in range(depth)]           # slices written back per level, in arrival order
        self.size = 1
        self.slices = {}
        self.computed = []                                 # (k, unit) in execution order
        self.next_id = 0


class SymWorld:
    """n_local == world: every rank here (deferred or immediate).  n_local == 1 with `dist`: one rank of a gloo world."""

    WORDS = 4                                              # payload: [k, unit, checksum, pad] int64

    def __init__(self, lib, sim, world, depth, batch, lag=None, first_rank=0, n_local=None, dist=None):
        self.lib, self.sim, self.world, self.depth, self.batch, self.dist = lib, sim, world, depth, batch, dist
        n_local = world if n_local is None else n_local
        self.reps = {first_rank + k: Replica(sim, depth, batch, world, first_rank + k) for k in range(n_local)}
        self.errors = []
        self.sched = Schedule(lib, world, depth + 1, lag)
        self._cb = Callbacks(**{name: typ(self._guard(getattr(self, "_" + name), name)) for name, typ in CB_TYPES})
        self.h = lib.sym_world_create(ctypes.byref(self._cb), world, first_rank, n_local, batch, depth, lag or 0, self.WORDS * 8)
        assert self.h, "sym_world_create failed"
        self.rounds = 0

    def _guard(self, fn, name):
        def wrapped(*a):
            try:
                r = fn(*a)
                return 0 if r is None else r
            except BaseException as e:          # an exception cannot cross the C frames: keep it, fail the call
                self.errors.append(f"{name}{a}: {type(e).__name__}: {e}")
                return -12
        return wrapped

    def _check(self, rc):
        assert not self.errors, self.errors[0]
        assert rc == 0, f"schedule call failed: {rc}"

    # ---- callbacks
    def _record(self, rank, ev, stream):
        rp = self.reps[rank]
        tok = [False]
        rp.events[ev] = tok
        self.sim.push(rp.streams[stream], ("record", tok))

    def _wait(self, rank, stream, ev):
        owner, ev = divmod(ev, 100000)
        tok = self.reps[owner].events.get(ev)
        if tok is not None:                    # HIP: a wait on an event never recorded is a no-op
            self.sim.push(self.reps[rank].streams[stream], ("wait", tok))

    def _event_sync(self, rank, ev):
        self.sim.run() if not self.sim.immediate else None
        tok = self.reps[rank].events.get(ev)
        assert tok is None or tok[0]

    def _alloc(self, rank, buf, nbytes):
        self.reps[rank].buffers[buf] = np.zeros(nbytes // 8, dtype=np.int64)

    def _copy(self, rank, dst, doff, src_rank, src, soff, nbytes, stream):
        d, s = self.reps[rank].buffers[dst], self.reps[src_rank].buffers[src]
        assert doff % 8 == 0 and soff % 8 == 0 and nbytes % 8 == 0

        def run():
            d[doff // 8:(doff + nbytes) // 8] = s[soff // 8:(soff + nbytes) // 8]
        self.sim.push(self.reps[rank].streams[stream], ("fn", run))

    def _tree_size(self, rank):
        return self.reps[rank].size

    def _unit_bytes(self, rank, size_before, n, q):
        # like the product: smaller payloads higher up (here: the full 32 bytes for the lower half of the units, 24 above)
        return self.WORDS * 8 if q <= (self.depth + 1) // 2 else 24

    def _fence(self, rank, slot, ring, stream):
        return 0

    def _prepare(self, rank, n_before, n_own, n_after, slot, slice_out):
        rp = self.reps[rank]
        assert n_before == rank * n_own and n_own == self.batch and n_after == (self.world - 1 - rank) * n_own
        k = (rp.size - 1 + n_before) // self.batch
        assert (rp.size - 1) % (self.batch * self.world) == 0
        sid = rp.next_id % (ROUNDS + 1)
        rp.next_id += 1
        assert sid not in rp.slices or rp.slices[sid]["next"] == rp.units, "plan set still open"
        rp.slices[sid] = dict(k=k, next=0)
        rp.size += n_before + n_own + n_after
        slice_out[0] = sid

    def _unit(self, rank, slice_id, q, payload, stream):
        rp = self.reps[rank]
        sl = rp.slices[slice_id]
        assert sl["next"] == q, "units out of order"
        sl["next"] = q + 1
        k = sl["k"]
        w = rp.buffers[payload]

        def run():
            if q >= 2:
                # the level it READS (children and siblings of the nodes it computes): every earlier slice's write-backs
                # and its own (its previous unit), nothing of a later slice yet
                # (arrivals are appended in strictly increasing order -- asserted where they are appended -- so length and
                # last element pin the whole list)
                rd = rp.levels[q - 2]
                assert len(rd) == k + 1 and rd[-1] == k, f"rank {rank}: slice {k} unit {q} reads level {q - 2} holding {rd[-6:]} (wants 0..{k})"
            if q >= 1:
                lvl = rp.levels[q - 1]
                assert len(lvl) == k and (k == 0 or lvl[-1] == k - 1), f"rank {rank}: slice {k} level {q - 1} sees {lvl[-6:]} (wants 0..{k - 1})"
                lvl.append(k)
            w[0], w[1], w[2] = k, q, k * 1000003 + q
            rp.computed.append((k, q))
        self.sim.push(rp.streams[stream], ("fn", run))

    def _apply_gathered(self, rank, gathered, stride, count, size_before, n, units, stream):
        rp = self.reps[rank]
        assert count == self.world and stride % 8 == 0
        exp = [((size_before[r] - 1) // self.batch) for r in range(count)]
        qs = [units[r] for r in range(count)]
        w = rp.buffers[gathered]

        def run():
            for r, q in enumerate(qs):
                if q < 0:
                    continue
                k, uq, chk = (int(x) for x in w[r * stride // 8: r * stride // 8 + 3])
                assert (k, uq) == (exp[r], q) and chk == k * 1000003 + q, \
                    f"rank {rank}: payload slot {r} holds slice {k} unit {uq}, expected slice {exp[r]} unit {q}"
                lvl = rp.levels[q - 1]
                assert not lvl or lvl[-1] < k, f"rank {rank}: level {q - 1} gets slice {k} after {lvl[-1]}"
                lvl.append(k)
        self.sim.push(rp.streams[stream], ("fn", run))

    def _sync(self, rank):
        if not self.sim.immediate:
            self.sim.run()

    def _all_gather(self, rank, slot, ring, send, recv, nbytes, stream):
        import torch
        rp = self.reps[rank]
        assert self.sim.immediate and nbytes % 8 == 0
        out = torch.from_numpy(rp.buffers[recv][:self.world * nbytes // 8])
        self.dist.all_gather_into_tensor(out, torch.from_numpy(rp.buffers[send][:nbytes // 8]))

    # ---- driving
    def step(self):
        R = c_u64()
        self._check(self.lib.sym_world_step(self.h, self.batch, ctypes.byref(R)))
        self.rounds += 1
        return int(R.value)

    def flush(self):
        self._check(self.lib.sym_world_flush(self.h))

    def run_all(self):
        self._check(self.lib.sym_world_run_all(self.h))

    def wait(self, k, R):
        self._check(self.lib.sym_world_wait(self.h, k, R))

    @property
    def collectives(self):
        return int(self.lib.sym_world_collectives(self.h))

    def close(self):
        if self.h:
            self.lib.sym_world_destroy(self.h)
            self.h = None
