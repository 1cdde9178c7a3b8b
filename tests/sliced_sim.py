"""Test infrastructure: a symbolic backend for indexed-merkle-tree-halo2_amd/sliced.py.

It hashes nothing.  A replica is, per tree level, the ordered list of slices (global slice number k = round * world +
rank) whose write-backs it holds; a unit asserts that its level holds exactly the slices before it, a payload names
(k, unit), an apply asserts order.  Two execution modes:

  deferred   streams are FIFO queues, events follow HIP semantics (a wait captures the latest record issued before
             it), buffers copy when their op runs; run() executes the queues in a RANDOM interleaving that respects
             only stream order and event waits -- so a missing event or a too-early buffer reuse shows up as a failed
             assertion for some seed.  Used with LocalWorld.
  immediate  ops run when issued, buffers are real torch CPU tensors: for the torch.distributed (gloo) transport.
"""
import contextlib
import random

import numpy as np
import torch

_current = [None]


class Event:
    def __init__(self, sim):
        self.sim, self.last = sim, None          # last = token of the latest record issued

    def record(self, stream):
        tok = [False]
        self.last = tok
        stream.push(("record", tok))


class Stream:
    def __init__(self, sim):
        self.sim, self.q = sim, []
        self.cuda_stream = 0

    def push(self, op):
        if self.sim.immediate:
            self.sim.execute(op)
        else:
            self.q.append(op)

    def wait_event(self, ev):
        if ev.last is not None:
            self.push(("wait", ev.last))

    def synchronize(self):
        self.sim.run()


class Buffer:
    """deferred-mode payload buffer: int64 words; slicing by BYTES like the uint8 tensors of the product"""

    def __init__(self, sim, arr):
        self.sim, self.a = sim, arr
        self.is_cuda = False

    def __getitem__(self, sl):
        return Buffer(self.sim, self.a[(sl.start or 0) // 8:sl.stop // 8])

    def copy_(self, src, non_blocking=False):
        dst, s = self.a, src.a
        _current[0].push(("fn", lambda: dst.__setitem__(slice(None), s)))

    def numel(self):
        return self.a.size * 8


class Sim:
    def __init__(self, immediate=False, seed=0):
        self.immediate, self.rng, self.streams = immediate, random.Random(seed), []
        self.executed = 0

    def stream(self):
        s = Stream(self)
        self.streams.append(s)
        return s

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
        while True:
            ready = [s for s in self.streams if s.q and not (s.q[0][0] == "wait" and not s.q[0][1][0])]
            if not ready:
                assert not any(s.q for s in self.streams), "deadlock: every stream waits for an event nobody records"
                return
            if budget <= 0:       # adversary: some streams get no time for a while (a slow GPU, a descheduled queue)
                starved = set(self.rng.sample(self.streams, self.rng.randrange(0, max(1, len(self.streams) * 2 // 3) + 1)))
                budget = self.rng.randrange(1, 400)
            budget -= 1
            pool = [s for s in ready if s not in starved] or ready
            s = self.rng.choice(pool)
            self.execute(s.q.pop(0))


class SymbolicBackend:
    """same methods as sliced.SliceGpuBackend"""

    WORDS = 4                                  # payload: [k, unit, checksum, pad] int64

    def __init__(self, sim, depth, batch, world, rank):
        self.sim, self.depth, self.batch, self.world, self.rank = sim, depth, batch, world, rank
        self.units = depth + 1
        self.payload_bytes = self.WORDS * 8
        self.streams = [sim.stream() for _ in range(4)]
        self.sets = [dict(slot=i) for i in range(5)]
        self.levels = [[] for _ in range(depth)]          # slices written back per level, in arrival order
        self._size = 1
        self.slices = {}
        self.computed = []                                # (k, unit) in execution order
        self.next_id = 0

    def size(self):
        return self._size

    def make_buffer(self, nbytes):
        if self.sim.immediate:
            return torch.zeros(nbytes, dtype=torch.uint8)
        return Buffer(self.sim, np.zeros(nbytes // 8, dtype=np.int64))

    def new_event(self):
        return Event(self.sim)

    @contextlib.contextmanager
    def stream_ctx(self, stream):
        prev, _current[0] = _current[0], stream
        try:
            yield
        finally:
            _current[0] = prev

    def prepare(self, vals, n_before, n_own, n_after, out_slot):
        assert n_before == self.rank * n_own and n_own == self.batch
        k = (self._size - 1 + n_before) // self.batch
        assert (self._size - 1) % (self.batch * self.world) == 0
        sid = self.next_id % 5
        self.next_id += 1
        assert sid not in self.slices or self.slices[sid]["next"] == self.units, "plan set still open"
        self.slices[sid] = dict(k=k, next=0)
        self._size += n_before + n_own + n_after
        return sid

    def _words(self, buf):
        return buf.view(torch.int64) if torch.is_tensor(buf) else buf.a

    def unit_bytes(self, size_before, n, q):
        # like the product: smaller payloads higher up (here: the full 32 bytes for the lower half of the units, 24 above)
        return self.payload_bytes if q <= self.units // 2 else 24

    def unit(self, slice_id, q, payload, stream):
        sl = self.slices[slice_id]
        assert sl["next"] == q, "units out of order"
        sl["next"] = q + 1
        k = sl["k"]

        def run():
            if q >= 1:
                lvl = self.levels[q - 1]
                assert lvl == list(range(k)), f"rank {self.rank}: slice {k} level {q - 1} sees {lvl[-6:]} (wants 0..{k - 1})"
                lvl.append(k)
            w = self._words(payload)
            w[0], w[1], w[2] = k, q, k * 1000003 + q
            self.computed.append((k, q))
        stream.push(("fn", run))

    def apply_gathered(self, gathered, stride, size_before, n, units, stream):
        exp = [((sb - 1) // self.batch) for sb in size_before]

        def run():
            w = self._words(gathered)
            for r, q in enumerate(units):
                if q < 0:
                    continue
                k, uq, chk = (int(x) for x in w[r * stride // 8: r * stride // 8 + 3])
                assert (k, uq) == (exp[r], q) and chk == k * 1000003 + q, \
                    f"rank {self.rank}: payload slot {r} holds slice {k} unit {uq}, expected slice {exp[r]} unit {q}"
                lvl = self.levels[q - 1]
                assert not lvl or lvl[-1] < k, f"rank {self.rank}: level {q - 1} gets slice {k} after {lvl[-1]}"
                lvl.append(k)
        stream.push(("fn", run))

    def outputs(self, slot, n=None):
        return dict(self.sets[slot])

    def sync(self):
        self.sim.run()
