"""Multi-GPU: one depth-D tree as `world` indexed subtrees, one per GPU (north_star's sharding by
leaf-index range).

Rank g owns the values with v % world == g and the leaf-index range [g << (D-k), (g+1) << (D-k)) of the
depth-D tree (k = log2 world) as an indexed subtree of height D-k with its own {0,0,0} sentinel at its
first leaf.  Insertions never cross ranks.  Every step the ranks all-gather their `world` subtree roots
(32 bytes each; RCCL over xGMI with the "nccl" backend) -- the one collective of the path -- and every
rank then LIFTS the witnesses of its own insertions to the full depth D (imt_itree_lift_batch): k more
hashes per root and the k upper siblings per proof, so that what comes out is exactly what the
reference's insert_leaf takes (src/indexed_merkle_tree.rs:231-245: depth-D proofs, depth-D roots, 2 + 2 D
hashes per insertion) with global leaf indices.  Order of a step across ranks: rank 0's insertions, then
rank 1's, ...; a rank's upper siblings therefore mix the other ranks' roots after the step (ranks below
it) and before the step (ranks above it).  The exchange runs one step behind the insertions so that it
waits for a finished batch while two more are in flight.

With world == 1 this is the reference's single tree (:632-671, :715-735).  For world > 1 the root
commits to `world` sorted lists instead of one (non-membership of v is decided by the subtree that owns
v); the reference has no multi-device form, so parity is defined against the CPU oracle building the same
subtrees and, at small depth, a dense rebuild of the whole tree after every event
(tests/test_sharded_gloo.py on CPU with gloo; tests/test_gpu_parity.py through GpuBackend).

The compute backend is pluggable so that the collective logic runs on CPU with gloo: `GpuBackend` (the
product: libimt_hip.so) or, in tests only, an oracle-backed stand-in with the same methods.
"""
import ctypes

import numpy as np
import torch


class GpuBackend:
    """The product path: one imt context and one placed imt_itree on this rank's GPU.  Witnesses stay in
    HBM (torch tensors), in `nbuf` rotating buffer sets so that the set of a batch is not rewritten
    before its lift has run."""

    FIELDS = ("low_index", "low_leaf", "is_largest", "old_root", "interim_root", "new_root", "new_leaf", "low_sib",
              "new_sib")

    def __init__(self, imt, device_index, depth, world, rank, capacity, batch, pipeline=True, inputs_ready=False,
                 nbuf=2, fmt=0, host_prep=False, pinned_outputs=False):
        self.imt, self.F = imt, imt._ffi
        self.depth, self.world, self.rank, self.batch, self.fmt = depth, world, rank, batch, fmt
        self.sub_height = depth - (world.bit_length() - 1)
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(device_index)
        self.ctx = imt.Context(device_index)
        self.stream_ptr = torch.cuda.current_stream().cuda_stream      # 0 = the null stream
        self.ctx.set_stream(self.stream_ptr)
        self.transport = None        # ShardedIndexedTree(transport=...): sync() then also asks it for a GPU-side timeout
        self.tree = imt.IndexedTree(self.ctx, self.sub_height, capacity)
        self.tree.set_placement(depth, rank)
        self.ctx._check(imt.lib.imt_itree_set_value_partition(self.tree.h, world, rank))
        u8 = dict(dtype=torch.uint8, device=self.device)
        # pinned_outputs: the witnesses land in page-locked HOST memory, written by the kernels over PCIe
        # (device-addressable, so they are still passed as IMT_DEVICE_PTRS): what a host-language caller wants
        o8 = dict(dtype=torch.uint8, device="cpu", pin_memory=True) if pinned_outputs else u8
        o64 = dict(dtype=torch.int64, device="cpu", pin_memory=True) if pinned_outputs else dict(dtype=torch.int64,
                                                                                                 device=self.device)
        self.sets = [dict(low_index=torch.empty(batch, **o64),
                          low_leaf=torch.empty((batch, 3, 32), **o8), is_largest=torch.empty(batch, **o8),
                          old_root=torch.empty((batch, 32), **o8), interim_root=torch.empty((batch, 32), **o8),
                          new_root=torch.empty((batch, 32), **o8), new_leaf=torch.empty((batch, 3, 32), **o8),
                          low_sib=torch.empty((depth, batch, 32), **o8), new_sib=torch.empty((depth, batch, 32), **o8))
                     for _ in range(nbuf)]
        self.structs = [self.F.InsertOut(**{k: t.data_ptr() for k, t in b.items()}) for b in self.sets]
        self.flags = self.F.DEVICE_PTRS | fmt
        self.ins_flags = (self.flags | (self.F.PIPELINE if pipeline else 0) | (self.F.INPUTS_READY if inputs_ready else 0)
                          | (self.F.HOST_PREP if host_prep else 0))
        self.next_slot = 0
        self.first_index = [0] * nbuf
        self._root = torch.empty(32, **u8)

    def initial_root(self):
        """root of a subtree that holds only its sentinel = the empty subtree of that height"""
        return torch.from_numpy(self.ctx.zero_hashes(self.sub_height, self.fmt)[self.sub_height].copy())

    def insert(self, vals, flags=None):
        """vals: uint8 [batch, 32] (torch on this device, or numpy).  Enqueues the batch; returns its slot."""
        if not torch.is_tensor(vals):
            vals = torch.from_numpy(np.ascontiguousarray(vals))
        vals = vals.to(self.device)
        if vals.shape != (self.batch, 32):
            raise ValueError(f"expected [{self.batch}, 32] values")
        slot = self.next_slot
        self.next_slot = (slot + 1) % len(self.sets)
        self.first_index[slot] = self.tree.index_base + self.tree.size
        rc = self.imt.lib.imt_itree_insert_batch(self.tree.h, ctypes.c_void_p(vals.data_ptr()), self.batch,
                                                 ctypes.byref(self.structs[slot]),
                                                 self.ins_flags if flags is None else flags)
        if rc == self.F.ERR["VALUE"]:
            raise ValueError(self.imt.lib.imt_last_error(self.ctx.h).decode())
        self.ctx._check(rc)
        return slot

    def root_after(self, lag):
        """this subtree's root after the batch inserted `lag` calls ago; orders the stream behind that batch"""
        self.ctx._check(self.imt.lib.imt_itree_root_lagged(self.tree.h, lag, ctypes.c_void_p(self._root.data_ptr()),
                                                           self.flags))
        return self._root

    def lift(self, slot, roots_before, roots_after):
        rb, ra = (r.to(self.device).contiguous() for r in (roots_before, roots_after))
        self.ctx._check(self.imt.lib.imt_itree_lift_batch(self.tree.h, ctypes.c_void_p(rb.data_ptr()),
                                                          ctypes.c_void_p(ra.data_ptr()), self.world, self.batch,
                                                          ctypes.byref(self.structs[slot]), self.flags))

    def combine(self, roots):
        out = torch.empty(32, dtype=torch.uint8, device=self.device)
        r = roots.to(self.device).contiguous()
        self.ctx._check(self.imt.lib.imt_combine_subtree_roots(self.ctx.h, ctypes.c_void_p(r.data_ptr()), self.world,
                                                               self.sub_height, self.depth,
                                                               ctypes.c_void_p(out.data_ptr()), self.flags))
        return out

    def non_membership_witness(self, vals, roots):
        """(low index, low leaf, siblings [depth, n, 32], is_largest), numpy: verify_non_inclusion's witness for values
        of this rank's residue against the GLOBAL root whose subtree roots are `roots` (the tree must be idle)"""
        self.sync()
        r = roots.cpu().numpy() if torch.is_tensor(roots) else roots
        v = vals.cpu().numpy() if torch.is_tensor(vals) else vals
        return self.tree.non_membership_witness(v, subtree_roots=r)

    def outputs(self, slot):
        """the slot's witness tensors (valid on the host after sync()) + the global index of its first new leaf"""
        d = dict(self.sets[slot])
        d["first_new_index"] = self.first_index[slot]
        return d

    def sync(self):
        self.ctx.sync()
        torch.cuda.synchronize()
        if self.transport is not None:
            # a GPU-side wait of the root exchange that gave up on a peer left the gathered roots unwritten (zeros) and
            # everything computed from them wrong: say so before anybody reads the outputs (imt_transport_poll_error)
            rc = self.imt.lib.imt_transport_poll_error(self.transport)
            if rc:
                raise self.imt.ImtError(rc, self.imt.lib.imt_transport_last_error(self.transport).decode())


class ShardedIndexedTree:
    """step(vals) inserts this rank's batch and finishes -- root exchange + lift -- the PREVIOUS step,
    whose depth-D witnesses it returns; flush() finishes the last one."""

    def __init__(self, backend, depth, world=1, rank=0, dist=None, via_host=False, transport=None, transport_ctx=None):
        """transport: an imt_transport handle (RCCL or IPC, include/imt.h) -- the root exchange then goes through the
        LIBRARY's communicators (imt_transport_all_gather: what a host without torch.distributed uses, e.g.
        examples/subtree_procs_demo.c) instead of `dist`.  The gather is enqueued on the BACKEND's stream, named
        explicitly; when that is the null stream (handle 0, which the C call reads as "the transport's own context's
        stream") the transport must have been created on the backend's context -- say so with transport_ctx."""
        if world & (world - 1):
            raise ValueError("world size must be a power of two")
        self.backend, self.depth, self.world, self.rank, self.dist = backend, depth, world, rank, dist
        self.via_host, self.transport = via_host, transport
        if transport is not None and backend is not None:
            if not backend.stream_ptr and transport_ctx is not None and transport_ctx is not backend.ctx:
                raise ValueError("the backend runs on the null stream and the transport was created on another context: its "
                                 "gather would run on that context's stream, unordered with root_after() / lift() -- create "
                                 "the transport on backend.ctx or run the backend on a stream of its own")
            backend.transport = transport
        self.k = world.bit_length() - 1
        self.sub_height = depth - self.k
        self.pending = None
        self.roots_prev = None if backend is None else torch.stack([backend.initial_root()] * world)
        self.global_root = None

    def owner(self, value):
        return int(value) % self.world

    def leaf_base(self):
        """first global leaf index of this rank's subtree"""
        return self.rank << self.sub_height

    def step(self, vals):
        slot = self.backend.insert(vals)        # the backend refuses values of another rank (all of them checked)
        done = self._finish(1) if self.pending is not None else None
        self.pending = slot
        return done

    def flush(self):
        if self.pending is None:
            return None
        done = self._finish(0)
        self.pending = None
        return done

    def non_membership_witness(self, vals):
        """BASELINE config 3 in the sharded layout: the depth-`depth` non-membership witness of values this rank owns,
        against `global_root` (call after flush(): no insertion pending)"""
        if self.pending is not None:
            raise RuntimeError("flush() first: a batch is still waiting for its root exchange")
        return self.backend.non_membership_witness(vals, self.roots_prev)

    def gather_roots(self, mine):
        """[world, 32]: every rank's subtree root -- the one collective of the path"""
        if self.transport is not None and self.world > 1:      # the library's own collective, on the backend's stream
            be = self.backend
            # zeros, not empty: if a GPU-side wait gives up the copies behind it are skipped, and what is read then must
            # not be whatever the allocator left there (backend.sync() reports the timeout; so does the next gather)
            out = torch.zeros((self.world, 32), dtype=torch.uint8, device=mine.device)
            src = mine.contiguous()
            rc = be.imt.lib.imt_transport_all_gather(self.transport, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                     32, ctypes.c_void_p(be.stream_ptr) if be.stream_ptr else None)
            if rc:
                raise be.imt.ImtError(rc, be.imt.lib.imt_transport_last_error(self.transport).decode())
            return out
        if self.dist is None:
            return mine.reshape(1, 32).clone()
        if self.via_host:       # gloo rehearsal: through host memory
            m = mine.cpu()
            parts = [torch.empty_like(m) for _ in range(self.world)]
            self.dist.all_gather(parts, m)
            return torch.stack(parts)
        out = torch.empty((self.world, 32), dtype=torch.uint8, device=mine.device)
        self.dist.all_gather_into_tensor(out.view(-1), mine.contiguous().view(-1))
        return out

    def _finish(self, lag):
        roots_after = self.gather_roots(self.backend.root_after(lag))
        self.backend.lift(self.pending, self.roots_prev, roots_after)
        self.global_root = self.backend.combine(roots_after)
        self.roots_prev = roots_after
        return self.backend.outputs(self.pending)


class ReplicatedIndexedTree:
    """ONE indexed tree (the reference's single sorted list, bit-exact at any world size) on `world`
    GPUs: every rank keeps a replica and runs the same hash-free preparation; the hashing of every
    level is split by slot range, and the ranks all-gather the level's node versions (E x 32 bytes,
    RCCL) before the next level.  Each rank returns the witnesses of its own share of the insertions.

    Uses the imt_itree_batch_* entry points; `via_host=True` routes the collectives through host
    memory (gloo rehearsal on a box without one GPU per rank)."""

    def __init__(self, imt, ctx, tree, world=1, rank=0, dist=None, via_host=False):
        import ctypes
        self.imt, self.ctx, self.tree, self.world, self.rank, self.dist = imt, ctx, tree, world, rank, dist
        self.via_host, self.ct = via_host, ctypes
        self.lib = imt.lib
        self.stream = torch.cuda.current_stream()
        ctx.set_stream(self.stream.cuda_stream)
        self.device = torch.device("cuda", torch.cuda.current_device())

    def _p(self, t):
        return self.ct.c_void_p(t.data_ptr())

    def _gather_rows(self, t, b, c):
        """rows [b, b+c) of t were computed here; afterwards every rank has all rows"""
        if self.world == 1:
            return
        if not self.via_host:
            self.dist.all_gather_into_tensor(t.view(-1), t[b:b + c].reshape(-1))
            return
        mine = t[b:b + c].cpu()
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        t.copy_(torch.cat(parts).to(t.device))

    def _bcast(self, t, src):
        if self.world == 1:
            return
        if not self.via_host:
            self.dist.broadcast(t, src)
            return
        h = t.cpu()
        self.dist.broadcast(h, src)
        t.copy_(h.to(t.device))

    def insert_batch(self, vals, proofs=True, fmt=0):
        """vals: uint8 [n, 32] (numpy or torch), the same on every rank, n a multiple of world.
        Returns a dict of torch tensors for this rank's insertions [rank*n/world, (rank+1)*n/world)."""
        ct, lib, imt = self.ct, self.lib, self.imt
        F = imt._ffi
        if not torch.is_tensor(vals):
            vals = torch.from_numpy(np.ascontiguousarray(vals))
        vals = vals.to(self.device)
        n = vals.shape[0]
        if n % self.world:
            raise ValueError("batch size must be a multiple of the world size")
        ev, l0 = ct.c_uint32(), ct.c_uint32()
        rc = lib.imt_itree_batch_begin(self.tree.h, self._p(vals), n, F.DEVICE_PTRS | fmt, ct.byref(ev), ct.byref(l0))
        if rc == F.ERR["VALUE"]:
            raise ValueError(lib.imt_last_error(self.ctx.h).decode())
        self.ctx._check(rc)
        E, L0, depth = ev.value, l0.value, self.tree.depth
        u8 = dict(dtype=torch.uint8, device=self.device)
        val = torch.empty((L0 + 1, E, 32), **u8)
        kb, kc = self.rank * (E // self.world), E // self.world
        self.ctx._check(lib.imt_itree_batch_leaves(self.tree.h, self._p(val[0]), kb, kc))
        self._gather_rows(val[0], kb, kc)
        for l in range(L0):
            self.ctx._check(lib.imt_itree_batch_level(self.tree.h, l, self._p(val[l]), self._p(val[l + 1]), kb, kc))
            self._gather_rows(val[l + 1], kb, kc)
        roots = torch.empty((E, 32), **u8)
        top_path = torch.zeros((depth - L0 + 1, 32), **u8)
        self.ctx._check(lib.imt_itree_batch_top(self.tree.h, self._p(val[L0]), kb, kc, self._p(roots), self._p(top_path)))
        self._gather_rows(roots, kb, kc)
        self._bcast(top_path, self.world - 1)          # the last event lies in the last rank's range
        ib, ic = self.rank * (n // self.world), n // self.world
        out = dict(low_index=torch.empty(ic, dtype=torch.int64, device=self.device), is_largest=torch.empty(ic, **u8),
                   low_leaf=torch.empty((ic, 3, 32), **u8), new_leaf=torch.empty((ic, 3, 32), **u8),
                   old_root=torch.empty((ic, 32), **u8), interim_root=torch.empty((ic, 32), **u8),
                   new_root=torch.empty((ic, 32), **u8))
        if proofs:
            out["low_sib"] = torch.empty((depth, ic, 32), **u8)
            out["new_sib"] = torch.empty((depth, ic, 32), **u8)
        st = F.InsertOut(**{k: v.data_ptr() for k, v in out.items()})
        ptrs = (ct.c_void_p * (L0 + 1))(*[val[l].data_ptr() for l in range(L0 + 1)])
        self.ctx._check(lib.imt_itree_batch_extract(self.tree.h, ptrs, self._p(roots), ib, ic, ct.byref(st),
                                                    F.DEVICE_PTRS | fmt))
        self.ctx._check(lib.imt_itree_batch_end(self.tree.h, ptrs, self._p(top_path)))
        self.ctx.sync()
        out["first_insertion"] = ib
        return out
