"""Multi-GPU, the reference's data structure: ONE indexed tree -- one sorted list, update_idx_leaf's sequential
semantics (src/indexed_merkle_tree.rs:632-660), insertion i at leaf `size + i` (:715) -- on `world` GPUs, bit-exact
with one GPU at any world size.

Everything that makes it correct -- the systolic schedule, its streams and events, the all-gather of each slice's
per-level write-backs (RCCL / IPC peer copies / in-process copies) -- lives behind the C ABI: imt_sliced_step,
imt_sliced_wait, imt_sliced_flush (include/imt.h; csrc/imt_sliced_sched.hpp is the schedule).  This module is a caller:
it owns the contexts, the replicas and rotating witness buffers (torch tensors), nothing else."""
import ctypes

import torch

FIELDS = ("low_index", "low_leaf", "is_largest", "old_root", "interim_root", "new_root", "new_leaf", "low_sib", "new_sib")


def local_transport(imt):
    h = ctypes.c_void_p()
    assert imt.lib.imt_transport_local_create(ctypes.byref(h)) == 0
    return h


def rccl_transport(imt, ctx, dist, world, rank, n_comms=4, device=None):
    """ncclAllGather inside the library on its own communicators; torch.distributed only carries the unique ids.
    `ctx` must stay alive until the transport has been destroyed (imt.h)."""
    ids = torch.zeros(n_comms * imt._ffi.RCCL_UNIQUE_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        for i in range(n_comms):
            ctx._check(imt.lib.imt_rccl_get_unique_id(ctypes.c_void_p(ids.data_ptr() + i * imt._ffi.RCCL_UNIQUE_ID_BYTES)))
    if world > 1:
        ids = ids.to(device) if device is not None else ids
        dist.broadcast(ids, src=0)
        ids = ids.cpu()
    h = ctypes.c_void_p()
    ctx._check(imt.lib.imt_transport_rccl_create(ctx.h, ctypes.c_void_p(ids.data_ptr()), n_comms, world, rank, ctypes.byref(h)))
    return h


def ipc_transport(imt, ctx, dist, world, rank, depth, batch, lag=None, device=None):
    """direct peer copies between the processes of one node (HIP IPC handles); torch.distributed carries the handle blobs.
    `ctx` must stay alive until the transport has been destroyed (imt.h)."""
    nb = int(imt.lib.imt_transport_ipc_blob_bytes())
    mine = torch.zeros(nb, dtype=torch.uint8)
    h = ctypes.c_void_p()
    ctx._check(imt.lib.imt_transport_ipc_create(ctx.h, world, rank, depth, batch, lag or 0, ctypes.byref(h), ctypes.c_void_p(mine.data_ptr())))
    allb = torch.zeros(world * nb, dtype=torch.uint8, device=device or "cpu")
    dist.all_gather_into_tensor(allb, mine.to(allb.device))
    allb = allb.cpu()
    ctx._check(imt.lib.imt_transport_ipc_connect(h, ctypes.c_void_p(allb.data_ptr())))
    return h


class SlicedTree:
    """The ranks of this process (one, or all `world` of them with the local transport) of one sliced tree.
    step(vals) starts a round (vals = the WHOLE step, world x n values, identical on every rank); outputs(R, k) are local
    rank k's witnesses of ITS slice of round R, valid after wait(R, k) or flush()."""

    def __init__(self, imt, device_index, depth, capacity, batch, world, first_rank=0, n_local=1, transport=None, lag=None,
                 nbuf=5, fmt=0, item_major=False):
        self.imt, self.F, self.lib = imt, imt._ffi, imt.lib
        self.depth, self.batch, self.world, self.first_rank, self.n_local, self.fmt = depth, batch, world, first_rank, n_local, fmt
        self.item_major = item_major         # sibling rows [n][depth] (the reference's per-proof Vec<F>) instead of [depth][n]
        if item_major:
            self.fmt |= imt._ffi.SIB_ITEM_MAJOR
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(device_index)
        self.ctxs = [imt.Context(device_index) for _ in range(n_local)]
        self.trees = [imt.IndexedTree(c, depth, capacity) for c in self.ctxs]
        self.tp = transport if transport is not None else local_transport(imt)
        u8 = dict(dtype=torch.uint8, device=self.device)
        mk = lambda: dict(low_index=torch.empty(batch, dtype=torch.int64, device=self.device),
                          low_leaf=torch.empty((batch, 3, 32), **u8), is_largest=torch.empty(batch, **u8),
                          old_root=torch.empty((batch, 32), **u8), interim_root=torch.empty((batch, 32), **u8),
                          new_root=torch.empty((batch, 32), **u8), new_leaf=torch.empty((batch, 3, 32), **u8),
                          low_sib=torch.empty((batch, depth, 32) if item_major else (depth, batch, 32), **u8),
                          new_sib=torch.empty((batch, depth, 32) if item_major else (depth, batch, 32), **u8))
        self.sets = [[mk() for _ in range(n_local)] for _ in range(nbuf)]       # [slot][local rank]
        self.rounds = []                     # per round: (n, size_before)
        arr = (ctypes.c_void_p * n_local)(*[t.h for t in self.trees])
        self.h = ctypes.c_void_p()
        rc = self.lib.imt_sliced_create(arr, n_local, world, first_rank, self.tp, batch, lag or 0, ctypes.byref(self.h))
        if rc:
            raise imt.ImtError(rc, self.lib.imt_last_error(self.ctxs[0].h).decode())

    def _check(self, rc):
        if rc == self.F.ERR["VALUE"]:
            raise ValueError(self.lib.imt_sliced_last_error(self.h).decode())
        if rc:
            raise self.imt.ImtError(rc, self.lib.imt_sliced_last_error(self.h).decode())

    def step(self, vals, flags=0):
        n = vals.shape[0] // self.world
        if vals.shape[0] != n * self.world or not 0 < n <= self.batch:
            raise ValueError(f"a step is world x n values, 0 < n <= batch = {self.batch} (a shorter step has shorter slices)")
        slot = len(self.rounds) % len(self.sets)
        outs = (self.F.InsertOut * self.n_local)(*[self.F.InsertOut(**{k: t.data_ptr() for k, t in s.items()}) for s in self.sets[slot]])
        R = ctypes.c_uint64()
        size_before = self.size()
        self._check(self.lib.imt_sliced_step(self.h, ctypes.c_void_p(vals.data_ptr()), n, outs, self.fmt | flags, ctypes.byref(R)))
        self.rounds.append((n, size_before))
        return int(R.value)

    def wait(self, R, k=0):
        self._check(self.lib.imt_sliced_wait(self.h, k, R))

    def flush(self):
        self._check(self.lib.imt_sliced_flush(self.h))

    def size(self):
        return int(self.lib.imt_itree_size(self.trees[0].h))

    def outputs(self, R, k=0):
        """rows [0, n) of every field (n = the round's slice length); sibling arrays are [depth, n, 32] (item_major:
        [n, depth, 32])"""
        n, size_before = self.rounds[R]
        d = dict(self.sets[R % len(self.sets)][k])
        if n != self.batch:
            for f, t in d.items():
                if f.endswith("_sib") and not self.item_major:
                    d[f] = t.view(-1)[:self.depth * n * 32].view(self.depth, n, 32)
                else:
                    d[f] = t[:n]
        d["first_insertion"] = size_before + (self.first_rank + k) * n      # = first new leaf index
        return d

    def info(self):
        o = self.F.SlicedInfo()
        self._check(self.lib.imt_sliced_get_info(self.h, ctypes.byref(o)))
        d = {f: getattr(o, f) for f, _ in o._fields_}
        d["queue_map"] = [[int(x) for x in row] for row in o.queue_map]       # [round / collective / apply stream][slot]
        d["placement"] = self.F.PLACEMENT.get(d["placement"], str(d["placement"]))
        return d

    def dump(self):
        """where the world stands (imt_sliced_dump): what a watchdog prints"""
        buf = ctypes.create_string_buffer(1 << 14)
        self.lib.imt_sliced_dump(self.h, buf, len(buf))
        return buf.value.decode(errors="replace")

    def set_option(self, option, value):
        self._check(self.lib.imt_sliced_set_option(self.h, option, value))

    def close(self, destroy_transport=True):
        if self.h:
            self.lib.imt_sliced_destroy(self.h)
            self.h = None
            if destroy_transport:
                self.lib.imt_transport_destroy(self.tp)
            for t in self.trees:
                t.close()
            for c in self.ctxs:
                c.close()
