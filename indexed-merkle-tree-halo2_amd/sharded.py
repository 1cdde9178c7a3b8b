"""Multi-GPU: one depth-D tree as `world` value-partitioned indexed subtrees, one per GPU.

Rank g owns the values with v % world == g and the leaf-index range
[g * 2^(D-k), (g+1) * 2^(D-k)) of the depth-D tree (k = log2 world) as an indexed subtree of
height D-k with its own {0,0,0} sentinel.  Insertions never cross ranks; the only exchange
is an all-gather of the `world` subtree roots (32 bytes each; RCCL over xGMI with the "nccl"
backend), after which every rank hashes the top k levels itself.  Non-membership of v is
decided by the one subtree v belongs to, so the partition keeps the indexed tree's guarantee.

With world == 1 this is exactly the reference's single tree (src/indexed_merkle_tree.rs
:632-671, :715-735).  For world > 1 the root commits to `world` sorted lists instead of one;
the reference has no multi-device form to compare with, so parity for it is defined against the
CPU oracle building the same subtrees and combining their roots (tests/test_sharded_gloo.py).

The compute backend is pluggable so the collective logic can be exercised on CPU with gloo:
`GpuBackend` (the product: libimt_hip.so) or, in tests only, an oracle-backed stand-in.
"""
import numpy as np
import torch


class GpuBackend:
    """The product path: one imt context + one imt_itree on this rank's GPU."""

    def __init__(self, imt, device_index, sub_height, capacity):
        self.imt = imt
        self.ctx = imt.Context(device_index)
        self.tree = imt.IndexedTree(self.ctx, sub_height, capacity)
        self.device = torch.device("cuda", device_index)

    def insert_batch(self, vals, proofs=True):
        return self.tree.insert_batch(vals, proofs=proofs)

    def root_bytes(self):
        return self.imt.to_bytes(self.tree.root())

    def combine(self, roots, sub_height, depth):
        return self.ctx.combine_subtree_roots(roots, sub_height, depth)


class ShardedIndexedTree:
    def __init__(self, backend, depth, world=1, rank=0, dist=None):
        if world & (world - 1):
            raise ValueError("world size must be a power of two")
        self.backend, self.depth, self.world, self.rank, self.dist = backend, depth, world, rank, dist
        self.k = world.bit_length() - 1
        self.sub_height = depth - self.k

    def owner(self, value):
        return int(value) % self.world

    def leaf_base(self):
        """first global leaf index of this rank's subtree"""
        return self.rank << self.sub_height

    def insert_batch(self, vals, proofs=True):
        """vals: ints owned by this rank (v % world == rank).  Returns the backend's witness dict;
        indices in it are local to the subtree (add leaf_base() for global positions)."""
        for v in vals[:16]:
            if self.owner(v) != self.rank:
                raise ValueError(f"value {v} belongs to rank {self.owner(v)}, not {self.rank}")
        return self.backend.insert_batch(vals, proofs=proofs)

    def gather_roots(self):
        """[world, 32] uint8: every rank's subtree root (the one collective of the path)."""
        mine = torch.from_numpy(np.ascontiguousarray(self.backend.root_bytes()))
        if self.world == 1:
            return mine.reshape(1, 32).numpy()
        mine = mine.to(self.backend.device)
        parts = [torch.empty(32, dtype=torch.uint8, device=self.backend.device) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        return torch.stack(parts).cpu().numpy()

    def global_root(self):
        roots = self.gather_roots()
        return self.backend.combine(roots, self.sub_height, self.depth)

    def top_proof(self, roots):
        """the k siblings that extend a subtree-root proof to the global root, for this rank"""
        level = [r for r in roots]
        sibs, idx = [], self.rank
        while len(level) > 1:
            sibs.append(level[idx ^ 1])
            level = [self.backend.combine(np.stack([level[2 * i], level[2 * i + 1]]), 0, 1)
                     for i in range(len(level) // 2)]
            idx >>= 1
        return sibs


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
