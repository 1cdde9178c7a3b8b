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
