"""CPU, world_size 2, gloo: the sharded driver (indexed-merkle-tree-halo2_amd/sharded.py) with an
oracle-backed stand-in for the GPU backend -- the collective logic, the value partition and the
root combination are the code under test; the hashing stand-in is the checker itself."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DEPTH = 32
N_PER_RANK = 24


class OracleBackend:
    """test-only stand-in with the GpuBackend interface, on the CPU oracle"""

    def __init__(self, orc, sub_height, capacity):
        self.orc, self.h, self.sub_height = orc, orc.sparse_new(sub_height, capacity), sub_height
        self.device = torch.device("cpu")

    def insert_batch(self, vals, proofs=True):
        rows = [self.orc.sparse_insert(self.h, self.sub_height, v) for v in vals]
        assert all(r["rc"] == 0 for r in rows)
        return rows

    def root_bytes(self):
        return np.frombuffer(self.orc.sparse_root(self.h).to_bytes(32, "little"), dtype=np.uint8).copy()

    def combine(self, roots, sub_height, depth):
        level = [int.from_bytes(bytes(r), "little") for r in np.asarray(roots, dtype=np.uint8)]
        z = self.orc.zero_hashes(depth)
        h = sub_height
        while len(level) > 1:
            level = [self.orc.hash([level[2 * i], level[2 * i + 1]]) for i in range(len(level) // 2)]
            h += 1
        cur = level[0]
        for l in range(h, depth):
            cur = self.orc.hash([cur, int.from_bytes(z[l].tobytes(), "little")])
        return np.frombuffer(cur.to_bytes(32, "little"), dtype=np.uint8).copy()


def _values(rank, world):
    import oracle_lib
    raw = oracle_lib.synth_values(N_PER_RANK * world * 3, 0x494D5404)
    return [v for v in raw if v % world == rank][:N_PER_RANK]


def _load_sharded():
    import importlib.util
    path = os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "sharded.py")
    spec = importlib.util.spec_from_file_location("imt_sharded", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)     # sharded.py itself does not need the HIP library
    return mod


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    orc = oracle_lib.load()
    sharded = _load_sharded()
    k = world.bit_length() - 1
    tree = sharded.ShardedIndexedTree(OracleBackend(orc, DEPTH - k, 64), DEPTH, world, rank, dist)
    vals = _values(rank, world)
    roots_seen = []
    for a in range(0, N_PER_RANK, 8):           # three steps, a root exchange after each
        tree.insert_batch(vals[a:a + 8])
        roots_seen.append(bytes(tree.global_root()))
    gathered = tree.gather_roots()
    try:
        tree.insert_batch([vals[0] + 1])          # a value owned by the other rank is refused
        refused = False
    except ValueError:
        refused = True
    q.put((rank, roots_seen, gathered.tobytes(), refused, [bytes(s) for s in tree.top_proof(gathered)]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_two_ranks_gloo(oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process expectation: both subtrees built here, roots combined with the oracle
    hs = [oracle.sparse_new(DEPTH - 1, 64) for _ in range(world)]
    want = []
    for a in range(0, N_PER_RANK, 8):
        for r in range(world):
            for v in _values(r, world)[a:a + 8]:
                assert oracle.sparse_insert(hs[r], DEPTH - 1, v)["rc"] == 0
        sub = [oracle.sparse_root(h) for h in hs]
        top = oracle.hash(sub)
        want.append(top.to_bytes(32, "little"))
    for rank, roots_seen, gathered, refused, top_proof in res:
        assert roots_seen == want                     # every rank computes the same global root per step
        assert refused
        assert gathered == b"".join(oracle.sparse_root(h).to_bytes(32, "little") for h in hs)
        assert top_proof == [oracle.sparse_root(hs[rank ^ 1]).to_bytes(32, "little")]
    # the combined root is a depth-32 root: a leaf proof of rank 1's subtree + the top sibling verifies
    proof = oracle.sparse_proof(hs[1], DEPTH - 1, 3)
    full = np.concatenate([proof, np.frombuffer(oracle.sparse_root(hs[0]).to_bytes(32, "little"), np.uint8)[None]])
    leaf = oracle.hash([int.from_bytes(oracle.sparse_preimage(hs[1], 3)[j].tobytes(), "little") for j in range(3)])
    assert oracle.path_root(leaf, (1 << (DEPTH - 1)) + 3, full).to_bytes(32, "little") == want[-1]
    for h in hs:
        oracle.sparse_free(h)


def test_owner_partition():
    sharded = _load_sharded()
    t = sharded.ShardedIndexedTree(None, 32, 8, 5)
    assert t.sub_height == 29 and t.leaf_base() == 5 << 29
    assert [t.owner(v) for v in (5, 13, 8, 7)] == [5, 5, 0, 7]
    try:
        sharded.ShardedIndexedTree(None, 32, 3, 0)
        assert False
    except ValueError:
        pass
