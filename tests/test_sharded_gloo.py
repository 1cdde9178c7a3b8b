"""CPU, world_size 2, gloo: the sharded driver (indexed-merkle-tree-halo2_amd/sharded.py) with an
oracle-backed stand-in for the GPU backend (tests/sharded_ref.py) -- the lagged root exchange, the
before/after bookkeeping of the lift and the value partition are the code under test; the hashing
stand-in is the checker itself.  Expectation: a dense rebuild of the whole depth-6 tree after every event
of the global order.  The same driver with the real GpuBackend: tests/test_gpu_sharded_procs.py."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DEPTH = 6          # small enough for a dense rebuild of the whole tree after every event
N_PER_STEP, STEPS = 5, 3


def _values(rank, world):
    import oracle_lib
    raw = oracle_lib.synth_values(N_PER_STEP * STEPS * world * 4, 0x494D5404)
    return [v for v in raw if v % world == rank][:N_PER_STEP * STEPS]


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
    from sharded_ref import OracleBackend
    orc = oracle_lib.load()
    sharded = _load_sharded()
    tree = sharded.ShardedIndexedTree(OracleBackend(orc, DEPTH, world, rank, 32), DEPTH, world, rank, dist, via_host=True)
    vals = _values(rank, world)
    finished, roots = [], []
    for st in range(STEPS):
        done = tree.step(vals[st * N_PER_STEP:(st + 1) * N_PER_STEP])     # returns the PREVIOUS step, lifted
        assert (done is None) == (st == 0)
        if done is not None:
            finished.append(done)
            roots.append(bytes(tree.global_root.numpy()))
    finished.append(tree.flush())
    roots.append(bytes(tree.global_root.numpy()))
    assert tree.flush() is None
    try:
        tree.step([vals[0] + 1] * N_PER_STEP)       # values owned by the other rank are refused
        refused = False
    except ValueError:
        refused = True
    # non-membership at full depth against the global root (BASELINE config 3 in the sharded layout): values of this
    # rank's residue that were never inserted; a pending batch, a foreign value and a stored value are refused
    cand = [v for v in oracle_lib.synth_values(400, 0x494D5405) if v % world == rank and v not in vals][:6]
    groot = int.from_bytes(bytes(tree.global_root.numpy()), "little")
    for v, w in zip(cand, tree.non_membership_witness(cand)):
        helper = np.zeros((DEPTH, 32), np.uint8)
        helper[:, 0] = [1 - ((w["low"] >> l) & 1) for l in range(DEPTH)]              # 1 = left child (src/utils.rs:79)
        fail, root_out = orc.verify_non_inclusion(groot, w["low_leaf"], w["proof"], helper, v, w["largest"])
        assert fail == 0 and root_out == groot and (w["low"] >> (DEPTH - 1)) == rank
    for bad in ([vals[0]], [cand[0] + 1]):
        try:
            tree.non_membership_witness(bad)
            refused = False
        except ValueError:
            pass
    tree.step([v for v in oracle_lib.synth_values(400, 0x494D5406) if v % world == rank and v not in vals][:N_PER_STEP])
    try:
        tree.non_membership_witness(cand[:1])
        refused = False
    except RuntimeError:
        tree.flush()
    q.put((rank, finished, roots, refused))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_two_ranks_gloo(oracle):
    from sharded_ref import dense_global_replay
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    steps = [[_values(r, world)[st * N_PER_STEP:(st + 1) * N_PER_STEP] for r in range(world)] for st in range(STEPS)]
    want, final_root = dense_global_replay(oracle, DEPTH, world, steps)
    for rank, finished, roots, refused in res:
        assert refused
        assert len(finished) == STEPS
        for st in range(STEPS):
            for got, exp in zip(finished[st], want[st][rank]):
                assert got["low"] == exp["low"] and got["new_index"] == exp["new_index"]
                assert [int.from_bytes(got["low_leaf"][j].tobytes(), "little") for j in range(3)] == exp["low_leaf"]
                assert got["largest"] == exp["largest"]
                for key in ("old_root", "interim_root", "new_root"):
                    assert got[key] == exp[key], (st, rank, key)
                assert (got["low_proof"] == exp["low_proof"]).all() and (got["new_proof"] == exp["new_proof"]).all()
            # the global root each rank computed after the step = the dense tree after the LAST rank's last event
            assert roots[st] == want[st][world - 1][-1]["new_root"].to_bytes(32, "little")
    assert res[0][2][-1] == final_root.to_bytes(32, "little")


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
