"""GPU, two processes on one MI355X (gloo collectives through host memory): the single-list
multi-GPU mode, ReplicatedIndexedTree, end to end.  Each rank returns the witnesses of its half of
every batch; together they must equal the one-process imt_itree_insert_batch results."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu
DEPTH, N_BATCH, BATCHES = 32, 512, 3


def _vals():
    import oracle_lib
    return oracle_lib.synth_values(N_BATCH * BATCHES, 0x494D5409)


def _load_sharded():
    import importlib.util
    spec = importlib.util.spec_from_file_location("imt_sharded", os.path.join(ROOT, "indexed-merkle-tree-halo2_amd",
                                                                               "sharded.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import imt_amd
    import oracle_lib
    ctx = imt_amd.Context(0)
    tree = imt_amd.IndexedTree(ctx, DEPTH, 4096)
    rep = _load_sharded().ReplicatedIndexedTree(imt_amd, ctx, tree, world, rank, dist, via_host=True)
    vals = _vals()
    res = []
    for b in range(BATCHES):
        chunk = oracle_lib.ints_to_arr(vals[b * N_BATCH:(b + 1) * N_BATCH])
        out = rep.insert_batch(chunk)
        res.append({k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    q.put((rank, res, tree.root(), tree.size))
    dist.barrier()
    dist.destroy_process_group()


def test_single_list_on_two_ranks(imt, ctx):
    world = 2
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = 29700 + (os.getpid() % 1000)
    procs = [mpctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import oracle_lib
    ref = imt.IndexedTree(ctx, DEPTH, 4096)
    vals = _vals()
    for b in range(BATCHES):
        want = ref.insert_batch(vals[b * N_BATCH:(b + 1) * N_BATCH])
        half = N_BATCH // world
        for rank, res, _, _ in got:
            r = res[b]
            assert r["first_insertion"] == rank * half
            sl = slice(rank * half, (rank + 1) * half)
            for k in ("low_index", "is_largest", "low_leaf", "new_leaf", "old_root", "interim_root", "new_root"):
                assert (r[k].astype(want[k].dtype) == want[k][sl]).all(), (b, rank, k)
            for k in ("low_sib", "new_sib"):
                assert (r[k] == want[k][:, sl]).all(), (b, rank, k)
    for rank, _, root, size in got:                      # both replicas hold the same, reference-equal tree
        assert root == ref.root() and size == ref.size
