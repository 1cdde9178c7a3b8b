"""GPU (MI355X): the multi-GPU single-list mode (indexed-merkle-tree-halo2_amd/sliced.py over imt_itree_slice_*)
against the ONE-GPU tree (imt_itree_insert_batch) on the same value sequence -- which is itself pinned to the CPU
oracle's sequential update_idx_leaf + rebuild (/root/reference/src/indexed_merkle_tree.rs:632-671, :715-735) by
tests/test_gpu_parity.py.  Bit-exact: every low index, preimage, flag, old / interim / new root and both proofs of
every insertion, and the stored tree of every replica.

 * LocalWorld: world = 1, 2, 4, 8 replicas in this process on the one GPU (device-to-device copies as the all-gather)
 * two processes over gloo (host-staged all-gather): tests the torch.distributed transport with real kernels
 * small depth where the tree fills up to its last level (l0 == depth), halo2curves' Montgomery format, refused values
"""
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

import oracle_lib  # noqa: E402
from test_sliced_schedule import load_sliced  # noqa: E402

pytestmark = pytest.mark.gpu
FIELDS = ("low_index", "low_leaf", "is_largest", "old_root", "interim_root", "new_root", "new_leaf", "low_sib", "new_sib")


def reference_run(imt, ctx, depth, cap, vals, per_call):
    """the one-GPU tree over the same sequence, `per_call` insertions per imt_itree_insert_batch"""
    ref = imt.IndexedTree(ctx, depth, cap)
    outs = [ref.insert_batch(vals[i:i + per_call]) for i in range(0, len(vals), per_call)]
    root = ref.root()
    ref.close()
    return outs, root


def check_round(want, got, lo, hi):
    """want: one-GPU results of the whole step; got: a rank's witnesses of insertions [lo, hi) of that step"""
    for k in FIELDS:
        g = got[k].cpu().numpy()
        w = np.asarray(want[k])
        w = w[:, lo:hi] if k in ("low_sib", "new_sib") else w[lo:hi]
        assert g.shape == w.shape, k
        assert (g == w).all(), k


@pytest.mark.parametrize("world,batch,rounds", [(1, 256, 6), (2, 256, 6), (4, 192, 6), (8, 64, 7), (2, 8192, 3)])
def test_local_world_equals_one_gpu_tree(imt, ctx, world, batch, rounds):
    sl = load_sliced()
    depth, cap = 32, 1 << 17
    vals = oracle_lib.synth_values(world * batch * rounds, 0x494D5431 + world)
    want, want_root = reference_run(imt, ctx, depth, cap, vals, world * batch)
    bes = [sl.SliceGpuBackend(imt, 0, depth, cap, batch) for _ in range(world)]
    w = sl.LocalWorld(bes)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    # witnesses must be read before their rotating buffer set is reused: check a round once it is `nbuf - 1` rounds old
    checked = 0
    for r in range(rounds):
        step = arr[r * world * batch:(r + 1) * world * batch]
        assert w.step([step] * world) == r
        if r == 2 and world != 4:
            w.flush()                    # run the schedule dry in the middle and go on (bench.py after its warm-up)
        while checked <= r - 3:
            for rk in w.ranks:
                rk.done_event(checked).synchronize()
                check_round(want[checked], rk.outputs(checked), rk.rank * batch, (rk.rank + 1) * batch)
            checked += 1
    w.flush()
    while checked < rounds:
        for rk in w.ranks:
            check_round(want[checked], rk.outputs(checked), rk.rank * batch, (rk.rank + 1) * batch)
        checked += 1
    for be in bes:                       # every replica is the same tree
        assert be.tree.root() == want_root
        assert be.size() == 1 + world * batch * rounds
    if world > 1:
        assert w.tp.collectives > 0
    # the replicas stay usable through the ordinary calls: a proof from replica 0 verifies against the root
    idx = np.array([1, 5, world * batch * rounds], dtype=np.uint64)
    sib = bes[0].tree.get_proof_batch(idx)
    leaves = bes[-1].tree.get_leaves(idx)
    h = ctx.hash3(leaves)
    roots = ctx.path_root(h, idx, sib, depth)
    assert all(int.from_bytes(bytes(x), "little") == want_root for x in roots)
    # non-membership (BASELINE config 3) on the replicated list needs no exchange: every replica holds the whole tree,
    # so the candidates are simply split between the ranks; each witness verifies against the common root
    cand = oracle_lib.synth_values(8 * world, 0x494D5499 + world)
    for g, be in enumerate(bes):
        mine = cand[g * 8:(g + 1) * 8]
        low, leaves, nsib, largest = be.tree.non_membership_witness(mine)
        fail = be.ctx.non_membership(imt.to_bytes(want_root), leaves, low, nsib, depth, imt.to_bytes(mine), largest)
        assert not fail.any()
    for be in bes:
        be.tree.close()
        be.ctx.close()


def test_local_world_at_bench_size(imt, ctx):
    """the size bench.py runs (BASELINE configs[1]: 2^16 insertions per GPU and step, depth 32): two replicas, three
    steps, the default lag -- every witness byte of both ranks against imt_itree_insert_batch over the same 2^17 values
    per step, compared on the GPU"""
    sl = load_sliced()
    depth, world, batch, rounds = 32, 2, 1 << 16, 3
    cap = 1 << 19
    import bench
    vals = torch.from_numpy(bench.synth_values(world * batch * rounds, 0, 1, 0x494D5491)).cuda()
    ref = imt.IndexedTree(ctx, depth, cap)
    u8 = dict(dtype=torch.uint8, device="cuda")
    gb = world * batch
    want = []
    F = imt._ffi
    for r in range(rounds):
        o = dict(low_index=torch.empty(gb, dtype=torch.int64, device="cuda"), low_leaf=torch.empty((gb, 3, 32), **u8),
                 is_largest=torch.empty(gb, **u8), old_root=torch.empty((gb, 32), **u8), interim_root=torch.empty((gb, 32), **u8),
                 new_root=torch.empty((gb, 32), **u8), new_leaf=torch.empty((gb, 3, 32), **u8),
                 low_sib=torch.empty((depth, gb, 32), **u8), new_sib=torch.empty((depth, gb, 32), **u8))
        st = F.InsertOut(**{k: t.data_ptr() for k, t in o.items()})
        import ctypes
        ctx._check(imt.lib.imt_itree_insert_batch(ref.h, ctypes.c_void_p(vals[r * gb:(r + 1) * gb].data_ptr()), gb, ctypes.byref(st),
                                                  F.DEVICE_PTRS))
        want.append(o)
    ctx.sync()
    bes = [sl.SliceGpuBackend(imt, 0, depth, cap, batch) for _ in range(world)]
    w = sl.LocalWorld(bes)
    assert w.sched.lag == 6
    for r in range(rounds):
        w.step([vals[r * gb:(r + 1) * gb]] * world)
    w.flush()
    for r in range(rounds):
        for rk in w.ranks:
            got = rk.outputs(r)
            lo, hi = rk.rank * batch, (rk.rank + 1) * batch
            for k in FIELDS:
                a = want[r][k][:, lo:hi] if k.endswith("_sib") else want[r][k][lo:hi]
                assert bool((got[k] == a).all()), (r, rk.rank, k)
    assert all(be.tree.root() == ref.root() for be in bes)
    ref.close()
    for be in bes:
        be.tree.close()
        be.ctx.close()


def test_config4_single_list_eight_slices_2pow22(imt, ctx):
    """BASELINE config 4's size on the reference's data structure: 2^22 insertions into ONE depth-32 tree, 8 slices of
    2^16 per step (8 replicas on this one GPU), 8 steps.  Size-independent properties at full size: every insertion's
    witnesses pass every insert_leaf constraint at depth 32 with global leaf indices (imt_insert_witness_batch), the
    roots chain insertion to insertion, slice to slice and step to step, all replicas end in the same root, and that
    root is the one-GPU tree's over the same 2^22 values."""
    import ctypes
    import bench
    sl = load_sliced()
    depth, world, batch, rounds = 32, 8, 1 << 16, 8
    cap = 1 << 23
    gb = world * batch
    vals = torch.from_numpy(bench.synth_values(gb * rounds, 0, 1, 0x494D5404)).cuda()
    bes = [sl.SliceGpuBackend(imt, 0, depth, cap, batch) for _ in range(world)]
    w = sl.LocalWorld(bes)
    F, lib = imt._ffi, imt.lib
    P_ = lambda x: ctypes.c_void_p(x.data_ptr())
    fail = torch.empty(batch, dtype=torch.uint8, device="cuda")
    prev_last = None
    checked = 0

    def check(r):
        nonlocal prev_last
        for rk in w.ranks:
            o = rk.outputs(r)
            first = o["first_insertion"]
            assert first == 1 + r * gb + rk.rank * batch
            new_index = torch.arange(first, first + batch, dtype=torch.int64, device="cuda")
            c = rk.be.ctx
            c._check(lib.imt_insert_witness_batch(c.h, P_(o["old_root"]), P_(o["low_leaf"]), P_(o["low_index"]), P_(o["low_sib"]),
                                                  P_(o["new_root"]), P_(o["new_leaf"]), P_(new_index), None, P_(o["new_sib"]),
                                                  P_(o["is_largest"]), depth, batch, P_(fail), None, F.DEVICE_PTRS))
            c.sync()
            assert int(fail.max()) == 0, (r, rk.rank)
            assert bool((o["old_root"][1:] == o["new_root"][:-1]).all())
            if prev_last is not None:
                assert bool((o["old_root"][0] == prev_last).all()), (r, rk.rank)      # slice to slice, step to step
            prev_last = o["new_root"][-1].clone()

    for r in range(rounds):
        w.step([vals[r * gb:(r + 1) * gb]] * world)
        while checked <= r - 3:
            for rk in w.ranks:
                rk.done_event(checked).synchronize()
            check(checked)
            checked += 1
    w.flush()
    while checked < rounds:
        check(checked)
        checked += 1
    roots = {be.tree.root() for be in bes}
    assert len(roots) == 1 and imt.to_int(prev_last.cpu().numpy()) in roots
    for be in bes:
        assert be.size() == 1 + gb * rounds
        be.tree.close()
        be.ctx.close()
    ref = imt.IndexedTree(ctx, depth, cap)
    for r in range(rounds):
        ctx._check(lib.imt_itree_insert_batch(ref.h, P_(vals[r * gb:(r + 1) * gb]), gb, None, F.DEVICE_PTRS | F.PIPELINE))
    ctx.sync()
    assert ref.root() in roots
    ref.close()


def test_local_world_fills_a_small_tree_to_its_last_level(imt, ctx):
    """depth 8, 4 replicas: l0 reaches the depth (no empty-subtree levels left), the root travels in the payload"""
    sl = load_sliced()
    depth, cap, world, batch = 8, 256, 4, 15
    rounds = 4                               # 1 + 240 leaves of 256
    vals = oracle_lib.synth_values(world * batch * rounds, 0x494D5441)
    want, want_root = reference_run(imt, ctx, depth, cap, vals, world * batch)
    bes = [sl.SliceGpuBackend(imt, 0, depth, cap, batch) for _ in range(world)]
    w = sl.LocalWorld(bes)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    for r in range(rounds):
        w.step([arr[r * world * batch:(r + 1) * world * batch]] * world)
    w.flush()
    for r in range(rounds):
        for rk in w.ranks:
            check_round(want[r], rk.outputs(r), rk.rank * batch, (rk.rank + 1) * batch)
    assert all(be.tree.root() == want_root for be in bes)
    # the oracle's sequential insertion agrees on the final root (depth 8 is cheap on the CPU)
    orc = oracle_lib.load()
    h = orc.sparse_new(depth, cap)
    for v in vals:
        assert orc.sparse_insert(h, depth, v)["rc"] == 0
    assert orc.sparse_root(h) == want_root
    orc.sparse_free(h)
    for be in bes:
        be.tree.close()
        be.ctx.close()


def test_local_world_ragged_steps(imt, ctx):
    """steps shorter than the buffers were sized for (the last, ragged step of a stream of insertions): slices of 64,
    10, 1 and 64 insertions per rank in consecutive steps"""
    sl = load_sliced()
    depth, cap, world, batch = 32, 1 << 10, 2, 64
    sizes = [64, 10, 1, 64, 7]
    vals = oracle_lib.synth_values(world * sum(sizes), 0x494D5481)
    bes = [sl.SliceGpuBackend(imt, 0, depth, cap, batch) for _ in range(world)]
    w = sl.LocalWorld(bes)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    ref = imt.IndexedTree(ctx, depth, cap)
    off = 0
    for r, n in enumerate(sizes):
        w.step([arr[off:off + world * n]] * world)
        w.flush()                        # witness sets rotate: read each round before the fifth after it is prepared
        want = ref.insert_batch(vals[off:off + world * n])
        for rk in w.ranks:
            got = rk.outputs(r)
            assert got["low_sib"].shape == (depth, n, 32) and got["new_root"].shape == (n, 32)
            check_round(want, got, rk.rank * n, (rk.rank + 1) * n)
        off += world * n
    with pytest.raises(ValueError):
        w.ranks[0]._start_round(arr[:world * batch + world])      # longer than the buffers
    with pytest.raises(ValueError):
        w.ranks[0]._start_round(arr[:3])                          # not a multiple of the world size
    assert all(be.tree.root() == ref.root() for be in bes)
    ref.close()
    for be in bes:
        be.tree.close()
        be.ctx.close()


def test_local_world_montgomery_format_and_refused_values(imt, ctx):
    sl = load_sliced()
    depth, cap, world, batch = 32, 1 << 12, 2, 128
    vals = oracle_lib.synth_values(world * batch * 2, 0x494D5451)
    F = imt._ffi
    bes = [sl.SliceGpuBackend(imt, 0, depth, cap, batch, fmt=F.FMT_MONT256) for _ in range(world)]
    w = sl.LocalWorld(bes)
    R = 1 << 256
    mont = torch.from_numpy(oracle_lib.ints_to_arr([v * R % oracle_lib.P for v in vals])).cuda()
    w.step([mont[:world * batch]] * world)
    # a step with a duplicate (of a stored value, in the second rank's slice) is refused by EVERY rank, nothing changes
    bad = mont[world * batch:].clone()
    bad[batch + 3] = mont[5]
    for rk in w.ranks:
        with pytest.raises(ValueError):
            rk._start_round(bad)
        assert rk.be.size() == 1 + world * batch
    w.step([mont[world * batch:]] * world)
    w.flush()
    ref = imt.IndexedTree(ctx, depth, cap)
    for r in range(2):
        want = ref.insert_batch(vals[r * world * batch:(r + 1) * world * batch])
        for rk in w.ranks:
            got = rk.outputs(r)
            lo, hi = rk.rank * batch, (rk.rank + 1) * batch
            for k in ("old_root", "interim_root", "new_root"):
                g = [int.from_bytes(bytes(x), "little") for x in got[k].cpu().numpy()]
                assert g == [int.from_bytes(bytes(x), "little") * R % oracle_lib.P for x in want[k][lo:hi]], k
            assert (got["low_index"].cpu().numpy() == want["low_index"][lo:hi]).all()
            g = [int.from_bytes(bytes(x), "little") for x in got["low_sib"].cpu().numpy()[:, 7]]
            assert g == [int.from_bytes(bytes(x), "little") * R % oracle_lib.P for x in want["low_sib"][:, lo + 7]]
    assert all(be.tree.root() == ref.root() for be in bes)
    ref.close()
    for be in bes:
        be.tree.close()
        be.ctx.close()


# ---------------------------------------------------------------- two processes, gloo
P_DEPTH, P_BATCH, P_ROUNDS = 32, 384, 5


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import imt_amd
    sl = load_sliced()
    be = sl.SliceGpuBackend(imt_amd, 0, P_DEPTH, 1 << 13, P_BATCH)
    tree = sl.SlicedIndexedTree(be, world, rank, sl.DistTransport(dist, via_host=True))
    vals = oracle_lib.synth_values(world * P_BATCH * P_ROUNDS, 0x494D5461)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    res = []
    for r in range(P_ROUNDS):
        tree.step(arr[r * world * P_BATCH:(r + 1) * world * P_BATCH])
        if r >= 3:                      # read a finished round while later ones are in flight
            tree.done_event(r - 3).synchronize()
            res.append({k: v.cpu().numpy().copy() for k, v in tree.outputs(r - 3).items() if torch.is_tensor(v)})
    tree.flush()
    for r in range(max(0, P_ROUNDS - 3), P_ROUNDS):
        res.append({k: v.cpu().numpy().copy() for k, v in tree.outputs(r).items() if torch.is_tensor(v)})
    tree.close()
    q.put((rank, res, be.tree.root(), tree.tp.collectives, tree.tp.bytes_moved))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_processes_over_gloo_equal_one_gpu_tree(imt, ctx, world):
    """one PROCESS per rank on the one GPU (2 and 4: the box admits six GPU processes), torch.distributed over gloo, the
    host-staged gather on its helper thread: every rank's witnesses and every replica's root equal the one-GPU tree"""
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [mpctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    vals = oracle_lib.synth_values(world * P_BATCH * P_ROUNDS, 0x494D5461)
    want, want_root = reference_run(imt, ctx, P_DEPTH, 1 << 13, vals, world * P_BATCH)
    for rank, res, root, ncoll, nbytes in got:
        assert root == want_root
        assert ncoll > 0 and nbytes > 0
        for r in range(P_ROUNDS):
            for k in FIELDS:
                w = np.asarray(want[r][k])
                w = w[:, rank * P_BATCH:(rank + 1) * P_BATCH] if k.endswith("_sib") else w[rank * P_BATCH:(rank + 1) * P_BATCH]
                assert (res[r][k] == w).all(), (rank, r, k)


def test_batches_after_slices_on_one_tree(imt, ctx):
    """a tree that has taken sliced steps goes on with ordinary pipelined batches (imt_itree_insert_batch with
    IMT_PIPELINE, no synchronisation in between): the batch is ordered behind the slice's last kernel"""
    import ctypes
    sl = load_sliced()
    depth, cap, batch = 32, 1 << 14, 2048
    vals = oracle_lib.synth_values(4 * batch, 0x494D5483)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    be = sl.SliceGpuBackend(imt, 0, depth, cap, batch)
    w = sl.LocalWorld([be])
    w.step([arr[:batch]])
    w.step([arr[batch:2 * batch]])
    w._run_ticks(w.ranks[0].starts[-1] + w.sched.round_ticks)       # everything issued, nothing waited for
    F = imt._ffi
    for k in (2, 3):
        be.ctx._check(imt.lib.imt_itree_insert_batch(be.tree.h, ctypes.c_void_p(arr[k * batch:(k + 1) * batch].data_ptr()), batch,
                                                     None, F.DEVICE_PTRS | F.PIPELINE))
    be.sync()
    ref = imt.IndexedTree(ctx, depth, cap)
    ref.insert_batch(vals)
    assert be.tree.root() == ref.root()
    ref.close()
    be.tree.close()
    be.ctx.close()


def test_slice_calls_refuse_bad_arguments(imt, ctx):
    """the C entry points directly: misaligned payloads / values, units out of order, a second preparation of too many
    slices, a placed tree -- documented codes, nothing reaches a kernel"""
    import ctypes
    F, lib = imt._ffi, imt.lib
    t = imt.IndexedTree(ctx, 32, 1 << 10)
    vals = torch.from_numpy(oracle_lib.ints_to_arr(oracle_lib.synth_values(64, 0x494D5471))).cuda()
    pay = torch.zeros(int(lib.imt_itree_slice_payload_bytes(16)) + 64, dtype=torch.uint8, device="cuda")
    sl = ctypes.c_int(-1)
    P_ = lambda x, off=0: ctypes.c_void_p(x.data_ptr() + off)
    assert lib.imt_itree_slice_prepare(t.h, P_(vals, 8), 0, 16, 0, None, F.DEVICE_PTRS, ctypes.byref(sl), None) == F.ERR["ARG"]
    assert lib.imt_itree_slice_prepare(t.h, P_(vals), 0, 16, 0, None, 0, ctypes.byref(sl), None) == F.ERR["ARG"]      # host pointers
    assert lib.imt_itree_slice_prepare(t.h, P_(vals), 0, 0, 16, None, F.DEVICE_PTRS, ctypes.byref(sl), None) == F.ERR["ARG"]
    assert lib.imt_itree_slice_prepare(t.h, P_(vals), 600, 16, 600, None, F.DEVICE_PTRS, ctypes.byref(sl), None) == F.ERR["FULL"]
    assert t.size == 1
    assert lib.imt_itree_slice_prepare(t.h, P_(vals), 0, 16, 0, None, F.DEVICE_PTRS, ctypes.byref(sl), None) == 0
    assert t.size == 17
    with pytest.raises(imt.ImtError):                                                         # a slice is open: batch calls wait
        t.insert_batch([12345])
    assert lib.imt_itree_slice_unit(t.h, sl.value, 1, P_(pay), None) == F.ERR["ARG"]          # unit 0 comes first
    assert lib.imt_itree_slice_unit(t.h, sl.value, 0, P_(pay, 8), None) == F.ERR["ARG"]       # misaligned payload
    assert lib.imt_itree_slice_unit(t.h, 7, 0, P_(pay), None) == F.ERR["ARG"]
    for q in range(33):
        assert lib.imt_itree_slice_unit(t.h, sl.value, q, P_(pay), None) == 0
    assert lib.imt_itree_slice_unit(t.h, sl.value, 33, P_(pay), None) == F.ERR["ARG"]         # the slice is closed
    assert lib.imt_itree_slice_apply(t.h, 1, 16, 5, P_(pay, 8), None) == F.ERR["ARG"]
    assert lib.imt_itree_slice_apply(t.h, 1, 16, 40, P_(pay), None) == F.ERR["RANGE"]
    assert lib.imt_itree_slice_apply(t.h, 1 << 10, 16, 5, P_(pay), None) == F.ERR["RANGE"]
    ctx.sync()
    ref = imt.IndexedTree(ctx, 32, 1 << 10)
    ref.insert_batch(oracle_lib.synth_values(64, 0x494D5471)[:16])
    assert t.root() == ref.root()          # one slice alone on the context's stream = an ordinary batch
    placed = imt.IndexedTree(ctx, 31, 1 << 10)
    placed.set_placement(32, 1)
    assert lib.imt_itree_slice_prepare(placed.h, P_(vals), 0, 16, 0, None, F.DEVICE_PTRS, ctypes.byref(sl), None) == F.ERR["ARG"]
    for x in (t, ref, placed):
        x.close()
