"""GPU (MI355X): the multi-GPU single-list mode BEHIND THE C ABI (imt_sliced_create / _step / _wait / _flush: the schedule,
its streams and events and the all-gather live in libimt_hip.so; indexed-merkle-tree-halo2_amd/sliced.py only owns buffers)

 * directly against the CPU oracle's sequential update_idx_leaf + rebuild
   (/root/reference/src/indexed_merkle_tree.rs:632-671, :715-735): every low index, flag, preimage, interim / new root and
   both proofs of every rank, world 2, 4 and 8, depth 32, a flush in the middle;
 * against the ONE-GPU tree (imt_itree_insert_batch, itself pinned to the oracle by tests/test_gpu_parity.py) on longer
   sequences: world = 1, 2, 4, 8 replicas in this process (the local transport: device-to-device copies as the
   all-gather), BASELINE config 2's and config 4's sizes;
 * two and four PROCESSES sharing the GPU over the IPC transport (peer copies through HIP IPC handles), one rank alone
   through the RCCL transport (ncclAllGather called by the library), a caller-supplied transport vtable;
 * small depth where the tree fills up to its last level (l0 == depth), halo2curves' Montgomery format, refused values,
   ordinary batches before and after slices, bad arguments.
"""
import ctypes
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

pytestmark = pytest.mark.gpu
FIELDS = ("low_index", "low_leaf", "is_largest", "old_root", "interim_root", "new_root", "new_leaf", "low_sib", "new_sib")


def load_sliced():
    import importlib.util
    spec = importlib.util.spec_from_file_location("imt_sliced", os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "sliced.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


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


@pytest.mark.parametrize("world,batch,rounds", [(2, 167, 6), (4, 84, 6), (8, 40, 7)])
def test_sliced_world_against_the_sequential_oracle(imt, world, batch, rounds):
    """>= 2000 insertions at depth 32, a flush in the middle: every rank's every witness against oracle.sparse_insert"""
    sl = load_sliced()
    depth, cap = 32, 1 << 12
    n_total = world * batch * rounds
    assert n_total >= 2000
    vals = oracle_lib.synth_values(n_total, 0x494D5439 + world)
    orc = oracle_lib.load()
    oh = orc.sparse_new(depth, cap)
    empty_root = orc.sparse_root(oh)
    rows = [orc.sparse_insert(oh, depth, v) for v in vals]
    assert all(r["rc"] == 0 for r in rows)
    t = sl.SlicedTree(imt, 0, depth, cap, batch, world, n_local=world, nbuf=rounds)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    gb = world * batch
    for r in range(rounds):
        assert t.step(arr[r * gb:(r + 1) * gb]) == r
        if r == rounds // 2:
            t.flush()
    t.flush()
    for r in range(rounds):
        for k in range(world):
            o = {f: v.cpu().numpy() for f, v in t.outputs(r, k).items() if torch.is_tensor(v)}
            base = r * gb + k * batch
            assert t.outputs(r, k)["first_insertion"] == 1 + base
            for j in range(batch):
                e = rows[base + j]
                assert imt.to_int(o["new_root"][j]) == e["new_root"] and imt.to_int(o["interim_root"][j]) == e["interim_root"], (r, k, j)
                assert imt.to_int(o["old_root"][j]) == (rows[base + j - 1]["new_root"] if base + j else empty_root)
                assert int(o["low_index"][j]) == e["low"] and int(o["is_largest"][j]) == e["largest"], (r, k, j)
                assert (o["low_sib"][:, j] == e["low_proof"]).all() and (o["new_sib"][:, j] == e["new_proof"]).all(), (r, k, j)
                assert (o["low_leaf"][j] == e["low_leaf"]).all(), (r, k, j)
    want_root = orc.sparse_root(oh)
    orc.sparse_free(oh)
    assert all(tr.root() == want_root for tr in t.trees)
    info = t.info()
    assert info["world"] == world and info["rounds"] == rounds and info["collectives"] > 0
    t.close()


@pytest.mark.parametrize("world,batch,rounds", [(1, 256, 6), (2, 256, 6), (4, 192, 6), (8, 64, 7), (2, 8192, 3)])
def test_local_world_equals_one_gpu_tree(imt, ctx, world, batch, rounds):
    sl = load_sliced()
    depth, cap = 32, 1 << 17
    vals = oracle_lib.synth_values(world * batch * rounds, 0x494D5431 + world)
    want, want_root = reference_run(imt, ctx, depth, cap, vals, world * batch)
    t = sl.SlicedTree(imt, 0, depth, cap, batch, world, n_local=world)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    # witnesses must be read before their rotating buffer set is reused: check a round once it is `nbuf - 2` rounds old
    checked = 0
    for r in range(rounds):
        step = arr[r * world * batch:(r + 1) * world * batch]
        assert t.step(step) == r
        if r == 2 and world != 4:
            t.flush()                    # run the schedule dry in the middle and go on (bench.py after its warm-up)
        while checked <= r - 3:
            for k in range(world):
                t.wait(checked, k)
                check_round(want[checked], t.outputs(checked, k), k * batch, (k + 1) * batch)
            checked += 1
    t.flush()
    while checked < rounds:
        for k in range(world):
            check_round(want[checked], t.outputs(checked, k), k * batch, (k + 1) * batch)
        checked += 1
    for tr in t.trees:                   # every replica is the same tree
        assert tr.root() == want_root
        assert tr.size == 1 + world * batch * rounds
    if world > 1:
        assert t.info()["collectives"] > 0 and t.info()["bytes_gathered"] > 0
    # the replicas stay usable through the ordinary calls: a proof from replica 0 verifies against the root
    idx = np.array([1, 5, world * batch * rounds], dtype=np.uint64)
    sib = t.trees[0].get_proof_batch(idx)
    leaves = t.trees[-1].get_leaves(idx)
    h = ctx.hash3(leaves)
    roots = ctx.path_root(h, idx, sib, depth)
    assert all(int.from_bytes(bytes(x), "little") == want_root for x in roots)
    # non-membership (BASELINE config 3) on the replicated list needs no exchange: every replica holds the whole tree,
    # so the candidates are simply split between the ranks; each witness verifies against the common root
    cand = oracle_lib.synth_values(8 * world, 0x494D5499 + world)
    for g, (tr, c) in enumerate(zip(t.trees, t.ctxs)):
        mine = cand[g * 8:(g + 1) * 8]
        low, leaves, nsib, largest = tr.non_membership_witness(mine)
        fail = c.non_membership(imt.to_bytes(want_root), leaves, low, nsib, depth, imt.to_bytes(mine), largest)
        assert not fail.any()
    t.close()


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
        ctx._check(imt.lib.imt_itree_insert_batch(ref.h, ctypes.c_void_p(vals[r * gb:(r + 1) * gb].data_ptr()), gb, ctypes.byref(st),
                                                  F.DEVICE_PTRS))
        want.append(o)
    ctx.sync()
    t = sl.SlicedTree(imt, 0, depth, cap, batch, world, n_local=world)
    assert t.info()["lag"] == 6
    for r in range(rounds):
        t.step(vals[r * gb:(r + 1) * gb])
    t.flush()
    for r in range(rounds):
        for k in range(world):
            got = t.outputs(r, k)
            lo, hi = k * batch, (k + 1) * batch
            for f in FIELDS:
                a = want[r][f][:, lo:hi] if f.endswith("_sib") else want[r][f][lo:hi]
                assert bool((got[f] == a).all()), (r, k, f)
    assert all(tr.root() == ref.root() for tr in t.trees)
    ref.close()
    t.close()


def test_config4_single_list_eight_slices_2pow22(imt, ctx):
    """BASELINE config 4's size on the reference's data structure: 2^22 insertions into ONE depth-32 tree, 8 slices of
    2^16 per step (8 replicas on this one GPU), 8 steps.  Size-independent properties at full size: every insertion's
    witnesses pass every insert_leaf constraint at depth 32 with global leaf indices (imt_insert_witness_batch), the
    roots chain insertion to insertion, slice to slice and step to step, all replicas end in the same root, and that
    root is the one-GPU tree's over the same 2^22 values.

    And, item by item at full size: every interim / new root, low index and flag of all 2^22 insertions
    equals the sequential CPU oracle's, through the committed per-step digests of its run
    (tests/golden/config4_oracle_digest.json, make_config4_digest.py: update_idx_leaf + rebuild,
    /root/reference/src/indexed_merkle_tree.rs:632-671, :715-735, about three core-hours)."""
    import hashlib
    import json
    import bench
    sl = load_sliced()
    depth, world, batch, rounds = 32, 8, 1 << 16, 8
    cap = 1 << 23
    gb = world * batch
    vals_h = bench.synth_values(gb * rounds, 0, 1, 0x494D5404)
    dg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config4_oracle_digest.json")))
    assert dg["n"] == gb * rounds and dg["steps"] == rounds and dg["depth"] == depth
    assert hashlib.sha256(vals_h.tobytes()).hexdigest() == dg["sha256_values"], "the value generator changed: regenerate the digest"
    vals = torch.from_numpy(vals_h).cuda()
    t = sl.SlicedTree(imt, 0, depth, cap, batch, world, n_local=world)
    F, lib = imt._ffi, imt.lib
    P_ = lambda x: ctypes.c_void_p(x.data_ptr())
    fail = torch.empty(batch, dtype=torch.uint8, device="cuda")
    prev_last = None
    checked = 0

    def check(r):
        nonlocal prev_last
        # the oracle's digests of this step: the 8 slices in insertion order
        want = dg["per_step"][r]
        for field, key, cast in (("interim_root", "sha256_interim_roots", None), ("new_root", "sha256_new_roots", None),
                                 ("low_index", "sha256_low_index", "<u8"), ("is_largest", "sha256_is_largest", None)):
            h = hashlib.sha256()
            for k in range(world):
                a = t.outputs(r, k)[field].cpu().numpy()
                h.update((a.astype(cast) if cast else a).tobytes())
            assert h.hexdigest() == want[key], (r, field)
        assert imt.to_int(t.outputs(r, world - 1)["new_root"][-1].cpu().numpy()) == int(want["root_after"]), r
        for k in range(world):
            o = t.outputs(r, k)
            first = o["first_insertion"]
            assert first == 1 + r * gb + k * batch
            new_index = torch.arange(first, first + batch, dtype=torch.int64, device="cuda")
            c = t.ctxs[k]
            c._check(lib.imt_insert_witness_batch(c.h, P_(o["old_root"]), P_(o["low_leaf"]), P_(o["low_index"]), P_(o["low_sib"]),
                                                  P_(o["new_root"]), P_(o["new_leaf"]), P_(new_index), None, P_(o["new_sib"]),
                                                  P_(o["is_largest"]), depth, batch, P_(fail), None, F.DEVICE_PTRS))
            c.sync()
            assert int(fail.max()) == 0, (r, k)
            assert bool((o["old_root"][1:] == o["new_root"][:-1]).all())
            if prev_last is not None:
                assert bool((o["old_root"][0] == prev_last).all()), (r, k)      # slice to slice, step to step
            prev_last = o["new_root"][-1].clone()

    for r in range(rounds):
        t.step(vals[r * gb:(r + 1) * gb])
        while checked <= r - 3:
            for k in range(world):
                t.wait(checked, k)
            check(checked)
            checked += 1
    t.flush()
    while checked < rounds:
        check(checked)
        checked += 1
    roots = {tr.root() for tr in t.trees}
    assert len(roots) == 1 and imt.to_int(prev_last.cpu().numpy()) in roots
    assert roots == {int(dg["final_root"])}
    assert all(tr.size == 1 + gb * rounds for tr in t.trees)
    for i, hx in dg["sha256_final_proofs"].items():
        assert hashlib.sha256(t.trees[0].get_proof_batch([int(i)], item_major=True).tobytes()).hexdigest() == hx, i
    t.close()
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
    t = sl.SlicedTree(imt, 0, depth, cap, batch, world, n_local=world)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    for r in range(rounds):
        t.step(arr[r * world * batch:(r + 1) * world * batch])
    t.flush()
    for r in range(rounds):
        for k in range(world):
            check_round(want[r], t.outputs(r, k), k * batch, (k + 1) * batch)
    assert all(tr.root() == want_root for tr in t.trees)
    # the oracle's sequential insertion agrees on the final root (depth 8 is cheap on the CPU)
    orc = oracle_lib.load()
    h = orc.sparse_new(depth, cap)
    for v in vals:
        assert orc.sparse_insert(h, depth, v)["rc"] == 0
    assert orc.sparse_root(h) == want_root
    orc.sparse_free(h)
    t.close()


def test_local_world_ragged_steps(imt, ctx):
    """steps shorter than the buffers were sized for (the last, ragged step of a stream of insertions): slices of 64,
    10, 1 and 64 insertions per rank in consecutive steps"""
    sl = load_sliced()
    depth, cap, world, batch = 32, 1 << 10, 2, 64
    sizes = [64, 10, 1, 64, 7]
    vals = oracle_lib.synth_values(world * sum(sizes), 0x494D5481)
    t = sl.SlicedTree(imt, 0, depth, cap, batch, world, n_local=world)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    ref = imt.IndexedTree(ctx, depth, cap)
    off = 0
    for r, n in enumerate(sizes):
        t.step(arr[off:off + world * n])
        t.flush()                        # witness sets rotate: read each round before the fifth after it is prepared
        want = ref.insert_batch(vals[off:off + world * n])
        for k in range(world):
            got = t.outputs(r, k)
            assert got["low_sib"].shape == (depth, n, 32) and got["new_root"].shape == (n, 32)
            check_round(want, got, k * n, (k + 1) * n)
        off += world * n
    with pytest.raises(ValueError):
        t.step(arr[:world * batch + world])      # longer than the buffers
    with pytest.raises(ValueError):
        t.step(arr[:3])                          # not a multiple of the world size
    F = imt._ffi
    assert imt.lib.imt_sliced_step(t.h, ctypes.c_void_p(arr.data_ptr()), batch + 1, None, 0, None) == F.ERR["RANGE"]
    assert imt.lib.imt_sliced_step(t.h, None, 4, None, 0, None) == F.ERR["ARG"]
    assert all(tr.root() == ref.root() for tr in t.trees)
    ref.close()
    t.close()


def test_local_world_item_major_siblings(imt, ctx):
    """IMT_SIB_ITEM_MAJOR through imt_sliced_step: each insertion's proof as one contiguous [depth][32] row (the reference's
    per-proof Vec<F>, src/utils.rs:63-85), ragged last step included"""
    sl = load_sliced()
    depth, cap, world, batch = 32, 1 << 11, 2, 96
    sizes = [96, 96, 17]
    vals = oracle_lib.synth_values(world * sum(sizes), 0x494D5485)
    t = sl.SlicedTree(imt, 0, depth, cap, batch, world, n_local=world, item_major=True)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    ref = imt.IndexedTree(ctx, depth, cap)
    off = 0
    for r, n in enumerate(sizes):
        t.step(arr[off:off + world * n])
        t.flush()
        want = ref.insert_batch(vals[off:off + world * n])
        for k in range(world):
            got = t.outputs(r, k)
            lo, hi = k * n, (k + 1) * n
            for f in ("low_sib", "new_sib"):
                assert got[f].shape == (n, depth, 32)
                assert (got[f].cpu().numpy() == np.asarray(want[f])[:, lo:hi].transpose(1, 0, 2)).all(), (r, k, f)
            assert (got["new_root"].cpu().numpy() == np.asarray(want["new_root"])[lo:hi]).all()
        off += world * n
    assert all(tr.root() == ref.root() for tr in t.trees)
    ref.close()
    t.close()


def test_local_world_montgomery_format_and_refused_values(imt, ctx):
    sl = load_sliced()
    depth, cap, world, batch = 32, 1 << 12, 2, 128
    vals = oracle_lib.synth_values(world * batch * 2, 0x494D5451)
    F = imt._ffi
    t = sl.SlicedTree(imt, 0, depth, cap, batch, world, n_local=world, fmt=F.FMT_MONT256)
    R = 1 << 256
    mont = torch.from_numpy(oracle_lib.ints_to_arr([v * R % oracle_lib.P for v in vals])).cuda()
    t.step(mont[:world * batch])
    # a step with a duplicate (of a stored value, in the second rank's slice) is refused, nothing changes on any replica
    bad = mont[world * batch:].clone()
    bad[batch + 3] = mont[5]
    with pytest.raises(ValueError):
        t.step(bad)
    assert all(tr.size == 1 + world * batch for tr in t.trees)
    t.step(mont[world * batch:])
    t.flush()
    ref = imt.IndexedTree(ctx, depth, cap)
    for r in range(2):
        want = ref.insert_batch(vals[r * world * batch:(r + 1) * world * batch])
        for k in range(world):
            got = t.outputs(r, k)
            lo, hi = k * batch, (k + 1) * batch
            for f in ("old_root", "interim_root", "new_root"):
                g = [int.from_bytes(bytes(x), "little") for x in got[f].cpu().numpy()]
                assert g == [int.from_bytes(bytes(x), "little") * R % oracle_lib.P for x in want[f][lo:hi]], f
            assert (got["low_index"].cpu().numpy() == want["low_index"][lo:hi]).all()
            g = [int.from_bytes(bytes(x), "little") for x in got["low_sib"].cpu().numpy()[:, 7]]
            assert g == [int.from_bytes(bytes(x), "little") * R % oracle_lib.P for x in want["low_sib"][:, lo + 7]]
    assert all(tr.root() == ref.root() for tr in t.trees)
    ref.close()
    t.close()


# ---------------------------------------------------------------- one process per rank, sharing the GPU: the IPC transport
P_DEPTH, P_BATCH, P_ROUNDS = 32, 384, 5


def _worker(rank, world, port, q, host_poll):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if host_poll is not None:
        os.environ["IMT_IPC_HOST_POLL"] = str(host_poll)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)       # carries the IPC handle blobs, nothing else
    import imt_amd
    sl = load_sliced()
    boot = imt_amd.Context(0)
    tp = sl.ipc_transport(imt_amd, boot, dist, world, rank, P_DEPTH, P_BATCH)
    tree = sl.SlicedTree(imt_amd, 0, P_DEPTH, 1 << 13, P_BATCH, world, first_rank=rank, n_local=1, transport=tp)
    vals = oracle_lib.synth_values(world * P_BATCH * P_ROUNDS, 0x494D5461)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    res = []
    for r in range(P_ROUNDS):
        tree.step(arr[r * world * P_BATCH:(r + 1) * world * P_BATCH])
        if r >= 3:                      # read a finished round while later ones are in flight
            tree.wait(r - 3)
            res.append({k: v.cpu().numpy().copy() for k, v in tree.outputs(r - 3).items() if torch.is_tensor(v)})
    tree.flush()
    for r in range(max(0, P_ROUNDS - 3), P_ROUNDS):
        res.append({k: v.cpu().numpy().copy() for k, v in tree.outputs(r).items() if torch.is_tensor(v)})
    info = tree.info()
    root = tree.trees[0].root()
    dist.barrier()                      # nobody closes its exported buffers while a peer may still read them
    tree.close()
    boot.close()
    q.put((rank, res, root, info["collectives"], info["bytes_gathered"], {k: info[k] for k in ("pools", "queue_map", "placement")}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,host_poll", [(2, None), (4, None), (2, 0), (3, 0)])
def test_processes_over_ipc_equal_one_gpu_tree(imt, ctx, world, host_poll):
    """one PROCESS per rank on the one GPU (2 to 4: the box admits six GPU processes), payloads copied peer to peer out
    of IPC-mapped device memory, ordered by counters in shared host pages: every rank's witnesses and every replica's
    root equal the one-GPU tree.  host_poll None = the library's choice (ranks share the device: a worker thread on the
    host watches the counters); 0 = the form for ranks on different GPUs (the GPUs poll the counters themselves)"""
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [mpctx.Process(target=_worker, args=(r, world, port, q, host_poll)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    vals = oracle_lib.synth_values(world * P_BATCH * P_ROUNDS, 0x494D5461)
    want, want_root = reference_run(imt, ctx, P_DEPTH, 1 << 13, vals, world * P_BATCH)
    for rank, res, root, ncoll, nbytes, place in got:
        assert root == want_root
        assert ncoll > 0 and nbytes > 0
        # one process per rank: the rounds in the HIGH-priority pool, the collectives in the LOW one (IMT_SLICED_OPT_POOLS)
        assert place["pools"] == 1 and sorted(place["queue_map"][0]) == [0, 1, 2, 3] and place["queue_map"][1] == [-2] * 4, place
        assert place["placement"] in ("as created", "repaired"), place
        for r in range(P_ROUNDS):
            for k in FIELDS:
                w = np.asarray(want[r][k])
                w = w[:, rank * P_BATCH:(rank + 1) * P_BATCH] if k.endswith("_sib") else w[rank * P_BATCH:(rank + 1) * P_BATCH]
                assert (res[r][k] == w).all(), (rank, r, k)


@pytest.mark.parametrize("layout", ["default", "pools", "one-pool", "one-communicator"])
def test_rccl_transport_with_one_rank(imt, ctx, layout):
    """the RCCL transport -- ncclCommInitRank and ncclAllGather called by the library on its own communicators and
    streams -- needs one GPU per rank (RCCL refuses two ranks on one device: profiles/r06_rccl_two_ranks_one_gpu.txt), so a
    one-GPU box runs it with a world of ONE: every tick's gather still goes through RCCL (a one-rank all-gather is a copy
    of the send buffer into the receive buffer).  layout: what a world of one resolves to by default; the THREE PRIORITY
    POOLS a real multi-GPU run gets (forced here: ncclAllGather on LOW-priority streams of the library's beside
    HIGH-priority round streams, RCCL's own bracket stream in the normal pool); every collective on its round's own
    stream (bench.py's later attempts); ONE communicator for all four round slots (IMT_BENCH_RCCL_COMMS=1: the slots'
    collectives then share one stream, in issue order)."""
    sl = load_sliced()
    depth, cap, batch, rounds = 32, 1 << 12, 256, 5
    F, lib = imt._ffi, imt.lib
    if layout == "pools":
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_POOLS, 1) == 0
    elif layout == "one-pool":
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_POOLS, 0) == 0
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_COMM_STREAMS, 0) == 0
    try:
        _rccl_one_rank(imt, ctx, sl, depth, cap, batch, rounds, layout)
    finally:
        lib.imt_sliced_set_option(None, F.SLICED_OPT_RESET, 0)


def _rccl_one_rank(imt, ctx, sl, depth, cap, batch, rounds, layout):
    F, lib = imt._ffi, imt.lib
    ver = ctypes.c_int(0)
    boot = imt.Context(0)
    tp = sl.rccl_transport(imt, boot, None, 1, 0, n_comms=1 if layout == "one-communicator" else 4)
    path = lib.imt_rccl_library(ctypes.byref(ver)).decode()
    assert "rccl" in path and ver.value >= 21000, (path, ver.value)
    vals = oracle_lib.synth_values(batch * rounds, 0x494D5463)
    want, want_root = reference_run(imt, ctx, depth, cap, vals, batch)
    t = sl.SlicedTree(imt, 0, depth, cap, batch, 1, transport=tp)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    for r in range(rounds):
        t.step(arr[r * batch:(r + 1) * batch])
    t.flush()
    for r in range(rounds):
        check_round(want[r], t.outputs(r), 0, batch)
    assert t.trees[0].root() == want_root
    info = t.info()
    assert info["collectives"] == rounds * depth          # unit 0 carries nothing; one gather per level
    if layout == "pools":
        assert info["pools"] == 1 and sorted(info["queue_map"][0]) == [0, 1, 2, 3] and info["queue_map"][1] == [-2] * 4, info
    elif layout == "one-pool":
        assert info["pools"] == 0 and info["comm_streams"] == 0, info
    t.close()
    boot.close()


def test_rccl_transport_over_communicators_the_host_owns(imt, ctx):
    """imt_transport_rccl_adopt: a host that already has communicators (its own NCCL program) hands them over; the library
    issues its all-gathers on them and does NOT destroy them.  One rank (RCCL takes one rank per GPU), two communicators
    made here with RCCL's C API through ctypes; afterwards they still work and are destroyed by their owner."""
    sl = load_sliced()
    F, lib = imt._ffi, imt.lib
    ver = ctypes.c_int(0)
    path = lib.imt_rccl_library(ctypes.byref(ver)).decode()
    rccl = ctypes.CDLL(path)

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rccl.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    torch.cuda.set_device(0)
    comms = []
    for _ in range(2):
        uid = UniqueId()
        assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
        c = ctypes.c_void_p()
        assert rccl.ncclCommInitRank(ctypes.byref(c), 1, uid, 0) == 0
        comms.append(c)
    tp = ctypes.c_void_p()
    arr_c = (ctypes.c_void_p * 2)(*[c.value for c in comms])
    assert lib.imt_transport_rccl_adopt(arr_c, 2, ctypes.byref(tp)) == 0
    depth, cap, batch, rounds = 32, 1 << 12, 128, 5
    vals = oracle_lib.synth_values(batch * rounds, 0x494D5464)
    want, want_root = reference_run(imt, ctx, depth, cap, vals, batch)
    t = sl.SlicedTree(imt, 0, depth, cap, batch, 1, transport=tp)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    for r in range(rounds):
        t.step(arr[r * batch:(r + 1) * batch])
    t.flush()
    for r in range(rounds):
        check_round(want[r], t.outputs(r), 0, batch)
    assert t.trees[0].root() == want_root and t.info()["collectives"] == rounds * depth
    t.close()                      # destroys the world and the transport -- not the communicators
    a = torch.arange(64, dtype=torch.uint8, device="cuda")
    b = torch.zeros(64, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for c in comms:
        assert rccl.ncclAllGather(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), 64, 2, c, ctypes.c_void_p(st)) == 0      # 2 = ncclUint8
        torch.cuda.synchronize()
        assert bool((a == b).all())
        b.zero_()
        assert rccl.ncclCommDestroy(c) == 0


def test_custom_transport_vtable(imt, ctx):
    """a caller-supplied collective through imt_transport_custom_create: here a world of one, whose all-gather is a
    device-to-device copy issued on the stream the library hands over"""
    sl = load_sliced()
    F, lib = imt._ffi, imt.lib
    calls = []
    hip = ctypes.CDLL("libamdhip64.so.7")       # the HIP runtime this process already holds
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]

    def all_gather(self_, channel, buffer, send, recv, nbytes, stream):
        calls.append((channel, buffer, nbytes))
        return 0 if hip.hipMemcpyAsync(recv, send, nbytes, 3, stream) == 0 else F.ERR["HIP"]

    ops = F.TransportOps(None, F.TransportOps.ALL_GATHER(all_gather), F.TransportOps.DESTROY())
    tp = ctypes.c_void_p()
    assert lib.imt_transport_custom_create(ctypes.byref(ops), ctypes.byref(tp)) == 0
    depth, cap, batch = 32, 1 << 10, 64
    vals = oracle_lib.synth_values(3 * batch, 0x494D5467)
    want, want_root = reference_run(imt, ctx, depth, cap, vals, batch)
    t = sl.SlicedTree(imt, 0, depth, cap, batch, 1, transport=tp)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    for r in range(3):
        t.step(arr[r * batch:(r + 1) * batch])
    t.flush()
    for r in range(3):
        check_round(want[r], t.outputs(r), 0, batch)
    assert t.trees[0].root() == want_root and len(calls) == 3 * depth
    assert {c[0] for c in calls} == {0, 1, 2} and all(0 <= c[1] <= t.info()["lag"] for c in calls)
    t.close()


def test_batches_before_and_after_slices_on_one_tree(imt, ctx):
    """a tree goes from ordinary pipelined batches (imt_itree_insert_batch with IMT_PIPELINE) to sliced steps and back
    with no synchronisation by the caller in between: slices are ordered behind the batches in flight (the sweep of a
    batch and a slice's units touch the same stored levels), and a batch behind the slices' last kernels"""
    sl = load_sliced()
    depth, cap, batch = 32, 1 << 15, 2048
    vals = oracle_lib.synth_values(8 * batch, 0x494D5483)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    t = sl.SlicedTree(imt, 0, depth, cap, batch, 1)
    tr, c = t.trees[0], t.ctxs[0]
    F = imt._ffi
    P_ = lambda k: ctypes.c_void_p(arr[k * batch:(k + 1) * batch].data_ptr())
    for k in (0, 1):        # pipelined batches first ...
        c._check(imt.lib.imt_itree_insert_batch(tr.h, P_(k), batch, None, F.DEVICE_PTRS | F.PIPELINE))
    t.step(arr[2 * batch:3 * batch])        # ... slices second, nothing waited for
    t.step(arr[3 * batch:4 * batch])
    t.flush()
    for k in (4, 5):        # and back: batches behind the slices
        c._check(imt.lib.imt_itree_insert_batch(tr.h, P_(k), batch, None, F.DEVICE_PTRS | F.PIPELINE))
    t.step(arr[6 * batch:7 * batch])        # a slice right behind two pipelined batches
    # mid-step the replica is not a tree anyone should read or write through the ordinary calls: they refuse
    with pytest.raises(imt.ImtError) as e:
        tr.root()
    assert e.value.code == F.ERR["ARG"] and "imt_sliced_flush" in str(e.value)
    assert imt.lib.imt_itree_insert_batch(tr.h, P_(7), batch, None, F.DEVICE_PTRS | F.PIPELINE) == F.ERR["ARG"]
    t.flush()
    c._check(imt.lib.imt_itree_insert_batch(tr.h, P_(7), batch, None, F.DEVICE_PTRS | F.PIPELINE))
    c.sync()
    ref = imt.IndexedTree(ctx, depth, cap)
    ref.insert_batch(vals)
    assert tr.root() == ref.root()
    ref.close()
    t.close()


def test_sliced_world_resumes_from_a_checkpoint(imt, ctx):
    """checkpoint / resume of the multi-GPU mode: the snapshot of ANY replica (imt_itree_get_leaves from its device index,
    right after a flush) loaded into every replica of a new world (imt_itree_load: checked and rebuilt on the GPU)
    continues bit-exactly -- witnesses and roots equal the one-GPU tree that never stopped."""
    sl = load_sliced()
    depth, cap, world, batch = 32, 1 << 13, 2, 300
    vals = oracle_lib.synth_values(6 * world * batch, 0x494D5484)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    step = world * batch
    want, root = reference_run(imt, ctx, depth, cap, vals, step)
    a = sl.SlicedTree(imt, 0, depth, cap, batch, world, n_local=world)
    for r in range(3):
        a.step(arr[r * step:(r + 1) * step])
    a.flush()
    snap = a.trees[1].snapshot()
    assert (snap == a.trees[0].snapshot()).all() and snap.shape[0] == 3 * step + 1
    a.close()
    b = sl.SlicedTree(imt, 0, depth, cap, batch, world, n_local=world)
    for t in b.trees:
        t.load(snap)
    for r in range(3, 6):
        R = b.step(arr[r * step:(r + 1) * step])
        b.flush()
        for k in range(world):
            check_round(want[r], b.outputs(R, k), k * batch, (k + 1) * batch)
    assert b.trees[0].root() == b.trees[1].root() == root
    b.close()


def test_rank_emulation_tool_runs(imt):
    """tools/rank_emulation.py (one rank of an N-rank run over a modelled transport, a timing tool): rank 1 of 4 goes
    through imt_sliced_step with a custom vtable whose receive slots hold its own payload -- the library must take that
    without a fault (the apply kernel clamps counts and node indices) and report a rate"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rank_emulation.py"), "4"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT, EMU_ROUNDS="3", EMU_RANKS="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("N = 4 rank 1 lag 3")]
    assert len(lines) == 2 and "free collectives" in lines[0] and "GB/s links" in lines[1], r.stdout
    assert all(float(l.split(": ")[1].split(" M insertions/s")[0]) > 0.5 for l in lines)


def test_slice_calls_refuse_bad_arguments(imt, ctx):
    """the C entry points directly: misaligned payloads / values, units out of order, a second preparation of too many
    slices, a placed tree, a short payload stride, worlds that are no schedule -- documented codes, nothing reaches a kernel"""
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
    # two payloads 256 bytes apart cannot hold unit 1 of a 16-insertion slice (128 + 36 * 17 bytes)
    sb, nn, un = (ctypes.c_uint64 * 2)(17, 33), (ctypes.c_uint64 * 2)(16, 16), (ctypes.c_int32 * 2)(1, 1)
    assert lib.imt_itree_slice_apply_gathered(t.h, P_(pay), 256, 2, sb, nn, un, None) == F.ERR["ARG"]
    root = np.zeros(32, dtype=np.uint8)
    assert lib.imt_itree_root_lagged(t.h, 0, root.ctypes.data_as(ctypes.c_void_p), 0) == F.ERR["ARG"]     # slices have no lagged root
    ctx.sync()
    ref = imt.IndexedTree(ctx, 32, 1 << 10)
    ref.insert_batch(oracle_lib.synth_values(64, 0x494D5471)[:16])
    assert t.root() == ref.root()          # one slice alone on the context's stream = an ordinary batch
    placed = imt.IndexedTree(ctx, 31, 1 << 10)
    placed.set_placement(32, 1)
    assert lib.imt_itree_slice_prepare(placed.h, P_(vals), 0, 16, 0, None, F.DEVICE_PTRS, ctypes.byref(sl), None) == F.ERR["ARG"]
    # imt_sliced_create: a placed tree, two ranks on one context, a lag that keeps 17 steps in flight, the wrong transport
    tp, h = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.imt_transport_local_create(ctypes.byref(tp)) == 0
    one = (ctypes.c_void_p * 1)(placed.h)
    assert lib.imt_sliced_create(one, 1, 1, 0, tp, 16, 0, ctypes.byref(h)) == F.ERR["ARG"]
    two = (ctypes.c_void_p * 2)(t.h, ref.h)
    assert lib.imt_sliced_create(two, 2, 2, 0, tp, 16, 0, ctypes.byref(h)) == F.ERR["ARG"]      # one context for both
    one = (ctypes.c_void_p * 1)(t.h)
    assert lib.imt_sliced_create(one, 1, 1, 0, tp, 16, 2, ctypes.byref(h)) == F.ERR["RANGE"]
    assert lib.imt_sliced_create(one, 1, 2, 0, tp, 16, 0, ctypes.byref(h)) == F.ERR["ARG"]      # local transport, one of two ranks
    assert lib.imt_sliced_create(one, 1, 1, 0, None, 16, 0, ctypes.byref(h)) == F.ERR["ARG"]
    lib.imt_transport_destroy(tp)
    for x in (t, ref, placed):
        x.close()


# ------------------------------------------------------------------------- where the world's streams sit, and hangs
def _hip():
    hip = ctypes.CDLL("libamdhip64.so.7")       # the HIP runtime this process already holds
    hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    return hip


@pytest.mark.parametrize("dummies", [0, 1, 2, 3, 5])
def test_queue_placement_is_measured_and_repaired(imt, ctx, dummies):
    """The HIP runtime gives a new stream the hardware queue with the fewest streams of its priority, so where the
    world's eight streams land depends on how many streams the HOST created before (torch, RCCL, the application) --
    tools/microbench/queue_map_probe.hip.  imt_sliced_create measures the placement on the device (a spinning wave on
    one stream, time stamps on the others) and re-creates streams until the four round streams sit on four different
    hardware queues and slot i's collective stream on round stream i's queue, whatever came before: here 0 .. 5 extra
    streams.  The witnesses stay those of the one-GPU tree."""
    sl = load_sliced()
    hip = _hip()
    extra = []
    for _ in range(dummies):
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0        # hipStreamNonBlocking
        extra.append(s)
    depth, cap, batch = 32, 1 << 10, 48
    vals = oracle_lib.synth_values(3 * batch, 0x494D5470 + dummies)
    want, want_root = reference_run(imt, ctx, depth, cap, vals, batch)
    t = sl.SlicedTree(imt, 0, depth, cap, batch, 1)
    info = t.info()
    qm = info["queue_map"]
    assert info["placement"] in ("as created", "repaired"), (info["placement"], qm, imt.lib.imt_sliced_last_error(t.h))
    assert info["hw_queues"] == 4 and sorted(qm[0]) == [0, 1, 2, 3], qm
    assert qm[1] == qm[0] and info["comm_streams"] == 4, qm          # the collectives' streams: on their rounds' queues
    assert qm[2] == [-1] * 4                                         # no apply streams by default
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    for r in range(3):
        t.step(arr[r * batch:(r + 1) * batch])
    t.flush()
    for r in range(3):
        check_round(want[r], t.outputs(r), 0, batch)
    assert t.trees[0].root() == want_root
    t.close()
    for s in extra:
        hip.hipStreamDestroy(s)


def test_collectives_get_queues_of_their_own_when_the_runtime_has_eight(imt, ctx):
    """IMT_SLICED_OPT_COMM_PLACEMENT's default: with GPU_MAX_HW_QUEUES=8 in the host's environment (read when the HIP runtime
    starts, hence a process of its own) the collectives' streams end up on four queues that no round stream and no other
    helper is on -- a gather then overlaps its round's next units instead of holding them up until the slowest rank has
    packed; with the runtime's default of four queues they sit on their rounds' queues (the test above).  Same tree either
    way."""
    import json
    import subprocess
    roots = []
    for nq, dummies in ((8, 0), (8, 3), (None, 1)):
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)
        if nq:
            env["GPU_MAX_HW_QUEUES"] = str(nq)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "placement_check.py"), str(dummies)], env=env, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        o = json.loads(r.stdout.strip().splitlines()[-1])
        qm = o["queue_map"]
        assert o["placement"] in ("as created", "repaired") and o["hw_queues"] == 4 and sorted(qm[0]) == [0, 1, 2, 3], o
        assert qm[1] == ([4, 5, 6, 7] if nq else qm[0]), o
        roots.append(o["root"])
    assert len(set(roots)) == 1


def test_sliced_options(imt, ctx):
    """imt_sliced_set_option: process-wide defaults for worlds created later (NULL handle), live options of a world, ranges"""
    sl = load_sliced()
    F, lib = imt._ffi, imt.lib
    assert lib.imt_sliced_set_option(None, F.SLICED_OPT_COMM_STREAMS, 5) == F.ERR["RANGE"]
    assert lib.imt_sliced_set_option(None, 99, 0) == F.ERR["ARG"]
    try:
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_APPLY_STREAMS, 1) == 0
        t = sl.SlicedTree(imt, 0, 32, 1 << 10, 32, 1)
        info = t.info()
        assert info["queue_map"][2] == info["queue_map"][0] and info["placement"] in ("as created", "repaired"), info
        assert lib.imt_sliced_set_option(t.h, F.SLICED_OPT_APPLY_STREAMS, 0) == F.ERR["ARG"]      # decides which streams exist
        t.set_option(F.SLICED_OPT_PREP_STREAM, 1)
        t.set_option(F.SLICED_OPT_WATCHDOG_MS, 5000)
        vals = oracle_lib.synth_values(64, 0x494D5479)
        want, want_root = reference_run(imt, ctx, 32, 1 << 10, vals, 32)
        arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
        for r in range(2):
            t.step(arr[r * 32:(r + 1) * 32])
        t.flush()
        assert t.trees[0].root() == want_root
        assert "global tick" in t.dump() and "everything issued is complete" in t.dump()
        t.close()
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_APPLY_STREAMS, 0) == 0
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_VERIFY_QUEUES, 0) == 0
        t = sl.SlicedTree(imt, 0, 32, 1 << 10, 32, 1)
        assert t.info()["placement"] == "unverified" and t.info()["queue_map"][0] == [-1] * 4
        t.close()
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_VERIFY_QUEUES, 1) == 0
        # three priority pools (what one process per GPU gets by default): the rounds alone on four queues of the HIGH
        # pool, the collectives' streams on four of the LOW one, the same tree
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_POOLS, 3) == F.ERR["RANGE"]
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_POOLS, 2) == 0      # rounds and collectives in the HIGH pool, on shared queues
        t = sl.SlicedTree(imt, 0, 32, 1 << 10, 32, 1)
        info = t.info()
        assert info["pools"] == 2 and sorted(info["queue_map"][0]) == [0, 1, 2, 3] and info["queue_map"][1] == info["queue_map"][0], info
        t.close()
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_POOLS, 1) == 0
        t = sl.SlicedTree(imt, 0, 32, 1 << 10, 32, 2, n_local=2)
        info = t.info()
        assert info["pools"] == 1 and sorted(info["queue_map"][0]) == [0, 1, 2, 3] and info["queue_map"][1] == [-2] * 4, info
        assert info["placement"] in ("as created", "repaired") and info["comm_streams"] == 4, info
        vals2 = oracle_lib.synth_values(2 * 32 * 5, 0x494D547A)
        want2, want_root2 = reference_run(imt, ctx, 32, 1 << 10, vals2, 64)
        arr2 = torch.from_numpy(oracle_lib.ints_to_arr(vals2)).cuda()
        for r in range(5):
            t.step(arr2[r * 64:(r + 1) * 64])
        t.flush()
        for r in range(2, 5):
            for k in range(2):
                check_round(want2[r], t.outputs(r, k), k * 32, (k + 1) * 32)
        assert t.trees[0].root() == want_root2 and t.trees[1].root() == want_root2
        t.close()
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_POOLS, -1) == 0
        t = sl.SlicedTree(imt, 0, 32, 1 << 10, 32, 2, n_local=2)           # replicas of one process: one pool
        assert t.info()["pools"] == 0 and t.info()["queue_map"][1] == t.info()["queue_map"][0]
        t.close()
    finally:
        lib.imt_sliced_set_option(None, F.SLICED_OPT_RESET, 0)


def test_transport_is_one_world_at_a_time(imt, ctx):
    """a transport serves one world at a time and cannot be destroyed under it"""
    sl = load_sliced()
    F, lib = imt._ffi, imt.lib
    t = sl.SlicedTree(imt, 0, 32, 1 << 10, 16, 1)
    c2 = imt.Context(0)
    tree2 = imt.IndexedTree(c2, 32, 1 << 10)
    h = ctypes.c_void_p()
    arr = (ctypes.c_void_p * 1)(tree2.h)
    assert lib.imt_sliced_create(arr, 1, 1, 0, t.tp, 16, 0, ctypes.byref(h)) == F.ERR["ARG"]
    assert lib.imt_transport_destroy(t.tp) == F.ERR["ARG"]           # still in use
    send = torch.zeros(32, dtype=torch.uint8, device="cuda")
    assert lib.imt_transport_all_gather(t.tp, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(send.data_ptr()), 32, None) == F.ERR["ARG"]
    tree2.close()
    c2.close()
    t.close()


def test_watchdog_reports_where_the_world_stands(imt, ctx, capfd):
    """A collective that does not complete (here: a custom transport whose fifth all-gather holds its stream for ~3 s,
    standing in for a peer that died) must not hang the caller for good: the host wait inside the next imt_sliced_step
    runs into the world's watchdog (set to 400 ms), the call returns IMT_ERR_TIMEOUT well before the collective ends,
    stderr and imt_sliced_last_error name the round slot, the first tick that has not completed and the pending
    collective with its channel, and the world refuses to go on."""
    import time
    sl = load_sliced()
    F, lib = imt._ffi, imt.lib
    hip = _hip()
    dev = torch.device("cuda", 0)
    # torch.cuda._sleep counts cycles of SOME clock: measure which
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(20_000_000)
    e1.record()
    torch.cuda.synchronize()
    clock_hz = 20_000_000 / (e0.elapsed_time(e1) * 1e-3)
    calls = []

    def all_gather(self_, channel, buffer, send, recv, nbytes, stream):
        calls.append((channel, buffer))
        if len(calls) == 5:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                torch.cuda._sleep(int(3.0 * clock_hz))
        return 0 if hip.hipMemcpyAsync(recv, send, nbytes, 3, stream) == 0 else F.ERR["HIP"]

    ops = F.TransportOps(None, F.TransportOps.ALL_GATHER(all_gather), F.TransportOps.DESTROY())
    tp = ctypes.c_void_p()
    assert lib.imt_transport_custom_create(ctypes.byref(ops), ctypes.byref(tp)) == 0
    depth, cap, batch = 32, 1 << 12, 64
    t = sl.SlicedTree(imt, 0, depth, cap, batch, 1, transport=tp)
    t.set_option(F.SLICED_OPT_WATCHDOG_MS, 400)
    vals = oracle_lib.synth_values(8 * batch, 0x494D5471)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    code = None
    for r in range(8):
        try:
            t.step(arr[r * batch:(r + 1) * batch])
        except imt.ImtError as e:
            code, msg = e.code, str(e)
            break
    dt = time.perf_counter() - t0
    assert code == F.ERR["TIMEOUT"], code
    assert dt < 2.0, f"the call came back after {dt:.2f} s"
    err = capfd.readouterr().err
    for text in (msg, err):
        assert "imt_sliced_step failed with -13" in text and "collective NOT COMPLETE on channel 0" in text, text
        assert "first incomplete: unit tick" in text and "global tick" in text
    with pytest.raises(imt.ImtError) as ei:
        t.step(arr[:batch])
    assert ei.value.code == F.ERR["INTERNAL"] and "cannot go on" in str(ei.value)
    with pytest.raises(imt.ImtError):
        t.flush()
    torch.cuda.synchronize()            # the long collective ends; everything drains
    t.close()


def _poison_worker(rank, world, port, q):
    """rank 1 stops after two steps (a rank that died, as far as rank 0 can tell) but keeps its buffers mapped"""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import imt_amd
    F, lib = imt_amd._ffi, imt_amd.lib
    sl = load_sliced()
    boot = imt_amd.Context(0)
    nb = int(lib.imt_transport_ipc_blob_bytes())
    mine = torch.zeros(nb, dtype=torch.uint8)
    tp = ctypes.c_void_p()
    assert lib.imt_transport_ipc_create(boot.h, world, rank, P_DEPTH, P_BATCH, 0, ctypes.byref(tp), ctypes.c_void_p(mine.data_ptr())) == 0
    assert lib.imt_transport_set_option(tp, F.TRANSPORT_OPT_HOST_POLL, 0) == 0         # the GPUs poll: the form real multi-GPU runs use
    assert lib.imt_transport_set_option(tp, F.TRANSPORT_OPT_TIMEOUT_MS, 300) == 0
    assert lib.imt_transport_set_option(tp, F.TRANSPORT_OPT_TIMEOUT_MS, 0) == F.ERR["RANGE"]
    allb = torch.zeros(world * nb, dtype=torch.uint8)
    dist.all_gather_into_tensor(allb, mine)
    assert lib.imt_transport_ipc_connect(tp, ctypes.c_void_p(allb.data_ptr())) == 0
    tree = sl.SlicedTree(imt_amd, 0, P_DEPTH, 1 << 14, P_BATCH, world, first_rank=rank, n_local=1, transport=tp)
    vals = oracle_lib.synth_values(world * P_BATCH * 14, 0x494D5472)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    gb = world * P_BATCH
    out = dict(rank=rank, code=None, at=None, msg="")
    t0 = time.perf_counter()
    try:
        for r in range(14):     # (a step five rounds later waits for the stuck round's plan set: the error is seen by then)
            if rank == 1 and r == 2:
                break
            tree.step(arr[r * gb:(r + 1) * gb])
            out["steps_ok"] = r + 1
        if rank == 0:
            tree.flush()
    except imt_amd.ImtError as e:
        out.update(code=e.code, at=time.perf_counter() - t0, msg=str(e)[:600])
        try:
            tree.step(arr[:gb])
        except imt_amd.ImtError as e2:
            out["again"] = e2.code
    q.put(out)
    dist.barrier()              # rank 1 keeps its exported buffers until rank 0 has reported
    os._exit(0)                 # no teardown of a world whose peer is gone


def test_a_vanished_peer_poisons_the_world_instead_of_corrupting_it(imt):
    """GPU-polled IPC transport, two processes, rank 1 stops stepping: rank 0's wait kernels give up after the
    transport's time limit (300 ms here), the sticky error word makes every copy / acknowledgement / apply behind them a
    no-op on the device (csrc/imt_flags.hip), and the host hears about it at its NEXT imt_sliced_step -- not only at
    flush -- as IMT_ERR_INTERNAL; the world then refuses further steps."""
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [mpctx.Process(target=_poison_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {o["rank"]: o for o in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    F = imt._ffi
    r0 = got[0]
    assert r0["code"] == F.ERR["INTERNAL"], r0
    assert "did not arrive" in r0["msg"] and r0["at"] < 30.0, r0
    assert r0.get("steps_ok", 0) < 14 and r0.get("again") == F.ERR["INTERNAL"], r0
    assert got[1]["code"] is None and got[1]["steps_ok"] == 2


def test_destroy_after_a_timeout_does_not_wait_for_the_device(imt, ctx, capfd):
    """after the watchdog has fired the documented way out is to destroy the world -- which must not itself block
    on the collective that never ends.  A custom transport whose fifth all-gather holds its stream for ~5 s; watchdog at
    300 ms; imt_sliced_destroy + imt_transport_destroy right after IMT_ERR_TIMEOUT return within a couple of watchdog
    periods (hipFree, which waits for the whole device, is skipped: the world's buffers and streams are left allocated and
    stderr says so), long before the collective ends."""
    import time
    sl = load_sliced()
    F, lib = imt._ffi, imt.lib
    hip = _hip()
    dev = torch.device("cuda", 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(20_000_000)
    e1.record()
    torch.cuda.synchronize()
    clock_hz = 20_000_000 / (e0.elapsed_time(e1) * 1e-3)
    calls = []

    def all_gather(self_, channel, buffer, send, recv, nbytes, stream):
        calls.append(channel)
        if len(calls) == 5:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                torch.cuda._sleep(int(5.0 * clock_hz))
        return 0 if hip.hipMemcpyAsync(recv, send, nbytes, 3, stream) == 0 else F.ERR["HIP"]

    ops = F.TransportOps(None, F.TransportOps.ALL_GATHER(all_gather), F.TransportOps.DESTROY())
    tp = ctypes.c_void_p()
    assert lib.imt_transport_custom_create(ctypes.byref(ops), ctypes.byref(tp)) == 0
    t = sl.SlicedTree(imt, 0, 32, 1 << 12, 64, 1, transport=tp)
    t.set_option(F.SLICED_OPT_WATCHDOG_MS, 300)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(oracle_lib.synth_values(8 * 64, 0x494D5474))).cuda()
    torch.cuda.synchronize()
    t_stall = None
    code = None
    for r in range(8):
        try:
            t.step(arr[r * 64:(r + 1) * 64])
        except imt.ImtError as e:
            code = e.code
            break
        if len(calls) >= 5 and t_stall is None:
            t_stall = time.perf_counter()
    assert code == F.ERR["TIMEOUT"]
    t0 = time.perf_counter()
    lib.imt_sliced_destroy(t.h)
    t.h = None
    assert lib.imt_transport_destroy(tp) == 0
    dt = time.perf_counter() - t0
    err = capfd.readouterr().err
    assert dt < 2.5, f"destroy took {dt:.2f} s"
    assert "did not drain" in err and "left allocated" in err
    # (this test lets the 5-s collective end and then closes trees and contexts; a real host would exit instead)
    torch.cuda.synchronize()
    for tr in t.trees:
        tr.close()
    for c in t.ctxs:
        c.close()


def test_pool_presets_leave_explicit_options_alone(imt, ctx):
    """IMT_SLICED_OPT_POOLS 1 presets ROUND_PRIORITIES / COMM_PRIORITY / PREP_STREAM -- but not over a value the
    caller has set explicitly; the new world's last_error says which preset was left out.  IMT_SLICED_OPT_RESET forgets."""
    sl = load_sliced()
    F, lib = imt._ffi, imt.lib
    try:
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_POOLS, 1) == 0
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_PREP_STREAM, 0) == 0
        t = sl.SlicedTree(imt, 0, 32, 1 << 10, 32, 2, n_local=2)
        msg = lib.imt_sliced_last_error(t.h).decode()
        assert "PREP_STREAM stays at the caller's 0" in msg and "ROUND_PRIORITIES" not in msg, msg
        info = t.info()
        assert info["pools"] == 1 and sorted(info["queue_map"][0]) == [0, 1, 2, 3] and info["queue_map"][1] == [-2] * 4, info
        vals = oracle_lib.synth_values(2 * 32 * 3, 0x494D547B)
        want, want_root = reference_run(imt, ctx, 32, 1 << 10, vals, 64)
        arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
        for r in range(3):
            t.step(arr[r * 64:(r + 1) * 64])
        t.flush()
        assert t.trees[0].root() == want_root and t.trees[1].root() == want_root
        t.close()
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_RESET, 1) == F.ERR["RANGE"]
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_RESET, 0) == 0
        assert lib.imt_sliced_set_option(None, F.SLICED_OPT_POOLS, 1) == 0
        t = sl.SlicedTree(imt, 0, 32, 1 << 10, 32, 2, n_local=2)
        assert lib.imt_sliced_set_option(t.h, F.SLICED_OPT_RESET, 0) == F.ERR["ARG"]
        assert "stays at the caller's" not in lib.imt_sliced_last_error(t.h).decode()
        t.close()
    finally:
        lib.imt_sliced_set_option(None, F.SLICED_OPT_RESET, 0)


def _gather_worker(rank, world, port, q):
    """the subtree layout's one collective over the GPU-polled IPC transport; rank 1 takes part in the first gather only"""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import imt_amd
    F, lib = imt_amd._ffi, imt_amd.lib
    boot = imt_amd.Context(0)
    nb = int(lib.imt_transport_ipc_blob_bytes())
    mine = torch.zeros(nb, dtype=torch.uint8)
    tp = ctypes.c_void_p()
    assert lib.imt_transport_ipc_create(boot.h, world, rank, 32, 64, 0, ctypes.byref(tp), ctypes.c_void_p(mine.data_ptr())) == 0
    assert lib.imt_transport_set_option(tp, F.TRANSPORT_OPT_HOST_POLL, 0) == 0
    assert lib.imt_transport_set_option(tp, F.TRANSPORT_OPT_TIMEOUT_MS, 300) == 0
    allb = torch.zeros(world * nb, dtype=torch.uint8)
    dist.all_gather_into_tensor(allb, mine)
    assert lib.imt_transport_ipc_connect(tp, ctypes.c_void_p(allb.data_ptr())) == 0
    send = torch.full((32,), 10 + rank, dtype=torch.uint8, device="cuda")
    out = dict(rank=rank)

    def gather():
        recv = torch.zeros((world, 32), dtype=torch.uint8, device="cuda")
        rc = lib.imt_transport_all_gather(tp, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()), 32, None)
        boot.sync()
        return rc, recv.cpu().numpy(), lib.imt_transport_poll_error(tp)
    rc, recv, pe = gather()
    out["first"] = (rc, recv[:, 0].tolist(), pe)
    dist.barrier()
    if rank == 0:
        t0 = time.perf_counter()
        rc, recv, pe = gather()                     # the peer never comes: the GPU-side wait gives up after 300 ms
        out["second"] = (rc, recv[:, 0].tolist(), pe, time.perf_counter() - t0, lib.imt_transport_last_error(tp).decode())
        rc, recv, pe = gather()                     # sticky: refused at the call, nothing enqueued
        out["third"] = (rc, recv[:, 0].tolist(), pe)
    q.put(out)
    dist.barrier()
    os._exit(0)


def test_subtree_gather_reports_a_peer_that_never_comes(imt):
    """imt_transport_all_gather over the GPU-polled IPC transport used to return IMT_OK for ever after a
    GPU-side wait had given up -- with `recv` never written.  Now: the call whose wait gives up still returns IMT_OK (the
    wait runs on the device), imt_transport_poll_error after the stream's sync says IMT_ERR_INTERNAL and `recv` holds
    the caller's zeros for the missing rank; every later call is refused at once."""
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [mpctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {o["rank"]: o for o in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    F = imt._ffi
    for r in (0, 1):
        assert got[r]["first"] == (0, [10, 11], 0), got[r]
    rc, col, pe, dt, msg = got[0]["second"]
    assert rc == 0 and pe == F.ERR["INTERNAL"] and col == [10, 0] and dt < 10.0 and "did not arrive" in msg, got[0]
    rc, col, pe = got[0]["third"]
    assert rc == F.ERR["INTERNAL"] and pe == F.ERR["INTERNAL"] and col == [0, 0], got[0]


def _hung_ipc_worker(rank, world, port, q):
    """GPU-polled IPC world of two; rank 1 stops after two steps.  Rank 0's wait kernels would poll for 8 s (the transport's
    limit) but its watchdog is at 500 ms: IMT_ERR_TIMEOUT, then the documented way out -- destroy world and transport while
    the wait kernel is STILL polling -- must come back at once, with the name in /dev/shm gone."""
    import glob
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import imt_amd
    F, lib = imt_amd._ffi, imt_amd.lib
    sl = load_sliced()
    boot = imt_amd.Context(0)
    nb = int(lib.imt_transport_ipc_blob_bytes())
    mine = torch.zeros(nb, dtype=torch.uint8)
    tp = ctypes.c_void_p()
    assert lib.imt_transport_ipc_create(boot.h, world, rank, P_DEPTH, P_BATCH, 0, ctypes.byref(tp), ctypes.c_void_p(mine.data_ptr())) == 0
    assert lib.imt_transport_set_option(tp, F.TRANSPORT_OPT_HOST_POLL, 0) == 0
    assert lib.imt_transport_set_option(tp, F.TRANSPORT_OPT_TIMEOUT_MS, 8000) == 0
    allb = torch.zeros(world * nb, dtype=torch.uint8)
    dist.all_gather_into_tensor(allb, mine)
    assert lib.imt_transport_ipc_connect(tp, ctypes.c_void_p(allb.data_ptr())) == 0
    tree = sl.SlicedTree(imt_amd, 0, P_DEPTH, 1 << 14, P_BATCH, world, first_rank=rank, n_local=1, transport=tp)
    tree.set_option(F.SLICED_OPT_WATCHDOG_MS, 500)
    vals = oracle_lib.synth_values(world * P_BATCH * 14, 0x494D5473)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    gb = world * P_BATCH
    out = dict(rank=rank, code=None)
    shm_before = glob.glob(f"/dev/shm/imt_ipc_{os.getpid()}_*")
    try:
        for r in range(14):
            if rank == 1 and r == 2:
                break
            tree.step(arr[r * gb:(r + 1) * gb])
        if rank == 0:
            tree.flush()
    except imt_amd.ImtError as e:
        out["code"] = e.code
        t0 = time.perf_counter()
        lib.imt_sliced_destroy(tree.h)
        tree.h = None
        out["destroy_rc"] = lib.imt_transport_destroy(tp)
        out["destroy_s"] = time.perf_counter() - t0
        out["shm_before"], out["shm_after"] = len(shm_before), len(glob.glob(f"/dev/shm/imt_ipc_{os.getpid()}_*"))
    q.put(out)
    dist.barrier()
    os._exit(0)                 # what a real host does next: report and exit, recovery in a fresh process


def test_a_hung_ipc_world_can_be_left_behind(imt):
    """ADVICE r5 (medium), the IPC transport's side of it: after IMT_ERR_TIMEOUT, imt_sliced_destroy + imt_transport_destroy
    return while a wait kernel is still polling for the vanished peer (no hipFree, no stream sync: the device side is
    leaked, the host side -- worker thread, the name in /dev/shm -- is cleaned up)."""
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [mpctx.Process(target=_hung_ipc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {o["rank"]: o for o in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    F = imt._ffi
    r0 = got[0]
    assert r0["code"] == F.ERR["TIMEOUT"], r0
    assert r0["destroy_rc"] == 0 and r0["destroy_s"] < 3.0, r0
    assert r0["shm_before"] == 1 and r0["shm_after"] == 0, r0
    assert got[1]["code"] is None
