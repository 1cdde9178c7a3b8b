"""GPU, two processes on one MI355X (gloo collectives through host memory).

* the subtree mode (bench.py's N > 1 path): sharded.ShardedIndexedTree over sharded.GpuBackend -- lagged
  root exchange, lift to depth 32 -- against the CPU oracle building the same two subtrees, and at depth 8
  against a dense rebuild of the whole tree after every event;
* the single-list mode, ReplicatedIndexedTree: each rank returns the witnesses of its half of every batch;
  together they must equal the one-process imt_itree_insert_batch results."""
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


# ------------------------------------------------------------------------------------------------
# subtree mode through ShardedIndexedTree + GpuBackend
# ------------------------------------------------------------------------------------------------
SUB_N, SUB_STEPS = 6, 3


def _sub_vals(rank, world, seed):
    import oracle_lib
    raw = oracle_lib.synth_values(SUB_N * SUB_STEPS * world * 4, seed)
    return [v for v in raw if v % world == rank][:SUB_N * SUB_STEPS]


def _sub_worker(rank, world, port, depth, q, library_gather=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import imt_amd
    sharded = _load_sharded()
    k = world.bit_length() - 1
    be = sharded.GpuBackend(imt_amd, 0, depth, world, rank, 64, SUB_N, pipeline=True, nbuf=2)
    tp = None
    if library_gather:      # the root exchange through the library's own communicators (imt_transport_all_gather over IPC);
        import importlib.util      # gloo only carries the IPC handle blobs
        spec = importlib.util.spec_from_file_location("imt_sliced", os.path.join(ROOT, "indexed-merkle-tree-halo2_amd", "sliced.py"))
        sl = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sl)
        tp = sl.ipc_transport(imt_amd, be.ctx, dist, world, rank, depth, 1)
    tree = sharded.ShardedIndexedTree(be, depth, world, rank, dist, via_host=True, transport=tp)
    vals = _sub_vals(rank, world, 0x494D5450 + depth)
    finished, roots, fails = [], [], []

    def take(done):
        be.sync()
        n = SUB_N
        fail = torch.empty(n, dtype=torch.uint8, device=be.device)
        new_index = torch.arange(done["first_new_index"], done["first_new_index"] + n, dtype=torch.int64, device=be.device)
        import ctypes
        P_ = lambda x: ctypes.c_void_p(x.data_ptr())
        rc = imt_amd.lib.imt_insert_witness_batch(be.ctx.h, P_(done["old_root"]), P_(done["low_leaf"]), P_(done["low_index"]),
                                                  P_(done["low_sib"]), P_(done["new_root"]), P_(done["new_leaf"]),
                                                  P_(new_index), None, P_(done["new_sib"]), P_(done["is_largest"]), depth,
                                                  n, P_(fail), None, imt_amd._ffi.DEVICE_PTRS)
        assert rc == 0
        be.sync()
        fails.append(int(fail.max()))
        finished.append({key: (v.cpu().numpy().copy() if torch.is_tensor(v) else v) for key, v in done.items()})
        roots.append(tree.global_root.cpu().numpy().tobytes())

    for st in range(SUB_STEPS):
        done = tree.step(imt_amd.to_bytes(vals[st * SUB_N:(st + 1) * SUB_N]))
        if done is not None:
            take(done)
    take(tree.flush())
    # config 3 in the same layout: values of this rank's residue that were never inserted, witnessed at full depth
    # against the global root and checked by the independent non-membership kernel
    import oracle_lib
    cand = [v for v in oracle_lib.synth_values(64 * world, 0x494D5460 + depth) if v % world == rank and v not in vals][:16]
    low, leaves, sib, largest = tree.non_membership_witness(imt_amd.to_bytes(cand))
    nm_fail = be.ctx.non_membership(tree.global_root.cpu().numpy(), leaves, low, sib, depth, imt_amd.to_bytes(cand), largest)
    assert sib.shape == (depth, 16, 32) and not nm_fail.any() and ((low >> (depth - k)) == rank).all()
    q.put((rank, finished, roots, fails))
    dist.barrier()              # nobody unmaps what a peer may still read
    if tp is not None:
        assert imt_amd.lib.imt_transport_destroy(tp) == 0
    dist.destroy_process_group()


@pytest.mark.parametrize("depth,library_gather", [(8, False), (32, False), (32, True)])
def test_subtree_mode_on_two_ranks(imt, ctx, oracle, depth, library_gather):
    """library_gather: the one collective of the layout through imt_transport_all_gather (the library's IPC transport)
    instead of torch.distributed -- what sharded.py shares with examples/subtree_procs_demo.c"""
    world = 2
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = 29800 + (os.getpid() % 1000) + depth + (50 if library_gather else 0)
    procs = [mpctx.Process(target=_sub_worker, args=(r, world, port, depth, q, library_gather)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    vals = [_sub_vals(r, world, 0x494D5450 + depth) for r in range(world)]
    steps = [[vals[r][st * SUB_N:(st + 1) * SUB_N] for r in range(world)] for st in range(SUB_STEPS)]
    ints = lambda a: [int.from_bytes(x.tobytes(), "little") for x in np.asarray(a, np.uint8).reshape(-1, 32)]
    for rank, finished, roots, fails in got:
        assert fails == [0] * SUB_STEPS and len(finished) == SUB_STEPS       # insert_leaf relations at full depth
    if depth <= 8:
        from sharded_ref import dense_global_replay
        want, final_root = dense_global_replay(oracle, depth, world, steps)
        for rank, finished, roots, _ in got:
            for st in range(SUB_STEPS):
                h = finished[st]
                for i, exp in enumerate(want[st][rank]):
                    assert int(h["low_index"][i]) == exp["low"] and h["first_new_index"] + i == exp["new_index"]
                    for key in ("old_root", "interim_root", "new_root"):
                        assert ints(h[key][i]) == [exp[key]], (st, rank, i, key)
                    assert (h["low_sib"][:, i] == exp["low_proof"]).all() and (h["new_sib"][:, i] == exp["new_proof"]).all()
                assert roots[st] == want[st][world - 1][-1]["new_root"].to_bytes(32, "little")
        assert got[0][2][-1] == final_root.to_bytes(32, "little")
    else:
        # depth 32: the oracle builds the same two height-31 subtrees; top level restated with oracle.hash
        sub = depth - 1
        hs = [oracle.sparse_new(sub, 64) for _ in range(world)]
        for r in range(world):
            oracle.sparse_set_index_base(hs[r], r << sub)
        sub_roots = [oracle.sparse_root(h) for h in hs]
        for st in range(SUB_STEPS):
            for r in range(world):
                h = got[r][1][st]
                for i, v in enumerate(steps[st][r]):
                    o = oracle.sparse_insert(hs[r], sub, v)
                    assert o["rc"] == 0
                    other = sub_roots[r ^ 1]          # rank 0 sees rank 1 before the step, rank 1 sees rank 0 after it
                    up = (lambda x: oracle.hash([x, other])) if r == 0 else (lambda x: oracle.hash([other, x]))
                    assert ints(h["interim_root"][i]) == [up(o["interim_root"])], (st, r, i)
                    assert ints(h["new_root"][i]) == [up(o["new_root"])]
                    assert int(h["low_index"][i]) == (r << sub) + o["low"]
                    assert (h["low_sib"][:sub, i] == o["low_proof"]).all() and (h["new_sib"][:sub, i] == o["new_proof"]).all()
                    assert ints(h["low_sib"][sub, i]) == [other] and ints(h["new_sib"][sub, i]) == [other]
                    assert (h["low_leaf"][i] == o["low_leaf"]).all()
                sub_roots[r] = oracle.sparse_root(hs[r])
            assert got[0][2][st] == oracle.hash(sub_roots).to_bytes(32, "little")
        for h in hs:
            oracle.sparse_free(h)


def test_bench_gpus2_as_typed_rehearsal():
    """`python3 bench.py --gpus 2 --steps 2 --warmup 1` exactly as the driver would type it (no torchrun around
    it): the process launches its two ranks as children, rank 0's JSON line comes back on stdout.  Rehearsal
    switches for the one-GPU box: both ranks on device 0, root exchange through gloo."""
    import json
    import subprocess
    env = dict(os.environ, IMT_BENCH_DEVICE="0", IMT_BENCH_COLLECTIVE="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["collective_backend"] == "gloo"
    assert res["verified"] is True
    # both multi-GPU modes in one invocation; `value` is the reference's single list
    assert res["value_is"].startswith("single-list") and res["config"]["hashes_per_insertion"] == 66
    sl, st = res["modes"]["single_list"], res["modes"]["subtrees"]
    assert sl["verified"] is True and st["verified"] is True and res["value"] == sl["value"]
    assert sl["hashes_per_insertion"] == 66 and st["hashes_per_insertion"] == 66
    assert sl["collectives_per_step"] > 30 and sl["bytes_gathered_per_step_per_rank"] > 0 and st["collectives_per_step"] == 1
    assert sl["roofline"]["frac"] > 0 and st["roofline"]["frac"] > 0 and res["cpu_baseline"] is None
    assert res["value"] > 0 and res["steps"] == 2 and res["scaling"] == "weak"
    # --gpus that contradicts the launcher's world size is an error, not silently ignored
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                        capture_output=True, text=True, timeout=300, env=env2, cwd=ROOT)
    assert r2.returncode != 0 and "does not match WORLD_SIZE" in (r2.stdout + r2.stderr)
    # an RCCL transport that cannot be created (here: more communicators than the library takes) -> every rank falls back
    # to the IPC transport, the line says so, the figure stands
    env3 = dict(env, IMT_BENCH_SLICED_TRANSPORT="rccl", IMT_BENCH_RCCL_COMMS="9", IMT_BENCH_MODE="single-list")
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                        capture_output=True, text=True, timeout=600, env=env3, cwd=ROOT)
    assert r3.returncode == 0, r3.stdout[-2000:] + r3.stderr[-4000:]
    res3 = json.loads([l for l in r3.stdout.splitlines() if l.startswith("{")][0])
    sch = res3["modes"]["single_list"]["schedule"]
    assert res3["verified"] is True and sch["transport"] == "ipc" and "fell back" in sch["transport_note"], sch


def test_bench_gpus4_as_typed_rehearsal():
    """`python3 bench.py --gpus 4 --steps 1 --warmup 1`, four ranks as children sharing device 0 over gloo: the largest
    multi-PROCESS rehearsal a one-GPU box allows (its process guard admits six GPU processes; eight ranks run as
    eight replicas in ONE process instead: test_gpu_sliced.py::test_local_world_equals_one_gpu_tree[8-...], and as eight
    gloo processes without a GPU: test_sliced_schedule.py::test_gloo_ranks_line_up[8-32-None]).  A rank that fails makes
    the launcher exit non-zero."""
    import json
    import subprocess
    env = dict(os.environ, IMT_BENCH_DEVICE="0", IMT_BENCH_COLLECTIVE="gloo", IMT_BENCH_NO_TRACE="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 4 and res["ranks_seen"] == 4 and res["verified"] is True
    assert res["modes"]["single_list"]["verified"] is True and res["modes"]["subtrees"]["verified"] is True
    assert res["modes"]["single_list"]["schedule"]["lag_levels"] == 3
    # a rank that dies takes the launcher's exit status with it (here: every rank refuses a world size it cannot shard)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], capture_output=True, text=True,
                         timeout=120, env=env, cwd=ROOT)
    assert bad.returncode != 0


def test_bench_keeps_the_first_leg_if_the_second_hangs():
    """A hung headline (single-list) leg must not look like a pass, and another mode's figure must never stand in for
    it.  A time limit far below what the leg needs stands in for the hang, in every attempt of the plan: every rank leaves
    with a non-zero status, rank 0 first prints the line with `value` null, `verified` false and the reason; the subtree
    leg -- which a worker runs BEHIND the headline leg since round 6 and therefore never reached -- is measured at the end
    in workers of its own and appears under `modes.subtrees` only."""
    import json
    import subprocess
    env = dict(os.environ, IMT_BENCH_DEVICE="0", IMT_BENCH_COLLECTIVE="gloo", IMT_BENCH_NO_TRACE="1",
               IMT_BENCH_SINGLE_LIST_TIMEOUT="0.3")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["value"] is None and res["verified"] is False and res["ms_per_step"] is None
    assert res["value_is"].startswith("single-list") and "did not finish" in res["value_failed"]
    assert res["modes"]["subtrees"]["verified"] is True and res["modes"]["subtrees"]["value"] > 0
    assert "did not finish" in res["modes"]["single_list"]["error"]


def test_bench_rccl_calls_with_one_rank():
    """The driver's N > 1 form (torch.distributed.run around bench.py, backend "nccl" = RCCL) needs one GPU per
    rank, so a one-GPU box can run it with ONE rank only: IMT_BENCH_FORCE_DIST makes that rank go through the
    process group, the device-tensor all-gather of the subtree roots, the lift, the combine and the reductions
    exactly as eight ranks would."""
    import json
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, IMT_BENCH_FORCE_DIST="1", IMT_BENCH_NO_TRACE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "IMT_BENCH_COLLECTIVE", "IMT_BENCH_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["ranks_seen"] == 1 and res["collective_backend"] == "nccl"
    assert res["verified"] is True and res["value"] > 0
    # the sliced single-list leg issued its all-gathers through RCCL (asynchronous, waited for on the round's stream)
    assert res["modes"]["single_list"]["verified"] is True and res["modes"]["single_list"]["collectives_per_step"] >= 32
    assert res["modes"]["subtrees"]["verified"] is True


def test_bench_says_where_a_hung_collective_stands():
    """What the first real multi-GPU run shows if a collective never completes: bench.py with a
    transport whose 40th all-gather holds its stream for 8 s (IMT_BENCH_SLICED_TRANSPORT=stall; one rank through the N > 1
    code path) and the library's watchdog at 1 s -- the single-list leg ends with IMT_ERR_TIMEOUT, stderr carries the
    world's state (global tick, the first incomplete tick per round slot, the pending collective and its channel), the
    line has `"value": null` with the reason and the subtree leg's figures under `modes` only, the exit status is
    non-zero, all of it well inside the 8 s the collective takes."""
    import json
    import socket
    import subprocess
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, IMT_BENCH_FORCE_DIST="1", IMT_BENCH_NO_TRACE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", IMT_BENCH_COLLECTIVE="gloo",
               IMT_BENCH_DEVICE="0", IMT_BENCH_SLICED_TRANSPORT="stall", IMT_BENCH_LIBRARY_WATCHDOG_S="1", IMT_BENCH_STALL_S="8")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["value"] is None and res["verified"] is False and "ImtError" in res["value_failed"] and "-13" in res["value_failed"]
    assert res["modes"]["subtrees"]["verified"] is True
    for text in ("failed with -13", "collective NOT COMPLETE on channel", "first incomplete: unit tick", "global tick"):
        assert text in r.stderr, r.stderr[-3000:]


def test_bench_second_attempt_in_fresh_workers_after_a_hung_collective():
    """the first multi-GPU run gets a second attempt.  Every launched rank of bench.py is a GPU-free
    supervisor that runs the measurement in a child process; when the single-list leg hangs (here: the stall transport as
    attempt 1, its 40th all-gather -- inside the PREFLIGHT's second short step -- holds its stream for 8 s, the preflight's
    library watchdog at 1 s) the supervisor starts a FRESH child with the next transport of the plan.  The one JSON line
    comes from the attempt that verifies, carries both attempts with the failed one's dump, and the exit status is 0."""
    import json
    import socket
    import subprocess
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, IMT_BENCH_FORCE_DIST="1", IMT_BENCH_NO_TRACE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", IMT_BENCH_COLLECTIVE="gloo",
               IMT_BENCH_DEVICE="0", IMT_BENCH_ATTEMPTS="stall,local", IMT_BENCH_LIBRARY_WATCHDOG_S="1", IMT_BENCH_STALL_S="8")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "IMT_BENCH_SLICED_TRANSPORT", "IMT_BENCH_WORKER"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    at = res["attempts"]
    assert len(at) == 2 and [a["outcome"] for a in at] == ["failed", "verified"], at
    assert at[0]["asked_for"] == "stall" and "-13" in at[0]["why"] and "collective NOT COMPLETE on channel" in at[0]["dump_tail"]
    assert at[0]["preflight"] is None and at[1]["preflight"]["verified"] is True and at[1]["transport"] == "local"
    assert res["value"] > 0 and res["verified"] is True and res["value"] == res["modes"]["single_list"]["value"]
    assert res["value_is"].startswith("single-list")
    # the other leg ran behind the headline leg of the attempt that verified
    assert res["modes"]["subtrees"]["verified"] is True and res["all_modes_verified"] is True
    assert "failed with -13" in r.stderr and "global tick" in r.stderr


@pytest.mark.parametrize("kind", ["die", "corrupt"])
def test_bench_retries_when_one_rank_of_two_fails(kind):
    """Two ranks as separate processes on the one GPU, exactly as typed (`python bench.py --gpus 2`: launcher ->
    torch.distributed.run -> two GPU-free supervisors -> two workers).  In attempt 0 rank 1's worker fails the way a real run
    can: "die" -- the process is gone after its first timed step, rank 0's worker is left waiting for payloads that never
    come and is killed by its own supervisor a few seconds after it hears of the peer's exit (not after the transport's
    60 s); "corrupt" -- one byte of rank 1's last proof is wrong, so the attempt does not verify although every process
    ends by itself.  Attempt 1 runs in FRESH workers with the plan's next layout (every collective on its round's own
    stream, one pool) on the same GPU and verifies; the line carries both attempts; exit status 0."""
    import json
    import subprocess
    env = dict(os.environ, IMT_BENCH_DEVICE="0", IMT_BENCH_COLLECTIVE="gloo", IMT_BENCH_NO_TRACE="1", IMT_BENCH_INJECT=f"0:1:{kind}",
               IMT_BENCH_PEER_GRACE="3", IMT_BENCH_MODE="single-list")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "IMT_BENCH_WORKER", "IMT_BENCH_ATTEMPTS", "IMT_BENCH_SLICED_TRANSPORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    at = res["attempts"]
    assert [a["outcome"] for a in at] == ["failed", "verified"], at
    assert (at[0]["layout"], at[1]["layout"]) == ("pools", "one-pool") and at[1]["transport"] == "ipc"
    assert at[1]["comm_streams"] == 0 and at[1]["pools"] == 0 and at[1]["preflight"]["verified"] is True
    if kind == "die":
        assert at[0]["exit_status"][1] == 17 and at[0]["exit_status"][0] != 0 and at[0]["seconds"] < 60, at[0]
        assert "rank 1: exit status 17" in at[0]["why"]
    else:
        assert "did not verify" in at[0]["why"], at[0]
    assert res["value"] > 0 and res["verified"] is True and res["n_gpus"] == 2 and res["ranks_seen"] == 2


def test_bench_headline_survives_a_hung_subtree_leg():
    """The other leg cannot take the headline down: the single list is measured and verified first; the subtree leg behind
    it gets 50 ms (standing in for a hang in its one collective): the line keeps `value` and `verified`, says what happened
    under `modes.subtrees.error`, `all_modes_verified` is false, exit status 0, one attempt."""
    import json
    import subprocess
    env = dict(os.environ, IMT_BENCH_DEVICE="0", IMT_BENCH_COLLECTIVE="gloo", IMT_BENCH_NO_TRACE="1", IMT_BENCH_SUBTREES_TIMEOUT="0.05")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "IMT_BENCH_WORKER", "IMT_BENCH_ATTEMPTS", "IMT_BENCH_SLICED_TRANSPORT", "IMT_BENCH_MODE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["value"] > 0 and res["verified"] is True and res["all_modes_verified"] is False
    assert res["value"] == res["modes"]["single_list"]["value"] and "did not finish" in res["modes"]["subtrees"]["error"]
    assert len(res["attempts"]) == 1 and res["attempts"][0]["outcome"] == "verified"


def test_bench_looks_once_at_the_other_stream_layout():
    """`python bench.py --gpus 2` with the exploration the RCCL runs have by default (IMT_BENCH_EXPLORE=1 here, on the gloo
    rehearsal): after the first verified attempt the same transport runs once more in the other stream layout (every
    collective on its round's own stream), in fresh workers, the single list alone; `value` is the better of the two
    verified figures and the line shows both; the subtree leg was measured once."""
    import json
    import subprocess
    env = dict(os.environ, IMT_BENCH_DEVICE="0", IMT_BENCH_COLLECTIVE="gloo", IMT_BENCH_NO_TRACE="1", IMT_BENCH_EXPLORE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "IMT_BENCH_WORKER", "IMT_BENCH_ATTEMPTS", "IMT_BENCH_SLICED_TRANSPORT", "IMT_BENCH_MODE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    at = res["attempts"]
    assert len(at) == 2 and all(a["outcome"] == "verified" for a in at), at
    assert (at[0]["layout"], at[1]["layout"]) == ("pools", "one-pool") and at[1]["exploratory"] is True
    assert at[1]["comm_streams"] == 0 and at[0]["comm_streams"] == 4
    best = max(range(2), key=lambda i: at[i]["value"])
    assert res["value"] == at[best]["value"] and res["value_from_attempt"] == at[best]["attempt"]
    assert res["value"] == res["modes"]["single_list"]["value"] and res["verified"] is True
    assert res["modes"]["single_list"]["schedule"]["comm_streams"] == at[best]["comm_streams"]
    assert res["modes"]["subtrees"]["verified"] is True


def test_bench_last_resort_layout_with_rccl_one_rank():
    """the plan's last entry, `rccl:one-comm` (ONE communicator for all round slots, one pool, collectives on one stream in
    issue order), through bench.py's worker with the only RCCL world a one-GPU box can have: one rank."""
    import json
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, IMT_BENCH_FORCE_DIST="1", IMT_BENCH_NO_TRACE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", IMT_BENCH_ATTEMPTS="rccl:one-comm",
               IMT_BENCH_MODE="single-list")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "IMT_BENCH_COLLECTIVE", "IMT_BENCH_DEVICE", "IMT_BENCH_WORKER", "IMT_BENCH_SLICED_TRANSPORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    sch = res["modes"]["single_list"]["schedule"]
    assert res["verified"] is True and sch["transport"] == "rccl" and sch["pools"] == 0 and sch["comm_streams"] == 1, sch
    assert res["attempts"][0]["layout"] == "one-comm" and res["attempts"][0]["preflight"]["verified"] is True
