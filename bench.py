#!/usr/bin/env python3
"""bench.py -- indexed-tree insertions/s at depth 32 over bn256::Fr on MI355X.

A step = one batch of 2^16 sequential-semantics insertions (BASELINE.json configs[1]) through
imt_itree_insert_batch: low-leaf search + leaf preimages on the host, all 2 + 2*32 hashes per
insertion on the GPU (level sweep), every per-insertion output written to HBM: old / interim /
new root and both 32-sibling proofs.  Values are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
      N > 1 without WORLD_SIZE in the environment: this process only LAUNCHES N ranks (python -m
      torch.distributed.run, child processes; the launcher itself never touches the GPU), relays rank 0's
      line and exits with their status.
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU: the
      driver's form; --gpus must equal WORLD_SIZE)

N > 1: the value space is partitioned by v mod N; rank g owns the leaf-index range
[g*2^(32-k), (g+1)*2^(32-k)) of the depth-32 tree as an indexed subtree of height 32-k (k = log2 N).
Every step the ranks all-gather their subtree roots (RCCL, 32 bytes each; one step behind the
insertions) and every rank lifts its own witnesses to depth 32 (imt_itree_lift_batch: k more hashes per
root, k more siblings per proof), so an insertion is the same 2 + 2*32 = 66 hashes and the same depth-32
outputs at every N.  Per-GPU work is fixed (weak scaling); there is no other data-path collective.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HBM, algorithmic
bytes of SURVEY.md 8d) and `cpu_baseline` (the C oracle, 1 thread, bounded sample) added, plus a
`valu` object: the path is integer-VALU bound, so that is the roofline that says something.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEPTH = 32
BATCH = 1 << 16
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BYTES_PER_INSERTION = 2320          # 2 paths x (32*32 + 96 + 8 + 32)
HASHES_PER_INSERTION = 66           # 2 + 2*32
BYTES_PER_PATH_LEVEL = 1160.0 / 33  # one event, one level: a path's 1160 B spread over its 33 hashes
MADS_PER_HASH = 2 * 76140           # v_mad_u64_u32 per 2-permutation hash
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_GMADS = 36443.0           # measured v_mad_u64_u32 lane-ops/ns (profiles/r01_valu_rates.txt, 8 waves/SIMD)
# HBM bytes of one k_sweep launch at E = 2^17 events from separate rocprofv3 --pmc passes: 2 x FETCH_SIZE (gfx950
# reports half of 16-B/lane reads, MI355X_MICROARCH.md "HBM") + WRITE_SIZE, counter unit KB, mean over the 99 k_sweep
# dispatches of the run (3 leaf launches, 51 levels below l0, 45 at and above).  bench.py cannot read PMC counters
# itself, so this is a STATIC figure, reported as roofline.traffic_static with its source.  (Round 1, three kernels
# calling a shared hash function: 22.0 MB, of which 4.6 MB call-ABI scratch; profiles/r01_pmc_hbm_traffic.txt.)
PMC_TRAFFIC_SWEEP_LEVEL = {"bytes": int((2 * 3580.1 + 8067.9) * 1024), "fetch_size_kb": 3580.1, "write_size_kb": 8067.9,
                           "source": "profiles/r02_pmc_hbm_traffic.txt", "measured_at_commit": "round-2 k_sweep build"}
TRACE_ROWS = 1208                   # witnesses per 2-input hash (imt_hash_trace_batch)


def synth_values(total, residue, modulus, seed):
    """Distinct random field elements v with 0 < v < p and v % modulus == residue (254-bit draws with
    the top 2 bits cleared, rejected until < p: mirrors src/indexed_merkle_tree.rs:381-386)."""
    rng = np.random.default_rng(seed)
    out, seen = [], set()
    while len(out) < total:
        limbs = rng.integers(0, 1 << 64, size=(total + 1024, 4), dtype=np.uint64)
        for row in limbs:
            v = int(row[0]) | (int(row[1]) << 64) | (int(row[2]) << 128) | ((int(row[3]) & ((1 << 62) - 1)) << 192)
            v = v - (v % modulus) + residue
            if 0 < v < P and v not in seen:
                seen.add(v)
                out.append(v)
                if len(out) == total:
                    break
    return np.frombuffer(b"".join(v.to_bytes(32, "little") for v in out), dtype=np.uint8).reshape(total, 32).copy()


def cpu_baseline(vals, budget_s=15.0):
    """The CPU oracle's sparse depth-32 insertion (oracle/sparse.c) on the first values of the same
    workload, one thread, for about `budget_s` seconds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    orc = oracle_lib.load()
    h = orc.sparse_new(DEPTH, 1 << 17)
    n, t0 = 0, time.perf_counter()
    lib = orc.lib
    low = ctypes.c_uint64()
    while n < vals.shape[0]:
        rc = lib.orc_sparse_insert(h, vals[n].ctypes.data_as(ctypes.c_void_p), ctypes.byref(low), None, None, None,
                                   None, None, None)
        assert rc == 0
        n += 1
        if (n & 63) == 0 and time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    orc.sparse_free(h)
    return {"value": n / dt, "unit": "insertions/s", "cores": 1, "kind": "port",
            "sample": f"first {n} insertions of the same depth-32 workload, C oracle (oracle/sparse.c), {dt:.1f} s"}


def cpu_baseline_all_cores(vals, budget_s=8.0):
    """The same oracle on every host core: T independent depth-32 trees, thread k inserting the values
    k, k+T, ... (the value-partitioned form the multi-GPU bench uses).  Extra information beside the
    one-thread `cpu_baseline` the contract asks for; ctypes releases the GIL during the C call."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    orc = oracle_lib.load()
    lib = orc.lib
    T = max(1, min(os.cpu_count() or 1, 64))
    counts = [0] * T
    t0 = time.perf_counter()

    def work(k):
        h = orc.sparse_new(DEPTH, 1 << 14)
        low = ctypes.c_uint64()
        i = k
        while i < vals.shape[0] and time.perf_counter() - t0 < budget_s:
            rc = lib.orc_sparse_insert(h, vals[i].ctypes.data_as(ctypes.c_void_p), ctypes.byref(low), None, None,
                                       None, None, None, None)
            assert rc == 0
            counts[k] += 1
            i += T
        orc.sparse_free(h)

    th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    n = sum(counts)
    return {"value": n / dt, "unit": "insertions/s", "cores": T, "kind": "port",
            "sample": f"{n} insertions over {T} threads, one depth-32 tree per thread, C oracle, {dt:.1f} s"}


def bench_single_list(args, world, rank, local_rank, dist, backend, ctx, imt_amd):
    """N > 1, IMT_BENCH_MODE=single-list: ONE depth-32 tree (the reference's single sorted list, bit-exact),
    replicated on every rank; a step inserts world x 2^16 values, each rank hashes 1/world of every level and
    the ranks all-gather the level's node versions (sharded.ReplicatedIndexedTree)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("imt_sharded", os.path.join(ROOT, "indexed-merkle-tree-halo2_amd",
                                                                               "sharded.py"))
    sharded = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharded)
    steps_total = args.warmup + args.steps
    gb = BATCH * world
    tree = imt_amd.IndexedTree(ctx, DEPTH, 1 << (steps_total * gb).bit_length())
    rep = sharded.ReplicatedIndexedTree(imt_amd, ctx, tree, world, rank, dist, via_host=(backend != "nccl"))
    vals = torch.from_numpy(synth_values(steps_total * gb, 0, 1, 0x494D5402)).to(torch.device("cuda", local_rank))

    def sync():
        ctx.sync()
        torch.cuda.synchronize()
        dist.barrier()

    for i in range(args.warmup):
        rep.insert_batch(vals[i * gb:(i + 1) * gb])
    sync()
    t0 = time.perf_counter()
    for i in range(args.warmup, steps_total):
        rep.insert_batch(vals[i * gb:(i + 1) * gb])
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=("cuda" if backend == "nccl" else "cpu"))
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        value = args.steps * gb / dt
        print(json.dumps({
            "metric": "indexed-tree insertions/sec at depth=32 (bn256::Fr)", "value": value, "unit": "insertions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 limbs (9 x 29-bit, Montgomery mod p), 64-bit accumulate", "data": "synthetic",
            "config": {"workload": "depth=32, ONE indexed tree, world x 2^16 sequential-semantics insertions per step; "
                                   "every rank returns roots + both proofs of its 2^16 insertions",
                       "batch_per_gpu": BATCH, "depth": DEPTH,
                       "parallelism": f"single sorted list replicated on {world} GPUs, per-level slot-range sharding + "
                                      "all-gather of node versions"},
            "roofline": None, "cpu_baseline": None,
            "valu": {"whole_step_frac": value / world * 66 * MADS_PER_HASH / 1e9 / VALU_PEAK_GMADS}}))
    dist.barrier()
    dist.destroy_process_group()


def launch_ranks(args):
    """`python bench.py --gpus N` typed as is: start the N ranks as CHILD processes (torch.distributed.run) before
    anything in this process has touched the GPU, relay their output, return their status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this host
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def load_sharded():
    import importlib.util
    spec = importlib.util.spec_from_file_location("imt_sharded", os.path.join(ROOT, "indexed-merkle-tree-halo2_amd",
                                                                               "sharded.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1 or args.gpus & (args.gpus - 1):
        raise SystemExit("--gpus must be a power of two (subtrees of equal height)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    # rehearsal switches (one-GPU box): IMT_BENCH_DEVICE pins every rank to one device and
    # IMT_BENCH_COLLECTIVE=gloo runs the root exchange through host memory.  The driver's runs use
    # neither: one rank per GPU, backend "nccl" (= RCCL over xGMI).
    if "IMT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["IMT_BENCH_DEVICE"])
    backend = os.environ.get("IMT_BENCH_COLLECTIVE", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    ranks_seen = 1
    # IMT_BENCH_FORCE_DIST: rehearsal of the N > 1 code path (process group, root all-gather, lift, reductions) with
    # whatever world size the launcher gave, 1 included -- the only way to run the RCCL calls on a one-GPU box
    if world > 1 or os.environ.get("IMT_BENCH_FORCE_DIST"):
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        ranks_seen = dist.get_world_size()

    import imt_amd
    from imt_amd import _ffi
    lib = imt_amd.lib

    mode = os.environ.get("IMT_BENCH_MODE", "subtrees")     # N > 1: "subtrees" (default) or "single-list"
    if world > 1 and mode == "single-list":
        ctx = imt_amd.Context(local_rank)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        return bench_single_list(args, world, rank, local_rank, dist, backend, ctx, imt_amd)
    sharded = load_sharded()
    k = world.bit_length() - 1
    sub_height = DEPTH - k
    steps_total = args.warmup + args.steps
    extra_steps = 2                                                  # kernel-attribution pass after the timed region
    cap = 1 << ((steps_total + extra_steps) * BATCH).bit_length()
    vals_h = synth_values((steps_total + extra_steps) * BATCH, rank, world, 0x494D5402 + rank)
    dev = torch.device("cuda", local_rank)
    vals = torch.from_numpy(vals_h).to(dev)
    out_pinned = os.environ.get("IMT_BENCH_OUT") == "pinned"        # secondary measurement (PCIe note)
    gpu_prep = os.environ.get("IMT_BENCH_PREP", "gpu") == "gpu"     # low-leaf search + event build on the GPU
    pipelined = not os.environ.get("IMT_NO_PIPELINE")
    # Two output sets: the two batches in flight never share rows, and the set of a batch is rewritten only by the
    # batch after next -- by then its lift (N > 1) has run.  The bench does not consume the outputs between steps, so
    # the hash-free output buffers are idle when a batch starts (IMT_INPUTS_READY).
    be = sharded.GpuBackend(imt_amd, local_rank, DEPTH, world, rank, cap, BATCH, pipeline=pipelined, inputs_ready=True,
                            nbuf=2, host_prep=not gpu_prep, pinned_outputs=out_pinned)
    ctx = be.ctx
    tree = sharded.ShardedIndexedTree(be, DEPTH, world, rank, dist, via_host=(backend != "nccl"))
    host_s = [0.0]
    last_slot = [None]

    def step(i, flags=None):
        th = time.perf_counter()
        v = vals[i * BATCH:(i + 1) * BATCH]
        if dist is None or flags is not None:
            last_slot[0] = be.insert(v, flags)
        else:
            tree.step(v)            # inserts batch i, then exchanges roots for and lifts batch i-1 (one step behind)
        host_s[0] += time.perf_counter() - th

    def sync():
        be.sync()
        if dist is not None:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    if dist is not None:
        tree.flush()                # the timed region then holds exactly `steps` inserts, exchanges and lifts
    sync()
    lib.imt_profile_enable(ctx.h, 1)
    host_s[0] = 0.0
    t0 = time.perf_counter()
    for i in range(args.warmup, steps_total):
        step(i)
    if dist is not None:
        last_slot[0] = tree.pending
        tree.flush()                # exchange + lift of the last batch: inside the timed region
    sync()
    dt = time.perf_counter() - t0
    prof = (ctypes.c_double * 12)()
    lib.imt_profile_read(ctx.h, prof)
    lib.imt_profile_enable(ctx.h, 0)

    # ---- verification of what the timed region produced (outside it): the LAST step's outputs, as they lie in
    # HBM, go through the independent witness kernels (imt_insert_witness_batch: 3 leaf hashes + 4 depth-32 paths
    # per insertion, every insert_leaf constraint) with global leaf indices, and its last new_root must be the
    # tree's root.
    o = be.outputs(last_slot[0])
    P_ = lambda x: ctypes.c_void_p(x.data_ptr())
    fail = torch.empty(BATCH, dtype=torch.uint8, device=dev)
    new_index = torch.arange(o["first_new_index"], o["first_new_index"] + BATCH, dtype=torch.int64, device=dev)
    ctx._check(lib.imt_insert_witness_batch(ctx.h, P_(o["old_root"]), P_(o["low_leaf"]), P_(o["low_index"]),
                                            P_(o["low_sib"]), P_(o["new_root"]), P_(o["new_leaf"]), P_(new_index), None,
                                            P_(o["new_sib"]), P_(o["is_largest"]), DEPTH, BATCH, P_(fail), None,
                                            _ffi.DEVICE_PTRS))
    be.sync()
    verified = int(fail.max()) == 0 and bool((o["old_root"][1:] == o["new_root"][:-1]).all())
    if dist is None:
        root_now = torch.from_numpy(imt_amd.to_bytes(be.tree.root()))
        verified = verified and bool((o["new_root"][-1].cpu() == root_now).all())
    elif rank == world - 1:         # the last rank's last insertion closes the step: its new root is the global root
        verified = verified and bool((o["new_root"][-1].cpu() == tree.global_root.cpu()).all())
    if dist is not None:
        vt = torch.tensor([1 if verified else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(vt, op=dist.ReduceOp.MIN)
        verified = bool(vt.item())

    # ---- kernel attribution pass (not part of `value`): two more steps WITHOUT IMT_PIPELINE, so that each
    # kernel has the GPU to itself and its HIP-event duration is a clean roofline input.  In the timed
    # region the hash kernels of up to four consecutive batches share the SIMDs and stretch each other.
    b2b = (ctypes.c_double * 12)()
    extra = 0 if os.environ.get("IMT_BENCH_NO_ATTRIBUTION") else extra_steps
    if extra:
        lib.imt_profile_enable(ctx.h, 1)
        alone = (be.ins_flags & ~_ffi.PIPELINE)
        for i in range(steps_total, steps_total + extra):
            step(i, alone)
        sync()
        lib.imt_profile_read(ctx.h, b2b)
        lib.imt_profile_enable(ctx.h, 0)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    mad_peak = ctypes.c_double(0.0)
    copy_gbps = None
    if rank == 0:
        lib.imt_measure_mad_peak(ctx.h, ctypes.byref(mad_peak))   # this device, this run (devices differ)
        # the HBM ceiling of a plain copy on this box, printed beside the vendor peak
        a = torch.empty(1 << 29, dtype=torch.uint8, device=dev)
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 2 * a.numel() * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
    trace_line = None
    if rank == 0 and not os.environ.get("IMT_BENCH_NO_TRACE"):
        # f1, secondary line: the witness-trace kernel (every new advice value of hash_fix_len_array, 38.7 KB per hash)
        # is the one kernel of this library with a meaningful HBM roofline.  2^18 hashes, rows in halo2curves' in-memory
        # form, row-major; algorithmic bytes = the rows it must deliver (PMC WRITE_SIZE equals them: profiles/).
        nt = 1 << 18
        tin = torch.randint(0, 256, (nt, 2, 32), dtype=torch.uint8, device=dev)
        tin[:, :, 31] &= 0x0f
        tout = torch.empty((TRACE_ROWS, nt, 32), dtype=torch.uint8, device=dev)
        tcall = lambda: ctx._check(lib.imt_hash_trace_batch(ctx.h, ctypes.c_void_p(tin.data_ptr()), 2, nt,
                                                            ctypes.c_void_p(tout.data_ptr()),
                                                            _ffi.DEVICE_PTRS | _ffi.FMT_MONT256))
        tcall()
        tcall()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            tcall()
        e1.record()
        torch.cuda.synchronize()
        tms = e0.elapsed_time(e1) / 5
        tgbps = nt * TRACE_ROWS * 32 / (tms * 1e-3) / 1e9
        trace_line = {"kernel": "k_hash_trace", "bound": "hbm", "hashes_per_launch": nt, "avg_launch_ms": tms,
                      "hashes_per_s": nt / (tms * 1e-3), "achieved": tgbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                      "frac": tgbps / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": nt * TRACE_ROWS * 32,
                      "traffic_static": {"write_bytes": 10135000000, "source": "profiles/r02_pmc_trace_traffic.txt"}}
        del tin, tout
    if rank == 0:
        n_ins = args.steps * BATCH * world
        value = n_ins / dt
        # one hash kernel, k_sweep, in three launch classes (HIP events per class): leaf hashes, the levels below l0
        # (table-driven) and the levels from l0 to the root (every event against the empty subtree)
        names = ["k_sweep[leaves]", "index(k_merge_level)", "k_sweep[level<l0]", "k_sweep[level>=l0]", "k_writeback",
                 "host_prepare"]
        kern = {names[c]: {"ms_total": prof[2 * c], "launches": int(prof[2 * c + 1])} for c in range(6)}
        gpu_ms = sum(v["ms_total"] for n_, v in kern.items() if n_ != "host_prepare")
        # dominant kernel by time: k_sweep; every one of its level launches hashes 2*BATCH events up one level
        sweep = [2, 3]                       # profile classes of the level launches
        pipe_ms = sum(prof[2 * c] for c in sweep) / max(sum(prof[2 * c + 1] for c in sweep), 1)
        alone_ms = (sum(b2b[2 * c] for c in sweep) / sum(b2b[2 * c + 1] for c in sweep)) if b2b[5] else None
        alg_bytes = 2 * BATCH * BYTES_PER_PATH_LEVEL

        def line(ms):
            gbps = alg_bytes / (ms * 1e-3) / 1e9
            hps = 2 * BATCH / (ms * 1e-3)
            return gbps, hps
        # roofline inputs: the kernel ALONE on the GPU (attribution pass).  The pipelined launches of the timed
        # region overlap a hash kernel of the neighbouring batch, so their durations add up to more than the wall
        # time and are reported as a secondary field only.
        ms = alone_ms if alone_ms else pipe_ms
        achieved, hashes_per_s = line(ms)
        pipe_gbps, _ = line(pipe_ms) if pipe_ms > 0 else (0.0, 0.0)
        hashes_per_insertion = 2 + 2 * sub_height + 2 * k
        res = {
            "metric": "indexed-tree insertions/sec at depth=32 (bn256::Fr)", "value": value,
            "unit": "insertions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 limbs (9 x 29-bit, Montgomery mod p), 64-bit accumulate",
            "data": "synthetic", "ranks_seen": ranks_seen, "collective_backend": backend if dist is not None else None,
            "verified": verified,
            "config": {"workload": "depth=32, 2^16 sequential-semantics insertions per step per GPU "
                                   "(BASELINE configs[1]); per insertion: old/interim/new depth-32 root + two 32-sibling "
                                   "proofs written to HBM; values resident in HBM",
                       "batch_per_gpu": BATCH, "depth": DEPTH, "subtree_height_per_gpu": sub_height,
                       "parallelism": "single tree" if world == 1 else
                       f"{world} value-partitioned subtrees by leaf-index range; per step one RCCL all-gather of the "
                       f"subtree roots (one step behind) + lift of every witness to depth 32 on its own rank",
                       "hashes_per_insertion": hashes_per_insertion,
                       "prepare": "gpu (imt_prep.hip)" if gpu_prep else "host",
                       "outputs": "pinned host memory, written by the kernels over PCIe" if out_pinned else "HBM",
                       "verified_how": "last timed step's outputs through imt_insert_witness_batch(depth=32, global "
                                       "indices) + root chain + tree root, after the timed region"},
            "roofline": {"bound": "hbm", "kernel": "k_sweep (level launches)", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "traffic_static": PMC_TRAFFIC_SWEEP_LEVEL,
                         "peak_copy_measured": copy_gbps,
                         "avg_launch_ms": ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "duration_source": ("attribution pass: 2 un-pipelined steps after the timed region, the kernel "
                                             "alone on the GPU" if alone_ms else "timed region (pipelined)"),
                         "pipelined": {"avg_launch_ms": pipe_ms, "achieved": pipe_gbps, "frac": pipe_gbps / HBM_PEAK_GBPS,
                                       "what": "the same kernel inside the timed region, sharing the SIMDs with the hash "
                                               "kernels of the neighbouring batches (up to four in flight)"},
                         "note": "declared HBM per the contract; the kernel is integer-VALU bound, see valu"},
            "valu": {"bound": "v_mad_u64_u32 issue", "kernel": "k_sweep (level launches)",
                     "peak_gmads_measured_now": mad_peak.value,
                     "whole_step_frac_of_measured": (value / world * hashes_per_insertion * MADS_PER_HASH / 1e9 /
                                                     mad_peak.value if mad_peak.value else None),
                     "achieved_gmads": hashes_per_s * MADS_PER_HASH / 1e9, "peak_gmads": VALU_PEAK_GMADS,
                     "frac": hashes_per_s * MADS_PER_HASH / 1e9 / VALU_PEAK_GMADS,
                     "hashes_per_s": hashes_per_s,
                     "whole_step_frac": value / world * hashes_per_insertion * MADS_PER_HASH / 1e9 / VALU_PEAK_GMADS,
                     "note": "kernel figures from the attribution pass (kernel alone); whole_step_frac = all hashes of "
                             "the step / wall time of the timed region"},
            "trace_roofline": trace_line,
            "kernels": kern, "gpu_kernel_ms_per_step": gpu_ms / args.steps,
            "host_call_ms_per_step": host_s[0] / args.steps * 1e3,
            "host_prepare_ms_per_step": kern["host_prepare"]["ms_total"] / args.steps,
            "whole_step_algorithmic_GBps": value * BYTES_PER_INSERTION / 1e9,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(vals_h)
            res["cpu_baseline_all_cores"] = cpu_baseline_all_cores(vals_h)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
