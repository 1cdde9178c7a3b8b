#!/usr/bin/env python3
"""bench.py -- indexed-tree insertions/s at depth 32 over bn256::Fr on MI355X.

A step = one batch of 2^16 sequential-semantics insertions (BASELINE.json configs[1]) through
imt_itree_insert_batch: low-leaf search + leaf preimages on the host, all 2 + 2*32 hashes per
insertion on the GPU (level sweep), every per-insertion output written to HBM: old / interim /
new root and both 32-sibling proofs.  Values are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

N > 1: the value space is partitioned by v mod N; rank g owns the leaf-index range
[g*2^(32-k), (g+1)*2^(32-k)) of the depth-32 tree as an indexed subtree of height 32-k (k = log2 N),
and after every step the ranks all-gather their subtree roots (RCCL, 32 bytes each) and hash the top
k levels.  Per-GPU work is fixed (weak scaling); there is no other data-path collective.

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
# HBM bytes of one k_sweep_level launch at E = 2^17 events from the PMC passes in
# profiles/r01_pmc_hbm_traffic.txt: 2 x FETCH_SIZE (gfx950 reports half of 16-B/lane reads,
# MI355X_MICROARCH.md "HBM") + WRITE_SIZE, counter unit KB
PMC_TRAFFIC_SWEEP_LEVEL = int((2 * 4096.2 + 13312.5) * 1024)   # profiles/r01_pmc_hbm_traffic.txt (final build)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if world & (world - 1):
        raise SystemExit("--gpus must be a power of two (subtrees of equal height)")
    # rehearsal switches (one-GPU box): IMT_BENCH_DEVICE pins every rank to one device and
    # IMT_BENCH_COLLECTIVE=gloo runs the root exchange through host memory.  The driver's runs use
    # neither: one rank per GPU, backend "nccl" (= RCCL over xGMI).
    if "IMT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["IMT_BENCH_DEVICE"])
    backend = os.environ.get("IMT_BENCH_COLLECTIVE", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import imt_amd
    from imt_amd import _ffi
    lib = imt_amd.lib
    ctx = imt_amd.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    mode = os.environ.get("IMT_BENCH_MODE", "subtrees")     # N > 1: "subtrees" (default) or "single-list"
    if world > 1 and mode == "single-list":
        return bench_single_list(args, world, rank, local_rank, dist, backend, ctx, imt_amd)
    k = world.bit_length() - 1
    depth = DEPTH - k
    steps_total = args.warmup + args.steps
    cap = 1 << ((steps_total + 2) * BATCH).bit_length()
    tree = imt_amd.IndexedTree(ctx, depth, cap)
    vals_h = synth_values((steps_total + 2) * BATCH, rank, world, 0x494D5402 + rank)
    dev = torch.device("cuda", local_rank)
    vals = torch.from_numpy(vals_h).to(dev)

    u8 = dict(dtype=torch.uint8, device=dev)
    out_pinned = os.environ.get("IMT_BENCH_OUT") == "pinned"
    if out_pinned:
        # secondary measurement (PCIe note): every per-insertion output lands in pinned HOST
        # memory, written by the kernels over PCIe (hipHostMalloc memory is device-addressable), which
        # is what a host-language caller that wants the witnesses in its own memory would do.
        u8o = dict(dtype=torch.uint8, device="cpu", pin_memory=True)
        i64o = dict(dtype=torch.int64, device="cpu", pin_memory=True)
    else:
        u8o, i64o = u8, dict(dtype=torch.int64, device=dev)
    bufs = dict(low_index=torch.empty(BATCH, **i64o),
                low_leaf=torch.empty((BATCH, 3, 32), **u8o), is_largest=torch.empty(BATCH, **u8o),
                old_root=torch.empty((BATCH, 32), **u8o), interim_root=torch.empty((BATCH, 32), **u8o),
                new_root=torch.empty((BATCH, 32), **u8o), new_leaf=torch.empty((BATCH, 3, 32), **u8o),
                low_sib=torch.empty((depth, BATCH, 32), **u8o), new_sib=torch.empty((depth, BATCH, 32), **u8o))
    out = _ffi.InsertOut(**{name: t.data_ptr() for name, t in bufs.items()})
    flags = _ffi.DEVICE_PTRS | _ffi.FMT_CANONICAL
    ins_flags = flags | (0 if os.environ.get("IMT_NO_PIPELINE") else _ffi.PIPELINE)
    gpu_prep = os.environ.get("IMT_BENCH_PREP", "gpu") == "gpu"     # low-leaf search + event build on the GPU
    if not gpu_prep:
        ins_flags |= _ffi.HOST_PREP
    roots_all = torch.empty((world, 32), **u8)
    root_buf = torch.empty(32, **u8)
    top_root = torch.empty(32, **u8)

    host_s = [0.0]

    def step(i):
        nonlocal ins_flags
        th = time.perf_counter()
        rc = lib.imt_itree_insert_batch(tree.h, ctypes.c_void_p(vals.data_ptr() + i * BATCH * 32), BATCH,
                                        ctypes.byref(out), ins_flags)
        host_s[0] += time.perf_counter() - th
        if rc != 0:
            raise RuntimeError(f"imt_itree_insert_batch: {rc} {lib.imt_last_error(ctx.h).decode()}")
        if world > 1 and i > 0:
            # the path's one exchange: subtree roots, then the top k levels on every rank.  It runs one
            # step behind (the root after batch i-1) so that it waits for a finished batch instead of
            # stalling the two in flight; the last batch's root is exchanged after the loop.
            exchange(1)

    def exchange(lag):
        ctx._check(lib.imt_itree_root_lagged(tree.h, lag, ctypes.c_void_p(root_buf.data_ptr()), flags))
        if backend == "nccl":
            dist.all_gather_into_tensor(roots_all, root_buf)
        else:   # rehearsal only
            parts = [torch.empty(32, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, root_buf.cpu())
            roots_all.copy_(torch.stack(parts))
        ctx._check(lib.imt_combine_subtree_roots(ctx.h, ctypes.c_void_p(roots_all.data_ptr()), world, depth, DEPTH,
                                                 ctypes.c_void_p(top_root.data_ptr()), flags))

    def sync():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    sync()
    lib.imt_profile_enable(ctx.h, 1)
    host_s[0] = 0.0
    t0 = time.perf_counter()
    for i in range(args.warmup, steps_total):
        step(i)
    if world > 1:
        exchange(0)
    sync()
    dt = time.perf_counter() - t0
    prof = (ctypes.c_double * 12)()
    lib.imt_profile_read(ctx.h, prof)
    # Kernel attribution pass (not part of `value`): two more steps WITHOUT IMT_PIPELINE, so that each
    # kernel has the GPU to itself and its HIP-event duration is a clean roofline input.  In the timed
    # region two hash kernels of consecutive batches share the SIMDs and stretch each other.
    b2b = (ctypes.c_double * 12)()
    extra = 2 if steps_total + 2 <= vals.shape[0] // BATCH else 0
    if os.environ.get("IMT_BENCH_NO_ATTRIBUTION"):
        extra = 0       # profiler runs: every k_sweep_level launch of the process is then a pipelined one
    if extra:
        saved = ins_flags
        ins_flags = flags | (0 if gpu_prep else _ffi.HOST_PREP)
        for i in range(steps_total, steps_total + extra):
            step(i)
        sync()
        lib.imt_profile_read(ctx.h, b2b)
        ins_flags = saved
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
    if rank == 0:
        n_ins = args.steps * BATCH * world
        value = n_ins / dt
        names = ["k_sweep_leaves", "index(k_merge_level)", "k_sweep_level", "k_sweep_top", "k_writeback", "host_prepare"]
        kern = {names[c]: {"ms_total": prof[2 * c], "launches": int(prof[2 * c + 1])} for c in range(6)}
        gpu_ms = sum(v["ms_total"] for n_, v in kern.items() if n_ != "host_prepare")
        # dominant kernel by time: one k_sweep_level launch hashes 2*BATCH events up one level
        lv = kern["k_sweep_level"]
        avg_ms = lv["ms_total"] / max(lv["launches"], 1)
        b2b_avg_ms = b2b[4] / b2b[5] if b2b[5] else None
        alg_bytes = 2 * BATCH * BYTES_PER_PATH_LEVEL
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        hashes_per_s = 2 * BATCH / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
        res = {
            "metric": "indexed-tree insertions/sec at depth=32 (bn256::Fr)", "value": value,
            "unit": "insertions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 limbs (9 x 29-bit, Montgomery mod p), 64-bit accumulate",
            "data": "synthetic",
            "config": {"workload": "depth=32, 2^16 sequential-semantics insertions per step per GPU "
                                   "(BASELINE configs[1]); per insertion: old/interim/new root + two 32-sibling "
                                   "proofs written to HBM; values resident in HBM",
                       "batch_per_gpu": BATCH, "depth": DEPTH, "subtree_height_per_gpu": depth,
                       "parallelism": "single tree" if world == 1 else
                       f"{world} value-partitioned subtrees by leaf-index range + RCCL all-gather of subtree roots per step",
                       "hashes_per_insertion": 2 + 2 * depth,
                       "prepare": "gpu (imt_prep.hip)" if gpu_prep else "host",
                       "outputs": "pinned host memory, written by the kernels over PCIe" if out_pinned else "HBM"},
            "roofline": {"bound": "hbm", "kernel": "k_sweep_level", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": PMC_TRAFFIC_SWEEP_LEVEL,
                         "peak_copy_measured": copy_gbps,
                         "traffic_source": "profiles/r01_pmc_hbm_traffic.txt (separate --pmc passes; value arrays, "
                                           "index tables, the proof store and 36 B/hash of call-ABI stack are counted, "
                                           "the 35 B/hash algorithmic figure counts only path inputs)",
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "back_to_back": None if not b2b_avg_ms else {
                             "avg_launch_ms": b2b_avg_ms, "achieved": alg_bytes / (b2b_avg_ms * 1e-3) / 1e9,
                             "frac": alg_bytes / (b2b_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "valu_frac": 2 * BATCH / (b2b_avg_ms * 1e-3) * MADS_PER_HASH / 1e9 / VALU_PEAK_GMADS,
                             "what": "2 extra un-pipelined steps after the timed region: the kernel alone on the GPU"},
                         "note": "declared HBM per the contract; the kernel is integer-VALU bound, see valu"},
            "valu": {"bound": "v_mad_u64_u32 issue", "kernel": "k_sweep_level",
                     "peak_gmads_measured_now": mad_peak.value,
                     "whole_step_frac_of_measured": (value / world * (2 + 2 * depth) * MADS_PER_HASH / 1e9 / mad_peak.value
                                                     if mad_peak.value else None),
                     "achieved_gmads": hashes_per_s * MADS_PER_HASH / 1e9, "peak_gmads": VALU_PEAK_GMADS,
                     "frac": hashes_per_s * MADS_PER_HASH / 1e9 / VALU_PEAK_GMADS,
                     "hashes_per_s": hashes_per_s,
                     "whole_step_frac": value / world * (2 + 2 * depth) * MADS_PER_HASH / 1e9 / VALU_PEAK_GMADS,
                     "note": "per-kernel figures include time shared with the pipelined k_sweep_top of the "
                             "previous batch; whole_step_frac = all hashes of the step / wall time"},
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
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
