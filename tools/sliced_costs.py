#!/usr/bin/env python3
"""One GPU: what the sliced single-list mode (sliced.py) costs a rank beyond its own hashing, measured piece by piece,
so that the N-GPU step can be priced without N GPUs:
  * imt_itree_slice_prepare for rank 0 of N = 1, 2, 4, 8 (index work for N x 2^16 values, events for 2^16)
  * imt_itree_slice_apply_gathered of one all-gather (N - 1 remote payloads of a level below l0)
  * a LocalWorld run (all N replicas on this GPU, so N x the hashing AND N x the replicated index work on one device)
Usage: python tools/sliced_costs.py [worlds...]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import imt_amd  # noqa: E402

BATCH, DEPTH = 1 << 16, 32
sliced = bench.load_module("sliced")
worlds = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
dev = torch.device("cuda", 0)

for world in worlds:
    rounds = 6
    cap = 1 << (rounds * world * BATCH).bit_length()
    vals = torch.from_numpy(bench.synth_values(rounds * world * BATCH, 0, 1, 77 + world)).to(dev)
    be = sliced.SliceGpuBackend(imt_amd, 0, DEPTH, cap, BATCH)
    lib = imt_amd.lib
    # ---- prepare alone (it blocks until the side stream is done: wall time = GPU time of the index work)
    ts = []
    for r in range(rounds):
        v = vals[r * world * BATCH:(r + 1) * world * BATCH]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sl = be.prepare(v, 0, BATCH, (world - 1) * BATCH, r % 5)
        ts.append((time.perf_counter() - t0) * 1e3)
        # run the slice so that the plan set frees up (single stream, no exchange: timing only)
        st = be.streams[0]
        pay, pay5 = be.make_buffer(be.payload_bytes), be.make_buffer(be.payload_bytes)
        size5 = be.size() - world * BATCH
        for q in range(DEPTH + 1):
            be.unit(sl, q, pay5 if q == 5 else pay, st)     # keep one real payload (level 4: 2^17 pairs) for the apply timing
        st.synchronize()
    print(f"world {world}: slice_prepare (rank 0: own 2^16 + {world - 1} x 2^16 foreign) {np.median(ts[1:]):.2f} ms "
          f"(per round; first {ts[0]:.1f})", flush=True)
    # ---- apply of one gathered level (world - 1 remote payloads below l0)
    if world > 1:
        # the last round's real level-4 payload in every remote slot, applied to a second replica
        g = pay5.repeat(world)
        units = [-1] + [5] * (world - 1)
        st = be.streams[0]
        be2 = sliced.SliceGpuBackend(imt_amd, 0, DEPTH, cap, BATCH)
        size_before = [size5] * world
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        be2.apply_gathered(g, be.payload_bytes, size_before, [BATCH] * world, units, st)
        e0.record(st)
        for _ in range(20):
            be2.apply_gathered(g, be.payload_bytes, size_before, [BATCH] * world, units, st)
        e1.record(st)
        st.synchronize()
        print(f"world {world}: apply of one gathered level ({world - 1} payloads x 2^17 (node, value) pairs) "
              f"{e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
        be2.tree.close(); be2.ctx.close()
    be.tree.close(); be.ctx.close()
    # ---- LocalWorld throughput on this one GPU
    if world <= 4:
        rounds = 8
        cap = 1 << ((rounds + 2) * world * BATCH).bit_length()
        vals = torch.from_numpy(bench.synth_values((rounds + 2) * world * BATCH, 0, 1, 99 + world)).to(dev)
        bes = [sliced.SliceGpuBackend(imt_amd, 0, DEPTH, cap, BATCH) for _ in range(world)]
        w = sliced.LocalWorld(bes)
        gb = world * BATCH
        for r in range(2):
            w.step([vals[r * gb:(r + 1) * gb]] * world)
        w.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(2, rounds + 2):
            w.step([vals[r * gb:(r + 1) * gb]] * world)
        w.flush()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"world {world}: LocalWorld on ONE GPU: {rounds * gb / dt / 1e6:.3f} M insertions/s in total "
              f"({dt / rounds * 1e3:.1f} ms per round of {world} x 2^16; lag {w.sched.lag}, "
              f"{w.tp.collectives} gathers, {w.tp.bytes_moved / 1e9:.2f} GB copied)", flush=True)
        for be in bes:
            be.tree.close(); be.ctx.close()
