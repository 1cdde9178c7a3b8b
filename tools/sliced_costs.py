#!/usr/bin/env python3
"""One GPU: the sliced single-list mode through the C ABI (imt_sliced_step) with all N replicas in this process (the
in-process transport), so N x the hashing AND N x the replicated index work sit on one device: what the library-side
schedule costs the host per step (`host_call_ms`), and the aggregate rate against the one-tree pipeline.
Usage: python tools/sliced_costs.py [worlds...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import imt_amd  # noqa: E402

BATCH, DEPTH = 1 << 16, 32
sliced = bench.load_module("sliced")
worlds = [int(x) for x in sys.argv[1:]] or [1, 2, 4]
dev = torch.device("cuda", 0)
F = imt_amd._ffi

for world in worlds:
    rounds = 12
    cap = 1 << ((rounds + 2) * world * BATCH).bit_length()
    vals = torch.from_numpy(bench.synth_values((rounds + 2) * world * BATCH, 0, 1, 99 + world)).to(dev)
    t = sliced.SlicedTree(imt_amd, 0, DEPTH, cap, BATCH, world, n_local=world)
    gb = world * BATCH
    for r in range(2):
        t.step(vals[r * gb:(r + 1) * gb], F.INPUTS_READY)
    t.flush()
    torch.cuda.synchronize()
    i0 = t.info()
    host = 0.0
    t0 = time.perf_counter()
    for r in range(2, rounds + 2):
        th = time.perf_counter()
        t.step(vals[r * gb:(r + 1) * gb], F.INPUTS_READY)
        host += time.perf_counter() - th
    t.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    i1 = t.info()
    print(f"world {world}: {world} replicas on ONE GPU through imt_sliced_step: {rounds * gb / dt / 1e6:.3f} M insertions/s in total "
          f"({dt / rounds * 1e3:.1f} ms per round of {world} x 2^16; lag {i1['lag']}, "
          f"{(i1['collectives'] - i0['collectives']) / rounds:.0f} gathers and "
          f"{(i1['bytes_gathered'] - i0['bytes_gathered']) / rounds / 1e9:.3f} GB copied per round); "
          f"host inside imt_sliced_step {host / rounds * 1e3:.2f} ms per round for all {world} ranks: "
          f"{(i1['host_issue_ms'] - i0['host_issue_ms']) / rounds:.2f} ms issuing, {(i1['host_wait_ms'] - i0['host_wait_ms']) / rounds:.2f} ms waiting for the GPU",
          flush=True)
    t.close()
