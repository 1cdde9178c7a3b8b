#!/usr/bin/env python3
"""Latency of ONE verify_proof (the reference calls it one proof at a time, src/indexed_merkle_tree.rs:397-400) and of
small path batches, one thread per path vs a quad of lanes per path (IMT_OPT_COOP_MAX_EVENTS)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imt_amd
rng = np.random.default_rng(1)
for depth in (3, 8, 32):
    for n in (1, 64, 1024):
        leaf = rng.integers(0, 256, (n, 32), dtype=np.uint8); leaf[:, 31] &= 0x0f
        sib = rng.integers(0, 256, (depth, n, 32), dtype=np.uint8); sib[:, :, 31] &= 0x0f
        idx = rng.integers(0, 1 << min(depth, 30), n).astype(np.uint64)
        row = []
        for coop in (0, 16384):
            c = imt_amd.Context(0)
            c.set_option(imt_amd._ffi.OPT_COOP_MAX_EVENTS, coop)
            c.path_root(leaf, idx, sib, depth)
            t0 = time.perf_counter()
            for _ in range(10):
                c.path_root(leaf, idx, sib, depth)
            row.append((time.perf_counter() - t0) / 10 * 1e3)
            c.close()
        print(f"depth {depth:2d}  n {n:5d}   one thread per path {row[0]:7.3f} ms   quad per path {row[1]:7.3f} ms   ({row[0] / row[1]:.2f}x)")
