#!/usr/bin/env python3
"""Insertion rate and latency vs batch size: depth 32, batches of 2^8 .. 2^16, device pointers,
all witnesses written.  Two columns per size: batches issued back to back without pipelining (latency of ONE batch
= what a caller waiting for its witnesses sees) and pipelined (IMT_PIPELINE, throughput).

  python tools/small_batch_rate.py [min_log2 [max_log2]]"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import imt_amd  # noqa: E402

F = imt_amd._ffi
lib = imt_amd.lib
dev = torch.device("cuda", 0)
DEPTH = 32
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = []
for lg in range(lo, hi + 1):
    bs = 1 << lg
    reps = max(4, min(32, (1 << 18) // bs))
    res = {}
    for mode, flags in (("alone", F.DEVICE_PTRS), ("pipelined", F.DEVICE_PTRS | F.PIPELINE)):
        ctx = imt_amd.Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        if "IMT_COOP_MAX_EVENTS" in os.environ:          # 0 = one thread per hash at every size
            ctx.set_option(F.OPT_COOP_MAX_EVENTS, int(os.environ["IMT_COOP_MAX_EVENTS"]))
        tree = imt_amd.IndexedTree(ctx, DEPTH, 1 << max(18, lg + 6))
        rng = np.random.default_rng(lg)
        raw = rng.integers(0, 256, size=((reps + 2) * bs, 32), dtype=np.uint8)
        raw[:, 31] &= 0x0f
        raw[:, 0] |= 1
        vals = torch.from_numpy(raw).to(dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        sets = [dict(low_index=torch.empty(bs, dtype=torch.int64, device=dev), low_leaf=torch.empty((bs, 3, 32), **u8),
                     is_largest=torch.empty(bs, **u8), old_root=torch.empty((bs, 32), **u8),
                     interim_root=torch.empty((bs, 32), **u8), new_root=torch.empty((bs, 32), **u8),
                     new_leaf=torch.empty((bs, 3, 32), **u8), low_sib=torch.empty((DEPTH, bs, 32), **u8),
                     new_sib=torch.empty((DEPTH, bs, 32), **u8)) for _ in range(2)]
        st = [F.InsertOut(**{k: t.data_ptr() for k, t in s.items()}) for s in sets]

        def run(i):
            rc = lib.imt_itree_insert_batch(tree.h, ctypes.c_void_p(vals.data_ptr() + i * bs * 32), bs, ctypes.byref(st[i & 1]), flags)
            assert rc == 0, lib.imt_last_error(ctx.h)
        run(0); run(1)
        ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2, reps + 2):
            run(i)
            if mode == "alone":
                ctx.sync()                   # the caller waits for this batch's witnesses
        ctx.sync(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res[mode] = dt
        tree.close(); ctx.close()
    rows.append(dict(log2_batch=lg, ms_per_batch_alone=res["alone"] * 1e3, ins_per_s_alone=bs / res["alone"],
                     ms_per_batch_pipelined=res["pipelined"] * 1e3, ins_per_s_pipelined=bs / res["pipelined"]))
    print(f"batch 2^{lg:<2}  alone {res['alone'] * 1e3:8.2f} ms/batch {bs / res['alone'] / 1e6:7.3f} M ins/s   "
          f"pipelined {res['pipelined'] * 1e3:8.2f} ms/batch {bs / res['pipelined'] / 1e6:7.3f} M ins/s", flush=True)
print(json.dumps(rows))
