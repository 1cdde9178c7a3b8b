#!/usr/bin/env python3
"""f1: device-side rate of imt_hash_trace_batch (every witness of hash_fix_len_array per hash), inputs and trace
resident in HBM.  The kernel writes 1208 x 32 B = 38.7 KB per 2-input hash, so unlike the hash kernels it has a
meaningful HBM roofline: achieved GB/s = hashes/s x 38 656 B (algorithmic bytes = the rows it must deliver).

  python tools/trace_rate.py [log2_n ...]      default 14 16 17
"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import imt_amd  # noqa: E402

F = imt_amd._ffi
lib = imt_amd.lib
ctx = imt_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
ROWS = 1208
HBM_PEAK = 8000.0
out = []
for lg in [int(a) for a in sys.argv[1:]] or [14, 16, 17]:
    n = 1 << lg
    inp = torch.randint(0, 256, (n, 2, 32), dtype=torch.uint8, device=dev)
    inp[:, :, 31] &= 0x0f
    tr = torch.empty((ROWS, n, 32), dtype=torch.uint8, device=dev)
    for name, flags in (("mont256 row-major", F.DEVICE_PTRS | F.FMT_MONT256),
                        ("mont256 item-major", F.DEVICE_PTRS | F.FMT_MONT256 | F.TRACE_ITEM_MAJOR),
                        ("device-format row-major", F.DEVICE_PTRS | F.FMT_DEVICE),
                        ("canonical row-major", F.DEVICE_PTRS | F.FMT_CANONICAL)):
        call = lambda: ctx._check(lib.imt_hash_trace_batch(ctx.h, ctypes.c_void_p(inp.data_ptr()), 2, n,
                                                           ctypes.c_void_p(tr.data_ptr()), flags))
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        rate = n / (ms * 1e-3)
        gbps = rate * ROWS * 32 / 1e9
        out.append(dict(log2_n=lg, variant=name, ms=ms, hashes_per_s=rate, trace_GBps=gbps, hbm_frac=gbps / HBM_PEAK))
        print(f"n=2^{lg:<2} {name:26s} {ms:9.3f} ms  {rate / 1e6:7.2f} Mhash/s  {gbps:7.1f} GB/s written  "
              f"{gbps / HBM_PEAK * 100:5.1f} % of 8 TB/s", flush=True)
    del tr, inp
ctx.sync()
print(json.dumps(out))
