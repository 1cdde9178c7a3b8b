#!/usr/bin/env python3
"""Where imt_sliced_create puts the world's streams on the runtime's hardware queues, and what it costs: creates `dummies`
extra streams first (a host that made streams of its own: torch, RCCL, an application), then a one-rank world, prints
imt_sliced_info's queue map / placement and -- with a batch argument -- the rate of a short run.

  python tools/placement_check.py [dummies [batch [rounds]]]          GPU_MAX_HW_QUEUES=8 in the environment: eight queues

Used by tests/test_gpu_sliced.py (in a subprocess: GPU_MAX_HW_QUEUES is read when the HIP runtime starts) and by
tools/gpu_round5.sh (the rate with 0 .. 3 dummy streams: profiles/r05_placement_rates.txt)."""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import imt_amd  # noqa: E402

dummies = int(sys.argv[1]) if len(sys.argv) > 1 else 0
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 12
sliced = bench.load_module("sliced")
hip = ctypes.CDLL("libamdhip64.so.7")
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
extra = []
for _ in range(dummies):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    extra.append(s)
n = batch or 64
cap = 1 << ((rounds + 3) * n).bit_length()
t = sliced.SlicedTree(imt_amd, 0, 32, cap, n, 1)
info = t.info()
out = dict(dummies=dummies, hw_queues_env=os.environ.get("GPU_MAX_HW_QUEUES"), placement=info["placement"], hw_queues=info["hw_queues"],
           comm_streams=info["comm_streams"], streams_recreated=info["streams_recreated"], queue_map=info["queue_map"])
vals = torch.from_numpy(bench.synth_values((rounds + 2) * n, 0, 1, 4242)).to("cuda")
for r in range(2):
    t.step(vals[r * n:(r + 1) * n], imt_amd._ffi.INPUTS_READY)
t.flush()
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(2, rounds + 2):
    t.step(vals[r * n:(r + 1) * n], imt_amd._ffi.INPUTS_READY)
t.flush()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out["root"] = hex(t.trees[0].root())
if batch:
    out["M_insertions_per_s"] = round(rounds * n / dt / 1e6, 4)
print(json.dumps(out), flush=True)
t.close()
