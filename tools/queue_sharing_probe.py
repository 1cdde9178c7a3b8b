#!/usr/bin/env python3
"""How long a tiny launch on an otherwise idle stream waits while other streams have a backlog of hash launches: the
HIP runtime multiplexes streams onto (by default four) hardware queues, and a hardware queue runs what it holds in
submission order -- a stream that shares a queue with a busy one stands behind that stream's WHOLE backlog, not behind
one kernel.  MI355X, ROCm 7: with 1 and with 4 busy streams the 64-hash launch completes when the busy stream has drained
(9.9 / 34.0 ms), with 2 or 3 it completes after one kernel (1.8 / 2.2 ms).  This is why the sliced mode issues one round
PERIOD of ticks per call and no more, and why a step's preparation is enqueued on the new round slot's collective stream
(DESIGN.md 8a)."""
import ctypes, os, sys, time
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import imt_amd
from imt_amd import _ffi
lib = imt_amd.lib
dev = torch.device("cuda", 0)
P = ctypes.c_void_p
def rnd(n):
    a = torch.randint(0, 256, (n, 2, 32), dtype=torch.uint8, device=dev); a[..., 31] &= 0x0f; return a
nbig = 1 << 17
for nstreams in (1, 2, 3, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams + 1)]
    ctxs = [imt_amd.Context(0) for _ in range(nstreams + 1)]
    for c, s in zip(ctxs, streams): c.set_stream(s.cuda_stream)
    ins = [rnd(nbig) for _ in range(nstreams)]
    outs = [torch.empty((nbig, 32), dtype=torch.uint8, device=dev) for _ in range(nstreams)]
    small_in, small_out = rnd(64), torch.empty((64, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    # warm
    for k in range(nstreams): lib.imt_hash2_batch(ctxs[k].h, P(ins[k].data_ptr()), P(outs[k].data_ptr()), nbig, _ffi.DEVICE_PTRS)
    lib.imt_hash2_batch(ctxs[-1].h, P(small_in.data_ptr()), P(small_out.data_ptr()), 64, _ffi.DEVICE_PTRS)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); eA = torch.cuda.Event(enable_timing=True)
    e0.record(streams[0])
    for rep in range(12):
        for k in range(nstreams):
            lib.imt_hash2_batch(ctxs[k].h, P(ins[k].data_ptr()), P(outs[k].data_ptr()), nbig, _ffi.DEVICE_PTRS)
    lib.imt_hash2_batch(ctxs[-1].h, P(small_in.data_ptr()), P(small_out.data_ptr()), 64, _ffi.DEVICE_PTRS)
    e1.record(streams[-1])
    eA.record(streams[0])
    torch.cuda.synchronize()
    print(f"{nstreams} busy stream(s) x 12 launches of 2^17 hashes: stream 0 drains after {e0.elapsed_time(eA):.2f} ms; "
          f"a 64-hash launch on ANOTHER stream, enqueued last, completes after {e0.elapsed_time(e1):.2f} ms", flush=True)
    for c in ctxs: c.close()
