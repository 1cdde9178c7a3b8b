#!/usr/bin/env python3
"""ONE rank of an N-GPU single-list run, alone on one GPU -- a TIMING emulation, not a correctness run.

What cannot be measured on a one-GPU box is what a rank does when it has a GPU to itself under the N-rank schedule: it
prepares the index for N x 2^16 values per step, hashes ITS slice (33 launches of 2^16 hashes), packs its write-backs,
applies the other N - 1 slices' payloads, and has only  33 / (N lag)  of its own launches in flight at a time (2.06 at
N = 8, lag 2, against 3-4 in the one-GPU pipeline) -- the wave count per SIMD that sets the issue rate.  All of that is
local work; only the collectives involve peers.  Here rank `g` of `world` runs through the real library
(imt_sliced_create with world = N, first_rank = g, one local rank, imt_sliced_step on all N x 2^16 values of a step) over
a CUSTOM transport (imt_transport_custom_create) whose all-gather is a model: wait  latency + bytes / link_rate  on the
collective's stream, then fill every peer's slot of the receive buffer with this rank's own payload (device-to-device
copies; the apply kernel clamps counts and node indices, so a foreign slot holding our payload is memory-safe and costs
what a real one costs).  The tree that results is NOT the N-GPU tree (the peers' write-backs are not theirs), so nothing
is verified here -- bit-exactness of the mode is what tests/test_gpu_sliced.py and tools/sliced_soak.py establish.

Output: insertions/s of the one rank, and N x that = what N such GPUs would deliver if every collective took the
modelled time.  The link model is a parameter, not a measurement: xGMI point-to-point, every peer's slot over its own
link (direct all-gather), EMU_LINK_GBPS per direction per peer (default 48 = 7 links x 153 GB/s per GPU / 7 / 2 x 0.9,
rounded down) and EMU_LATENCY_US (default 40) per collective; EMU_LINK_GBPS=0 means free collectives.

Usage: python tools/rank_emulation.py [world ...]      (default 2 4 8; EMU_RANKS="first last" by default; EMU_LAG
overrides the schedule's lag)"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import imt_amd  # noqa: E402

BATCH, DEPTH = 1 << 16, 32
ROUNDS = int(os.environ.get("EMU_ROUNDS", "20"))
WARM = 4
sliced = bench.load_module("sliced")
F, lib = imt_amd._ffi, imt_amd.lib
dev = torch.device("cuda", 0)
hip = ctypes.CDLL("libamdhip64.so.7")       # the HIP runtime this process already holds
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
CLOCK_HZ = torch.cuda.get_device_properties(0).clock_rate * 1e3 if hasattr(torch.cuda.get_device_properties(0), "clock_rate") else 2.4e9


class _Raw:
    """a device pointer as something torch.as_tensor takes (zero copy)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


# How the slots are filled.  "memcpy" (default): one hipMemcpyAsync per slot, eight short full-width blit kernels per
# collective at N = 8 -- what the queue model's emulated transport describes and was calibrated with.  "fused": ONE
# broadcast copy kernel for all slots (a torch elementwise copy of 38 MB): cheaper for the host, but one fat low-priority
# kernel competes with the hashing -- 2.2 - 2.5 against 2.6 - 2.7 M/s at N = 8 -- and without the modelled link time it
# starts the moment the pack ends (then "free collectives" come out SLOWER than modelled links).
# "rccl": ONE kernel of RCCL's shape per collective (tools/microbench/emu_gather.hip: EMU_GATHER_WGS workgroups of 512 lanes
# that need wave slots beside the resident hash kernels, poll a flag until the modelled link time has passed, then copy)
# instead of a one-wave sleep + world blits: what a real ncclAllGather asks of the device.
FILL = os.environ.get("EMU_FILL", "memcpy")
GATHER_WGS = int(os.environ.get("EMU_GATHER_WGS", "28"))
emu = None
if FILL == "rccl":
    import subprocess
    so = os.path.join(ROOT, "tools", "microbench", "libemu_gather.so")
    src = os.path.join(ROOT, "tools", "microbench", "emu_gather.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, src], check=True)
    emu = ctypes.CDLL(so)
    emu.emu_gather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_double, ctypes.c_int]
    emu.emu_gather_spreads.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(ctypes.c_float)]
    emu.emu_gather_spreads.restype = ctypes.c_uint
    torch.cuda.set_device(0)
    assert emu.emu_gather_init() == 0


class ModelTransport:
    """all-gather = modelled wait + own payload into every slot"""

    def __init__(self, world, rank, link_gbps, latency_us):
        self.world, self.rank, self.link, self.lat = world, rank, link_gbps, latency_us
        self.calls = 0
        self.bytes = 0
        self.model_ms = 0.0
        self.streams = {}
        self.ops = F.TransportOps(None, F.TransportOps.ALL_GATHER(self.all_gather), F.TransportOps.DESTROY())
        self.h = ctypes.c_void_p()
        assert lib.imt_transport_custom_create(ctypes.byref(self.ops), ctypes.byref(self.h)) == 0

    def all_gather(self, self_, channel, buffer, send, recv, nbytes, stream):
        try:
            self.calls += 1
            self.bytes += nbytes * (self.world - 1)
            if FILL == "rccl":
                # EMU_EXTRA_WAIT_US: every collective completes this much LATER than latency + bytes / link rate -- the
                # slowest of the N - 1 peers' kernels got its wave slots that much after this rank's did
                us = (self.lat + nbytes / (self.link * 1e3) if self.link > 0 else 0.0) + float(os.environ.get("EMU_EXTRA_WAIT_US", "0"))
                self.model_ms += us * 1e-3
                return 0 if emu.emu_gather(stream, recv, send, nbytes, self.world, us, GATHER_WGS) == 0 else F.ERR["HIP"]
            if self.link > 0:
                us = self.lat + nbytes / (self.link * 1e3)             # every peer's slot arrives over its own link
                self.model_ms += us * 1e-3
                ext = self.streams.get(stream)
                if ext is None:
                    ext = self.streams[stream] = torch.cuda.ExternalStream(stream, device=dev)
                with torch.cuda.stream(ext):
                    torch.cuda._sleep(int(us * 1e-6 * CLOCK_HZ))
            if FILL == "memcpy":
                for k in range(self.world):
                    if hip.hipMemcpyAsync(recv + k * nbytes, send, nbytes, 3, stream):
                        return F.ERR["HIP"]
                return 0
            ext = self.streams.get(stream)
            if ext is None:
                ext = self.streams[stream] = torch.cuda.ExternalStream(stream, device=dev)
            with torch.cuda.stream(ext):
                src = torch.as_tensor(_Raw(send, nbytes), device=dev)
                dst = torch.as_tensor(_Raw(recv, nbytes * self.world), device=dev).view(self.world, nbytes)
                dst.copy_(src.expand(self.world, nbytes))
            return 0
        except Exception as e:                                         # never let an exception cross the C boundary
            print("model transport:", repr(e), flush=True)
            return F.ERR["INTERNAL"]


def one_rank(world, rank, link, lat):
    steps = ROUNDS + WARM
    cap = 1 << (steps * world * BATCH + 1).bit_length()
    vals = torch.from_numpy(bench.synth_values(steps * world * BATCH, 0, 1, 7000 + world)).to(dev)
    tp = ModelTransport(world, rank, link, lat)
    t = sliced.SlicedTree(imt_amd, 0, DEPTH, cap, BATCH, world, first_rank=rank, n_local=1, transport=tp.h,
                          lag=int(os.environ["EMU_LAG"]) if os.environ.get("EMU_LAG") else None)
    gb = world * BATCH
    for r in range(WARM):
        t.step(vals[r * gb:(r + 1) * gb], F.INPUTS_READY)
    t.flush()
    torch.cuda.synchronize()
    c0, m0 = tp.calls, tp.model_ms
    g0 = emu.emu_gather_spreads(0, 0, None) if emu is not None else 0
    i0 = t.info()
    t0 = time.perf_counter()
    for r in range(WARM, steps):
        t.step(vals[r * gb:(r + 1) * gb], F.INPUTS_READY)
    t.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    i1 = t.info()
    info = dict(rate=ROUNDS * BATCH / dt, ms=dt / ROUNDS * 1e3, lag=i1["lag"], gathers=(tp.calls - c0) / ROUNDS,
                model_ms=(tp.model_ms - m0) / ROUNDS, gb=(i1["bytes_gathered"] - i0["bytes_gathered"]) / ROUNDS / 1e9,
                issue=(i1["host_issue_ms"] - i0["host_issue_ms"]) / ROUNDS, wait=(i1["host_wait_ms"] - i0["host_wait_ms"]) / ROUNDS)
    if emu is not None:          # how long the collectives' kernels waited for their wave slots (last workgroup's start - first's)
        n = min(tp.calls - c0, 1 << 16)
        buf = (ctypes.c_float * n)()
        emu.emu_gather_spreads(g0, n, buf)
        sp = sorted(buf)
        info["spread"] = dict(n=n, p50=sp[n // 2], p90=sp[int(n * 0.9)], p99=sp[int(n * 0.99)], max=sp[-1])
    t.close()
    del vals
    torch.cuda.empty_cache()
    return info


def main():
    worlds = [int(x) for x in sys.argv[1:]] or [2, 4, 8]
    link = float(os.environ.get("EMU_LINK_GBPS", "48"))
    lat = float(os.environ.get("EMU_LATENCY_US", "40"))
    print(f"one rank of an N-rank single-list run alone on the GPU, {ROUNDS} timed steps of N x 2^16 insertions after {WARM}; "
          f"collectives modelled as {lat:.0f} us + bytes / {link:.0f} GB/s per peer link (EMU_LINK_GBPS=0: free); fill = {FILL}"
          + (f" ({GATHER_WGS} workgroups x 512 lanes per collective)" if FILL == "rccl" else ""), flush=True)
    for world in worlds:
        which = os.environ.get("EMU_RANKS", "first last").split()
        ranks = sorted({0 if w == "first" else world - 1 if w == "last" else int(w) for w in which})
        for rank in ranks:
            for lk in ([0.0, link] if link > 0 else [0.0]):
                r = one_rank(world, rank, lk, lat)
                print(f"N = {world} rank {rank} lag {r['lag']} {'free collectives' if lk == 0 else f'{lk:.0f} GB/s links'}: "
                      f"{r['rate'] / 1e6:.3f} M insertions/s per rank ({r['ms']:.2f} ms per step) -> x {world} = "
                      f"{world * r['rate'] / 1e6:.2f} M/s; {r['gathers']:.0f} all-gathers receiving {r['gb']:.3f} GB per step"
                      f"{'' if lk == 0 else ', modelled at %.2f ms of link time per step in total' % r['model_ms']}; "
                      f"host inside imt_sliced_step: {r['issue']:.2f} ms issuing + {r['wait']:.2f} ms waiting for the step's "
                      f"value check per step"
                      + ("" if "spread" not in r else "; the collectives' kernels had ALL their workgroups running p50 %.0f / p90 %.0f / p99 %.0f / max %.0f us "
                         "after the first (%d collectives)" % (r["spread"]["p50"], r["spread"]["p90"], r["spread"]["p99"], r["spread"]["max"], r["spread"]["n"])),
                      flush=True)


if __name__ == "__main__":
    main()
