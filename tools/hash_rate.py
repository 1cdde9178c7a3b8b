"""Hash throughput vs batch size (device pointers, HIP events via torch) -- a tuning aid."""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imt_amd
from imt_amd import _ffi
ctx = imt_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
for lg in [int(x) for x in os.environ.get("LGS", "15,16,17,18,19,20,21").split(",")]:
    n = 1 << lg
    a = torch.randint(0, 256, (n, 2, 32), dtype=torch.uint8, device=dev)
    a[:, :, 31] &= 0x0f
    out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    fl = _ffi.DEVICE_PTRS | _ffi.FMT_DEVICE
    for _ in range(2):
        imt_amd.lib.imt_hash2_batch(ctx.h, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), n, fl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(1, (1 << 21) // n)
    e0.record()
    for _ in range(reps):
        imt_amd.lib.imt_hash2_batch(ctx.h, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), n, fl)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"n=2^{lg} {ms:8.3f} ms  {n / ms / 1e3:8.1f} Mhash/s  waves/SIMD={n / 65536:.2f}")
