"""One-off scale check: 2^20 insertions into a depth-32 tree as 4 pipelined batches of 2^18, all
per-insertion outputs kept; every insert_leaf constraint re-checked by the witness kernels; final
root against a bulk rebuild from the snapshot.  (Sizes above what the pytest suite uses.)"""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import imt_amd
from imt_amd import _ffi
lib = imt_amd.lib
ctx = imt_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
depth, nb, bs = 32, 4, 1 << 18
t = imt_amd.IndexedTree(ctx, depth, 1 << 21)
rng = np.random.default_rng(5)
raw = rng.integers(0, 256, size=(nb * bs, 32), dtype=np.uint8)
raw[:, 31] &= 0x0f
raw[:, 0] |= 1
vals = torch.from_numpy(raw).to(dev)
P = lambda x: ctypes.c_void_p(x.data_ptr())
outs = []
t0 = time.perf_counter()
for b in range(nb):
    o = dict(low_index=torch.empty(bs, dtype=torch.int64, device=dev), is_largest=torch.empty(bs, dtype=torch.uint8, device=dev),
             low_leaf=torch.empty((bs, 3, 32), dtype=torch.uint8, device=dev), new_leaf=torch.empty((bs, 3, 32), dtype=torch.uint8, device=dev),
             old_root=torch.empty((bs, 32), dtype=torch.uint8, device=dev), interim_root=torch.empty((bs, 32), dtype=torch.uint8, device=dev),
             new_root=torch.empty((bs, 32), dtype=torch.uint8, device=dev),
             low_sib=torch.empty((depth, bs, 32), dtype=torch.uint8, device=dev), new_sib=torch.empty((depth, bs, 32), dtype=torch.uint8, device=dev))
    st = _ffi.InsertOut(**{k: v.data_ptr() for k, v in o.items()})
    rc = lib.imt_itree_insert_batch(t.h, ctypes.c_void_p(vals.data_ptr() + b * bs * 32), bs, ctypes.byref(st),
                                    _ffi.DEVICE_PTRS | _ffi.PIPELINE)
    assert rc == 0, lib.imt_last_error(ctx.h)
    outs.append(o)
ctx.sync(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"inserted {nb * bs} in {dt * 1e3:.1f} ms = {nb * bs / dt / 1e6:.2f} M insertions/s (batches of 2^18)")
prev = None
for b, o in enumerate(outs):
    new_index = torch.arange(1 + b * bs, 1 + (b + 1) * bs, dtype=torch.int64, device=dev)
    fail = torch.empty(bs, dtype=torch.uint8, device=dev)
    rc = lib.imt_insert_witness_batch(ctx.h, P(o["old_root"]), P(o["low_leaf"]), P(o["low_index"]), P(o["low_sib"]),
                                      P(o["new_root"]), P(o["new_leaf"]), P(new_index), None, P(o["new_sib"]),
                                      P(o["is_largest"]), depth, bs, P(fail), None, _ffi.DEVICE_PTRS)
    assert rc == 0
    ctx.sync()
    assert int(fail.max()) == 0, b
    assert bool((o["old_root"][1:] == o["new_root"][:-1]).all())
    if prev is not None:
        assert bool((o["old_root"][0] == prev).all())
    prev = o["new_root"][-1].clone()
print("all", nb * bs, "insert_leaf witnesses satisfied; root chain continuous")
t2 = imt_amd.IndexedTree(ctx, depth, 1 << 21)
t2.load(t.snapshot())
assert t2.root() == t.root() == imt_amd.to_int(prev.cpu().numpy())
print("bulk rebuild from snapshot gives the same root", hex(t.root()))
