"""Secondary rates (device pointers, HIP events): hashes, paths, non-membership (BASELINE config 3),
dense build, bulk load.  Not the headline metric; numbers go to DESIGN.md section 7."""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import imt_amd
from imt_amd import _ffi
import oracle_lib

lib = imt_amd.lib
ctx = imt_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
P = ctypes.c_void_p
D = _ffi.DEVICE_PTRS


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def rnd(shape):
    a = torch.randint(0, 256, shape + (32,), dtype=torch.uint8, device=dev)
    a[..., 31] &= 0x0f
    return a

depth = 32
t = imt_amd.IndexedTree(ctx, depth, 1 << 17)          # the config-2 tree
vals = oracle_lib.synth_values(1 << 16, 0x494D5402)
t.insert_batch(vals, proofs=False)
root = torch.from_numpy(imt_amd.to_bytes(t.root())).to(dev)

n = 1 << 20
cand = rnd((n,)).cpu().numpy()
t0 = time.perf_counter()
low, leaves, sib, largest = t.non_membership_witness(cand)
t_wit = time.perf_counter() - t0
d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in
     dict(low=low.astype(np.int64), leaves=leaves, sib=sib, largest=largest, cand=cand).items()}
fail = torch.empty(n, dtype=torch.uint8, device=dev)
ms = timed(lambda: lib.imt_non_membership_batch(ctx.h, P(root.data_ptr()), P(d["leaves"].data_ptr()), P(d["low"].data_ptr()),
                                                P(d["sib"].data_ptr()), depth, P(d["cand"].data_ptr()),
                                                P(d["largest"].data_ptr()), n, P(fail.data_ptr()), None, D))
assert int(fail.max()) == 0
print(f"non-membership 2^20 items depth 32: {ms:.2f} ms  {n / ms / 1e3:.2f} M items/s  {n * 33 / ms / 1e3:.1f} Mhash/s  "
      f"{n * 1192 / ms / 1e6:.2f} GB/s algorithmic   (host witness prep {t_wit:.2f} s)")

# f3 for the same gadget: the glue rows of verify_non_inclusion for 2^18 of those items, written to HBM
m = 1 << 18
rows = lib.imt_non_inclusion_gadget_rows(depth, 18)
gtr = torch.empty((rows, m, 32), dtype=torch.uint8, device=dev)
sib_m = d["sib"][:, :m].contiguous()
ms = timed(lambda: lib.imt_non_inclusion_gadget_trace_batch(ctx.h, P(d["leaves"].data_ptr()), P(d["low"].data_ptr()), P(sib_m.data_ptr()),
                                                            P(d["cand"].data_ptr()), P(d["largest"].data_ptr()), depth, 18, m,
                                                            P(gtr.data_ptr()), D))
assert int(gtr[-1, :, 0].min()) == 1 and int(gtr[-1, :, 1:].max()) == 0          # last row: low.val < candidate holds
print(f"verify_non_inclusion glue rows (f3) 2^18 items depth 32: {ms:.2f} ms  {m / ms / 1e3:.2f} M items/s  {rows} rows/item  "
      f"{m * rows * 32 / ms / 1e6:.1f} GB/s written  (32 path hashes per item recomputed for dual_mux's inputs)")
del gtr, sib_m

leaf = rnd((n,)); idx = torch.randint(0, 1 << 32, (n,), dtype=torch.int64, device=dev)
out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
ms = timed(lambda: lib.imt_path_root_batch(ctx.h, P(leaf.data_ptr()), P(idx.data_ptr()), P(d["sib"].data_ptr()), depth, n,
                                           P(out.data_ptr()), D))
print(f"path_root 2^20 paths depth 32: {ms:.2f} ms  {n / ms / 1e3:.2f} M paths/s  {n * 32 / ms / 1e3:.1f} Mhash/s")

a = rnd((1 << 21, 2)); o2 = torch.empty((1 << 21, 32), dtype=torch.uint8, device=dev)
ms = timed(lambda: lib.imt_hash2_batch(ctx.h, P(a.data_ptr()), P(o2.data_ptr()), 1 << 21, D))
print(f"hash2 2^21: {ms:.2f} ms  {(1 << 21) / ms / 1e3:.1f} Mhash/s")

lv = rnd((1 << 20,)).cpu().numpy()
t0 = time.perf_counter()
dt = imt_amd.IndexedMerkleTree.new(ctx, lv)
print(f"IndexedMerkleTree.new 2^20 leaves (host pointers, incl. H2D): {(time.perf_counter() - t0) * 1e3:.1f} ms")

t0 = time.perf_counter()
snap = t.snapshot()
t1 = time.perf_counter()
t2 = imt_amd.IndexedTree(ctx, depth, 1 << 17)
t2.load(snap); t2.close()
t2 = imt_amd.IndexedTree(ctx, depth, 1 << 17)
t1b = time.perf_counter()
t2.load(snap)
print(f"imt_itree_get_leaves {snap.shape[0]} leaves to the host: {(t1 - t0) * 1e3:.1f} ms; imt_itree_load of them from the host "
      f"(list check + rebuild on the GPU): {(time.perf_counter() - t1b) * 1e3:.1f} ms")
assert t2.root() == t.root()
