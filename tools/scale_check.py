#!/usr/bin/env python3
"""The timed path on BIG trees: prefill a depth-32 tree with imt_itree_load to 2^k leaves (k from SCALE_SIZES, default
20 24 26 27 28; the snapshot is generated, checked and loaded in HBM -- no host copy of the tree exists), then 50 pipelined batches of 2^16 insertions exactly as bench.py issues them (device pointers,
IMT_PIPELINE | IMT_INPUTS_READY, every witness written), and report the rate, the per-class kernel times
(imt_profile_read) and a correctness check at that size: the last batch's witnesses through imt_insert_witness_batch
(every insert_leaf constraint at depth 32), the root chain, the tree's root.

Above 2^22 leaves every level's sibling read is a random 32-byte access into an array of >= 128 MB and k_merge_level /
k_writeback run for more levels (L0 = ceil(log2(size))): this is where a cliff would show.

Prefill: leaf i holds v_i = (i << 64) + r_i (r_i random 64 bits), so leaf order = value order and the snapshot is
{v_i, v_{i+1}, i+1} without a sort; the inserted values are (j << 64) + r with j uniform over the prefilled leaves, i.e.
every insertion's low leaf is a uniformly random stored leaf -- random paths through the whole stored tree."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import imt_amd  # noqa: E402
from imt_amd import _ffi  # noqa: E402

lib = imt_amd.lib
DEPTH, BATCH = 32, 1 << 16
STEPS, WARM = int(os.environ.get("SCALE_STEPS", "50")), 4
sizes = [int(x) for x in os.environ.get("SCALE_SIZES", "20 24 26 27 28").split()]
dev = torch.device("cuda", 0)
P_ = lambda x: ctypes.c_void_p(x.data_ptr())


def snapshot(M, seed):
    """[M][3][32] uint8 on the device: the sorted linked list of M leaves described above"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    pre = torch.zeros((M, 3, 4), dtype=torch.int64, device=dev)          # 4 x u64 little-endian limbs per element
    r = torch.randint(1, 1 << 62, (M,), dtype=torch.int64, device=dev, generator=g)
    r[0] = 0
    idx = torch.arange(M, dtype=torch.int64, device=dev)
    pre[:, 0, 0], pre[:, 0, 1] = r, idx                   # val = (i << 64) + r_i
    pre[:-1, 1, 0], pre[:-1, 1, 1] = r[1:], idx[1:]       # next_val
    pre[:-1, 2, 0] = idx[1:]                              # next_idx
    return pre.view(torch.uint8).reshape(M, 3, 32)


def new_values(n, M, rng):
    v = np.zeros((n, 4), dtype=np.uint64)
    v[:, 0] = rng.integers(1, 1 << 63, size=n, dtype=np.uint64) | np.uint64(1 << 63)      # never equal to a stored r_i (< 2^63)
    v[:, 1] = rng.integers(0, M, size=n, dtype=np.uint64)
    # distinct inside the run with overwhelming probability; the library refuses a batch with a duplicate anyway
    return v.view(np.uint8).reshape(n, 32)


print(f"steps {STEPS} x 2^16 after {WARM} warm-up batches", flush=True)
for k in sizes:
    M = 1 << k
    rng = np.random.default_rng(1000 + k)
    total = (STEPS + WARM) * BATCH
    cap = 1 << (M + total).bit_length()
    ctx = imt_amd.Context(0)
    tree = imt_amd.IndexedTree(ctx, DEPTH, cap)
    snap = snapshot(M, 1000 + k)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    tree.load_device(snap.data_ptr(), M)                  # returns after the rebuild has been checked (synchronous)
    t2 = time.perf_counter()
    # the checkpoint read back from the device index equals what was loaded; a corrupted one is refused
    back = torch.empty_like(snap)
    tree.snapshot_into(back.data_ptr())
    torch.cuda.synchronize()
    t2b = time.perf_counter()
    same = torch.equal(back, snap)
    del back
    snap[M // 3, 1, 0] ^= 1
    try:
        tree.load_device(snap.data_ptr(), M)
        refused = False
    except ValueError as e:
        refused = f"leaf {M // 3} " in str(e)
    print(f"2^{k} leaves: imt_itree_load {t2 - t1:.2f} s from HBM ({M / (t2 - t1) / 1e6:.0f} M leaves/s), "
          f"imt_itree_get_leaves of all of them {t2b - t2:.2f} s, equal {same}; corrupted snapshot refused {refused}", flush=True)
    del snap
    torch.cuda.empty_cache()
    vals = torch.from_numpy(new_values(total, M, rng)).to(dev)
    u8 = dict(dtype=torch.uint8, device=dev)
    sets = [dict(low_index=torch.empty(BATCH, dtype=torch.int64, device=dev), low_leaf=torch.empty((BATCH, 3, 32), **u8),
                 is_largest=torch.empty(BATCH, **u8), old_root=torch.empty((BATCH, 32), **u8),
                 interim_root=torch.empty((BATCH, 32), **u8), new_root=torch.empty((BATCH, 32), **u8),
                 new_leaf=torch.empty((BATCH, 3, 32), **u8), low_sib=torch.empty((DEPTH, BATCH, 32), **u8),
                 new_sib=torch.empty((DEPTH, BATCH, 32), **u8)) for _ in range(3)]
    structs = [_ffi.InsertOut(**{f: t.data_ptr() for f, t in s.items()}) for s in sets]
    flags = _ffi.DEVICE_PTRS | _ffi.PIPELINE | _ffi.INPUTS_READY
    torch.cuda.synchronize()

    def batch(i, slot):
        ctx._check(lib.imt_itree_insert_batch(tree.h, ctypes.c_void_p(vals.data_ptr() + i * BATCH * 32), BATCH,
                                              ctypes.byref(structs[slot]), flags))

    tw = time.perf_counter()
    for i in range(WARM):
        batch(i, i & 1)
    ctx.sync()
    torch.cuda.synchronize()
    first_ms = (time.perf_counter() - tw) * 1e3
    lib.imt_profile_enable(ctx.h, 1)
    t3 = time.perf_counter()
    for i in range(WARM, WARM + STEPS):
        batch(i, 2 if i == WARM + STEPS - 1 else (i & 1))
    ctx.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t3
    prof = (ctypes.c_double * 12)()
    lib.imt_profile_read(ctx.h, prof)
    lib.imt_profile_enable(ctx.h, 0)
    names = ["k_sweep[leaves]", "k_merge_level", "k_sweep[l<l0]", "k_sweep[l>=l0]", "k_writeback", "host"]
    per = {names[c]: (prof[2 * c] / STEPS, int(prof[2 * c + 1]) // STEPS) for c in range(6)}
    # ---- the last batch, as it lies in HBM, through the independent witness kernels
    o = sets[2]
    first = M + (WARM + STEPS - 1) * BATCH
    new_index = torch.arange(first, first + BATCH, dtype=torch.int64, device=dev)
    fail = torch.empty(BATCH, dtype=torch.uint8, device=dev)
    ctx._check(lib.imt_insert_witness_batch(ctx.h, P_(o["old_root"]), P_(o["low_leaf"]), P_(o["low_index"]), P_(o["low_sib"]),
                                            P_(o["new_root"]), P_(o["new_leaf"]), P_(new_index), None, P_(o["new_sib"]),
                                            P_(o["is_largest"]), DEPTH, BATCH, P_(fail), None, _ffi.DEVICE_PTRS))
    ctx.sync()
    ok = int(fail.max()) == 0 and bool((o["old_root"][1:] == o["new_root"][:-1]).all())
    ok = ok and imt_amd.to_int(o["new_root"][-1].cpu().numpy()) == tree.root()
    spread = int(o["low_index"].max()) - int(o["low_index"].min())
    l0 = (M + total - 1).bit_length()
    print(f"2^{k} leaves prefilled (first {WARM} batches {first_ms:.0f} ms): "
          f"{STEPS * BATCH / dt / 1e6:.3f} M insertions/s, {dt / STEPS * 1e3:.2f} ms per 2^16 batch, L0 = {l0}; "
          f"verified {ok} (last batch: witness kernels + root chain + tree root; low leaves span {spread} positions)", flush=True)
    print("    per batch: " + "; ".join(f"{n_} {ms:.2f} ms / {cnt} launches" for n_, (ms, cnt) in per.items()), flush=True)
    del vals, sets, structs
    tree.close()
    ctx.close()
    torch.cuda.empty_cache()
