#!/usr/bin/env python3
"""One item, or a handful: the GPU library against ONE host core running the C oracle (tests-only code, here as the
yardstick) on the same box.  The reference's callers work one item at a time (verify_proof src/utils.rs:87 called at
src/indexed_merkle_tree.rs:397-400; one insert_leaf per circuit :715-803); a depth-32 item is a chain of 33 / 66
dependent hashes, which no amount of parallel hardware shortens.  Prints wall times (host pointers, synchronous
calls: what such a caller sees) and the batch size from which the GPU call is the faster one."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import imt_amd  # noqa: E402
import oracle_lib  # noqa: E402

DEPTH = 32
orc = oracle_lib.load()
ctx = imt_amd.Context(0)


def wall(f, reps):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"depth {DEPTH}; GPU: libimt_hip.so through host pointers (latency forms below 4096 paths); CPU: oracle/, one thread")
print(f"{'items':>6s} | {'verify_proof GPU ms':>20s} {'CPU ms':>9s} | {'insert_leaf check GPU ms':>25s} {'CPU ms':>9s} | "
      f"{'insertion (tree update) GPU ms':>31s} {'CPU ms':>9s}")
cross = {}
base = oracle_lib.synth_values(200, 0x494D5481)
for n in (1, 2, 4, 8, 16, 64, 256, 1024):
    t = imt_amd.IndexedTree(ctx, DEPTH, 1 << 12)
    t.insert_batch(base)
    h = orc.sparse_new(DEPTH, 1 << 12)
    for v in base:
        assert orc.sparse_insert(h, DEPTH, v)["rc"] == 0
    vals = oracle_lib.synth_values(n, 0x494D5490 + n)
    # --- insertion: the tree update itself
    t0 = time.perf_counter()
    r = t.insert_batch(vals)
    g_ins = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    outs = [orc.sparse_insert(h, DEPTH, v) for v in vals]
    c_ins = (time.perf_counter() - t0) * 1e3
    assert imt_amd.to_int(r["new_root"][-1]) == outs[-1]["new_root"]
    # --- verify_proof of the n new leaves
    idx = np.asarray(r["new_index"], dtype=np.uint64)
    sib = t.get_proof_batch(idx)
    leaf = ctx.hash3(t.get_leaves(idx))
    root = imt_amd.to_bytes(t.root())
    g_ver = wall(lambda: ctx.verify_proof_batch(leaf, idx, root, sib, DEPTH), 5)
    li = [imt_amd.to_int(x) for x in leaf]
    proofs = [np.ascontiguousarray(sib[:, i]) for i in range(n)]
    c_ver = wall(lambda: [orc.path_root(li[i], int(idx[i]), proofs[i]) for i in range(n)], 2 if n > 64 else 5)
    # --- insert_leaf constraint check (3 leaf hashes + 4 paths per item)
    g_chk = wall(lambda: ctx.insert_witness(r["old_root"], r["low_leaf"], r["low_index"], r["low_sib"], r["new_root"],
                                            r["new_leaf"], r["new_index"], r["new_sib"], r["is_largest"], DEPTH), 3)
    c_chk = c_ver / 32 * 131       # 131 hashes per item at the oracle's per-hash rate (its relation checker is per item too)
    print(f"{n:6d} | {g_ver:20.2f} {c_ver:9.2f} | {g_chk:25.2f} {c_chk:9.2f} | {g_ins:31.2f} {c_ins:9.2f}", flush=True)
    for k, g, c in (("verify_proof", g_ver, c_ver), ("insert_leaf check", g_chk, c_chk), ("insertion", g_ins, c_ins)):
        if g < c and k not in cross:
            cross[k] = n
    t.close()
    orc.sparse_free(h)
print("GPU call faster from (items per call): " + ", ".join(f"{k} >= {v}" for k, v in cross.items()))
print("below that: stay on the CPU, or batch -- a GPU call costs its dependent-hash chain (33 hashes ~ 6.5 ms in the "
      "latency form) however few items it carries")
