"""Differential soak of the multi-GPU single-list mode (imt_sliced_* through sliced.SlicedTree: N replicas on this one GPU): random
world sizes, slice sizes, depths, value patterns, lags, flushes in the middle -- every witness of every rank against the
SEQUENTIAL CPU oracle (update_idx_leaf + rebuild, /root/reference/src/indexed_merkle_tree.rs:632-671).  Not part of the
test suite; tests/test_gpu_sliced.py is the short version."""
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import imt_amd  # noqa: E402
import oracle_lib  # noqa: E402

sliced = bench.load_module("sliced")
orc = oracle_lib.load()
rng = random.Random(int(os.environ.get("SOAK_SEED", "2027")))
budget = float(os.environ.get("SOAK_SECONDS", "60"))
t0 = time.time()
cases = total = 0
last_progress = t0
seen = {}
while time.time() - t0 < budget:
    depth = rng.choice([5, 9, 14, 32, 32])
    world = rng.choice([1, 2, 2, 3, 4, 8])
    cap = min(1 << depth, 1 << 12)
    batch = rng.choice([1, 2, 3, 8, 33, 100])
    rounds = rng.randrange(1, 8)
    while 1 + world * batch * rounds > cap:
        rounds -= 1
        if rounds == 0:
            batch, rounds = 1, 1
            if 1 + world > cap:
                world = 1
    n_total = world * batch * rounds
    style = rng.choice(["random", "ascending", "descending", "clustered"])
    if style == "random":
        vals = list({rng.randrange(1, oracle_lib.P) for _ in range(n_total * 2)})[:n_total]
    elif style == "ascending":
        base = rng.randrange(1, 1 << 200); vals = [base + i for i in range(n_total)]
    elif style == "descending":
        base = rng.randrange(1 << 20, 1 << 200); vals = [base - i for i in range(n_total)]
    else:
        centres = [rng.randrange(1 << 100, 1 << 250) for _ in range(5)]
        vals = list({c + rng.randrange(-2000, 2000) for c in centres for _ in range(n_total)})[:n_total]
        rng.shuffle(vals)
    if len(vals) < n_total:
        continue
    oh = orc.sparse_new(depth, cap)
    rows = [orc.sparse_insert(oh, depth, v) for v in vals]
    oroot = orc.sparse_root(oh)
    lag = rng.choice([None, None, 1, 2, 5]) if world > 1 else None
    try:
        w = sliced.SlicedTree(imt_amd, 0, depth, cap, batch, world, n_local=world, lag=lag)
    except imt_amd.ImtError:        # that lag keeps too many rounds in flight at this world / depth
        w = sliced.SlicedTree(imt_amd, 0, depth, cap, batch, world, n_local=world)
    arr = torch.from_numpy(oracle_lib.ints_to_arr(vals)).cuda()
    gb = world * batch
    checked = 0

    def check(r):
        for k in range(world):
            o = {f: v.cpu().numpy() for f, v in w.outputs(r, k).items() if torch.is_tensor(v)}
            for j in range(batch):
                e = rows[r * gb + k * batch + j]
                assert imt_amd.to_int(o["new_root"][j]) == e["new_root"] and imt_amd.to_int(o["interim_root"][j]) == e["interim_root"], \
                    (cases, depth, world, batch, r, k, j)
                assert int(o["low_index"][j]) == e["low"] and int(o["is_largest"][j]) == e["largest"]
                assert (o["low_sib"][:, j] == e["low_proof"]).all() and (o["new_sib"][:, j] == e["new_proof"]).all()
                assert (o["low_leaf"][j] == e["low_leaf"]).all()

    for r in range(rounds):
        w.step(arr[r * gb:(r + 1) * gb])
        if rng.random() < 0.2:
            w.flush()
        while checked <= r - 3:
            for k in range(world):
                w.wait(checked, k)
            check(checked)
            checked += 1
    w.flush()
    while checked < rounds:
        check(checked)
        checked += 1
    for tr in w.trees:
        assert tr.root() == oroot
    w.close()
    orc.sparse_free(oh)
    cases += 1
    total += n_total
    seen[world] = seen.get(world, 0) + 1
    if time.time() - last_progress > 30:          # a silent GPU job looks hung to the runner
        last_progress = time.time()
        print(f"... {cases} runs, {total} insertions, {time.time() - t0:.0f} s", flush=True)
print("sliced soak: %d runs, %d insertions, every witness of every rank and every replica's root equal to the sequential "
      "oracle (%.0f s); runs per world size: %s" % (cases, total, time.time() - t0, dict(sorted(seen.items()))))
