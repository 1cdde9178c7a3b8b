"""One-off differential soak: random batch splits of random insertions, every per-insertion output of
the GPU path against the sequential CPU oracle (about a minute of oracle time).  Not part of the test
suite; the suite's differential tests are the short version of this."""
import os, random, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import imt_amd, oracle_lib

orc = oracle_lib.load()
ctx = imt_amd.Context(0)
rng = random.Random(int(os.environ.get("SOAK_SEED", "2026")))
budget = float(os.environ.get("SOAK_SECONDS", "60"))
t0 = time.time()
total = 0
last_progress = t0
case = 0
snaps = refused = 0
forms = {}
while time.time() - t0 < budget:
    depth = rng.choice([4, 9, 14, 32, 32])
    cap = min(1 << depth, 1 << 13)
    n_total = rng.randrange(1, min(cap - 1, 3000) + 1)
    style = rng.choice(["random", "ascending", "descending", "clustered"])
    if style == "random":
        vals = rng.sample(range(1, oracle_lib.P), n_total) if False else list({rng.randrange(1, oracle_lib.P) for _ in range(n_total)})
    elif style == "ascending":
        base = rng.randrange(1, 1 << 200); vals = [base + i for i in range(n_total)]
    elif style == "descending":
        base = rng.randrange(1 << 20, 1 << 200); vals = [base - i for i in range(n_total)]
    else:
        centres = [rng.randrange(1 << 100, 1 << 250) for _ in range(5)]
        vals = list({c + rng.randrange(-2000, 2000) for c in centres for _ in range(n_total // 5 + 1)})[:n_total]
    rng.shuffle(vals) if style in ("random", "clustered") else None
    n_total = len(vals)
    oh = orc.sparse_new(depth, cap)
    rows = [orc.sparse_insert(oh, depth, v) for v in vals]
    oroot = orc.sparse_root(oh)
    # which form of the hash kernel the batches of this tree run: always a quad of lanes per hash (the default for
    # batches this small), never, or split at 1000 insertions
    coop_max = rng.choice([16384, 0, 2000])
    ctx.set_option(imt_amd._ffi.OPT_COOP_MAX_EVENTS, coop_max)
    forms[coop_max] = forms.get(coop_max, 0) + 1
    t = imt_amd.IndexedTree(ctx, depth, cap)
    i = 0
    while i < n_total:
        b = min(n_total - i, rng.choice([1, 2, 3, 17, 64, 255, 1000, 3000]))
        host_prep = rng.random() < 0.3
        r = t.insert_batch(vals[i:i + b], host_prep=host_prep)
        for j in range(b):
            o = rows[i + j]
            got = {k: imt_amd.to_int(r[k][j]) for k in ("old_root", "interim_root", "new_root")}
            assert got["new_root"] == o["new_root"] and got["interim_root"] == o["interim_root"], (case, depth, style, i + j)
            assert int(r["low_index"][j]) == o["low"] and int(r["is_largest"][j]) == o["largest"], (case, depth, style, i + j)
            if (i + j) % 97 == 0:      # proofs and the rewritten low leaf on a sample
                assert (r["low_sib"][:, j] == o["low_proof"]).all() and (r["new_sib"][:, j] == o["new_proof"]).all()
                assert (r["low_leaf"][j] == o["low_leaf"]).all()
        i += b
        if rng.random() < 0.15:
            # checkpoint here and continue on the reloaded tree (imt_itree_get_leaves from the device index ->
            # imt_itree_load: list check + rebuild on the GPU); the snapshot itself against the oracle's leaves
            snap = t.snapshot(); snaps += 1
            t2 = imt_amd.IndexedTree(ctx, depth, cap)
            t2.load(snap)
            assert t2.root() == t.root() == rows[i - 1]["new_root"], (case, depth, style, i, "resume")
            if i < n_total:
                probe = vals[i]                      # not stored yet: its low leaf must be the oracle's for the next insertion
                assert int(t2.find_low([probe])[0]) == rows[i]["low"], (case, depth, style, i, "find_low")
            if rng.random() < 0.5:                   # one flipped bit anywhere in a next pointer is refused
                bad = snap.copy()
                bad[rng.randrange(i + 1), rng.choice([1, 2]), rng.randrange(4)] ^= 1 << rng.randrange(8)
                try:
                    t2.load(bad); raise AssertionError("corrupted snapshot accepted")
                except ValueError:
                    refused += 1
                assert t2.root() == t.root()
            t.close(); t = t2
    assert t.root() == oroot
    final = t.snapshot()
    for q in rng.sample(range(n_total + 1), min(n_total + 1, 16)):
        assert (final[q] == orc.sparse_preimage(oh, q)).all(), (case, depth, style, q, "leaf")
    orc.sparse_free(oh)
    t.close()
    total += n_total
    case += 1
    if time.time() - last_progress > 30:          # a silent GPU job looks hung to the runner
        last_progress = time.time()
        print(f"... {case} trees, {total} insertions, {time.time() - t0:.0f} s", flush=True)
print("differential soak: %d trees, %d insertions, every root and low index equal to the oracle (%.0f s); "
      "%d checkpoints reloaded mid-stream (root, next low leaf and final leaves equal), %d corrupted snapshots refused; "
      "trees per IMT_OPT_COOP_MAX_EVENTS setting: %s" % (case, total, time.time() - t0, snaps, refused, dict(sorted(forms.items()))))
