"""Regenerates tests/golden/config4_oracle_digest.json: the CPU oracle (oracle/sparse.c, sequential semantics of
update_idx_leaf + rebuild, /root/reference/src/indexed_merkle_tree.rs:632-671, :715-735) over ALL 2^22 insertions of
BASELINE config 4 on the reference's data structure (ONE depth-32 indexed tree, one sorted list), reduced to digests per
step of 2^19 insertions, so that tests/test_gpu_sliced.py::test_config4_single_list_eight_slices_2pow22 can compare every
interim / new root, low index and flag of the 8-slice world with the oracle's, not only properties.

The values are the test's: bench.synth_values(2^22, 0, 1, 0x494D5404) (numpy PCG64; their sha256 is in the digest, the
test checks it before anything else).

One sequential run is about four hours of one core (2.7 ms of hashing per insertion + the index memmove).  The run is
therefore cut at the step boundaries into 8 SEGMENTS which run side by side: segment s starts from the state after
s * 2^19 insertions, REBUILT from its preimages the way the reference's test rebuilds its tree after every insertion
(orc_sparse_load = hash_nullifier_pre_images :662-671 + IndexedMerkleTree::new src/utils.rs:38-51; the preimages are the
sorted list of the first s * 2^19 values, written down here with numpy), and then inserts its 2^19 values one by one with
orc_sparse_insert.  What makes the cut sound: the root a segment is loaded with must EQUAL the root the segment before
it reached by sequential insertion (a root commits to every leaf preimage, hence to the whole list) -- checked below for
all 7 joints, the run fails otherwise -- and segment 0 starts from the empty tree.

    python tests/golden/make_config4_digest.py [--workers 4] [--scratch /tmp/config4_digest]
"""
import argparse
import ctypes
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

DEPTH, STEPS, PER_STEP, SEED, CAP = 32, 8, 1 << 19, 0x494D5404, 1 << 23
N = STEPS * PER_STEP


def values():
    import bench
    return bench.synth_values(N, 0, 1, SEED)


def preimages_after(vals, k):
    """The list after the first k insertions as leaf preimages [k + 1][3][32]: leaf 0 the sentinel, leaf 1 + i holds
    vals[i]; next_val / next_idx name the next larger value's leaf, 0 / 0 at the largest (:647-657 applied k times)."""
    pre = np.zeros((k + 1, 3, 32), np.uint8)
    if k == 0:
        return pre
    pre[1:, 0] = vals[:k]
    limbs = vals[:k].view("<u8").reshape(k, 4)
    order = np.lexsort((limbs[:, 0], limbs[:, 1], limbs[:, 2], limbs[:, 3]))      # ascending by 256-bit value
    chain = np.concatenate(([0], order + 1)).astype(np.uint64)                   # leaf indices in value order
    nxt = chain[1:]
    pre[chain[:-1], 1] = pre[nxt, 0]
    idx = np.zeros((k, 32), np.uint8)
    idx[:, :8] = nxt.astype("<u8").view(np.uint8).reshape(k, 8)
    pre[chain[:-1], 2] = idx
    return pre


def run_segment(args):
    s, scratch = args
    import oracle_lib
    orc = oracle_lib.load()
    lib = orc.lib
    vals = values()
    h = orc.sparse_new(DEPTH, CAP)
    t0 = time.time()
    rc = orc.sparse_load(h, preimages_after(vals, s * PER_STEP))
    assert rc == 0, ("load", s, rc)
    start_root = orc.sparse_root(h)
    t_load = time.time() - t0
    interim = np.empty((PER_STEP, 32), np.uint8)
    new = np.empty((PER_STEP, 32), np.uint8)
    low = np.empty(PER_STEP, np.uint64)
    largest = np.empty(PER_STEP, np.uint8)
    lo, lg = ctypes.c_uint64(), ctypes.c_int()
    seg = vals[s * PER_STEP:(s + 1) * PER_STEP]
    for i in range(PER_STEP):
        rc = lib.orc_sparse_insert(h, seg[i].ctypes.data_as(ctypes.c_void_p), ctypes.byref(lo), None, ctypes.byref(lg),
                                   interim[i].ctypes.data_as(ctypes.c_void_p), new[i].ctypes.data_as(ctypes.c_void_p), None, None)
        assert rc == 0, (s, i, rc)
        low[i], largest[i] = lo.value, lg.value
        if (i & 0xFFFF) == 0xFFFF:
            print(f"segment {s}: {i + 1} / {PER_STEP} after {time.time() - t0:.0f} s", flush=True)
    end_root = orc.sparse_root(h)
    proofs = {}
    if s == STEPS - 1:
        proofs = {str(i): hashlib.sha256(orc.sparse_proof(h, DEPTH, i).tobytes()).hexdigest() for i in (0, 1, 1234567, N)}
    orc.sparse_free(h)
    np.save(os.path.join(scratch, f"interim_{s}.npy"), interim)
    np.save(os.path.join(scratch, f"new_{s}.npy"), new)
    np.save(os.path.join(scratch, f"low_{s}.npy"), low)
    np.save(os.path.join(scratch, f"largest_{s}.npy"), largest)
    out = dict(step=s, start_root=str(start_root), end_root=str(end_root), load_seconds=round(t_load, 1),
               seconds=round(time.time() - t0, 1), final_proofs=proofs)
    json.dump(out, open(os.path.join(scratch, f"segment_{s}.json"), "w"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--scratch", default="/tmp/config4_digest")
    a = ap.parse_args()
    os.makedirs(a.scratch, exist_ok=True)
    todo = [s for s in range(STEPS) if not os.path.exists(os.path.join(a.scratch, f"segment_{s}.json"))]
    t0 = time.time()
    if todo:
        with mp.get_context("spawn").Pool(a.workers) as pool:
            # longest first: a later segment loads more leaves and moves a longer index
            for r in pool.imap_unordered(run_segment, [(s, a.scratch) for s in sorted(todo, reverse=True)]):
                print("done", r, flush=True)
    segs = [json.load(open(os.path.join(a.scratch, f"segment_{s}.json"))) for s in range(STEPS)]
    import oracle_lib
    orc = oracle_lib.load()
    h = orc.sparse_new(DEPTH, 2)
    empty_root = orc.sparse_root(h)
    orc.sparse_free(h)
    assert int(segs[0]["start_root"]) == empty_root
    for s in range(1, STEPS):        # the joints: the rebuilt state IS the state the sequential run before it ended in
        assert segs[s]["start_root"] == segs[s - 1]["end_root"], f"joint {s}: loaded root differs from the sequential run's"
    vals = values()
    tot = {k: hashlib.sha256() for k in ("interim", "new", "low", "largest")}
    steps = []
    for s in range(STEPS):
        d = {}
        for k in tot:
            arr = np.load(os.path.join(a.scratch, f"{k}_{s}.npy"))
            raw = arr.astype("<u8").tobytes() if k == "low" else arr.tobytes()
            tot[k].update(raw)
            d[k] = hashlib.sha256(raw).hexdigest()
        d["root_after"] = segs[s]["end_root"]
        steps.append(d)
    out = dict(depth=DEPTH, n=N, steps=STEPS, insertions_per_step=PER_STEP, seed=hex(SEED), capacity=CAP,
               values="bench.synth_values(2^22, 0, 1, seed)", sha256_values=hashlib.sha256(vals.tobytes()).hexdigest(),
               provenance="oracle (derived, KAT-anchored; unpinned by the reference); 8 segments of the sequential run, "
                          "each started from orc_sparse_load of the list before it, all 7 joints root-equal",
               sha256_interim_roots=tot["interim"].hexdigest(), sha256_new_roots=tot["new"].hexdigest(),
               sha256_low_index=tot["low"].hexdigest(), sha256_is_largest=tot["largest"].hexdigest(),
               per_step=steps_fmt(steps), final_root=segs[-1]["end_root"], sha256_final_proofs=segs[-1]["final_proofs"],
               joints=[segs[s]["start_root"] for s in range(STEPS)],
               oracle_core_seconds=round(sum(x["seconds"] for x in segs), 1), wall_seconds=round(time.time() - t0, 1))
    path = os.path.join(HERE, "config4_oracle_digest.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


def steps_fmt(steps):
    return [dict(sha256_interim_roots=d["interim"], sha256_new_roots=d["new"], sha256_low_index=d["low"],
                 sha256_is_largest=d["largest"], root_after=d["root_after"]) for d in steps]


if __name__ == "__main__":
    main()
