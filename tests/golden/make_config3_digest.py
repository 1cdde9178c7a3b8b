"""Regenerates tests/golden/config3_oracle_digest.json: the CPU oracle's restatement of verify_non_inclusion
(oracle/indexed.c, /root/reference/src/indexed_merkle_tree.rs:127-229) over BASELINE config 3 at its full size -- 2^20
items against the config-2 tree, a MIXED batch (tests/golden/config3_mix.py: honest items, members, zero, a wrong
is_largest, a forged sibling, a wrong path position, a changed preimage) -- reduced to digests of the fail masks and of
the recomputed roots, and of the honest witnesses themselves (low index, preimage, is_largest, all 32 siblings), so that
the GPU test can compare every one of the 2^20 answers.  About 35 M CPU hashes: some twenty minutes of one core, a few
minutes on eight.      python tests/golden/make_config3_digest.py [workers]"""
import bisect
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
sys.path.insert(0, HERE)
import oracle_lib  # noqa: E402
import config3_mix as C  # noqa: E402


def verify_chunk(args):
    root, cand, low, leaves, sib, largest = args
    orc = oracle_lib.load()
    lib = orc.lib
    n = cand.shape[0]
    fail = np.empty(n, np.uint8)
    roots = np.empty((n, 32), np.uint8)
    helper = np.zeros((C.DEPTH, 32), np.uint8)
    lh = ctypes.create_string_buffer(32)
    rootb = oracle_lib.b32(root)
    for i in range(n):
        idx = int(low[i])
        for l in range(C.DEPTH):
            helper[l, 0] = 1 - ((idx >> l) & 1)
        proof = np.ascontiguousarray(sib[:, i])
        fail[i] = lib.orc_verify_non_inclusion(rootb, leaves[i].ctypes.data_as(ctypes.c_void_p), proof.ctypes.data_as(ctypes.c_void_p),
                                               helper.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(C.DEPTH),
                                               cand[i].ctypes.data_as(ctypes.c_void_p), int(largest[i]), lh,
                                               roots[i].ctypes.data_as(ctypes.c_void_p))
    return fail, roots


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, (os.cpu_count() or 2) - 1)
    t0 = time.time()
    orc = oracle_lib.load()
    lib = orc.lib
    vals = oracle_lib.synth_values(C.N_TREE, C.TREE_SEED)
    h = orc.sparse_new(C.DEPTH, 1 << 17)
    for v in vals:
        assert lib.orc_sparse_insert(h, oracle_lib.b32(v), None, None, None, None, None, None, None) == 0
    root = orc.sparse_root(h)
    print(f"tree built ({time.time() - t0:.0f} s), root {root:#x}", flush=True)
    # ---- the honest witnesses from the oracle's tree: predecessor in the sorted list, its preimage and proof
    order = sorted((v, i + 1) for i, v in enumerate(vals))
    sorted_vals = [0] + [v for v, _ in order]
    sorted_idx = [0] + [i for _, i in order]
    cand = C.candidates()
    cints = oracle_lib.arr_ints(cand)
    low = np.empty(C.N_ITEMS, np.uint64)
    for j, c in enumerate(cints):
        low[j] = sorted_idx[bisect.bisect_left(sorted_vals, c) - 1]
    assert not (set(cints) & set(sorted_vals)), "a candidate is a member: choose another seed"
    uniq, inv = np.unique(low, return_inverse=True)
    pre = np.stack([orc.sparse_preimage(h, int(i)) for i in uniq])              # [u, 3, 32]
    proofs = np.stack([orc.sparse_proof(h, C.DEPTH, int(i)) for i in uniq])     # [u, depth, 32]
    leaves = pre[inv]
    sib = np.ascontiguousarray(proofs[inv].transpose(1, 0, 2))                   # [depth, n, 32]
    largest = (~leaves[:, 1].any(axis=1)).astype(np.uint8)
    orc.sparse_free(h)
    digest = dict(depth=C.DEPTH, n_items=C.N_ITEMS, tree_seed=hex(C.TREE_SEED), candidate_seed=C.CAND_SEED, tree_root=str(root),
                  provenance="oracle (derived, KAT-anchored; unpinned by the reference)",
                  sha256_honest_low_index=hashlib.sha256(low.astype("<u8").tobytes()).hexdigest(),
                  sha256_honest_low_leaf=hashlib.sha256(leaves.tobytes()).hexdigest(),
                  sha256_honest_is_largest=hashlib.sha256(largest.tobytes()).hexdigest(),
                  sha256_honest_siblings=hashlib.sha256(sib.tobytes()).hexdigest())
    cls = C.mix(cand, low, leaves, sib, largest)
    print(f"witnesses built ({time.time() - t0:.0f} s); verifying {C.N_ITEMS} items on {workers} workers", flush=True)
    step = 1 << 13
    jobs = [(root, cand[a:a + step], low[a:a + step], leaves[a:a + step], np.ascontiguousarray(sib[:, a:a + step]), largest[a:a + step])
            for a in range(0, C.N_ITEMS, step)]
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(verify_chunk, jobs, chunksize=1)
    fail = np.concatenate([r[0] for r in res])
    roots = np.concatenate([r[1] for r in res])
    per_class = {}
    for c in sorted(set(cls.tolist())):
        m = fail[cls == c]
        per_class[str(c)] = dict(what=C.CLASSES.get(c, "honest"), items=int(m.size), masks={f"0x{int(k):02x}": int((m == k).sum()) for k in np.unique(m)})
    assert per_class["0"]["masks"] == {"0x00": per_class["0"]["items"]}, "an honest item was refused"
    assert all("0x00" not in v["masks"] for k, v in per_class.items() if k != "0"), "a broken item was accepted"
    digest.update(sha256_fail_masks=hashlib.sha256(fail.tobytes()).hexdigest(), sha256_recomputed_roots=hashlib.sha256(roots.tobytes()).hexdigest(),
                  per_class=per_class, oracle_seconds=round(time.time() - t0, 1), workers=workers)
    path = os.path.join(HERE, "config3_oracle_digest.json")
    json.dump(digest, open(path, "w"), indent=1)
    print("wrote", path, digest["oracle_seconds"], "s")


if __name__ == "__main__":
    main()
