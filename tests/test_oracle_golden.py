"""CPU: pins the oracle (oracle/) to the reference's own known answers and to the fixtures.

Reference pins for this path: ONE absolute KAT, the zero-leaf hash literal at
src/indexed_merkle_tree.rs:247-250, plus the relative pins of test_insert_leaf (:360-596) and
test_insert_leaf_multiple_round (:679-803): native tree == circuit values, proofs verify,
every insert_leaf relation holds."""
import json
import os
import random

import numpy as np

import oracle_lib
from oracle_lib import P, KAT_ZERO, ints_to_arr

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))


def test_reference_kat_zero_leaf(oracle):
    assert oracle.hash([0, 0, 0]) == KAT_ZERO


def test_golden_vectors(oracle):
    n = 0
    for e in GOLD["entries"]:
        ins = [int(x) for x in e["in"]]
        if e["kind"] in ("hash2", "hash3"):
            assert oracle.hash(ins) == int(e["out"]), e
        elif e["kind"] == "permute":
            assert oracle.permute(ins) == [int(x) for x in e["out"]]
        elif e["kind"] == "permute_lane0":          # public circomlib known answers
            assert oracle.permute(ins)[0] == int(e["out"]), e
        elif e["kind"] == "empty_root":
            assert int.from_bytes(oracle.zero_hashes(ins[0])[ins[0]].tobytes(), "little") == int(e["out"])
        n += 1
    assert n > 50


def test_field_constants(oracle):
    import ctypes
    p = (ctypes.c_uint64 * 4)(); r = (ctypes.c_uint64 * 4)(); r2 = (ctypes.c_uint64 * 4)(); inv = ctypes.c_uint64()
    oracle.lib.ofr_raw_constants(p, r, r2, ctypes.byref(inv))
    as_int = lambda a: sum(int(a[i]) << (64 * i) for i in range(4))
    assert as_int(p) == P
    assert as_int(r) == 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb
    assert as_int(r2) == 0x0216d0b17f4e44a58c49833d53bb808553fe3ab1e35c59e31bb8e645ae216da7
    assert inv.value == 0xc2e1f593efffffff


def test_poseidon_constants_spot(oracle):
    import ctypes
    rc = ctypes.create_string_buffer(195 * 32); mds = ctypes.create_string_buffer(9 * 32)
    oracle.lib.orc_poseidon_params(rc, mds)
    assert int.from_bytes(rc.raw[:32], "little") == 0x0ee9a592ba9a9518d05986d656f40c2114c4993c11bb29938d21d47304cd8e6e
    assert int.from_bytes(mds.raw[:32], "little") == 0x109b7f411ba0e4c9b2b70caf5c36a7b194be7c11ad24378bfedb68592ba8118b


def test_dense_tree_errors_and_shape(oracle):
    # src/utils.rs:24-36 and the :45 panic
    assert oracle.tree_new(np.zeros((0, 32), np.uint8))[0] == -1
    assert oracle.tree_new(ints_to_arr([1, 2, 3]))[0] == -2
    assert oracle.tree_new(ints_to_arr([1, 2, 3, 4, 5, 6]))[0] == -3
    rc, h = oracle.tree_new(ints_to_arr([7]))
    assert rc == 0 and oracle.tree_root(h) == 7         # :27-33
    oracle.tree_free(h)
    leaves = [oracle.hash([i, i + 1, i + 2]) for i in range(8)]
    rc, h = oracle.tree_new(ints_to_arr(leaves))
    assert rc == 0
    l1 = [oracle.hash([leaves[2 * i], leaves[2 * i + 1]]) for i in range(4)]
    l2 = [oracle.hash([l1[0], l1[1]]), oracle.hash([l1[2], l1[3]])]
    assert oracle.tree_root(h) == oracle.hash(l2)
    for idx in range(8):
        proof, helper = oracle.tree_proof(h, idx)
        assert [int(x[0]) for x in helper] == [1 - ((idx >> l) & 1) for l in range(3)]   # :79
        assert oracle.path_root(leaves[idx], idx, proof) == oracle.tree_root(h)          # verify_proof
        rc2, r2 = oracle.compute_merkle_root(leaves[idx], proof, helper)                   # circuit form
        assert rc2 == 0 and r2 == oracle.tree_root(h)
    oracle.tree_free(h)


def test_is_less_than_limb_formula(oracle):
    # test_limbs_logic (:597-630) at a size that runs in seconds, plus boundary pairs
    rng = random.Random(5)
    pairs = [(rng.getrandbits(254), rng.getrandbits(254)) for _ in range(20000)]
    pairs += [(0, 0), (0, 1), (1, 0), (1 << 128, (1 << 128) - 1), ((1 << 128) - 1, 1 << 128), (5 << 128, 5 << 128),
              ((5 << 128) + 1, (5 << 128) + 2), (P - 1, P - 2), (P - 2, P - 1)]
    for a, b in pairs:
        assert oracle.is_less_than_limbs(a, b) == (1 if a < b else 0)


def _run_rounds(oracle, vals, depth):
    """test_insert_leaf_multiple_round (:679-803): update_idx_leaf + full rebuild each round,
    every insert_leaf relation checked, against the sparse builder."""
    n = 1 << depth
    pre = np.zeros((n, 3, 32), np.uint8)
    leaves = oracle.hash3_batch(pre)
    rc, tree = oracle.tree_new(leaves)
    assert rc == 0
    sp = oracle.sparse_new(depth, n)
    out = []
    for rnd, v in enumerate(vals):
        old_root = oracle.tree_root(tree)
        old_pre = pre.copy()
        low = oracle.update_idx_leaf(pre, v, rnd + 1)
        low_proof, low_helper = oracle.tree_proof(tree, low)
        leaves = oracle.hash3_batch(pre)
        oracle.tree_free(tree)
        rc, tree = oracle.tree_new(leaves)
        new_proof, new_helper = oracle.tree_proof(tree, rnd + 1)
        new_root = oracle.tree_root(tree)
        new_leaf = [int.from_bytes(pre[rnd + 1, j].tobytes(), "little") for j in range(3)]
        low_leaf = [int.from_bytes(old_pre[low, j].tobytes(), "little") for j in range(3)]
        largest = 1 if new_leaf[1] == 0 else 0
        f, trace = oracle.insert_leaf(old_root, low_leaf, low_proof, low_helper, new_root, new_leaf, rnd + 1,
                                      new_proof, new_helper, largest)
        assert f == 0, (rnd, hex(f))
        s = oracle.sparse_insert(sp, depth, v)
        assert s["rc"] == 0 and s["low"] == low and s["largest"] == largest
        assert s["new_root"] == new_root and s["interim_root"] == trace[3]
        assert (s["low_proof"] == low_proof).all() and (s["new_proof"] == new_proof).all()
        out.append((low, largest, trace[3], new_root))
    oracle.tree_free(tree)
    oracle.sparse_free(sp)
    return out


def test_multiple_round_matches_golden(oracle):
    got = _run_rounds(oracle, [30, 10, 20, 5, 50, 35], 3)
    for g, e in zip(got, GOLD["multi_round_depth3"]):
        assert g == (e["low_idx"], e["largest"], int(e["interim_root"]), int(e["new_root"]))


def test_test_insert_leaf_shape(oracle):
    # test_insert_leaf (:360-596): a random 254-bit value mod r as largest, then 42 as non-largest
    rng = random.Random(11)
    a = rng.getrandbits(254) % P
    got = _run_rounds(oracle, [a, 42], 3)
    assert got[0][:2] == (0, 1) and got[1][:2] == (0, 0)


def test_sparse_equals_dense_random(oracle):
    vals = oracle_lib.synth_values(15, 77)
    _run_rounds(oracle, vals, 4)


def test_insert_run_depth32_golden(oracle):
    run = GOLD["insert_run_depth32"]["rounds"]
    h = oracle.sparse_new(32, 64)
    for e in run:
        r = oracle.sparse_insert(h, 32, int(e["val"]))
        assert (r["low"], r["largest"], r["interim_root"], r["new_root"]) == (
            e["low_idx"], e["largest"], int(e["interim_root"]), int(e["new_root"]))
    oracle.sparse_free(h)


def test_relation_checker_detects_failures(oracle):
    # negative cases the reference does not test: each broken input flips its own bit
    depth = 3
    sp = oracle.sparse_new(depth, 8)
    for v in (30, 10):
        assert oracle.sparse_insert(sp, depth, v)["rc"] == 0
    old_root = oracle.sparse_root(sp)
    r = oracle.sparse_insert(sp, depth, 20)
    low_leaf = [int.from_bytes(r["low_leaf"][j].tobytes(), "little") for j in range(3)]
    new_leaf = [20, low_leaf[1], low_leaf[2]]
    hl = lambda idx: ints_to_arr([1 - ((idx >> l) & 1) for l in range(depth)])
    args = dict(old_root=old_root, low_leaf3=low_leaf, low_proof=r["low_proof"], low_helper=hl(r["low"]),
                new_root=r["new_root"], new_leaf3=new_leaf, new_index=3, new_proof=r["new_proof"], new_helper=hl(3),
                largest=r["largest"])
    assert oracle.insert_leaf(**args)[0] == 0
    bad = dict(args); bad["new_root"] = args["new_root"] ^ 1
    assert oracle.insert_leaf(**bad)[0] == 0x40
    bad = dict(args); bad["old_root"] = args["old_root"] ^ 1
    assert oracle.insert_leaf(**bad)[0] == 0x02
    bad = dict(args); bad["largest"] = 1 - args["largest"]
    assert oracle.insert_leaf(**bad)[0] & 0x01
    bad = dict(args); bad["new_leaf3"] = [20, low_leaf[1] + 1, low_leaf[2]]
    assert oracle.insert_leaf(**bad)[0] & 0x10
    assert oracle.sparse_insert(sp, depth, 20)["rc"] == -10    # duplicate
    assert oracle.sparse_insert(sp, depth, 0)["rc"] == -10     # zero
    oracle.sparse_free(sp)


def test_insert_trace_depth32_golden(oracle):
    """BASELINE config 5 at the size a circuit assigns: the 158 251-row witness trace of insert_leaf at depth 32
    (src/indexed_merkle_tree.rs:92, :194, :271-275, :299-303 are its hash_fix_len_array call sites) for insertions 38 and
    39 of the seeded run, recomputed from the oracle: digests as committed, every vertical gate of a sample holds
    through the chain ends (roots)."""
    import hashlib
    gold = {g["insertion"]: g for g in GOLD["insert_trace_depth32"]}
    h = oracle.sparse_new(32, 64)
    prev_root = None
    for i, v in enumerate(oracle_lib.synth_values(40, 0x494D5402)):
        r = oracle.sparse_insert(h, 32, v)
        assert r["rc"] == 0
        if i in gold:
            low3 = oracle_lib.arr_ints(r["low_leaf"])
            new3 = oracle_lib.arr_ints(oracle.sparse_preimage(h, i + 1))
            rows, roots = oracle_lib.insert_leaf_trace(oracle, low3, r["low"], r["low_proof"], new3, i + 1, r["new_proof"], 32)
            g = gold[i]
            assert rows.shape == (g["n_rows"], 32) and g["n_rows"] == 3 * 1209 + 4 * 32 * 1208
            assert hashlib.sha256(rows.tobytes()).hexdigest() == g["sha256_rows"]
            assert hashlib.sha256(rows[:g["non_inclusion_rows"]].tobytes()).hexdigest() == g["sha256_non_inclusion_rows"]
            assert roots == [prev_root, r["interim_root"], r["interim_root"], r["new_root"]]
        prev_root = r["new_root"]
    oracle.sparse_free(h)


def test_rebuild_from_preimages_equals_the_sequential_state(oracle):
    """orc_sparse_load = the reference's rebuild after every insertion (hash_nullifier_pre_images
    src/indexed_merkle_tree.rs:662-671 + IndexedMerkleTree::new src/utils.rs:38-51) on a sparse depth-32 tree: loaded with
    the list after k sequential insertions (written down independently, by sorting) it has the sequential run's root, and
    goes on exactly like it.  This is what lets tests/golden/make_config4_digest.py cut the 2^22-insertion oracle run into
    segments; a list that is not one sorted chain from the sentinel is refused."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_config4_digest import preimages_after
    n, depth = 400, 32
    vals = ints_to_arr(oracle_lib.synth_values(n, 0x494D5404))
    ints = oracle_lib.arr_ints(vals)
    h = oracle.sparse_new(depth, 1024)
    rows = [oracle.sparse_insert(h, depth, v) for v in ints]
    assert all(r["rc"] == 0 for r in rows)
    for k in (0, 1, 2, 3, 127, 128, 129, 333):
        g = oracle.sparse_new(depth, 1024)
        pre = preimages_after(vals, k)
        assert oracle.sparse_load(g, pre) == 0
        if k:
            assert oracle.sparse_root(g) == rows[k - 1]["new_root"], k
        for i in range(k, min(k + 40, n)):
            o = oracle.sparse_insert(g, depth, ints[i])
            assert (o["rc"], o["low"], o["largest"], o["interim_root"], o["new_root"]) == \
                (0, rows[i]["low"], rows[i]["largest"], rows[i]["interim_root"], rows[i]["new_root"]), (k, i)
            assert (o["low_proof"] == rows[i]["low_proof"]).all() and (o["new_proof"] == rows[i]["new_proof"]).all()
        assert oracle.sparse_load(g, pre) != 0              # only into a fresh tree
        oracle.sparse_free(g)
    for breakage in ("pointer", "value", "sentinel", "duplicate"):
        pre = preimages_after(vals, 10).copy()
        if breakage == "pointer":
            pre[3, 2, 0] ^= 1
        elif breakage == "value":
            pre[4, 1, 5] ^= 1
        elif breakage == "sentinel":
            pre[0, 0, 0] = 1
        else:
            pre[7, 0] = pre[6, 0]
        g = oracle.sparse_new(depth, 1024)
        assert oracle.sparse_load(g, pre) == -10, breakage
        oracle.sparse_free(g)
    oracle.sparse_free(h)
