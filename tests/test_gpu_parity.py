"""GPU (MI355X): the HIP path, called through the C ABI, against the CPU oracle on the same
inputs -- bit-exact, since everything is integer arithmetic mod p.  Mirrors the reference's
own tests (src/indexed_merkle_tree.rs:349-810) and adds the cases it leaves out."""
import json
import os
import sys
import random

import numpy as np
import pytest

import oracle_lib
from oracle_lib import P, KAT_ZERO, ints_to_arr

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))


def ints(a):
    a = np.asarray(a, dtype=np.uint8)
    return [int.from_bytes(x.tobytes(), "little") for x in a.reshape(-1, 32)]


# ---------------------------------------------------------------- a1 / a10
@pytest.mark.parametrize("form", ["quad", "thread"])
def test_kat_and_golden_vectors(imt, ctx, ctx_thread_per_hash, form):
    ctx = ctx if form == "quad" else ctx_thread_per_hash
    assert ints(ctx.hash3(ints_to_arr([0, 0, 0]).reshape(1, 3, 32))) == [KAT_ZERO]      # reference :248
    h2 = [e for e in GOLD["entries"] if e["kind"] == "hash2"]
    h3 = [e for e in GOLD["entries"] if e["kind"] == "hash3"]
    pm = [e for e in GOLD["entries"] if e["kind"] == "permute"]
    out2 = ctx.hash2(ints_to_arr([int(x) for e in h2 for x in e["in"]]).reshape(-1, 2, 32))
    assert ints(out2) == [int(e["out"]) for e in h2]
    out3 = ctx.hash3(ints_to_arr([int(x) for e in h3 for x in e["in"]]).reshape(-1, 3, 32))
    assert ints(out3) == [int(e["out"]) for e in h3]
    outp = ctx.permute(ints_to_arr([int(x) for e in pm for x in e["in"]]).reshape(-1, 3, 32))
    assert ints(outp) == [int(x) for e in pm for x in e["out"]]
    pub = [e for e in GOLD["entries"] if e["kind"] == "permute_lane0"]     # public circomlib known answers
    assert len(pub) == 2
    outq = ctx.permute(ints_to_arr([int(x) for e in pub for x in e["in"]]).reshape(-1, 3, 32))
    assert ints(outq[:, 0]) == [int(e["out"]) for e in pub]
    z = ctx.zero_hashes(32)
    for e in GOLD["entries"]:
        if e["kind"] == "empty_root":
            assert ints(z[int(e["in"][0])]) == [int(e["out"])]


@pytest.mark.parametrize("form", ["quad", "thread"])
def test_hash_batches_random_and_ragged(imt, ctx, ctx_thread_per_hash, oracle, form):
    """few hashes run a quad of lanes per hash (k_hash_batch_coop, up to 4096 hashes by default), many one thread"""
    ctx = ctx if form == "quad" else ctx_thread_per_hash
    rng = random.Random(10)
    for n in (1, 63, 64, 65, 257, 1000, 4096, 4097):       # ragged against the 256-thread blocks, across the switch
        a = ints_to_arr([rng.randrange(P) for _ in range(2 * n)]).reshape(n, 2, 32)
        assert (ctx.hash2(a) == oracle.hash2_batch(a)).all()
        b = ints_to_arr([rng.randrange(P) for _ in range(3 * n)]).reshape(n, 3, 32)
        assert (ctx.hash3(b) == oracle.hash3_batch(b)).all()
    assert ctx.hash2(np.zeros((0, 2, 32), np.uint8)).shape == (0, 32)     # empty batch


def test_formats_and_noncanonical(imt, ctx, oracle):
    rng = random.Random(12)
    R = 1 << 256
    vals = [rng.randrange(P) for _ in range(64)]
    want = oracle.hash2_batch(ints_to_arr(vals).reshape(-1, 2, 32))
    got = ctx.hash2(ints_to_arr([v * R % P for v in vals]).reshape(-1, 2, 32), fmt=imt._ffi.FMT_MONT256)
    assert ints(got) == [h * R % P for h in ints(want)]
    bad = ints_to_arr([1, P]).reshape(1, 2, 32)
    with pytest.raises(imt.ImtError) as ei:
        ctx.hash2(bad)
    assert ei.value.code == imt._ffi.ERR["NONCANONICAL"]
    assert ints(ctx.hash2(ints_to_arr([1, 2]).reshape(1, 2, 32))) == [oracle.hash([1, 2])]   # context still usable


# ---------------------------------------------------------------- a2 / a3 / a4 / a5
def test_dense_tree_new_errors(imt, ctx):
    with pytest.raises(ValueError, match="Cannot create Merkle Tree with no leaves"):
        imt.IndexedMerkleTree.new(ctx, [])
    with pytest.raises(ValueError, match="Leaves must be even"):
        imt.IndexedMerkleTree.new(ctx, [1, 2, 3])
    with pytest.raises(IndexError):
        imt.IndexedMerkleTree.new(ctx, [1, 2, 3, 4, 5, 6])
    t = imt.IndexedMerkleTree.new(ctx, [7])
    assert t.get_root() == 7 and t.num_levels() == 1 and t.get_proof(0) == ([], [])


@pytest.mark.parametrize("form", ["quad", "thread"])
@pytest.mark.parametrize("depth", [1, 3, 8])
def test_dense_tree_matches_oracle(imt, ctx, ctx_thread_per_hash, oracle, depth, form):
    ctx = ctx if form == "quad" else ctx_thread_per_hash
    rng = random.Random(depth)
    n = 1 << depth
    leaves = ints_to_arr([rng.randrange(P) for _ in range(n)])
    t = imt.IndexedMerkleTree.new(ctx, leaves)
    rc, ot = oracle.tree_new(leaves)
    assert rc == 0 and t.num_levels() == depth + 1
    for l in range(depth + 1):
        assert (t.get_level(l) == oracle.tree_level(ot, l)).all()
    assert t.get_root() == oracle.tree_root(ot)
    for idx in sorted({0, 1, n - 1, rng.randrange(n)}):
        proof, helper = t.get_proof(idx)
        op, oh = oracle.tree_proof(ot, idx)
        assert proof == ints(op) and helper == ints(oh)
        leaf = ints(leaves[idx])[0]
        assert t.verify_proof(leaf, idx, t.get_root(), proof)
        assert not t.verify_proof(leaf ^ 1, idx, t.get_root(), proof)
    with pytest.raises(IndexError):
        t.get_proof(n)
    oracle.tree_free(ot)


def test_config1_depth8_16_insertions(imt, ctx, oracle):
    """BASELINE config 1: depth 8, 16 sequential insertions through the reference's own dense
    builder path (update_idx_leaf -> rehash -> IndexedMerkleTree::new), root after each."""
    vals = oracle_lib.synth_values(16, 0x494D5401)
    pre = np.zeros((256, 3, 32), np.uint8)
    it = imt.IndexedTree(ctx, 8, 256)
    for rnd, v in enumerate(vals):
        oracle.update_idx_leaf(pre, v, rnd + 1)
        dense = imt.IndexedMerkleTree.new(ctx, ctx.hash3(pre))
        rc, ot = oracle.tree_new(oracle.hash3_batch(pre))
        assert dense.get_root() == oracle.tree_root(ot)
        it.insert_batch([v])
        assert it.root() == dense.get_root()
        oracle.tree_free(ot)


def test_path_root_layouts_and_helpers(imt, ctx, oracle):
    rng = random.Random(21)
    n, d = 300, 32
    leaf = [rng.randrange(P) for _ in range(n)]
    idx = [rng.randrange(1 << d) for _ in range(n)]
    sib = np.stack([ints_to_arr([rng.randrange(P) for _ in range(d)]) for _ in range(n)])   # [n][d][32]
    want = [oracle.path_root(leaf[i], idx[i], sib[i]) for i in range(n)]
    assert ints(ctx.path_root(ints_to_arr(leaf), idx, sib, d, item_major=True)) == want
    lm = np.ascontiguousarray(sib.transpose(1, 0, 2))
    assert ints(ctx.path_root(ints_to_arr(leaf), idx, lm, d)) == want
    hm = [(~i) & ((1 << d) - 1) for i in idx]
    assert ints(ctx.compute_merkle_root(ints_to_arr(leaf), hm, lm, d)) == want
    ok = ctx.verify_proof_batch(ints_to_arr(leaf), idx, ints_to_arr(want), lm, d)
    assert ok.all()
    wrong = list(want); wrong[5] ^= 1
    ok = ctx.verify_proof_batch(ints_to_arr(leaf), idx, ints_to_arr(wrong), lm, d)
    assert not ok[5] and ok.sum() == n - 1
    assert ints(ctx.path_root(ints_to_arr(leaf[:3]), idx[:3], np.zeros((0, 32), np.uint8), 0)) == leaf[:3]   # depth 0


# ---------------------------------------------------------------- a15 + a13 + a14
def _oracle_run(oracle, depth, cap, vals):
    h = oracle.sparse_new(depth, cap)
    rows = [oracle.sparse_insert(h, depth, v) for v in vals]
    root = oracle.sparse_root(h)
    return h, rows, root


@pytest.mark.parametrize("depth,n,batches", [(3, 6, 1), (8, 40, 3), (32, 200, 1), (32, 300, 4)])
def test_insert_batch_matches_sequential_oracle(imt, ctx, oracle, depth, n, batches):
    if depth == 3:
        vals = [30, 10, 20, 5, 50, 35]          # test_insert_leaf_multiple_round :683-690
    else:
        vals = oracle_lib.synth_values(n, 0x494D5402 + depth + batches)
    cap = min(512, 1 << depth)
    oh, rows, oroot = _oracle_run(oracle, depth, cap, vals)
    t = imt.IndexedTree(ctx, depth, cap)
    assert t.root() == ints(oracle.zero_hashes(depth)[depth])[0]
    got = []
    step = (len(vals) + batches - 1) // batches
    for s in range(0, len(vals), step):
        r = t.insert_batch(vals[s:s + step], item_major=True)
        for i in range(len(vals[s:s + step])):
            got.append({k: r[k][i] for k in r})
    assert t.root() == oroot and t.size == len(vals) + 1
    prev_root = ints(oracle.zero_hashes(depth)[depth])[0]
    for i, (g, o) in enumerate(zip(got, rows)):
        assert int(g["low_index"]) == o["low"] and int(g["is_largest"]) == o["largest"], i
        assert (g["low_leaf"] == o["low_leaf"]).all(), i
        assert ints(g["old_root"]) == [prev_root], i
        assert ints(g["interim_root"]) == [o["interim_root"]], i
        assert ints(g["new_root"]) == [o["new_root"]], i
        assert (g["low_sib"] == o["low_proof"]).all(), i
        assert (g["new_sib"] == o["new_proof"]).all(), i
        assert int(g["new_index"]) == i + 1
        prev_root = o["new_root"]
    # stored tree == oracle tree: proofs and preimages of every filled leaf and the next empty one
    idx = list(range(len(vals) + 2))
    sib = t.get_proof_batch(idx, item_major=True)
    leaves = t.get_leaves(idx)
    for i in idx:
        assert (sib[i] == oracle.sparse_proof(oh, depth, i)).all()
        assert (leaves[i] == oracle.sparse_preimage(oh, i)).all()
    oracle.sparse_free(oh)
    if depth == 3:
        for g, e in zip(got, GOLD["multi_round_depth3"]):
            assert ints(g["new_root"]) == [int(e["new_root"])] and int(g["low_index"]) == e["low_idx"]


def test_insert_batch_golden_depth32(imt, ctx):
    run = GOLD["insert_run_depth32"]["rounds"]
    t = imt.IndexedTree(ctx, 32, 64)
    r = t.insert_batch([int(e["val"]) for e in run], proofs=False)
    assert ints(r["new_root"]) == [int(e["new_root"]) for e in run]
    assert ints(r["interim_root"]) == [int(e["interim_root"]) for e in run]
    assert [int(x) for x in r["low_index"]] == [e["low_idx"] for e in run]


def test_insert_batch_adversarial_orders(imt, ctx, oracle):
    # ascending (every insertion is the new maximum), descending (leaf 0 is always the low leaf:
    # one run of N events), and a sawtooth
    for vals in (list(range(1, 65)), list(range(64, 0, -1)), [((i * 37) % 64) + 1 for i in range(64)]):
        oh, rows, oroot = _oracle_run(oracle, 32, 128, vals)
        t = imt.IndexedTree(ctx, 32, 128)
        r = t.insert_batch(vals, proofs=False)
        assert ints(r["new_root"]) == [o["new_root"] for o in rows]
        assert [int(x) for x in r["low_index"]] == [o["low"] for o in rows]
        assert t.root() == oroot
        oracle.sparse_free(oh)


def test_random_small_batches_differential(imt, ctx, oracle):
    """Many small batches with values that share their top limbs (the sorted index compares the top
    limb first and falls back to the full value), growing through several L0 transitions."""
    rng = random.Random(99)
    for trial in range(8):
        depth = rng.choice([6, 9, 32])
        cap = 64
        shape = trial % 3
        if shape == 0:
            pool = list(range(1, 400))                                  # top limbs all zero
        elif shape == 1:
            pool = [(7 << 192) + x for x in range(1, 400)]              # equal non-zero top limb
        else:
            pool = [rng.randrange(1, P) for _ in range(300)] + list(range(1, 100))
        rng.shuffle(pool)
        t = imt.IndexedTree(ctx, depth, cap)
        oh = oracle.sparse_new(depth, cap)
        used = 0
        while t.size < cap:
            n = min(rng.randrange(1, 10), cap - t.size)
            vals = pool[used:used + n]
            used += n
            r = t.insert_batch(vals, item_major=True)
            for i, v in enumerate(vals):
                o = oracle.sparse_insert(oh, depth, v)
                assert o["rc"] == 0
                assert int(r["low_index"][i]) == o["low"] and int(r["is_largest"][i]) == o["largest"]
                assert ints(r["new_root"][i]) == [o["new_root"]] and ints(r["interim_root"][i]) == [o["interim_root"]]
                assert (r["low_sib"][i] == o["low_proof"]).all() and (r["new_sib"][i] == o["new_proof"]).all()
                assert (r["low_leaf"][i] == o["low_leaf"]).all()
            assert t.root() == oracle.sparse_root(oh)
        with pytest.raises(imt.ImtError):
            t.insert_batch([pool[used]])            # full
        oracle.sparse_free(oh)
        t.close()


def test_quad_per_hash_kernel_is_bit_identical(imt, ctx, oracle):
    """Small launches use the latency form of the hash kernel (four lanes per hash, imt_coop_device.hpp); the same
    batches with it switched off (IMT_OPT_COOP_MAX_EVENTS = 0) and with it forced on for every size must give the same
    bytes, and both equal the oracle.  Covers leaf hashes (3 inputs), table-driven levels and the levels above l0."""
    depth = 32
    vals = oracle_lib.synth_values(700, 0x494D5444)
    cuts = [0, 1, 2, 5, 64, 65, 300, 700]
    res = {}
    for mode, coop_max in (("off", 0), ("default", 16384), ("always", 1 << 30)):
        c2 = imt.Context(0)
        c2.set_option(imt._ffi.OPT_COOP_MAX_EVENTS, coop_max)
        t = imt.IndexedTree(c2, depth, 1024)
        outs = [t.insert_batch(vals[a:b], host_prep=(i % 3 == 2)) for i, (a, b) in enumerate(zip(cuts, cuts[1:]))]
        res[mode] = ({k: np.concatenate([o[k] for o in outs], axis=1 if k.endswith("_sib") else 0) for k in outs[0]}, t.root())
        t.close(); c2.close()
    for mode in ("default", "always"):
        for k in res["off"][0]:
            assert (res[mode][0][k] == res["off"][0][k]).all(), (mode, k)
        assert res[mode][1] == res["off"][1]
    oh, rows, root = _oracle_run(oracle, depth, 1024, vals)
    r = res["always"][0]
    assert ints(r["new_root"]) == [o["new_root"] for o in rows] and ints(r["interim_root"]) == [o["interim_root"] for o in rows]
    assert all((r["low_sib"][:, i] == rows[i]["low_proof"]).all() and (r["new_sib"][:, i] == rows[i]["new_proof"]).all()
               for i in range(0, 700, 37))
    assert res["always"][1] == root
    oracle.sparse_free(oh)
    # the path kernel has the same two forms (few paths: a quad per path)
    rng = random.Random(3)
    n, d = 7, 32
    leaf = [rng.randrange(P) for _ in range(n)]
    sib = [[rng.randrange(P) for _ in range(n)] for _ in range(d)]
    idx = [rng.randrange(1 << d) for _ in range(n)]
    want = [oracle.path_root(leaf[i], idx[i], imt.to_bytes([sib[l][i] for l in range(d)])) for i in range(n)]
    for coop_max in (0, 16384):
        c2 = imt.Context(0)
        c2.set_option(imt._ffi.OPT_COOP_MAX_EVENTS, coop_max)
        assert ints(c2.path_root(imt.to_bytes(leaf), idx, imt.to_bytes(sib), d)) == want
        hm = [(~i) & ((1 << d) - 1) for i in idx]
        assert ints(c2.compute_merkle_root(imt.to_bytes(leaf), hm, imt.to_bytes(sib), d)) == want
        ok = c2.verify_proof_batch(imt.to_bytes(leaf), idx, imt.to_bytes(want[:1] * n), imt.to_bytes(sib), d)
        assert ok.tolist() == [True] + [False] * (n - 1)
        c2.close()
    with pytest.raises(imt.ImtError):
        ctx.set_option(99, 1)


def test_gpu_prepare_equals_host_prepare(imt, ctx, oracle):
    """The default device-side low-leaf search and event building against IMT_HOST_PREP, batch by
    batch, including adversarial orders, mixed modes on one tree, and the error cases."""
    rng = random.Random(123)
    keys = ("low_index", "is_largest", "low_leaf", "new_leaf", "old_root", "interim_root", "new_root", "low_sib",
            "new_sib")
    streams = [oracle_lib.synth_values(900, 0x494D5407), list(range(1, 901)), list(range(900, 0, -1)),
               [((i * 389) % 900) + 1 for i in range(900)], [(9 << 200) + x for x in range(1, 901)]]
    for vals in streams:
        a = imt.IndexedTree(ctx, 32, 1024)     # host prepare
        b = imt.IndexedTree(ctx, 32, 1024)     # GPU prepare
        m = imt.IndexedTree(ctx, 32, 1024)     # alternating
        pos, k = 0, 0
        while pos < len(vals):
            n = min(rng.choice([1, 2, 5, 33, 64, 200]), len(vals) - pos)
            chunk = vals[pos:pos + n]
            ra = a.insert_batch(chunk, host_prep=True)
            rb = b.insert_batch(chunk)
            rm = m.insert_batch(chunk, host_prep=(k % 2 == 0))
            for key in keys:
                assert (ra[key] == rb[key]).all(), (key, pos)
                assert (ra[key] == rm[key]).all(), (key, pos)
            pos += n
            k += 1
        assert a.root() == b.root() == m.root()
        probe = [v + 1 for v in vals[:50] if v + 1 not in set(vals)] or [P - 5]
        assert (a.find_low(probe) == b.find_low(probe)).all()           # mirror rebuilt from the device index
        assert (a.snapshot() == b.snapshot()).all() and (a.snapshot() == m.snapshot()).all()
    # the sequential oracle agrees (one stream is enough: the host path is checked against it elsewhere)
    oh, rows, oroot = _oracle_run(oracle, 32, 256, streams[0][:150])
    t = imt.IndexedTree(ctx, 32, 256)
    r = t.insert_batch(streams[0][:150])
    assert ints(r["new_root"]) == [o["new_root"] for o in rows] and t.root() == oroot
    assert [int(x) for x in r["low_index"]] == [o["low"] for o in rows]
    oracle.sparse_free(oh)
    # error cases leave the tree untouched
    root, size = t.root(), t.size
    for bad in ([0], [7, 7], [streams[0][3]], [5, streams[0][10], 6]):
        with pytest.raises(ValueError):
            t.insert_batch(bad)
    with pytest.raises(imt.ImtError) as ei:
        t.insert_batch([P + 1])
    assert ei.value.code == imt._ffi.ERR["NONCANONICAL"]
    assert t.root() == root and t.size == size
    r2 = t.insert_batch([5, 6])                             # and it still works afterwards
    assert t.size == size + 2


@pytest.mark.parametrize("host_prep", [False, True])
def test_insert_batch_rejects_bad_values(imt, ctx, host_prep):
    t = imt.IndexedTree(ctx, 8, 16)
    t.insert_batch([5, 9], host_prep=host_prep)
    root = t.root()
    for bad in ([0], [7, 7], [9], [3, 5]):
        with pytest.raises(ValueError):
            t.insert_batch(bad, host_prep=host_prep)
    assert t.root() == root and t.size == 3            # nothing changed
    with pytest.raises(imt.ImtError) as ei:
        t.insert_batch(list(range(100, 114)), host_prep=host_prep)          # 3 + 14 > 16
    assert ei.value.code == imt._ffi.ERR["FULL"]
    with pytest.raises(imt.ImtError) as ei:
        t.insert_batch([P], host_prep=host_prep)
    assert ei.value.code == imt._ffi.ERR["NONCANONICAL"]
    root = t.root()
    t.insert_batch([], host_prep=host_prep)                                 # empty batch: a no-op
    assert t.root() == root and t.size == 3
    t.insert_batch(list(range(100, 113)), host_prep=host_prep)              # exactly full
    assert t.size == 16
    with pytest.raises(imt.ImtError) as ei:
        t.insert_batch([200], host_prep=host_prep)                          # one past full
    assert ei.value.code == imt._ffi.ERR["FULL"]


def test_reference_tests_through_the_mirror_api(imt, ctx, oracle):
    """test_insert_leaf / test_insert_leaf_multiple_round (:360-596, :679-803): native tree and
    witnesses from the product, every insert_leaf constraint satisfied; then negative cases."""
    rng = random.Random(31)
    a = rng.getrandbits(254) % P
    for vals in ([a, 42], [30, 10, 20, 5, 50, 35]):
        n = 8
        pre = np.zeros((n, 3, 32), np.uint8)
        tree = imt.IndexedMerkleTree.new(ctx, ctx.hash3(pre))
        for rnd, v in enumerate(vals):
            old_root = tree.get_root()
            old_pre = pre.copy()
            low = oracle.update_idx_leaf(pre, v, rnd + 1)        # host list logic (hash-free)
            low_proof, low_helper = tree.get_proof(low)
            assert tree.verify_proof(ints(ctx.hash3(old_pre[low:low + 1]))[0], low, old_root, low_proof)
            tree = imt.IndexedMerkleTree.new(ctx, ctx.hash3(pre))
            new_proof, new_helper = tree.get_proof(rnd + 1)
            new_root = tree.get_root()
            new_leaf = ints(pre[rnd + 1])
            low_leaf = ints(old_pre[low])
            largest = new_leaf[1] == 0
            trace = imt.insert_leaf(ctx, old_root, low_leaf, low_proof, low_helper, new_root, new_leaf, rnd + 1,
                                    new_proof, new_helper, largest)
            f, otrace = oracle.insert_leaf(old_root, low_leaf, ints_to_arr(low_proof), ints_to_arr(low_helper),
                                           new_root, new_leaf, rnd + 1, ints_to_arr(new_proof),
                                           ints_to_arr(new_helper), largest)
            assert f == 0 and ints(trace[:, 0]) == otrace
            imt.verify_non_inclusion(ctx, old_root, low_leaf, low_proof, low_helper, v, largest)
            # negative cases: each breaks exactly the constraint it should
            with pytest.raises(imt.ConstraintError) as ei:
                imt.insert_leaf(ctx, old_root, low_leaf, low_proof, low_helper, new_root ^ 1, new_leaf, rnd + 1,
                                new_proof, new_helper, largest)
            assert ei.value.mask == imt._ffi.F_NEW_ROOT
            with pytest.raises(imt.ConstraintError) as ei:
                imt.verify_non_inclusion(ctx, old_root ^ 1, low_leaf, low_proof, low_helper, v, largest)
            assert ei.value.mask == imt._ffi.F_LOW_IN_ROOT
            with pytest.raises(imt.ConstraintError) as ei:
                imt.verify_non_inclusion(ctx, old_root, low_leaf, low_proof, low_helper, v, not largest)
            assert ei.value.mask & imt._ffi.F_RANGE_PRED
            with pytest.raises(imt.ConstraintError) as ei:     # a value below the low leaf
                imt.verify_non_inclusion(ctx, old_root, low_leaf, low_proof, low_helper, low_leaf[0], largest)
            assert ei.value.mask & imt._ffi.F_LOW_LT_NEW


@pytest.fixture(scope="module")
def ctx_thread_per_hash(imt):
    """a context with the quad-per-hash (latency) kernels switched off: small launches run one thread per hash / item"""
    c = imt.Context(0)
    c.set_option(imt._ffi.OPT_COOP_MAX_EVENTS, 0)
    yield c
    c.close()


@pytest.mark.parametrize("form", ["quad", "thread"])
def test_non_membership_batch_vs_oracle(imt, ctx, ctx_thread_per_hash, oracle, form):
    """both forms of the kernels: few items run a quad of lanes per item (k_non_membership_coop), many one thread"""
    ctx = ctx if form == "quad" else ctx_thread_per_hash
    depth, cap = 32, 256
    vals = oracle_lib.synth_values(100, 0x494D5403)
    t = imt.IndexedTree(ctx, depth, cap)
    t.insert_batch(vals[:60], proofs=False)
    root = t.root()
    cand = vals[60:]
    low, leaves, sib, largest = t.non_membership_witness(cand)
    fail, rout = ctx.non_membership(imt.to_bytes(root), leaves, low, sib, depth, ints_to_arr(cand), largest,
                                    want_root=True)
    assert not fail.any() and ints(rout) == [root] * len(cand)
    for i in range(len(cand)):
        helper = ints_to_arr([1 - ((int(low[i]) >> l) & 1) for l in range(depth)])
        f, r = oracle.verify_non_inclusion(root, ints(leaves[i]), sib[:, i], helper, cand[i], int(largest[i]))
        assert f == 0 and r == root
    # the host-mirror route gives the same witness
    hl, hleaves, hsib, hlg = t.non_membership_witness(cand, host=True)
    assert (hl == low).all() and (hleaves == leaves).all() and (hsib == sib).all() and (hlg == largest).all()
    # members are not non-members: the predicate fails for every inserted value
    with pytest.raises(ValueError):
        t.find_low(vals[:1])
    for bad in (vals[:1], [0]):
        with pytest.raises(ValueError):
            t.non_membership_witness(bad)
    with pytest.raises(imt.ImtError):
        t.non_membership_witness([P])
    fail = ctx.non_membership(imt.to_bytes(root), leaves, low, sib, depth, leaves[:, 0, :].copy(), largest)
    assert (fail & imt._ffi.F_LOW_LT_NEW).all()
    # per-item roots and a bad largest flag value
    lg = largest.copy(); lg[0] = 2
    fail = ctx.non_membership(np.repeat(imt.to_bytes(root)[None], len(cand), 0), leaves, low, sib, depth,
                              ints_to_arr(cand), lg)
    assert fail[0] & imt._ffi.F_BAD_BIT and not fail[1:].any()


@pytest.mark.parametrize("form", ["quad", "thread"])
def test_insert_witness_batch_vs_oracle(imt, ctx, ctx_thread_per_hash, oracle, form):
    """both forms: few items run a quad of lanes per (item, chain) (k_insert_chains_coop), many one thread per chain"""
    ctx = ctx if form == "quad" else ctx_thread_per_hash
    depth, cap, n = 32, 256, 120
    vals = oracle_lib.synth_values(n, 0x494D5404)
    t = imt.IndexedTree(ctx, depth, cap)
    r = t.insert_batch(vals)
    fail, trace = ctx.insert_witness(r["old_root"], r["low_leaf"], r["low_index"], r["low_sib"], r["new_root"],
                                     r["new_leaf"], r["new_index"], r["new_sib"], r["is_largest"], depth,
                                     want_trace=True)
    assert not fail.any()
    assert (trace[3] == r["interim_root"]).all() and (trace[6] == r["new_root"]).all()
    assert (trace[4] == r["interim_root"]).all()
    for i in (0, 1, n // 2, n - 1):
        hl = lambda idx: ints_to_arr([1 - ((int(idx) >> l) & 1) for l in range(depth)])
        f, otr = oracle.insert_leaf(ints(r["old_root"][i])[0], ints(r["low_leaf"][i]), r["low_sib"][:, i],
                                    hl(r["low_index"][i]), ints(r["new_root"][i])[0], ints(r["new_leaf"][i]),
                                    int(r["new_index"][i]), r["new_sib"][:, i], hl(r["new_index"][i]),
                                    int(r["is_largest"][i]))
        assert f == 0 and ints(trace[:, i]) == otr
    # corrupt one witness field per item class and see the right bit
    nr = r["new_root"].copy(); nr[3, 0] ^= 1
    nl = r["new_leaf"].copy(); nl[4, 2, 0] ^= 1
    fail = ctx.insert_witness(r["old_root"], r["low_leaf"], r["low_index"], r["low_sib"], nr, nl, r["new_index"],
                              r["new_sib"], r["is_largest"], depth)
    assert fail[3] == imt._ffi.F_NEW_ROOT and fail[4] == (imt._ffi.F_NEXT_IDX | imt._ffi.F_NEW_ROOT)
    assert not np.delete(fail, [3, 4]).any()


def test_device_pointer_mode_matches_host_mode(imt, ctx):
    """IMT_DEVICE_PTRS (what bench.py uses): torch-allocated buffers, asynchronous on torch's stream,
    pipelined over 3 batches, halo2curves Montgomery format in and out."""
    import ctypes
    import torch
    depth, n, R = 32, 512, 1 << 256
    vals = oracle_lib.synth_values(3 * n, 0x494D5405)
    t_host = imt.IndexedTree(ctx, depth, 4096)
    want = [t_host.insert_batch(vals[i * n:(i + 1) * n]) for i in range(3)]
    c2 = imt.Context(0)
    c2.set_stream(torch.cuda.current_stream().cuda_stream)
    t_dev = imt.IndexedTree(c2, depth, 4096)
    dev = torch.device("cuda", 0)
    vm = torch.from_numpy(ints_to_arr([v * R % P for v in vals])).to(dev)
    outs = []
    flags = imt._ffi.DEVICE_PTRS | imt._ffi.FMT_MONT256 | imt._ffi.PIPELINE   # top kernel on its own stream
    for i in range(3):
        b = dict(low_index=torch.empty(n, dtype=torch.int64, device=dev),
                 is_largest=torch.empty(n, dtype=torch.uint8, device=dev),
                 low_leaf=torch.empty((n, 3, 32), dtype=torch.uint8, device=dev),
                 new_leaf=torch.empty((n, 3, 32), dtype=torch.uint8, device=dev),
                 old_root=torch.empty((n, 32), dtype=torch.uint8, device=dev),
                 interim_root=torch.empty((n, 32), dtype=torch.uint8, device=dev),
                 new_root=torch.empty((n, 32), dtype=torch.uint8, device=dev),
                 low_sib=torch.empty((depth, n, 32), dtype=torch.uint8, device=dev),
                 new_sib=torch.empty((depth, n, 32), dtype=torch.uint8, device=dev))
        o = imt._ffi.InsertOut(**{k: v.data_ptr() for k, v in b.items()})
        rc = imt.lib.imt_itree_insert_batch(t_dev.h, ctypes.c_void_p(vm.data_ptr() + i * n * 32), n, ctypes.byref(o),
                                            flags | (imt._ffi.HOST_PREP if i == 1 else 0))
        assert rc == 0, imt.lib.imt_last_error(c2.h)
        outs.append(b)
    c2.sync()
    torch.cuda.synchronize()
    to_mont = lambda a: [x * R % P for x in ints(a)]
    for i in range(3):
        for k in ("old_root", "interim_root", "new_root"):
            assert ints(outs[i][k].cpu().numpy()) == to_mont(want[i][k]), (i, k)
        for k in ("low_leaf", "new_leaf"):
            assert ints(outs[i][k].cpu().numpy()) == to_mont(want[i][k]), (i, k)
        assert (outs[i]["low_index"].cpu().numpy().astype(np.uint64) == want[i]["low_index"]).all()
        assert (outs[i]["is_largest"].cpu().numpy() == want[i]["is_largest"]).all()
        for k in ("low_sib", "new_sib"):
            got = outs[i][k].cpu().numpy()
            assert ints(got[:, ::37]) == to_mont(want[i][k][:, ::37]), (i, k)
    # lagged roots: after 3 batches, lag 0 / 1 are the roots after batch 3 / batch 2
    for lag, want_root in ((0, want[2]["new_root"][-1]), (1, want[1]["new_root"][-1])):
        buf = np.empty(32, np.uint8)
        assert imt.lib.imt_itree_root_lagged(t_dev.h, lag, buf.ctypes.data_as(ctypes.c_void_p), 0) == 0
        assert (buf == want_root).all()
    assert imt.lib.imt_itree_root_lagged(t_dev.h, 2, buf.ctypes.data_as(ctypes.c_void_p), 0) == imt._ffi.ERR["RANGE"]
    assert t_dev.root() == t_host.root()
    t_dev.close()
    c2.close()


def test_device_pointer_inputs_may_still_be_in_production_on_the_callers_stream(imt, ctx):
    """imt.h: with IMT_DEVICE_PTRS the work is enqueued on the context's stream.  The batch's preparation runs on an
    internal side stream, so it must first be ordered behind what the caller has enqueued: here `vals` is all zeros
    (value 0 = IMT_ERR_VALUE) until a copy lands that sits behind ~50 ms of busy-waiting on the caller's stream."""
    import ctypes
    import torch
    depth, n = 32, 2048
    dev = torch.device("cuda", 0)
    c2 = imt.Context(0)
    c2.set_stream(torch.cuda.current_stream().cuda_stream)
    want_t = imt.IndexedTree(c2, depth, 8192)
    real = imt.to_bytes(oracle_lib.synth_values(2 * n, 0x494D5443))
    want = [want_t.insert_batch(real[i * n:(i + 1) * n], proofs=False)["new_root"] for i in range(2)]
    t = imt.IndexedTree(c2, depth, 8192)
    src = torch.from_numpy(real).to(dev)
    for b, flags in enumerate((imt._ffi.DEVICE_PTRS, imt._ffi.DEVICE_PTRS | imt._ffi.PIPELINE)):
        vals = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
        new_root = torch.empty((n, 32), dtype=torch.uint8, device=dev)
        out = imt._ffi.InsertOut(new_root=new_root.data_ptr())
        torch.cuda.synchronize()
        torch.cuda._sleep(120_000_000)                        # ~50 ms on the caller's stream
        vals.copy_(src[b * n:(b + 1) * n], non_blocking=True)
        rc = imt.lib.imt_itree_insert_batch(t.h, ctypes.c_void_p(vals.data_ptr()), n, ctypes.byref(out), flags)
        assert rc == 0, imt.lib.imt_last_error(c2.h)          # read too early it would be "value 0 cannot be inserted"
        c2.sync()
        torch.cuda.synchronize()
        assert (new_root.cpu().numpy() == want[b]).all()
    t.close(); want_t.close(); c2.close()


def test_snapshot_load_roundtrip(imt, ctx, oracle):
    depth = 32
    vals = oracle_lib.synth_values(700, 0x494D5406)
    a = imt.IndexedTree(ctx, depth, 1024)
    a.insert_batch(vals[:500], proofs=False)
    snap = a.snapshot()
    b = imt.IndexedTree(ctx, depth, 2048)            # a different capacity on purpose
    b.load(snap)
    assert b.root() == a.root() and b.size == a.size
    idx = [0, 1, 250, 500, 501, 777]
    assert (b.get_proof_batch(idx) == a.get_proof_batch(idx)).all()
    ra = a.insert_batch(vals[500:], proofs=True)     # both continue identically after the resume
    rb = b.insert_batch(vals[500:], proofs=True)
    for k in ("low_index", "new_root", "interim_root", "low_sib", "new_sib", "low_leaf"):
        assert (ra[k] == rb[k]).all(), k
    assert a.root() == b.root()
    # corrupted snapshots are refused and leave the tree as it was
    root_before = b.root()
    bad = snap.copy(); bad[3, 1, 0] ^= 1             # next_val no longer the successor
    with pytest.raises(ValueError):
        b.load(bad)
    bad = snap.copy(); bad[0, 0, 0] = 1              # leaf 0 is not the sentinel
    with pytest.raises(ValueError):
        b.load(bad)
    assert b.root() == root_before
    # bulk build == oracle: small tree, every level
    c = imt.IndexedTree(ctx, 8, 64)
    c.insert_batch(list(range(10, 40)), proofs=False)
    d = imt.IndexedTree(ctx, 8, 64)
    d.load(c.snapshot())
    oh = oracle.sparse_new(8, 64)
    for v in range(10, 40):
        oracle.sparse_insert(oh, 8, v)
    assert d.root() == oracle.sparse_root(oh)
    for i in (0, 5, 30, 31, 63):
        assert (d.get_proof_batch([i])[:, 0] == oracle.sparse_proof(oh, 8, i)).all()
    oracle.sparse_free(oh)


def test_snapshot_and_load_on_the_device(imt, ctx, oracle):
    """The checkpoint without the host: imt_itree_get_leaves / imt_itree_load with device pointers and with index = NULL,
    in every boundary format; the list check (k_load_check) refuses each kind of broken snapshot with its own message and
    writes nothing; values that agree in their top 64 bits take the second, fully compared sort."""
    import ctypes
    import torch
    depth, n = 32, 600
    vals = oracle_lib.synth_values(n, 0x494D5421)
    top = (vals[10] >> 192) << 192
    for k, low in ((20, 9), (30, 4), (40, 7), (50, 1)):      # same top limb, inserted in non-monotone order
        vals[k] = top | low
    a = imt.IndexedTree(ctx, depth, 1024)
    a.insert_batch(vals, proofs=False)
    oh = oracle.sparse_new(depth, 1024)
    for v in vals:
        oracle.sparse_insert(oh, depth, v)
    # get_leaves from the device index == the oracle's leaves, empty slots included
    idx = np.array([0, 1, 21, 31, 41, 51, n, n + 1, 1023], dtype=np.uint64)
    got = a.get_leaves(idx)
    for k, i in enumerate(idx):
        want = oracle.sparse_preimage(oh, int(i)) if i <= n else np.zeros((3, 32), np.uint8)
        assert (got[k] == np.asarray(want).reshape(3, 32)).all(), int(i)
    with pytest.raises(imt.ImtError):
        a.get_leaves(np.array([1024], dtype=np.uint64))
    snap = a.snapshot()
    assert snap.shape == (n + 1, 3, 32) and (snap[idx[:6].astype(np.int64)] == got[:6]).all()
    # device to device, Montgomery bytes on the way
    for fmt in (0, 1, 2):
        d_snap = torch.empty((n + 1, 3, 32), dtype=torch.uint8, device="cuda")
        a.snapshot_into(d_snap.data_ptr(), fmt)
        if fmt == 0:
            assert (d_snap.cpu().numpy() == snap).all()
        b = imt.IndexedTree(ctx, depth, 2048)
        b.load_device(d_snap.data_ptr(), n + 1, fmt)
        assert b.root() == a.root() == oracle.sparse_root(oh) and b.size == n + 1
        assert (b.snapshot() == snap).all()
        more = oracle_lib.synth_values(64, 0x494D5422)
        ra, rb = imt.IndexedTree(ctx, depth, 1024), b
        ra.load(snap)
        x, y = ra.insert_batch(more, proofs=True), rb.insert_batch(more, proofs=True)
        for k in ("low_index", "new_root", "low_sib", "new_sib", "low_leaf"):
            assert (x[k] == y[k]).all(), k
        ra.close(); b.close()
    # every kind of broken snapshot, refused with the tree untouched
    b = imt.IndexedTree(ctx, depth, 1024)
    b.load(snap)
    root = b.root()
    # the low-leaf search of a loaded tree runs in the device index (k_find_low): host and device pointers
    stored = [0] + vals
    probe = [v + 1 for v in vals[:40] if v + 1 not in set(vals)] + [top | 5, top | 8, P - 1, 1]
    want = [max((i for i in range(n + 1) if stored[i] < x), key=lambda i: stored[i]) for x in probe]
    assert [int(x) for x in b.find_low(probe)] == want
    d_probe = torch.from_numpy(imt.to_bytes(probe)).cuda()
    d_low = torch.empty(len(probe), dtype=torch.int64, device="cuda")
    ctx._check(imt.lib.imt_itree_find_low_batch(b.h, ctypes.c_void_p(d_probe.data_ptr()), len(probe),
                                                ctypes.c_void_p(d_low.data_ptr()), imt._ffi.DEVICE_PTRS))
    assert d_low.cpu().tolist() == want
    for member in (vals[3], 0):
        with pytest.raises(ValueError):
            b.find_low([member])
    order = sorted(range(n + 1), key=lambda i: int.from_bytes(snap[i, 0].tobytes(), "little"))
    def refused(bad, text, exc=ValueError):
        with pytest.raises(exc, match=text):
            b.load(bad)
        assert b.root() == root and (b.snapshot() == snap).all()
    bad = snap.copy(); bad[3, 1, 0] ^= 1;              refused(bad, "leaf 3 does not point")
    bad = snap.copy(); bad[9, 2, 0] ^= 1;              refused(bad, "leaf 9 does not point")
    bad = snap.copy(); bad[0, 0, 0] = 1;               refused(bad, "sentinel")
    bad = snap.copy(); bad[order[-1], 2, 0] = 7;       refused(bad, "largest leaf")
    bad = snap.copy(); bad[order[5], 0] = bad[order[6], 0]; refused(bad, "duplicate")
    bad = snap.copy(); bad[21, 0, 0] ^= 2;             refused(bad, "does not point|duplicate")   # inside the tied group
    bad = snap.copy(); bad[17, 0] = 0xFF;              refused(bad, "not reduced", imt.ImtError)
    # one leaf: the sentinel alone
    e = imt.IndexedTree(ctx, depth, 16)
    e.load(np.zeros((1, 3, 32), np.uint8))
    f = imt.IndexedTree(ctx, depth, 16)
    assert e.root() == f.root() and e.size == 1
    with pytest.raises(imt.ImtError):
        e.load(np.zeros((17, 3, 32), np.uint8))          # over capacity
    oracle.sparse_free(oh)
    for t in (a, b, e, f):
        t.close()


def _load_sharded_module():
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "indexed-merkle-tree-halo2_amd",
                        "sharded.py")
    spec = importlib.util.spec_from_file_location("imt_sharded", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_sharded_single_list_stepping_in_one_process(imt, ctx, oracle):
    """imt_itree_batch_*: the sharded single-list batch driven by one process, first as one rank, then as
    two and four 'ranks' taking their slot ranges in turn on the same arrays (what the all-gathers would
    produce) -- every output equals imt_itree_insert_batch, which equals the sequential oracle."""
    import ctypes
    import torch
    sharded = _load_sharded_module()
    depth = 32
    vals = oracle_lib.synth_values(1200, 0x494D5408)
    ref = imt.IndexedTree(ctx, depth, 2048)
    want = [ref.insert_batch(vals[a:a + 400]) for a in (0, 400, 800)]
    dev = torch.device("cuda", 0)
    F = imt._ffi
    for parts in (1, 2, 4):
        t = imt.IndexedTree(ctx, depth, 2048)
        for bi, a in enumerate((0, 400, 800)):
            chunk = torch.from_numpy(ints_to_arr(vals[a:a + 400])).to(dev)
            n = 400
            if parts == 1:
                r = sharded.ReplicatedIndexedTree(imt, ctx, t).insert_batch(chunk)
                got = {k: v.cpu().numpy() for k, v in r.items() if k != "first_insertion"}
            else:
                ev, l0 = ctypes.c_uint32(), ctypes.c_uint32()
                P_ = lambda x: ctypes.c_void_p(x.data_ptr())
                assert imt.lib.imt_itree_batch_begin(t.h, P_(chunk), n, F.DEVICE_PTRS, ctypes.byref(ev), ctypes.byref(l0)) == 0
                E, L0 = ev.value, l0.value
                val = torch.empty((L0 + 1, E, 32), dtype=torch.uint8, device=dev)
                kc = E // parts
                for q in reversed(range(parts)):        # any order: slots of one level are independent
                    assert imt.lib.imt_itree_batch_leaves(t.h, P_(val[0]), q * kc, kc) == 0
                for l in range(L0):
                    for q in range(parts):
                        assert imt.lib.imt_itree_batch_level(t.h, l, P_(val[l]), P_(val[l + 1]), q * kc, kc) == 0
                roots = torch.empty((E, 32), dtype=torch.uint8, device=dev)
                tops = [torch.zeros((depth - L0 + 1, 32), dtype=torch.uint8, device=dev) for _ in range(parts)]
                for q in range(parts):
                    assert imt.lib.imt_itree_batch_top(t.h, P_(val[L0]), q * kc, kc, P_(roots), P_(tops[q])) == 0
                ptrs = (ctypes.c_void_p * (L0 + 1))(*[val[l].data_ptr() for l in range(L0 + 1)])
                got = None
                ic = n // parts
                for q in range(parts):
                    o = dict(low_index=torch.empty(ic, dtype=torch.int64, device=dev),
                             is_largest=torch.empty(ic, dtype=torch.uint8, device=dev),
                             low_leaf=torch.empty((ic, 3, 32), dtype=torch.uint8, device=dev),
                             new_leaf=torch.empty((ic, 3, 32), dtype=torch.uint8, device=dev),
                             old_root=torch.empty((ic, 32), dtype=torch.uint8, device=dev),
                             interim_root=torch.empty((ic, 32), dtype=torch.uint8, device=dev),
                             new_root=torch.empty((ic, 32), dtype=torch.uint8, device=dev),
                             low_sib=torch.empty((depth, ic, 32), dtype=torch.uint8, device=dev),
                             new_sib=torch.empty((depth, ic, 32), dtype=torch.uint8, device=dev))
                    st = F.InsertOut(**{k: v.data_ptr() for k, v in o.items()})
                    assert imt.lib.imt_itree_batch_extract(t.h, ptrs, P_(roots), q * ic, ic, ctypes.byref(st), F.DEVICE_PTRS) == 0
                    ctx.sync()
                    o = {k: v.cpu().numpy() for k, v in o.items()}
                    if got is None:
                        got = {k: [v] for k, v in o.items()}
                    else:
                        for k, v in o.items():
                            got[k].append(v)
                got = {k: np.concatenate(v, axis=1 if k.endswith("_sib") else 0) for k, v in got.items()}
                assert imt.lib.imt_itree_batch_end(t.h, ptrs, P_(tops[parts - 1])) == 0
                ctx.sync()
            for k in ("low_index", "is_largest", "low_leaf", "new_leaf", "old_root", "interim_root", "new_root",
                      "low_sib", "new_sib"):
                assert (got[k].astype(want[bi][k].dtype) == want[bi][k]).all(), (parts, bi, k)
            assert t.root() == imt.to_int(want[bi]["new_root"][-1])
        assert t.root() == ref.root()
        assert (t.snapshot() == ref.snapshot()).all()
        r = t.insert_batch([3, 1, 2])                 # the ordinary path continues on the same tree
        assert t.size == 1204
    # a batch that must be refused leaves nothing open
    t = imt.IndexedTree(ctx, depth, 64)
    bad = torch.from_numpy(ints_to_arr([5, 5])).to(dev)
    ev, l0 = ctypes.c_uint32(), ctypes.c_uint32()
    assert imt.lib.imt_itree_batch_begin(t.h, ctypes.c_void_p(bad.data_ptr()), 2, F.DEVICE_PTRS, ctypes.byref(ev),
                                         ctypes.byref(l0)) == F.ERR["VALUE"]
    assert imt.lib.imt_itree_batch_leaves(t.h, ctypes.c_void_p(bad.data_ptr()), 0, 1) == F.ERR["ARG"]
    t.insert_batch([7, 9])
    assert t.size == 3


def test_split128_limb_witnesses(imt, ctx):
    rng = random.Random(61)
    vals = [0, 1, (1 << 128) - 1, 1 << 128, (1 << 128) + 1, P - 1] + [rng.randrange(P) for _ in range(500)]
    q, r = ctx.split128(ints_to_arr(vals))
    assert ints(q) == [v >> 128 for v in vals] and ints(r) == [v & ((1 << 128) - 1) for v in vals]
    R = 1 << 256
    q, r = ctx.split128(ints_to_arr([v * R % P for v in vals]), fmt=imt._ffi.FMT_MONT256)
    assert ints(q) == [(v >> 128) * R % P for v in vals] and ints(r) == [(v & ((1 << 128) - 1)) * R % P for v in vals]


def test_c_example_runs(imt):
    """examples/insert_demo.c: the C ABI from plain C, end to end on the GPU."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "insert_demo")
    csrc = os.path.join(root, "indexed-merkle-tree-halo2_amd", "csrc")
    r = subprocess.run(["gcc", "-std=c11", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "insert_demo.c"),
                        "-L", csrc, "-limt_hip", "-Wl,-rpath," + csrc, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all satisfied" in r.stdout and "trace rows ok" in r.stdout and "4506 cells" in r.stdout
    assert "7 leaves reloaded, root equal; corrupted snapshot refused (leaf 2 does not point to its successor)" in r.stdout
    want = int(GOLD["multi_round_depth3"][-1]["new_root"])
    assert f"{want:064x}" in r.stdout          # the last root of test_insert_leaf_multiple_round
    # examples/slice_demo.c: two replicas of ONE list driven through imt_sliced_step from plain C (schedule and exchange
    # inside the library, the in-process transport); equal to the ordinary tree
    exe = os.path.join(root, "examples", "slice_demo")
    r = subprocess.run(["gcc", "-std=c11", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "slice_demo.c"),
                        "-L", csrc, "-limt_hip", "-Wl,-rpath," + csrc, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "replicas equal the one-GPU tree (49 leaves; lag 6, " in r.stdout, r.stdout + r.stderr
    # examples/sliced_procs_demo.c: a multi-PROCESS host in plain C (fork, pipes for the bootstrap bytes, no Python, no
    # torch): two and three ranks sharing this GPU over the IPC transport, one rank over RCCL (ncclCommInitRank and
    # ncclAllGather called by the library); every replica's root = the one-tree batch's
    exe = os.path.join(root, "examples", "sliced_procs_demo")
    r = subprocess.run(["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-I", os.path.join(root, "include"),
                        os.path.join(root, "examples", "sliced_procs_demo.c"), "-L", csrc, "-limt_hip", "-Wl,-rpath," + csrc, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, IMT_DEMO_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for args in (["2", "ipc", "6", "256"], ["3", "ipc", "5", "100"], ["1", "rccl", "4", "128"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and "replicas equal" in r.stdout and "root " in r.stdout, (args, r.stdout + r.stderr)
    # examples/subtree_procs_demo.c: north_star's layout (placed subtrees by leaf-index range, ONE all-gather of subtree roots
    # per step through the library's own communicators: imt_transport_all_gather) from a plain-C multi-process host: every
    # witness lifted to depth 32 passes insert_leaf's constraints (imt_insert_witness_batch), the roots of all ranks and
    # steps form one chain -- two and four ranks sharing this GPU over IPC, one rank over RCCL
    exe = os.path.join(root, "examples", "subtree_procs_demo")
    r = subprocess.run(["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-I", os.path.join(root, "include"),
                        os.path.join(root, "examples", "subtree_procs_demo.c"), "-L", csrc, "-limt_hip", "-Wl,-rpath," + csrc, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for args in (["2", "ipc", "5", "192"], ["4", "ipc", "4", "96"], ["1", "rccl", "3", "128"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and "the roots form one chain" in r.stdout and "witnesses lifted to depth 32" in r.stdout, (args, r.stdout + r.stderr)


def test_c_abi_survives_null_and_nonsense_arguments(imt):
    """tests/native/null_args_probe.py in a child process: every int-returning entry point with a NULL handle, with a
    valid handle and NULL everything else, and with the unknown format 3.  Nothing may crash; a NULL handle is always an
    error; every return value is a documented code; the handles work afterwards."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "native", "null_args_probe.py")], capture_output=True,
                       text=True, timeout=300, cwd=root, env=dict(os.environ, PYTHONPATH=root))
    calls = [l.split()[1:3] for l in r.stdout.splitlines() if l.startswith("CALL ")]
    rcs = {(l.split()[1], l.split()[2]): int(l.split()[3]) for l in r.stdout.splitlines() if l.startswith("RC ")}
    last = calls[-1] if calls else None
    assert r.returncode == 0 and "ALIVE" in r.stdout, f"died in {last}: " + r.stderr[-1500:]
    assert len(rcs) == len(calls) > 120
    for (name, label), rc in rcs.items():
        assert -12 <= rc <= 0, (name, label, rc)
        if label == "null-handle":
            assert rc < 0, (name, rc)
    # spot checks of the documented codes
    assert rcs[("imt_hash2_batch", "null-args")] == imt._ffi.ERR["ARG"]
    assert rcs[("imt_itree_insert_batch", "null-args")] == imt._ffi.ERR["ARG"]
    assert rcs[("imt_hash2_batch", "bad-format")] == imt._ffi.ERR["ARG"]
    assert rcs[("imt_tree_new", "null-args")] == imt._ffi.ERR["ARG"]
    # device pointers to field elements that are not 16-byte aligned are an argument error, not a GPU fault
    for name in ("imt_hash2_batch", "imt_hash3_batch", "imt_permute_batch", "imt_hash_trace_batch", "imt_path_trace_batch",
                 "imt_insert_trace_batch", "imt_tree_new", "imt_tree_build", "imt_tree_get_root", "imt_tree_get_proof_batch",
                 "imt_path_root_batch", "imt_compute_merkle_root_batch", "imt_verify_proof_batch", "imt_non_membership_batch",
                 "imt_split128_batch", "imt_insert_witness_batch", "imt_itree_root", "imt_itree_insert_batch",
                 "imt_itree_get_proof_batch", "imt_itree_non_membership_witness", "imt_itree_find_low_batch",
                 "imt_combine_subtree_roots", "imt_zero_hashes", "imt_itree_batch_begin",
                 "imt_itree_slice_prepare"):
        assert rcs[(name, "odd-offset")] == imt._ffi.ERR["ARG"], (name, rcs[(name, "odd-offset")])


def test_reference_tests_in_cpp(imt, oracle, tmp_path):
    """tests/native/reference_tests.cpp: the reference's own tests (test_hash_zero, test_insert_leaf,
    test_insert_leaf_multiple_round, test_limbs_logic, the two Err strings of IndexedMerkleTree::new) re-enacted
    on include/imt.hpp, the compiled-language host side above the C ABI, g++ only.  The program checks what the
    reference's tests check; the values it prints are compared here with the reference's KAT, the golden roots and the
    oracle."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "reference_tests")
    csrc = os.path.join(root, "indexed-merkle-tree-halo2_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                        os.path.join(root, "tests", "native", "reference_tests.cpp"), "-L", csrc, "-limt_hip",
                        "-Wl,-rpath," + csrc, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "reference tests: ok" in r.stdout
    kat = next(e for e in GOLD["entries"] if e["kind"] == "hash3" and e["provenance"] == "reference")
    assert f"hash_zero={int(kat['out']):064x}" in r.stdout                  # src/indexed_merkle_tree.rs:248
    rounds = re.findall(r"round (\d) low_leaf_idx=(\d+) new_root=([0-9a-f]{64})", r.stdout)
    assert len(rounds) == len(GOLD["multi_round_depth3"]) == 6
    for (k, low, new_root), g in zip(rounds, GOLD["multi_round_depth3"]):
        assert int(low) == g["low_idx"] and int(new_root, 16) == int(g["new_root"])
    m = re.search(r"insert_leaf new_val=([0-9a-f]{64}) new_root=([0-9a-f]{64})", r.stdout)
    v = int(m.group(1), 16)                                                    # test_insert_leaf's random value
    z = oracle.hash([0, 0, 0])
    leaves = [oracle.hash([0, v, 1]), oracle.hash([v, 0, 0])] + [z] * 6
    rc, ot = oracle.tree_new(ints_to_arr(leaves))
    assert rc == 0 and int(m.group(2), 16) == oracle.tree_root(ot)
    oracle.tree_free(ot)


def test_combine_subtree_roots(imt, ctx, oracle):
    rng = random.Random(41)
    leaves = ints_to_arr([rng.randrange(P) for _ in range(64)])
    rc, ot = oracle.tree_new(leaves)
    sub = oracle.tree_level(ot, 3)                      # 8 subtree roots of height 3
    z = oracle.zero_hashes(32)
    want = oracle.tree_root(ot)
    for l in range(6, 32):
        want = oracle.hash([want, ints(z[l])[0]])
    assert ints(ctx.combine_subtree_roots(sub, 3, 32)) == [want]
    assert ints(ctx.combine_subtree_roots(sub, 3, 6)) == [oracle.tree_root(ot)]
    oracle.tree_free(ot)


# ---------------------------------------------------------------- f1: witness trace
def test_hash_trace_vs_oracle_formats_and_layouts(imt, ctx, oracle):
    """imt_hash_trace_batch against oracle/trace.c on the golden inputs and random ones: every row, all three
    formats, both layouts, ragged batch sizes; the output row equals imt_hash2/3_batch; the column rebuilt from the
    product's own layout satisfies every vertical gate."""
    R256, R261 = (1 << 256) % P, (1 << 261) % P
    for arity in (2, 3):
        gold = [[int(x) for x in g["in"]] for g in GOLD["hash_trace"] if len(g["in"]) == arity]
        rnd = oracle_lib.synth_values(3 * 70, 0x494D5460 + arity)
        items = gold + [[0] * arity, [P - 1] * arity] + [rnd[i * arity:(i + 1) * arity] for i in range(70)]
        inp = np.stack([imt.to_bytes(x) for x in items])
        want = [ints(oracle.hash_trace(x)["witness"]) for x in items]
        rows = 1208 if arity == 2 else 1209
        tr = ctx.hash_trace(inp)
        assert tr.shape == (rows, len(items), 32)
        for i in range(len(items)):
            assert ints(tr[:, i]) == want[i], (arity, i)
        hashes = ctx.hash2(inp) if arity == 2 else ctx.hash3(inp)
        assert (tr[rows - 4] == hashes).all()
        for n in (1, 63, 65):                                   # ragged: partial waves
            assert (ctx.hash_trace(inp[:n]) == tr[:, :n]).all()
        tim = ctx.hash_trace(inp, item_major=True)
        assert tim.shape == (len(items), rows, 32) and (tim.transpose(1, 0, 2) == tr).all()
        # the format applies to inputs and rows alike: MONT256 = halo2curves' in-memory x * 2^256, DEVICE = x * 2^261
        for fmt, r in ((1, R256), (2, R261)):
            inp_f = imt.to_bytes([[(v * r) % P for v in x] for x in items[:6]])
            got = ctx.hash_trace(inp_f, fmt=fmt)
            for i in range(6):
                assert ints(got[:, i]) == [(v * r) % P for v in want[i]], (arity, fmt, i)
        cells, consts, out_row = ctx.hash_trace_layout(arity)
        assert out_row == rows - 4
        for i in (0, len(items) - 1):
            col = imt.rebuild_advice_column(cells, consts, items[i], tr[:, i])
            assert imt.check_vertical_gates(cells, col) == rows
            assert col == ints(oracle.hash_trace(items[i])["cells"])
    for g in GOLD["hash_trace"]:                                # the committed digests, through the GPU
        import hashlib
        xs = [int(x) for x in g["in"]]
        t = ctx.hash_trace(np.stack([imt.to_bytes(xs)]))[:, 0]
        assert hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest() == g["sha256_rows"]
    bad = imt.to_bytes([[1, 2]]).copy()
    bad[0, 1] = 0xff                                            # >= p
    with pytest.raises(imt.ImtError):
        ctx.hash_trace(bad)


def test_path_trace_is_the_trace_of_every_hash_on_the_path(imt, ctx, oracle):
    """imt_path_trace_batch = leaf-hash trace + one hash2 trace per level with the (left, right) inputs dual_mux
    selects (src/indexed_merkle_tree.rs:47-63,78-96); checked against the oracle hash by hash."""
    depth, n = 5, 9
    rng = random.Random(11)
    leaf3 = [[rng.randrange(P) for _ in range(3)] for _ in range(n)]
    sib = [[rng.randrange(P) for _ in range(n)] for _ in range(depth)]
    index = [rng.randrange(1 << depth) for _ in range(n)]
    tr, roots = ctx.path_trace(index, imt.to_bytes(sib), depth, leaf3=imt.to_bytes(leaf3))
    assert tr.shape == (1209 + depth * 1208, n, 32)
    for i in range(n):
        t = oracle.hash_trace(leaf3[i])
        assert ints(tr[:1209, i]) == ints(t["witness"])
        cur, off = oracle.hash(leaf3[i]), 1209
        for l in range(depth):
            pair = [sib[l][i], cur] if (index[i] >> l) & 1 else [cur, sib[l][i]]
            assert ints(tr[off:off + 1208, i]) == ints(oracle.hash_trace(pair)["witness"]), (i, l)
            cur = oracle.hash(pair)
            off += 1208
        assert ints(roots[i]) == [cur]
    # item-major trace = item-major siblings too (one flag bit, as in imt_insert_trace_batch): n > 1 and depth > 1 tell
    # sib[item][level] from sib[level][item]
    sib_im = imt.to_bytes(sib).transpose(1, 0, 2).copy()
    tim, rim = ctx.path_trace(index, sib_im, depth, leaf3=imt.to_bytes(leaf3), item_major=True)
    assert (tim.transpose(1, 0, 2) == tr).all() and (rim == roots).all()
    leaf = [rng.randrange(P) for _ in range(n)]
    tr2, roots2 = ctx.path_trace(index, imt.to_bytes(sib), depth, leaf=imt.to_bytes(leaf))
    assert tr2.shape == (depth * 1208, n, 32)
    assert ints(roots2) == [oracle.path_root(leaf[i], index[i], imt.to_bytes([sib[l][i] for l in range(depth)])) for i in range(n)]


def test_insert_trace_is_what_the_circuit_would_assign(imt, ctx, oracle):
    """imt_insert_trace_batch on real insertion witnesses: the 3 + 4 d traces per insertion, in insert_leaf's call
    order, each equal to the oracle's trace of the hash the circuit computes there; the chains end in the roots."""
    depth, n = 4, 6
    t = imt.IndexedTree(ctx, depth, 16)
    t.insert_batch([77, 5])
    r = t.insert_batch([30, 10, 20, 5000, 50, 35])
    tr = ctx.insert_trace(r["low_leaf"], r["low_index"], r["low_sib"], r["new_leaf"], r["new_index"], r["new_sib"], depth)
    assert tr.shape == (imt.lib.imt_insert_trace_rows(depth), n, 32) and tr.shape[0] == 3 * 1209 + 4 * depth * 1208
    zero_leaf = oracle.hash([0, 0, 0])
    for i in range(n):
        low, new = ints(r["low_leaf"][i]), ints(r["new_leaf"][i])
        li, ni = int(r["low_index"][i]), int(r["new_index"][i])
        chains = [(low, li, r["low_sib"][:, i], ints(r["old_root"][i])[0]),
                  ([low[0], new[0], ni], li, r["low_sib"][:, i], ints(r["interim_root"][i])[0]),
                  (None, ni, r["new_sib"][:, i], ints(r["interim_root"][i])[0]),
                  (new, ni, r["new_sib"][:, i], ints(r["new_root"][i])[0])]
        off = 0
        for pre, idx, sib, root in chains:
            if pre is not None:
                assert ints(tr[off:off + 1209, i]) == ints(oracle.hash_trace(pre)["witness"]), (i, off)
                cur = oracle.hash(pre)
                off += 1209
            else:
                cur = zero_leaf
            for l in range(depth):
                s_ = ints(sib[l])[0]
                pair = [s_, cur] if (idx >> l) & 1 else [cur, s_]
                assert ints(tr[off:off + 1208, i]) == ints(oracle.hash_trace(pair)["witness"]), (i, off, l)
                cur = oracle.hash(pair)
                off += 1208
            assert cur == root
        assert off == tr.shape[0]
    tim = ctx.insert_trace(r["low_leaf"], r["low_index"], r["low_sib"].transpose(1, 0, 2).copy(), r["new_leaf"],
                           r["new_index"], r["new_sib"].transpose(1, 0, 2).copy(), depth, item_major=True)
    assert (tim.transpose(1, 0, 2) == tr).all()
    t.close()


def test_insert_trace_at_depth_32_every_row(imt, ctx, oracle):
    """BASELINE config 5 at the size a k=17 circuit assigns: imt_insert_trace_batch at depth 32 for two REAL insertions
    (158 251 rows each: the call sites src/indexed_merkle_tree.rs:92, :194, :271-275, :299-303) and imt_path_trace_batch
    for their verify_non_inclusion part (:193-204), every row against the oracle's trace of the same hashes, in
    canonical form and in halo2curves' Montgomery form; digests as committed in tests/golden/vectors.json."""
    import hashlib
    depth = 32
    gold = {g["insertion"]: g for g in GOLD["insert_trace_depth32"]}
    vals = oracle_lib.synth_values(40, 0x494D5402)
    t = imt.IndexedTree(ctx, depth, 64)
    t.insert_batch(vals[:38])
    r = t.insert_batch(vals[38:40])
    n = 2
    tr = ctx.insert_trace(r["low_leaf"], r["low_index"], r["low_sib"], r["new_leaf"], r["new_index"], r["new_sib"], depth)
    assert tr.shape == (158251, n, 32)
    R = 1 << 256
    mont = lambda a: oracle_lib.ints_to_arr([x * R % P for x in ints(a)]).reshape(np.asarray(a).shape)
    trm = ctx.insert_trace(mont(r["low_leaf"]), r["low_index"], mont(r["low_sib"]), mont(r["new_leaf"]), r["new_index"],
                           mont(r["new_sib"]), depth, fmt=imt._ffi.FMT_MONT256)
    nm_rows = 1209 + depth * 1208
    ptr, proots = ctx.path_trace(r["low_index"], r["low_sib"], depth, leaf3=r["low_leaf"])
    assert ptr.shape == (nm_rows, n, 32) and (proots == r["old_root"]).all()
    for i in range(n):
        low3, new3 = ints(r["low_leaf"][i]), ints(r["new_leaf"][i])
        want, roots = oracle_lib.insert_leaf_trace(oracle, low3, int(r["low_index"][i]), r["low_sib"][:, i], new3,
                                                   int(r["new_index"][i]), r["new_sib"][:, i], depth)
        assert want.shape == (158251, 32)
        assert (tr[:, i] == want).all(), f"insertion {38 + i}: a trace row differs from the oracle's"
        assert roots == [ints(r["old_root"][i])[0], ints(r["interim_root"][i])[0], ints(r["interim_root"][i])[0],
                         ints(r["new_root"][i])[0]]
        g = gold[38 + i]
        assert hashlib.sha256(np.ascontiguousarray(tr[:, i]).tobytes()).hexdigest() == g["sha256_rows"]
        assert hashlib.sha256(np.ascontiguousarray(trm[:, i]).tobytes()).hexdigest() == g["sha256_rows_mont256"]
        assert (trm[:, i] == oracle_lib.ints_to_arr([x * R % P for x in ints(want)])).all()
        assert (ptr[:, i] == want[:nm_rows]).all()
        assert hashlib.sha256(np.ascontiguousarray(ptr[:, i]).tobytes()).hexdigest() == g["sha256_non_inclusion_rows"]
    t.close()


def test_trace_edge_cases(imt, ctx, oracle):
    """empty batches, depth 0 (the leaf hash alone), wrong arity, device pointers"""
    import ctypes
    import torch
    assert ctx.hash_trace(np.zeros((0, 2, 32), np.uint8)).shape == (1208, 0, 32)
    leaf3 = imt.to_bytes([[5, 6, 7]])
    tr, roots = ctx.path_trace([0], np.zeros((0, 32), np.uint8), 0, leaf3=leaf3)
    assert tr.shape == (1209, 1, 32) and ints(tr[:, 0]) == ints(oracle.hash_trace([5, 6, 7])["witness"])
    assert ints(roots) == [oracle.hash([5, 6, 7])]
    assert imt.lib.imt_hash_trace_rows(4) == 0
    out = np.empty((1208, 1, 32), np.uint8)
    assert imt.lib.imt_hash_trace_batch(ctx.h, leaf3.ctypes.data_as(ctypes.c_void_p), 4, 1, out.ctypes.data_as(ctypes.c_void_p), 0) == imt._ffi.ERR["ARG"]
    assert imt.lib.imt_insert_trace_rows(32) == 3 * 1209 + 128 * 1208
    # device pointers, asynchronous on the context's stream; input errors surface at imt_ctx_sync
    dev = torch.device("cuda", 0)
    c2 = imt.Context(0)
    c2.set_stream(torch.cuda.current_stream().cuda_stream)
    inp = torch.from_numpy(imt.to_bytes([[1, 2], [3, 4]])).to(dev)
    tr_d = torch.empty((1208, 2, 32), dtype=torch.uint8, device=dev)
    rc = imt.lib.imt_hash_trace_batch(c2.h, ctypes.c_void_p(inp.data_ptr()), 2, 2, ctypes.c_void_p(tr_d.data_ptr()), imt._ffi.DEVICE_PTRS)
    assert rc == 0
    c2.sync()
    assert ints(tr_d[:, 1].cpu().numpy()) == ints(oracle.hash_trace([3, 4])["witness"])
    inp[0, 0, :] = 0xff                                                    # >= p
    assert imt.lib.imt_hash_trace_batch(c2.h, ctypes.c_void_p(inp.data_ptr()), 2, 2, ctypes.c_void_p(tr_d.data_ptr()), imt._ffi.DEVICE_PTRS) == 0
    with pytest.raises(imt.ImtError):
        c2.sync()
    c2.close()


def test_hash_trace_2pow14_properties(imt, ctx):
    """2^14 traces in one launch (634 MB of rows): the output row of every item equals imt_hash2_batch, and sampled
    items satisfy every gate of the rebuilt column."""
    n = 1 << 14
    rng = np.random.default_rng(5)
    inp = rng.integers(0, 256, size=(n, 2, 32), dtype=np.uint8)
    inp[:, :, 31] &= 0x0f
    tr = ctx.hash_trace(inp)
    assert (tr[1204] == ctx.hash2(inp)).all()
    cells, consts, _ = ctx.hash_trace_layout(2)
    for i in (0, 4097, n - 1):
        col = imt.rebuild_advice_column(cells, consts, ints(inp[i]), tr[:, i])
        assert imt.check_vertical_gates(cells, col) == 1208


def test_hash_trace_mont256_rows_are_canonical_2pow18(imt, ctx):
    """store_mont256 (imt_trace_device.hpp) decides by the top limbs whether a row leaves as a + m p or a + (m - 32) p
    and falls back to an exact test when they are equal -- about once in 2^25 rows, so a launch of 2^18 hashes (3.2e8
    rows, 10 GB, device pointers) meets that path a few times.  A wrong decision in either direction leaves a row
    outside [0, p): every row of the launch is range-checked on the GPU, the output rows equal imt_hash2_batch, and
    sampled items equal the canonical-format trace times 2^256."""
    import ctypes
    import torch
    n = 1 << 18
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(7)
    inp = torch.randint(0, 256, (n, 2, 32), dtype=torch.uint8, generator=g)
    inp[:, :, 31] &= 0x0f
    d_in = inp.to(dev)
    rows = 1208
    tr = torch.empty((rows, n, 32), dtype=torch.uint8, device=dev)
    c2 = imt.Context(0)
    c2.set_stream(torch.cuda.current_stream().cuda_stream)
    F = imt._ffi
    c2._check(imt.lib.imt_hash_trace_batch(c2.h, ctypes.c_void_p(d_in.data_ptr()), 2, n, ctypes.c_void_p(tr.data_ptr()),
                                           F.DEVICE_PTRS | F.FMT_MONT256))
    out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    c2._check(imt.lib.imt_hash2_batch(c2.h, ctypes.c_void_p(d_in.data_ptr()), ctypes.c_void_p(out.data_ptr()), n,
                                      F.DEVICE_PTRS | F.FMT_MONT256))
    c2.sync()
    assert bool((tr[1204] == out).all())
    # unsigned 256-bit "row < p" on little-endian 64-bit words (sign bit flipped: signed compare = unsigned compare)
    flip = -(1 << 63)
    pw = [((P >> (64 * k)) & ((1 << 64) - 1)) for k in range(4)]
    pw = [x - (1 << 64) if x >= (1 << 63) else x for x in pw]
    pw = [torch.tensor(x, dtype=torch.int64, device=dev) ^ flip for x in pw]
    bad = 0
    for r0 in range(0, rows, 151):
        w = tr[r0:r0 + 151].view(torch.int64).view(-1, n, 4) ^ flip
        lt = w[..., 0] < pw[0]
        for k in (1, 2, 3):
            lt = (w[..., k] < pw[k]) | ((w[..., k] == pw[k]) & lt)
        bad += int((~lt).sum())
    assert bad == 0
    # sampled items against the canonical-format trace of the same inputs
    idx = [0, 77777, n - 1]
    R256 = (1 << 256) % P
    inv = pow(R256, -1, P)                              # the launch read the bytes as x * 2^256
    can = ctx.hash_trace(imt.to_bytes([[v * inv % P for v in ints(inp[i].numpy())] for i in idx]))
    for j, i in enumerate(idx):
        assert ints(tr[:, i].cpu().numpy()) == [v * R256 % P for v in ints(can[:, j])]
    c2.close()


# ---------------------------------------------------------------- BASELINE-size properties
def test_config2_full_size_properties(imt, ctx, oracle):
    """depth 32, 2^16 insertions (BASELINE config 2), checked through size-independent properties:
    (1) every insert_leaf constraint holds for every insertion (witness kernels, independent of the
    sweep); (2) roots chain: old_root[i+1] == new_root[i]; (3) the final root equals an independent
    bulk build (leaf hashes -> dense level kernels -> zero extension); (4) batch-split invariance;
    (5) EVERY one of the 2^16 interim / new roots, low indices and flags equals the sequential CPU oracle's
    (digests of its full run, tests/golden/config2_oracle_digest.json), as do final proofs; a prefix value by value."""
    depth, n = 32, 1 << 16
    vals = oracle_lib.synth_values(n, 0x494D5402)
    t = imt.IndexedTree(ctx, depth, 1 << 17)
    r = t.insert_batch(vals)
    fail = ctx.insert_witness(r["old_root"], r["low_leaf"], r["low_index"], r["low_sib"], r["new_root"],
                              r["new_leaf"], r["new_index"], r["new_sib"], r["is_largest"], depth)
    assert not fail.any()
    assert (r["old_root"][1:] == r["new_root"][:-1]).all()
    assert ints(r["new_root"][-1]) == [t.root()]
    # (3) independent bulk build of the final state
    pre = t.get_leaves(np.arange(1 << 17))
    dense = imt.IndexedMerkleTree.new(ctx, ctx.hash3(pre))
    sub = imt.to_bytes([dense.get_root()])
    assert ints(ctx.combine_subtree_roots(sub, 17, depth)) == [t.root()]
    # (4) same values in 5 uneven batches
    t2 = imt.IndexedTree(ctx, depth, 1 << 17)
    cuts = [0, 1, 1000, 30000, 30001, n]
    roots = [t2.insert_batch(vals[a:b], proofs=False)["new_root"] for a, b in zip(cuts, cuts[1:])]
    assert (np.concatenate(roots) == r["new_root"]).all() and t2.root() == t.root()
    # (5) ALL 65 536 insertions against the sequential CPU oracle, through the committed digests of its full run
    #     (tests/golden/make_config2_digest.py: six and a half minutes of one core, too long for a test)
    import hashlib
    dg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config2_oracle_digest.json")))
    assert dg["n"] == n and dg["depth"] == depth
    assert hashlib.sha256(r["interim_root"].tobytes()).hexdigest() == dg["sha256_interim_roots"]
    assert hashlib.sha256(r["new_root"].tobytes()).hexdigest() == dg["sha256_new_roots"]
    assert hashlib.sha256(r["low_index"].astype("<u8").tobytes()).hexdigest() == dg["sha256_low_index"]
    assert hashlib.sha256(r["is_largest"].tobytes()).hexdigest() == dg["sha256_is_largest"]
    assert t.root() == int(dg["final_root"])
    for k, v in dg["root_after"].items():
        assert ints(r["new_root"][int(k) - 1]) == [int(v)]
    for i, hx in dg["sha256_final_proofs"].items():
        assert hashlib.sha256(t.get_proof_batch([int(i)], item_major=True).tobytes()).hexdigest() == hx
    # (5b) and a prefix value by value
    oh, rows, _ = _oracle_run(oracle, depth, 512, vals[:256])
    assert ints(r["new_root"][:256]) == [o["new_root"] for o in rows]
    assert ints(r["interim_root"][:256]) == [o["interim_root"] for o in rows]
    oracle.sparse_free(oh)


def test_config2_size_descending_values_against_the_oracle_digest(imt, ctx, oracle):
    """one full-size ORDERED case.  The 2^16 config-2 values inserted in DESCENDING order: every
    insertion's low leaf is leaf 0 (the sentinel is rewritten 2^16 times, each new leaf points at the one inserted before
    it) -- the opposite extreme of random values for the sort / lower bound / nearest-smaller search and for the (node, time)
    runs (one run of 2^16 versions per level) that replace update_idx_leaf's scan (src/indexed_merkle_tree.rs:639-658).
    Every interim / new root, low index and flag against the sequential CPU oracle's digests
    (tests/golden/config2_descending_oracle_digest.json, make_config2_digest.py descending: six minutes of one core), in one
    batch and in uneven batches; every insert_leaf constraint through the witness kernels; a prefix value by value."""
    import hashlib
    depth, n = 32, 1 << 16
    vals = sorted(oracle_lib.synth_values(n, 0x494D5402), reverse=True)
    dg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config2_descending_oracle_digest.json")))
    assert dg["n"] == n and dg["depth"] == depth and dg["order"] == "descending"
    t = imt.IndexedTree(ctx, depth, 1 << 17)
    r = t.insert_batch(vals)
    assert (r["low_index"] == 0).all() and r["is_largest"][0] == 1 and not r["is_largest"][1:].any()
    fail = ctx.insert_witness(r["old_root"], r["low_leaf"], r["low_index"], r["low_sib"], r["new_root"],
                              r["new_leaf"], r["new_index"], r["new_sib"], r["is_largest"], depth)
    assert not fail.any()
    assert hashlib.sha256(r["interim_root"].tobytes()).hexdigest() == dg["sha256_interim_roots"]
    assert hashlib.sha256(r["new_root"].tobytes()).hexdigest() == dg["sha256_new_roots"]
    assert hashlib.sha256(r["low_index"].astype("<u8").tobytes()).hexdigest() == dg["sha256_low_index"]
    assert hashlib.sha256(r["is_largest"].tobytes()).hexdigest() == dg["sha256_is_largest"]
    assert t.root() == int(dg["final_root"])
    for k, v in dg["root_after"].items():
        assert ints(r["new_root"][int(k) - 1]) == [int(v)]
    for i, hx in dg["sha256_final_proofs"].items():
        assert hashlib.sha256(t.get_proof_batch([int(i)], item_major=True).tobytes()).hexdigest() == hx
    t2 = imt.IndexedTree(ctx, depth, 1 << 17)
    cuts = [0, 1, 1000, 30000, 30001, n]
    roots = [t2.insert_batch(vals[a:b], proofs=False)["new_root"] for a, b in zip(cuts, cuts[1:])]
    assert (np.concatenate(roots) == r["new_root"]).all() and t2.root() == t.root()
    oh, rows, _ = _oracle_run(oracle, depth, 512, vals[:256])
    assert ints(r["new_root"][:256]) == [o["new_root"] for o in rows]
    assert ints(r["interim_root"][:256]) == [o["interim_root"] for o in rows]
    oracle.sparse_free(oh)


def test_pipelined_growth_stress_properties(imt, ctx):
    """2^18 insertions as 8 pipelined device-pointer batches of 2^15 (L0 grows 15 -> 18 on the way, so
    both the overlapped and the joined schedule run), GPU prepare; every insert_leaf constraint is then
    re-checked by the independent witness kernels, the root chain is continuous and the final root
    equals a bulk rebuild from the snapshot."""
    import ctypes
    import torch
    depth, nb, bs = 32, 8, 1 << 15
    dev = torch.device("cuda", 0)
    c2 = imt.Context(0)
    c2.set_stream(torch.cuda.current_stream().cuda_stream)
    t = imt.IndexedTree(c2, depth, 1 << 19)
    rng = np.random.default_rng(17)
    raw = rng.integers(0, 256, size=(nb * bs, 32), dtype=np.uint8)
    raw[:, 31] &= 0x0f
    raw[:, 0] |= 1                                      # non-zero; 252-bit randoms are distinct w.h.p.
    vals = torch.from_numpy(raw).to(dev)
    outs = []
    flags = imt._ffi.DEVICE_PTRS | imt._ffi.PIPELINE
    for b in range(nb):
        o = dict(low_index=torch.empty(bs, dtype=torch.int64, device=dev),
                 is_largest=torch.empty(bs, dtype=torch.uint8, device=dev),
                 low_leaf=torch.empty((bs, 3, 32), dtype=torch.uint8, device=dev),
                 new_leaf=torch.empty((bs, 3, 32), dtype=torch.uint8, device=dev),
                 old_root=torch.empty((bs, 32), dtype=torch.uint8, device=dev),
                 interim_root=torch.empty((bs, 32), dtype=torch.uint8, device=dev),
                 new_root=torch.empty((bs, 32), dtype=torch.uint8, device=dev),
                 low_sib=torch.empty((depth, bs, 32), dtype=torch.uint8, device=dev),
                 new_sib=torch.empty((depth, bs, 32), dtype=torch.uint8, device=dev))
        st = imt._ffi.InsertOut(**{k: v.data_ptr() for k, v in o.items()})
        rc = imt.lib.imt_itree_insert_batch(t.h, ctypes.c_void_p(vals.data_ptr() + b * bs * 32), bs, ctypes.byref(st), flags)
        assert rc == 0, imt.lib.imt_last_error(c2.h)
        outs.append(o)
    c2.sync()
    torch.cuda.synchronize()
    prev = None
    for b, o in enumerate(outs):
        new_index = torch.arange(1 + b * bs, 1 + (b + 1) * bs, dtype=torch.int64, device=dev)
        fail = torch.empty(bs, dtype=torch.uint8, device=dev)
        P_ = lambda x: ctypes.c_void_p(x.data_ptr())
        rc = imt.lib.imt_insert_witness_batch(c2.h, P_(o["old_root"]), P_(o["low_leaf"]), P_(o["low_index"]), P_(o["low_sib"]),
                                              P_(o["new_root"]), P_(o["new_leaf"]), P_(new_index), None, P_(o["new_sib"]),
                                              P_(o["is_largest"]), depth, bs, P_(fail), None, imt._ffi.DEVICE_PTRS)
        assert rc == 0
        c2.sync()
        assert int(fail.max()) == 0, b
        assert bool((o["old_root"][1:] == o["new_root"][:-1]).all())
        if prev is not None:
            assert bool((o["old_root"][0] == prev).all())
        prev = o["new_root"][-1].clone()
    assert ints(prev.cpu().numpy()) == [t.root()]
    t2 = imt.IndexedTree(c2, depth, 1 << 19)
    t2.load(t.snapshot())
    assert t2.root() == t.root()
    t.close(); t2.close(); c2.close()


def test_config3_non_membership_2pow20_properties(imt, ctx):
    """depth 32, 2^20 non-membership items against the config-2 tree: all accepted, every recomputed
    root equals the tree root, and flipping the candidate to the low leaf's own value is rejected."""
    depth = 32
    t = imt.IndexedTree(ctx, depth, 1 << 17)
    t.insert_batch(oracle_lib.synth_values(1 << 16, 0x494D5402), proofs=False)
    root = t.root()
    rng = np.random.default_rng(3)
    cand = rng.integers(0, 256, size=(1 << 20, 32), dtype=np.uint8)
    cand[:, 31] &= 0x0f                                    # < 2^252 < p, non-zero w.h.p., not members w.h.p.
    low, leaves, sib, largest = t.non_membership_witness(cand)
    fail, rout = ctx.non_membership(imt.to_bytes(root), leaves, low, sib, depth, cand, largest, want_root=True)
    assert not fail.any()
    assert (rout == imt.to_bytes(root)[None]).all()
    fail = ctx.non_membership(imt.to_bytes(root), leaves, low, sib, depth, leaves[:, 0, :].copy(), largest)
    assert (fail & imt._ffi.F_LOW_LT_NEW).all()


def test_config3_mixed_batch_2pow20_against_the_oracle_digest(imt, ctx):
    """BASELINE config 3 at its full size AGAINST THE ORACLE, item by item: 2^20 verify_non_inclusion items against the
    config-2 tree, a mixed batch (tests/golden/config3_mix.py: honest ones, members shown with their own leaf or their
    predecessor, zero, a wrong is_largest, a forged sibling, a wrong path position, a changed preimage).  The CPU oracle's
    fail mask and recomputed root of every item were computed in the build container
    (tests/golden/make_config3_digest.py, ~35 M CPU hashes) and committed as digests; so were the honest witnesses
    (low index, preimage, is_largest, all 32 siblings of all 2^20 items), which pins imt_itree_non_membership_witness at
    this size too.  /root/reference/src/indexed_merkle_tree.rs:127-229."""
    import hashlib
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import config3_mix as C
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config3_oracle_digest.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    t = imt.IndexedTree(ctx, C.DEPTH, 1 << 17)
    t.insert_batch(oracle_lib.synth_values(C.N_TREE, C.TREE_SEED), proofs=False)
    root = t.root()
    assert str(root) == gold["tree_root"]
    cand = C.candidates()
    low, leaves, sib, largest = t.non_membership_witness(cand)
    assert sha(low.astype("<u8")) == gold["sha256_honest_low_index"] and sha(leaves) == gold["sha256_honest_low_leaf"]
    assert sha(largest) == gold["sha256_honest_is_largest"] and sha(sib) == gold["sha256_honest_siblings"]
    cls = C.mix(cand, low, leaves, sib, largest)
    fail, rout = ctx.non_membership(imt.to_bytes(root), leaves, low, sib, C.DEPTH, cand, largest, want_root=True)
    per_class = {str(c): {f"0x{int(k):02x}": int((fail[cls == c] == k).sum()) for k in np.unique(fail[cls == c])} for c in sorted(set(cls.tolist()))}
    assert per_class == {c: v["masks"] for c, v in gold["per_class"].items()}, per_class       # says WHICH class differs, if one does
    assert sha(fail) == gold["sha256_fail_masks"]
    assert sha(rout) == gold["sha256_recomputed_roots"]
    t.close()


def _load_sharded():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("imt_sharded", os.path.join(root, "indexed-merkle-tree-halo2_amd",
                                                                               "sharded.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _witness_fail(imt, be, o, depth):
    """imt_insert_witness_batch at the FULL depth on a lifted output set of GpuBackend, global indices"""
    import ctypes
    import torch
    P_ = lambda x: ctypes.c_void_p(x.data_ptr())
    n = o["new_root"].shape[0]
    fail = torch.empty(n, dtype=torch.uint8, device=o["new_root"].device)
    new_index = torch.arange(o["first_new_index"], o["first_new_index"] + n, dtype=torch.int64, device=fail.device)
    rc = imt.lib.imt_insert_witness_batch(be.ctx.h, P_(o["old_root"]), P_(o["low_leaf"]), P_(o["low_index"]),
                                          P_(o["low_sib"]), P_(o["new_root"]), P_(o["new_leaf"]), P_(new_index), None,
                                          P_(o["new_sib"]), P_(o["is_largest"]), depth, n, P_(fail), None,
                                          imt._ffi.DEVICE_PTRS)
    assert rc == 0, imt.lib.imt_last_error(be.ctx.h)
    be.sync()
    return fail


@pytest.mark.parametrize("world,depth", [(4, 8), (2, 6), (8, 8)])
def test_subtree_lift_matches_dense_replay(imt, ctx, oracle, world, depth):
    """Subtree sharding end to end through sharded.GpuBackend, `world` shards one after the other on one GPU:
    every lifted witness (global indices, depth-`depth` roots and proofs) equals a dense rebuild of the WHOLE
    tree (the oracle's IndexedMerkleTree::new) after every single event of the global order, and passes
    imt_insert_witness_batch at the full depth."""
    import torch
    from sharded_ref import dense_global_replay
    sharded = _load_sharded()
    n_step, steps_n = 5, 3
    k = world.bit_length() - 1
    raw = oracle_lib.synth_values(n_step * steps_n * world * 6, 0x494D5440 + world)
    vals = [[v for v in raw if v % world == g][:n_step * steps_n] for g in range(world)]
    steps = [[vals[g][st * n_step:(st + 1) * n_step] for g in range(world)] for st in range(steps_n)]
    want, final_root = dense_global_replay(oracle, depth, world, steps)
    be = [sharded.GpuBackend(imt, 0, depth, world, g, 1 << (depth - k), n_step, pipeline=(g % 2 == 0)) for g in range(world)]
    roots_prev = torch.stack([be[0].initial_root()] * world)
    for st in range(steps_n):
        slots = [be[g].insert(imt.to_bytes(steps[st][g])) for g in range(world)]
        roots_after = torch.stack([be[g].root_after(0).clone() for g in range(world)])
        for g in range(world):
            be[g].lift(slots[g], roots_prev, roots_after)
            o = be[g].outputs(slots[g])
            assert int(_witness_fail(imt, be[g], o, depth).max()) == 0
            h = {key: (v.cpu().numpy() if torch.is_tensor(v) else v) for key, v in o.items()}
            for i, exp in enumerate(want[st][g]):
                assert int(h["low_index"][i]) == exp["low"] and h["first_new_index"] + i == exp["new_index"]
                assert ints(h["low_leaf"][i]) == exp["low_leaf"] and ints(h["new_leaf"][i]) == exp["new_leaf"]
                assert int(h["is_largest"][i]) == exp["largest"]
                for key in ("old_root", "interim_root", "new_root"):
                    assert ints(h[key][i]) == [exp[key]], (st, g, i, key)
                assert (h["low_sib"][:, i] == exp["low_proof"]).all(), (st, g, i)
                assert (h["new_sib"][:, i] == exp["new_proof"]).all(), (st, g, i)
        roots_prev = roots_after
    assert ints(be[0].combine(roots_prev).cpu().numpy()) == [final_root]
    # snapshot of a placed tree: global next_idx fields; reload reproduces the subtree root
    g = world - 1
    snap = be[g].tree.snapshot()
    nxt = [x for x in ints(snap[:, 2]) if x]
    assert nxt and all((g << (depth - k)) < x < ((g + 1) << (depth - k)) for x in nxt)
    t2 = imt.IndexedTree(be[g].ctx, depth - k, 1 << (depth - k))
    t2.set_placement(depth, g)
    t2.load(snap)
    assert t2.root() == be[g].tree.root()
    # the value partition is enforced for EVERY value of a batch, on the GPU (k_scatter)
    fresh = [vals[0][j] + world for j in range(n_step - 1)]
    size0 = be[0].tree.size
    with pytest.raises(ValueError, match="another subtree"):
        be[0].insert(imt.to_bytes(fresh + [vals[1][1]]))           # only the last one is misrouted
    assert be[0].tree.size == size0
    be[0].insert(imt.to_bytes(fresh + [vals[0][0] + 2 * world]))   # the tree is still usable afterwards
    be[0].sync()
    assert be[0].tree.size == size0 + n_step
    for b in be:
        b.tree.close(); b.ctx.close()


def test_montgomery_inputs_ready_with_lagged_lift(imt, ctx):
    """halo2curves' format + IMT_INPUTS_READY + IMT_PIPELINE through the sharded driver with the lift one step behind
    (bench.py's N > 1 schedule): the canonical copy of a batch's values is made on the side stream, unordered behind the
    context's stream on which the previous step's imt_itree_lift_batch still runs -- it must not share scratch with it
    (it has the plan's own buffer).  Many small steps back to back; everything equals the canonical-format run."""
    import torch
    sharded = _load_sharded()
    depth, world, n_step, steps_n = 32, 2, 96, 12
    R = 1 << 256
    raw = oracle_lib.synth_values(n_step * steps_n * 3, 0x494D5447)
    runs = {}
    for fmt in (0, imt._ffi.FMT_MONT256):
        outs = []
        trees = []
        for g in range(world):
            vals = [v for v in raw if v % world == g][:n_step * steps_n]
            be = sharded.GpuBackend(imt, 0, depth, world, g, 1 << 12, n_step, pipeline=True, inputs_ready=True, nbuf=3, fmt=fmt)
            trees.append((be, vals))
        # one process stands in for both ranks: the "all-gather" is a stack of the two lagged roots
        prev = torch.stack([trees[0][0].initial_root()] * world)
        pending = None
        for st in range(steps_n + 1):
            slots = None
            if st < steps_n:
                slots = []
                for be, vals in trees:
                    chunk = vals[st * n_step:(st + 1) * n_step]
                    arr = oracle_lib.ints_to_arr([v * R % P for v in chunk] if fmt else chunk)
                    slots.append(be.insert(torch.from_numpy(arr).cuda()))
            if pending is not None:
                lag = 1 if st < steps_n else 0
                after = torch.stack([be.root_after(lag).clone() for be, _ in trees])
                for (be, _), slot in zip(trees, pending):
                    be.lift(slot, prev, after)
                    be.sync()
                    o = be.outputs(slot)
                    outs.append({k: o[k].cpu().numpy().copy() for k in ("old_root", "new_root", "low_sib", "low_leaf")})
                prev = after
            pending = slots
        runs[fmt] = outs
        for be, _ in trees:
            be.tree.close(); be.ctx.close()
    unmont = lambda a: [x * pow(R, -1, P) % P for x in ints(a)]
    assert len(runs[0]) == world * steps_n
    for a, b in zip(runs[0], runs[imt._ffi.FMT_MONT256]):
        for k in a:
            assert ints(a[k]) == unmont(b[k]), k


def test_subtree_layout_needs_the_owner_constraint(imt, ctx, oracle):
    """The soundness note of DESIGN.md 8b / INTEGRATION.md sec. 4 as an executable fact.  Two value-partitioned subtrees
    under one depth-32 root: 10 (even) is stored in subtree 0, 11 (odd) in subtree 1.  A non-inclusion witness for 10
    built from subtree 1's SENTINEL {0, 11, idx} passes every constraint of the reference's verify_non_inclusion
    (src/indexed_merkle_tree.rs:127-229: range predicates, low leaf in root) although 10 is in the tree -- the reference's
    circuit never ties a leaf's position to a value.  The library refuses to PRODUCE that witness; the extra constraint
    `low_index >> (32 - k) == v mod 2^k` (chip.rs::constrain_owner_subtree) rejects it; the honest witness from subtree
    0 cannot exist (10 is present).  In the single-list layout (sliced.py) none of this arises."""
    import ctypes
    depth, k = 32, 1
    sub = depth - k
    trees = []
    for g in range(2):
        t = imt.IndexedTree(ctx, sub, 64)
        t.set_placement(depth, g)
        ctx._check(imt.lib.imt_itree_set_value_partition(t.h, 2, g))
        trees.append(t)
    trees[0].insert_batch([10, 1000])
    trees[1].insert_batch([11, 1001])
    roots = ints_to_arr([trees[0].root(), trees[1].root()])
    global_root = ctx.combine_subtree_roots(roots, sub, depth)
    # the library: a candidate of the other residue has no witness in this list
    with pytest.raises(ValueError, match="another subtree"):
        trees[1].non_membership_witness([10], subtree_roots=roots)
    with pytest.raises(ValueError):
        trees[0].non_membership_witness([10], subtree_roots=roots)         # present in its own list
    # the forged witness: subtree 1's sentinel as the low leaf, its proof lifted to the global root by hand
    base1 = 1 << sub
    idx = np.array([base1], dtype=np.uint64)
    leaf = trees[1].get_leaves(idx)
    assert ints(leaf[0]) == [0, 11, base1 + 1]
    sib = np.zeros((depth, 1, 32), np.uint8)
    sib[:sub] = trees[1].get_proof_batch(idx)
    out = imt._ffi.InsertOut(low_sib=sib.ctypes.data)
    ctx._check(imt.lib.imt_itree_lift_batch(trees[1].h, roots.ctypes.data_as(ctypes.c_void_p), roots.ctypes.data_as(ctypes.c_void_p),
                                            2, 1, ctypes.byref(out), 0))
    fail = ctx.non_membership(global_root, leaf, idx, sib, depth, imt.to_bytes([10]), [0])
    assert int(fail[0]) == 0, "the reference's constraints alone reject it after all?"
    # the oracle's restatement of verify_non_inclusion agrees: satisfied
    helper = ints_to_arr([1 - ((base1 >> l) & 1) for l in range(depth)])
    assert oracle.verify_non_inclusion(int.from_bytes(bytes(global_root), "little"), ints(leaf[0]), sib[:, 0].copy(), helper, 10, 0)[0] == 0
    # the owner constraint a circuit over this layout must add
    assert (int(idx[0]) >> sub) != 10 % 2          # violated by the forged witness ...
    low, leaves, hsib, largest = trees[1].non_membership_witness([13], subtree_roots=roots)
    assert (int(low[0]) >> sub) == 13 % 2          # ... and satisfied by an honest one
    assert not ctx.non_membership(global_root, leaves, low, hsib, depth, imt.to_bytes([13]), largest).any()
    for t in trees:
        t.close()


def test_lift_batch_host_pointers_and_item_major(imt, ctx, oracle):
    """imt_itree_lift_batch with host pointers, both sibling layouts, equal to the device-pointer path; a
    placed tree's low-leaf queries speak global indices."""
    depth, world, g, n = 10, 4, 2, 7
    sub = depth - 2
    vals = [v for v in oracle_lib.synth_values(200, 0x494D5441) if v % world == g][:2 * n]
    rng = random.Random(5)
    before = ints_to_arr([rng.randrange(P) for _ in range(world)])
    after = ints_to_arr([rng.randrange(P) for _ in range(world)])
    res = {}
    for item_major in (False, True):
        t = imt.IndexedTree(ctx, sub, 64)
        t.set_placement(depth, g)
        t.insert_batch(vals[:n], item_major=item_major)
        r = t.insert_batch(vals[n:], item_major=item_major)
        assert r["low_sib"].shape == ((n, depth, 32) if item_major else (depth, n, 32))
        sub_new_root = r["new_root"].copy()
        t.lift_batch(r, before, after, item_major=item_major)
        res[item_major] = r
        # the climb, restated with the oracle: ranks below g after the step, ranks above before it
        mixed = [ints(after)[q] if q < g else ints(before)[q] for q in range(world)]
        s0, s1 = mixed[g ^ 1], oracle.hash([mixed[0], mixed[1]])
        for i in range(n):
            x = oracle.hash([ints(sub_new_root[i])[0], s0])          # g = 2: left child, then right child
            assert ints(r["new_root"][i]) == [oracle.hash([s1, x])]
        sib = r["new_sib"] if not item_major else r["new_sib"].transpose(1, 0, 2)
        assert ints(sib[sub, 0]) == [s0] and ints(sib[sub + 1, n - 1]) == [s1]
        low = t.find_low([vals[0] + world])                         # global index of the low leaf
        assert (g << sub) <= int(low[0]) < ((g + 1) << sub)
        assert (t.get_leaves(low)[0, 0] == imt.to_bytes(vals[0])).all()
        lw, leaves, sibs, largest = t.non_membership_witness([vals[0] + world])
        assert int(lw[0]) == int(low[0]) and sibs.shape == (sub, 1, 32)
        t.close()
    assert (res[True]["low_sib"].transpose(1, 0, 2) == res[False]["low_sib"]).all()
    assert (res[True]["old_root"] == res[False]["old_root"]).all()


def test_sharded_non_membership_at_depth_32(imt, ctx, oracle):
    """BASELINE config 3 in the sharded layout: four value-partitioned subtrees of height 30 (one GPU, one after the
    other), non-membership witnesses for values none of them holds, lifted to depth 32 with the four subtree roots:
    every item passes verify_non_inclusion (imt_non_membership_batch, all constraints) against the GLOBAL root with
    its global low-leaf index; a value that IS in the tree is refused, one of another rank's residue too."""
    depth, world, n_ins, n_q = 32, 4, 300, 200
    sub = depth - 2
    pool = oracle_lib.synth_values(6 * (n_ins + n_q), 0x494D5443)
    trees, roots, held = [], [], []
    for g in range(world):
        mine = [v for v in pool if v % world == g]
        t = imt.IndexedTree(ctx, sub, 1 << 10)
        t.set_placement(depth, g)
        ctx._check(imt.lib.imt_itree_set_value_partition(t.h, world, g))
        t.insert_batch(mine[:n_ins], proofs=False)
        trees.append(t)
        roots.append(t.root())
        held.append(mine)
    roots_arr = ints_to_arr(roots)
    global_root = ctx.combine_subtree_roots(roots_arr, sub, depth)
    # the oracle builds the same four subtrees and agrees on the global root
    oroots = []
    for g in range(world):
        oh = oracle.sparse_new(sub, 1 << 10)
        oracle.sparse_set_index_base(oh, g << sub)
        for v in held[g][:n_ins]:
            oracle.sparse_insert(oh, sub, v)
        oroots.append(oracle.sparse_root(oh))
        oracle.sparse_free(oh)
    assert oroots == roots
    top = oracle.hash([oracle.hash(oroots[0:2]), oracle.hash(oroots[2:4])])
    assert ints(global_root) == [top]
    for g in range(world):
        q = held[g][n_ins:n_ins + n_q]
        low, leaves, sib, largest = trees[g].non_membership_witness(q, subtree_roots=roots_arr)
        assert sib.shape == (depth, n_q, 32)
        assert ((low >> sub) == g).all()                               # global indices inside subtree g
        fail = ctx.non_membership(global_root, leaves, low, sib, depth, imt.to_bytes(q), largest)
        assert not fail.any()
        # the two appended siblings are the neighbouring subtree's root and the other pair's hash
        assert ints(sib[sub, 0]) == [oroots[g ^ 1]]
        other = oracle.hash(oroots[2:4]) if g < 2 else oracle.hash(oroots[0:2])
        assert ints(sib[sub + 1, n_q - 1]) == [other]
        with pytest.raises(ValueError):
            trees[g].non_membership_witness([held[g][0]], subtree_roots=roots_arr)          # present
        with pytest.raises(ValueError):
            trees[g].non_membership_witness([held[g ^ 1][n_ins]], subtree_roots=roots_arr)  # another rank's value
    for t in trees:
        t.close()


def test_lift_above_the_subtrees_meets_empty_subtrees_and_world_one_is_a_no_op(imt, ctx, oracle):
    """global depth > subtree height + log2(n_subtrees): the levels above the subtrees' common root climb against
    Z[l]; a single subtree of full height is lifted by nothing.  Also: batch_abort, a non-power-of-two value
    partition on both prepare paths."""
    sub, world, g, depth, n = 5, 4, 1, 10, 6                     # k = 2, three more levels against Z[7], Z[8], Z[9]
    vals = [v for v in oracle_lib.synth_values(200, 0x494D5442) if v % 3 == 2][:2 * n]
    z = ints(oracle.zero_hashes(depth))
    for host_prep in (False, True):
        t = imt.IndexedTree(ctx, sub, 32)
        t.set_placement(depth, g)
        assert imt.lib.imt_itree_set_value_partition(t.h, 3, 2) == 0       # modulus 3: mod_small, not a mask
        r = t.insert_batch(vals[:n], host_prep=host_prep)
        with pytest.raises(ValueError, match="another subtree"):
            t.insert_batch([vals[0] + 1], host_prep=host_prep)
        sub_new = ints(r["new_root"])
        before = ints_to_arr([oracle.hash([q, 1]) for q in range(world)])
        after = ints_to_arr([oracle.hash([q, 2]) for q in range(world)])
        t.lift_batch(r, before, after)
        mixed = [ints(after)[q] if q < g else ints(before)[q] for q in range(world)]
        for i in range(n):
            x = oracle.hash([mixed[0], sub_new[i]])                          # g = 1: right child at the first level
            x = oracle.hash([x, oracle.hash([mixed[2], mixed[3]])])          # then left child
            for l in range(sub + 2, depth):
                x = oracle.hash([x, z[l]])
            assert ints(r["new_root"][i]) == [x], (host_prep, i)
        assert ints(r["new_sib"][sub + 2:, 0]) == z[sub + 2:depth] and ints(r["low_sib"][sub + 1, n - 1]) == [
            oracle.hash([mixed[2], mixed[3]])]
        fail = ctx.insert_witness(r["old_root"], r["low_leaf"], r["low_index"], r["low_sib"], r["new_root"], r["new_leaf"],
                                  r["new_index"], r["new_sib"], r["is_largest"], depth)
        assert not fail.any()
        t.close()
    t = imt.IndexedTree(ctx, 6, 32)
    t.set_placement(6, 0)                                          # world 1: the tree IS the global tree
    r = t.insert_batch(vals[:n])
    keep = {k: v.copy() for k, v in r.items()}
    t.lift_batch(r, imt.to_bytes([[0]]).reshape(1, 32), imt.to_bytes([[0]]).reshape(1, 32))
    assert all((r[k] == keep[k]).all() for k in keep)
    # an open sharded batch can be given up; the tree is as before
    import ctypes
    ev, l0 = ctypes.c_uint32(), ctypes.c_uint32()
    v = imt.to_bytes(vals[n:n + 4])
    assert imt.lib.imt_itree_batch_begin(t.h, v.ctypes.data_as(ctypes.c_void_p), 4, 0, ctypes.byref(ev), ctypes.byref(l0)) == 0
    assert imt.lib.imt_itree_insert_batch(t.h, v.ctypes.data_as(ctypes.c_void_p), 4, None, 0) == imt._ffi.ERR["ARG"]
    assert imt.lib.imt_itree_batch_abort(t.h) == 0
    root0, size0 = t.root(), t.size
    r2 = t.insert_batch(vals[n:n + 4])
    oh, rows, _ = _oracle_run(oracle, 6, 32, vals[:n + 4])
    assert size0 == n + 1 and ints(r2["new_root"]) == [o["new_root"] for o in rows[n:]]
    oracle.sparse_free(oh)
    with pytest.raises(imt.ImtError):
        t.set_placement(6, 0)                                      # placement only on an empty tree
    t.close()


def test_config4_eight_shards_2pow22_properties(imt, ctx, oracle):
    """BASELINE config 4 on one GPU: 2^22 insertions as 8 value-partitioned shards (v mod 8) x 8 steps x 2^16,
    height-29 subtrees through sharded.GpuBackend (pipelined device-pointer batches), the roots exchanged per
    step as the all-gather does and every batch lifted to depth 32.  Every insert_leaf constraint of every
    insertion holds AT DEPTH 32 with global leaf indices (independent witness kernels: 131 hashes each); the
    global root sequence is continuous inside a batch, from rank to rank inside a step and from step to step;
    the final root equals the combination of the subtree roots (checked against the oracle, 7 hashes) and every
    subtree equals a bulk rebuild from its snapshot."""
    import torch
    sharded = _load_sharded()
    shards, depth, steps_n, bs = 8, 32, 8, 1 << 16
    sub_depth = depth - 3
    dev = torch.device("cuda", 0)
    be = [sharded.GpuBackend(imt, 0, depth, shards, g, 1 << 20, bs, pipeline=True) for g in range(shards)]
    vals = []
    for s in range(shards):
        rng = np.random.default_rng(400 + s)
        raw = rng.integers(0, 256, size=(steps_n * bs, 32), dtype=np.uint8)
        raw[:, 31] &= 0x0f
        raw[:, 0] = (raw[:, 0] & 0xf8) | s                 # this shard's residue class mod 8
        raw[:, 1] |= 1                                      # non-zero
        vals.append(torch.from_numpy(raw).to(dev))
    roots_prev = torch.stack([be[0].initial_root()] * shards).to(dev)
    last_root = None
    for st in range(steps_n):
        slots = [be[g].insert(vals[g][st * bs:(st + 1) * bs]) for g in range(shards)]
        roots_after = torch.stack([be[g].root_after(0).clone() for g in range(shards)])
        for g in range(shards):
            be[g].lift(slots[g], roots_prev, roots_after)
            o = be[g].outputs(slots[g])
            assert o["first_new_index"] == (g << sub_depth) + 1 + st * bs
            assert int(_witness_fail(imt, be[g], o, depth).max()) == 0, (st, g)
            assert bool((o["old_root"][1:] == o["new_root"][:-1]).all())
            if last_root is not None:
                assert bool((o["old_root"][0] == last_root).all()), (st, g)      # rank to rank, step to step
            last_root = o["new_root"][-1].clone()
        roots_prev = roots_after
    top = ints(be[0].combine(roots_prev).cpu().numpy())[0]
    assert ints(last_root.cpu().numpy()) == [top]
    roots = ints(roots_prev.cpu().numpy())
    lvl = roots
    while len(lvl) > 1:
        lvl = [oracle.hash([lvl[i], lvl[i + 1]]) for i in range(0, len(lvl), 2)]
    assert top == lvl[0]
    for g in (0, 5):
        assert be[g].tree.size == steps_n * bs + 1 and be[g].tree.root() == roots[g]
        t2 = imt.IndexedTree(be[g].ctx, sub_depth, 1 << 20)
        t2.set_placement(depth, g)
        t2.load(be[g].tree.snapshot())
        assert t2.root() == roots[g]
        t2.close()
    for b in be:
        b.tree.close(); b.ctx.close()


def test_library_loaded_before_torch_leaves_torch_usable():
    """The mirror package loaded BEFORE torch touches the GPU (fresh process): both must then see the
    device -- PyTorch-ROCm bundles its own HIP runtime with the same soname as the one libimt_hip.so
    links against (see _ffi._preload_torch_hip_runtime)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, imt_amd\n"
        "assert 'torch' not in sys.modules\n"
        "c = imt_amd.Context(0)\n"
        "h = c.hash2(np.zeros((4, 2, 32), np.uint8))\n"
        "import torch\n"
        "s = torch.cuda.current_stream().cuda_stream\n"
        "x = torch.arange(8, device='cuda:0').sum().item()\n"
        "assert x == 28\n"
        "c.set_stream(s)\n"
        "h2 = c.hash2(np.zeros((4, 2, 32), np.uint8))\n"
        "assert (h == h2).all()\n"
        "print('ok')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_pinned_host_buffers_as_device_pointers(imt, ctx, oracle):
    """imt_host_alloc: values read from and witnesses written to page-locked HOST memory by the kernels
    (IMT_DEVICE_PTRS | IMT_PIPELINE, three batches in flight), equal to the synchronous host-pointer
    call and to the oracle."""
    import ctypes
    depth, n, nb = 32, 1024, 3
    vals = oracle_lib.synth_values(nb * n, 0x494D5407)
    c2 = imt.Context(0)
    t_ref = imt.IndexedTree(c2, depth, 8192)
    want = [t_ref.insert_batch(vals[i * n:(i + 1) * n]) for i in range(nb)]
    t = imt.IndexedTree(c2, depth, 8192)
    pv = c2.host_alloc((nb * n, 32))
    pv[:] = imt.to_bytes(vals)
    outs = []
    for b in range(nb):
        o = dict(low_index=c2.host_alloc(n, np.uint64), is_largest=c2.host_alloc(n),
                 low_leaf=c2.host_alloc((n, 3, 32)), new_leaf=c2.host_alloc((n, 3, 32)),
                 old_root=c2.host_alloc((n, 32)), interim_root=c2.host_alloc((n, 32)), new_root=c2.host_alloc((n, 32)),
                 low_sib=c2.host_alloc((depth, n, 32)), new_sib=c2.host_alloc((depth, n, 32)))
        for a in o.values():
            a[...] = 0xee
        st = imt._ffi.InsertOut(**{k: v.ctypes.data for k, v in o.items()})
        rc = imt.lib.imt_itree_insert_batch(t.h, ctypes.c_void_p(pv.ctypes.data + b * n * 32), n, ctypes.byref(st),
                                            imt._ffi.DEVICE_PTRS | imt._ffi.PIPELINE)
        assert rc == 0, imt.lib.imt_last_error(c2.h)
        outs.append(o)
    c2.sync()
    for b, o in enumerate(outs):
        for k in o:
            assert (np.asarray(o[k]) == want[b][k]).all(), (b, k)
    assert t.root() == t_ref.root()
    oh, rows, _ = _oracle_run(oracle, depth, 8192, vals[:64])
    assert ints(outs[0]["new_root"][:64]) == [r["new_root"] for r in rows]
    oracle.sparse_free(oh)
    for o in outs:
        for a in o.values():
            c2.host_free(a)
    c2.host_free(pv)
    t.close(); t_ref.close(); c2.close()
