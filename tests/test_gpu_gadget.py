"""GPU (MI355X): f3 -- imt_less_than_trace_batch / imt_insert_gadget_trace_batch / imt_insert_column_segments against the
CPU oracle (oracle/gadget.c) and the committed digests: every new advice value the reference's insert_leaf
(/root/reference/src/indexed_merkle_tree.rs:231-314) assigns OUTSIDE hash_fix_len_array, in assignment order.  Together
with the f1 traces (imt_insert_trace_batch) that is the whole advice column: the last test walks the segment table and
checks that the two traces cover it.  ORDER UNPINNED BY THE REFERENCE (halo2-base is un-vendored), like f1."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from oracle_lib import P  # noqa: E402
from test_gadget_cpu import EDGE  # noqa: E402

pytestmark = pytest.mark.gpu
R256 = (1 << 256) % P
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "gadget_digest.json")))


@pytest.mark.parametrize("lookup_bits", [18, 8])
def test_less_than_trace_vs_oracle_formats_and_layouts(imt, ctx, oracle, lookup_bits):
    import random
    rng = random.Random(31 + lookup_bits)
    pairs = EDGE + [(rng.randrange(P), rng.randrange(P)) for _ in range(200)]
    pairs += [((q << 128) + rng.randrange(1 << 128), (q << 128) + rng.randrange(1 << 128)) for q in [rng.randrange(1 << 125) for _ in range(60)]]
    pairs += [(x, x) for x in (rng.randrange(P) for _ in range(20))]
    a, b = imt.to_bytes([p[0] for p in pairs]), imt.to_bytes([p[1] for p in pairs])
    want = np.stack([oracle.less_than_trace(x, y, lookup_bits)["witness"] for x, y in pairs], axis=1)     # [rows, n, 32]
    got, lt = ctx.less_than_trace(a, b, lookup_bits)
    assert got.shape == want.shape and (got == want).all()
    assert list(lt) == [1 if x < y else 0 for x, y in pairs]
    got_im, _ = ctx.less_than_trace(a, b, lookup_bits, item_major=True)
    assert (got_im == want.transpose(1, 0, 2)).all()
    F = imt._ffi
    am, bm = imt.to_bytes([x * R256 % P for x, _ in pairs]), imt.to_bytes([y * R256 % P for _, y in pairs])
    got_m, lt_m = ctx.less_than_trace(am, bm, lookup_bits, fmt=F.FMT_MONT256)
    assert oracle_lib.arr_ints(got_m) == [v * R256 % P for v in oracle_lib.arr_ints(want)] and (lt_m == lt).all()
    # the layout the library reports rebuilds a column whose every gate holds (the chip's view of one comparison)
    cells, consts, out_row = ctx.less_than_layout(lookup_bits)
    for j in (0, 3, len(EDGE) + 5, len(pairs) - 1):
        x, y = pairs[j]
        col = imt.rebuild_advice_column(cells, consts, [x >> 128, x & ((1 << 128) - 1), y >> 128, y & ((1 << 128) - 1)], got[:, j])
        assert imt.check_vertical_gates(cells, col) == int(cells["gate"].sum())
        assert oracle_lib.arr_ints(got[out_row, j])[0] == (1 if x < y else 0)
    # the rows the RangeChip would send to its lookup table are the lookup_bits-wide limbs, for all 2^8+ pairs at once
    lk = ctx.less_than_lookup_rows(lookup_bits)
    limbs = got[lk]                                          # [2 L, n, 32]
    assert (limbs[:, :, 4:] == 0).all()
    assert (limbs[:, :, :4].copy().view("<u4")[..., 0] < (1 << lookup_bits)).all()
    if lookup_bits == GOLD["lookup_bits"]:
        for g in GOLD["less_than"]:
            r, _ = ctx.less_than_trace(imt.to_bytes([int(g["a"])]), imt.to_bytes([int(g["b"])]), lookup_bits)
            assert hashlib.sha256(np.ascontiguousarray(r[:, 0]).tobytes()).hexdigest() == g["sha256_rows"]


def test_less_than_trace_refuses_bad_arguments(imt, ctx):
    F, lib = imt._ffi, imt.lib
    big = np.frombuffer(P.to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32).copy()
    one = imt.to_bytes([1])
    with pytest.raises(imt.ImtError) as e:
        ctx.less_than_trace(big, one)
    assert e.value.code == F.ERR["NONCANONICAL"]
    with pytest.raises(imt.ImtError) as e:
        ctx.less_than_trace(one, one, lookup_bits=29)
    assert e.value.code == F.ERR["RANGE"]
    assert lib.imt_less_than_trace_rows(0) == 0 and lib.imt_less_than_trace_rows(18) == 63
    assert lib.imt_insert_gadget_rows(0, 18) == 0 and lib.imt_insert_gadget_rows(32, 18) == 658


def _real_insertions(imt, ctx, depth, vals):
    t = imt.IndexedTree(ctx, depth, 1 << min(depth, 10))
    res = t.insert_batch(vals)
    t.close()
    return res


@pytest.mark.parametrize("depth,vals", [(3, [30, 10, 20, 5, 50, 35]), (32, None)])
def test_insert_gadget_trace_vs_oracle(imt, ctx, oracle, depth, vals):
    vals = vals or oracle_lib.synth_values(40, 0x494D54B2)
    res = _real_insertions(imt, ctx, depth, vals)
    n = len(vals)
    got = ctx.insert_gadget_trace(res["low_leaf"], res["low_index"], res["low_sib"], res["new_leaf"], res["new_index"],
                                  res["new_sib"], res["is_largest"], depth)
    K = 63
    assert got.shape == (20 + 2 * K + 16 * depth, n, 32)
    want_segs = None
    for i in range(n):
        low3, new3 = oracle_lib.arr_ints(res["low_leaf"][i]), oracle_lib.arr_ints(res["new_leaf"][i])
        want, segs = oracle.insert_gadget_trace(low3, int(res["low_index"][i]), res["low_sib"][:, i], new3, int(res["new_index"][i]),
                                                res["new_sib"][:, i], int(res["is_largest"][i]), depth)
        assert (got[:, i] == want).all(), i
        want_segs = segs
    assert imt.insert_column_segments(depth) == want_segs
    key = "depth3_reference_sequence" if depth == 3 else "depth32_seed_0x494D54B2"
    for i, d in enumerate(GOLD["insert"][key]):
        assert hashlib.sha256(np.ascontiguousarray(got[:, i]).tobytes()).hexdigest() == d
    # halo2curves' in-memory form, item-major (siblings item-major too): what a Rust chip reads per insertion
    F = imt._ffi
    mont = lambda x: imt.to_bytes([v * R256 % P for v in oracle_lib.arr_ints(x)]).reshape(np.asarray(x).shape)
    ls_im = np.ascontiguousarray(np.asarray(res["low_sib"]).transpose(1, 0, 2))
    ns_im = np.ascontiguousarray(np.asarray(res["new_sib"]).transpose(1, 0, 2))
    got_m = ctx.insert_gadget_trace(mont(res["low_leaf"]), res["low_index"], mont(ls_im), mont(res["new_leaf"]), res["new_index"],
                                    mont(ns_im), res["is_largest"], depth, fmt=F.FMT_MONT256, item_major=True)
    assert got_m.shape == (n, got.shape[0], 32)
    assert oracle_lib.arr_ints(got_m[3]) == [v * R256 % P for v in oracle_lib.arr_ints(got[:, 3])]
    assert oracle_lib.arr_ints(got_m[n - 1]) == [v * R256 % P for v in oracle_lib.arr_ints(got[:, n - 1])]


@pytest.mark.parametrize("depth", [5, 32])
def test_non_inclusion_gadget_trace_vs_oracle(imt, ctx, oracle, depth):
    """BASELINE config 3's gadget: every new advice value of verify_non_inclusion (:127-229) outside its hashes for real
    non-members of a real tree -- the witness imt_itree_non_membership_witness returns, through
    imt_non_inclusion_gadget_trace_batch -- against oracle/gadget.c, row by row; the rows are the first rows of the
    insert_leaf call that would insert the same value; the segment table alternates with imt_path_trace_batch's blocks
    and every hash block hashes exactly the (left, right) the glue rows before it hold."""
    stored = oracle_lib.synth_values(24 if depth == 5 else 300, 0x494D54C0 + depth)
    t = imt.IndexedTree(ctx, depth, 1 << min(depth, 10))
    t.insert_batch(stored, proofs=False)
    S = set(stored)
    cand = [v + 1 for v in stored[:20] if v + 1 not in S] + [1, P - 1, max(stored) + 5, min(stored) - 1]
    n = len(cand)
    low, leaves, sib, largest = t.non_membership_witness(cand)
    K, rows = 63, 17 + 2 * 63 + 4 * depth
    got = ctx.non_inclusion_gadget_trace(leaves, low, sib, imt.to_bytes(cand), largest, depth)
    assert got.shape == (rows, n, 32) and imt.lib.imt_non_inclusion_gadget_rows(depth, 18) == rows
    want_segs = None
    for i in range(n):
        want, want_segs = oracle.non_inclusion_gadget_trace(oracle_lib.arr_ints(leaves[i]), int(low[i]), sib[:, i], cand[i],
                                                            int(largest[i]), depth)
        assert (got[:, i] == want).all(), i
    segs = imt.non_inclusion_column_segments(depth)
    assert segs == want_segs and len(segs) == 3 + 2 * depth
    assert segs == imt.insert_column_segments(depth)[:len(segs)]                 # insert_leaf begins with this gadget
    assert int(max(oracle_lib.arr_ints(got[10 + K + 2]))) == 1 == int(min(oracle_lib.arr_ints(got[10 + K + 2])))   # select: satisfied
    assert set(oracle_lib.arr_ints(got[-1])) == {1}                               # low.val < new: the last row of the gadget
    # the lookup cells: same row numbers as in the insert_leaf column, all below 2^18
    lk = ctx.insert_gadget_lookup_rows(depth)
    assert lk.max() < rows and (got[lk][:, :, 3:] == 0).all() and (got[lk][:, :, 2] < 4).all()
    # walk the column of one candidate: glue rows and the hash blocks of imt_path_trace_batch
    tr, root = ctx.path_trace(low, sib, depth, leaf3=leaves)
    assert oracle_lib.arr_ints(root) == [t.root()] * n
    out2, out3 = ctx.hash_trace_layout(2)[2], ctx.hash_trace_layout(3)[2]
    for i in (0, n - 1):
        g, h = oracle_lib.arr_ints(got[:, i]), oracle_lib.arr_ints(tr[:, i])
        cur, gi = None, 0
        for kind, arity, first, cnt in segs:
            if kind == 0:
                gi = first + cnt
            elif arity == 3:
                cur = h[first + out3]
            else:
                left, right = g[gi - 2], g[gi - 1]                                # dual_mux's outputs, the rows right before
                assert oracle.hash([left, right]) == h[first + out2]
                assert cur in (left, right)
                cur = h[first + out2]
        assert cur == t.root()
    # halo2curves' form, item-major
    F = imt._ffi
    mont = lambda x: imt.to_bytes([v * R256 % P for v in oracle_lib.arr_ints(x)]).reshape(np.asarray(x).shape)
    sib_im = np.ascontiguousarray(np.asarray(sib).transpose(1, 0, 2))
    got_m = ctx.non_inclusion_gadget_trace(mont(leaves), low, mont(sib_im), mont(imt.to_bytes(cand)), largest, depth,
                                           fmt=F.FMT_MONT256, item_major=True)
    assert got_m.shape == (n, rows, 32)
    assert oracle_lib.arr_ints(got_m[2]) == [v * R256 % P for v in oracle_lib.arr_ints(got[:, 2])]
    with pytest.raises(ValueError):
        ctx.non_inclusion_gadget_trace(leaves, low, sib, imt.to_bytes(cand), largest, 0)
    t.close()


def test_the_two_traces_cover_the_whole_advice_column_of_insert_leaf(imt, ctx, oracle):
    """insert_leaf at depth 32: walking imt_insert_column_segments, the glue rows (f3) and the hash blocks (f1) alternate,
    each hash block's inputs are what the glue rows before it say (the leaf preimage, or dual_mux's left / right), its
    output is the next path's current node, and the last block's output is the new root -- i.e. the two GPU traces
    together are every new advice value of the call, in order, with no host arithmetic in between."""
    depth = 32
    vals = oracle_lib.synth_values(3, 0x494D54B3)
    res = _real_insertions(imt, ctx, depth, vals)
    glue = ctx.insert_gadget_trace(res["low_leaf"], res["low_index"], res["low_sib"], res["new_leaf"], res["new_index"],
                                   res["new_sib"], res["is_largest"], depth)
    hashes = ctx.insert_trace(res["low_leaf"], res["low_index"], res["low_sib"], res["new_leaf"], res["new_index"], res["new_sib"], depth)
    segs = imt.insert_column_segments(depth)
    assert sum(s[3] for s in segs if s[0] == 0) == glue.shape[0] and sum(s[3] for s in segs if s[0] == 1) == hashes.shape[0]
    i = 2
    g = oracle_lib.arr_ints(glue[:, i])
    out_row = {2: 1208 - 4, 3: 1209 - 4}
    roots, prev_out, last_glue = [], None, None
    for kind, arity, first, rows in segs:
        if kind == 0:
            last_glue = g[first:first + rows]
            left, right = last_glue[-2], last_glue[-1]
            if rows == 4:                                         # a path continues: dual_mux of (previous hash, sibling)
                assert prev_out in (left, right)
            elif rows == 5:                                       # a path starts: load_witness(leaf), then dual_mux
                assert last_glue[0] in (left, right)
                assert last_glue[0] in (prev_out, oracle.hash([0, 0, 0]))      # the leaf hash just computed, or the zero leaf
        else:
            block = oracle_lib.arr_ints(hashes[first:first + rows, i])
            out = block[out_row[arity]]
            if arity == 2:
                assert oracle.hash(last_glue[-2:]) == out          # hashed exactly dual_mux's (left, right)
            prev_out = out
            roots.append(out)
    assert roots[-1] == imt.to_int(res["new_root"][i])             # :313
    assert roots[1 + depth - 1] == imt.to_int(res["old_root"][i])  # the low leaf's path ends in the old root :196-204


def test_limbs_logic_at_the_references_size(imt, ctx):
    """The reference's own test of the limb formula, test_limbs_logic (/root/reference/src/indexed_merkle_tree.rs:597-630), at
    its own size: 10^7 random pairs -- here through the GPU kernels that stand for the circuit's is_less_than: the f3 trace
    kernel's result row (imt_less_than_trace_batch: lt_out and the column's output cell) and the relation checker's range
    predicates (k_non_membership_pred through imt_non_membership_batch: low.val < new, new < next), against 256-bit integer
    comparison (numpy on four 64-bit limbs; exact Python integers on a sample of every chunk).  The reference draws
    254-bit integers; field elements have to be < p, so a draw >= p is halved.  A twentieth of the pairs share their high
    limb, a hundredth are equal, and a hundredth have a_r == b_q -- the one case where the formula AS TYPED in the
    reference's test (`are_lsb_eq = a_r == b_q`, :615) differs from the circuit's (`is_equal(a_r, b_r)`, :112): the
    kernels follow the circuit."""
    import ctypes
    import torch
    F, lib = imt._ffi, imt.lib
    total, chunk = 10_000_000, 1_000_000
    rows = int(lib.imt_less_than_trace_rows(18))
    _, _, out_row = ctx.less_than_layout(18)
    dev = torch.device("cuda", 0)
    p_limbs = np.array([(P >> (64 * k)) & (2 ** 64 - 1) for k in range(4)], dtype=np.uint64)

    def lt256(x, y):                     # [n, 4] little-endian uint64 limbs
        res = np.zeros(x.shape[0], bool)
        decided = np.zeros(x.shape[0], bool)
        for k in (3, 2, 1, 0):
            res |= ~decided & (x[:, k] < y[:, k])
            decided |= x[:, k] != y[:, k]
        return res

    def draw(rng, n):
        v = rng.integers(0, 2 ** 64, size=(n, 4), dtype=np.uint64)
        v[:, 3] >>= np.uint64(2)                                          # 254 bits
        big = ~lt256(v, np.broadcast_to(p_limbs, v.shape))                # >= p: halve (a 253-bit value is < p)
        carry = (v[big, 1:] & np.uint64(1)) << np.uint64(63)
        v[big] >>= np.uint64(1)
        v[big, :3] |= carry
        return v

    trace = torch.empty((rows, chunk, 32), dtype=torch.uint8, device=dev)
    lt = torch.empty(chunk, dtype=torch.uint8, device=dev)
    fail = torch.empty(chunk, dtype=torch.uint8, device=dev)
    leaf = torch.zeros((chunk, 3, 32), dtype=torch.uint8, device=dev)
    zeros64 = torch.zeros(chunk, dtype=torch.int64, device=dev)
    zeros8 = torch.zeros(chunk, dtype=torch.uint8, device=dev)
    root = torch.zeros(32, dtype=torch.uint8, device=dev)
    P_ = lambda t: ctypes.c_void_p(t.data_ptr())
    rng = np.random.default_rng(597)
    seen = dict(lt=0, eq=0, same_high=0, typo=0)
    for c in range(total // chunk):
        a, b = draw(rng, chunk), draw(rng, chunk)
        k = np.arange(chunk)
        s = k % 20 == 7
        b[s, 2:] = a[s, 2:]                                               # same high limb: the low limbs decide
        s = k % 100 == 13
        b[s] = a[s]
        s = k % 100 == 57
        a[s, :2] = b[s, 2:]                                               # a_r == b_q
        want = lt256(a, b)
        ab, bb = np.ascontiguousarray(a).view(np.uint8).reshape(chunk, 32), np.ascontiguousarray(b).view(np.uint8).reshape(chunk, 32)
        ta, tb = torch.from_numpy(ab).to(dev), torch.from_numpy(bb).to(dev)
        ctx._check(lib.imt_less_than_trace_batch(ctx.h, P_(ta), P_(tb), chunk, 18, P_(trace), P_(lt), F.DEVICE_PTRS))
        # the relation checker: low leaf {val = a, next_val = a}, candidate b, not the largest:
        #   IMT_F_LOW_LT_NEW clear <=> a < b;  IMT_F_RANGE_PRED clear <=> b < a
        leaf[:, 0] = ta
        leaf[:, 1] = ta
        fail.zero_()
        ctx._check(lib.imt_non_membership_batch(ctx.h, P_(root), P_(leaf), P_(zeros64), None, 0, P_(tb), P_(zeros8), chunk, P_(fail), None,
                                                F.DEVICE_PTRS))
        ctx.sync()
        got = lt.cpu().numpy().astype(bool)
        cell = trace[out_row].cpu().numpy()
        f = fail.cpu().numpy()
        assert (got == want).all(), f"chunk {c}: lt_out differs at {np.nonzero(got != want)[0][:5]}"
        assert (cell[:, 0].astype(bool) == want).all() and not cell[:, 1:].any()          # the column's output cell is the bit
        assert (((f & F.F_LOW_LT_NEW) == 0) == want).all()
        assert (((f & F.F_RANGE_PRED) == 0) == lt256(b, a)).all()
        # exact integers on a sample, boundary classes included
        for j in list(range(0, chunk, 97))[:3000] + [7, 13, 57, 107, 113, 157]:
            x, y = int.from_bytes(ab[j].tobytes(), "little"), int.from_bytes(bb[j].tobytes(), "little")
            assert x < P and y < P and bool(got[j]) == (x < y)
        seen["lt"] += int(want.sum())
        seen["eq"] += int((~want & ~lt256(b, a)).sum())
        seen["same_high"] += int((a[:, 2:] == b[:, 2:]).all(axis=1).sum())
        seen["typo"] += int((a[:, :2] == b[:, 2:]).all(axis=1).sum())
    assert seen["eq"] >= total // 100 and seen["same_high"] >= total // 20 and seen["typo"] >= total // 100
    assert 0.4 * total < seen["lt"] < 0.6 * total
