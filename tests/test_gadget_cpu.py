"""CPU: f3, the advice cells of insert_leaf OUTSIDE hash_fix_len_array (oracle/gadget.c) -- the oracle against itself
and against an independent big-integer model (tests/oracle_lib.py: less_than_rows_model, insert_gadget_rows_model):

 * is_less_than (/root/reference/src/indexed_merkle_tree.rs:98-125): every vertical gate a + b*c = d of the emitted
   column holds, the new witnesses equal the model's, the result equals the reference's boolean limb formula
   (oracle/indexed.c, orc_is_less_than_limbs) and the integer comparison; edge cases (equal values, equal high limbs,
   0, p - 1, values straddling 2^128), several lookup_bits (the reference's tests use 18, :436);
 * insert_leaf's glue rows (:231-314): the oracle's column walk against the model for real insertions (depth 3 and
   32), the segment table (glue rows / hash blocks) adds up, the row count is 20 + 2 K + 16 depth.
ORDER UNPINNED BY THE REFERENCE (halo2-base is un-vendored), exactly like the f1 trace."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from oracle_lib import P  # noqa: E402

EDGE = [(0, 0), (0, 1), (1, 0), (5, 5), (P - 1, P - 1), (P - 1, 0), (0, P - 1), ((1 << 128) - 1, 1 << 128),
        (1 << 128, (1 << 128) - 1), ((7 << 128) + 3, (7 << 128) + 4), ((7 << 128) + 4, (7 << 128) + 3),
        ((1 << 253) + 9, (1 << 253) + 9), ((3 << 128), (3 << 128) + 1), (P - 2, P - 1)]


def column_of(tr):
    """the advice column as ints; CONST cells carry their value in `cells` already"""
    return [int.from_bytes(c.tobytes(), "little") for c in tr["cells"]]


@pytest.mark.parametrize("lookup_bits", [18, 8, 17, 28])
def test_less_than_column_gates_model_and_result(lookup_bits):
    orc = oracle_lib.load()
    rng = random.Random(1800 + lookup_bits)
    pairs = EDGE + [(rng.randrange(P), rng.randrange(P)) for _ in range(40)]
    pairs += [((q << 128) + rng.randrange(1 << 128), (q << 128) + rng.randrange(1 << 128)) for q in (0, 1, rng.randrange(1 << 120))]
    L = -(-128 // lookup_bits) + 1
    for a, b in pairs:
        tr = orc.less_than_trace(a, b, lookup_bits)
        col = column_of(tr)
        gates = 0
        for i in np.nonzero(tr["gate"])[0]:
            x, y, z, w = col[i:i + 4]
            assert (x + y * z - w) % P == 0, (a, b, int(i))
            gates += 1
        assert gates == 2 * (2 + (L - 1) + 2) + 2 * 3 + 4 + 4 + 2         # range.is_less_than x2, is_equal x2, not x4, and x4, or
        want, out = oracle_lib.less_than_rows_model(a, b, lookup_bits)
        got = oracle_lib.arr_ints(tr["witness"])
        assert got == want, (a, b)
        assert len(got) == 4 * L + 27
        assert got[tr["out_row"]] == out == (1 if a < b else 0)
        assert orc.is_less_than_limbs(a, b) == out                        # the reference's formula, :98-125
        # witness cells appear in row order; copies point backwards; inputs are a_q, a_r, b_q, b_r
        rows = [int(ix) for k, ix in zip(tr["kind"], tr["index"]) if k == 3]
        assert rows == list(range(len(got)))
        ins = {int(ix): col[j] for j, (k, ix) in enumerate(zip(tr["kind"], tr["index"])) if k == 1}
        assert ins == {0: a >> 128, 1: a & ((1 << 128) - 1), 2: b >> 128, 3: b & ((1 << 128) - 1)}
        seen = -1
        for j, (k, ix) in enumerate(zip(tr["kind"], tr["index"])):
            if k == 3:
                seen = int(ix)
            elif k == 4:
                assert int(ix) <= seen and col[j] == got[int(ix)]


def _insertions(depth, n, seed):
    orc = oracle_lib.load()
    h = orc.sparse_new(depth, 1 << min(depth, 10))
    vals = oracle_lib.synth_values(n, seed)
    rows = []
    for i, v in enumerate(vals):
        o = orc.sparse_insert(h, depth, v)
        assert o["rc"] == 0
        rows.append((v, 1 + i, o))
    orc.sparse_free(h)
    return orc, rows


@pytest.mark.parametrize("depth,n", [(3, 6), (32, 5)])
def test_insert_gadget_rows_against_the_model(imt, depth, n):
    orc, ins = _insertions(depth, n, 0x494D54A0 + depth)
    K = 4 * 9 + 27
    lk = imt.Context.insert_gadget_lookup_rows(depth, 18)
    assert len(lk) == 4 * 9 and len(set(lk.tolist())) == len(lk)
    for v, new_index, o in ins:
        low3 = oracle_lib.arr_ints(o["low_leaf"])
        new3 = [v, low3[1], low3[2]]
        got, segs = orc.insert_gadget_trace(low3, o["low"], o["low_proof"], new3, new_index, o["new_proof"], o["largest"], depth)
        want = oracle_lib.insert_gadget_rows_model(orc, low3, o["low"], o["low_proof"], new3, new_index, o["new_proof"],
                                                   o["largest"], depth)
        assert oracle_lib.arr_ints(got) == want
        assert len(want) == 20 + 2 * K + 16 * depth
        # the select row: is_largest ? next_val == 0 : new < next_val -- 1 for a real insertion (:182-191)
        assert want[10 + K + 2] == 1
        # the lookup rows are the 18-bit limbs of the four shifted differences of the two comparisons (:180, :226)
        M128 = (1 << 128) - 1
        pairs = [(v >> 128, low3[1] >> 128), (v & M128, low3[1] & M128), (low3[0] >> 128, v >> 128), (low3[0] & M128, v & M128)]
        for g, (x, y) in enumerate(pairs):
            limbs = [want[r] for r in lk[9 * g:9 * g + 9]]
            assert all(t < (1 << 18) for t in limbs)
            assert sum(t << (18 * i) for i, t in enumerate(limbs)) == (1 << 144) + x - y
        # segments: glue rows and hash blocks alternate, cover both traces exactly, hash blocks in imt_insert_trace_batch's order
        glue = [s for s in segs if s[0] == 0]
        hsh = [s for s in segs if s[0] == 1]
        assert sum(s[3] for s in glue) == len(want) and [s[2] for s in glue] == list(np.cumsum([0] + [s[3] for s in glue[:-1]]))
        assert [s[1] for s in hsh] == ([3] + [2] * depth) * 2 + [2] * depth + [3] + [2] * depth
        assert [s[2] for s in hsh] == list(np.cumsum([0] + [s[3] for s in hsh[:-1]]))
        assert sum(s[3] for s in hsh) == 3 * 1209 + 4 * depth * 1208
        assert segs[0][0] == 0 and segs[0][3] == 10 + K + 3 and segs[1] == (1, 3, 0, 1209)
        assert all(x[0] != y[0] for x, y in zip(segs, segs[1:]) if x[0] == 0)           # never two glue segments in a row


@pytest.mark.parametrize("depth,n", [(3, 6), (32, 3)])
def test_non_inclusion_gadget_is_the_head_of_the_insert_gadget(imt, depth, n):
    """verify_non_inclusion (:127-229) on its own = the first 17 + 2 K + 4 depth glue rows and the first 3 + 2 depth
    segments of the insert_leaf that inserts the same value (insert_leaf begins with it, :253-257) -- oracle against
    oracle, and the product's segment table and row count (host arithmetic, no GPU) against both"""
    orc, ins = _insertions(depth, n, 0x494D54A8 + depth)
    K = 4 * 9 + 27
    for v, new_index, o in ins:
        low3 = oracle_lib.arr_ints(o["low_leaf"])
        full, fsegs = orc.insert_gadget_trace(low3, o["low"], o["low_proof"], [v, low3[1], low3[2]], new_index, o["new_proof"],
                                              o["largest"], depth)
        head, hsegs = orc.non_inclusion_gadget_trace(low3, o["low"], o["low_proof"], v, o["largest"], depth)
        assert len(head) == 17 + 2 * K + 4 * depth == imt.lib.imt_non_inclusion_gadget_rows(depth, 18)
        assert (head == full[:len(head)]).all()
        assert hsegs == fsegs[:len(hsegs)] and len(hsegs) == 3 + 2 * depth
        assert imt.non_inclusion_column_segments(depth) == hsegs
        assert sum(s[3] for s in hsegs if s[0] == 1) == 1209 + depth * 1208
    assert imt.lib.imt_non_inclusion_gadget_rows(0, 18) == 0 and imt.lib.imt_non_inclusion_gadget_rows(32, 29) == 0


def test_less_than_rejects_bad_arguments():
    orc = oracle_lib.load()
    import ctypes
    nc = ctypes.c_size_t()
    assert orc.lib.orc_less_than_trace(oracle_lib.b32(1), oracle_lib.b32(2), ctypes.c_uint(0), None, None, ctypes.c_size_t(0),
                                       ctypes.byref(nc), None, ctypes.c_size_t(0), None, None) != 0
    assert orc.lib.orc_less_than_trace((P).to_bytes(32, "little"), oracle_lib.b32(2), ctypes.c_uint(18), None, None,
                                       ctypes.c_size_t(0), ctypes.byref(nc), None, ctypes.c_size_t(0), None, None) != 0


@pytest.mark.parametrize("lookup_bits", [18, 17, 8, 4, 28])
def test_lookup_rows_are_the_limbs_of_both_range_checks(imt, lookup_bits):
    """imt_less_than_lookup_rows (closed form, no GPU) = the cells oracle/gadget.c's RangeChip restatement adds to the
    lookup; on real columns those rows hold values below 2^lookup_bits that recompose both shifted differences --
    exactly what range_check(shifted, padded + lookup_bits) constrains (halo2-lib v0.4.x gates/range.rs)."""
    orc = oracle_lib.load()
    L = -(-128 // lookup_bits) + 1
    rows = imt.Context.less_than_lookup_rows(lookup_bits)
    assert rows.tolist() == orc.less_than_lookup_rows(lookup_bits).tolist() and len(rows) == 2 * L
    assert len(set(rows.tolist())) == 2 * L and list(rows) == sorted(rows)
    with pytest.raises(ValueError):
        imt.Context.less_than_lookup_rows(0)
    for a, b in EDGE[:6] + list(zip(oracle_lib.synth_values(4, 93), oracle_lib.synth_values(4, 94))):
        w = oracle_lib.arr_ints(orc.less_than_trace(a, b, lookup_bits)["witness"])
        for h, (x, y) in enumerate(((a >> 128, b >> 128), (a & ((1 << 128) - 1), b & ((1 << 128) - 1)))):
            limbs = [w[r] for r in rows[h * L:(h + 1) * L]]
            assert all(v < (1 << lookup_bits) for v in limbs)
            shifted = (1 << ((L - 1) * lookup_bits)) + x - y
            assert sum(v << (lookup_bits * i) for i, v in enumerate(limbs)) == shifted == w[rows[h * L] - 2]
    # no other row of the column needs the table: every remaining witness is tied down by a vertical gate alone
    assert imt.lib.imt_less_than_trace_rows(lookup_bits) == 4 * L + 27


@pytest.mark.parametrize("lookup_bits", [18, 8, 28])
def test_product_less_than_layout_equals_oracle_and_rebuilds_a_satisfied_column(emul, imt, lookup_bits):
    """csrc/imt_gadget_layout.cpp (host code, no GPU needed) against oracle/gadget.c: same cells, kinds, gates, regions
    and indices; the column a chip would assign from the layout + its constants + the four limb inputs + the rows equals
    the oracle's column cell by cell and satisfies every gate"""
    import ctypes
    orc = oracle_lib.load()
    emul.emul_less_than_layout.restype = ctypes.c_int
    cells, consts, out_row = imt.trace_layout(lambda *a: emul.emul_less_than_layout(*a), lookup_bits, 0,
                                              lambda rc: (_ for _ in ()).throw(AssertionError(rc)) if rc else None)
    for a, b in EDGE[:8] + [(v, w) for v, w in zip(oracle_lib.synth_values(6, 91), oracle_lib.synth_values(6, 92))]:
        t = orc.less_than_trace(a, b, lookup_bits)
        assert out_row == t["out_row"] and len(cells) == len(t["cells"])
        assert (cells["gate"] == t["gate"]).all() and (cells["kind"] == t["kind"]).all() and (cells["region"] == t["region"]).all()
        nonconst = cells["kind"] != imt._ffi.CELL_CONST
        assert (cells["index"][nonconst] == t["index"][nonconst]).all()
        ins = [a >> 128, a & ((1 << 128) - 1), b >> 128, b & ((1 << 128) - 1)]
        col = imt.rebuild_advice_column(cells, consts, ins, t["witness"])
        assert col == column_of(t)
        assert imt.check_vertical_gates(cells, col) == int(t["gate"].sum())
    # a chip assigns region by region: a copy inside a region never refers to a row of the same region -- EXCEPT where
    # the gadget itself ties two cells of one region together (is_zero's and or's repeated witness are new cells, not copies)
    starts = list(np.nonzero(cells["region"])[0]) + [len(cells)]
    for x, y in zip(starts, starts[1:]):
        own = {int(c["index"]) for c in cells[x:y] if c["kind"] == imt._ffi.CELL_WITNESS}
        assert not own & {int(c["index"]) for c in cells[x:y] if c["kind"] == imt._ffi.CELL_COPY}
    R256 = (1 << 256) % P
    _, consts_m, _ = imt.trace_layout(lambda *a: emul.emul_less_than_layout(*a), lookup_bits, 1, lambda rc: None)
    assert oracle_lib.arr_ints(consts_m) == [(v * R256) % P for v in oracle_lib.arr_ints(consts)]


def test_gadget_golden_digests():
    """tests/golden/gadget_digest.json (make_gadget_digest.py): the oracle reproduces its committed digests"""
    import hashlib
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_gadget_digest as mk
    orc = oracle_lib.load()
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "gadget_digest.json")))
    for g in gold["less_than"]:
        t = orc.less_than_trace(int(g["a"]), int(g["b"]), gold["lookup_bits"])
        assert hashlib.sha256(t["witness"].tobytes()).hexdigest() == g["sha256_rows"]
        assert hashlib.sha256(t["cells"].tobytes()).hexdigest() == g["sha256_cells"]
    assert [hashlib.sha256(r.tobytes()).hexdigest() for r in mk.insertions(orc, 3, [30, 10, 20, 5, 50, 35])] == \
        gold["insert"]["depth3_reference_sequence"]
    assert [hashlib.sha256(r.tobytes()).hexdigest() for r in mk.insertions(orc, 32, oracle_lib.synth_values(4, 0x494D54B2))] == \
        gold["insert"]["depth32_seed_0x494D54B2"]
