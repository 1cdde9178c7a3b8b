"""CPU: f1, the witness trace of halo2-base's PoseidonHasher::hash_fix_len_array
(/root/reference/src/indexed_merkle_tree.rs:92,194,271-275,299-303 call it; the gadget is un-vendored).

Three restatements are compared with each other and with what little the reference pins:
  oracle/trace.c             C, derives the optimised spec from the plain Grain constants itself, emits the whole column
  csrc/imt_trace_device.hpp  the product's device code (host build, tests/native), emits the new values only
  csrc/imt_trace_layout.cpp  the product's cell map
Pins: the output row is the hash (reference KAT for [0,0,0]); every vertical gate of every column holds."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib
from oracle_lib import P, KAT_ZERO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "vectors.json")))
R256 = (1 << 256) % P
R261 = (1 << 261) % P
CASES = [[0, 0, 0], [1, 2], [1, 2, 3], [P - 1, P - 2], [P - 1, 0, P - 1], [1 << 64, 0]]


def _ints(a):
    return [int.from_bytes(x.tobytes(), "little") for x in np.asarray(a, np.uint8).reshape(-1, 32)]


def test_oracle_trace_output_is_the_hash_and_every_gate_holds(oracle):
    cases = CASES + [oracle_lib.synth_values(3, 77)[:k] for k in (2, 3)]
    for xs in cases:
        t = oracle.hash_trace(xs)
        w = _ints(t["witness"])
        assert len(w) == (1208 if len(xs) == 2 else 1209) and len(t["cells"]) == (4506 if len(xs) == 2 else 4509)
        assert t["out_row"] == len(w) - 4
        assert w[t["out_row"]] == oracle.hash(xs)                       # = the plain 65-round sponge
        col = _ints(t["cells"])
        gates = 0
        for i in np.nonzero(t["gate"])[0]:
            assert (col[i] + col[i + 1] * col[i + 2] - col[i + 3]) % P == 0, (xs, i)
            gates += 1
        assert gates == len(w)                                           # every new value is the d of exactly one gate
        # witness cells appear in order and carry the trace rows; copies repeat an EARLIER row
        wi = [int(j) for k, j in zip(t["kind"], t["index"]) if k == 3]
        assert wi == list(range(len(w)))
        seen = -1
        for k, j, v in zip(t["kind"], t["index"], col):
            if k == 3:
                seen = int(j)
            if k == 4:
                assert int(j) <= seen and v == w[int(j)]
    assert _ints(oracle.hash_trace([0, 0, 0])["witness"])[1205] == KAT_ZERO      # reference :247-250


def test_trace_golden_digests(oracle):
    for g in GOLD["hash_trace"]:
        xs = [int(x) for x in g["in"]]
        t = oracle.hash_trace(xs)
        assert hashlib.sha256(t["witness"].tobytes()).hexdigest() == g["sha256_rows"]
        assert hashlib.sha256(t["cells"].tobytes()).hexdigest() == g["sha256_cells"]
        w = _ints(t["witness"])
        for r, v in g["rows"].items():
            assert w[int(r)] == int(v)


def _emul_trace(emul, xs, fmt_out):
    rows = 1208 if len(xs) == 2 else 1209
    out = np.zeros((rows + 2, 32), np.uint8)
    n = emul.emul_hash_trace(b"".join(oracle_lib.b32(x) for x in xs), len(xs), out.ctypes.data_as(ctypes.c_void_p), 0, fmt_out)
    assert n == rows
    return out[:rows]


def test_product_trace_code_equals_oracle_in_every_format(oracle, emul):
    """imt_trace_device.hpp (host build of the same source the kernel compiles) against oracle/trace.c.  Format 1 is
    what a Rust chip transmutes into Fr: x * 2^256 mod p; format 2 the library's own x * 2^261 mod p."""
    for xs in CASES + [oracle_lib.synth_values(3, 78)[:k] for k in (2, 3)] + [[int(x) for x in g["in"]] for g in GOLD["hash_trace"]]:
        want = _ints(oracle.hash_trace(xs)["witness"])
        assert _ints(_emul_trace(emul, xs, 0)) == want
        assert _ints(_emul_trace(emul, xs, 1)) == [(v * R256) % P for v in want]
        assert _ints(_emul_trace(emul, xs, 2)) == [(v * R261) % P for v in want]


@pytest.mark.parametrize("arity", [2, 3])
def test_product_cell_layout_equals_oracle_and_rebuilds_a_satisfied_column(oracle, emul, imt, arity):
    emul.emul_trace_layout.restype = ctypes.c_int
    cells, consts, out_row = imt.trace_layout(lambda *a: emul.emul_trace_layout(*a), arity, 0,
                                              lambda rc: (_ for _ in ()).throw(AssertionError(rc)) if rc else None)
    xs = oracle_lib.synth_values(3, 79)[:arity]
    t = oracle.hash_trace(xs)
    assert out_row == t["out_row"] and len(cells) == len(t["cells"])
    assert (cells["gate"] == t["gate"]).all() and (cells["kind"] == t["kind"]).all()
    assert (cells["region"] == t["region"]).all()
    # a chip assigns region by region: a copy inside a region never refers to a row of the same region
    starts = list(np.nonzero(cells["region"])[0]) + [len(cells)]
    assert starts[0] == 0 and set(np.diff(starts)) <= {4, 7, 10}          # add / mul / mul_add, sum, inner_product
    for a, b in zip(starts, starts[1:]):
        own = {int(c["index"]) for c in cells[a:b] if c["kind"] == imt._ffi.CELL_WITNESS}
        assert not own & {int(c["index"]) for c in cells[a:b] if c["kind"] == imt._ffi.CELL_COPY}
    nonconst = cells["kind"] != imt._ffi.CELL_CONST
    assert (cells["index"][nonconst] == t["index"][nonconst]).all()
    # the column a chip would assign from the product's own trace + layout + constants equals the oracle's, cell by cell
    col = imt.rebuild_advice_column(cells, consts, xs, _emul_trace(emul, xs, 0))
    assert col == _ints(t["cells"])
    assert imt.check_vertical_gates(cells, col) == (1208 if arity == 2 else 1209)
    assert col[[i for i, c in enumerate(cells) if c["kind"] == 3 and c["index"] == out_row][0]] == oracle.hash(xs)
    # the constants in Rust's in-memory form
    _, consts_m, _ = imt.trace_layout(lambda *a: emul.emul_trace_layout(*a), arity, 1, lambda rc: None)
    assert _ints(consts_m) == [(v * R256) % P for v in _ints(consts)]
    assert len(set(_ints(consts))) == len(consts) < 420          # de-duplicated table


def test_store_mont256_division_by_32_edge_cases(emul):
    """store_mont256 (imt_trace_device.hpp): a value a < 4p in nine 29-bit limbs leaves as the canonical a / 32 mod p,
    as a + ms p with the signed multiplier ms = m or m - 32, never normalised.  Which one is decided by the top limbs
    except when they are equal -- about once in 2^25 rows, so never in a random test: crafted here, together with
    the values right at the boundaries a = (32 - m) p for m = 29, 30, 31."""
    import ctypes
    import random
    inv32 = pow(32, -1, P)

    def run(a):
        limbs = (ctypes.c_uint32 * 9)(*[(a >> (29 * i)) & ((1 << 29) - 1) if i < 8 else a >> 232 for i in range(9)])
        out = (ctypes.c_uint8 * 32)()
        emul.emul_store_mont256(limbs, out)
        return int.from_bytes(bytes(out), "little")

    rng = random.Random(5)
    cases = []
    for k in (1, 2, 3):                     # a = k p (+- multiples of 32 keep m): the flip point of the subtraction
        for d in (-64, -32, 0, 32, 64, 32 * 12345):
            cases.append(k * P + d)
        top = (k * P) >> 232                # same top limb as k p, anything below: the slow path on both sides
        for _ in range(200):
            low = rng.randrange(1 << 232)
            a = (top << 232) | low
            cases.append(a - ((a - k * P) % 32))          # congruent to k p mod 32 -> m = 32 - k
    cases += [0, 1, 31, 32, P - 1, P, P + 1, 4 * P - 1, 2 * P + 12345]
    cases += [rng.randrange(4 * P) for _ in range(20000)]
    for a in cases:
        assert 0 <= a < 4 * P
        got = run(a)
        assert got == a * inv32 % P, hex(a)
