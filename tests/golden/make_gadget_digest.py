"""Writes tests/golden/gadget_digest.json: sha256 digests of the oracle's f3 rows (oracle/gadget.c) for fixed inputs --
the is_less_than rows of a fixed set of pairs and the glue rows of the first insertions of a seeded depth-32 run and of
the reference's depth-3 sequence 30, 10, 20, 5, 50, 35 (src/indexed_merkle_tree.rs:683-690).  Provenance "oracle-gadget":
derived by the CPU oracle, KAT-anchored through its hashes, row ORDER unpinned by the reference."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402

P = oracle_lib.P
PAIRS = [(0, 0), (0, 1), (1, 0), (P - 1, P - 2), ((1 << 128) - 1, 1 << 128), ((7 << 128) + 3, (7 << 128) + 4),
         (12345678901234567890, 98765432109876543210), ((1 << 253) + 9, (1 << 253) + 9)]


def insertions(orc, depth, vals):
    h = orc.sparse_new(depth, 1 << min(depth, 10))
    out = []
    for i, v in enumerate(vals):
        o = orc.sparse_insert(h, depth, v)
        assert o["rc"] == 0
        low3 = oracle_lib.arr_ints(o["low_leaf"])
        rows, segs = orc.insert_gadget_trace(low3, o["low"], o["low_proof"], [v, low3[1], low3[2]], 1 + i, o["new_proof"],
                                             o["largest"], depth)
        out.append(rows)
    orc.sparse_free(h)
    return out


def main():
    orc = oracle_lib.load()
    res = {"provenance": "oracle-gadget", "lookup_bits": 18, "less_than": [], "insert": {}}
    for a, b in PAIRS:
        t = orc.less_than_trace(a, b, 18)
        res["less_than"].append({"a": str(a), "b": str(b), "lt": int(a < b), "sha256_rows": hashlib.sha256(t["witness"].tobytes()).hexdigest(),
                                 "sha256_cells": hashlib.sha256(t["cells"].tobytes()).hexdigest()})
    res["insert"]["depth3_reference_sequence"] = [hashlib.sha256(r.tobytes()).hexdigest() for r in insertions(orc, 3, [30, 10, 20, 5, 50, 35])]
    res["insert"]["depth32_seed_0x494D54B2"] = [hashlib.sha256(r.tobytes()).hexdigest()
                                                for r in insertions(orc, 32, oracle_lib.synth_values(4, 0x494D54B2))]
    with open(os.path.join(ROOT, "tests", "golden", "gadget_digest.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("wrote gadget_digest.json:", len(res["less_than"]), "pairs,", {k: len(v) for k, v in res["insert"].items()})


if __name__ == "__main__":
    main()
