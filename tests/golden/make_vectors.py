"""Regenerates tests/golden/vectors.json.

The reference (Rust, un-vendored git-branch dependencies, no cargo/rustc in the image)
cannot run here, so the fixture has three provenance classes, recorded per entry:
  "reference"      the zero-leaf hash literal at /root/reference/src/indexed_merkle_tree.rs:248
  "survey-script"  values SURVEY.md sec. B lists, computed there by an independent big-int script
  "oracle"         values produced by oracle/ (KAT-anchored, unpinned by the reference itself)
  "public-circomlib"  the widely published known answers of circomlib's Poseidon for two inputs,
                   poseidon([1,2]) and poseidon([0,0]): lane 0 of THIS permutation (same Grain
                   constants and MDS, t=3, R_F=8, R_P=57) applied to [0,a,b].  They pin the
                   permutation independently of the reference's own known answer, which in turn pins
                   the sponge around it (capacity 2^64, padding, output lane 1).
  "oracle-trace"   f1: digests and checkpoints of the witness trace of hash_fix_len_array as oracle/trace.c
                   restates halo2-base's gadget -- UNPINNED by the reference (its output row is pinned: it is the
                   hash); kept so that a change of the trace order or of the spec derivation is noticed
Run from the repo root:  python tests/golden/make_vectors.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_lib  # noqa: E402

O = oracle_lib.load()
P = oracle_lib.P
vec = {"modulus": str(P), "entries": []}


def add(kind, inputs, out, prov):
    vec["entries"].append({"kind": kind, "in": [str(x) for x in inputs], "out": str(out), "provenance": prov})


add("hash3", [0, 0, 0], oracle_lib.KAT_ZERO, "reference")
add("permute_lane0", [0, 1, 2], 7853200120776062878684798364095072458815029376092732009249414926327459813530,
    "public-circomlib")      # 0x115cc0f5e7d690413df64c6b9662e9cf2a3617f2743245519e19607a4417189a
add("permute_lane0", [0, 0, 0], 14744269619966411208579211824598458697587494354926760081771325075741142829156,
    "public-circomlib")      # 0x2098f5fb9e239eab3ceac3f27b81e481dc3124d55ffed523a839ee8446b64864
z = oracle_lib.KAT_ZERO
add("hash2", [z, z], 4631070890700890603680124140378602853676871767725085684580160391942752072668, "survey-script")
add("hash2", [1, 2], 21877010470986031768387685515622483058891036836834541740519926154448980606803, "survey-script")
add("hash3", [1, 2, 3], 13779850769162876186950433148717826617852336339119390266180431467389789105393, "survey-script")
for d, v in ((3, 11221770372622818334043267142716753777637498248460122103656425965372530197968),
             (8, 1727427431492614990563994011315675929958651134447243072823769748003472186163),
             (32, 5762754648593443595065451807490613754617800564295275528461006591588932157774)):
    add("empty_root", [d], v, "survey-script")
rounds = [(30, 0, 1, 19890339583349038801538071266233213153819311616889133208835138342744190216647,
           12751690945445451579176064519072020512765665772250977239717014295594884463011),
          (10, 0, 0, 6875051297863511546087049228527067390264001432937566062295159985852886357893,
           2448913255876797507841652245687653406814049511746920551032110906066661031685),
          (20, 2, 0, 8162022922075942844691504419762784016391357891178871619214902034003587793086,
           14555469528445384252702120751234523748523903391985378473704887708953966730192),
          (5, 0, 0, 6629459660764510586913266416405043502041952990633563426284789774283114998010,
           8076804783578882893962767795276902497200865872284723950653180282503273674732),
          (50, 1, 1, 16173909447417337533912277683871324895377949992061308900710298367322227991906,
           20419076845670931973612403412227258777823597395802461730854189096430290082514),
          (35, 1, 0, 258121065344524826517132414517089181031476928898407825009423481804777726013,
           14929561429163008590837870191858813375294709329236271969877117441260023133805)]
vec["multi_round_depth3"] = [dict(val=str(a), low_idx=b, largest=c, interim_root=str(d), new_root=str(e),
                                  provenance="survey-script") for a, b, c, d, e in rounds]

# oracle-derived: edge and seeded random hashes, permutations, a depth-32 insertion run
edge = [0, 1, 2, P - 1, P - 2, 1 << 64, (1 << 128) - 1, (1 << 253), P >> 1]
for a in edge:
    for b in (0, 1, P - 1):
        add("hash2", [a, b], O.hash([a, b]), "oracle")
        add("hash3", [a, b, a], O.hash([a, b, a]), "oracle")
vals = oracle_lib.synth_values(48, 0x494D5400)
for i in range(0, 24, 2):
    add("hash2", vals[i:i + 2], O.hash(vals[i:i + 2]), "oracle")
for i in range(0, 24, 3):
    add("hash3", vals[i:i + 3], O.hash(vals[i:i + 3]), "oracle")
for i in range(24, 36, 3):
    add("permute", vals[i:i + 3], 0, "oracle")
    vec["entries"][-1]["out"] = [str(x) for x in O.permute(vals[i:i + 3])]
# f1: witness traces (sha256 over the rows as canonical 32-byte little-endian, plus a few rows verbatim)
import hashlib  # noqa: E402
vec["hash_trace"] = []
for xs in ([0, 0, 0], [1, 2], [1, 2, 3], vals[36:38], vals[38:41]):
    t = O.hash_trace(xs)
    w = t["witness"]
    rows = [0, 1, 4, 5, 23, 604, 605, 608, len(w) - 9, t["out_row"], len(w) - 1]
    vec["hash_trace"].append(dict(**{"in": [str(x) for x in xs]}, n_cells=int(len(t["cells"])), n_rows=int(len(w)),
                                  out_row=int(t["out_row"]), sha256_rows=hashlib.sha256(w.tobytes()).hexdigest(),
                                  sha256_cells=hashlib.sha256(t["cells"].tobytes()).hexdigest(),
                                  rows={str(r): str(int.from_bytes(w[r].tobytes(), "little")) for r in rows},
                                  provenance="oracle-trace"))
h = O.sparse_new(32, 64)
run = []
big = []
R256 = 1 << 256
for i, v in enumerate(oracle_lib.synth_values(40, 0x494D5402)):
    r = O.sparse_insert(h, 32, v)
    assert r["rc"] == 0
    run.append(dict(val=str(v), low_idx=r["low"], largest=r["largest"], interim_root=str(r["interim_root"]),
                    new_root=str(r["new_root"])))
    if i >= 38:
        # BASELINE config 5 at the size a circuit assigns: the whole witness trace of this insert_leaf call at depth
        # 32 (3 + 128 hashes, 158 251 rows), and of its verify_non_inclusion part alone (the low leaf's hash + path)
        low3 = oracle_lib.arr_ints(r["low_leaf"])
        new3 = oracle_lib.arr_ints(O.sparse_preimage(h, i + 1))
        rows, roots = oracle_lib.insert_leaf_trace(O, low3, r["low"], r["low_proof"], new3, i + 1, r["new_proof"], 32)
        assert roots[1] == roots[2] == r["interim_root"] and roots[3] == r["new_root"]
        assert roots[0] == (int(run[-2]["new_root"]) if i else 0)
        mont = oracle_lib.ints_to_arr([x * R256 % P for x in oracle_lib.arr_ints(rows)])
        nm = 1209 + 32 * 1208
        big.append(dict(insertion=i, n_rows=int(rows.shape[0]), sha256_rows=hashlib.sha256(rows.tobytes()).hexdigest(),
                        sha256_rows_mont256=hashlib.sha256(mont.tobytes()).hexdigest(),
                        non_inclusion_rows=nm, sha256_non_inclusion_rows=hashlib.sha256(rows[:nm].tobytes()).hexdigest(),
                        provenance="oracle-trace"))
O.sparse_free(h)
vec["insert_run_depth32"] = dict(provenance="oracle", seed="0x494D5402", rounds=run)
vec["insert_trace_depth32"] = big

out = os.path.join(os.path.dirname(__file__), "vectors.json")
with open(out, "w") as f:
    json.dump(vec, f, indent=1)
print("wrote", out, len(vec["entries"]), "entries")
