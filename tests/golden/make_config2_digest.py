"""Regenerates tests/golden/config2_oracle_digest.json: the CPU oracle (oracle/sparse.c, sequential semantics of
update_idx_leaf + rebuild, /root/reference/src/indexed_merkle_tree.rs:632-671, :715-735) run over ALL 2^16 insertions of
BASELINE config 2 (depth 32, values = oracle_lib.synth_values(2^16, 0x494D5402)), reduced to digests so that the GPU
test can check every one of the 65 536 interim / new roots, low indices and flags, not only a prefix.
Takes about three minutes of one core.   python tests/golden/make_config2_digest.py

    python tests/golden/make_config2_digest.py descending
writes config2_descending_oracle_digest.json: the SAME 2^16 values inserted in descending order (one
full-size ordered case in the GPU suite).  Every insertion's low leaf is then leaf 0 -- the sentinel is rewritten 2^16
times, each new leaf points at the one inserted before it -- the opposite extreme of the random case for everything that
replaces update_idx_leaf's scan (:639-658): one (node, time) run of 2^16 versions per level."""
import ctypes
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_lib  # noqa: E402

DEPTH, N, SEED = 32, 1 << 16, 0x494D5402
orc = oracle_lib.load()
lib = orc.lib
vals = oracle_lib.synth_values(N, SEED)
ORDER = sys.argv[1] if len(sys.argv) > 1 else "random"
assert ORDER in ("random", "descending")
if ORDER == "descending":
    vals = sorted(vals, reverse=True)
h = orc.sparse_new(DEPTH, 1 << 17)
interim = np.empty((N, 32), np.uint8)
new = np.empty((N, 32), np.uint8)
low = np.empty(N, np.uint64)
largest = np.empty(N, np.uint8)
lo, lg = ctypes.c_uint64(), ctypes.c_int()
t0 = time.time()
for i, v in enumerate(vals):
    rc = lib.orc_sparse_insert(h, oracle_lib.b32(v), ctypes.byref(lo), None, ctypes.byref(lg),
                               interim[i].ctypes.data_as(ctypes.c_void_p), new[i].ctypes.data_as(ctypes.c_void_p), None, None)
    assert rc == 0
    low[i], largest[i] = lo.value, lg.value
# the final state's proofs for a few leaves (siblings at every level)
proofs = {str(i): hashlib.sha256(orc.sparse_proof(h, DEPTH, i).tobytes()).hexdigest() for i in (0, 1, 12345, N)}
out = dict(depth=DEPTH, n=N, seed=hex(SEED), order=ORDER, provenance="oracle (derived, KAT-anchored; unpinned by the reference)",
           sha256_interim_roots=hashlib.sha256(interim.tobytes()).hexdigest(),
           sha256_new_roots=hashlib.sha256(new.tobytes()).hexdigest(),
           sha256_low_index=hashlib.sha256(low.astype("<u8").tobytes()).hexdigest(),
           sha256_is_largest=hashlib.sha256(largest.tobytes()).hexdigest(),
           final_root=str(orc.sparse_root(h)), last_new_root=str(int.from_bytes(new[-1].tobytes(), "little")),
           root_after={str(k): str(int.from_bytes(new[k - 1].tobytes(), "little")) for k in (1, 256, 4096, 32768, 65535, 65536)},
           sha256_final_proofs=proofs, oracle_seconds=round(time.time() - t0, 1))
orc.sparse_free(h)
path = os.path.join(os.path.dirname(__file__), "config2_oracle_digest.json" if ORDER == "random" else f"config2_{ORDER}_oracle_digest.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, out["oracle_seconds"], "s")
