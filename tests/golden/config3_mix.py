"""BASELINE config 3 at its full size as a MIXED batch: 2^20 verify_non_inclusion items against the config-2 tree
(depth 32, 2^16 insertions of oracle_lib.synth_values(2^16, 0x494D5402)), most of them honest, the rest broken in the ways
the reference's constraints exist to catch (/root/reference/src/indexed_merkle_tree.rs:127-229).  Shared by
tests/golden/make_config3_digest.py (the CPU oracle's masks and roots, committed as digests) and
tests/test_gpu_parity.py::test_config3_mixed_batch_2pow20_against_the_oracle_digest (the GPU's): both build the HONEST
witnesses from their own tree and then call mix(), which only rearranges bytes."""
import numpy as np

DEPTH, N_TREE, TREE_SEED, N_ITEMS, CAND_SEED = 32, 1 << 16, 0x494D5402, 1 << 20, 3
CLASSES = {1: "the low leaf's own value as the candidate (a member shown with its own leaf)",
           2: "the low leaf's next value as the candidate (a member shown with its predecessor)",
           3: "candidate 0",
           4: "is_largest flipped",
           5: "one sibling replaced (level = item mod 32)",
           6: "low index with its lowest bit flipped (path position)",
           7: "next_idx of the low leaf's preimage changed (leaf hash)"}


def candidates():
    rng = np.random.default_rng(CAND_SEED)
    cand = rng.integers(0, 256, size=(N_ITEMS, 32), dtype=np.uint8)
    cand[:, 31] &= 0x0f                                    # < 2^252 < p, non-zero w.h.p., not members w.h.p.
    return cand


def mix(cand, low, leaves, sib, largest):
    """in place: item i with i mod 64 in CLASSES is broken as that class says; returns the class of every item (0 = honest).
    cand [n, 32], low [n] uint64, leaves [n, 3, 32], sib [depth, n, 32], largest [n] uint8"""
    n = cand.shape[0]
    cls = np.zeros(n, np.uint8)
    i = np.arange(n)
    for c in CLASSES:
        cls[i % 64 == c] = c
    s = cls == 1
    cand[s] = leaves[s, 0]
    s = cls == 2
    cand[s] = leaves[s, 1]
    s = cls == 3
    cand[s] = 0
    s = cls == 4
    largest[s] ^= 1
    for j in np.nonzero(cls == 5)[0]:
        sib[j % DEPTH, j] = cand[j]
    s = cls == 6
    low[s] ^= np.uint64(1)
    s = cls == 7
    leaves[s, 2, 0] ^= 1
    return cls
