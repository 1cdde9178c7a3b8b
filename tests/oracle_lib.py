"""ctypes wrapper of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY:
the checker the HIP path is compared against, never the thing shipped or measured."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
KAT_ZERO = 1960587138944869480785025106734196872454309951825657414575195034687326603497  # reference :248


def b32(x):
    return int(x).to_bytes(32, "little")


def ints_to_arr(xs):
    return np.frombuffer(b"".join(b32(x) for x in xs), dtype=np.uint8).reshape(-1, 32).copy()


class InsertTrace(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint8 * 32) for n in ("low_leaf_hash", "root_from_low", "new_low_leaf_hash",
                                                   "interim_root", "zero_slot_root", "new_leaf_hash", "new_root")]


class TraceCell(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint8), ("gate", ctypes.c_uint8), ("region", ctypes.c_uint16), ("index", ctypes.c_uint32)]


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.orc_poseidon_init()
        lib.orc_sparse_size.restype = ctypes.c_uint64
        lib.orc_tree_num_levels.restype = ctypes.c_size_t

    # ---- hashes ----
    def hash(self, xs):
        out = ctypes.create_string_buffer(32)
        rc = self.lib.orc_hash_var(out, b"".join(b32(x) for x in xs), ctypes.c_size_t(len(xs)))
        assert rc == 0, rc
        return int.from_bytes(out.raw, "little")

    def hash2_batch(self, arr):  # uint8 [n,2,32] -> [n,32]
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        n = arr.shape[0]
        out = np.empty((n, 32), dtype=np.uint8)
        rc = self.lib.orc_hash2_batch(out.ctypes.data_as(ctypes.c_void_p), arr.ctypes.data_as(ctypes.c_void_p),
                                      ctypes.c_size_t(n))
        assert rc == 0, rc
        return out

    def hash3_batch(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        n = arr.shape[0]
        out = np.empty((n, 32), dtype=np.uint8)
        rc = self.lib.orc_hash3_batch(out.ctypes.data_as(ctypes.c_void_p), arr.ctypes.data_as(ctypes.c_void_p),
                                      ctypes.c_size_t(n))
        assert rc == 0, rc
        return out

    def permute(self, state3):
        buf = ctypes.create_string_buffer(b"".join(b32(x) for x in state3), 96)
        self.lib.orc_permute_bytes(buf)
        return [int.from_bytes(buf.raw[32 * i:32 * i + 32], "little") for i in range(3)]

    def zero_hashes(self, depth):
        out = np.empty((depth + 1, 32), dtype=np.uint8)
        self.lib.orc_zero_hashes(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint(depth))
        return out

    # ---- f1: cell-by-cell witness trace of hash_fix_len_array (oracle/trace.c) ----
    def hash_trace(self, xs):
        """dict(cells uint8 [nc,32], kind/gate/index uint arrays [nc], witness uint8 [nw,32], out_row)"""
        cap, wcap = 6000, 2000
        cells = np.empty((cap, 32), np.uint8)
        desc = (TraceCell * cap)()
        wit = np.empty((wcap, 32), np.uint8)
        nc, nw, row = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_uint32()
        rc = self.lib.orc_hash_trace(b"".join(b32(x) for x in xs), len(xs), cells.ctypes.data_as(ctypes.c_void_p), desc,
                                     ctypes.c_size_t(cap), ctypes.byref(nc), wit.ctypes.data_as(ctypes.c_void_p),
                                     ctypes.c_size_t(wcap), ctypes.byref(nw), ctypes.byref(row))
        assert rc == 0, rc
        d = np.frombuffer(bytes(desc), dtype=np.dtype([("kind", "u1"), ("gate", "u1"), ("region", "<u2"), ("index", "<u4")]))
        d = d[:nc.value]
        return dict(cells=cells[:nc.value].copy(), kind=d["kind"].copy(), gate=d["gate"].copy(), index=d["index"].copy(),
                    region=d["region"].copy(),
                    witness=wit[:nw.value].copy(), out_row=row.value)

    # ---- f3: the cells of insert_leaf outside its hashes (oracle/gadget.c) ----
    def less_than_lookup_rows(self, lookup_bits=18):
        n = ctypes.c_size_t()
        assert self.lib.orc_less_than_lookup_rows(ctypes.c_uint(lookup_bits), None, ctypes.c_size_t(0), ctypes.byref(n)) == 0
        rows = np.empty(n.value, dtype=np.uint32)
        assert self.lib.orc_less_than_lookup_rows(ctypes.c_uint(lookup_bits), rows.ctypes.data_as(ctypes.c_void_p),
                                                  ctypes.c_size_t(n.value), None) == 0
        return rows

    def less_than_trace(self, a, b, lookup_bits=18):
        """the column of ONE is_less_than(a_q, a_r, b_q, b_r) (src/indexed_merkle_tree.rs:98-125): same dict as hash_trace"""
        cap, wcap = 1200, 400
        cells = np.empty((cap, 32), np.uint8)
        desc = (TraceCell * cap)()
        wit = np.empty((wcap, 32), np.uint8)
        nc, nw, row = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_uint32()
        rc = self.lib.orc_less_than_trace(b32(a), b32(b), ctypes.c_uint(lookup_bits), cells.ctypes.data_as(ctypes.c_void_p), desc,
                                          ctypes.c_size_t(cap), ctypes.byref(nc), wit.ctypes.data_as(ctypes.c_void_p),
                                          ctypes.c_size_t(wcap), ctypes.byref(nw), ctypes.byref(row))
        assert rc == 0, rc
        d = np.frombuffer(bytes(desc), dtype=np.dtype([("kind", "u1"), ("gate", "u1"), ("region", "<u2"), ("index", "<u4")]))
        d = d[:nc.value]
        assert nw.value == self.lib.orc_less_than_trace_rows(ctypes.c_uint(lookup_bits))
        return dict(cells=cells[:nc.value].copy(), kind=d["kind"].copy(), gate=d["gate"].copy(), index=d["index"].copy(),
                    region=d["region"].copy(), witness=wit[:nw.value].copy(), out_row=row.value)

    def insert_gadget_trace(self, low_leaf3, low_index, low_proof, new_leaf3, new_index, new_proof, largest, depth,
                            lookup_bits=18, new_path_index=None):
        """(glue rows uint8 [g, 32], segments [(kind, arity, first_row, n_rows)]) of one insert_leaf (:231-314)"""
        self.lib.orc_insert_gadget_rows.restype = ctypes.c_size_t
        g = self.lib.orc_insert_gadget_rows(ctypes.c_size_t(depth), ctypes.c_uint(lookup_bits))
        wit = np.empty((g, 32), np.uint8)
        segs = np.zeros(8 * depth + 16, dtype=np.dtype([("kind", "<u4"), ("arity", "<u4"), ("first_row", "<u8"), ("n_rows", "<u8")]))
        nw, ns = ctypes.c_size_t(), ctypes.c_size_t()
        lp = np.ascontiguousarray(low_proof, dtype=np.uint8)
        npf = np.ascontiguousarray(new_proof, dtype=np.uint8)
        rc = self.lib.orc_insert_gadget_trace(b"".join(b32(x) for x in low_leaf3), ctypes.c_uint64(low_index),
                                              lp.ctypes.data_as(ctypes.c_void_p), b"".join(b32(x) for x in new_leaf3),
                                              ctypes.c_uint64(new_index),
                                              ctypes.c_uint64(new_index if new_path_index is None else new_path_index),
                                              npf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(1 if largest else 0),
                                              ctypes.c_size_t(depth), ctypes.c_uint(lookup_bits),
                                              wit.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(g), ctypes.byref(nw),
                                              segs.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(segs)), ctypes.byref(ns))
        assert rc == 0 and nw.value == g, (rc, nw.value, g)
        return wit, [tuple(int(x) for x in s_) for s_ in segs[:ns.value]]

    def non_inclusion_gadget_trace(self, low_leaf3, low_index, low_proof, new_val, largest, depth, lookup_bits=18):
        """(glue rows uint8 [g, 32], segments) of one verify_non_inclusion alone (:127-229)"""
        self.lib.orc_non_inclusion_gadget_rows.restype = ctypes.c_size_t
        g = self.lib.orc_non_inclusion_gadget_rows(ctypes.c_size_t(depth), ctypes.c_uint(lookup_bits))
        wit = np.empty((g, 32), np.uint8)
        segs = np.zeros(2 * depth + 8, dtype=np.dtype([("kind", "<u4"), ("arity", "<u4"), ("first_row", "<u8"), ("n_rows", "<u8")]))
        nw, ns = ctypes.c_size_t(), ctypes.c_size_t()
        lp = np.ascontiguousarray(low_proof, dtype=np.uint8)
        rc = self.lib.orc_non_inclusion_gadget_trace(b"".join(b32(x) for x in low_leaf3), ctypes.c_uint64(low_index),
                                                     lp.ctypes.data_as(ctypes.c_void_p), b32(new_val),
                                                     ctypes.c_int(1 if largest else 0), ctypes.c_size_t(depth),
                                                     ctypes.c_uint(lookup_bits), wit.ctypes.data_as(ctypes.c_void_p),
                                                     ctypes.c_size_t(g), ctypes.byref(nw), segs.ctypes.data_as(ctypes.c_void_p),
                                                     ctypes.c_size_t(len(segs)), ctypes.byref(ns))
        assert rc == 0 and nw.value == g, (rc, nw.value, g)
        return wit, [tuple(int(x) for x in s_) for s_ in segs[:ns.value]]

    # ---- dense tree (src/utils.rs) ----
    def tree_new(self, leaves_arr):
        leaves_arr = np.ascontiguousarray(leaves_arr, dtype=np.uint8)
        h = ctypes.c_void_p()
        rc = self.lib.orc_tree_new(ctypes.byref(h), leaves_arr.ctypes.data_as(ctypes.c_void_p),
                                   ctypes.c_size_t(leaves_arr.shape[0] if leaves_arr.size else 0))
        return rc, h

    def tree_free(self, h):
        self.lib.orc_tree_free(h)

    def tree_root(self, h):
        out = ctypes.create_string_buffer(32)
        self.lib.orc_tree_get_root(h, out)
        return int.from_bytes(out.raw, "little")

    def tree_level(self, h, level):
        n = ctypes.c_size_t()
        assert self.lib.orc_tree_level(h, ctypes.c_size_t(level), None, ctypes.byref(n)) == 0
        out = np.empty((n.value, 32), dtype=np.uint8)
        self.lib.orc_tree_level(h, ctypes.c_size_t(level), out.ctypes.data_as(ctypes.c_void_p), None)
        return out

    def tree_proof(self, h, index):
        d = self.lib.orc_tree_num_levels(h) - 1
        proof = np.empty((d, 32), dtype=np.uint8)
        helper = np.empty((d, 32), dtype=np.uint8)
        rc = self.lib.orc_tree_get_proof(h, ctypes.c_size_t(index), proof.ctypes.data_as(ctypes.c_void_p),
                                         helper.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, rc
        return proof, helper

    def path_root(self, leaf, index, proof_arr):
        proof_arr = np.ascontiguousarray(proof_arr, dtype=np.uint8)
        out = ctypes.create_string_buffer(32)
        rc = self.lib.orc_path_root(out, b32(leaf), ctypes.c_uint64(index),
                                    proof_arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(proof_arr.shape[0]))
        assert rc == 0, rc
        return int.from_bytes(out.raw, "little")

    def compute_merkle_root(self, leaf, proof_arr, helper_arr):
        proof_arr = np.ascontiguousarray(proof_arr, dtype=np.uint8)
        helper_arr = np.ascontiguousarray(helper_arr, dtype=np.uint8)
        out = ctypes.create_string_buffer(32)
        rc = self.lib.orc_compute_merkle_root(out, b32(leaf), proof_arr.ctypes.data_as(ctypes.c_void_p),
                                              helper_arr.ctypes.data_as(ctypes.c_void_p),
                                              ctypes.c_size_t(proof_arr.shape[0]))
        return rc, int.from_bytes(out.raw, "little")

    def is_less_than_limbs(self, a, b):
        return self.lib.orc_is_less_than_limbs(b32(a), b32(b))

    # ---- circuit relations ----
    def verify_non_inclusion(self, root, low_leaf3, proof_arr, helper_arr, new_val, largest):
        proof_arr = np.ascontiguousarray(proof_arr, dtype=np.uint8)
        helper_arr = np.ascontiguousarray(helper_arr, dtype=np.uint8)
        lh = ctypes.create_string_buffer(32)
        ro = ctypes.create_string_buffer(32)
        ll = b"".join(b32(x) for x in low_leaf3)
        f = self.lib.orc_verify_non_inclusion(b32(root), ll, proof_arr.ctypes.data_as(ctypes.c_void_p),
                                              helper_arr.ctypes.data_as(ctypes.c_void_p),
                                              ctypes.c_size_t(proof_arr.shape[0]), b32(new_val), int(largest), lh, ro)
        return f, int.from_bytes(ro.raw, "little")

    def insert_leaf(self, old_root, low_leaf3, low_proof, low_helper, new_root, new_leaf3, new_index, new_proof,
                    new_helper, largest):
        lp = np.ascontiguousarray(low_proof, dtype=np.uint8)
        lh = np.ascontiguousarray(low_helper, dtype=np.uint8)
        np_ = np.ascontiguousarray(new_proof, dtype=np.uint8)
        nh = np.ascontiguousarray(new_helper, dtype=np.uint8)
        tr = InsertTrace()
        f = self.lib.orc_insert_leaf(b32(old_root), b"".join(b32(x) for x in low_leaf3),
                                     lp.ctypes.data_as(ctypes.c_void_p), lh.ctypes.data_as(ctypes.c_void_p),
                                     b32(new_root), b"".join(b32(x) for x in new_leaf3), ctypes.c_uint64(new_index),
                                     np_.ctypes.data_as(ctypes.c_void_p), nh.ctypes.data_as(ctypes.c_void_p),
                                     int(largest), ctypes.c_size_t(lp.shape[0]), ctypes.byref(tr))
        trace = [int.from_bytes(bytes(getattr(tr, n)), "little") for n, _ in InsertTrace._fields_]
        return f, trace

    # ---- test-module insertion (:632-671) ----
    def update_idx_leaf(self, pre_arr, new_val, new_val_idx):
        """pre_arr uint8 [n,3,32], modified in place; returns low idx."""
        low = ctypes.c_uint64()
        rc = self.lib.orc_update_idx_leaf(pre_arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(pre_arr.shape[0]),
                                          b32(new_val), ctypes.c_uint64(new_val_idx), ctypes.byref(low))
        assert rc == 0, rc
        return low.value

    # ---- sparse depth-d tree ----
    def sparse_new(self, depth, cap):
        h = ctypes.c_void_p()
        rc = self.lib.orc_sparse_new(ctypes.byref(h), ctypes.c_uint(depth), ctypes.c_uint64(cap))
        assert rc == 0, rc
        return h

    def sparse_free(self, h):
        self.lib.orc_sparse_free(h)

    def sparse_set_index_base(self, h, base):
        self.lib.orc_sparse_set_index_base(h, ctypes.c_uint64(base))

    def sparse_root(self, h):
        out = ctypes.create_string_buffer(32)
        self.lib.orc_sparse_root(h, out)
        return int.from_bytes(out.raw, "little")

    def sparse_insert(self, h, depth, val):
        low = ctypes.c_uint64()
        largest = ctypes.c_int()
        ll = np.empty((3, 32), np.uint8)
        ir = ctypes.create_string_buffer(32)
        nr = ctypes.create_string_buffer(32)
        lp = np.empty((depth, 32), np.uint8)
        npf = np.empty((depth, 32), np.uint8)
        rc = self.lib.orc_sparse_insert(h, b32(val), ctypes.byref(low), ll.ctypes.data_as(ctypes.c_void_p),
                                        ctypes.byref(largest), ir, nr, lp.ctypes.data_as(ctypes.c_void_p),
                                        npf.ctypes.data_as(ctypes.c_void_p))
        return dict(rc=rc, low=low.value, largest=largest.value, low_leaf=ll,
                    interim_root=int.from_bytes(ir.raw, "little"), new_root=int.from_bytes(nr.raw, "little"),
                    low_proof=lp, new_proof=npf)

    def sparse_load(self, h, preimages):
        """hash_nullifier_pre_images + IndexedMerkleTree::new over the first n leaves (oracle/sparse.c orc_sparse_load)"""
        pre = np.ascontiguousarray(preimages, dtype=np.uint8).reshape(-1, 3, 32)
        return self.lib.orc_sparse_load(h, pre.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(pre.shape[0]))

    def sparse_proof(self, h, depth, index):
        out = np.empty((depth, 32), np.uint8)
        assert self.lib.orc_sparse_proof(h, ctypes.c_uint64(index), out.ctypes.data_as(ctypes.c_void_p)) == 0
        return out

    def sparse_preimage(self, h, index):
        out = np.empty((3, 32), np.uint8)
        assert self.lib.orc_sparse_preimage(h, ctypes.c_uint64(index), out.ctypes.data_as(ctypes.c_void_p)) == 0
        return out


def arr_ints(a):
    return [int.from_bytes(x.tobytes(), "little") for x in np.asarray(a, dtype=np.uint8).reshape(-1, 32)]


def less_than_rows_model(a, b, lookup_bits=18):
    """Test-side restatement of the NEW witnesses of is_less_than(a_q, a_r, b_q, b_r) (src/indexed_merkle_tree.rs:98-125)
    with Python integers, written from the gadget descriptions in oracle/gadget.c's header and independent of its code:
    per limb pair range.is_less_than (shifted difference, shifted a, the limbs of the range check with their running
    sums, is_zero of the top limb), gate.is_equal (difference, is_zero), then not x4, and x3, and, or."""
    k = -(-128 // lookup_bits)
    padded, L = k * lookup_bits, k + 1
    inv = lambda x: pow(x, P - 2, P) if x % P else 1

    def is_zero(x):
        z = 1 if x % P == 0 else 0
        return [z, inv(x), z], z

    def range_lt(x, y):
        sab, sa = (1 << padded) + x - y, (1 << padded) + x
        rows, acc, limbs = [sab, sa], 0, []
        for i in range(L):
            limb = (sab >> (lookup_bits * i)) & ((1 << lookup_bits) - 1)
            limbs.append(limb)
            acc += limb << (lookup_bits * i)
            rows += [limb] if i == 0 else [limb, acc]
        assert acc == sab
        zr, z = is_zero(limbs[-1])
        return rows + zr, z

    def is_equal(x, y):
        d = (x - y) % P
        zr, z = is_zero(d)
        return [d] + zr, z

    a_q, a_r, b_q, b_r = a >> 128, a & ((1 << 128) - 1), b >> 128, b & ((1 << 128) - 1)
    rows = []
    r, msb_lt = range_lt(a_q, b_q); rows += r
    r, msb_eq = is_equal(a_q, b_q); rows += r
    r, lsb_lt = range_lt(a_r, b_r); rows += r
    r, lsb_eq = is_equal(a_r, b_r); rows += r
    c_not, a_not = 1 - msb_eq, 1 - msb_lt
    c, d_not = 1 - c_not, 1 - lsb_eq
    rows += [c_not, a_not, c, d_not]
    t1 = a_not * lsb_lt; t2 = t1 * c; rhs = t2 * d_not
    lhs = msb_lt * c_not
    rows += [t1, t2, rhs, lhs]
    out = lhs + rhs - lhs * rhs
    rows += [1 - rhs, 1 - rhs, out]
    assert out == (1 if a < b else 0)
    return [x % P for x in rows], out


def insert_gadget_rows_model(orc, low_leaf3, low_index, low_proof, new_leaf3, new_index, new_proof, largest, depth,
                             lookup_bits=18, new_path_index=None):
    """Test-side restatement of the glue rows of insert_leaf (:231-314), Python integers + the oracle's HASH only."""
    new_path_index = new_index if new_path_index is None else new_path_index
    inv = lambda x: pow(x, P - 2, P) if x % P else 1
    M = (1 << 128) - 1
    low_val, low_next, _ = low_leaf3
    nv = new_leaf3[0]
    iz = 1 if low_next == 0 else 0
    rows = [low_next % P, iz, inv(low_next), iz, nv >> 128, nv & M, low_next >> 128, low_next & M, nv, low_next]
    r, lt = less_than_rows_model(nv, low_next, lookup_bits)
    rows += r
    s = 1 if largest else 0
    rows += [iz * s, 1 - s, (1 - s) * lt + iz * s]

    def chain(leaf, index, proof):
        out = [leaf]
        cur = leaf
        for l, sib in enumerate(arr_ints(proof)[:depth]):
            sw = ((index >> l) & 1) ^ 1
            left, right = (cur, sib) if sw else (sib, cur)
            out += [(cur - sib) % P, (sib - cur) % P, left, right]
            cur = orc.hash([left, right])
        return out
    rows += chain(orc.hash(low_leaf3), low_index, low_proof)
    rows += [low_val >> 128, low_val & M, low_val]
    r, _ = less_than_rows_model(low_val, nv, lookup_bits)
    rows += r
    rows += chain(orc.hash([low_val, nv, new_index]), low_index, low_proof)
    rows += chain(orc.hash([0, 0, 0]), new_path_index, new_proof)
    rows += chain(orc.hash(new_leaf3), new_path_index, new_proof)
    return rows


def path_trace(orc, start, leaf3, index, proof, depth):
    """The witness rows of every hash_fix_len_array call of one compute_merkle_root
    (/root/reference/src/indexed_merkle_tree.rs:78-96), in call order: the 3-input leaf hash first when `leaf3` is given
    (:193-194), else the chain starts from the value `start`; then one hash per level with the (left, right) pair
    dual_mux selects (:47-63).  Returns (rows uint8 [k, 32] canonical, root)."""
    parts = []
    if leaf3 is not None:
        parts.append(orc.hash_trace(leaf3)["witness"])
        cur = orc.hash(leaf3)
    else:
        cur = start
    sib = arr_ints(proof)
    for l in range(depth):
        pair = [sib[l], cur] if (index >> l) & 1 else [cur, sib[l]]
        parts.append(orc.hash_trace(pair)["witness"])
        cur = orc.hash(pair)
    return (np.concatenate(parts) if parts else np.zeros((0, 32), np.uint8)), cur


def insert_leaf_trace(orc, low_leaf3, low_index, low_proof, new_leaf3, new_index, new_proof, depth):
    """All 3 + 4 * depth hash traces of one insert_leaf call (:231-314) in the order the circuit reaches
    hash_fix_len_array: low leaf + path (:193-204), rewritten low leaf {low.val, new.val, new_index} + the same path
    (:265-284), the zero leaf's path at the new slot (:286-294; the zero-leaf hash is a constant, no trace), new leaf +
    path (:299-312).  Returns (rows uint8 [3*1209 + 4*depth*1208, 32], [old_root, interim_root, interim_root', new_root])."""
    new_low = [low_leaf3[0], new_leaf3[0], new_index]
    chains = [(None, low_leaf3, low_index, low_proof), (None, new_low, low_index, low_proof),
              (orc.hash([0, 0, 0]), None, new_index, new_proof), (None, new_leaf3, new_index, new_proof)]
    rows, roots = [], []
    for start, leaf3, idx, proof in chains:
        r, root = path_trace(orc, start, leaf3, idx, proof, depth)
        rows.append(r)
        roots.append(root)
    return np.concatenate(rows), roots


_cached = None


def load():
    global _cached
    if _cached is not None:
        return _cached
    so = os.path.join(ODIR, "liboracle.so")
    srcs = [os.path.join(ODIR, f) for f in ("fr.c", "poseidon.c", "tree.c", "indexed.c", "sparse.c", "trace.c", "gadget.c",
                                            "column.h", "imt_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        r = subprocess.run(["make", "-C", ODIR, "liboracle.so"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    _cached = Oracle(ctypes.CDLL(so))
    return _cached


def synth_values(n, seed):
    """Synthetic insertion values: xoshiro256** seeded from `seed`, 4 limbs, top 2
    bits cleared, rejected until 0 < v < p, de-duplicated (mirrors the reference's random draw
    at src/indexed_merkle_tree.rs:381-386)."""
    mask = (1 << 64) - 1

    def rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & mask

    # splitmix64 seeding
    s = []
    z = seed & mask
    for _ in range(4):
        z = (z + 0x9E3779B97F4A7C15) & mask
        x = z
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & mask
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & mask
        s.append(x ^ (x >> 31))
    out, seen = [], set()
    while len(out) < n:
        limbs = []
        for _ in range(4):
            r = (rotl((s[1] * 5) & mask, 7) * 9) & mask
            t = (s[1] << 17) & mask
            s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
            s[2] ^= t
            s[3] = rotl(s[3], 45)
            limbs.append(r)
        v = limbs[0] | (limbs[1] << 64) | (limbs[2] << 128) | ((limbs[3] & ((1 << 62) - 1)) << 192)
        if 0 < v < P and v not in seen:
            seen.add(v)
            out.append(v)
    return out
