"""Host-side mirror of the reference's interface for the hot path, on top of the C ABI.

Names, argument meaning and error behaviour follow /root/reference:
  IndexedMerkleTree.new / get_root / get_proof / verify_proof      src/utils.rs:19-108
  verify_non_inclusion, insert_leaf                                src/indexed_merkle_tree.rs:127, :231
  IndexedTree.insert_batch  = update_idx_leaf + rebuild, batched   src/indexed_merkle_tree.rs:632-671, :715-735
Field elements are Python ints (canonical) at this level and numpy uint8[..., 32]
little-endian arrays underneath.  All arithmetic happens in libimt_hip.so on the GPU.
"""
import ctypes
import weakref

import numpy as np

from . import _ffi
from ._ffi import lib

P_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class ImtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"imt error {code}: {msg}")
        self.code = code


class ConstraintError(AssertionError):
    """A constraint / assert of the reference's circuit does not hold; .mask has the IMT_F_* bits."""

    def __init__(self, mask):
        super().__init__(f"unsatisfied constraints, mask=0x{mask:02x}")
        self.mask = mask


def to_bytes(x):
    """int or iterable of ints (any nesting) -> uint8 array [..., 32] little-endian."""
    if isinstance(x, (int, np.integer)):
        return np.frombuffer(int(x).to_bytes(32, "little"), dtype=np.uint8).copy()
    x = list(x)
    if x and all(isinstance(v, (int, np.integer)) for v in x):      # flat list: one join instead of n arrays
        return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in x), dtype=np.uint8).reshape(len(x), 32).copy()
    a = [to_bytes(v) for v in x]
    return np.stack(a) if a else np.zeros((0, 32), dtype=np.uint8)


def to_int(b):
    """uint8 array [..., 32] -> int or nested list of ints."""
    b = np.asarray(b, dtype=np.uint8)
    if b.ndim == 1:
        return int.from_bytes(b.tobytes(), "little")
    return [to_int(x) for x in b]


def _arr(x, shape_tail):
    a = np.ascontiguousarray(x, dtype=np.uint8)
    if a.ndim < len(shape_tail) or tuple(a.shape[-len(shape_tail):]) != tuple(shape_tail):
        raise ValueError(f"expected trailing shape {shape_tail}, got {a.shape}")
    return a


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Context:
    """imt_ctx: one per host thread; owns the Poseidon tables on one GPU."""

    def __init__(self, device=0):
        h = ctypes.c_void_p()
        rc = lib.imt_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise ImtError(rc, "imt_ctx_create failed (no usable HIP device?)" if rc == _ffi.ERR["NO_DEVICE"]
                           else "imt_ctx_create failed")
        self.h = h
        self.device = device
        self._children = weakref.WeakSet()      # trees of this context: imt.h wants them destroyed before it

    def close(self):
        if getattr(self, "h", None):
            for child in list(getattr(self, "_children", ())):
                child.close()
            lib.imt_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise ImtError(rc, lib.imt_last_error(self.h).decode())

    def set_stream(self, stream_ptr):
        self._check(lib.imt_ctx_set_stream(self.h, ctypes.c_void_p(stream_ptr)))

    def sync(self):
        self._check(lib.imt_ctx_sync(self.h))

    def set_option(self, option, value):
        self._check(lib.imt_ctx_set_option(self.h, option, value))

    def host_alloc(self, shape, dtype=np.uint8):
        """numpy array over page-locked, device-addressable host memory (imt_host_alloc); pass its
        `.ctypes.data` wherever IMT_DEVICE_PTRS expects a device pointer.  Free with host_free(arr)."""
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        self._check(lib.imt_host_alloc(self.h, max(nbytes, 1), ctypes.byref(p)))
        buf = (ctypes.c_uint8 * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self._check(lib.imt_host_free(self.h, ctypes.c_void_p(p)))

    # ---- a1 / a10 ----
    def hash2(self, pairs, fmt=0):
        a = _arr(pairs, (2, 32))
        n = a.shape[0]
        out = np.empty((n, 32), dtype=np.uint8)
        self._check(lib.imt_hash2_batch(self.h, _p(a), _p(out), n, fmt))
        return out

    def hash3(self, triples, fmt=0):
        a = _arr(triples, (3, 32))
        n = a.shape[0]
        out = np.empty((n, 32), dtype=np.uint8)
        self._check(lib.imt_hash3_batch(self.h, _p(a), _p(out), n, fmt))
        return out

    def permute(self, states, fmt=0):
        a = _arr(states, (3, 32))
        out = np.empty_like(a)
        self._check(lib.imt_permute_batch(self.h, _p(a), _p(out), a.shape[0], fmt))
        return out

    # ---- f1: witness trace of hash_fix_len_array ----
    def hash_trace(self, inputs, fmt=0, item_major=False):
        """inputs uint8 [n, 2 or 3, 32] -> every new advice value of halo2-base's hash_fix_len_array per hash,
        in assignment order: uint8 [rows, n, 32] (item_major: [n, rows, 32]); rows = 1208 / 1209."""
        a = np.ascontiguousarray(inputs, dtype=np.uint8)
        if a.ndim != 3 or a.shape[1] not in (2, 3) or a.shape[2] != 32:
            raise ValueError(f"expected [n, 2|3, 32], got {a.shape}")
        n, arity = a.shape[0], a.shape[1]
        rows = lib.imt_hash_trace_rows(arity)
        out = np.empty((n, rows, 32) if item_major else (rows, n, 32), dtype=np.uint8)
        self._check(lib.imt_hash_trace_batch(self.h, _p(a), arity, n, _p(out),
                                             fmt | (_ffi.TRACE_ITEM_MAJOR if item_major else 0)))
        return out

    def path_trace(self, index, sib, depth, leaf=None, leaf3=None, fmt=0, item_major=False):
        """Traces of every hash of compute_merkle_root for n paths (imt_path_trace_batch); returns (trace, roots).
        trace: uint8 [total_rows, n, 32] (blocks: leaf hash if leaf3, then the levels) or item-major [n, total_rows, 32]
        (siblings then item-major too: sib[item][level])."""
        src = _arr(leaf3, (3, 32)) if leaf3 is not None else _arr(leaf, (32,))
        n = src.shape[0]
        idx = np.ascontiguousarray(index, dtype=np.uint64)
        sb = _arr(sib, (32,)) if depth else np.zeros((0, 32), np.uint8)
        total = (lib.imt_hash_trace_rows(3) if leaf3 is not None else 0) + depth * lib.imt_hash_trace_rows(2)
        out = np.empty((n, total, 32) if item_major else (total, n, 32), dtype=np.uint8)
        roots = np.empty((n, 32), dtype=np.uint8)
        self._check(lib.imt_path_trace_batch(self.h, _p(src) if leaf3 is None else None, _p(src) if leaf3 is not None else None,
                                             _p(idx), _p(sb), depth, n, _p(out), _p(roots),
                                             fmt | (_ffi.TRACE_ITEM_MAJOR if item_major else 0)))
        return out, roots

    def insert_trace(self, low_leaf, low_index, low_sib, new_leaf, new_index, new_sib, depth, new_path_index=None,
                     fmt=0, item_major=False):
        """All 3 + 4*depth hash traces of insert_leaf for n insertions (imt_insert_trace_batch), in the circuit's call
        order; uint8 [rows_total, n, 32] or item-major [n, rows_total, 32] (siblings then item-major too)."""
        ll = _arr(low_leaf, (3, 32))
        n = ll.shape[0]
        args = [ll, np.ascontiguousarray(low_index, dtype=np.uint64), _arr(low_sib, (32,)), _arr(new_leaf, (3, 32)),
                np.ascontiguousarray(new_index, dtype=np.uint64),
                None if new_path_index is None else np.ascontiguousarray(new_path_index, dtype=np.uint64),
                _arr(new_sib, (32,))]
        total = lib.imt_insert_trace_rows(depth)
        out = np.empty((n, total, 32) if item_major else (total, n, 32), dtype=np.uint8)
        self._check(lib.imt_insert_trace_batch(self.h, *[_p(a) for a in args], depth, n, _p(out),
                                               fmt | (_ffi.TRACE_ITEM_MAJOR if item_major else 0)))
        return out

    # ---- f3: the advice values of insert_leaf outside its hashes ----
    def less_than_trace(self, a, b, lookup_bits=18, fmt=0, item_major=False):
        """rows of is_less_than(a_q, a_r, b_q, b_r) (src/indexed_merkle_tree.rs:98-125) for n pairs of 256-bit values:
        (uint8 [rows, n, 32] or item-major [n, rows, 32], lt uint8 [n])"""
        a, b = _arr(a, (32,)), _arr(b, (32,))
        n = a.shape[0]
        rows = lib.imt_less_than_trace_rows(lookup_bits)
        out = np.empty((n, rows, 32) if item_major else (rows, n, 32), dtype=np.uint8)
        lt = np.empty(n, dtype=np.uint8)
        self._check(lib.imt_less_than_trace_batch(self.h, _p(a), _p(b), n, lookup_bits, _p(out), _p(lt),
                                                  fmt | (_ffi.TRACE_ITEM_MAJOR if item_major else 0)))
        return out, lt

    def less_than_layout(self, lookup_bits=18, fmt=0):
        """(cells, constants, out_row) of one is_less_than call (imt_less_than_trace_layout); inputs 0..3 = a_q, a_r, b_q, b_r"""
        return trace_layout(lambda *a: lib.imt_less_than_trace_layout(self.h, *a), lookup_bits, fmt, self._check)

    @staticmethod
    def less_than_lookup_rows(lookup_bits=18):
        """trace rows of one is_less_than that the RangeChip also adds to its lookup table (the limbs of both range
        checks), in column order (imt_less_than_lookup_rows)"""
        n = ctypes.c_size_t()
        if lib.imt_less_than_lookup_rows(lookup_bits, None, 0, ctypes.byref(n)):
            raise ValueError("1 <= lookup_bits <= 28")
        rows = np.empty(n.value, dtype=np.uint32)
        assert lib.imt_less_than_lookup_rows(lookup_bits, rows.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n.value, None) == 0
        return rows

    @staticmethod
    def insert_gadget_lookup_rows(depth, lookup_bits=18):
        """glue rows of one insert_leaf (insert_gadget_trace) that are also lookup cells: both comparisons' limbs"""
        n = ctypes.c_size_t()
        if lib.imt_insert_gadget_lookup_rows(depth, lookup_bits, None, 0, ctypes.byref(n)):
            raise ValueError("depth >= 1 and 1 <= lookup_bits <= 28")
        rows = np.empty(n.value, dtype=np.uint32)
        assert lib.imt_insert_gadget_lookup_rows(depth, lookup_bits, rows.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n.value, None) == 0
        return rows

    def insert_gadget_trace(self, low_leaf, low_index, low_sib, new_leaf, new_index, new_sib, is_largest, depth,
                            lookup_bits=18, new_path_index=None, fmt=0, item_major=False):
        """every new advice value of insert_leaf OUTSIDE its hashes for n insertions (imt_insert_gadget_trace_batch):
        uint8 [rows, n, 32] or item-major [n, rows, 32] (siblings then item-major too)"""
        ll = _arr(low_leaf, (3, 32))
        n = ll.shape[0]
        args = [ll, np.ascontiguousarray(low_index, dtype=np.uint64), _arr(low_sib, (32,)), _arr(new_leaf, (3, 32)),
                np.ascontiguousarray(new_index, dtype=np.uint64),
                None if new_path_index is None else np.ascontiguousarray(new_path_index, dtype=np.uint64),
                _arr(new_sib, (32,)), np.ascontiguousarray(is_largest, dtype=np.uint8)]
        rows = lib.imt_insert_gadget_rows(depth, lookup_bits)
        if not rows:
            raise ValueError("depth >= 1 and 1 <= lookup_bits <= 28")
        out = np.empty((n, rows, 32) if item_major else (rows, n, 32), dtype=np.uint8)
        self._check(lib.imt_insert_gadget_trace_batch(self.h, *[_p(a) for a in args], depth, lookup_bits, n, _p(out),
                                                      fmt | (_ffi.TRACE_ITEM_MAJOR if item_major else 0)))
        return out

    def non_inclusion_gadget_trace(self, low_leaf, low_index, low_sib, new_val, is_largest, depth, lookup_bits=18, fmt=0,
                                   item_major=False):
        """every new advice value of ONE verify_non_inclusion outside its hashes, for n candidates
        (imt_non_inclusion_gadget_trace_batch): uint8 [rows, n, 32] or item-major [n, rows, 32]"""
        ll = _arr(low_leaf, (3, 32))
        n = ll.shape[0]
        rows = lib.imt_non_inclusion_gadget_rows(depth, lookup_bits)
        if not rows:
            raise ValueError("depth >= 1 and 1 <= lookup_bits <= 28")
        out = np.empty((n, rows, 32) if item_major else (rows, n, 32), dtype=np.uint8)
        self._check(lib.imt_non_inclusion_gadget_trace_batch(
            self.h, _p(ll), _p(np.ascontiguousarray(low_index, dtype=np.uint64)), _p(_arr(low_sib, (32,))), _p(_arr(new_val, (32,))),
            _p(np.ascontiguousarray(is_largest, dtype=np.uint8)), depth, lookup_bits, n, _p(out),
            fmt | (_ffi.TRACE_ITEM_MAJOR if item_major else 0)))
        return out

    def hash_trace_layout(self, arity, fmt=0):
        """(cells, constants, out_row): the advice column of one hash, cell by cell (imt_hash_trace_layout).
        cells: structured array with fields kind (_ffi.CELL_*), gate, index; constants: uint8 [k, 32]."""
        return trace_layout(lambda *a: lib.imt_hash_trace_layout(self.h, *a), arity, fmt, self._check)

    # ---- a5 / a8 / a9 ----
    def path_root(self, leaf, index, sib, depth, item_major=False, fmt=0):
        leaf = _arr(leaf, (32,))
        n = leaf.shape[0]
        idx = np.ascontiguousarray(index, dtype=np.uint64)
        sib = _arr(sib, (32,)) if depth else np.zeros((0, 32), np.uint8)
        assert sib.size == depth * n * 32 and idx.size == n
        out = np.empty((n, 32), dtype=np.uint8)
        flags = fmt | (_ffi.SIB_ITEM_MAJOR if item_major else 0)
        self._check(lib.imt_path_root_batch(self.h, _p(leaf), _p(idx), _p(sib), depth, n, _p(out), flags))
        return out

    def compute_merkle_root(self, leaf, helper_mask, sib, depth, item_major=False, fmt=0):
        leaf = _arr(leaf, (32,))
        n = leaf.shape[0]
        hm = np.ascontiguousarray(helper_mask, dtype=np.uint64)
        sib = _arr(sib, (32,)) if depth else np.zeros((0, 32), np.uint8)
        out = np.empty((n, 32), dtype=np.uint8)
        flags = fmt | (_ffi.SIB_ITEM_MAJOR if item_major else 0)
        self._check(lib.imt_compute_merkle_root_batch(self.h, _p(leaf), _p(hm), _p(sib), depth, n, _p(out), flags))
        return out

    def verify_proof_batch(self, leaf, index, root, sib, depth, item_major=False, fmt=0):
        leaf = _arr(leaf, (32,))
        n = leaf.shape[0]
        idx = np.ascontiguousarray(index, dtype=np.uint64)
        root = _arr(root, (32,))
        per_item = root.ndim == 2
        sib = _arr(sib, (32,)) if depth else np.zeros((0, 32), np.uint8)
        ok = np.empty(n, dtype=np.uint8)
        flags = fmt | (_ffi.SIB_ITEM_MAJOR if item_major else 0) | (_ffi.ROOT_PER_ITEM if per_item else 0)
        self._check(lib.imt_verify_proof_batch(self.h, _p(leaf), _p(idx), _p(root), _p(sib), depth, n, _p(ok), flags))
        return ok.astype(bool)

    # ---- a13 ----
    def non_membership(self, root, low_leaf, low_index, low_sib, depth, new_val, is_largest, item_major=False,
                       fmt=0, want_root=False):
        low_leaf = _arr(low_leaf, (3, 32))
        n = low_leaf.shape[0]
        root = _arr(root, (32,))
        per_item = root.ndim == 2
        idx = np.ascontiguousarray(low_index, dtype=np.uint64)
        sib = _arr(low_sib, (32,)) if depth else np.zeros((0, 32), np.uint8)
        nv = _arr(new_val, (32,))
        lg = np.ascontiguousarray(is_largest, dtype=np.uint8)
        fail = np.empty(n, dtype=np.uint8)
        rout = np.empty((n, 32), dtype=np.uint8) if want_root else None
        flags = fmt | (_ffi.SIB_ITEM_MAJOR if item_major else 0) | (_ffi.ROOT_PER_ITEM if per_item else 0)
        self._check(lib.imt_non_membership_batch(self.h, _p(root), _p(low_leaf), _p(idx), _p(sib), depth, _p(nv),
                                                 _p(lg), n, _p(fail), _p(rout), flags))
        return (fail, rout) if want_root else fail

    def split128(self, vals, fmt=0):
        """(q, r) limb witnesses of verify_non_inclusion (src/indexed_merkle_tree.rs:145-178)."""
        v = _arr(vals, (32,))
        q = np.empty_like(v)
        r = np.empty_like(v)
        self._check(lib.imt_split128_batch(self.h, _p(v), _p(q), _p(r), v.shape[0], fmt))
        return q, r

    # ---- a14 ----
    def insert_witness(self, old_root, low_leaf, low_index, low_sib, new_root, new_leaf, new_index, new_sib,
                       is_largest, depth, item_major=False, fmt=0, want_trace=False, new_path_index=None):
        low_leaf = _arr(low_leaf, (3, 32))
        n = low_leaf.shape[0]
        args = [_arr(old_root, (32,)), low_leaf, np.ascontiguousarray(low_index, dtype=np.uint64),
                _arr(low_sib, (32,)), _arr(new_root, (32,)), _arr(new_leaf, (3, 32)),
                np.ascontiguousarray(new_index, dtype=np.uint64),
                None if new_path_index is None else np.ascontiguousarray(new_path_index, dtype=np.uint64),
                _arr(new_sib, (32,)), np.ascontiguousarray(is_largest, dtype=np.uint8)]
        fail = np.empty(n, dtype=np.uint8)
        trace = np.empty((7, n, 32), dtype=np.uint8) if want_trace else None
        flags = fmt | (_ffi.SIB_ITEM_MAJOR if item_major else 0)
        self._check(lib.imt_insert_witness_batch(self.h, *[_p(a) for a in args], depth, n, _p(fail), _p(trace), flags))
        return (fail, trace) if want_trace else fail

    # ---- e ----
    def zero_hashes(self, depth, fmt=0):
        out = np.empty((depth + 1, 32), dtype=np.uint8)
        self._check(lib.imt_zero_hashes(self.h, depth, _p(out), fmt))
        return out

    def combine_subtree_roots(self, sub_roots, sub_height, depth, fmt=0):
        r = _arr(sub_roots, (32,))
        out = np.empty(32, dtype=np.uint8)
        self._check(lib.imt_combine_subtree_roots(self.h, _p(r), r.shape[0], sub_height, depth, _p(out), fmt))
        return out


CELL_DTYPE = np.dtype([("kind", "u1"), ("gate", "u1"), ("region", "<u2"), ("index", "<u4")])


def trace_layout(call, arity, fmt, check):
    """shared by Context.hash_trace_layout and the CPU tests (which call the same C function on a GPU-less build)"""
    nc, nk, row = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_uint32()
    check(call(arity, None, 0, ctypes.byref(nc), None, 0, ctypes.byref(nk), ctypes.byref(row), fmt))
    cells = np.zeros(nc.value, dtype=CELL_DTYPE)
    consts = np.empty((nk.value, 32), dtype=np.uint8)
    check(call(arity, cells.ctypes.data_as(ctypes.POINTER(_ffi.TraceCell)), nc.value, ctypes.byref(nc),
               consts.ctypes.data_as(ctypes.c_void_p), nk.value, ctypes.byref(nk), ctypes.byref(row), fmt))
    return cells, consts, row.value


def _column_segments(fn, name, depth, lookup_bits):
    n = ctypes.c_size_t()
    rc = fn(depth, lookup_bits, None, 0, ctypes.byref(n))
    if rc:
        raise ImtError(rc, name)
    segs = (_ffi.ColumnSegment * n.value)()
    rc = fn(depth, lookup_bits, segs, n.value, ctypes.byref(n))
    if rc:
        raise ImtError(rc, name)
    return [(s.kind, s.arity, s.first_row, s.n_rows) for s in segs]


def insert_column_segments(depth, lookup_bits=18):
    """[(kind, arity, first_row, n_rows)]: how the glue rows (kind 0, imt_insert_gadget_trace_batch) and the hash blocks
    (kind 1, imt_insert_trace_batch) interleave in insert_leaf's advice column"""
    return _column_segments(lib.imt_insert_column_segments, "imt_insert_column_segments", depth, lookup_bits)


def non_inclusion_column_segments(depth, lookup_bits=18):
    """the same for one verify_non_inclusion alone: glue rows of imt_non_inclusion_gadget_trace_batch, hash blocks of
    imt_path_trace_batch (leaf3 form)"""
    return _column_segments(lib.imt_non_inclusion_column_segments, "imt_non_inclusion_column_segments", depth, lookup_bits)


def rebuild_advice_column(cells, consts, inputs, trace_rows):
    """The full advice column of one hash as Python ints from its layout, the constants (canonical), the hash inputs
    and the trace rows (canonical): what a chip assigns.  INIT cells are the hasher's initial state [2^64, 0, 0]."""
    init = [1 << 64, 0, 0]
    k = [int.from_bytes(c.tobytes(), "little") for c in consts]
    w = [int.from_bytes(r.tobytes(), "little") for r in np.asarray(trace_rows, dtype=np.uint8).reshape(-1, 32)]
    col = []
    for c in cells:
        kind, idx = int(c["kind"]), int(c["index"])
        col.append(k[idx] if kind == _ffi.CELL_CONST else inputs[idx] if kind == _ffi.CELL_INPUT
                   else init[idx] if kind == _ffi.CELL_INIT else w[idx])
    return col


def check_vertical_gates(cells, col):
    """halo2-base's only custom gate: q * (a + b * c - d) = 0 over four consecutive cells; returns the gate count"""
    n = 0
    for i in np.nonzero(cells["gate"])[0]:
        a, b, c, d = col[i:i + 4]
        if (a + b * c - d) % P_MODULUS:
            raise AssertionError(f"gate at cell {i} does not hold")
        n += 1
    return n


class IndexedMerkleTree:
    """The reference's dense native tree (src/utils.rs:5-108) with the build on the GPU."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        ctx._children.add(self)

    @classmethod
    def new(cls, ctx, leaves):
        """IndexedMerkleTree::new (src/utils.rs:20-57); `leaves` = ints or uint8[n,32]."""
        a = to_bytes(leaves) if not isinstance(leaves, np.ndarray) else _arr(leaves, (32,))
        h = ctypes.c_void_p()
        rc = lib.imt_tree_new(ctx.h, _p(a) if a.size else None, a.shape[0], 0, ctypes.byref(h))
        if rc == _ffi.ERR["NO_LEAVES"]:
            raise ValueError("Cannot create Merkle Tree with no leaves")       # src/utils.rs:25
        if rc == _ffi.ERR["ODD_LEAVES"]:
            raise ValueError("Leaves must be even")                            # src/utils.rs:35
        if rc == _ffi.ERR["NOT_POW2"]:
            raise IndexError("index out of bounds (leaf count not a power of two)")   # panic at src/utils.rs:45
        ctx._check(rc)
        return cls(ctx, h)

    def close(self):
        if self.h:
            lib.imt_tree_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_levels(self):
        return lib.imt_tree_num_levels(self.h)

    def get_root(self):
        out = np.empty(32, dtype=np.uint8)
        self.ctx._check(lib.imt_tree_get_root(self.h, _p(out), 0))
        return to_int(out)

    def get_level(self, level):
        n = ctypes.c_size_t()
        self.ctx._check(lib.imt_tree_get_level(self.h, level, None, ctypes.byref(n), 0))
        out = np.empty((n.value, 32), dtype=np.uint8)
        self.ctx._check(lib.imt_tree_get_level(self.h, level, _p(out), None, 0))
        return out

    def get_proof(self, index):
        """(proof, proof_helper) as lists of ints: src/utils.rs:63-85."""
        d = self.num_levels() - 1
        proof = np.empty((d, 32), dtype=np.uint8)
        helper = np.empty((d, 32), dtype=np.uint8)
        rc = lib.imt_tree_get_proof(self.h, index, _p(proof), _p(helper), 0)
        if rc == _ffi.ERR["RANGE"]:
            raise IndexError("leaf index out of range")
        self.ctx._check(rc)
        return to_int(proof) if d else [], to_int(helper) if d else []

    def get_proof_batch(self, index, item_major=False):
        idx = np.ascontiguousarray(index, dtype=np.uint64)
        d = self.num_levels() - 1
        out = np.empty((idx.size, d, 32) if item_major else (d, idx.size, 32), dtype=np.uint8)
        self.ctx._check(lib.imt_tree_get_proof_batch(self.h, _p(idx), idx.size, _p(out),
                                                     _ffi.SIB_ITEM_MAJOR if item_major else 0))
        return out

    def verify_proof(self, leaf, index, root, proof):
        """src/utils.rs:87-107 (the helper vector is not used, as in the reference)."""
        p = to_bytes(proof) if len(proof) else np.zeros((0, 32), np.uint8)
        ok = self.ctx.verify_proof_batch(to_bytes([leaf]), [index], to_bytes(root), p, len(proof), item_major=True)
        return bool(ok[0])


def _helper_mask(helper):
    m = 0
    for l, h in enumerate(helper):
        if h not in (0, 1):
            raise ConstraintError(_ffi.F_BAD_BIT)     # gate.assert_bit, src/indexed_merkle_tree.rs:54
        m |= int(h) << l
    return m


def _helpers_to_index(helper):
    # helper 1 = left child (src/utils.rs:79) -> index bit 0
    m = _helper_mask(helper)
    return (~m) & ((1 << len(helper)) - 1)


def verify_non_inclusion(ctx, root, low_leaf, low_leaf_proof, low_leaf_proof_helper, new_leaf_value,
                         is_new_leaf_largest):
    """src/indexed_merkle_tree.rs:127-229 on one item.  low_leaf = (val, next_val, next_idx).
    Raises ConstraintError where the reference panics (:190) or leaves a constraint unsatisfied."""
    d = len(low_leaf_proof)
    fail = ctx.non_membership(to_bytes(root), to_bytes([list(low_leaf)]), [_helpers_to_index(low_leaf_proof_helper)],
                              to_bytes(low_leaf_proof), d, to_bytes([new_leaf_value]),
                              [int(is_new_leaf_largest)], item_major=True)
    if fail[0]:
        raise ConstraintError(int(fail[0]))


def insert_leaf(ctx, old_root, low_leaf, low_leaf_proof, low_leaf_proof_helper, new_root, new_leaf,
                new_leaf_index, new_leaf_proof, new_leaf_proof_helper, is_new_leaf_largest):
    """src/indexed_merkle_tree.rs:231-314 on one item; raises ConstraintError when the circuit
    would be unsatisfied.  As in the reference, new_leaf_index is not tied to the helper bits."""
    d = len(low_leaf_proof)
    fail, trace = ctx.insert_witness(
        to_bytes([old_root]), to_bytes([list(low_leaf)]), [_helpers_to_index(low_leaf_proof_helper)],
        to_bytes(low_leaf_proof), to_bytes([new_root]), to_bytes([list(new_leaf)]), [new_leaf_index],
        to_bytes(new_leaf_proof), [int(is_new_leaf_largest)], d, item_major=True, want_trace=True,
        new_path_index=[_helpers_to_index(new_leaf_proof_helper)])
    if fail[0]:
        raise ConstraintError(int(fail[0]))
    return trace


class IndexedTree:
    """Depth-d append-only indexed tree (imt_itree): leaf 0 is the {0,0,0} sentinel, insertion i
    lands on leaf `size + i`, empty slots hash to H(0,0,0) (src/indexed_merkle_tree.rs:373-376)."""

    def __init__(self, ctx, depth, capacity):
        self.ctx, self.depth, self.capacity = ctx, depth, capacity
        self.global_depth, self.index_base = depth, 0
        h = ctypes.c_void_p()
        ctx._check(lib.imt_itree_new(ctx.h, depth, capacity, ctypes.byref(h)))
        self.h = h
        ctx._children.add(self)

    def set_placement(self, global_depth, subtree_index):
        """Make this (empty) tree subtree `subtree_index` at height `depth` of a tree of depth `global_depth`:
        leaf indices crossing the API, including the hashed next_idx fields, become global
        (imt_itree_set_placement); sibling arrays get global_depth rows, filled up by lift_batch()."""
        self.ctx._check(lib.imt_itree_set_placement(self.h, global_depth, subtree_index))
        self.global_depth, self.index_base = global_depth, subtree_index << self.depth

    def lift_batch(self, res, roots_before, roots_after, item_major=False):
        """In place: the dict insert_batch() returned becomes depth-`global_depth` witnesses
        (imt_itree_lift_batch).  roots_before / roots_after: uint8 [n_subtrees, 32], every subtree's root
        before / after the step."""
        rb, ra = _arr(roots_before, (32,)), _arr(roots_after, (32,))
        n = res["new_root"].shape[0]
        out = _ffi.InsertOut(**{k: res[k].ctypes.data for k in ("old_root", "interim_root", "new_root", "low_sib",
                                                               "new_sib") if k in res})
        self.ctx._check(lib.imt_itree_lift_batch(self.h, _p(rb), _p(ra), rb.shape[0], n, ctypes.byref(out),
                                                 _ffi.SIB_ITEM_MAJOR if item_major else 0))
        return res

    def close(self):
        if getattr(self, "h", None):
            lib.imt_itree_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return lib.imt_itree_size(self.h)

    def root(self):
        out = np.empty(32, dtype=np.uint8)
        self.ctx._check(lib.imt_itree_root(self.h, _p(out), 0))
        return to_int(out)

    def insert_batch(self, vals, proofs=True, item_major=False, host_prep=False):
        """n sequential insertions (update_idx_leaf semantics); returns a dict of numpy arrays."""
        v = to_bytes(vals) if not isinstance(vals, np.ndarray) else _arr(vals, (32,))
        n, d = v.shape[0], self.global_depth      # a placed tree fills rows [0, depth); lift_batch() the rest
        res = dict(low_index=np.empty(n, np.uint64), low_leaf=np.empty((n, 3, 32), np.uint8),
                   is_largest=np.empty(n, np.uint8), old_root=np.empty((n, 32), np.uint8),
                   interim_root=np.empty((n, 32), np.uint8), new_root=np.empty((n, 32), np.uint8),
                   new_leaf=np.empty((n, 3, 32), np.uint8))
        if proofs:
            shape = (n, d, 32) if item_major else (d, n, 32)
            res["low_sib"] = np.empty(shape, np.uint8)
            res["new_sib"] = np.empty(shape, np.uint8)
        out = _ffi.InsertOut(**{k: a.ctypes.data for k, a in res.items()})
        flags = (_ffi.SIB_ITEM_MAJOR if item_major else 0) | (_ffi.HOST_PREP if host_prep else 0)
        rc = lib.imt_itree_insert_batch(self.h, _p(v), n, ctypes.byref(out), flags)
        if rc == _ffi.ERR["VALUE"]:
            raise ValueError(lib.imt_last_error(self.ctx.h).decode())
        self.ctx._check(rc)
        res["new_index"] = np.arange(self.size - n, self.size, dtype=np.uint64) + np.uint64(self.index_base)
        return res

    def get_proof_batch(self, index, item_major=False):
        idx = np.ascontiguousarray(index, dtype=np.uint64)
        out = np.empty((idx.size, self.depth, 32) if item_major else (self.depth, idx.size, 32), dtype=np.uint8)
        self.ctx._check(lib.imt_itree_get_proof_batch(self.h, _p(idx), idx.size, _p(out),
                                                      _ffi.SIB_ITEM_MAJOR if item_major else 0))
        return out

    def get_leaves(self, index):
        idx = np.ascontiguousarray(index, dtype=np.uint64)
        out = np.empty((idx.size, 3, 32), dtype=np.uint8)
        self.ctx._check(lib.imt_itree_get_leaves(self.h, _p(idx), idx.size, _p(out), 0))
        return out

    def snapshot(self, fmt=0):
        """Leaf preimages [size, 3, 32] in index order: the checkpoint of the tree (read from the device index)."""
        out = np.empty((self.size, 3, 32), dtype=np.uint8)
        self.ctx._check(lib.imt_itree_get_leaves(self.h, None, self.size, _p(out), fmt))
        return out

    def snapshot_into(self, device_ptr, fmt=0):
        """The same into device memory the caller owns ([size][3][32] bytes, 16-byte aligned): nothing crosses PCIe."""
        self.ctx._check(lib.imt_itree_get_leaves(self.h, None, self.size, ctypes.c_void_p(device_ptr), fmt | _ffi.DEVICE_PTRS))

    def load(self, preimages, fmt=0):
        """Replace the contents with a snapshot: checked (one sorted linked list from the sentinel) and rebuilt on the
        GPU; a refused snapshot leaves the tree as it was."""
        a = _arr(preimages, (3, 32))
        self._load(_p(a), a.shape[0], fmt)

    def load_device(self, device_ptr, n, fmt=0):
        """load() of n leaf preimages already in device memory (16-byte aligned)."""
        self._load(ctypes.c_void_p(device_ptr), n, fmt | _ffi.DEVICE_PTRS)

    def _load(self, ptr, n, flags):
        rc = lib.imt_itree_load(self.h, ptr, n, flags)
        if rc == _ffi.ERR["VALUE"]:
            raise ValueError(lib.imt_last_error(self.ctx.h).decode())
        self.ctx._check(rc)

    def find_low(self, vals):
        v = to_bytes(vals) if not isinstance(vals, np.ndarray) else _arr(vals, (32,))
        out = np.empty(v.shape[0], dtype=np.uint64)
        rc = lib.imt_itree_find_low_batch(self.h, _p(v), v.shape[0], _p(out), 0)
        if rc == _ffi.ERR["VALUE"]:
            raise ValueError(lib.imt_last_error(self.ctx.h).decode())
        self.ctx._check(rc)
        return out

    def non_membership_witness(self, vals, host=False, subtree_roots=None):
        """Witness for verify_non_inclusion of every value: (low index, low leaf, siblings, is_largest).
        Built on the GPU from the device-resident index; host=True uses the host mirror instead.
        A placed tree (set_placement) returns `depth` siblings against its own root; with subtree_roots (uint8
        [n_subtrees, 32], every subtree's current root) the siblings above it are appended (imt_itree_lift_batch)
        and the witness is one of depth `global_depth` against the global root."""
        if subtree_roots is not None:
            if host:
                raise ValueError("subtree_roots needs the GPU path")
            v = to_bytes(vals) if not isinstance(vals, np.ndarray) else _arr(vals, (32,))
            n = v.shape[0]
            low = np.empty(n, np.uint64)
            leaves = np.empty((n, 3, 32), np.uint8)
            largest = np.empty(n, np.uint8)
            sib = np.zeros((self.global_depth, n, 32), np.uint8)       # rows [0, depth) from the tree, the rest lifted
            rc = lib.imt_itree_non_membership_witness(self.h, _p(v), n, _p(low), _p(leaves), _p(largest), _p(sib), 0)
            if rc == _ffi.ERR["VALUE"]:
                raise ValueError(lib.imt_last_error(self.ctx.h).decode())
            self.ctx._check(rc)
            r = _arr(subtree_roots, (32,))
            out = _ffi.InsertOut(low_sib=sib.ctypes.data)
            self.ctx._check(lib.imt_itree_lift_batch(self.h, _p(r), _p(r), r.shape[0], n, ctypes.byref(out), 0))
            return low, leaves, sib, largest
        if host:
            low = self.find_low(vals)
            leaves = self.get_leaves(low)
            sib = self.get_proof_batch(low)
            largest = (leaves[:, 1, :].max(axis=1) == 0).astype(np.uint8)
            return low, leaves, sib, largest
        v = to_bytes(vals) if not isinstance(vals, np.ndarray) else _arr(vals, (32,))
        n = v.shape[0]
        low = np.empty(n, np.uint64)
        leaves = np.empty((n, 3, 32), np.uint8)
        largest = np.empty(n, np.uint8)
        sib = np.empty((self.depth, n, 32), np.uint8)
        rc = lib.imt_itree_non_membership_witness(self.h, _p(v), n, _p(low), _p(leaves), _p(largest), _p(sib), 0)
        if rc == _ffi.ERR["VALUE"]:
            raise ValueError(lib.imt_last_error(self.ctx.h).decode())
        self.ctx._check(rc)
        return low, leaves, sib, largest
