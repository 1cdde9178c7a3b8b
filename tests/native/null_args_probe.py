"""Calls every int-returning entry point of libimt_hip.so with hostile arguments -- a NULL handle; then a valid handle
with every other pointer NULL, sizes 1, depth 1, flags 0 and flags with the unknown format 3 -- and prints one line per
call.  Run by tests/test_gpu_parity.py::test_c_abi_survives_null_and_nonsense_arguments in a child process: a crash is a
missing line.  No exception or abort may cross the ABI (include/imt.h)."""
import ctypes
import sys

import imt_amd
from imt_amd import _ffi

lib = imt_amd.lib
ctx = imt_amd.Context(0)
leaves = imt_amd.to_bytes([1, 2, 3, 4])
dense = imt_amd.IndexedMerkleTree.new(ctx, leaves)
itree = imt_amd.IndexedTree(ctx, 4, 8)
SKIP = {"imt_ctx_create"}                 # takes no handle; covered by test_no_cpu_fallback
# the first argument of these is not an opaque handle (an ops table, an out-pointer, an id buffer the call WRITES): the
# generic loop would hand them a context as scratch memory; they get their own hostile calls below
SLICED = {n for n in _ffi.SIGNATURES if n.startswith(("imt_transport_", "imt_sliced_", "imt_rccl_"))}
SKIP |= SLICED
SKIP.add("imt_insert_column_segments")    # no handle at all: (depth, lookup_bits, ...)
SKIP.add("imt_less_than_lookup_rows")     # likewise: (lookup_bits, rows, cap, n_rows)
SKIP.add("imt_insert_gadget_lookup_rows")
SKIP.add("imt_non_inclusion_column_segments")


def handle_for(name):
    if name.startswith("imt_tree_") and name not in ("imt_tree_new", "imt_tree_build"):
        return dense.h
    if name.startswith("imt_itree_") and name != "imt_itree_new":
        return itree.h
    return ctx.h


def hostile(argtypes, handle, flags):
    out = [handle]
    for i, t in enumerate(argtypes[1:], start=1):
        if t is ctypes.c_void_p or hasattr(t, "contents") or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
            out.append(None)
        elif t is ctypes.c_uint and i == len(argtypes) - 1:
            out.append(flags)             # the trailing `unsigned flags`
        else:
            out.append(1)
    return out


for name, (res, args) in _ffi.SIGNATURES.items():
    if res is not ctypes.c_int or name in SKIP:
        continue
    fn = getattr(lib, name)
    for label, handle, flags in (("null-handle", None, 0), ("null-args", handle_for(name), 0), ("bad-format", handle_for(name), 3)):
        print(f"CALL {name} {label}", flush=True)
        rc = fn(*hostile(args, handle, flags))
        print(f"RC {name} {label} {rc}", flush=True)
# Every pointer a valid, device-addressable address that is NOT 16-byte aligned (8 bytes into a page-locked buffer),
# IMT_DEVICE_PTRS set: the entry points that take field elements must refuse it (IMT_ERR_ARG, include/imt.h "16-byte
# aligned") instead of handing it to a kernel.  Pointers that are not data are left out: a stream, a pointer to free.
NOT_DATA = {"imt_ctx_set_stream", "imt_host_free", "imt_itree_slice_unit", "imt_itree_slice_apply",
            "imt_itree_slice_apply_gathered", "imt_itree_batch_extract", "imt_itree_batch_end"}
buf = ctx.host_alloc(1 << 20)
odd = buf.ctypes.data + 8


FLAGS_AT = {"imt_tree_new": 3, "imt_itree_batch_begin": 3, "imt_itree_slice_prepare": 6}     # flags not the last argument


def misaligned(argtypes, handle, flags_at):
    out = [handle]
    for i, t in enumerate(argtypes[1:], start=1):
        if t is ctypes.c_void_p or hasattr(t, "contents") or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
            out.append(ctypes.cast(ctypes.c_void_p(odd), t) if t is not ctypes.c_void_p else odd)
        elif i == flags_at:
            out.append(_ffi.DEVICE_PTRS)
        else:
            out.append(1)
    return out


for name, (res, args) in _ffi.SIGNATURES.items():
    if res is not ctypes.c_int or name in SKIP or name in NOT_DATA:
        continue
    flags_at = FLAGS_AT.get(name, len(args) - 1 if len(args) and args[-1] is ctypes.c_uint else None)
    if flags_at is None:
        continue                          # no flags argument: host pointers or device-only calls covered elsewhere
    print(f"CALL {name} odd-offset", flush=True)
    rc = getattr(lib, name)(*misaligned(args, handle_for(name), flags_at))
    print(f"RC {name} odd-offset {rc}", flush=True)
# ---- imt_sliced_* / imt_transport_*: NULL everything, then valid handles with NULL / nonsense for the rest
P, vp = ctypes.POINTER, ctypes.c_void_p
calls = [
    ("imt_transport_custom_create", (None, None)), ("imt_transport_custom_create", (ctypes.byref(_ffi.TransportOps()), None)),
    ("imt_transport_local_create", (None,)), ("imt_rccl_get_unique_id", (None,)),
    ("imt_transport_rccl_create", (None, None, 1, 1, 0, None)), ("imt_transport_rccl_create", (ctx.h, None, 1, 1, 0, ctypes.byref(vp()))),
    ("imt_transport_rccl_create", (ctx.h, ctypes.create_string_buffer(128), 9, 1, 0, ctypes.byref(vp()))),
    ("imt_transport_rccl_adopt", (None, 1, None)), ("imt_transport_rccl_adopt", ((vp * 1)(None), 1, ctypes.byref(vp()))),
    ("imt_transport_ipc_create", (None, 2, 0, 32, 8, 0, None, None)), ("imt_transport_ipc_create", (ctx.h, 1, 0, 32, 8, 0, ctypes.byref(vp()), ctypes.create_string_buffer(1 << 14))),
    ("imt_transport_ipc_create", (ctx.h, 2, 5, 32, 8, 0, ctypes.byref(vp()), ctypes.create_string_buffer(1 << 14))),
    ("imt_transport_ipc_connect", (None, None)), ("imt_transport_poll_error", (None,)),
    ("imt_transport_all_gather", (None, None, None, 32, None)),
    ("imt_sliced_set_option", (None, 11, 1)),                # IMT_SLICED_OPT_RESET takes 0 only
    ("imt_sliced_create", (None, 1, 1, 0, None, 8, 0, None)), ("imt_sliced_create", ((vp * 1)(itree.h), 1, 1, 0, None, 8, 0, ctypes.byref(vp()))),
    ("imt_sliced_step", (None, None, 1, None, 0, None)), ("imt_sliced_wait", (None, 0, 0)), ("imt_sliced_flush", (None,)),
    ("imt_sliced_get_info", (None, None)),
    ("imt_insert_column_segments", (0, 18, None, 0, None)), ("imt_insert_column_segments", (32, 0, None, 0, None)),
    ("imt_insert_column_segments", (32, 18, (_ffi.ColumnSegment * 2)(), 2, None)),       # too small a table
    ("imt_non_inclusion_column_segments", (0, 18, None, 0, None)), ("imt_non_inclusion_column_segments", (32, 18, (_ffi.ColumnSegment * 2)(), 2, None)),
    ("imt_less_than_lookup_rows", (0, None, 0, None)), ("imt_less_than_lookup_rows", (29, None, 0, None)),
    ("imt_less_than_lookup_rows", (18, (ctypes.c_uint32 * 3)(), 3, None)),                # too small an array
    ("imt_insert_gadget_lookup_rows", (0, 18, None, 0, None)), ("imt_insert_gadget_lookup_rows", (32, 18, (ctypes.c_uint32 * 3)(), 3, None)),
]
for k, (name, args) in enumerate(calls):
    print(f"CALL {name} hostile-{k}", flush=True)
    rc = getattr(lib, name)(*args)
    assert rc < 0, (name, rc)
    print(f"RC {name} hostile-{k} {rc}", flush=True)
tp, w = vp(), vp()
assert lib.imt_transport_local_create(ctypes.byref(tp)) == 0
assert lib.imt_transport_ipc_connect(tp, ctypes.create_string_buffer(64)) == _ffi.ERR["ARG"]      # not an IPC transport
assert lib.imt_transport_poll_error(tp) == 0                                                       # a transport without a GPU-side wait
big = imt_amd.IndexedTree(ctx, 32, 64)
assert lib.imt_sliced_create((vp * 1)(big.h), 1, 1, 0, tp, 8, 0, ctypes.byref(w)) == 0
for k, (name, args) in enumerate((("imt_sliced_step", (w, None, 1, None, 0, None)), ("imt_sliced_step", (w, vp(odd), 1, None, _ffi.DEVICE_PTRS, None)),
                   ("imt_sliced_step", (w, vp(buf.ctypes.data), 99, None, 0, None)), ("imt_sliced_step", (w, vp(buf.ctypes.data), 1, None, 0x4000, None)),
                   ("imt_sliced_wait", (w, 3, 0)), ("imt_sliced_wait", (w, 0, 7)), ("imt_sliced_wait", (w, -1, 0)),
                   ("imt_sliced_get_info", (w, None)))):
    print(f"CALL {name} hostile-live-{k}", flush=True)
    rc = getattr(lib, name)(*args)
    assert rc < 0, (name, args, rc)
    print(f"RC {name} hostile-live-{k} {rc}", flush=True)
assert lib.imt_sliced_flush(w) == 0
lib.imt_sliced_destroy(w)
lib.imt_sliced_destroy(None)
lib.imt_transport_destroy(tp)
lib.imt_transport_destroy(None)
big.close()
ctx.sync()
# the handles still work afterwards
assert imt_amd.to_int(ctx.hash2(imt_amd.to_bytes([[1, 2]]))[0]) > 0
itree.insert_batch([5, 9])
assert itree.size == 3
print("ALIVE", flush=True)
