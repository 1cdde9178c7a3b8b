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
ctx.sync()
# the handles still work afterwards
assert imt_amd.to_int(ctx.hash2(imt_amd.to_bytes([[1, 2]]))[0]) > 0
itree.insert_batch([5, 9])
assert itree.size == 3
print("ALIVE", flush=True)
