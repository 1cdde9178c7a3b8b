"""ctypes binding of libimt_hip.so (the C ABI in include/imt.h).

The library is the product; there is no Python or CPU fallback.  If the shared object is
missing or cannot be loaded this module raises at import time, and creating a context
without a usable MI355X raises ImtError(IMT_ERR_NO_DEVICE).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IMT_LIB_PATH") or os.path.join(_HERE, "csrc", "libimt_hip.so")   # override: tuning builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
        "(or __graft_entry__.build()); there is no fallback implementation")



def _preload_torch_hip_runtime():
    """A process can hold one HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so.N
    (same soname as /opt/rocm's, which libimt_hip.so links against); whichever is loaded first serves
    both.  If this library came first, a later `import torch` would run on a runtime it was not built
    with and report "No HIP GPUs are available".  So when torch is installed, its runtime is loaded
    first -- without importing torch itself -- and libimt_hip.so binds to it, which is the order
    bench.py and the GPU tests have always run in.  The same holds for RCCL (librccl.so.1: the library
    calls ncclAllGather itself, imt_sliced_rccl.cpp): one copy per process, torch's when torch is there,
    so that torch.distributed and this library never run two RCCL builds over one HIP runtime."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libamdhip64.so",):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


if not os.environ.get("IMT_NO_TORCH_RUNTIME_PRELOAD"):
    _preload_torch_hip_runtime()
lib = ctypes.CDLL(LIB_PATH)

c_void_p, c_size_t, c_uint, c_int, c_u64 = (ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int,
                                            ctypes.c_uint64)
P = ctypes.POINTER


class TraceCell(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint8), ("gate", ctypes.c_uint8), ("region", ctypes.c_uint16), ("index", ctypes.c_uint32)]


class TransportOps(ctypes.Structure):
    ALL_GATHER = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_size_t, ctypes.c_void_p)
    DESTROY = ctypes.CFUNCTYPE(None, ctypes.c_void_p)
    _fields_ = [("self", ctypes.c_void_p), ("all_gather", ALL_GATHER), ("destroy", DESTROY)]


class SlicedInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("world", "n_local", "lag", "period", "gathers_per_round", "round_ticks",
                                            "rounds_in_flight")] + \
               [("payload_bytes", ctypes.c_size_t)] + [(n, ctypes.c_uint64) for n in ("rounds", "collectives", "bytes_gathered")] + \
               [(n, ctypes.c_double) for n in ("host_issue_ms", "host_wait_ms")] + \
               [(n, ctypes.c_int) for n in ("placement", "hw_queues", "comm_streams", "streams_recreated")] + \
               [("queue_map", (ctypes.c_int * 4) * 3), ("pools", ctypes.c_int)]


class ColumnSegment(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("arity", ctypes.c_uint32), ("first_row", ctypes.c_uint64), ("n_rows", ctypes.c_uint64)]


class InsertOut(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("low_index", "low_leaf", "is_largest", "old_root", "interim_root",
                                        "new_root", "new_leaf", "low_sib", "new_sib")]


# every symbol include/imt.h declares, with its signature
SIGNATURES = {
    "imt_version": (ctypes.c_char_p, []),
    "imt_ctx_create": (c_int, [c_int, P(c_void_p)]),
    "imt_ctx_destroy": (None, [c_void_p]),
    "imt_last_error": (ctypes.c_char_p, [c_void_p]),
    "imt_ctx_set_stream": (c_int, [c_void_p, c_void_p]),
    "imt_host_alloc": (c_int, [c_void_p, c_size_t, P(c_void_p)]),
    "imt_host_free": (c_int, [c_void_p, c_void_p]),
    "imt_ctx_sync": (c_int, [c_void_p]),
    "imt_ctx_set_option": (c_int, [c_void_p, c_int, c_u64]),
    "imt_measure_mad_peak": (c_int, [c_void_p, P(ctypes.c_double)]),
    "imt_profile_enable": (c_int, [c_void_p, c_int]),
    "imt_profile_read": (c_int, [c_void_p, P(ctypes.c_double)]),
    "imt_hash2_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_uint]),
    "imt_hash3_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_uint]),
    "imt_permute_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_uint]),
    "imt_hash_trace_rows": (c_size_t, [c_int]),
    "imt_hash_trace_batch": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p, c_uint]),
    "imt_path_trace_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint, c_size_t, c_void_p, c_void_p,
                                     c_uint]),
    "imt_insert_trace_rows": (c_size_t, [c_uint]),
    "imt_insert_trace_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint,
                                       c_size_t, c_void_p, c_uint]),
    "imt_hash_trace_layout": (c_int, [c_void_p, c_int, P(TraceCell), c_size_t, P(c_size_t), c_void_p, c_size_t,
                                      P(c_size_t), P(ctypes.c_uint32), c_uint]),
    "imt_less_than_trace_rows": (c_size_t, [c_uint]),
    "imt_less_than_trace_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_uint, c_void_p, c_void_p, c_uint]),
    "imt_less_than_trace_layout": (c_int, [c_void_p, c_uint, P(TraceCell), c_size_t, P(c_size_t), c_void_p, c_size_t,
                                           P(c_size_t), P(ctypes.c_uint32), c_uint]),
    "imt_less_than_lookup_rows": (c_int, [c_uint, P(ctypes.c_uint32), c_size_t, P(c_size_t)]),
    "imt_insert_gadget_rows": (c_size_t, [c_uint, c_uint]),
    "imt_insert_gadget_lookup_rows": (c_int, [c_uint, c_uint, P(ctypes.c_uint32), c_size_t, P(c_size_t)]),
    "imt_insert_gadget_trace_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_uint, c_uint, c_size_t, c_void_p, c_uint]),
    "imt_insert_column_segments": (c_int, [c_uint, c_uint, P(ColumnSegment), c_size_t, P(c_size_t)]),
    "imt_non_inclusion_gadget_rows": (c_size_t, [c_uint, c_uint]),
    "imt_non_inclusion_gadget_trace_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint, c_uint,
                                                     c_size_t, c_void_p, c_uint]),
    "imt_non_inclusion_column_segments": (c_int, [c_uint, c_uint, P(ColumnSegment), c_size_t, P(c_size_t)]),
    "imt_tree_new": (c_int, [c_void_p, c_void_p, c_size_t, c_uint, P(c_void_p)]),
    "imt_tree_free": (None, [c_void_p]),
    "imt_tree_num_levels": (c_size_t, [c_void_p]),
    "imt_tree_get_root": (c_int, [c_void_p, c_void_p, c_uint]),
    "imt_tree_get_proof": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_uint]),
    "imt_tree_get_proof_batch": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_uint]),
    "imt_tree_get_level": (c_int, [c_void_p, c_size_t, c_void_p, P(c_size_t), c_uint]),
    "imt_tree_build": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_uint]),
    "imt_path_root_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint, c_size_t, c_void_p, c_uint]),
    "imt_compute_merkle_root_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint, c_size_t, c_void_p,
                                              c_uint]),
    "imt_verify_proof_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint, c_size_t, c_void_p,
                                       c_uint]),
    "imt_non_membership_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint, c_void_p,
                                         c_void_p, c_size_t, c_void_p, c_void_p, c_uint]),
    "imt_split128_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_uint]),
    "imt_insert_witness_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_uint, c_size_t, c_void_p, c_void_p,
                                         c_uint]),
    "imt_itree_new": (c_int, [c_void_p, c_uint, c_u64, P(c_void_p)]),
    "imt_itree_free": (None, [c_void_p]),
    "imt_itree_size": (c_u64, [c_void_p]),
    "imt_itree_root": (c_int, [c_void_p, c_void_p, c_uint]),
    "imt_itree_root_lagged": (c_int, [c_void_p, c_uint, c_void_p, c_uint]),
    "imt_itree_insert_batch": (c_int, [c_void_p, c_void_p, c_size_t, P(InsertOut), c_uint]),
    "imt_itree_get_proof_batch": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_uint]),
    "imt_itree_get_leaves": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_uint]),
    "imt_itree_non_membership_witness": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_uint]),
    "imt_itree_load": (c_int, [c_void_p, c_void_p, c_u64, c_uint]),
    "imt_itree_find_low_batch": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_uint]),
    "imt_itree_batch_begin": (c_int, [c_void_p, c_void_p, c_size_t, c_uint, P(ctypes.c_uint32), P(ctypes.c_uint32)]),
    "imt_itree_batch_leaves": (c_int, [c_void_p, c_void_p, ctypes.c_uint32, ctypes.c_uint32]),
    "imt_itree_batch_level": (c_int, [c_void_p, c_uint, c_void_p, c_void_p, ctypes.c_uint32, ctypes.c_uint32]),
    "imt_itree_batch_top": (c_int, [c_void_p, c_void_p, ctypes.c_uint32, ctypes.c_uint32, c_void_p, c_void_p]),
    "imt_itree_batch_extract": (c_int, [c_void_p, P(c_void_p), c_void_p, ctypes.c_uint32, ctypes.c_uint32, P(InsertOut),
                                        c_uint]),
    "imt_itree_batch_end": (c_int, [c_void_p, P(c_void_p), c_void_p]),
    "imt_itree_batch_abort": (c_int, [c_void_p]),
    "imt_itree_slice_payload_bytes": (c_size_t, [c_size_t]),
    "imt_itree_slice_unit_bytes": (c_size_t, [c_void_p, c_u64, c_size_t, c_uint]),
    "imt_itree_slice_prepare": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, P(InsertOut), c_uint, P(c_int),
                                        P(ctypes.c_uint32)]),
    "imt_itree_slice_unit": (c_int, [c_void_p, c_int, c_uint, c_void_p, c_void_p]),
    "imt_itree_slice_apply": (c_int, [c_void_p, c_u64, c_size_t, c_uint, c_void_p, c_void_p]),
    "imt_itree_slice_apply_gathered": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, P(c_u64), P(c_u64),
                                               P(ctypes.c_int32), c_void_p]),
    "imt_transport_custom_create": (c_int, [P(TransportOps), P(c_void_p)]),
    "imt_transport_local_create": (c_int, [P(c_void_p)]),
    "imt_rccl_get_unique_id": (c_int, [c_void_p]),
    "imt_transport_rccl_create": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, P(c_void_p)]),
    "imt_transport_rccl_adopt": (c_int, [P(c_void_p), c_int, P(c_void_p)]),
    "imt_rccl_library": (ctypes.c_char_p, [P(c_int)]),
    "imt_transport_ipc_blob_bytes": (c_size_t, []),
    "imt_transport_ipc_create": (c_int, [c_void_p, c_int, c_int, c_uint, c_size_t, c_int, P(c_void_p), c_void_p]),
    "imt_transport_ipc_connect": (c_int, [c_void_p, c_void_p]),
    "imt_transport_destroy": (c_int, [c_void_p]),
    "imt_transport_set_option": (c_int, [c_void_p, c_int, ctypes.c_long]),
    "imt_transport_all_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "imt_transport_poll_error": (c_int, [c_void_p]),
    "imt_sliced_set_option": (c_int, [c_void_p, c_int, ctypes.c_long]),
    "imt_sliced_dump": (c_int, [c_void_p, ctypes.c_char_p, c_size_t]),
    "imt_transport_last_error": (ctypes.c_char_p, [c_void_p]),
    "imt_sliced_create": (c_int, [P(c_void_p), c_int, c_int, c_int, c_void_p, c_size_t, c_int, P(c_void_p)]),
    "imt_sliced_step": (c_int, [c_void_p, c_void_p, c_size_t, P(InsertOut), c_uint, P(c_u64)]),
    "imt_sliced_wait": (c_int, [c_void_p, c_int, c_u64]),
    "imt_sliced_flush": (c_int, [c_void_p]),
    "imt_sliced_get_info": (c_int, [c_void_p, P(SlicedInfo)]),
    "imt_sliced_last_error": (ctypes.c_char_p, [c_void_p]),
    "imt_sliced_destroy": (None, [c_void_p]),
    "imt_itree_set_placement": (c_int, [c_void_p, c_uint, c_u64]),
    "imt_itree_set_value_partition": (c_int, [c_void_p, ctypes.c_uint32, ctypes.c_uint32]),
    "imt_itree_lift_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, P(InsertOut), c_uint]),
    "imt_combine_subtree_roots": (c_int, [c_void_p, c_void_p, c_size_t, c_uint, c_uint, c_void_p, c_uint]),
    "imt_zero_hashes": (c_int, [c_void_p, c_uint, c_void_p, c_uint]),
}

for _name, (_res, _args) in SIGNATURES.items():
    if os.environ.get("IMT_LIB_PATH") and not hasattr(lib, _name):
        continue                  # a tuning build of an older revision (explicit override only)
    _fn = getattr(lib, _name)     # AttributeError here = the library does not export the symbol
    _fn.restype = _res
    _fn.argtypes = _args

# error codes / flags of include/imt.h
IMT_OK = 0
ERR = dict(NO_LEAVES=-1, ODD_LEAVES=-2, NOT_POW2=-3, RANGE=-4, NONCANONICAL=-5, ALLOC=-6, NO_DEVICE=-7, HIP=-8,
           ARG=-9, VALUE=-10, FULL=-11, INTERNAL=-12, TIMEOUT=-13)
FMT_CANONICAL, FMT_MONT256, FMT_DEVICE = 0, 1, 2
DEVICE_PTRS, SIB_ITEM_MAJOR, ROOT_PER_ITEM, PIPELINE, HOST_PREP, INPUTS_READY = 0x10, 0x20, 0x40, 0x80, 0x100, 0x200
F_RANGE_PRED, F_LOW_IN_ROOT, F_LOW_LT_NEW, F_ZERO_SLOT, F_NEXT_VAL, F_NEXT_IDX, F_NEW_ROOT, F_BAD_BIT = (
    0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80)
CELL_CONST, CELL_INPUT, CELL_INIT, CELL_WITNESS, CELL_COPY = 0, 1, 2, 3, 4
TRACE_ITEM_MAJOR = SIB_ITEM_MAJOR
OPT_COOP_MAX_EVENTS = 1
SEG_GLUE, SEG_HASH = 0, 1
SLICED_ROUNDS = 4
(SLICED_OPT_COMM_STREAMS, SLICED_OPT_COMM_PRIORITY, SLICED_OPT_ROUND_PRIORITIES, SLICED_OPT_APPLY_STREAMS, SLICED_OPT_PREP_STREAM,
 SLICED_OPT_VERIFY_QUEUES, SLICED_OPT_WATCHDOG_MS, SLICED_OPT_TIMING, SLICED_OPT_COMM_PLACEMENT, SLICED_OPT_POOLS,
 SLICED_OPT_RESET) = range(1, 12)
TRANSPORT_OPT_TIMEOUT_MS, TRANSPORT_OPT_HOST_POLL = 1, 2
PLACEMENT = {0: "unverified", 1: "as created", 2: "repaired", 3: "degraded"}
RCCL_UNIQUE_ID_BYTES = 128
