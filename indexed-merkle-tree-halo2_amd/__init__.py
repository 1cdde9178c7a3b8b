"""indexed-merkle-tree-halo2_amd: the MI355X-native hot path of
aerius-labs/indexed-merkle-tree-halo2 (Poseidon over bn256::Fr + batched depth-d Merkle
paths + batched indexed insertion), behind the reference's own interface.

The directory name has dashes, so import it through the `imt_amd` alias module at the
repository root (or importlib).  Importing requires csrc/libimt_hip.so (no fallback).
"""
from . import _ffi
from ._ffi import lib, LIB_PATH
from .api import (Context, IndexedMerkleTree, IndexedTree, ImtError, ConstraintError, verify_non_inclusion,
                  insert_leaf, to_bytes, to_int, P_MODULUS, trace_layout, rebuild_advice_column, check_vertical_gates,
                  insert_column_segments, non_inclusion_column_segments)

__all__ = ["Context", "IndexedMerkleTree", "IndexedTree", "ImtError", "ConstraintError", "verify_non_inclusion",
           "insert_leaf", "to_bytes", "to_int", "P_MODULUS", "lib", "LIB_PATH"]
