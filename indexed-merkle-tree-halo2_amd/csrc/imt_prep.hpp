// imt_prep.hpp -- GPU-side preparation of a batch insertion ("GPU low-leaf
// search"): everything imt_itree.cpp otherwise does on the host between receiving the values and
// launching the hash sweep -- the low-leaf search of update_idx_leaf
// (/root/reference/src/indexed_merkle_tree.rs:639-658), the event preimages and the
// (position, time) order of the events -- as device kernels over a device-resident index.
//
// Device index of a tree: val[cap][32] (canonical integers, leaf order) and sorted[size] (leaf
// indices in ascending value order).  Leaf indices ARE insertion times, so "the low leaf of v at
// the time it is inserted" is the nearest element to the left of v, in value order, with a
// smaller leaf index: an all-nearest-smaller-values problem, solved per batch with a sparse
// min-table over the batch in value order plus the gap (number of stored values below) of each
// new value.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdint>

namespace imt {
namespace prep {

constexpr uint32_t NONE = 0xffffffffu;
// error bits written by the kernels
constexpr int ERR_NONCANONICAL = 1, ERR_ZERO = 2, ERR_DUPLICATE = 4, ERR_FOREIGN = 8;
// 16 / 32 / 64 / 128: LOAD_SENTINEL / LOAD_LINK / LOAD_LAST / LOAD_TIE of the snapshot check (imt_prep_logic.hpp)
constexpr int ERR_RANGE = 256;

struct Workspace {            // per plan set, sized for `cap_n` insertions
    size_t cap_n = 0;
    uint32_t part_mod = 0, part_res = 0;   // value partition: accept only v % part_mod == part_res (0/1 = everything)
    uint32_t* iota = nullptr;      // [n]   M+i
    uint32_t* bsorted = nullptr;   // [n]   new leaf indices in value order
    uint32_t* gap = nullptr;       // [n]   stored values below each new value
    uint32_t* st = nullptr;        // [levels][n] sparse min-table of insertion times
    uint32_t* low = nullptr;       // [n]   by insertion time
    uint32_t* succ = nullptr;      // [n]
    uint64_t* keys = nullptr;      // [2n]  (position << 32) | event
    uint64_t* keys_sorted = nullptr;
    void* tmp = nullptr;           // rocPRIM temporary storage
    size_t tmp_bytes = 0;
    int* err = nullptr;            // device word
    // hash-free outputs kept for imt_itree_batch_extract (sharded mode)
    uint64_t* o_low = nullptr;     // [n]
    uint8_t* o_largest = nullptr;  // [n]
    uint8_t* o_lowleaf = nullptr;  // [n][96]
    uint8_t* o_newleaf = nullptr;  // [n][96]
};

size_t temp_bytes_needed(size_t n, size_t max_size);

// Everything up to (and including) the level-0 tables and the new sorted index.
//   vals      [n][32] canonical, device
//   d_val     device index values; rows [M, M+n) are written
//   sorted_old[M] -> sorted_new[M+n]
//   pre       [2n][96] event preimages (canonical)
//   node/time/rs/re  level-0 tables [2n]
//   o_* user outputs (device pointers, any may be NULL): low_index u64[n], is_largest u8[n],
//        low_leaf / new_leaf [n][3][32] canonical
//   base      added to every leaf index that leaves the tree: the next_idx fields of the preimages and
//             o_low_index (a tree placed as a subtree of a deeper one, imt_itree_set_placement)
// Errors are OR-ed into ws.err (ERR_*); nothing outside rows [M, M+n) of d_val, sorted_new and the
// workspace is written, so a failed batch leaves the tree untouched.  Returns the first failure of a
// rocPRIM call or kernel launch (hipErrorInvalidValue if the workspace is too small for (n, M)).
// All 32-byte rows (vals, d_val, pre, o_*_leaf) must be 16-byte aligned: they move as two 16-byte words.
hipError_t run(hipStream_t s, Workspace& ws, const uint8_t* vals, uint8_t* d_val, const uint32_t* sorted_old,
               uint32_t* sorted_new, uint32_t M, uint32_t n, uint64_t base, uint8_t* pre, uint32_t* node, uint32_t* time,
               uint32_t* rs, uint32_t* re, uint64_t* o_low_index, uint8_t* o_is_largest, uint8_t* o_low_leaf,
               uint8_t* o_new_leaf);

// The index part of run() alone -- values into rows [M, M+n) of d_val, the checks (non-canonical, zero, duplicate),
// sorted_old[M] -> sorted_new[M+n] -- for values whose hashing is another GPU's share (imt_itree_slice_prepare): needs
// only iota / bsorted / gap / st[n] / tmp / err of the workspace.
hipError_t index_only(hipStream_t s, Workspace& ws, const uint8_t* vals, uint8_t* d_val, const uint32_t* sorted_old,
                      uint32_t* sorted_new, uint32_t M, uint32_t n);

// non-membership witness: low leaf index, its preimage {val, next_val, next_idx} (canonical) and the
// is_largest flag of every candidate; any output may be NULL.  part_mod > 1: candidates with v % part_mod != part_res
// belong to another subtree's list (ERR_FOREIGN)
void nm_witness(hipStream_t s, const uint8_t* vals, const uint8_t* d_val, const uint32_t* sorted, uint32_t M, uint32_t n,
                uint64_t base, uint32_t part_mod, uint32_t part_res, uint64_t* low_index, uint8_t* low_leaf,
                uint8_t* is_largest, int* err);

// predecessor search only (imt_itree_find_low_batch): low[i] = leaf index of the greatest value < vals[i]
void find_low(hipStream_t s, const uint8_t* vals, const uint8_t* d_val, const uint32_t* sorted, uint32_t M, uint32_t n,
              uint64_t base, uint32_t part_mod, uint32_t part_res, uint64_t* low_index, int* err);

// ---- snapshot (imt_itree_load, imt_itree_get_leaves): the tree's list checked and read on the device ----
// load_check: pre = [n][3][32] canonical leaf preimages in index order.  Orders the leaves by val (radix sort on the top
// 64 bits; full_sort: merge sort with the 256-bit comparator -- the caller's second attempt after LOAD_TIE) and checks
// every rank (load_check_rank, imt_prep_logic.hpp); error bits are OR-ed into *err, the smallest leaf index with a broken
// link into **bad (0xffffffff if none).  *sorted = the leaf indices in value order, inside `ws` (load_ws_bytes(n) bytes of
// device memory).  Writes nothing but ws / err.
size_t load_ws_bytes(size_t n);
hipError_t load_check(hipStream_t s, const uint8_t* pre, uint32_t n, uint64_t base, uint32_t part_mod, uint32_t part_res,
                      void* ws, size_t ws_bytes, bool full_sort, int* err, uint32_t** bad, const uint32_t** sorted);
// d_val[i] = pre[i].val
void load_commit(hipStream_t s, const uint8_t* pre, uint32_t n, uint8_t* d_val);
// out[i] = canonical preimage of leaf index[i] (index == NULL: first + i), all-zero for an empty slot below `cap`;
// ERR_RANGE for an index outside [base, base + cap)
void leaves(hipStream_t s, const uint64_t* index, uint64_t first, uint32_t n, const uint8_t* d_val, const uint32_t* sorted,
            uint32_t M, uint64_t cap, uint64_t base, uint8_t* out, int* err);

}  // namespace prep
}  // namespace imt
