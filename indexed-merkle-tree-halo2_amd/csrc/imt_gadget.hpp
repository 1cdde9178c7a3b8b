// imt_gadget.hpp -- launchers of imt_gadget.hip (f3: the advice values of insert_leaf outside its hashes).  All inputs
// and outputs are CANONICAL 32-byte little-endian integers; the C entries (imt_capi.cpp) convert around the launch.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdint>

namespace imt {
namespace launch {

// rows of is_less_than(a_q, a_r, b_q, b_r) for n pairs; row r of item i at trace + r * row_stride + i * item_stride
void less_than_trace(hipStream_t s, const uint8_t* a, const uint8_t* b, size_t n, unsigned lookup_bits, uint8_t* trace,
                     uint64_t row_stride, uint64_t item_stride, uint8_t* lt_out);
// the glue rows of insert_leaf; pairs = the (left, right) inputs of every path hash of its four chains,
// [4][depth][n][2][32] canonical (launch::path_pairs, converted)
void insert_gadget(hipStream_t s, const uint8_t* low_leaf, const uint64_t* low_index, const uint8_t* new_leaf,
                   const uint64_t* new_path_index, const uint8_t* is_largest, const uint8_t* pairs, unsigned depth,
                   unsigned lookup_bits, size_t n, uint8_t* trace, uint64_t row_stride, uint64_t item_stride);
// the glue rows of ONE verify_non_inclusion (the first 17 + 2 K + 4 depth of the above); new_val [n][32]; pairs =
// [depth][n][2][32] of the low leaf's chain alone
void non_inclusion_gadget(hipStream_t s, const uint8_t* low_leaf, const uint64_t* low_index, const uint8_t* new_val,
                          const uint8_t* is_largest, const uint8_t* pairs, unsigned depth, unsigned lookup_bits, size_t n,
                          uint8_t* trace, uint64_t row_stride, uint64_t item_stride);

}  // namespace launch
}  // namespace imt
