// imt_consts.hpp -- plain structs shared by the host table generator (imt_params.cpp)
// and the device code (imt_device.hpp).  No HIP types here.
#pragma once
#include <cstdint>

namespace imt {
namespace dev {

constexpr int NL = 9;
constexpr uint32_t MASK29 = (1u << 29) - 1;

// one field element: nine 29-bit limbs, Montgomery form with R = 2^261
struct Fe {
    uint32_t v[NL];
};

constexpr int RF = 8, RP = 57;
struct PoseidonConsts {
    Fe rc_full[RF][3];    // rounds 0..3 as generated; round 61 carries the partial rounds' leftover
    Fe rc_h2p2[3];        // rc_full[0] + (0,1,0): second permutation of a 2-input hash
    Fe k_partial[RP];     // lane-0 constants of the partial rounds
    Fe mats[2][3][3];     // [0] = MDS; [1] = matrix of full round 3 (M followed by the first N').  One
                          // array so that the round's matrix is an address, not a select over both.
    Fe sp_row[RP][3];     // sparse round: new s0 = row . (y, s1, s2)
    Fe sp_col[RP][2];     // sparse round: new s_i = s_i + col_i * y
    Fe sp_gamma[RP];      // row[p][1]*col[p-1][0] + row[p][2]*col[p-1][1]: lets round p use the linear
                          // lanes as they were BEFORE round p-1 (two rounds share one reduction each)
    Fe cap0;              // 2^64 in device form (initial capacity lane)
    Fe one;               // 1 in device form
    Fe from_canon;        // R^2: canonical integer -> device form
    Fe from_mont256;      // R^2 / 2^256: halo2curves Montgomery form -> device form
    Fe to_mont256;        // 2^256 as integer: device form -> halo2curves Montgomery form
    Fe int_one;           // integer 1: device form -> canonical integer
    Fe zero_leaf;         // H(0,0,0) in device form
};

// ---- f1: tables of the witness-trace kernel (imt_trace_device.hpp) -------------------------------
// halo2-base's OptimizedPoseidonSpec (= pse-poseidon's Spec): constants are added AFTER the S-box, the
// full-round matrix is M except for the last round of the first half (pre_sparse_mds), every partial round
// has its own sparse matrix {row, col_hat}.  Eight full rounds as one table: round f adds full_c[f] after
// its S-boxes and multiplies by mats[f == RF/2 - 1].
struct TraceConsts {
    Fe absorb[3][3];      // constants of absorb_with_pre_constants for 2 / 1 / 0 inputs: start[0] with the
                          // padding 1 folded into the first free lane ([0] = start[0] itself)
    Fe full_c[RF][3];     // start[1..4], end[0..2], 0
    Fe partial[RP];
    Fe mats[2][3][3];     // [0] = mds, [1] = pre_sparse_mds
    Fe row[RP][3];
    Fe col_hat[RP][2];
};
constexpr int TRACE_ROWS_H2 = 1208, TRACE_ROWS_H3 = 1209;   // witnesses per 2- / 3-input hash

}  // namespace dev
}  // namespace imt
