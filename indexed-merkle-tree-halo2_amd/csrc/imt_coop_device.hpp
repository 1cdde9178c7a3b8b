// imt_coop_device.hpp -- the LATENCY form of the hash: three lanes of a quad per hash, one state lane each.
//
// Why: a batch insertion is 33 dependent launches (leaf level + 32 tree levels) and with one thread per hash a launch
// cannot take less than one hash's dependent instruction chain, 186 k instructions = 0.39 ms, however few events it
// carries (profiles/r02_small_batch_rates.txt: 13 ms per batch from 2^8 to 2^15 insertions).  For small batches the
// chip is empty, so lanes are free: here the three state lanes of the permutation live in three lanes of a DPP quad
// and every round costs the critical lane
//   full round     3 products (x^2, x^4, x^5) + one 3-term row                       (was 9 products + 3 rows)
//   partial round  2 products + ONE fused product: with y = x^5 = x^4 (x), every product of the linear layer that
//                  involves y is regrouped as x^4 (constant x), and "constant x" does not wait for the S-box: step 1
//                  lane 0 squares x while lanes 1, 2 form col_i x and the FOURTH lane row_0 x; step 2 lane 0 squares
//                  again while lanes 1, 2 multiply their row entries into their state lanes; step 3 all lanes run the
//                  same "x^4 (.) + addend": lane 0 finishes the row, lanes 1, 2 take their column update
//                                                                                        (was 8 products)
// i.e. 4 dependent products per full round and 3 per partial round.
// ~54 k instructions per permutation on every lane instead of 92 k on one: the hash takes 0.6x the time and 2.3x the
// lane-instructions, so it is used only while the launch fits one wave per SIMD (imt_launch: coop_max_events).
// Same values as hash23 (plain-form-identical schedule; all products with 29-bit quotient digits so that the linear
// lanes, which only accumulate for 57 rounds, stay below 2^261: s_i grows by < p per round, the S-box lane is reset by
// every squaring).
//
// A lane's "constant" is its neighbour's "state", so no factor can be an SGPR operand here: the tables are copied from
// __constant__ memory into LDS once per block and every lane reads the entry its role needs (lanes of one role read
// one address: a broadcast, no bank conflicts); cross-lane traffic is DPP quad_perm moves (one VALU instruction per
// limb).  Device-only (DPP has no host form); checked against the oracle through every small-batch GPU test.
#pragma once
#if !defined(__HIP_DEVICE_COMPILE__)
#error "device-only header"
#endif
#include <cstddef>
#include "imt_device.hpp"

namespace imt {
namespace dev {
namespace coop {

constexpr unsigned TAB_DWORDS = sizeof(PoseidonConsts) / 4;     // the whole struct is an array of Fe
#define IMT_COOP_E(member) ((unsigned)(offsetof(PoseidonConsts, member) / sizeof(Fe)))

__device__ __forceinline__ void tab_fill(uint32_t* tab, const PoseidonConsts& pc) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&pc);
    for (unsigned i = threadIdx.x; i < TAB_DWORDS; i += blockDim.x) tab[i] = src[i];
    __syncthreads();
}
__device__ __forceinline__ void tab_fe(Fe& r, const uint32_t* tab, unsigned entry) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = tab[entry * NL + i];
}
// every lane of a quad receives lane J's value
template <int J>
__device__ __forceinline__ void quad_bcast(Fe& r, const Fe& x) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.v[i], J * 0x55, 0xf, 0xf, true);
}
__device__ __forceinline__ void sel(Fe& r, bool c, const Fe& a, const Fe& b) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = c ? a.v[i] : b.v[i];
}

// One permutation; S = this lane's state lane (normalised limbs), ri = 0, 1, 2 (the fourth lane of a quad shadows
// lane 0).  first_rc = table entry of the round-0 constants (rc_full[0], or rc_h2p2 for the padding permutation).
__device__ __forceinline__ void permute(const uint32_t* tab, Fe& S, unsigned ri, unsigned first_rc) {
    const bool is0 = ri == 0;
#pragma unroll 1
    for (int st = 0; st < RF + RP; st++) {
        if (st < RF / 2 || st >= RF / 2 + RP) {
            const int f = st < RF / 2 ? st : st - RP;
            Fe k, v, x2, x4, y, Y[3], M[3];
            tab_fe(k, tab, (st == 0 ? first_rc : IMT_COOP_E(rc_full) + 3u * (unsigned)f) + ri);
            add_lazy(v, S, k);
            masm::sqr_v_narrow(x2, v);
            masm::sqr_v_narrow(x4, x2);
            masm::mul_vv_narrow(y, &x4, &v);
            quad_bcast<0>(Y[0], y);
            quad_bcast<1>(Y[1], y);
            quad_bcast<2>(Y[2], y);
            const unsigned row = IMT_COOP_E(mats) + ((f == RF / 2 - 1 ? 3u : 0u) + ri) * 3u;
            tab_fe(M[0], tab, row);
            tab_fe(M[1], tab, row + 1);
            tab_fe(M[2], tab, row + 2);
            masm::dot3_vv_narrow(S, M, Y);
        } else {
            // s0' = row0 y + row1 s1 + row2 s2 and s_i' = s_i + col_i y with y = x^5, regrouped as x^4 (row0 x) and
            // x^4 (col_i x): the products with x do not wait for the S-box, so the round is THREE dependent products
            // (x^2 | col_i x | row0 x;  x^4 | row_i s_i;  x^4 (.) + addend) instead of four, the fourth lane of the
            // quad taking row0 x.  Lane 3 otherwise shadows lane 0 (same constants, same state), also through here.
            const unsigned p = (unsigned)(st - RF / 2);
            const bool lane0 = (threadIdx.x & 3u) == 0u;        // is0 is true on lanes 0 AND 3
            Fe k, v, X, a1, e1, a2, b2, e2, X4, T, U1, U2, yf, add;
            tab_fe(k, tab, IMT_COOP_E(k_partial) + p);
            add_lazy(v, S, k);                                   // x = s0 + k (lanes 0, 3)
            quad_bcast<0>(X, v);
            tab_fe(a1, tab, is0 ? IMT_COOP_E(sp_row) + 3u * p : IMT_COOP_E(sp_col) + 2u * p + (ri - 1u));
            sel(a1, lane0, X, a1);
            masm::mul_vv_narrow(e1, &a1, &X);                    // lane 0: x^2   lane i: col[i] x   lane 3: row[0] x
            tab_fe(a2, tab, IMT_COOP_E(sp_row) + 3u * p + ri);
            sel(a2, lane0, e1, a2);
            sel(b2, lane0, e1, S);
            masm::mul_vv_narrow(e2, &a2, &b2);                   // lane 0: x^4   lane i: row[i] s_i   (lane 3: unused)
            quad_bcast<0>(X4, e2);
            quad_bcast<3>(T, e1);
            quad_bcast<1>(U1, e2);
            quad_bcast<2>(U2, e2);
            add_lazy(add, U1, U2);
            sel(yf, is0, T, e1);                                 // lanes 0, 3: row[0] x   lane i: col[i] x
            sel(add, is0, add, S);                               // lanes 0, 3: the rest of the row   lane i: s_i
            masm::mul_vv_add_narrow(S, &X4, &yf, add);           // lanes 0, 3: new s_0   lane i: s_i + col[i] x^5
        }
    }
}

// X: this lane's input (lane 1: first, lane 2: second); C3: the third input (read on lane 1, for `three`).
// Returns the hash on lane 1 (canonical); other lanes return their own state lane.
__device__ __forceinline__ void hash23(const uint32_t* tab, Fe& out, const Fe& X, const Fe& C3, bool three, unsigned ri) {
    Fe S, cap;
    tab_fe(cap, tab, IMT_COOP_E(cap0));
    sel(S, ri == 0, cap, X);
    permute(tab, S, ri, IMT_COOP_E(rc_full));
    unsigned first_rc = IMT_COOP_E(rc_h2p2);                     // 2 inputs: the padding 1 rides on the constants
    if (three) {                                                 // absorb [c, 1] into lanes 1, 2
        Fe one, z;
        tab_fe(one, tab, IMT_COOP_E(one));
#pragma unroll
        for (int i = 0; i < NL; i++) z.v[i] = ri == 1 ? C3.v[i] : (ri == 2 ? one.v[i] : 0u);
        add_lazy(S, S, z);
        normalize(S);
        first_rc = IMT_COOP_E(rc_full);
    }
    permute(tab, S, ri, first_rc);
    out = S;
    canonicalize(out);
}

}  // namespace coop
}  // namespace dev
}  // namespace imt
