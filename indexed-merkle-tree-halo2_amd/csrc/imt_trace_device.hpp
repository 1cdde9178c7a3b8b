// imt_trace_device.hpp -- f1: the witness trace of halo2-base's
// PoseidonHasher::hash_fix_len_array, every NEW advice value in assignment order, so that a chip can assign
// precomputed cells where the reference recomputes each Poseidon on the CPU:
//   hasher.hash_fix_len_array(ctx, gate, &inp)   /root/reference/src/indexed_merkle_tree.rs:92 (path loop),
//                                                :194, :271-275, :299-303 (leaf hashes)
// halo2-base is un-vendored (aerius-labs/halo2-lib, branch feat/secp256k1-hash2curve, Cargo.toml:14); the order
// restated here is the published halo2-lib v0.4.x gadget (poseidon/hasher/state.rs over gates/flex_gate.rs):
//
//   permutation(inputs):
//     absorb   s0 + c                                    -> 1 row
//              per input lane: s + in, (s + in) + c      -> 2 rows          (gate.sum of three)
//              per free lane: s + c (the first one carries the padding 1)   -> 1 row
//     4 x      per lane x^2, x^4, x * x^4 + c            -> 9 rows          (sbox_full)
//              per output lane m0 s0, + m1 s1, + m2 s2   -> 9 rows          (apply_mds: gate.inner_product)
//     57 x     lane 0: x^2, x^4, x * x^4 + c             -> 3 rows          (sbox_part)
//              r0 s0, + r1 s1, + r2 s2                   -> 3 rows          (apply_sparse_mds, lane 0)
//              c1 s0 + s1 ; c2 s0 + s2                   -> 2 rows          (lanes 1, 2: gate.mul_add)
//     4 x      as the first four
//   hash2 = permutation([a, b]), permutation([])      : 5 + 600 + 3 + 600 = 1208 rows
//   hash3 = permutation([a, b]), permutation([c])     : 5 + 600 + 4 + 600 = 1209 rows
//   result = lane 1 of the last apply_mds = row (rows - 4).
//
// Parity status: UNPINNED BY THE REFERENCE (no trace vector exists in it, halo2-base cannot be built here).
// Pinned instead by: the last row equals the hash (reference KAT); every vertical gate a + b c = d of the
// reconstructed advice column holds (tests); equality with the oracle's independent restatement (oracle/trace.c).
//
// Arithmetic: same radix-2^29 / R = 2^261 domain as the hash kernels, but every intermediate is an output, so the
// lazy tricks of permute() do not apply.  All products use 29-bit quotient digits: for factors below 4p the result
// is below 16 p^2 / R + p = 1.125 p (+ the addend of the fused forms); sums are brought below ~2.1 p by one
// conditional subtraction of 2p (red2) only where they would otherwise keep growing (the linear lanes of the partial
// rounds, the absorbed lanes), so every operand stays below 4p and every stored limb below 2^29.
// The kernel is bound by its HBM writes (38.7 KB per hash) about as much as by the VALU: DESIGN.md section 6.
#pragma once
#include "imt_device.hpp"

namespace imt {
namespace dev {

// r = a * b / R + addend   (addend may be a wave-uniform constant)
IMT_HD void t_mul_add(Fe& r, const Fe& a, const Fe& b, const Fe& addend_uniform) {
#ifdef IMT_MONT_ASM
    masm::mul_vv_adds_narrow(r, &a, &b, addend_uniform);
#else
    mont_dot<1, true, false>(r, &a, &b, addend_uniform);
#endif
}
IMT_HD void t_sqr(Fe& r, const Fe& a) {
#ifdef IMT_MONT_ASM
    masm::sqr_v_narrow(r, a);
#else
    mont_dot<1, false, false>(r, &a, &a, a);
#endif
}
// c wave-uniform (a __constant__ table entry indexed by the round counter)
IMT_HD void t_mulc(Fe& r, const Fe& c_uniform, const Fe& v) {
#ifdef IMT_MONT_ASM
    masm::mul_uc_narrow(r, &c_uniform, &v);
#else
    mont_dot<1, false, false>(r, &c_uniform, &v, v);
#endif
}
IMT_HD void t_mulc_add(Fe& r, const Fe& c_uniform, const Fe& v, const Fe& addend) {
#ifdef IMT_MONT_ASM
    masm::mul_uc_add_narrow(r, &c_uniform, &v, addend);
#else
    mont_dot<1, true, false>(r, &c_uniform, &v, addend);
#endif
}

// a -= (p << SH) if a >= (p << SH), branch-free: the difference is formed with a borrow chain (3 instructions per
// limb) and selected by the final borrow, instead of compare-then-subtract under a divergent branch.
// Normalised limbs in (top limb: whatever is left), normalised limbs out.
template <int SH>
IMT_HD void csub(Fe& a) {
    uint32_t d[NL];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        const uint32_t x = a.v[i] - p29_shl<SH>(i) - borrow;
        borrow = x >> 31;                              // limbs < 2^29: bit 31 set <=> went negative
        d[i] = x & MASK29;
    }
    const uint32_t top = a.v[NL - 1] - p29_shl<SH>(NL - 1) - borrow;      // negative <=> a < (p << SH)
    const bool keep = (int32_t)top < 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) a.v[i] = keep ? a.v[i] : d[i];
    a.v[NL - 1] = keep ? a.v[NL - 1] : top;
}
IMT_HD void red2(Fe& a) { csub<1>(a); }               // a < 6p -> a < max(2p, a - 2p)

// ---- x * 2^261 (limbs normalised, value < 4p)  ->  the eight words of canonical x * 2^256: what halo2curves keeps in
// memory.  Dividing by 2^5 is one 5-bit Montgomery step: p = 1 mod 32, so m = -a mod 32 clears the low five bits of
// t = a + m p, and t / 32 is canonical once t < 32 p, i.e. after subtracting 32 p where a >= (32 - m) p: in one go,
// t = a + ms p with the SIGNED multiplier ms = m or m - 32.  Which of the two: a < 4p, so a >= (32 - m) p needs
// m >= 29, and the top limbs decide it except when they are equal (about once in 2^25 rows: then the sign of the
// exact sum decides).  Neither the sum nor the division is ever normalised on the limbs: the output words are cut out
// of a signed 64-bit running sum of a_i + ms p_i, five bits up.
IMT_HD constexpr uint32_t top_limb_of_multiple(uint32_t k) {      // limb 8 of k * p
    uint64_t acc = 0;
    for (int i = 0; i < NL; i++) acc = (acc >> 29) + (uint64_t)k * p29(i);
    return (uint32_t)acc;
}
IMT_HD void store_mont256(void* ptr, const Fe& a) {
    const uint32_t m = (0u - a.v[0]) & 31u;
    constexpr uint32_t Q1 = top_limb_of_multiple(1), Q2 = top_limb_of_multiple(2), Q3 = top_limb_of_multiple(3);
    const uint32_t q8 = m == 31u ? Q1 : m == 30u ? Q2 : m == 29u ? Q3 : 0xffffffffu;     // top limb of (32 - m) p
    bool neg = a.v[NL - 1] > q8;
    if (a.v[NL - 1] == q8) {                  // rare: the sign of a + (m - 32) p, exactly
        int64_t s = 0;
        for (int i = 0; i < NL; i++) s = (s >> 29) + (int64_t)a.v[i] + ((int64_t)m - 32) * (int64_t)p29(i);
        neg = s >= 0;
    }
    const int32_t ms = (int32_t)m - (neg ? 32 : 0);               // a + ms p lies in [0, 32 p) and is a multiple of 32
    uint32_t w[8];
    // limb j + 1 starts at bit 29 (j + 1) - 5 = 32 j + (24 - 3 j) of the output: word j is complete once limb j + 1 is in
    int64_t acc = ((int64_t)a.v[0] + (int64_t)ms * (int64_t)p29(0)) >> 5;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int sh = 24 - 3 * j;
        acc += ((int64_t)a.v[j + 1] << sh) + (int64_t)(ms * (1 << sh)) * (int64_t)p29(j + 1);      // |ms| 2^sh < 2^29
        w[j] = (uint32_t)acc;
        acc >>= 32;
    }
    Word4* q = reinterpret_cast<Word4*>(ptr);
    q[0] = Word4{w[0], w[1], w[2], w[3]};
    q[1] = Word4{w[4], w[5], w[6], w[7]};
}

struct TraceSink {
    uint8_t* p;              // next row of this hash
    uint64_t stride;         // bytes from one row to the next
};
// v: normalised limbs, value < 4p.  The output format is a template parameter all the way up to the kernel: each of the
// ~30 inlined copies of this function then holds one store path instead of three (the loop bodies shrink to what runs).
template <unsigned FMT>
IMT_HD void t_emit(const PoseidonConsts& pc, TraceSink& o, const Fe& v) {
    if (FMT == FMT_MONT256) {
        store_mont256(o.p, v);
    } else {
        Fe y;
        if (FMT == FMT_DEVICE) {
            y = v;
            csub<1>(y);
            csub<0>(y);
        } else {
#ifdef IMT_MONT_ASM
            masm::redc_v_narrow(y, v);                                 // v / R + p
#else
            mont_redc(y, v);
#endif
            csub<0>(y);
        }
        store_packed(o.p, y);
    }
    o.p += o.stride;
}

// a + b (a < 3.2p, b < p), normalised and below 2.2p
IMT_HD void t_add(Fe& r, const Fe& a, const Fe& b) {
    add_lazy(r, a, b);
    normalize(r);
    red2(r);
}

// x^5 + c with its three rows
template <unsigned FMT>
IMT_HD void t_x5c(const PoseidonConsts& pc, TraceSink& o, Fe& x, const Fe& c_uniform) {
    Fe x2, x4;
    t_sqr(x2, x);
    t_emit<FMT>(pc, o, x2);
    t_sqr(x4, x2);
    t_emit<FMT>(pc, o, x4);
    t_mul_add(x, x, x4, c_uniform);      // x < 4p, x4 < 1.04p: < 4.2 p^2 / R + c + p < 2.04p, no reduction needed
    t_emit<FMT>(pc, o, x);
}

// gate.inner_product(s, row): the three running sums are rows
template <unsigned FMT>
IMT_HD void t_inner(const PoseidonConsts& pc, TraceSink& o, Fe& r, const Fe* row_uniform, const Fe s[3]) {
    Fe acc;
    t_mulc(acc, row_uniform[0], s[0]);
    t_emit<FMT>(pc, o, acc);
    t_mulc_add(acc, row_uniform[1], s[1], acc);      // < 0.04p + 1.04p + p
    t_emit<FMT>(pc, o, acc);
    t_mulc_add(r, row_uniform[2], s[2], acc);        // < 0.04p + 2.08p + p = 3.12p: fine as a factor (< 4p) and as a row
    t_emit<FMT>(pc, o, r);
}

// One permutation.  `absorb` = tc.absorb[2 - n_in]; in0 / in1 are read only for n_in >= 1 / 2.
// Entry: lanes normalised, < 4p.
template <unsigned FMT>
IMT_HD void permute_trace(const PoseidonConsts& pc, const TraceConsts& tc, TraceSink& o, Fe s[3], int n_in,
                          const Fe& in0, const Fe& in1) {
    const Fe* ab = tc.absorb[2 - n_in];
    t_add(s[0], s[0], ab[0]);
    t_emit<FMT>(pc, o, s[0]);
    if (n_in >= 1) {
        t_add(s[1], s[1], in0);
        t_emit<FMT>(pc, o, s[1]);
    }
    t_add(s[1], s[1], ab[1]);
    t_emit<FMT>(pc, o, s[1]);
    if (n_in >= 2) {
        t_add(s[2], s[2], in1);
        t_emit<FMT>(pc, o, s[2]);
    }
    t_add(s[2], s[2], ab[2]);
    t_emit<FMT>(pc, o, s[2]);
#pragma unroll 1
    for (int st = 0; st < RF + RP; st++) {
        if (st < RF / 2 || st >= RF / 2 + RP) {
            const int f = st < RF / 2 ? st : st - RP;
            t_x5c<FMT>(pc, o, s[0], tc.full_c[f][0]);
            t_x5c<FMT>(pc, o, s[1], tc.full_c[f][1]);
            t_x5c<FMT>(pc, o, s[2], tc.full_c[f][2]);
            const Fe(*mat)[3] = tc.mats[f == RF / 2 - 1 ? 1 : 0];
            Fe n0, n1, n2;
            t_inner<FMT>(pc, o, n0, mat[0], s);
            t_inner<FMT>(pc, o, n1, mat[1], s);
            t_inner<FMT>(pc, o, n2, mat[2], s);
            s[0] = n0; s[1] = n1; s[2] = n2;
        } else {
            const int p = st - RF / 2;
            t_x5c<FMT>(pc, o, s[0], tc.partial[p]);
            Fe n0;
            t_inner<FMT>(pc, o, n0, tc.row[p], s);
            t_mulc_add(s[1], tc.col_hat[p][0], s[0], s[1]);   // gate.mul_add(s0, col_hat, s_i)
            red2(s[1]);
            t_emit<FMT>(pc, o, s[1]);
            t_mulc_add(s[2], tc.col_hat[p][1], s[0], s[2]);
            red2(s[2]);
            t_emit<FMT>(pc, o, s[2]);
            s[0] = n0;
        }
    }
}

// a, b, c: canonical device-form inputs (c only for three).  `three` must be wave-uniform.
template <unsigned FMT>
IMT_HD void hash_trace(const PoseidonConsts& pc, const TraceConsts& tc, TraceSink& o, const Fe& a, const Fe& b,
                       const Fe& c, bool three) {
    Fe s[3] = {pc.cap0, a, b};
#pragma unroll
    for (int i = 0; i < NL; i++) s[1].v[i] = s[2].v[i] = 0;     // PoseidonState::default: [2^64, 0, 0]
#pragma unroll 1
    for (int blk = 0; blk < 2; blk++) {
        const int n_in = blk == 0 ? 2 : (three ? 1 : 0);
        permute_trace<FMT>(pc, tc, o, s, n_in, blk == 0 ? a : c, b);
    }
}

}  // namespace dev
}  // namespace imt
