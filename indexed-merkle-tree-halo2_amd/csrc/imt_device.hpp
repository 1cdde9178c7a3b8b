// imt_device.hpp -- gfx950 device code: bn256::Fr in radix 2^29 and the Poseidon
// T=3 / R_F=8 / R_P=57 permutation + rate-2 sponge used by every kernel.
//
// Replaces, on the device, the arithmetic behind the reference call sites
//   hash.update(..); hash.squeeze_and_reset()   /root/reference/src/utils.rs:46-47,96-100
//   hasher.hash_fix_len_array(ctx, gate, &inp)  /root/reference/src/indexed_merkle_tree.rs:92,194,271-275,299-303
// (pse-poseidon / halo2-base, un-vendored; algorithm in SURVEY.md sec. A).
//
// Why radix 2^29 (measured on MI355X, tools/microbench/valu_rates.hip,
// profiles/r01_valu_rates.txt): v_mad_u64_u32 issues at the same ~4 cycles per
// wave64 as a 32-bit add-with-carry, so the cheapest 254-bit multiplier is the one
// with the fewest instructions of any kind.  Nine 29-bit limbs leave 6 spare bits in
// a 64-bit column accumulator: up to 36 limb products (58 bits each) are summed with
// one v_mad_u64_u32 apiece and NO carry instructions; the carry is taken once per
// column (v_lshrrev_b64 + v_and).  A Montgomery product is 81 + 81 mads + 9 mul_lo +
// ~34 shifts/ands instead of 128 mads + 128 carry ops with 32-bit limbs.  FP64-FMA
// limbs (5.2 cycles per v_fma_f64, 2 FMAs + 1 add + 2 integer adds per limb product)
// and 24-bit multiplies were measured slower.
//
// Value domain: Montgomery with R = 2^261.  Stored field elements (HBM, API buffers in
// device format) are fully reduced (< p) and packed as 8 x u32 little-endian.
#pragma once
#include <cstddef>
#include <cstdint>
#include "imt_consts.hpp"

// The arithmetic below is plain integer C++, so tests/native/ can also compile it for the
// host (g++) and check it against the oracle without a GPU.  Under hipcc everything is
// force-inlined device code.
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define IMT_HD __host__ __device__ __forceinline__
#else
#define IMT_HD inline
#endif

namespace imt {
namespace dev {

// p = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001 in 29-bit limbs
#define IMT_P29_0 0x10000001u
#define IMT_P29_1 0x1f0fac9fu
#define IMT_P29_2 0x0e5c2450u
#define IMT_P29_3 0x07d090f3u
#define IMT_P29_4 0x1585d283u
#define IMT_P29_5 0x02db40c0u
#define IMT_P29_6 0x00a6e141u
#define IMT_P29_7 0x0e5c2634u
#define IMT_P29_8 0x0030644eu
constexpr uint32_t N0INV29 = 0x0fffffffu;   // -p^-1 mod 2^29
constexpr uint32_t N0INV32 = 0xefffffffu;   // -p^-1 mod 2^32

IMT_HD constexpr uint32_t p29(int i) {
    return i == 0 ? IMT_P29_0 : i == 1 ? IMT_P29_1 : i == 2 ? IMT_P29_2 : i == 3 ? IMT_P29_3 :
           i == 4 ? IMT_P29_4 : i == 5 ? IMT_P29_5 : i == 6 ? IMT_P29_6 : i == 7 ? IMT_P29_7 : IMT_P29_8;
}


// ---------------------------------------------------------------------------------
// Montgomery reduction of a sum of NT limb products plus (optionally) addend * R.
//   r = (sum_t a[t]*b[t] + addend*R + m*p) / R
// The quotient digits m_k only have to clear the low 29 bits of their column.  With WIDE_M they
// are taken as full 32-bit words, m_k = lo32(column) * (-p^-1 mod 2^32), which saves the mask per
// digit; the column then has its low 32 bits clear and the result grows by up to 8p instead of p
// (m < 2^264 = 8R):
//   WIDE_M : r < sum/R + addend + 8p        !WIDE_M : r < sum/R + addend + p
// Column accumulators must stay below 2^64.  A column holds each limb of p at most once, so its
// m*p part is below 2^32 * (sum of the limbs of p) = 0.4251 * 2^64 (WIDE_M) or 9 * 2^58 (!WIDE_M);
// the a*b part is at most NT * 9 * max(a limb) * max(b limb), and the carry-in is below 2^36:
//   NT = 4, 29-bit limbs : 36 * 2^58 = 0.5625 * 2^64   -> 0.9876 * 2^64 with WIDE_M
//   NT = 1, 30-bit limbs on both sides : 9 * 2^60 = 0.5625 * 2^64 -> the same
// Output limbs are normalised (< 2^29; the top limb holds whatever is left, so values must stay
// below 2^261 = 169p: every lane that is only ever ADDED to between multiplications -- the linear
// lanes of the partial rounds -- uses !WIDE_M).
// ---------------------------------------------------------------------------------
template <bool WIDE_M>
IMT_HD uint32_t mont_digit(uint64_t acc) {
    return WIDE_M ? (uint32_t)acc * N0INV32 : (((uint32_t)acc * N0INV29) & MASK29);
}

}  // namespace dev
}  // namespace imt
// On the device the products below are replaced by hand-laid single-chain assembly with the same
// values (generated; see tools/gen_mont_asm.py).  -DIMT_NO_MONT_ASM keeps the C++ forms.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(IMT_NO_MONT_ASM)
#define IMT_MONT_ASM 1
#include "imt_mont_asm.hpp"
#endif
namespace imt {
namespace dev {

template <int NT, bool ADD, bool WIDE_M = true>
IMT_HD void mont_dot(Fe& r, const Fe* a, const Fe* b, const Fe& addend) {
    uint32_t m[NL];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (uint64_t)a[t].v[i] * b[t].v[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p29(k - i);
        m[k] = mont_digit<WIDE_M>(acc);
        acc += (uint64_t)m[k] * p29(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int i = k - (NL - 1); i < NL; i++) acc += (uint64_t)a[t].v[i] * b[t].v[k - i];
        }
#pragma unroll
        for (int i = k - (NL - 1); i < NL; i++) acc += (uint64_t)m[i] * p29(k - i);
        if (ADD) acc += addend.v[k - NL];
        r.v[k - NL] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    if (ADD) acc += addend.v[NL - 1];
    r.v[NL - 1] = (uint32_t)acc;
}

IMT_HD void mont_mul(Fe& r, const Fe& a, const Fe& b) {
#ifdef IMT_MONT_ASM
    masm::mul_vv(r, &a, &b);
#else
    mont_dot<1, false>(r, &a, &b, a);
#endif
}

// mont_dot whose first factors `c` are wave-uniform constants (entries of the __constant__ Poseidon
// tables indexed by the round counter).  Only permute() may use it: see imt_mont_asm.hpp.
template <int NT, bool ADD, bool WIDE_M = true>
IMT_HD void mont_dot_uc(Fe& r, const Fe* c, const Fe* v, const Fe& addend) {
#ifdef IMT_MONT_ASM
    static_assert((NT == 3 && !ADD && WIDE_M) || (NT == 4 && !ADD && WIDE_M) || (NT == 2 && ADD && !WIDE_M),
                  "no assembly form for this shape");
    if constexpr (NT == 3) masm::dot3_uc(r, c, v);
    else if constexpr (NT == 4) masm::dot4_uc(r, c, v);
    else masm::dot2_add_uc_narrow(r, c, v, addend);
#else
    mont_dot<NT, ADD, WIDE_M>(r, c, v, addend);
#endif
}

// r = a^2 / R (+ up to 8p: wide quotient digits).  36 doubled cross products + 9 squares instead
// of 81 products.  Precondition: limbs of a < 2^30 (column 8: 4 * 2^31 * 2^30 + 2^60 = 0.5625 * 2^64).
IMT_HD void mont_sqr(Fe& r, const Fe& a) {
#ifdef IMT_MONT_ASM
    masm::sqr_v(r, a);
    return;
#endif
    uint32_t m[NL], a2[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) a2[i] = a.v[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a2[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p29(k - i);
        m[k] = mont_digit<true>(acc);
        acc += (uint64_t)m[k] * p29(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - (NL - 1); 2 * i < k; i++) acc += (uint64_t)a2[i] * a.v[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
        for (int i = k - (NL - 1); i < NL; i++) acc += (uint64_t)m[i] * p29(k - i);
        r.v[k - NL] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r.v[NL - 1] = (uint32_t)acc;
}

// limb-wise add without carry propagation (limbs grow by one bit)
// r = a / R mod p (+ < p): out of the Montgomery domain without the 81 limb products of a multiplication by 1.
// a: normalised limbs (29-bit digits).  Same column structure as mont_dot with the value in the low nine columns.
IMT_HD void mont_redc(Fe& r, const Fe& a) {
    uint32_t m[NL];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        acc += a.v[k];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * p29(k - i);
        m[k] = mont_digit<false>(acc);
        acc += (uint64_t)m[k] * p29(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - (NL - 1); i < NL; i++) acc += (uint64_t)m[i] * p29(k - i);
        r.v[k - NL] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r.v[NL - 1] = (uint32_t)acc;
}

IMT_HD void add_lazy(Fe& r, const Fe& a, const Fe& b) {
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + b.v[i];
}

IMT_HD void normalize(Fe& a) {
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        a.v[i + 1] += a.v[i] >> 29;
        a.v[i] &= MASK29;
    }
}

// a >= p ?  (normalised limbs)
IMT_HD bool geq_p(const Fe& a) {
    bool ge = true;   // equal so far => >=
#pragma unroll
    for (int i = 0; i < NL; i++) {   // from least significant: later limbs override
        uint32_t pi = p29(i);
        if (a.v[i] != pi) ge = a.v[i] > pi;
    }
    return ge;
}

IMT_HD void sub_p(Fe& a) {
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t d = a.v[i] - p29(i) - borrow;
        borrow = (i < NL - 1) ? (d >> 31) : 0;        // limbs < 2^29: bit 31 set <=> negative
        a.v[i] = (i < NL - 1) ? (d & MASK29) : d;
    }
}

// limb i of (p << SH), SH = 0..3
template <int SH>
IMT_HD constexpr uint32_t p29_shl(int i) {
    return i == 0 ? ((p29(0) << SH) & MASK29)
         : i < NL - 1 ? (((p29(i) << SH) | (p29(i - 1) >> (29 - SH))) & MASK29)
                      : ((p29(i) << SH) | (p29(i - 1) >> (29 - SH)));
}
// a -= (p << SH) if a >= (p << SH)   (normalised limbs)
template <int SH>
IMT_HD void cond_sub_p_shl(Fe& a) {
    bool ge = true;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t pi = p29_shl<SH>(i);
        if (a.v[i] != pi) ge = a.v[i] > pi;
    }
    if (ge) {
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            uint32_t d = a.v[i] - p29_shl<SH>(i) - borrow;
            borrow = (i < NL - 1) ? (d >> 31) : 0;
            a.v[i] = (i < NL - 1) ? (d & MASK29) : d;
        }
    }
}

// bring a normalised value < 16p into [0, p)
IMT_HD void canonicalize(Fe& a) {
    cond_sub_p_shl<3>(a);
    cond_sub_p_shl<2>(a);
    cond_sub_p_shl<1>(a);
    cond_sub_p_shl<0>(a);
}

// 8 x u32 packed (value < 2^256) <-> 9 x 29-bit limbs
IMT_HD void unpack(Fe& r, const uint32_t w[8]) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        uint32_t lo = w[wi] >> sh;
        if (sh > 3 && wi + 1 < 8) lo |= w[wi + 1] << (32 - sh);
        r.v[i] = lo & MASK29;
    }
}
IMT_HD void pack(uint32_t w[8], const Fe& a) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        // word j covers bits [32j, 32j+32)
        const int lo_limb = (32 * j) / 29, off = 32 * j - 29 * lo_limb;
        uint32_t x = a.v[lo_limb] >> off;
        const int have = 29 - off;
        if (have < 32 && lo_limb + 1 < NL) x |= a.v[lo_limb + 1] << have;
        if (have + 29 < 32 && lo_limb + 2 < NL) x |= a.v[lo_limb + 2] << (have + 29);
        w[j] = x;
    }
}

struct alignas(16) Word4 { uint32_t x, y, z, w; };   // one 16-byte global access
IMT_HD void load_packed(Fe& r, const void* ptr) {
    const Word4* q = reinterpret_cast<const Word4*>(ptr);
    Word4 x = q[0], y = q[1];
    uint32_t w[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
    unpack(r, w);
}
IMT_HD void store_packed(void* ptr, const Fe& a) {
    uint32_t w[8];
    pack(w, a);
    Word4* q = reinterpret_cast<Word4*>(ptr);
    q[0] = Word4{w[0], w[1], w[2], w[3]};
    q[1] = Word4{w[4], w[5], w[6], w[7]};
}

// ---------------------------------------------------------------------------------
// Constant tables (filled by the host from the Grain LFSR; imt_params.cpp).  All
// lanes of a wave read the same entry at the same time, so these are scalar loads.
// ---------------------------------------------------------------------------------


IMT_HD void sbox(Fe& x) {   // x <- x^5; limbs of x < 2^30
    Fe x2, x4;
    mont_sqr(x2, x);
    mont_sqr(x4, x2);
    mont_mul(x, x4, x);
}

// One permutation, optimised schedule (same values as the plain 65-round form):
//   rounds 0..3   : s += c_r; x^5 on all lanes; s = M s  (round 3 uses PRE = N' M)
//   rounds 4..60  : s0 += k_p; y = s0^5; s0 = row_p . (y, s1, s2); s_i += col_p,i * y
//   rounds 61..64 : as 0..3 (round 61's constants carry the partial rounds' leftover)
// The 57 partial rounds run in pairs: the second round of a pair takes the linear lanes as they
// were before the first one (its row product gets the extra term gamma * y_first), so each linear
// lane is reduced once per pair, REDC(s_i R + col_p,i y_first + col_p+1,i y_second), instead of once
// per round: 11 products + 4 reductions per pair instead of 10 + 6.  The odd 57th round runs
// through the same code with y_second = 0.
// One loop so that each body exists once in the instruction stream.
// `first_rc` replaces the round-0 constants (sponge padding folded in by the caller).
// Entry: limbs normalised, lanes < 16p.  Exit: limbs normalised, every lane < 9p.
IMT_HD void permute(const PoseidonConsts& pc, Fe s[3], const Fe* first_rc) {
    constexpr int NSTEP = RF + (RP + 1) / 2;   // 4 full, 29 partial pairs, 4 full
#pragma unroll 1
    for (int st = 0; st < NSTEP; st++) {
        if (st < RF / 2 || st >= RF / 2 + (RP + 1) / 2) {
            const int fr = st < RF / 2 ? st : st - (RP + 1) / 2;
            const Fe* rc = (st == 0) ? first_rc : pc.rc_full[fr];
            add_lazy(s[0], s[0], rc[0]);
            add_lazy(s[1], s[1], rc[1]);
            add_lazy(s[2], s[2], rc[2]);
            sbox(s[0]); sbox(s[1]); sbox(s[2]);
            const Fe(*mat)[3] = pc.mats[st == RF / 2 - 1 ? 1 : 0];
            Fe n0, n1, n2;
            mont_dot_uc<3, false>(n0, mat[0], s, s[0]);
            mont_dot_uc<3, false>(n1, mat[1], s, s[0]);
            mont_dot_uc<3, false>(n2, mat[2], s, s[0]);
            s[0] = n0; s[1] = n1; s[2] = n2;
        } else {
            const int p = 2 * (st - RF / 2);
            const bool second = p + 1 < RP;
            Fe v[4], y[2], n0;
            add_lazy(v[0], s[0], pc.k_partial[p]);
            sbox(v[0]);
            v[1] = s[1]; v[2] = s[2];
            mont_dot_uc<3, false>(n0, pc.sp_row[p], v, v[0]);
            y[0] = v[0];
#pragma unroll
            for (int i = 0; i < NL; i++) y[1].v[i] = 0;
            if (second) {
                add_lazy(v[0], n0, pc.k_partial[p + 1]);
                sbox(v[0]);
                v[3] = y[0];
                const Fe c4[4] = {pc.sp_row[p + 1][0], pc.sp_row[p + 1][1], pc.sp_row[p + 1][2], pc.sp_gamma[p + 1]};
                mont_dot_uc<4, false>(n0, c4, v, v[0]);
                y[1] = v[0];
            }
            const int q = second ? p + 1 : p;    // with y[1] = 0 the second column constant is unused
            const Fe c1[2] = {pc.sp_col[p][0], pc.sp_col[q][0]};
            const Fe c2[2] = {pc.sp_col[p][1], pc.sp_col[q][1]};
            mont_dot_uc<2, true, false>(s[1], c1, y, s[1]);   // narrow digits: these lanes only accumulate
            mont_dot_uc<2, true, false>(s[2], c2, y, s[2]);
            s[0] = n0;
        }
    }
}

// Poseidon::update(&[a,b]) or update(&[a,b,c]) followed by squeeze_and_reset():
// two permutations either way.  `three` must be wave-uniform.
IMT_HD void hash23(const PoseidonConsts& pc, Fe& out, const Fe& a, const Fe& b, const Fe& c, bool three) {
    Fe s[3] = {pc.cap0, a, b};
#pragma unroll 1
    for (int blk = 0; blk < 2; blk++) {
        const Fe* rc0 = pc.rc_full[0];
        if (blk == 1) {
            if (three) {               // absorb [c, 1] into lanes 1, 2
                add_lazy(s[1], s[1], c);
                add_lazy(s[2], s[2], pc.one);
                normalize(s[1]);
                normalize(s[2]);
            } else {                   // absorb the padding 1 into lane 1 (folded into rc)
                rc0 = pc.rc_h2p2;
            }
        }
        permute(pc, s, rc0);
    }
    out = s[1];
    canonicalize(out);
}
// The same with the third input parked in a per-thread slot of `stash` (LDS on the device: stash[limb * stride]),
// written by the caller and read back only where the second permutation absorbs it: nine registers fewer are live
// across the first permutation, which is what lets ONE copy of the hash serve 2- and 3-input callers at 96 VGPRs.
IMT_HD void hash23_stashed(const PoseidonConsts& pc, Fe& out, const Fe& a, const Fe& b, bool three,
                           const uint32_t* stash, unsigned stride) {
    Fe s[3] = {pc.cap0, a, b};
#pragma unroll 1
    for (int blk = 0; blk < 2; blk++) {
        const Fe* rc0 = pc.rc_full[0];
        if (blk == 1) {
            if (three) {               // absorb [c, 1] into lanes 1, 2
                Fe c;
#pragma unroll
                for (int i = 0; i < NL; i++) c.v[i] = stash[(size_t)i * stride];
                add_lazy(s[1], s[1], c);
                add_lazy(s[2], s[2], pc.one);
                normalize(s[1]);
                normalize(s[2]);
            } else {                   // absorb the padding 1 into lane 1 (folded into rc)
                rc0 = pc.rc_h2p2;
            }
        }
        permute(pc, s, rc0);
    }
    out = s[1];
    canonicalize(out);
}
IMT_HD void hash2(const PoseidonConsts& pc, Fe& out, const Fe& a, const Fe& b) { hash23(pc, out, a, b, a, false); }
IMT_HD void hash3(const PoseidonConsts& pc, Fe& out, const Fe& a, const Fe& b, const Fe& c) { hash23(pc, out, a, b, c, true); }

// ---- boundary formats ----------------------------------------------------------
enum : unsigned { FMT_CANONICAL = 0, FMT_MONT256 = 1, FMT_DEVICE = 2 };

// returns false if the stored integer is >= p (canonical / mont256 inputs must be reduced)
IMT_HD bool load_fe(const PoseidonConsts& pc, Fe& r, const void* ptr, unsigned fmt) {
    Fe raw;
    load_packed(raw, ptr);
    bool ok = !geq_p(raw) && (raw.v[NL - 1] >> 24) == 0;   // top limb holds bits 232..255
    if (fmt == FMT_DEVICE) { r = raw; return ok; }
    mont_mul(r, raw, fmt == FMT_CANONICAL ? pc.from_canon : pc.from_mont256);
    canonicalize(r);
    return ok;
}
IMT_HD void store_fe(const PoseidonConsts& pc, void* ptr, const Fe& a, unsigned fmt) {
    if (fmt == FMT_DEVICE) { store_packed(ptr, a); return; }
    Fe t;
    mont_mul(t, a, fmt == FMT_CANONICAL ? pc.int_one : pc.to_mont256);
    canonicalize(t);
    store_packed(ptr, t);
}

IMT_HD bool fe_eq(const Fe& a, const Fe& b) {   // both canonical
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) d |= a.v[i] ^ b.v[i];
    return d == 0;
}
IMT_HD bool fe_is_zero(const Fe& a) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) d |= a.v[i];
    return d == 0;
}

}  // namespace dev
}  // namespace imt
