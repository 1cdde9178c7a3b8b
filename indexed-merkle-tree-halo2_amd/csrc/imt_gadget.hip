// imt_gadget.hip -- f3: every NEW advice value the reference's insert_leaf assigns OUTSIDE hash_fix_len_array
// (imt_less_than_trace_batch, imt_insert_gadget_trace_batch; the hashes' own cells are imt_trace_device.hpp).
//
//   is_less_than          /root/reference/src/indexed_merkle_tree.rs:98-125  (two range.is_less_than(.,.,128), two
//                         gate.is_equal, not x4, and x3, and, or)
//   verify_non_inclusion  :127-229   insert_leaf :231-314   select :33-45   dual_mux :47-63   compute_merkle_root :78-96
//
// Row order = the order in which halo2-base's GateChip / RangeChip assign Witness cells (halo2-lib v0.4.x,
// gates/flex_gate.rs, gates/range.rs; UNPINNED BY THE REFERENCE like the f1 trace: the crate is not vendored):
//   sub(a,b) [W a-b, b, 1, a]   mul / and [0, a, b, W ab]   mul_add [c, a, b, W ab+c]   not(a) = sub(1, a)
//   or(a,b) [W 1-b, 1, b, 1, b, a, W 1-b, W a+b-ab]         is_zero(a) [W z, a, W 1/a, 1, 0, a, W z, 0]
//   range.is_less_than(a, b, 128): [W 2^padded+a-b, b, 1, W 2^padded+a, -2^padded, 1, a], the k + 1 limbs of the first
//   cell with their running sums [W l0, W l1, 2^lb, W s1, ...], is_zero(top limb)         (k = ceil(128 / lookup_bits))
//
// Nothing here is a hash: integer work on 256-bit values (limbs, shifted differences, booleans), subtractions mod p and
// one modular inverse per is_equal (binary extended Euclid on canonical integers).  One thread per item; rows are
// written canonical, row-major [rows][n] (a wave's 64 stores of one row are contiguous) or item-major; the C entry
// converts to the caller's format afterwards.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "imt_gadget.hpp"

namespace imt {
namespace {

constexpr int BLOCK = 128;

struct U256 {
    uint32_t w[8];
};

__device__ __forceinline__ U256 u256_zero() {
    U256 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.w[i] = 0;
    return r;
}
__device__ __forceinline__ U256 u256_small(uint32_t v) {
    U256 r = u256_zero();
    r.w[0] = v;
    return r;
}
// p = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001 (src/indexed_merkle_tree.rs:383)
__device__ __forceinline__ U256 modulus() {
    U256 p;
    p.w[0] = 0xf0000001u; p.w[1] = 0x43e1f593u; p.w[2] = 0x79b97091u; p.w[3] = 0x2833e848u;
    p.w[4] = 0x8181585du; p.w[5] = 0xb85045b6u; p.w[6] = 0xe131a029u; p.w[7] = 0x30644e72u;
    return p;
}
__device__ __forceinline__ U256 load256(const uint8_t* p) {
    U256 r;
    const uint4* q = (const uint4*)p;
    const uint4 a = q[0], b = q[1];
    r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w;
    r.w[4] = b.x; r.w[5] = b.y; r.w[6] = b.z; r.w[7] = b.w;
    return r;
}
__device__ __forceinline__ void store256(uint8_t* p, const U256& v) {
    uint4* q = (uint4*)p;
    q[0] = make_uint4(v.w[0], v.w[1], v.w[2], v.w[3]);
    q[1] = make_uint4(v.w[4], v.w[5], v.w[6], v.w[7]);
}
__device__ __forceinline__ U256 add256(const U256& a, const U256& b) {
    U256 r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.w[i] + b.w[i];
        r.w[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}
__device__ __forceinline__ U256 sub256(const U256& a, const U256& b, uint32_t* borrow_out = nullptr) {
    U256 r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.w[i] - (int64_t)b.w[i];
        r.w[i] = (uint32_t)c;
        c >>= 32;
    }
    if (borrow_out) *borrow_out = c ? 1u : 0u;
    return r;
}
__device__ __forceinline__ bool geq256(const U256& a, const U256& b) {
    uint32_t borrow;
    sub256(a, b, &borrow);
    return !borrow;
}
__device__ __forceinline__ bool is_zero256(const U256& a) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= a.w[i];
    return d == 0;
}
__device__ __forceinline__ bool is_one256(const U256& a) {
    uint32_t d = a.w[0] ^ 1u;
#pragma unroll
    for (int i = 1; i < 8; i++) d |= a.w[i];
    return d == 0;
}
__device__ __forceinline__ U256 shr1(const U256& a) {
    U256 r;
#pragma unroll
    for (int i = 0; i < 7; i++) r.w[i] = (a.w[i] >> 1) | (a.w[i + 1] << 31);
    r.w[7] = a.w[7] >> 1;
    return r;
}
// (a - b) mod p for a, b < p
__device__ __forceinline__ U256 sub_mod(const U256& a, const U256& b) {
    uint32_t borrow;
    U256 r = sub256(a, b, &borrow);
    return borrow ? add256(r, modulus()) : r;
}
// x / 2 mod p for x < p
__device__ __forceinline__ U256 half_mod(const U256& x) { return (x.w[0] & 1u) ? shr1(add256(x, modulus())) : shr1(x); }
// 1 / a mod p for 0 < a < p: binary extended Euclid (the modulus is odd); u, v shrink by at least one bit every other
// step, so the loop ends after at most 2 * 254 rounds
__device__ U256 inv_mod(const U256& a) {
    U256 u = a, v = modulus(), x1 = u256_small(1), x2 = u256_zero();
    for (int guard = 0; guard < 1100 && !is_one256(u) && !is_one256(v); guard++) {
        while (!(u.w[0] & 1u)) { u = shr1(u); x1 = half_mod(x1); }
        while (!(v.w[0] & 1u)) { v = shr1(v); x2 = half_mod(x2); }
        if (geq256(u, v)) { u = sub256(u, v); x1 = sub_mod(x1, x2); }
        else { v = sub256(v, u); x2 = sub_mod(x2, x1); }
    }
    return is_one256(u) ? x1 : x2;
}
// bits [pos, pos + nbits) of v, nbits <= 28
__device__ __forceinline__ uint32_t bits_at(const U256& v, unsigned pos, unsigned nbits) {
    const unsigned wi = pos >> 5, sh = pos & 31;
    uint64_t x = wi < 8 ? v.w[wi] : 0;
    if (wi + 1 < 8) x |= (uint64_t)v.w[wi + 1] << 32;
    return (uint32_t)(x >> sh) & ((1u << nbits) - 1u);
}
// v with every bit at or above `nbits` cleared
__device__ __forceinline__ U256 low_bits(const U256& v, unsigned nbits) {
    U256 r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const unsigned lo = 32u * i;
        r.w[i] = nbits >= lo + 32 ? v.w[i] : nbits <= lo ? 0u : (v.w[i] & ((1u << (nbits - lo)) - 1u));
    }
    return r;
}
__device__ __forceinline__ U256 pow2_256(unsigned e) {
    U256 r = u256_zero();
    r.w[e >> 5] = 1u << (e & 31);
    return r;
}
__device__ __forceinline__ U256 high128(const U256& v) {
    U256 r = u256_zero();
    r.w[0] = v.w[4]; r.w[1] = v.w[5]; r.w[2] = v.w[6]; r.w[3] = v.w[7];
    return r;
}
__device__ __forceinline__ U256 low128(const U256& v) {
    U256 r = u256_zero();
    r.w[0] = v.w[0]; r.w[1] = v.w[1]; r.w[2] = v.w[2]; r.w[3] = v.w[3];
    return r;
}

// where row r of item i goes
struct Out {
    uint8_t* base;
    uint64_t row_stride, item_stride;      // in bytes
    uint64_t item;
    uint32_t row;
    __device__ __forceinline__ void put(const U256& v) {
        store256(base + row * row_stride + item * item_stride, v);
        row++;
    }
    __device__ __forceinline__ void put_bit(uint32_t b) { put(u256_small(b)); }
};

// is_zero(a): rows z, 1/a (1 if a = 0), z; returns z
__device__ uint32_t emit_is_zero(Out& o, const U256& a) {
    const uint32_t z = is_zero256(a) ? 1u : 0u;
    o.put_bit(z);
    o.put(z ? u256_small(1) : inv_mod(a));
    o.put_bit(z);
    return z;
}
// range.is_less_than(a, b, 128) for a, b < 2^128; returns a < b
__device__ uint32_t emit_range_lt(Out& o, const U256& a, const U256& b, unsigned lb) {
    const unsigned k = (128 + lb - 1) / lb, padded = k * lb, L = k + 1;
    const U256 sa = add256(pow2_256(padded), a);
    const U256 sab = sub256(sa, b);
    o.put(sab);
    o.put(sa);
    uint32_t top = 0;
    for (unsigned i = 0; i < L; i++) {
        top = bits_at(sab, i * lb, lb);
        o.put(u256_small(top));
        if (i) o.put(low_bits(sab, (i + 1) * lb));           // the running sum = the low (i + 1) limbs
    }
    return emit_is_zero(o, u256_small(top));                 // top limb 0 <=> no carry into bit `padded` <=> a < b
}
// gate.is_equal(a, b): rows a - b mod p, then is_zero
__device__ uint32_t emit_is_equal(Out& o, const U256& a, const U256& b) {
    const U256 d = sub_mod(a, b);
    o.put(d);
    return emit_is_zero(o, d);
}
// is_less_than(a_q, a_r, b_q, b_r) :98-125; returns a < b
__device__ uint32_t emit_less_than(Out& o, const U256& a, const U256& b, unsigned lb) {
    const U256 a_q = high128(a), a_r = low128(a), b_q = high128(b), b_r = low128(b);
    const uint32_t msb_lt = emit_range_lt(o, a_q, b_q, lb);
    const uint32_t msb_eq = emit_is_equal(o, a_q, b_q);
    const uint32_t lsb_lt = emit_range_lt(o, a_r, b_r, lb);
    const uint32_t lsb_eq = emit_is_equal(o, a_r, b_r);
    const uint32_t c_not = 1u - msb_eq, a_not = 1u - msb_lt, c = 1u - c_not, d_not = 1u - lsb_eq;
    o.put_bit(c_not); o.put_bit(a_not); o.put_bit(c); o.put_bit(d_not);
    const uint32_t t1 = a_not & lsb_lt, t2 = t1 & c, rhs = t2 & d_not, lhs = msb_lt & c_not;
    o.put_bit(t1); o.put_bit(t2); o.put_bit(rhs); o.put_bit(lhs);
    const uint32_t out = lhs | rhs;
    o.put_bit(1u - rhs); o.put_bit(1u - rhs); o.put_bit(out);
    return out;
}

__global__ void __launch_bounds__(BLOCK) k_less_than_trace(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, size_t n,
                                                           unsigned lb, uint8_t* __restrict__ trace, uint64_t row_stride,
                                                           uint64_t item_stride, uint8_t* __restrict__ lt_out) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    Out o{trace, row_stride, item_stride, i, 0};
    const uint32_t lt = emit_less_than(o, load256(a + i * 32), load256(b + i * 32), lb);
    if (lt_out) lt_out[i] = (uint8_t)lt;
}

// compute_merkle_root :78-96 outside its hashes: load_witness(leaf), then per level dual_mux's a-b, b-a, left, right
__device__ void emit_chain(Out& o, const uint8_t* __restrict__ pairs, uint64_t index, unsigned depth, size_t n, size_t i) {
    for (unsigned l = 0; l < depth; l++) {
        const uint8_t* pp = pairs + ((size_t)l * n + i) * 64;
        const U256 left = load256(pp), right = load256(pp + 32);
        const bool sw = ((index >> l) & 1) == 0;             // helper = 1 <=> the node is a left child (src/utils.rs:79)
        const U256 cur = sw ? left : right, sib = sw ? right : left;
        if (l == 0) o.put(cur);                              // ctx.load_witness(*leaf.value()) :88
        o.put(sub_mod(cur, sib));
        o.put(sub_mod(sib, cur));
        o.put(left);
        o.put(right);
    }
}

__global__ void __launch_bounds__(BLOCK)
k_insert_gadget(const uint8_t* __restrict__ low_leaf, const uint64_t* __restrict__ low_index, const uint8_t* __restrict__ new_leaf,
                unsigned new_stride /*96: leaves; 32: bare values*/, const uint64_t* __restrict__ new_path_index,
                const uint8_t* __restrict__ is_largest, const uint8_t* __restrict__ pairs /*[chains][depth][n][2][32] canonical*/,
                unsigned depth, unsigned lb, size_t n, int whole_insert /*0: verify_non_inclusion alone (one chain)*/,
                uint8_t* __restrict__ trace, uint64_t row_stride, uint64_t item_stride) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    Out o{trace, row_stride, item_stride, i, 0};
    const U256 low_val = load256(low_leaf + i * 96), low_next = load256(low_leaf + i * 96 + 32);
    const U256 nv = load256(new_leaf + i * (size_t)new_stride);
    // verify_non_inclusion :127-229
    const uint32_t iz = emit_is_equal(o, low_next, u256_zero());                       // :143
    o.put(high128(nv)); o.put(low128(nv)); o.put(high128(low_next)); o.put(low128(low_next));   // :169-172
    o.put(nv); o.put(low_next);                                                        // mul_add: q 2^128 + r :175-178
    const uint32_t lt = emit_less_than(o, nv, low_next, lb);                            // :180
    const uint32_t s = is_largest[i] ? 1u : 0u;                                         // select :182-189
    o.put_bit(iz & s); o.put_bit(1u - s); o.put_bit(((1u - s) & lt) | (iz & s));
    const size_t ps = (size_t)depth * n * 64;
    emit_chain(o, pairs, low_index[i], depth, n, i);                                    // :196-204
    o.put(high128(low_val)); o.put(low128(low_val)); o.put(low_val);                    // :219-224
    emit_less_than(o, low_val, nv, lb);                                                 // :226
    if (!whole_insert) return;
    // insert_leaf :277-312
    emit_chain(o, pairs + ps, low_index[i], depth, n, i);
    emit_chain(o, pairs + 2 * ps, new_path_index[i], depth, n, i);
    emit_chain(o, pairs + 3 * ps, new_path_index[i], depth, n, i);
}

}  // namespace

namespace launch {

void less_than_trace(hipStream_t s, const uint8_t* a, const uint8_t* b, size_t n, unsigned lookup_bits, uint8_t* trace,
                     uint64_t row_stride, uint64_t item_stride, uint8_t* lt_out) {
    if (!n) return;
    hipLaunchKernelGGL(k_less_than_trace, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, a, b, n, lookup_bits, trace,
                       row_stride, item_stride, lt_out);
}

void insert_gadget(hipStream_t s, const uint8_t* low_leaf, const uint64_t* low_index, const uint8_t* new_leaf,
                   const uint64_t* new_path_index, const uint8_t* is_largest, const uint8_t* pairs, unsigned depth,
                   unsigned lookup_bits, size_t n, uint8_t* trace, uint64_t row_stride, uint64_t item_stride) {
    if (!n) return;
    hipLaunchKernelGGL(k_insert_gadget, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, low_leaf, low_index, new_leaf,
                       96u, new_path_index, is_largest, pairs, depth, lookup_bits, n, 1, trace, row_stride, item_stride);
}

void non_inclusion_gadget(hipStream_t s, const uint8_t* low_leaf, const uint64_t* low_index, const uint8_t* new_val,
                          const uint8_t* is_largest, const uint8_t* pairs, unsigned depth, unsigned lookup_bits, size_t n,
                          uint8_t* trace, uint64_t row_stride, uint64_t item_stride) {
    if (!n) return;
    hipLaunchKernelGGL(k_insert_gadget, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, low_leaf, low_index, new_val,
                       32u, low_index, is_largest, pairs, depth, lookup_bits, n, 0, trace, row_stride, item_stride);
}

}  // namespace launch
}  // namespace imt
