// imt_kernels.hip -- hand-written gfx950 (MI355X) kernels of the indexed-Merkle-tree path.
//
// One thread owns one hash chain: its 3x9-limb Poseidon state stays in VGPRs for all 65
// rounds, round constants arrive as wave-uniform scalar loads (SGPR operands of
// v_mad_u64_u32), and HBM is touched only for the 32-byte inputs / siblings / outputs.
// The path is VALU-integer bound (about 76k v_mad_u64_u32 per permutation against
// 32..100 bytes of traffic), so there is no LDS staging and no MFMA: nothing is reused
// across lanes and there is no dense contraction ("Kernels").
//
// Reference rows: a1/a10 hash_batch, a2 tree_level, a4 gather_proof,
// a5/a8/a9 path_root, a13 non_membership, a14 insert_witness, a15 sweep_* (index logic in imt_sweep.hpp;
// the hash-free batch preparation is a separate translation unit, imt_prep.hip).
#include <algorithm>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "imt_device.hpp"
#include "imt_trace_device.hpp"
#include "imt_launch.hpp"
#include "imt_sweep.hpp"
#if defined(__HIP_DEVICE_COMPILE__)
#include "imt_coop_device.hpp"
#endif

namespace imt {

__constant__ dev::PoseidonConsts g_pc;
__constant__ dev::TraceConsts g_tc;      // halo2-base form of the same permutation (witness trace, f1)

namespace {
using namespace dev;

#ifndef IMT_BLOCK
#define IMT_BLOCK 256
#endif
// Every hash kernel asks for 5 waves per SIMD (96 VGPRs).  With
// the assembly multipliers the hash needs < 90 registers; measured on MI355X (bench.py, 16 steps,
// same box): 4 waves 2.93 M insertions/s, 5 waves 3.05 M, 6 waves 3.02-3.07 M but the kernel alone
// 2 % slower, 8 waves 2.93 M.  The fifth wave is what the small kernels of the other batch and the
// scalar-load waits of the hash overlap with.
#ifndef IMT_HASH_WAVES
#define IMT_HASH_WAVES __attribute__((amdgpu_waves_per_eu(5, 5)))
#endif
// The latency forms of the PATH kernels (a quad of lanes per item, imt_coop_device.hpp) run with at most one wave per
// SIMD by construction -- the launcher picks them only while the launch leaves the chip mostly empty -- so registers are
// free: with the 96 of the throughput kernels the two inlined copies of the quad hash plus the loop-carried path state
// spilled 13-88 VGPRs inside the 33-hash chain; with up to 256 nothing spills.  (k_sweep_coop and k_hash_batch_coop
// fit 96 without spills and keep it: they run up to 16 384 events, where a second wave per SIMD still pays.)
#ifndef IMT_COOP_WAVES
#define IMT_COOP_WAVES __attribute__((amdgpu_waves_per_eu(1, 2)))
#endif
constexpr int BLOCK = IMT_BLOCK;   // 256 = 4 waves = one per SIMD of a CU

__device__ __forceinline__ size_t gtid() { return (size_t)blockIdx.x * blockDim.x + threadIdx.x; }
// the same, recomputed where it is called: behind a loop of inlined hashes the item number costs three instructions to
// rebuild but two registers (and every address the compiler would hoist from it) to keep across the loop
__device__ __forceinline__ size_t gtid_again() {
    uint32_t lo = blockIdx.x * blockDim.x + threadIdx.x, hi = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return ((size_t)hi << 32) | lo;
}
__device__ __forceinline__ void flag_err(int* err, bool ok) {
    if (!ok && err) atomicOr(err, 1);
}

// integer value (not Montgomery) of a device-form element, canonical limbs
__device__ __forceinline__ void to_int(Fe& r, const Fe& a) {
    mont_mul(r, a, g_pc.int_one);
    canonicalize(r);
}
__device__ __forceinline__ bool int_lt(const Fe& a, const Fe& b) {   // canonical limbs
    bool lt = false;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        if (a.v[i] != b.v[i]) lt = a.v[i] < b.v[i];
    }
    return lt;
}
__device__ __forceinline__ void fe_from_u64(Fe& r, uint64_t x) {
    Fe t;
#pragma unroll
    for (int i = 0; i < NL; i++) t.v[i] = 0;
    t.v[0] = (uint32_t)x & MASK29;
    t.v[1] = (uint32_t)(x >> 29) & MASK29;
    t.v[2] = (uint32_t)(x >> 58);
    mont_mul(r, t, g_pc.from_canon);
    canonicalize(r);
}

// ---------------------------------------------------------------------------------
// One hash chain with the hash INLINED, exactly once per kernel (round 4; what k_sweep got in round 2): one loop whose
// iteration -1 is the 3-input leaf hash (when `with_leaf`, wave-uniform) and whose iterations 0 .. depth-1 climb the
// path -- right child at level l <=> bit l of idx is 1 (src/utils.rs:93-101).  No call, hence no call ABI: rounds 1-3
// had ONE shared, non-inlined hash function for these kernels (so that different hash kernels of consecutive batches
// would not evict each other's 47 KB from the 64 KB instruction cache -- k_sweep is one kernel for that reason); its
// callers kept 1-3 VGPRs spilled and 80-156 B of scratch per thread (the callee owns the registers;
// profiles/r03_kernel_resources.txt), 0 / 0 now.  The third input of
// the leaf hash waits in LDS for the second permutation (hash23_stashed); `on_level` sees the (left, right) pair of
// every level before it is hashed (k_path_pairs stores it).  These kernels run alone on the chip (a batch of paths, a
// tree level), so a private copy of the 47 KB hash body per kernel costs no instruction-cache sharing.
// ---------------------------------------------------------------------------------
struct NoLevelHook {
    __device__ __forceinline__ void operator()(unsigned, const Fe&, const Fe&) const {}
};
struct NoLeafHook {
    __device__ __forceinline__ void operator()(const Fe&) const {}
};
// `index` / `flip`: bit l of index[item] ^ flip = the node is a right child at level l.  The index word is re-read and
// the item number rebuilt (gtid_again) inside every iteration on purpose: a cached 8-byte load and three scalar-ish
// instructions per hash instead of four registers live across it.
template <class LoadLeaf, class OnLeaf = NoLeafHook, class OnLevel = NoLevelHook>
__device__ __forceinline__ void chain_inline(Fe& cur, bool with_leaf, LoadLeaf load_leaf, OnLeaf on_leaf,
                                             const uint64_t* __restrict__ index, uint64_t flip, const uint8_t* sib,
                                             launch::SibLayout lay, unsigned depth, unsigned fmt_in, bool& ok, uint32_t* stash,
                                             OnLevel on_level = OnLevel()) {
#pragma unroll 1
    for (int it = with_leaf ? -1 : 0; it < (int)depth; it++) {
        Fe A, B;
        const bool three = it < 0;
        if (three) {
            Fe C;
            load_leaf(A, B, C, ok);
#pragma unroll
            for (int i = 0; i < NL; i++) stash[(size_t)i * BLOCK] = C.v[i];
        } else {
            const size_t item = gtid_again();
            Fe sv;
            ok &= load_fe(g_pc, sv, sib + ((uint64_t)it * lay.level_stride + item * lay.item_stride) * 32, fmt_in);
            const bool right = ((index[item] ^ flip) >> it) & 1;
#pragma unroll
            for (int i = 0; i < NL; i++) {
                A.v[i] = right ? sv.v[i] : cur.v[i];
                B.v[i] = right ? cur.v[i] : sv.v[i];
            }
            on_level((unsigned)it, A, B);
        }
        hash23_stashed(g_pc, cur, A, B, three, stash, BLOCK);
        if (three) on_leaf(cur);
    }
}
// the leaf preimage at p ([3][32], fmt_in) as the three inputs
struct LeafAt {
    const uint8_t* base;     // [n][3][32]
    unsigned fmt_in;
    __device__ __forceinline__ void operator()(Fe& a, Fe& b, Fe& c, bool& ok) const {
        const uint8_t* p = base + gtid_again() * 96;
        ok &= load_fe(g_pc, a, p, fmt_in);
        ok &= load_fe(g_pc, b, p + 32, fmt_in);
        ok &= load_fe(g_pc, c, p + 64, fmt_in);
    }
};

// ---- a1 / a10 --------------------------------------------------------------------
__global__ IMT_HASH_WAVES void __launch_bounds__(BLOCK) k_hash_batch(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                      size_t n, int arity, unsigned fmt_in, unsigned fmt_out,
                                                      int* err) {
    const size_t i = gtid();
    if (i >= n) return;
    const uint8_t* p = in + i * 32 * (size_t)arity;
    __shared__ uint32_t stash[NL][BLOCK];
    Fe a, b, o;
    bool ok = load_fe(g_pc, a, p, fmt_in);
    ok &= load_fe(g_pc, b, p + 32, fmt_in);
    if (arity == 3) {
        Fe c;
        ok &= load_fe(g_pc, c, p + 64, fmt_in);
#pragma unroll
        for (int q = 0; q < NL; q++) stash[q][threadIdx.x] = c.v[q];
    }
    hash23_stashed(g_pc, o, a, b, arity == 3, &stash[0][threadIdx.x], BLOCK);
    store_fe(g_pc, out + i * 32, o, fmt_out);
    flag_err(err, ok);
}

// the same for few hashes: a quad of lanes per hash (imt_coop_device.hpp): one hasher.update / squeeze_and_reset pair
// of the reference (src/utils.rs:46-47) is one hash, and its latency is all there is to it
__global__ IMT_HASH_WAVES void __launch_bounds__(BLOCK) k_hash_batch_coop(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                           size_t n, int arity, unsigned fmt_in, unsigned fmt_out,
                                                           int* err) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint32_t tab[coop::TAB_DWORDS];
    coop::tab_fill(tab, g_pc);
    const size_t t = gtid();
    const size_t i = t >> 2;
    if (i >= n) return;
    const unsigned role = (unsigned)t & 3u, ri = role == 3u ? 0u : role;
    const uint8_t* p = in + i * 32 * (size_t)arity;
    Fe X, C3, o;
    C3 = g_pc.one;
    bool ok = load_fe(g_pc, X, p + (ri == 2u ? 32 : 0), fmt_in);            // lane 1: first input, lane 2: second
    if (ri == 1u) ok &= load_fe(g_pc, o, p + 32, fmt_in);                   // (every element validated once)
    if (arity == 3) ok &= load_fe(g_pc, C3, p + 64, fmt_in);
    coop::hash23(tab, o, X, C3, arity == 3, ri);
    if (role == 1u) store_fe(g_pc, out + i * 32, o, fmt_out);
    flag_err(err, ok);
#endif
}

__global__ void __launch_bounds__(BLOCK) k_permute_batch(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                         size_t n, unsigned fmt_in, unsigned fmt_out, int* err) {
    const size_t i = gtid();
    if (i >= n) return;
    Fe s[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; j++) ok &= load_fe(g_pc, s[j], in + (i * 3 + j) * 32, fmt_in);
    permute(g_pc, s, g_pc.rc_full[0]);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        canonicalize(s[j]);
        store_fe(g_pc, out + (i * 3 + j) * 32, s[j], fmt_out);
    }
    flag_err(err, ok);
}

__global__ void __launch_bounds__(BLOCK) k_convert(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n,
                                                   unsigned fmt_in, unsigned fmt_out, int* err) {
    const size_t i = gtid();
    if (i >= n) return;
    Fe x;
    bool ok = load_fe(g_pc, x, in + i * 32, fmt_in);
    store_fe(g_pc, out + i * 32, x, fmt_out);
    flag_err(err, ok);
}

// ---- f1: witness trace of hash_fix_len_array (imt_trace_device.hpp) -----------------
// One thread = one hash = 1208 / 1209 rows of 32 bytes.  The n_items hashes form blocks of n_per items (the levels
// of a path trace; one block for a plain batch); block l starts at row row0 + l * rows of the trace, whose layout is
// row-major ([rows_total][n_per]: a wave's 64 stores of one row are 2 KiB contiguous) or item-major
// ([n_per][rows_total]: the order a per-hash consumer reads).
// No register cap: 140 VGPRs = 3 waves/SIMD without spills measured the same as 128 = 4 waves with 12 spilled
// (one wave's dependent chain already fills 78 % of a SIMD's issue slots).
#ifndef IMT_TRACE_WAVES
#define IMT_TRACE_WAVES
#endif
template <unsigned FMT_OUT>
__global__ IMT_TRACE_WAVES void __launch_bounds__(BLOCK) k_hash_trace(launch::TraceJobs a) {
    const launch::TraceJobs::Job jb = a.j[blockIdx.y];         // wave-uniform
    const size_t q = gtid();
    if (q >= jb.n_items) return;
    const uint8_t* p = jb.in + q * 32 * (size_t)jb.arity;
    Fe x, y, z;
    bool ok = load_fe(g_pc, x, p, jb.fmt_in);
    ok &= load_fe(g_pc, y, p + 32, jb.fmt_in);
    z = x;
    if (jb.arity == 3) ok &= load_fe(g_pc, z, p + 64, jb.fmt_in);
    const size_t l = q / a.n_per, i = q - l * a.n_per;
    const size_t first = jb.row0 + l * (size_t)(jb.arity == 3 ? TRACE_ROWS_H3 : TRACE_ROWS_H2);
    TraceSink o{a.item_major ? a.trace + (i * a.rows_total + first) * 32 : a.trace + (first * a.n_per + i) * 32,
                a.item_major ? (uint64_t)32 : (uint64_t)a.n_per * 32};
    hash_trace<FMT_OUT>(g_pc, g_tc, o, x, y, z, jb.arity == 3);
    flag_err(a.err, ok);
}

// (left, right) inputs of every hash2 along n paths, for the trace of a whole path: pairs[l][i][2] in device
// format (the leaf hash's output is the level-0 start).  Same walk as k_path_root.  blockIdx.y selects one of up to
// four chains (the four compute_merkle_root calls of insert_leaf run as one launch).
// (4 or 5 waves per SIMD: with the pair stores inside the loop, 96 registers left the work-item id in scratch)
__global__ __attribute__((amdgpu_waves_per_eu(4, 5))) void __launch_bounds__(BLOCK) k_path_pairs(launch::PathChains a) {
    const size_t i = gtid();
    if (i >= a.n) return;
    const launch::PathChains::Chain ch = a.c[blockIdx.y];
    __shared__ uint32_t stash[NL][BLOCK];
    bool ok = true;
    Fe cur;
    if (!ch.leaf3) ok &= load_fe(g_pc, cur, ch.leaf + i * 32, a.fmt_in);
    uint8_t* const pairs = ch.pairs;
    const size_t n = a.n;
    auto store_pair = [pairs, n](unsigned l, const Fe& x, const Fe& y) {
        uint8_t* dst = pairs + ((size_t)l * n + gtid_again()) * 64;
        store_packed(dst, x);
        store_packed(dst + 32, y);
    };
    chain_inline(cur, ch.leaf3 != nullptr, LeafAt{ch.leaf3, a.fmt_in}, NoLeafHook(), ch.index, 0, ch.sib, a.lay, a.depth,
                 a.fmt_in, ok, &stash[0][threadIdx.x], store_pair);
    if (ch.root_out) store_fe(g_pc, ch.root_out + gtid_again() * 32, cur, a.fmt_out);
    flag_err(a.err, ok);
}

// the same for few paths: a quad of lanes per path (imt_coop_device.hpp); lane 1 holds the left input, lane 2 the right
__global__ IMT_COOP_WAVES void __launch_bounds__(BLOCK) k_path_pairs_coop(launch::PathChains a) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint32_t tab[coop::TAB_DWORDS];
    coop::tab_fill(tab, g_pc);
    const size_t t = gtid();
    const size_t i = t >> 2;
    if (i >= a.n) return;
    const launch::PathChains::Chain ch = a.c[blockIdx.y];
    const unsigned role = (unsigned)t & 3u, ri = role == 3u ? 0u : role;
    bool ok = true;
    Fe cur, X, C3, o;
    C3 = g_pc.one;
    if (ch.leaf3) {
        ok &= load_fe(g_pc, X, ch.leaf3 + i * 96 + (ri == 2u ? 32 : 0), a.fmt_in);
        ok &= load_fe(g_pc, C3, ch.leaf3 + i * 96 + 64, a.fmt_in);
        if (ri == 1u) ok &= load_fe(g_pc, o, ch.leaf3 + i * 96 + 32, a.fmt_in);
        coop::hash23(tab, o, X, C3, true, ri);
        coop::quad_bcast<1>(cur, o);
    } else {
        ok &= load_fe(g_pc, cur, ch.leaf + i * 32, a.fmt_in);
    }
    const uint64_t idx = ch.index[i];
#pragma unroll 1
    for (unsigned l = 0; l < a.depth; l++) {
        Fe sv;
        ok &= load_fe(g_pc, sv, ch.sib + ((uint64_t)l * a.lay.level_stride + i * a.lay.item_stride) * 32, a.fmt_in);
        const bool right = (idx >> l) & 1;
        const bool take_sv = (ri == 2u) != right;
#pragma unroll
        for (int q = 0; q < NL; q++) X.v[q] = take_sv ? sv.v[q] : cur.v[q];
        if (role == 1u || role == 2u) store_packed(ch.pairs + ((size_t)l * a.n + i) * 64 + (role == 2u ? 32 : 0), X);
        coop::hash23(tab, o, X, C3, false, ri);
        coop::quad_bcast<1>(cur, o);
    }
    if (role == 1u && ch.root_out) store_fe(g_pc, ch.root_out + i * 32, cur, a.fmt_out);
    flag_err(a.err, ok);
#endif
}

// inputs of imt_insert_trace_batch that no caller buffer holds: the rewritten low leaf and the zero-leaf hash
__global__ void __launch_bounds__(BLOCK)
k_insert_trace_inputs(const uint8_t* __restrict__ low_leaf, const uint8_t* __restrict__ new_leaf,
                      const uint64_t* __restrict__ new_index, size_t n, uint8_t* __restrict__ new_low,
                      uint8_t* __restrict__ zero_leaf, unsigned fmt, int* err) {
    const size_t i = gtid();
    if (i >= n) return;
    Fe a, b, c;
    bool ok = load_fe(g_pc, a, low_leaf + (i * 3 + 0) * 32, fmt);
    ok &= load_fe(g_pc, b, new_leaf + (i * 3 + 0) * 32, fmt);
    fe_from_u64(c, new_index[i]);
    store_fe(g_pc, new_low + (i * 3 + 0) * 32, a, fmt);
    store_fe(g_pc, new_low + (i * 3 + 1) * 32, b, fmt);
    store_fe(g_pc, new_low + (i * 3 + 2) * 32, c, fmt);
    store_fe(g_pc, zero_leaf + i * 32, g_pc.zero_leaf, fmt);
    flag_err(err, ok);
}

// ---- a5 / a8 / a9 ----------------------------------------------------------------
__global__ IMT_HASH_WAVES void __launch_bounds__(BLOCK)
k_path_root(const uint8_t* __restrict__ leaf, const uint8_t* __restrict__ leaf3, const uint64_t* __restrict__ index,
            int is_helper, const uint8_t* __restrict__ sib, launch::SibLayout lay, unsigned depth, size_t n,
            uint8_t* __restrict__ root_out, const uint8_t* __restrict__ expect, unsigned expect_stride,
            uint8_t* __restrict__ ok_out, unsigned fmt_in, unsigned fmt_out, int* err) {
    const size_t i0 = gtid();
    if (i0 >= n) return;
    __shared__ uint32_t stash[NL][BLOCK];
    bool ok = true;
    Fe cur;
    if (!leaf3) ok &= load_fe(g_pc, cur, leaf + i0 * 32, fmt_in);
    // helper 1 = left child (src/utils.rs:79): the helper mask is the complement of the index
    chain_inline(cur, leaf3 != nullptr, LeafAt{leaf3, fmt_in}, NoLeafHook(), index, is_helper ? ~(uint64_t)0 : 0, sib, lay, depth,
                 fmt_in, ok, &stash[0][threadIdx.x]);
    const size_t i = gtid_again();
    if (root_out) store_fe(g_pc, root_out + i * 32, cur, fmt_out);
    if (ok_out) {
        Fe e;
        ok &= load_fe(g_pc, e, expect + i * (size_t)expect_stride, fmt_in);
        ok_out[i] = fe_eq(cur, e) ? 1 : 0;
    }
    flag_err(err, ok);
}

// The same for FEW paths: a quad of lanes per path, the latency form of the hash (imt_coop_device.hpp).  A path is
// `depth` (+1) hashes one after the other in one thread -- 13 ms at depth 32 with one thread per path however few
// paths there are (the reference calls verify_proof one proof at a time, src/indexed_merkle_tree.rs:397-400); this
// form takes 0.6x that.  Lane 1 of a quad ends up with every hash and hands it to its neighbours for the next level.
__global__ IMT_COOP_WAVES void __launch_bounds__(BLOCK)
k_path_root_coop(const uint8_t* __restrict__ leaf, const uint8_t* __restrict__ leaf3, const uint64_t* __restrict__ index,
                 int is_helper, const uint8_t* __restrict__ sib, launch::SibLayout lay, unsigned depth, size_t n,
                 uint8_t* __restrict__ root_out, const uint8_t* __restrict__ expect, unsigned expect_stride,
                 uint8_t* __restrict__ ok_out, unsigned fmt_in, unsigned fmt_out, int* err) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint32_t tab[coop::TAB_DWORDS];
    coop::tab_fill(tab, g_pc);
    const size_t t = gtid();
    const size_t i = t >> 2;
    if (i >= n) return;
    const unsigned role = (unsigned)t & 3u, ri = role == 3u ? 0u : role;
    bool ok = true;
    Fe cur, X, C3, o;
    C3 = g_pc.one;
    if (leaf3) {
        ok &= load_fe(g_pc, X, leaf3 + i * 96 + (ri == 2u ? 32 : 0), fmt_in);
        ok &= load_fe(g_pc, C3, leaf3 + i * 96 + 64, fmt_in);
        if (ri == 1u) ok &= load_fe(g_pc, o, leaf3 + i * 96 + 32, fmt_in);     // every element validated once
        coop::hash23(tab, o, X, C3, true, ri);
        coop::quad_bcast<1>(cur, o);
    } else {
        ok &= load_fe(g_pc, cur, leaf + i * 32, fmt_in);
    }
    uint64_t idx = index[i];
    if (is_helper) idx = ~idx;
#pragma unroll 1
    for (unsigned l = 0; l < depth; l++) {
        Fe sv;
        ok &= load_fe(g_pc, sv, sib + ((uint64_t)l * lay.level_stride + i * lay.item_stride) * 32, fmt_in);
        const bool right = (idx >> l) & 1;
        const bool take_sv = (ri == 2u) != right;        // lane 1 holds the left input, lane 2 the right one
#pragma unroll
        for (int q = 0; q < NL; q++) X.v[q] = take_sv ? sv.v[q] : cur.v[q];
        coop::hash23(tab, o, X, C3, false, ri);
        coop::quad_bcast<1>(cur, o);
    }
    if (role == 1u) {
        if (root_out) store_fe(g_pc, root_out + i * 32, cur, fmt_out);
        if (ok_out) {
            Fe e;
            ok &= load_fe(g_pc, e, expect + i * (size_t)expect_stride, fmt_in);
            ok_out[i] = fe_eq(cur, e) ? 1 : 0;
        }
    }
    flag_err(err, ok);
#endif
}

// ---- a13: verify_non_inclusion (src/indexed_merkle_tree.rs:127-229) ---------------
__global__ IMT_HASH_WAVES void __launch_bounds__(BLOCK)
k_non_membership(const uint8_t* __restrict__ root, unsigned root_stride, const uint8_t* __restrict__ low_leaf,
                 const uint64_t* __restrict__ low_index, const uint8_t* __restrict__ sib, launch::SibLayout lay,
                 unsigned depth, const uint8_t* __restrict__ new_val, const uint8_t* __restrict__ is_largest,
                 size_t n, uint8_t* __restrict__ fail_out, uint8_t* __restrict__ root_out, unsigned fmt_in,
                 unsigned fmt_out, int* err) {
    const size_t i0 = gtid();
    if (i0 >= n) return;
    bool ok = true;
    unsigned fail = 0;
    // The hash chain first, the range predicates after it: nothing but `ok` is live across the 33 hashes.
    __shared__ uint32_t stash[NL][BLOCK];
    Fe cur;
    chain_inline(cur, true, LeafAt{low_leaf, fmt_in}, NoLeafHook(), low_index, 0, sib, lay, depth, fmt_in, ok,
                 &stash[0][threadIdx.x]);                           // :193-204
    const size_t i = gtid_again();
    {
        Fe rt;
        ok &= load_fe(g_pc, rt, root + i * (size_t)root_stride, fmt_in);
        if (!fe_eq(cur, rt)) fail |= 0x02;
    }
    if (root_out) store_fe(g_pc, root_out + i * 32, cur, fmt_out);
    fail_out[i] = (uint8_t)fail;
    flag_err(err, ok);
}
// ... and the range predicates of the same call (:143, :180-191, :206-228) as a kernel of their own, after the chain
// kernel: three integer comparisons per item, no hash.  Inside the chain kernel they needed 54 registers of operands on
// top of the inlined hash's budget and spilled; here they have the whole register file.  ORs its bits into fail_out.
__global__ void __launch_bounds__(BLOCK)
k_non_membership_pred(const uint8_t* __restrict__ low_leaf, const uint8_t* __restrict__ new_val, const uint8_t* __restrict__ is_largest,
                      size_t n, uint8_t* __restrict__ fail_out, unsigned fmt_in, int* err) {
    const size_t i = gtid();
    if (i >= n) return;
    bool ok = true;
    unsigned fail = 0;
    Fe v, nx, nv, nvi, lvi, lni;
    ok &= load_fe(g_pc, v, low_leaf + (i * 3 + 0) * 32, fmt_in);
    ok &= load_fe(g_pc, nx, low_leaf + (i * 3 + 1) * 32, fmt_in);
    ok &= load_fe(g_pc, nv, new_val + i * 32, fmt_in);
    to_int(nvi, nv); to_int(lvi, v); to_int(lni, nx);
    const unsigned s = is_largest[i];
    if (s > 1) fail |= 0x80;                                    // assert_bit :41
    const bool is_zero = fe_is_zero(nx);                        // :143
    const bool next_gr = int_lt(nvi, lni);                      // :180
    if (!(s ? is_zero : next_gr)) fail |= 0x01;                 // :182-191
    if (!int_lt(lvi, nvi)) fail |= 0x04;                        // :206-228
    fail_out[i] |= (uint8_t)fail;
    flag_err(err, ok);
}

// the same for few items: a quad of lanes per item (imt_coop_device.hpp) -- one verify_non_inclusion call is a chain of
// 33 dependent hashes, which is all its time
__global__ IMT_COOP_WAVES void __launch_bounds__(BLOCK)
k_non_membership_coop(const uint8_t* __restrict__ root, unsigned root_stride, const uint8_t* __restrict__ low_leaf,
                      const uint64_t* __restrict__ low_index, const uint8_t* __restrict__ sib, launch::SibLayout lay,
                      unsigned depth, const uint8_t* __restrict__ new_val, const uint8_t* __restrict__ is_largest,
                      size_t n, uint8_t* __restrict__ fail_out, uint8_t* __restrict__ root_out, unsigned fmt_in,
                      unsigned fmt_out, int* err) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint32_t tab[coop::TAB_DWORDS];
    coop::tab_fill(tab, g_pc);
    const size_t t = gtid();
    const size_t i = t >> 2;
    if (i >= n) return;
    const unsigned role = (unsigned)t & 3u, ri = role == 3u ? 0u : role;
    bool ok = true;
    unsigned fail = 0;
    {   // the range predicates, on every lane of the quad alike (lane 1 reports)
        Fe v, nx, nv, nvi, lvi, lni;
        ok &= load_fe(g_pc, v, low_leaf + (i * 3 + 0) * 32, fmt_in);
        ok &= load_fe(g_pc, nx, low_leaf + (i * 3 + 1) * 32, fmt_in);
        ok &= load_fe(g_pc, nv, new_val + i * 32, fmt_in);
        to_int(nvi, nv); to_int(lvi, v); to_int(lni, nx);
        const unsigned s = is_largest[i];
        if (s > 1) fail |= 0x80;
        if (!(s ? fe_is_zero(nx) : int_lt(nvi, lni))) fail |= 0x01;
        if (!int_lt(lvi, nvi)) fail |= 0x04;
    }
    Fe cur, X, C3, o;
    ok &= load_fe(g_pc, X, low_leaf + i * 96 + (ri == 2u ? 32 : 0), fmt_in);
    ok &= load_fe(g_pc, C3, low_leaf + i * 96 + 64, fmt_in);
    coop::hash23(tab, o, X, C3, true, ri);
    coop::quad_bcast<1>(cur, o);
    const uint64_t idx = low_index[i];
#pragma unroll 1
    for (unsigned l = 0; l < depth; l++) {
        Fe sv;
        ok &= load_fe(g_pc, sv, sib + ((uint64_t)l * lay.level_stride + i * lay.item_stride) * 32, fmt_in);
        const bool right = (idx >> l) & 1;
        const bool take_sv = (ri == 2u) != right;
#pragma unroll
        for (int q = 0; q < NL; q++) X.v[q] = take_sv ? sv.v[q] : cur.v[q];
        coop::hash23(tab, o, X, C3, false, ri);
        coop::quad_bcast<1>(cur, o);
    }
    if (role == 1u) {
        Fe rt;
        ok &= load_fe(g_pc, rt, root + i * (size_t)root_stride, fmt_in);
        if (!fe_eq(cur, rt)) fail |= 0x02;
        fail_out[i] = (uint8_t)fail;
        if (root_out) store_fe(g_pc, root_out + i * 32, cur, fmt_out);
    }
    flag_err(err, ok);
#endif
}

// ---- a11: 128-bit limb split (src/indexed_merkle_tree.rs:145-178) ------------------
__global__ void __launch_bounds__(BLOCK) k_split128(const uint8_t* __restrict__ vals, uint8_t* __restrict__ q,
                                                    uint8_t* __restrict__ r, size_t n, unsigned fmt, int* err) {
    const size_t i = gtid();
    if (i >= n) return;
    Fe x, xi;
    bool ok = load_fe(g_pc, x, vals + i * 32, fmt);
    to_int(xi, x);                                   // canonical integer, 29-bit limbs
    uint32_t w[8];
    pack(w, xi);
    Fe lo, hi, t;
    const uint32_t wl[8] = {w[0], w[1], w[2], w[3], 0, 0, 0, 0};
    const uint32_t wh[8] = {w[4], w[5], w[6], w[7], 0, 0, 0, 0};
    unpack(t, wl);
    mont_mul(lo, t, g_pc.from_canon);
    canonicalize(lo);
    unpack(t, wh);
    mont_mul(hi, t, g_pc.from_canon);
    canonicalize(hi);
    store_fe(g_pc, r + i * 32, lo, fmt);
    store_fe(g_pc, q + i * 32, hi, fmt);
    flag_err(err, ok);
}

// ---- a14: insert_leaf (src/indexed_merkle_tree.rs:231-314) ------------------------
// blockIdx.y selects one of the four chains of an item, so leaf-hash selection is uniform.
// trace rows (device format): 0 low_leaf_hash, 1 root_from_low, 2 new_low_leaf_hash,
// 3 interim_root, 4 zero_slot_root, 5 new_leaf_hash, 6 new_root_recomputed.
__global__ IMT_HASH_WAVES void __launch_bounds__(BLOCK)
k_insert_chains(const uint8_t* __restrict__ low_leaf, const uint64_t* __restrict__ low_index,
                const uint8_t* __restrict__ low_sib, const uint8_t* __restrict__ new_leaf,
                const uint64_t* __restrict__ new_index, const uint64_t* __restrict__ new_path_index,
                const uint8_t* __restrict__ new_sib, launch::SibLayout lay, unsigned depth, size_t n,
                uint8_t* __restrict__ trace, unsigned fmt_in, int* err) {
    const size_t i = gtid();
    if (i >= n) return;
    const int chain = blockIdx.y;         // wave-uniform: which of the four compute_merkle_root calls
    __shared__ uint32_t stash[NL][BLOCK];
    bool ok = true;
    Fe cur;
    const uint8_t* sib = chain < 2 ? low_sib : new_sib;
    const uint64_t* index = chain < 2 ? low_index : new_path_index;
    // chain 0: low leaf as given :193-204; 1: {low.val, new.val, new_leaf_index} :265-284; 2: the zero leaf at the new
    // slot :286-294 (no leaf hash); 3: the new leaf :299-312.  Trace rows: leaf hash at 0 / 2 / - / 5, root at 1 / 3 / 4 / 6.
    const int leaf_row = chain == 0 ? 0 : chain == 1 ? 2 : 5, root_row = chain == 0 ? 1 : chain == 1 ? 3 : chain == 2 ? 4 : 6;
    if (chain == 2) cur = g_pc.zero_leaf;
    auto load_leaf = [=](Fe& a, Fe& b, Fe& c, bool& okk) {
        const size_t it = gtid_again();
        const uint8_t* const p0 = (chain == 3 ? new_leaf : low_leaf) + it * 96;
        okk &= load_fe(g_pc, a, p0, fmt_in);
        if (chain == 1) {
            okk &= load_fe(g_pc, b, new_leaf + it * 96, fmt_in);
            fe_from_u64(c, new_index[it]);
        } else {
            okk &= load_fe(g_pc, b, p0 + 32, fmt_in);
            okk &= load_fe(g_pc, c, p0 + 64, fmt_in);
        }
    };
    auto store_leaf = [=](const Fe& h) { store_packed(trace + ((size_t)leaf_row * n + gtid_again()) * 32, h); };
    chain_inline(cur, chain != 2, load_leaf, store_leaf, index, 0, sib, lay, depth, fmt_in, ok, &stash[0][threadIdx.x]);
    store_packed(trace + ((size_t)root_row * n + gtid_again()) * 32, cur);
    flag_err(err, ok);
}

// the same for few items: a quad of lanes per (item, chain)
__global__ IMT_COOP_WAVES void __launch_bounds__(BLOCK)
k_insert_chains_coop(const uint8_t* __restrict__ low_leaf, const uint64_t* __restrict__ low_index,
                     const uint8_t* __restrict__ low_sib, const uint8_t* __restrict__ new_leaf,
                     const uint64_t* __restrict__ new_index, const uint64_t* __restrict__ new_path_index,
                     const uint8_t* __restrict__ new_sib, launch::SibLayout lay, unsigned depth, size_t n,
                     uint8_t* __restrict__ trace, unsigned fmt_in, int* err) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint32_t tab[coop::TAB_DWORDS];
    coop::tab_fill(tab, g_pc);
    const size_t t = gtid();
    const size_t i = t >> 2;
    if (i >= n) return;
    const unsigned role = (unsigned)t & 3u, ri = role == 3u ? 0u : role;
    const int chain = blockIdx.y;
    bool ok = true;
    const uint8_t* sib = chain < 2 ? low_sib : new_sib;
    const uint64_t idx = chain < 2 ? low_index[i] : new_path_index[i];
    uint8_t* leaf_out = nullptr;
    uint8_t* root_out;
    Fe cur, X, C3, o;
    C3 = g_pc.one;
    if (chain == 2) {            // the zero leaf at the new slot           :286-294
        cur = g_pc.zero_leaf;
        root_out = trace + (4 * n + i) * 32;
    } else {
        // lane 1: first input, lane 2: second input, C3: third input (absorbed by lane 1)
        if (chain == 1) {        // {low.val, new.val, new_leaf_index}      :265-284
            ok &= load_fe(g_pc, X, (ri == 2u ? new_leaf : low_leaf) + (i * 3 + 0) * 32, fmt_in);
            fe_from_u64(C3, new_index[i]);
            leaf_out = trace + (2 * n + i) * 32;
            root_out = trace + (3 * n + i) * 32;
        } else {                 // a leaf as given: the low leaf :193-204, the new leaf :299-312
            const uint8_t* lf = chain == 0 ? low_leaf : new_leaf;
            ok &= load_fe(g_pc, X, lf + i * 96 + (ri == 2u ? 32 : 0), fmt_in);
            ok &= load_fe(g_pc, C3, lf + i * 96 + 64, fmt_in);
            if (ri == 1u) ok &= load_fe(g_pc, o, lf + i * 96 + 32, fmt_in);       // (every element validated once)
            leaf_out = trace + ((chain == 0 ? 0 : 5) * n + i) * 32;
            root_out = trace + ((chain == 0 ? 1 : 6) * n + i) * 32;
        }
        coop::hash23(tab, o, X, C3, true, ri);
        coop::quad_bcast<1>(cur, o);
        if (role == 1u) store_packed(leaf_out, cur);
    }
#pragma unroll 1
    for (unsigned l = 0; l < depth; l++) {
        Fe sv;
        ok &= load_fe(g_pc, sv, sib + ((uint64_t)l * lay.level_stride + i * lay.item_stride) * 32, fmt_in);
        const bool right = (idx >> l) & 1;
        const bool take_sv = (ri == 2u) != right;
#pragma unroll
        for (int q = 0; q < NL; q++) X.v[q] = take_sv ? sv.v[q] : cur.v[q];
        coop::hash23(tab, o, X, C3, false, ri);
        coop::quad_bcast<1>(cur, o);
    }
    if (role == 1u) store_packed(root_out, cur);
    flag_err(err, ok);
#endif
}

__global__ void __launch_bounds__(BLOCK)
k_insert_check(const uint8_t* __restrict__ old_root, const uint8_t* __restrict__ low_leaf,
               const uint8_t* __restrict__ new_root, const uint8_t* __restrict__ new_leaf,
               const uint8_t* __restrict__ is_largest, size_t n, const uint8_t* __restrict__ trace,
               uint8_t* __restrict__ fail_out, unsigned fmt_in, int* err) {
    const size_t i = gtid();
    if (i >= n) return;
    bool ok = true;
    Fe low[3], nl[3], r0, r1, t;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        ok &= load_fe(g_pc, low[j], low_leaf + (i * 3 + j) * 32, fmt_in);
        ok &= load_fe(g_pc, nl[j], new_leaf + (i * 3 + j) * 32, fmt_in);
    }
    ok &= load_fe(g_pc, r0, old_root + i * 32, fmt_in);
    ok &= load_fe(g_pc, r1, new_root + i * 32, fmt_in);
    unsigned fail = 0;
    Fe nvi, lvi, lni;
    to_int(nvi, nl[0]); to_int(lvi, low[0]); to_int(lni, low[1]);
    const unsigned s = is_largest[i];
    if (s > 1) fail |= 0x80;
    if (!(s ? fe_is_zero(low[1]) : int_lt(nvi, lni))) fail |= 0x01;
    if (!int_lt(lvi, nvi)) fail |= 0x04;
    load_packed(t, trace + (1 * n + i) * 32);
    if (!fe_eq(t, r0)) fail |= 0x02;
    Fe interim, z;
    load_packed(interim, trace + (3 * n + i) * 32);
    load_packed(z, trace + (4 * n + i) * 32);
    if (!fe_eq(z, interim)) fail |= 0x08;
    if (!fe_eq(nl[1], low[1])) fail |= 0x10;
    if (!fe_eq(nl[2], low[2])) fail |= 0x20;
    load_packed(t, trace + (6 * n + i) * 32);
    if (!fe_eq(t, r1)) fail |= 0x40;
    fail_out[i] = (uint8_t)fail;
    flag_err(err, ok);
}

// ---- a2: one level of the dense build (src/utils.rs:43-48) ------------------------
__global__ IMT_HASH_WAVES void __launch_bounds__(BLOCK) k_tree_level(const uint8_t* __restrict__ prev, uint8_t* __restrict__ next,
                                                      size_t n_parents) {
    const size_t i = gtid();
    if (i >= n_parents) return;
    Fe a, b, o;
    load_packed(a, prev + (2 * i) * 32);
    load_packed(b, prev + (2 * i + 1) * 32);
    hash23_stashed(g_pc, o, a, b, false, nullptr, 0);      // inlined: a 2-input hash never reads the stash
    store_packed(next + i * 32, o);
}

// keeps a wave-uniform chain on the vector ALU: hipcc otherwise runs an all-constant chain on the
// scalar unit, where one hash takes ~1.6 ms (64 of them made context creation take 100 ms)
__device__ __forceinline__ void force_vector(Fe& x) {
#pragma unroll
    for (int i = 0; i < NL; i++) asm volatile("" : "+v"(x.v[i]));
}

// Z[0] = H(0,0,0), Z[l+1] = H(Z[l], Z[l]): `depth` hashes one after the other at context creation, so they run in the
// latency form (one quad of lanes, imt_coop_device.hpp): 64 levels in ~15 ms instead of ~26.
__global__ IMT_HASH_WAVES void k_zero_chain(uint8_t* out, unsigned depth) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint32_t tab[coop::TAB_DWORDS];
    coop::tab_fill(tab, g_pc);
    if (blockIdx.x != 0 || threadIdx.x >= 4) return;
    const unsigned role = threadIdx.x, ri = role == 3u ? 0u : role;
    Fe cur = g_pc.zero_leaf;
    force_vector(cur);
    if (role == 1u) store_packed(out, cur);
#pragma unroll 1
    for (unsigned l = 0; l < depth; l++) {
        Fe o;
        coop::hash23(tab, o, cur, cur, false, ri);      // lanes 1 and 2 both hold Z[l]
        coop::quad_bcast<1>(cur, o);
        if (role == 1u) store_packed(out + (size_t)(l + 1) * 32, cur);
    }
#endif
}

__global__ IMT_HASH_WAVES void k_extend_root(uint8_t* cur_io, const uint8_t* zero, unsigned from, unsigned to) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Fe cur;
    load_packed(cur, cur_io);
    force_vector(cur);
#pragma unroll 1
    for (unsigned l = from; l < to; l++) {
        Fe z, o;
        load_packed(z, zero + (size_t)l * 32);
        hash23_stashed(g_pc, o, cur, z, false, nullptr, 0);
        cur = o;
    }
    store_packed(cur_io, cur);
}

// ---- a4: get_proof (src/utils.rs:63-85) as a gather ------------------------------
__global__ void __launch_bounds__(BLOCK) k_gather_proof(launch::TreeView tv, const uint64_t* __restrict__ index,
                                                        size_t n, unsigned depth, uint8_t* __restrict__ out,
                                                        launch::SibLayout lay, unsigned fmt_out) {
    const size_t t = gtid();
    if (t >= n * depth) return;
    const size_t i = t % n;            // items fastest: coalesced level-major stores
    const unsigned l = (unsigned)(t / n);
    const uint64_t s = ((index[i] - tv.index_base) >> l) ^ 1;
    const uint8_t* src = s < tv.len[l] ? tv.nodes + (tv.off[l] + s) * 32 : tv.zero + (size_t)l * 32;
    Fe x;
    load_packed(x, src);
    store_fe(g_pc, out + ((uint64_t)l * lay.level_stride + i * lay.item_stride) * 32, x, fmt_out);
}

__global__ void k_write_helpers(uint64_t index, unsigned depth, uint8_t* out, unsigned fmt_out) {
    const unsigned l = threadIdx.x + blockIdx.x * blockDim.x;
    if (l >= depth) return;
    Fe x;
    fe_from_u64(x, ((index >> l) & 1) ? 0 : 1);
    store_fe(g_pc, out + (size_t)l * 32, x, fmt_out);
}


// ===================================================================================
// a15: batch insertion as a level sweep over time-versioned nodes (imt_sweep.hpp)
// ===================================================================================
__global__ void __launch_bounds__(BLOCK) k_fill_level(uint8_t* __restrict__ nodes, size_t n,
                                                      const uint8_t* __restrict__ zero_l) {
    const size_t i = gtid();
    if (i >= n) return;
    const Word4* z = reinterpret_cast<const Word4*>(zero_l);
    Word4* o = reinterpret_cast<Word4*>(nodes + i * 32);
    o[0] = z[0];
    o[1] = z[1];
}

__global__ void __launch_bounds__(BLOCK) k_merge_level(sweep::LevelTable in, sweep::LevelOut out, uint32_t total) {
    const size_t k = gtid();
    if (k >= total) return;
    sweep::merge_element(in, out, (uint32_t)k, total);
}

// final version of every touched node of level l goes back to the stored tree
__global__ void __launch_bounds__(BLOCK)
k_writeback(const uint8_t* __restrict__ val_l, const uint32_t* __restrict__ from,
            const uint32_t* __restrict__ node_below, uint8_t* __restrict__ tree_l, uint32_t total) {
    const size_t kp = gtid();
    if (kp >= total) return;
    const uint32_t f = from[kp];
    if (!(f & sweep::LAST_BIT)) return;
    const Word4* s = reinterpret_cast<const Word4*>(val_l + (size_t)(f & ~sweep::LAST_BIT) * 32);
    Word4* d = reinterpret_cast<Word4*>(tree_l + (size_t)node_below[kp] * 32);
    d[0] = s[0];
    d[1] = s[1];
}

// -----------------------------------------------------------------------------------------------
// THE hash kernel of a batch insertion: one hash per thread, the hash inlined exactly once.
//   LEAVES  slot k of level 0 = H(preimage of event time0[k])                                   (:662-671)
//   LEVEL   level l -> l+1: one hash2 per event version; the sibling read IS the proof element.  Below l0 the
//           (from, sibsrc, node_below, time_next) tables of imt_sweep.hpp say who meets whom; at and above l0
//           (from == NULL) every event is alone in node 0, its slot is its event id and its sibling the empty
//           subtree of that height.
// Why one kernel: under IMT_PIPELINE two of these launches (consecutive batches, any two phases) share every CU.
// As separate kernels with the hash inlined they would be 47 KB copies evicting each other from the 64 KB
// instruction cache; as separate kernels CALLING a shared hash function (round 1) the call ABI cost 36 B of scratch
// per hash for the second argument plus callee-saved VGPR saves -- 4.6 MB of the 13.3 MB a level launch wrote to HBM
// (profiles/r01_pmc_hbm_traffic.txt).  One kernel is one copy of the code for every phase of every batch in
// flight and has no call.  Round 1 climbed the levels above l0 in a loop inside one launch; the loop-carried state
// on top of the hash's ~90 registers spilled, so those levels are ordinary launches now (same hashes, and
// consecutive batches overlap there level by level too).  The mode is a kernel argument: wave-uniform branches.
// The third input of a leaf hash waits in LDS for the second permutation (hash23_stashed).
// The other, much rarer hash kernels (paths, dense levels, lift) run alone on the chip and inline the hash once each
// too (chain_inline, round 4; rounds 1-3 called a shared function there).
// -----------------------------------------------------------------------------------------------
__global__ IMT_HASH_WAVES void __launch_bounds__(BLOCK) k_sweep(launch::SweepArgs a) {
    __shared__ uint32_t stash[NL][BLOCK];
    const size_t t = gtid();
    if (t >= a.count) return;
    const uint32_t x = a.begin + (uint32_t)t;        // slot (= event id at and above l0)
    Fe A, B, o;
    if (a.mode == launch::SWEEP_LEAVES) {
        const uint8_t* p = a.pre + (size_t)a.time0[x] * 96;
        Fe C;
        bool ok = load_fe(g_pc, A, p, a.fmt_in);
        ok &= load_fe(g_pc, B, p + 32, a.fmt_in);
        ok &= load_fe(g_pc, C, p + 64, a.fmt_in);
#pragma unroll
        for (int i = 0; i < NL; i++) stash[i][threadIdx.x] = C.v[i];
        flag_err(a.err, ok);
    } else {
        uint32_t k = x, n = 0, e = x;
        const uint8_t* sp = a.zero_l;
        if (a.from) {
            k = a.from[x] & ~sweep::LAST_BIT;
            n = a.node_below[x];
            e = a.time_next[x];
            const int32_t ss = a.sibsrc[x];
            const uint64_t sn = (uint64_t)(n ^ 1u);
            sp = ss >= 0 ? a.val_in + (size_t)ss * 32 : (sn < a.len_l ? a.tree_l + sn * 32 : a.zero_l);
        }
        Fe cur, sv;
        load_packed(cur, a.val_in + (size_t)k * 32);
        load_packed(sv, sp);
        if (x == a.last_event && a.node_in) store_packed(a.node_in, cur);
        const bool right = n & 1u;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            A.v[i] = right ? sv.v[i] : cur.v[i];
            B.v[i] = right ? cur.v[i] : sv.v[i];
        }
        uint8_t* row = (e & 1u) ? a.new_sib : a.low_sib;
        if (row) store_fe(g_pc, row + ((uint64_t)a.level * a.lay.level_stride + (uint64_t)(e >> 1) * a.lay.item_stride) * 32, sv, a.fmt_out);
    }
    hash23_stashed(g_pc, o, A, B, a.mode == launch::SWEEP_LEAVES, &stash[0][threadIdx.x], BLOCK);
    store_packed(a.val_out + (size_t)x * 32, o);
    if (x == a.last_event && a.node_out) store_packed(a.node_out, o);
}

// The same launch for SMALL batches: four lanes (a DPP quad) per event, three of them holding one state lane of the
// permutation each (imt_coop_device.hpp).  0.6x the time per launch while the launch fits one wave per SIMD, 2.3x the
// lane-instructions: chosen by the launcher below coop_max_events.  Lane 1 of a quad ends up with the hash.
__global__ IMT_HASH_WAVES void __launch_bounds__(BLOCK) k_sweep_coop(launch::SweepArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ uint32_t tab[coop::TAB_DWORDS];
    coop::tab_fill(tab, g_pc);                       // before anyone leaves: it ends in a barrier
    const size_t t = gtid();
    const size_t q = t >> 2;
    if (q >= a.count) return;                        // whole quads leave together
    const unsigned role = (unsigned)t & 3u, ri = role == 3u ? 0u : role;
    const uint32_t x = a.begin + (uint32_t)q;
    Fe X, C3, o;
    C3 = g_pc.one;                                   // any value: only read for LEAVES
    if (a.mode == launch::SWEEP_LEAVES) {
        const uint8_t* p = a.pre + (size_t)a.time0[x] * 96;
        bool ok = load_fe(g_pc, X, p + (ri == 2u ? 32 : 0), a.fmt_in);     // lane 1: val, lane 2: next_val
        ok &= load_fe(g_pc, C3, p + 64, a.fmt_in);                         // next_idx, absorbed by lane 1
        if (ri == 1u) ok &= load_fe(g_pc, o, p + 32, a.fmt_in);            // (every element validated once)
        flag_err(a.err, ok);
    } else {
        uint32_t k = x, n = 0, e = x;
        const uint8_t* sp = a.zero_l;
        if (a.from) {
            k = a.from[x] & ~sweep::LAST_BIT;
            n = a.node_below[x];
            e = a.time_next[x];
            const int32_t ss = a.sibsrc[x];
            const uint64_t sn = (uint64_t)(n ^ 1u);
            sp = ss >= 0 ? a.val_in + (size_t)ss * 32 : (sn < a.len_l ? a.tree_l + sn * 32 : a.zero_l);
        }
        Fe cur, sv;
        load_packed(cur, a.val_in + (size_t)k * 32);
        load_packed(sv, sp);
        const bool right = n & 1u;
        // lane 1 holds the left input of the hash, lane 2 the right one
        const bool take_sv = (ri == 2u) != right;    // lane 2 & left child, or lane 1 & right child: the sibling
#pragma unroll
        for (int i = 0; i < NL; i++) X.v[i] = take_sv ? sv.v[i] : cur.v[i];
        if (role == 0u) {
            if (x == a.last_event && a.node_in) store_packed(a.node_in, cur);
            uint8_t* row = (e & 1u) ? a.new_sib : a.low_sib;
            if (row) store_fe(g_pc, row + ((uint64_t)a.level * a.lay.level_stride + (uint64_t)(e >> 1) * a.lay.item_stride) * 32, sv, a.fmt_out);
        }
    }
    coop::hash23(tab, o, X, C3, a.mode == launch::SWEEP_LEAVES, ri);
    if (role == 1u) {
        store_packed(a.val_out + (size_t)x * 32, o);
        if (x == a.last_event && a.node_out) store_packed(a.node_out, o);
    }
#endif
}

// Roots of events [e_begin, e_begin + e_count) from the top values (indexed by event id), no hashing: event 2i is
// "low leaf rewritten" (interim root of insertion i), event 2i+1 "new leaf written" (its new root = the old root of
// insertion i+1).  Sharded mode (roots_dev != NULL): device format, one row per event.
__global__ void __launch_bounds__(BLOCK)
k_emit_roots(const uint8_t* __restrict__ val, uint32_t e_begin, uint32_t e_count, uint32_t total,
             uint8_t* __restrict__ old_root, uint8_t* __restrict__ interim_root, uint8_t* __restrict__ new_root,
             unsigned fmt_out, uint8_t* __restrict__ roots_dev, uint8_t* __restrict__ node_store) {
    const size_t t = gtid();
    if (t >= e_count) return;
    const uint32_t e = e_begin + (uint32_t)t;
    Fe cur;
    load_packed(cur, val + (size_t)e * 32);
    if (e == total - 1 && node_store) store_packed(node_store, cur);
    if (roots_dev) { store_packed(roots_dev + (size_t)e * 32, cur); return; }
    const uint32_t i = e >> 1;
    if (e & 1u) {
        if (new_root) store_fe(g_pc, new_root + (size_t)i * 32, cur, fmt_out);
        if (old_root && (size_t)i + 1 < (size_t)(total >> 1)) store_fe(g_pc, old_root + ((size_t)i + 1) * 32, cur, fmt_out);
    } else {
        if (interim_root) store_fe(g_pc, interim_root + (size_t)i * 32, cur, fmt_out);
    }
}

// ---- subtree placement (imt_itree_lift_batch) ---------------------------------------------------
// A tree placed as subtree g of a deeper tree produces subtree-level roots; the enclosing tree's root
// after the same event is `levels` more hash2 up a path whose siblings are the same for the whole batch.
// Jobs: [0, n) interim_root, [n, 2n) new_root (+ old_root[i+1]), then old_root (1 row, or n when new_root
// is absent).
__global__ IMT_HASH_WAVES void __launch_bounds__(BLOCK)
k_lift_roots(uint8_t* __restrict__ old_root, uint8_t* __restrict__ interim_root, uint8_t* __restrict__ new_root,
             uint32_t n, const uint8_t* __restrict__ top, uint64_t pos_bits, unsigned levels, unsigned fmt, int* err) {
    const size_t t = gtid();
    const uint32_t n_int = interim_root ? n : 0, n_new = new_root ? n : 0;
    const uint32_t n_old = old_root ? (new_root ? 1u : n) : 0;
    if (t >= (size_t)n_int + n_new + n_old) return;
    uint8_t* row;
    uint8_t* also = nullptr;
    if (t < n_int) {
        row = interim_root + t * 32;
    } else if (t < (size_t)n_int + n_new) {
        const size_t i = t - n_int;
        row = new_root + i * 32;
        if (old_root && i + 1 < n) also = old_root + (i + 1) * 32;
    } else {
        row = old_root + (t - n_int - n_new) * 32;
    }
    Fe cur;
    bool ok = load_fe(g_pc, cur, row, fmt);
#pragma unroll 1
    for (unsigned j = 0; j < levels; j++) {
        Fe sv, a, b, o;
        load_packed(sv, top + (size_t)j * 32);
        const bool right = (pos_bits >> j) & 1;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            a.v[i] = right ? sv.v[i] : cur.v[i];
            b.v[i] = right ? cur.v[i] : sv.v[i];
        }
        hash23_stashed(g_pc, o, a, b, false, nullptr, 0);
        cur = o;
    }
    store_fe(g_pc, row, cur, fmt);
    if (also) store_fe(g_pc, also, cur, fmt);
    flag_err(err, ok);
}

__global__ void __launch_bounds__(BLOCK)
k_fill_sib_rows(uint8_t* __restrict__ sib, launch::SibLayout lay, unsigned first_level, unsigned levels, uint32_t n,
                const uint8_t* __restrict__ top, unsigned fmt_out) {
    const size_t t = gtid();
    if (t >= (size_t)n * levels) return;
    const uint32_t i = (uint32_t)(t % n);          // items fastest: coalesced level-major stores
    const unsigned j = (unsigned)(t / n);
    Fe x;
    load_packed(x, top + (size_t)j * 32);
    store_fe(g_pc, sib + ((uint64_t)(first_level + j) * lay.level_stride + (uint64_t)i * lay.item_stride) * 32, x, fmt_out);
}

__global__ void k_mix_roots(const uint8_t* __restrict__ before, const uint8_t* __restrict__ after,
                            uint8_t* __restrict__ mixed, uint32_t n_sub, uint32_t self, unsigned fmt_in, int* err) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_sub) return;
    Fe x;
    const bool ok = load_fe(g_pc, x, (r < self ? after : before) + (size_t)r * 32, fmt_in);
    store_packed(mixed + (size_t)r * 32, x);
    flag_err(err, ok);
}

// levels_buf: the dense tree over the mixed roots, level j at offset (2 n_sub - (2 n_sub >> j)) rows
__global__ void k_pick_top(const uint8_t* __restrict__ levels_buf, uint32_t n_sub, uint32_t self, unsigned k,
                           unsigned levels, const uint8_t* __restrict__ zero, unsigned sub_depth,
                           uint8_t* __restrict__ top) {
    const unsigned j = threadIdx.x;
    if (j >= levels) return;
    const uint8_t* src;
    if (j < k) {
        const size_t off = 2 * (size_t)n_sub - ((2 * (size_t)n_sub) >> j);
        src = levels_buf + (off + ((self >> j) ^ 1u)) * 32;
    } else {
        src = zero + (size_t)(sub_depth + j) * 32;
    }
    const Word4* s4 = reinterpret_cast<const Word4*>(src);
    Word4* d4 = reinterpret_cast<Word4*>(top + (size_t)j * 32);
    d4[0] = s4[0];
    d4[1] = s4[1];
}

// ---- sharded single-list batch (imt_itree_batch_*): helpers ------------------------------------
__global__ void __launch_bounds__(BLOCK) k_slot0(const uint32_t* __restrict__ time0, uint32_t* __restrict__ slot0,
                                                 uint32_t total) {
    const size_t k = gtid();
    if (k < total) slot0[time0[k]] = (uint32_t)k;
}

// Proof rows and roots of insertions [ins_begin, ins_begin + ins_count) from the gathered value arrays of
// every level.  One thread per (event, level); the stored tree still holds the pre-batch nodes.
struct ExtractArgs {
    const uint8_t* const* val;          // [l0 + 1] device pointers, level l in level-l slot order
    const uint32_t* slot;               // [l0 + 1][stride]: slot[l][event]
    const int32_t* sibsrc;              // [l0][stride]
    const uint32_t* node_below;         // [l0][stride]
    size_t stride;
    const uint8_t* tree_nodes;
    const uint64_t* tree_off;
    const uint64_t* tree_len;
    const uint8_t* zero;
    const uint8_t* roots;               // [events][32] device format: root after each event
    unsigned l0, depth;
    uint32_t ins_begin, ins_count, n_total;
    uint8_t *old_root, *interim_root, *new_root, *low_sib, *new_sib;   // rows indexed from ins_begin
    launch::SibLayout lay;
    unsigned fmt_out;
};
__global__ void __launch_bounds__(BLOCK) k_extract(ExtractArgs a) {
    const size_t t = gtid();
    const size_t per = (size_t)a.depth + 1;      // depth proof levels + one "roots" task per event
    if (t >= (size_t)a.ins_count * 2 * per) return;
    const uint32_t ev_local = (uint32_t)(t % ((size_t)a.ins_count * 2));   // events fastest: coalesced rows
    const unsigned task = (unsigned)(t / ((size_t)a.ins_count * 2));
    const uint32_t i_local = ev_local >> 1, odd = ev_local & 1u;
    const uint32_t e = (a.ins_begin + i_local) * 2 + odd;
    Fe x;
    if (task == a.depth) {                       // roots
        load_packed(x, a.roots + (size_t)e * 32);
        if (odd) {
            if (a.new_root) store_fe(g_pc, a.new_root + (size_t)i_local * 32, x, a.fmt_out);
        } else {
            if (a.interim_root) store_fe(g_pc, a.interim_root + (size_t)i_local * 32, x, a.fmt_out);
            if (a.old_root) {                    // root before insertion i = after event 2i-1, or the stored root
                Fe o;
                if (e > 0) load_packed(o, a.roots + (size_t)(e - 1) * 32);
                else load_packed(o, a.tree_nodes + a.tree_off[a.depth] * 32);
                store_fe(g_pc, a.old_root + (size_t)i_local * 32, o, a.fmt_out);
            }
        }
        return;
    }
    uint8_t* dst = odd ? a.new_sib : a.low_sib;
    if (!dst) return;
    const unsigned l = task;
    const uint8_t* sp;
    if (l < a.l0) {
        const uint32_t kp = a.slot[(size_t)(l + 1) * a.stride + e];
        const int32_t ss = a.sibsrc[(size_t)l * a.stride + kp];
        const uint64_t sn = (uint64_t)(a.node_below[(size_t)l * a.stride + kp] ^ 1u);
        sp = ss >= 0 ? a.val[l] + (size_t)ss * 32
                     : (sn < a.tree_len[l] ? a.tree_nodes + (a.tree_off[l] + sn) * 32 : a.zero + (size_t)l * 32);
    } else {
        sp = a.zero + (size_t)l * 32;
    }
    load_packed(x, sp);
    store_fe(g_pc, dst + ((uint64_t)l * a.lay.level_stride + (uint64_t)i_local * a.lay.item_stride) * 32, x, a.fmt_out);
}

// ---- time-sliced single list (imt_itree_slice_*): a level's write-back AND the same as a compact payload ----
// What k_writeback writes -- the last version of every node the slice touched at this level -- goes to the stored tree
// and, as (node, value) pairs, into the payload the other replicas apply (one kernel, one pass over the tables),
// packed: level l has at most min(events, nodes of level l) of them, so the payloads the GPUs exchange shrink by half
// per level once a level has fewer nodes than the slice has events.  Order is whatever the atomic counter gives; the
// nodes are distinct.
__global__ void __launch_bounds__(BLOCK)
k_pack_writeback(const uint8_t* __restrict__ val_l, const uint32_t* __restrict__ from, const uint32_t* __restrict__ node_below,
                 uint32_t total, uint8_t* __restrict__ tree_l, uint8_t* __restrict__ out_vals,
                 uint32_t* __restrict__ out_nodes, uint32_t* counter, uint32_t cap) {
    const size_t kp = gtid();
    const uint32_t f = kp < total ? from[kp] : 0u;
    const bool last = (f & sweep::LAST_BIT) != 0;
    // one atomic per wavefront, not per pair: 2^17 increments of one address would serialise in L2 (tens of
    // microseconds per level); the wave's leader reserves a run of slots and every lane takes its rank in the ballot
    const uint64_t mask = __ballot(last);
    if (!last) return;
    const unsigned lane = __lane_id();
    const int leader = __ffsll((unsigned long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    const uint32_t i = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    const Word4* s = reinterpret_cast<const Word4*>(val_l + (size_t)(f & ~sweep::LAST_BIT) * 32);
    const Word4 lo = s[0], hi = s[1];
    const uint32_t node = node_below[kp];
    Word4* t = reinterpret_cast<Word4*>(tree_l + (size_t)node * 32);     // this replica's own write-back (k_writeback)
    t[0] = lo;
    t[1] = hi;
    if (i >= cap) return;                 // cannot happen: cap bounds the nodes of the level (kept as a fence)
    Word4* d = reinterpret_cast<Word4*>(out_vals + (size_t)i * 32);
    d[0] = lo;
    d[1] = hi;
    out_nodes[i] = node;
}
__global__ void __launch_bounds__(BLOCK)
k_apply_packed(const uint8_t* __restrict__ vals, const uint32_t* __restrict__ nodes, const uint32_t* __restrict__ counter,
               uint32_t cap, uint8_t* __restrict__ tree_l, uint64_t len_l) {
    const size_t i = gtid();
    const uint32_t n = *counter < cap ? *counter : cap;
    if (i >= n) return;
    const uint64_t node = nodes[i];
    if (node >= len_l) return;            // a well-formed payload never names a node outside the stored level
    const Word4* s = reinterpret_cast<const Word4*>(vals + i * 32);
    Word4* d = reinterpret_cast<Word4*>(tree_l + node * 32);
    d[0] = s[0];
    d[1] = s[1];
}

// all payloads of one all-gather in ONE launch (blockIdx.y = payload): below its slice's l0 a payload is a list of
// (node, value) pairs for one stored level; at and above l0 it is the one or two nodes the level's launch stored
// (node at l0, node above) and, when the tree is full to its depth, the root
__global__ void __launch_bounds__(BLOCK) k_apply_gathered(launch::ApplyJobs a) {
    if (a.poison && __hip_atomic_load(a.poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
    const launch::ApplyJobs::Job j = a.j[blockIdx.y];
    const size_t i = gtid();
    if (j.pairs) {
        const uint32_t* counter = reinterpret_cast<const uint32_t*>(j.payload + 96);
        const uint32_t n = *counter < j.cap ? *counter : j.cap;
        if (i >= n) return;
        const uint32_t* nodes = reinterpret_cast<const uint32_t*>(j.payload + 128 + (size_t)j.cap * 32);
        const uint64_t node = nodes[i];
        if (node >= j.len_l) return;
        const Word4* s = reinterpret_cast<const Word4*>(j.payload + 128 + i * 32);
        Word4* d = reinterpret_cast<Word4*>(j.tree_l + node * 32);
        d[0] = s[0];
        d[1] = s[1];
        return;
    }
    if (i >= 3) return;
    uint8_t* dst = i == 0 ? j.node_in : (i == 1 ? j.node_out : j.root);
    if (!dst) return;
    const Word4* s = reinterpret_cast<const Word4*>(j.payload + i * 32);
    Word4* d = reinterpret_cast<Word4*>(dst);
    d[0] = s[0];
    d[1] = s[1];
}

__global__ void k_store_top_path(const uint8_t* __restrict__ top_path, uint8_t* __restrict__ tree_nodes,
                                 const uint64_t* __restrict__ tree_off, unsigned l0, unsigned depth) {
    const unsigned l = l0 + threadIdx.x;
    if (l > depth) return;
    const Word4* s = reinterpret_cast<const Word4*>(top_path + (size_t)(l - l0) * 32);
    Word4* d = reinterpret_cast<Word4*>(tree_nodes + tree_off[l] * 32);
    d[0] = s[0];
    d[1] = s[1];
}

// issue-rate probe for the VALU roofline: nothing but independent v_mad_u64_u32 chains
__global__ void __launch_bounds__(256) k_mad_peak(uint32_t* out, uint32_t seed, int iters) {
    uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    uint64_t a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    const uint32_t b = (uint32_t)a0 | 1u, c = seed * 77u + 12345u;
    for (int i = 0; i < iters; i++) {
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
                     "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
                     "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c) : "vcc");
    }
    const uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}

inline unsigned nblk(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

}  // namespace

// ===================================================================================
namespace launch {

hipError_t upload_consts(const dev::PoseidonConsts& pc) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pc), &pc, sizeof(pc), 0, hipMemcpyHostToDevice);
}
hipError_t upload_trace_consts(const dev::TraceConsts& tc) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tc), &tc, sizeof(tc), 0, hipMemcpyHostToDevice);
}
void hash_trace_jobs(hipStream_t s, const TraceJobs& a, unsigned fmt_out) {
    size_t most = 0;
    for (int k = 0; k < a.n_jobs; k++) most = std::max(most, a.j[k].n_items);
    if (!most || a.n_jobs <= 0) return;
    auto* k = fmt_out == FMT_MONT256 ? k_hash_trace<FMT_MONT256>
              : fmt_out == FMT_DEVICE ? k_hash_trace<FMT_DEVICE> : k_hash_trace<FMT_CANONICAL>;
    hipLaunchKernelGGL(k, dim3(nblk(most), (unsigned)a.n_jobs), dim3(BLOCK), 0, s, a);
}
void hash_trace(hipStream_t s, const uint8_t* in, size_t n_items, int arity, uint8_t* trace, size_t n_per, size_t row0,
                size_t rows_total, bool item_major, unsigned fmt_in, unsigned fmt_out, int* err) {
    TraceJobs a{};
    a.j[0] = {in, n_items, arity, row0, fmt_in};
    a.n_jobs = 1;
    a.trace = trace; a.n_per = n_per; a.rows_total = rows_total; a.item_major = item_major ? 1 : 0; a.err = err;
    hash_trace_jobs(s, a, fmt_out);
}
void insert_trace_inputs(hipStream_t s, const uint8_t* low_leaf, const uint8_t* new_leaf, const uint64_t* new_index,
                         size_t n, uint8_t* new_low, uint8_t* zero_leaf, unsigned fmt, int* err) {
    if (!n) return;
    hipLaunchKernelGGL(k_insert_trace_inputs, dim3(nblk(n)), dim3(BLOCK), 0, s, low_leaf, new_leaf, new_index, n, new_low,
                       zero_leaf, fmt, err);
}
void path_pairs(hipStream_t s, const PathChains& a, uint32_t coop_max) {
    if (!a.n || a.n_chains <= 0) return;
    if (a.n * 4 * (size_t)a.n_chains <= coop_max)
        hipLaunchKernelGGL(k_path_pairs_coop, dim3(nblk(a.n * 4), (unsigned)a.n_chains), dim3(BLOCK), 0, s, a);
    else
        hipLaunchKernelGGL(k_path_pairs, dim3(nblk(a.n), (unsigned)a.n_chains), dim3(BLOCK), 0, s, a);
}

void hash_batch(hipStream_t s, const uint8_t* in, uint8_t* out, size_t n, int arity, unsigned fmt_in,
                unsigned fmt_out, int* err, uint32_t coop_max) {
    if (!n) return;
    if (n * 4 <= coop_max)
        hipLaunchKernelGGL(k_hash_batch_coop, dim3(nblk(n * 4)), dim3(BLOCK), 0, s, in, out, n, arity, fmt_in, fmt_out, err);
    else
        hipLaunchKernelGGL(k_hash_batch, dim3(nblk(n)), dim3(BLOCK), 0, s, in, out, n, arity, fmt_in, fmt_out, err);
}
void permute_batch(hipStream_t s, const uint8_t* in, uint8_t* out, size_t n, unsigned fmt_in, unsigned fmt_out,
                   int* err) {
    if (!n) return;
    hipLaunchKernelGGL(k_permute_batch, dim3(nblk(n)), dim3(BLOCK), 0, s, in, out, n, fmt_in, fmt_out, err);
}
void convert(hipStream_t s, const uint8_t* in, uint8_t* out, size_t n, unsigned fmt_in, unsigned fmt_out,
             int* err) {
    if (!n) return;
    hipLaunchKernelGGL(k_convert, dim3(nblk(n)), dim3(BLOCK), 0, s, in, out, n, fmt_in, fmt_out, err);
}
void split128(hipStream_t s, const uint8_t* vals, uint8_t* q, uint8_t* r, size_t n, unsigned fmt, int* err) {
    if (!n) return;
    hipLaunchKernelGGL(k_split128, dim3(nblk(n)), dim3(BLOCK), 0, s, vals, q, r, n, fmt, err);
}
void path_root(hipStream_t s, const uint8_t* leaf, const uint8_t* leaf3, const uint64_t* index, bool is_helper,
               const uint8_t* sib, SibLayout lay, unsigned depth, size_t n, uint8_t* root_out,
               const uint8_t* expect, unsigned expect_stride, uint8_t* ok_out, unsigned fmt_in, unsigned fmt_out,
               int* err, uint32_t coop_max) {
    if (!n) return;
    if (n * 4 <= coop_max)     // few paths: four lanes per path (the same one-wave-per-SIMD budget as the sweep)
        hipLaunchKernelGGL(k_path_root_coop, dim3(nblk(n * 4)), dim3(BLOCK), 0, s, leaf, leaf3, index, is_helper ? 1 : 0,
                           sib, lay, depth, n, root_out, expect, expect_stride, ok_out, fmt_in, fmt_out, err);
    else
        hipLaunchKernelGGL(k_path_root, dim3(nblk(n)), dim3(BLOCK), 0, s, leaf, leaf3, index, is_helper ? 1 : 0, sib,
                           lay, depth, n, root_out, expect, expect_stride, ok_out, fmt_in, fmt_out, err);
}
void non_membership(hipStream_t s, const uint8_t* root, unsigned root_stride, const uint8_t* low_leaf,
                    const uint64_t* low_index, const uint8_t* sib, SibLayout lay, unsigned depth,
                    const uint8_t* new_val, const uint8_t* is_largest, size_t n, uint8_t* fail_out,
                    uint8_t* root_out, unsigned fmt_in, unsigned fmt_out, int* err, uint32_t coop_max) {
    if (!n) return;
    if (n * 4 <= coop_max)
        hipLaunchKernelGGL(k_non_membership_coop, dim3(nblk(n * 4)), dim3(BLOCK), 0, s, root, root_stride, low_leaf,
                           low_index, sib, lay, depth, new_val, is_largest, n, fail_out, root_out, fmt_in, fmt_out, err);
    else {
        hipLaunchKernelGGL(k_non_membership, dim3(nblk(n)), dim3(BLOCK), 0, s, root, root_stride, low_leaf, low_index,
                           sib, lay, depth, new_val, is_largest, n, fail_out, root_out, fmt_in, fmt_out, err);
        hipLaunchKernelGGL(k_non_membership_pred, dim3(nblk(n)), dim3(BLOCK), 0, s, low_leaf, new_val, is_largest, n, fail_out,
                           fmt_in, err);
    }
}
void insert_witness(hipStream_t s, const uint8_t* old_root, const uint8_t* low_leaf, const uint64_t* low_index,
                    const uint8_t* low_sib, const uint8_t* new_root, const uint8_t* new_leaf,
                    const uint64_t* new_index, const uint64_t* new_path_index, const uint8_t* new_sib, SibLayout lay,
                    const uint8_t* is_largest, unsigned depth, size_t n, uint8_t* fail_out, uint8_t* trace,
                    unsigned fmt_in, unsigned fmt_out, int* err, uint32_t coop_max) {
    (void)fmt_out;
    if (!n) return;
    if (n * 16 <= coop_max)
        hipLaunchKernelGGL(k_insert_chains_coop, dim3(nblk(n * 4), 4), dim3(BLOCK), 0, s, low_leaf, low_index, low_sib,
                           new_leaf, new_index, new_path_index, new_sib, lay, depth, n, trace, fmt_in, err);
    else
        hipLaunchKernelGGL(k_insert_chains, dim3(nblk(n), 4), dim3(BLOCK), 0, s, low_leaf, low_index, low_sib, new_leaf,
                           new_index, new_path_index, new_sib, lay, depth, n, trace, fmt_in, err);
    hipLaunchKernelGGL(k_insert_check, dim3(nblk(n)), dim3(BLOCK), 0, s, old_root, low_leaf, new_root, new_leaf,
                       is_largest, n, trace, fail_out, fmt_in, err);
}
void tree_level(hipStream_t s, const uint8_t* prev, uint8_t* next, size_t n_parents, uint32_t coop_max) {
    if (!n_parents) return;
    if (n_parents * 4 <= coop_max)      // a level of a small tree is n_parents two-input hashes in device format
        hipLaunchKernelGGL(k_hash_batch_coop, dim3(nblk(n_parents * 4)), dim3(BLOCK), 0, s, prev, next, n_parents, 2,
                           (unsigned)FMT_DEVICE, (unsigned)FMT_DEVICE, (int*)nullptr);
    else
        hipLaunchKernelGGL(k_tree_level, dim3(nblk(n_parents)), dim3(BLOCK), 0, s, prev, next, n_parents);
}
void zero_chain(hipStream_t s, uint8_t* out, unsigned depth) {
    hipLaunchKernelGGL(k_zero_chain, dim3(1), dim3(64), 0, s, out, depth);
}
void extend_root(hipStream_t s, uint8_t* cur, const uint8_t* zero, unsigned from, unsigned to) {
    hipLaunchKernelGGL(k_extend_root, dim3(1), dim3(64), 0, s, cur, zero, from, to);
}
void gather_proof(hipStream_t s, TreeView tv, const uint64_t* index, size_t n, unsigned depth, uint8_t* out,
                  SibLayout lay, unsigned fmt_out) {
    if (!n || !depth) return;
    hipLaunchKernelGGL(k_gather_proof, dim3(nblk(n * depth)), dim3(BLOCK), 0, s, tv, index, n, depth, out, lay,
                       fmt_out);
}
void write_helpers(hipStream_t s, uint64_t index, unsigned depth, uint8_t* out, unsigned fmt_out) {
    if (!depth) return;
    hipLaunchKernelGGL(k_write_helpers, dim3((depth + 63) / 64), dim3(64), 0, s, index, depth, out, fmt_out);
}


void mad_peak(hipStream_t s, uint32_t* out, unsigned blocks, int iters) {
    hipLaunchKernelGGL(k_mad_peak, dim3(blocks), dim3(256), 0, s, out, 1u, iters);
}
void fill_level(hipStream_t s, uint8_t* nodes, size_t n, const uint8_t* zero_l) {
    if (!n) return;
    hipLaunchKernelGGL(k_fill_level, dim3(nblk(n)), dim3(BLOCK), 0, s, nodes, n, zero_l);
}
// one thread per event, or -- while the launch is small enough to leave most SIMDs idle -- one quad per event
static void launch_sweep(hipStream_t s, const SweepArgs& a, uint32_t coop_max) {
    if (a.count <= coop_max)
        hipLaunchKernelGGL(k_sweep_coop, dim3(nblk((size_t)a.count * 4)), dim3(BLOCK), 0, s, a);
    else
        hipLaunchKernelGGL(k_sweep, dim3(nblk(a.count)), dim3(BLOCK), 0, s, a);
}
void sweep_leaves(hipStream_t s, const uint8_t* pre, const uint32_t* time0, uint8_t* val0, uint32_t k_begin,
                  uint32_t k_count, unsigned fmt_in, int* err, uint32_t coop_max) {
    if (!k_count) return;
    SweepArgs a{};
    a.mode = SWEEP_LEAVES;
    a.begin = k_begin; a.count = k_count;
    a.pre = pre; a.time0 = time0; a.val_out = val0; a.fmt_in = fmt_in; a.err = err;
    a.last_event = 0xffffffffu;
    launch_sweep(s, a, coop_max);
}
void merge_level(hipStream_t s, sweep::LevelTable in, sweep::LevelOut out, uint32_t total) {
    if (!total) return;
    hipLaunchKernelGGL(k_merge_level, dim3(nblk(total)), dim3(BLOCK), 0, s, in, out, total);
}
void sweep_level(hipStream_t s, const uint8_t* val_in, uint8_t* val_out, const uint32_t* from, const int32_t* sibsrc,
                 const uint32_t* node_below, const uint32_t* time_next, const uint8_t* tree_l, uint64_t len_l,
                 const uint8_t* zero_l, uint32_t k_begin, uint32_t k_count, uint8_t* low_sib, uint8_t* new_sib,
                 SibLayout lay, unsigned level, unsigned fmt_out, uint32_t coop_max) {
    if (!k_count) return;
    SweepArgs a{};
    a.mode = SWEEP_LEVEL;
    a.begin = k_begin; a.count = k_count;
    a.val_in = val_in; a.val_out = val_out; a.from = from; a.sibsrc = sibsrc; a.node_below = node_below;
    a.time_next = time_next; a.tree_l = tree_l; a.len_l = len_l; a.zero_l = zero_l; a.level = level;
    a.low_sib = low_sib; a.new_sib = new_sib; a.lay = lay; a.fmt_out = fmt_out;
    a.last_event = 0xffffffffu;
    launch_sweep(s, a, coop_max);
}
void sweep_upper(hipStream_t s, const uint8_t* val_in, uint8_t* val_out, const uint8_t* zero_l, uint32_t e_begin,
                 uint32_t e_count, uint32_t last_event, uint8_t* node_in, uint8_t* node_out, uint8_t* low_sib,
                 uint8_t* new_sib, SibLayout lay, unsigned level, unsigned fmt_out, uint32_t coop_max) {
    if (!e_count) return;
    SweepArgs a{};
    a.mode = SWEEP_LEVEL;
    a.begin = e_begin; a.count = e_count;
    a.val_in = val_in; a.val_out = val_out; a.zero_l = zero_l; a.level = level;
    a.low_sib = low_sib; a.new_sib = new_sib; a.lay = lay; a.fmt_out = fmt_out;
    a.last_event = last_event; a.node_in = node_in; a.node_out = node_out;
    launch_sweep(s, a, coop_max);
}
void emit_roots(hipStream_t s, const uint8_t* val, uint32_t e_begin, uint32_t e_count, uint32_t total, uint8_t* old_root,
                uint8_t* interim_root, uint8_t* new_root, unsigned fmt_out, uint8_t* roots_dev, uint8_t* node_store) {
    if (!e_count) return;
    hipLaunchKernelGGL(k_emit_roots, dim3(nblk(e_count)), dim3(BLOCK), 0, s, val, e_begin, e_count, total, old_root,
                       interim_root, new_root, fmt_out, roots_dev, node_store);
}
void writeback(hipStream_t s, const uint8_t* val_l, const uint32_t* from, const uint32_t* node_below, uint8_t* tree_l,
               uint32_t total) {
    if (!total) return;
    hipLaunchKernelGGL(k_writeback, dim3(nblk(total)), dim3(BLOCK), 0, s, val_l, from, node_below, tree_l, total);
}
void lift_roots(hipStream_t s, uint8_t* old_root, uint8_t* interim_root, uint8_t* new_root, uint32_t n,
                const uint8_t* top, uint64_t pos_bits, unsigned levels, unsigned fmt, int* err) {
    const size_t jobs = (interim_root ? n : 0) + (size_t)(new_root ? n : 0) + (old_root ? (new_root ? 1u : n) : 0);
    if (!jobs) return;
    hipLaunchKernelGGL(k_lift_roots, dim3(nblk(jobs)), dim3(BLOCK), 0, s, old_root, interim_root, new_root, n, top,
                       pos_bits, levels, fmt, err);
}
void fill_sib_rows(hipStream_t s, uint8_t* sib, SibLayout lay, unsigned first_level, unsigned levels, uint32_t n,
                   const uint8_t* top, unsigned fmt_out) {
    if (!sib || !n || !levels) return;
    hipLaunchKernelGGL(k_fill_sib_rows, dim3(nblk((size_t)n * levels)), dim3(BLOCK), 0, s, sib, lay, first_level, levels,
                       n, top, fmt_out);
}
void mix_roots(hipStream_t s, const uint8_t* before, const uint8_t* after, uint8_t* mixed, uint32_t n_sub, uint32_t self,
               unsigned fmt_in, int* err) {
    hipLaunchKernelGGL(k_mix_roots, dim3((n_sub + 63) / 64), dim3(64), 0, s, before, after, mixed, n_sub, self, fmt_in,
                       err);
}
void pick_top(hipStream_t s, const uint8_t* levels_buf, uint32_t n_sub, uint32_t self, unsigned k, unsigned levels,
              const uint8_t* zero, unsigned sub_depth, uint8_t* top) {
    hipLaunchKernelGGL(k_pick_top, dim3(1), dim3(64), 0, s, levels_buf, n_sub, self, k, levels, zero, sub_depth, top);
}
void slot0(hipStream_t s, const uint32_t* time0, uint32_t* slot0_out, uint32_t total) {
    if (!total) return;
    hipLaunchKernelGGL(k_slot0, dim3(nblk(total)), dim3(BLOCK), 0, s, time0, slot0_out, total);
}
void extract(hipStream_t s, const ExtractParams& p) {
    if (!p.ins_count) return;
    ExtractArgs a{p.val, p.slot, p.sibsrc, p.node_below, p.stride, p.tree_nodes, p.tree_off, p.tree_len, p.zero, p.roots,
                  p.l0, p.depth, p.ins_begin, p.ins_count, p.n_total, p.old_root, p.interim_root, p.new_root, p.low_sib,
                  p.new_sib, p.lay, p.fmt_out};
    const size_t threads = (size_t)p.ins_count * 2 * ((size_t)p.depth + 1);
    hipLaunchKernelGGL(k_extract, dim3(nblk(threads)), dim3(BLOCK), 0, s, a);
}
void pack_writeback(hipStream_t s, const uint8_t* val_l, const uint32_t* from, const uint32_t* node_below, uint32_t total,
                    uint8_t* tree_l, uint8_t* out_vals, uint32_t* out_nodes, uint32_t* counter, uint32_t cap, hipEvent_t done) {
    if (!total) return;
    if (done)
        hipExtLaunchKernelGGL(k_pack_writeback, dim3(nblk(total)), dim3(BLOCK), 0, s, nullptr, done, 0, val_l, from, node_below, total,
                              tree_l, out_vals, out_nodes, counter, cap);
    else
        hipLaunchKernelGGL(k_pack_writeback, dim3(nblk(total)), dim3(BLOCK), 0, s, val_l, from, node_below, total, tree_l,
                           out_vals, out_nodes, counter, cap);
}
void apply_packed(hipStream_t s, const uint8_t* vals, const uint32_t* nodes, const uint32_t* counter, uint32_t cap,
                  uint8_t* tree_l, uint64_t len_l) {
    if (!cap) return;
    hipLaunchKernelGGL(k_apply_packed, dim3(nblk(cap)), dim3(BLOCK), 0, s, vals, nodes, counter, cap, tree_l, len_l);
}
void apply_gathered(hipStream_t s, const ApplyJobs& a, hipEvent_t done) {
    if (!a.n_jobs) return;
    uint32_t widest = 3;
    for (int k = 0; k < a.n_jobs; k++)
        if (a.j[k].pairs && a.j[k].cap > widest) widest = a.j[k].cap;
    if (done) hipExtLaunchKernelGGL(k_apply_gathered, dim3(nblk(widest), a.n_jobs), dim3(BLOCK), 0, s, nullptr, done, 0, a);
    else hipLaunchKernelGGL(k_apply_gathered, dim3(nblk(widest), a.n_jobs), dim3(BLOCK), 0, s, a);
}
void store_top_path(hipStream_t s, const uint8_t* top_path, uint8_t* tree_nodes, const uint64_t* tree_off, unsigned l0,
                    unsigned depth) {
    hipLaunchKernelGGL(k_store_top_path, dim3(1), dim3(64), 0, s, top_path, tree_nodes, tree_off, l0, depth);
}

}  // namespace launch
}  // namespace imt
