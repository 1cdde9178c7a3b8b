// imt_prep.hip -- device kernels of the GPU batch preparation (see imt_prep.hpp).  Index and byte
// work only (no field arithmetic): bound by HBM latency, a few hundred microseconds per 2^16 batch.
// Sorting and merging use rocPRIM (AMD's device primitives) with a 256-bit comparator.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "imt_prep.hpp"
#include "imt_prep_logic.hpp"

namespace imt {
namespace prep {
namespace {

constexpr int BLOCK = 256;
inline unsigned nblk(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

struct ValLess {                       // order leaf indices by the value they index
    const uint8_t* val;
    __device__ bool operator()(uint32_t a, uint32_t b) const {
        return lt256(val + (uint64_t)a * 32, val + (uint64_t)b * 32);
    }
};

struct alignas(16) W4 { uint32_t x, y, z, w; };
__device__ __forceinline__ void copy32(uint8_t* dst, const uint8_t* src) {
    const W4* s = reinterpret_cast<const W4*>(src);
    W4* d = reinterpret_cast<W4*>(dst);
    d[0] = s[0];
    d[1] = s[1];
}
__device__ __forceinline__ void zero32(uint8_t* dst) {
    W4* d = reinterpret_cast<W4*>(dst);
    d[0] = W4{0, 0, 0, 0};
    d[1] = W4{0, 0, 0, 0};
}
__device__ __forceinline__ void put_u64(uint8_t* dst, uint64_t v) {
    W4* d = reinterpret_cast<W4*>(dst);
    d[0] = W4{(uint32_t)v, (uint32_t)(v >> 32), 0, 0};
    d[1] = W4{0, 0, 0, 0};
}

// p, little-endian 64-bit limbs
__device__ __forceinline__ bool geq_p(const uint8_t* v) {
    const uint64_t P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    const uint64_t* x = reinterpret_cast<const uint64_t*>(v);
    for (int i = 3; i >= 0; i--)
        if (x[i] != P[i]) return x[i] > P[i];
    return true;
}

__global__ void __launch_bounds__(BLOCK) k_scatter(const uint8_t* __restrict__ vals, uint8_t* __restrict__ d_val,
                                                   uint32_t M, uint32_t n, uint32_t part_mod, uint32_t part_res,
                                                   uint32_t* __restrict__ iota, int* err) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint8_t* v = vals + (uint64_t)i * 32;
    const uint64_t* x = reinterpret_cast<const uint64_t*>(v);
    if (geq_p(v)) atomicOr(err, ERR_NONCANONICAL);
    if ((x[0] | x[1] | x[2] | x[3]) == 0) atomicOr(err, ERR_ZERO);
    if (part_mod > 1 && mod_small(v, part_mod) != part_res) atomicOr(err, ERR_FOREIGN);
    copy32(d_val + (uint64_t)(M + i) * 32, v);
    iota[i] = M + i;
}

__global__ void __launch_bounds__(BLOCK) k_gap(const uint8_t* __restrict__ d_val, const uint32_t* __restrict__ sorted_old,
                                               uint32_t M, const uint32_t* __restrict__ bsorted, uint32_t n,
                                               uint32_t* __restrict__ gap, uint32_t* __restrict__ st0, int* err) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n) return;
    const uint8_t* x = d_val + (uint64_t)bsorted[j] * 32;
    const uint32_t g = count_below(d_val, sorted_old, M, x);
    gap[j] = g;
    st0[j] = bsorted[j] - M;                                  // insertion time of the j-th smallest new value
    if (g < M && eq256(d_val + (uint64_t)sorted_old[g] * 32, x)) atomicOr(err, ERR_DUPLICATE);
    if (j + 1 < n && eq256(d_val + (uint64_t)bsorted[j + 1] * 32, x)) atomicOr(err, ERR_DUPLICATE);
}

__global__ void __launch_bounds__(BLOCK) k_sparse_level(uint32_t* __restrict__ st, uint32_t n, int k) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t w = 1u << k, h = w >> 1;
    if (j >= n || j + w > n) return;
    const uint32_t* prev = st + (uint64_t)(k - 1) * n;
    const uint32_t a = prev[j], b = prev[j + h];
    st[(uint64_t)k * n + j] = a < b ? a : b;
}

// low leaf and successor of every insertion, at the time it is inserted
__global__ void __launch_bounds__(BLOCK) k_neighbours(const uint32_t* __restrict__ st, int levels,
                                                      const uint32_t* __restrict__ bsorted,
                                                      const uint32_t* __restrict__ gap,
                                                      const uint32_t* __restrict__ sorted_old, uint32_t M, uint32_t n,
                                                      uint32_t* __restrict__ low, uint32_t* __restrict__ succ) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n) return;
    const uint32_t g = gap[j], t = st[j];
    const uint32_t jl = nearest_smaller_left(st, n, levels, j);
    const uint32_t jr = nearest_smaller_right(st, n, levels, j);
    // a new neighbour counts only if no stored value lies between (same gap).  g >= 1 because the sentinel 0
    // is stored; g == 0 only for the rejected value 0 (ERR_ZERO is already set), keep the read in bounds.
    low[t] = (jl < n && gap[jl] == g) ? bsorted[jl] : sorted_old[g > 0 ? g - 1 : 0];
    succ[t] = (jr < n && gap[jr] == g) ? bsorted[jr] : (g < M ? sorted_old[g] : NONE);
}

// preimages at every time step (:650-656), event keys, and the hash-free outputs
__global__ void __launch_bounds__(BLOCK)
k_events(const uint8_t* __restrict__ d_val, uint32_t M, uint32_t n, uint64_t base, const uint32_t* __restrict__ low,
         const uint32_t* __restrict__ succ, uint8_t* __restrict__ pre, uint64_t* __restrict__ keys,
         uint64_t* __restrict__ o_low_index, uint8_t* __restrict__ o_is_largest, uint8_t* __restrict__ o_low_leaf,
         uint8_t* __restrict__ o_new_leaf) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t lo = low[i], su = succ[i];
    const uint8_t* v = d_val + (uint64_t)(M + i) * 32;
    const uint8_t* lv = d_val + (uint64_t)lo * 32;
    uint8_t* e0 = pre + (uint64_t)(2 * i) * 96;
    uint8_t* e1 = e0 + 96;
    copy32(e0, lv);                      // low leaf rewritten: {low.val, v, base+M+i}
    copy32(e0 + 32, v);
    put_u64(e0 + 64, base + M + i);      // `base`: the tree is a subtree of a deeper one (imt_itree_set_placement)
    copy32(e1, v);                       // new leaf inherits the low leaf's old pointers
    if (su != NONE) { copy32(e1 + 32, d_val + (uint64_t)su * 32); put_u64(e1 + 64, base + su); }
    else { zero32(e1 + 32); zero32(e1 + 64); }
    keys[2 * i] = ((uint64_t)lo << 32) | (uint64_t)(2 * i);
    keys[2 * i + 1] = ((uint64_t)(M + i) << 32) | (uint64_t)(2 * i + 1);
    if (o_low_index) o_low_index[i] = base + lo;
    if (o_is_largest) o_is_largest[i] = su == NONE ? 1 : 0;
    if (o_low_leaf) {                    // the low leaf BEFORE this insertion: {low.val, succ.val, succ.idx}
        uint8_t* o = o_low_leaf + (uint64_t)i * 96;
        copy32(o, lv);
        copy32(o + 32, e1 + 32);
        copy32(o + 64, e1 + 64);
    }
    if (o_new_leaf) {
        uint8_t* o = o_new_leaf + (uint64_t)i * 96;
        copy32(o, e1);
        copy32(o + 32, e1 + 32);
        copy32(o + 64, e1 + 64);
    }
}

// level-0 tables from the sorted keys: node, time and the run [rs, re) of equal positions
__global__ void __launch_bounds__(BLOCK) k_runs(const uint64_t* __restrict__ keys, uint32_t E, uint32_t* __restrict__ node,
                                                uint32_t* __restrict__ time, uint32_t* __restrict__ rs,
                                                uint32_t* __restrict__ re) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= E) return;
    const uint64_t key = keys[k];
    const uint32_t pos = (uint32_t)(key >> 32);
    node[k] = pos;
    time[k] = (uint32_t)key;
    const uint64_t lo_key = (uint64_t)pos << 32, hi_key = ((uint64_t)pos + 1) << 32;
    uint32_t lo = 0, hi = k;                        // first index with key >= lo_key
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] < lo_key) lo = mid + 1; else hi = mid; }
    rs[k] = lo;
    lo = k + 1; hi = E;                             // first index with key >= hi_key
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] < hi_key) lo = mid + 1; else hi = mid; }
    re[k] = lo;
}

__global__ void __launch_bounds__(BLOCK) k_find_low(const uint8_t* __restrict__ vals, const uint8_t* __restrict__ d_val,
                                                    const uint32_t* __restrict__ sorted, uint32_t M, uint32_t n,
                                                    uint64_t base, uint32_t part_mod, uint32_t part_res,
                                                    uint64_t* __restrict__ low_index, int* err) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint8_t* x = vals + (uint64_t)i * 32;
    if (geq_p(x)) atomicOr(err, ERR_NONCANONICAL);
    if (part_mod > 1 && mod_small(x, part_mod) != part_res) atomicOr(err, ERR_FOREIGN);
    const uint32_t g = count_below(d_val, sorted, M, x);
    if (g == 0) { atomicOr(err, ERR_ZERO); low_index[i] = base; return; }
    if (g < M && eq256(d_val + (uint64_t)sorted[g] * 32, x)) atomicOr(err, ERR_DUPLICATE);
    low_index[i] = base + sorted[g - 1];
}

__global__ void __launch_bounds__(BLOCK)
k_nm_witness(const uint8_t* __restrict__ vals, const uint8_t* __restrict__ d_val, const uint32_t* __restrict__ sorted,
             uint32_t M, uint32_t n, uint64_t base, uint32_t part_mod, uint32_t part_res, uint64_t* __restrict__ low_index,
             uint8_t* __restrict__ low_leaf, uint8_t* __restrict__ is_largest, int* err) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint8_t* x = vals + (uint64_t)i * 32;
    if (geq_p(x)) atomicOr(err, ERR_NONCANONICAL);
    // a value of another subtree's residue is absent from THIS list whether or not the tree holds it
    if (part_mod > 1 && mod_small(x, part_mod) != part_res) atomicOr(err, ERR_FOREIGN);
    uint32_t g = count_below(d_val, sorted, M, x);
    if (g == 0) { atomicOr(err, ERR_ZERO); g = 1; }
    if (g < M && eq256(d_val + (uint64_t)sorted[g] * 32, x)) atomicOr(err, ERR_DUPLICATE);
    const uint32_t lo = sorted[g - 1];
    if (low_index) low_index[i] = base + lo;
    if (is_largest) is_largest[i] = g == M ? 1 : 0;
    if (low_leaf) {        // the stored list: a leaf points at its successor in value order
        uint8_t* o = low_leaf + (uint64_t)i * 96;
        copy32(o, d_val + (uint64_t)lo * 32);
        if (g < M) { copy32(o + 32, d_val + (uint64_t)sorted[g] * 32); put_u64(o + 64, base + sorted[g]); }
        else { zero32(o + 32); zero32(o + 64); }
    }
}

// ---- snapshot: imt_itree_load's list check and imt_itree_get_leaves, on the device ----
struct PreLess {                       // order leaf indices by the val field of their [3][32] preimage
    const uint8_t* pre;
    __device__ bool operator()(uint32_t a, uint32_t b) const {
        return lt256(pre + (uint64_t)a * 96, pre + (uint64_t)b * 96);
    }
};

__global__ void __launch_bounds__(BLOCK) k_load_keys(const uint8_t* __restrict__ pre, uint32_t n, uint32_t part_mod,
                                                     uint32_t part_res, uint64_t* __restrict__ keys,
                                                     uint32_t* __restrict__ idx, int* err) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = pre + (uint64_t)i * 96;
    if (geq_p(p) || geq_p(p + 32) || geq_p(p + 64)) atomicOr(err, ERR_NONCANONICAL);
    if (part_mod > 1 && i > 0 && mod_small(p, part_mod) != part_res) atomicOr(err, ERR_FOREIGN);
    keys[i] = reinterpret_cast<const uint64_t*>(p)[3];
    idx[i] = i;
}

__global__ void __launch_bounds__(BLOCK) k_load_check(const uint8_t* __restrict__ pre, uint32_t n, uint64_t base,
                                                      const uint32_t* __restrict__ idx, int* err, uint32_t* bad) {
    const uint32_t r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    const int e = load_check_rank(pre, n, base, idx, r);
    if (e) {
        atomicOr(err, e);
        if (e & LOAD_LINK) atomicMin(bad, idx[r]);
    }
}

__global__ void __launch_bounds__(BLOCK) k_load_commit(const uint8_t* __restrict__ pre, uint32_t n,
                                                       uint8_t* __restrict__ d_val) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    copy32(d_val + (uint64_t)i * 32, pre + (uint64_t)i * 96);
}

// preimage {val, next_val, next_idx} of leaves: the successor of a stored value is the next one in value order
__global__ void __launch_bounds__(BLOCK) k_leaves(const uint64_t* __restrict__ index, uint64_t first, uint32_t n,
                                                  const uint8_t* __restrict__ d_val, const uint32_t* __restrict__ sorted,
                                                  uint32_t M, uint64_t cap, uint64_t base, uint8_t* __restrict__ out,
                                                  int* err) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint64_t li = (index ? index[i] : first + i) - base;     // wraps below the base: out of range
    uint8_t* o = out + (uint64_t)i * 96;
    if (li >= cap) atomicOr(err, ERR_RANGE);
    if (li >= M) { zero32(o); zero32(o + 32); zero32(o + 64); return; }
    const uint8_t* x = d_val + li * 32;
    const uint32_t r = count_below(d_val, sorted, M, x);            // its rank: sorted[r] == li
    copy32(o, x);
    if (r + 1 < M) { const uint32_t su = sorted[r + 1]; copy32(o + 32, d_val + (uint64_t)su * 32); put_u64(o + 64, base + su); }
    else { zero32(o + 32); zero32(o + 64); }
}

int levels_for(uint32_t n) {
    int k = 1;
    while ((1u << k) <= n) k++;
    return k;   // table rows 0 .. k-1, 2^(k-1) <= n
}

}  // namespace

size_t temp_bytes_needed(size_t n, size_t max_size) {
    size_t a = 0, b = 0, c = 0;
    (void)rocprim::merge_sort(nullptr, a, (uint32_t*)nullptr, (uint32_t*)nullptr, n, ValLess{nullptr}, nullptr);
    (void)rocprim::merge(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, max_size, n,
                         ValLess{nullptr}, nullptr);
    (void)rocprim::radix_sort_keys(nullptr, c, (uint64_t*)nullptr, (uint64_t*)nullptr, 2 * n, 0, 64, nullptr);
    size_t m = a > b ? a : b;
    return (m > c ? m : c) + 256;
}

hipError_t run(hipStream_t s, Workspace& ws, const uint8_t* vals, uint8_t* d_val, const uint32_t* sorted_old,
               uint32_t* sorted_new, uint32_t M, uint32_t n, uint64_t base, uint8_t* pre, uint32_t* node, uint32_t* time,
               uint32_t* rs, uint32_t* re, uint64_t* o_low_index, uint8_t* o_is_largest, uint8_t* o_low_leaf,
               uint8_t* o_new_leaf) {
    const int levels = levels_for(n);
    hipError_t e;
    // the workspace was sized for (cap_n, tree capacity); a larger request must not run into it
    if (n > ws.cap_n || temp_bytes_needed(n, (size_t)M) > ws.tmp_bytes) return hipErrorInvalidValue;
    (void)hipGetLastError();       // the thread's sticky error may be a stale one from an unrelated earlier call
    hipLaunchKernelGGL(k_scatter, dim3(nblk(n)), dim3(BLOCK), 0, s, vals, d_val, M, n, ws.part_mod, ws.part_res, ws.iota,
                       ws.err);
    size_t tb = ws.tmp_bytes;
    if ((e = rocprim::merge_sort(ws.tmp, tb, ws.iota, ws.bsorted, (size_t)n, ValLess{d_val}, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_gap, dim3(nblk(n)), dim3(BLOCK), 0, s, d_val, sorted_old, M, ws.bsorted, n, ws.gap, ws.st,
                       ws.err);
    for (int k = 1; k < levels; k++)
        hipLaunchKernelGGL(k_sparse_level, dim3(nblk(n)), dim3(BLOCK), 0, s, ws.st, n, k);
    hipLaunchKernelGGL(k_neighbours, dim3(nblk(n)), dim3(BLOCK), 0, s, ws.st, levels, ws.bsorted, ws.gap, sorted_old, M,
                       n, ws.low, ws.succ);
    hipLaunchKernelGGL(k_events, dim3(nblk(n)), dim3(BLOCK), 0, s, d_val, M, n, base, ws.low, ws.succ, pre, ws.keys,
                       o_low_index, o_is_largest, o_low_leaf, o_new_leaf);
    tb = ws.tmp_bytes;
    if ((e = rocprim::radix_sort_keys(ws.tmp, tb, ws.keys, ws.keys_sorted, (size_t)2 * n, 0, 64, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_runs, dim3(nblk(2 * (size_t)n)), dim3(BLOCK), 0, s, ws.keys_sorted, 2 * n, node, time, rs, re);
    tb = ws.tmp_bytes;
    if ((e = rocprim::merge(ws.tmp, tb, sorted_old, ws.bsorted, sorted_new, (size_t)M, (size_t)n, ValLess{d_val}, s)) !=
        hipSuccess)
        return e;
    return hipGetLastError();      // a failed launch anywhere in the sequence
}

hipError_t index_only(hipStream_t s, Workspace& ws, const uint8_t* vals, uint8_t* d_val, const uint32_t* sorted_old,
                      uint32_t* sorted_new, uint32_t M, uint32_t n) {
    hipError_t e;
    if (n > ws.cap_n || temp_bytes_needed(n, (size_t)M) > ws.tmp_bytes) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_scatter, dim3(nblk(n)), dim3(BLOCK), 0, s, vals, d_val, M, n, ws.part_mod, ws.part_res, ws.iota,
                       ws.err);
    size_t tb = ws.tmp_bytes;
    if ((e = rocprim::merge_sort(ws.tmp, tb, ws.iota, ws.bsorted, (size_t)n, ValLess{d_val}, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_gap, dim3(nblk(n)), dim3(BLOCK), 0, s, d_val, sorted_old, M, ws.bsorted, n, ws.gap, ws.st,
                       ws.err);
    tb = ws.tmp_bytes;
    if ((e = rocprim::merge(ws.tmp, tb, sorted_old, ws.bsorted, sorted_new, (size_t)M, (size_t)n, ValLess{d_val}, s)) !=
        hipSuccess)
        return e;
    return hipGetLastError();
}

void nm_witness(hipStream_t s, const uint8_t* vals, const uint8_t* d_val, const uint32_t* sorted, uint32_t M, uint32_t n,
                uint64_t base, uint32_t part_mod, uint32_t part_res, uint64_t* low_index, uint8_t* low_leaf,
                uint8_t* is_largest, int* err) {
    if (!n) return;
    hipLaunchKernelGGL(k_nm_witness, dim3(nblk(n)), dim3(BLOCK), 0, s, vals, d_val, sorted, M, n, base, part_mod, part_res,
                       low_index, low_leaf, is_largest, err);
}

void find_low(hipStream_t s, const uint8_t* vals, const uint8_t* d_val, const uint32_t* sorted, uint32_t M, uint32_t n,
              uint64_t base, uint32_t part_mod, uint32_t part_res, uint64_t* low_index, int* err) {
    if (!n) return;
    hipLaunchKernelGGL(k_find_low, dim3(nblk(n)), dim3(BLOCK), 0, s, vals, d_val, sorted, M, n, base, part_mod, part_res,
                       low_index, err);
}

size_t load_ws_bytes(size_t n) {
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                                    (uint32_t*)nullptr, n, 0, 64, nullptr);
    (void)rocprim::merge_sort(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, n, PreLess{nullptr}, nullptr);
    const size_t r = (n + 63) / 64 * 64;
    return 2 * r * 8 + 2 * r * 4 + 256 + (a > b ? a : b) + 256;
}

hipError_t load_check(hipStream_t s, const uint8_t* pre, uint32_t n, uint64_t base, uint32_t part_mod, uint32_t part_res,
                      void* ws, size_t ws_bytes, bool full_sort, int* err, uint32_t** bad, const uint32_t** sorted) {
    if (load_ws_bytes(n) > ws_bytes) return hipErrorInvalidValue;
    const size_t r = ((size_t)n + 63) / 64 * 64;
    uint8_t* w = (uint8_t*)ws;
    uint64_t* keys = (uint64_t*)w;
    uint64_t* keys2 = keys + r;
    uint32_t* idx = (uint32_t*)(keys2 + r);
    uint32_t* idx2 = idx + r;
    *bad = idx2 + r;
    void* tmp = (uint8_t*)(*bad) + 256;
    size_t tb = ws_bytes - (size_t)((uint8_t*)tmp - w);
    hipError_t e;
    (void)hipGetLastError();
    if ((e = hipMemsetAsync(*bad, 0xff, 4, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_load_keys, dim3(nblk(n)), dim3(BLOCK), 0, s, pre, n, part_mod, part_res, keys, idx, err);
    if (full_sort) e = rocprim::merge_sort(tmp, tb, idx, idx2, (size_t)n, PreLess{pre}, s);
    else e = rocprim::radix_sort_pairs(tmp, tb, keys, keys2, idx, idx2, (size_t)n, 0, 64, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_load_check, dim3(nblk(n)), dim3(BLOCK), 0, s, pre, n, base, idx2, err, *bad);
    *sorted = idx2;
    return hipGetLastError();
}

void load_commit(hipStream_t s, const uint8_t* pre, uint32_t n, uint8_t* d_val) {
    if (!n) return;
    hipLaunchKernelGGL(k_load_commit, dim3(nblk(n)), dim3(BLOCK), 0, s, pre, n, d_val);
}

void leaves(hipStream_t s, const uint64_t* index, uint64_t first, uint32_t n, const uint8_t* d_val, const uint32_t* sorted,
            uint32_t M, uint64_t cap, uint64_t base, uint8_t* out, int* err) {
    if (!n) return;
    hipLaunchKernelGGL(k_leaves, dim3(nblk(n)), dim3(BLOCK), 0, s, index, first, n, d_val, sorted, M, cap, base, out, err);
}

}  // namespace prep
}  // namespace imt
