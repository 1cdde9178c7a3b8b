// imt_prep_logic.hpp -- per-element logic of the GPU batch preparation (imt_prep.hip), kept free of
// HIP types so tests/native/ can run it on the host against a brute force.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define IMT_PL_HD __host__ __device__ __forceinline__
#else
#define IMT_PL_HD inline
#endif

namespace imt {
namespace prep {

// 256-bit little-endian integers stored as 32 bytes
IMT_PL_HD bool lt256(const uint8_t* a, const uint8_t* b) {
    const uint64_t* x = reinterpret_cast<const uint64_t*>(a);
    const uint64_t* y = reinterpret_cast<const uint64_t*>(b);
    for (int i = 3; i >= 0; i--)
        if (x[i] != y[i]) return x[i] < y[i];
    return false;
}
IMT_PL_HD bool eq256(const uint8_t* a, const uint8_t* b) {
    const uint64_t* x = reinterpret_cast<const uint64_t*>(a);
    const uint64_t* y = reinterpret_cast<const uint64_t*>(b);
    return x[0] == y[0] && x[1] == y[1] && x[2] == y[2] && x[3] == y[3];
}

// (256-bit little-endian integer) mod m, m < 2^32
IMT_PL_HD uint32_t mod_small(const uint8_t* a, uint32_t m) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(a);
    uint64_t r = 0;
    for (int i = 7; i >= 0; i--) r = ((r << 32) | w[i]) % m;
    return (uint32_t)r;
}

// number of stored values (sorted[0..M) indexes val) strictly below x
IMT_PL_HD uint32_t count_below(const uint8_t* val, const uint32_t* sorted, uint32_t M, const uint8_t* x) {
    uint32_t lo = 0, hi = M;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (lt256(val + (uint64_t)sorted[mid] * 32, x)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Sparse min-table st[k][j] = min(t[j .. j + 2^k)), rows of length n (entries with j + 2^k > n unused).
// Nearest position left of j whose value is < t[j]; returns n ("none") if there is none.
IMT_PL_HD uint32_t nearest_smaller_left(const uint32_t* st, uint32_t n, int levels, uint32_t j) {
    const uint32_t t = st[j];
    uint32_t p = j;                       // invariant: every entry in [p, j) is > t
    for (int k = levels - 1; k >= 0; k--) {
        const uint32_t w = 1u << k;
        if (p >= w && st[(uint64_t)k * n + (p - w)] > t) p -= w;
    }
    return p > 0 ? p - 1 : n;
}
IMT_PL_HD uint32_t nearest_smaller_right(const uint32_t* st, uint32_t n, int levels, uint32_t j) {
    const uint32_t t = st[j];
    uint32_t p = j + 1;                   // invariant: every entry in (j, p) is > t
    for (int k = levels - 1; k >= 0; k--) {
        const uint32_t w = 1u << k;
        if (p + w <= n && st[(uint64_t)k * n + p] > t) p += w;
    }
    return p < n ? p : n;
}

// ---- snapshot check (imt_itree_load): the leaves, visited in value order, must be ONE linked list ----------------
// pre = [n][3][32] canonical {val, next_val, next_idx}; idx[r] = leaf index of the r-th smallest val (candidate order:
// sorted by the top 64 bits at least).  Rank r checks leaf idx[r] against its successor idx[r + 1]: strictly larger
// val (equal: duplicate; smaller under an equal top limb: the candidate order was too coarse, LOAD_TIE -> the caller
// re-sorts with the full comparator), next_val = the successor's val, next_idx = base + its index; the last leaf points
// to {0, 0}; rank 0 is leaf 0 = the {0,..} sentinel (the reference's first leaf, src/indexed_merkle_tree.rs:710-713).
constexpr int LOAD_SENTINEL = 16, LOAD_LINK = 32, LOAD_LAST = 64, LOAD_TIE = 128, LOAD_DUP = 4;
IMT_PL_HD bool is_u64_256(const uint8_t* a, uint64_t v) {
    const uint64_t* x = reinterpret_cast<const uint64_t*>(a);
    return x[0] == v && (x[1] | x[2] | x[3]) == 0;
}
IMT_PL_HD int load_check_rank(const uint8_t* pre, uint32_t n, uint64_t base, const uint32_t* idx, uint32_t r) {
    const uint32_t i = idx[r];
    const uint8_t* p = pre + (uint64_t)i * 96;
    int e = 0;
    if (r == 0 && (i != 0 || !is_u64_256(p, 0))) e |= LOAD_SENTINEL;
    if (r + 1 < n) {
        const uint32_t j = idx[r + 1];
        const uint8_t* q = pre + (uint64_t)j * 96;
        if (eq256(p, q)) return e | LOAD_DUP;
        if (!lt256(p, q)) return e | LOAD_TIE;
        if (!eq256(p + 32, q) || !is_u64_256(p + 64, base + j)) e |= LOAD_LINK;
    } else if (!is_u64_256(p + 32, 0) || !is_u64_256(p + 64, 0)) {
        e |= LOAD_LAST;
    }
    return e;
}

}  // namespace prep
}  // namespace imt
