// imt_sweep.hpp -- batch insertion with sequential semantics as a level sweep over
// time-versioned nodes (SURVEY.md sec. 7 "time-versioned level sweep").
//
// What it replaces: the test module's per-insertion sequence
//     update_idx_leaf -> hash_nullifier_pre_images -> IndexedMerkleTree::new -> get_proof
// (/root/reference/src/indexed_merkle_tree.rs:632-671, :715-735), which rehashes the
// whole tree for every insertion.  Here insertion i contributes two EVENTS,
//     e = 2i   : the low leaf is rewritten          (position low_idx(i))
//     e = 2i+1 : the new leaf is written            (position size + i)
// and the version of node (l, pos >> l) created by event e is
//     H( latest version <= e of the left child , latest version <= e of the right child ).
// One of the children is event e's own version one level down; the other is the newest
// version of the sibling node with time < e, or the stored tree value if there is none.
// Sorting events by (node, time) level by level is a merge of the two children's runs,
// so all of it is index arithmetic (merge_element below), done before any hashing.
// Then level l is 2N independent hashes (k_sweep, one launch per level), and above the highest level at
// which two events can meet every event climbs alone against empty-subtree constants
// (the same kernel, one launch per upper level).  Every sibling read on the way IS the insertion's Merkle proof.
#pragma once
#include <cstdint>
#include "imt_consts.hpp"

#if defined(__HIPCC__)
#define IMT_SW_HD __host__ __device__ __forceinline__
#else
#define IMT_SW_HD inline
#endif

namespace imt {
namespace sweep {

// One level's event table, in the order "sorted by (node at this level, time)".
struct LevelTable {
    const uint32_t* node;   // node index at this level
    const uint32_t* time;   // event id e
    const uint32_t* rs;     // start of the run (same node) containing this slot
    const uint32_t* re;     // end (exclusive) of that run
};
struct LevelOut {
    uint32_t* node;         // next level's table
    uint32_t* time;
    uint32_t* rs;
    uint32_t* re;
    uint32_t* from;         // slot of the same event one level down | (last-of-run << 31)
    int32_t* sibsrc;        // slot (one level down) of the sibling version, or -1 = stored tree
    uint32_t* node_below;   // node index one level down (its low bit = right child)
    uint32_t* slot;         // optional: slot[event] = position of the event in the next level's order
};

constexpr uint32_t LAST_BIT = 0x80000000u;

// Places element k of level l into level l+1.  `total` = number of events.
IMT_SW_HD void merge_element(const LevelTable& in, const LevelOut& out, uint32_t k, uint32_t total) {
    const uint32_t n = in.node[k], a = in.rs[k], b = in.re[k], t = in.time[k];
    uint32_t s0, s1, prs, pre;
    if ((n & 1u) == 0) {           // left child: the sibling run, if any, starts where ours ends
        s0 = s1 = b;
        if (b < total && in.node[b] == n + 1) s1 = in.re[b];
        prs = a;
        pre = s1;
    } else {                       // right child: the sibling run, if any, ends where ours starts
        s0 = s1 = a;
        if (a > 0 && in.node[a - 1] == n - 1) s0 = in.rs[a - 1];
        prs = s0;
        pre = b;
    }
    // r = number of sibling versions older than t (times are unique)
    uint32_t lo = s0, hi = s1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (in.time[mid] < t) lo = mid + 1; else hi = mid;
    }
    const uint32_t r = lo - s0;
    const uint32_t kp = prs + (k - a) + r;
    out.node[kp] = n >> 1;
    out.time[kp] = t;
    out.rs[kp] = prs;
    out.re[kp] = pre;
    out.from[kp] = k | ((k == b - 1) ? LAST_BIT : 0u);
    out.sibsrc[kp] = r > 0 ? (int32_t)(s0 + r - 1) : -1;
    out.node_below[kp] = n;
    if (out.slot) out.slot[t] = kp;
}

}  // namespace sweep
}  // namespace imt
