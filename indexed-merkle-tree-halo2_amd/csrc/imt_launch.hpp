// imt_launch.hpp -- host-callable launchers of the gfx950 kernels (imt_kernels.hip).
// Everything is asynchronous on `stream`; pointers are device pointers.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdint>
#include "imt_consts.hpp"
#include "imt_sweep.hpp"

namespace imt {
namespace launch {

// sibling array addressing in units of 32-byte elements:
//   element(level, item) = sib + (level * level_stride + item * item_stride) * 32
struct SibLayout {
    uint64_t level_stride, item_stride;
};

// description of a stored tree: level l has `len[l]` nodes at nodes + off[l]*32; nodes
// outside [0, len[l]) read as zero[l] (the empty-subtree hash of that height)
struct TreeView {
    const uint8_t* nodes;        // device format
    const uint64_t* off;         // [depth+1] device array
    const uint64_t* len;         // [depth+1] device array
    const uint8_t* zero;         // [depth+1][32] device format, may be NULL when never needed
    uint64_t index_base = 0;     // leaf indices handed to gather_proof are global: local = index - index_base
};

hipError_t upload_consts(const dev::PoseidonConsts& pc);
hipError_t upload_trace_consts(const dev::TraceConsts& tc);

// f1: every witness of hash_fix_len_array for n_items hashes of `arity` inputs.  The items form blocks of n_per (the
// levels of a path; a single block otherwise); block l occupies rows [row0 + l * rows, ...) of a trace of rows_total
// rows per item column, row-major [rows_total][n_per] or item-major [n_per][rows_total].
void hash_trace(hipStream_t s, const uint8_t* in, size_t n_items, int arity, uint8_t* trace, size_t n_per, size_t row0,
                size_t rows_total, bool item_major, unsigned fmt_in, unsigned fmt_out, int* err);
// The same for up to eight groups of hashes in ONE launch (blockIdx.y = group): the 3 leaf hashes and 4 paths of an
// insert_leaf call do not depend on each other once the path inputs are known, and for a few items one thread per hash
// leaves the chip empty -- seven launches in a row would cost seven times one hash's 0.9 ms.
struct TraceJobs {
    struct Job {
        const uint8_t* in;       // [n_items][arity][32]
        size_t n_items;          // item q -> block q / n_per (rows row0 + (q / n_per) * rows(arity) ...) of column q % n_per
        int arity;
        size_t row0;
        unsigned fmt_in;
    } j[8];
    int n_jobs;
    uint8_t* trace;
    size_t n_per, rows_total;
    int item_major;
    int* err;
};
void hash_trace_jobs(hipStream_t s, const TraceJobs& a, unsigned fmt_out);
// {low.val, new.val, new_index} per item -> new_low [n][3][32], and the zero-leaf hash -> zero_leaf [n][32] (fmt)
void insert_trace_inputs(hipStream_t s, const uint8_t* low_leaf, const uint8_t* new_leaf, const uint64_t* new_index,
                         size_t n, uint8_t* new_low, uint8_t* zero_leaf, unsigned fmt, int* err);
// the (left, right) inputs of every hash2 along n paths, for up to four chains in one launch:
// chain k writes pairs[depth][n][2][32] (device format) and optionally its roots
struct PathChains {
    struct Chain {
        const uint8_t* leaf;      // [n][32], or
        const uint8_t* leaf3;     // [n][3][32]: the chain starts from H(leaf3)
        const uint64_t* index;
        const uint8_t* sib;
        uint8_t* pairs;
        uint8_t* root_out;        // or NULL
    } c[4];
    int n_chains;
    SibLayout lay;
    unsigned depth;
    size_t n;
    unsigned fmt_in, fmt_out;
    int* err;
};
void path_pairs(hipStream_t s, const PathChains& a, uint32_t coop_max);

void hash_batch(hipStream_t s, const uint8_t* in, uint8_t* out, size_t n, int arity, unsigned fmt_in,
                unsigned fmt_out, int* err, uint32_t coop_max = 0);
void permute_batch(hipStream_t s, const uint8_t* in, uint8_t* out, size_t n, unsigned fmt_in,
                   unsigned fmt_out, int* err);
void convert(hipStream_t s, const uint8_t* in, uint8_t* out, size_t n, unsigned fmt_in,
             unsigned fmt_out, int* err);

// `blocks` x 256 threads, each 8 x `iters` multiply-adds
void mad_peak(hipStream_t s, uint32_t* out, unsigned blocks, int iters);
void split128(hipStream_t s, const uint8_t* vals, uint8_t* q, uint8_t* r, size_t n, unsigned fmt, int* err);

// Path recompute.  leaf3 != NULL: the start value is hash3(leaf3[i]) ([n][3][32]);
// otherwise leaf[i].  index bit l (or ~helper_mask bit l when is_helper) = node is a right
// child at level l.  expect (optional, stride 0 or 32 bytes) -> ok_out[i].
void path_root(hipStream_t s, const uint8_t* leaf, const uint8_t* leaf3, const uint64_t* index,
               bool is_helper, const uint8_t* sib, SibLayout lay, unsigned depth, size_t n,
               uint8_t* root_out, const uint8_t* expect, unsigned expect_stride, uint8_t* ok_out,
               unsigned fmt_in, unsigned fmt_out, int* err, uint32_t coop_max = 0);

void non_membership(hipStream_t s, const uint8_t* root, unsigned root_stride, const uint8_t* low_leaf,
                    const uint64_t* low_index, const uint8_t* sib, SibLayout lay, unsigned depth,
                    const uint8_t* new_val, const uint8_t* is_largest, size_t n, uint8_t* fail_out,
                    uint8_t* root_out, unsigned fmt_in, unsigned fmt_out, int* err, uint32_t coop_max);

// 4 chains per item; trace [7][n][32] (device buffer, required), then the check kernel
void insert_witness(hipStream_t s, const uint8_t* old_root, const uint8_t* low_leaf,
                    const uint64_t* low_index, const uint8_t* low_sib, const uint8_t* new_root,
                    const uint8_t* new_leaf, const uint64_t* new_index, const uint64_t* new_path_index,
                    const uint8_t* new_sib, SibLayout lay, const uint8_t* is_largest, unsigned depth, size_t n,
                    uint8_t* fail_out, uint8_t* trace, unsigned fmt_in, unsigned fmt_out, int* err, uint32_t coop_max);

// next[i] = hash2(prev[2i], prev[2i+1]), device format
void tree_level(hipStream_t s, const uint8_t* prev, uint8_t* next, size_t n_parents, uint32_t coop_max = 0);
// chain: out[l+1] = hash2(out[l], out[l]) for l < depth, out[0] = H(0,0,0); one thread
void zero_chain(hipStream_t s, uint8_t* out, unsigned depth);
// cur = hash2(cur, zero[l]) for l in [from, to): extends a subtree root up the left spine
void extend_root(hipStream_t s, uint8_t* cur, const uint8_t* zero, unsigned from, unsigned to);

// siblings of `index[i]` at every level of a stored tree
void gather_proof(hipStream_t s, TreeView tv, const uint64_t* index, size_t n, unsigned depth,
                  uint8_t* out, SibLayout lay, unsigned fmt_out);
// helper[l] = 1 iff (index >> l) is even, as field elements (get_proof, src/utils.rs:79)
void write_helpers(hipStream_t s, uint64_t index, unsigned depth, uint8_t* out, unsigned fmt_out);


// ---- batch insertion level sweep (imt_sweep.hpp) ----
// argument block of k_sweep, the one hash kernel behind sweep_leaves / sweep_level / sweep_upper
enum : int { SWEEP_LEAVES = 0, SWEEP_LEVEL = 1 };
struct SweepArgs {
    int mode;
    uint32_t begin, count;                // slots [begin, begin + count)
    // LEAVES
    const uint8_t* pre;
    const uint32_t* time0;
    unsigned fmt_in;
    int* err;
    // LEVEL
    const uint8_t* val_in;
    uint8_t* val_out;                     // (LEAVES writes it too)
    const uint32_t* from;                 // NULL at and above l0: slot = event, node 0, sibling = zero_l
    const int32_t* sibsrc;
    const uint32_t* node_below;
    const uint32_t* time_next;
    const uint8_t* tree_l;
    uint64_t len_l;
    const uint8_t* zero_l;
    unsigned level;
    uint32_t last_event;                  // the event whose node goes back to the stored tree (above l0), or ~0
    uint8_t *node_in, *node_out;          // where its input / output value is stored (device format), or NULL
    uint8_t *low_sib, *new_sib;           // proof rows
    SibLayout lay;
    unsigned fmt_out;
};
void fill_level(hipStream_t s, uint8_t* nodes, size_t n, const uint8_t* zero_l);
// coop_max: launches of at most this many events use the latency form of the kernel (a quad of lanes per event,
// imt_coop_device.hpp); 0 = never
void sweep_leaves(hipStream_t s, const uint8_t* pre, const uint32_t* time0, uint8_t* val0, uint32_t k_begin,
                  uint32_t k_count, unsigned fmt_in, int* err, uint32_t coop_max);
void merge_level(hipStream_t s, sweep::LevelTable in, sweep::LevelOut out, uint32_t total);
// hashes slots [k_begin, k_begin + k_count) of level `level`+1 (a rank's share when sharded)
void sweep_level(hipStream_t s, const uint8_t* val_in, uint8_t* val_out, const uint32_t* from, const int32_t* sibsrc,
                 const uint32_t* node_below, const uint32_t* time_next, const uint8_t* tree_l, uint64_t len_l,
                 const uint8_t* zero_l, uint32_t k_begin, uint32_t k_count, uint8_t* low_sib, uint8_t* new_sib,
                 SibLayout lay, unsigned level, unsigned fmt_out, uint32_t coop_max);
void writeback(hipStream_t s, const uint8_t* val_l, const uint32_t* from, const uint32_t* node_below, uint8_t* tree_l,
               uint32_t total);
// level l >= l0 for events [e_begin, e_begin + e_count): val indexed by event id, sibling = zero_l; the values
// of `last_event` go to node_in (its input, i.e. the node at level l) / node_out (the node at level l + 1)
void sweep_upper(hipStream_t s, const uint8_t* val_in, uint8_t* val_out, const uint8_t* zero_l, uint32_t e_begin,
                 uint32_t e_count, uint32_t last_event, uint8_t* node_in, uint8_t* node_out, uint8_t* low_sib,
                 uint8_t* new_sib, SibLayout lay, unsigned level, unsigned fmt_out, uint32_t coop_max);
// roots per event from the top values (no hashing); sharded mode: roots_dev[e] in device format
void emit_roots(hipStream_t s, const uint8_t* val, uint32_t e_begin, uint32_t e_count, uint32_t total, uint8_t* old_root,
                uint8_t* interim_root, uint8_t* new_root, unsigned fmt_out, uint8_t* roots_dev, uint8_t* node_store);

// ---- subtree placement: lift subtree-level witnesses to the depth of the enclosing tree ----
// top[j] (device format, j < levels) = sibling of this subtree's ancestor at height sub_depth + j;
// bit j of pos_bits = that ancestor is a RIGHT child.  Roots are lifted in place (format fmt):
// interim_root[i], new_root[i] and old_root[i] climb `levels` hashes each; with new_root and old_root
// both given, old_root[i + 1] is written from new_root[i] and only old_root[0] climbs on its own.
void lift_roots(hipStream_t s, uint8_t* old_root, uint8_t* interim_root, uint8_t* new_root, uint32_t n,
                const uint8_t* top, uint64_t pos_bits, unsigned levels, unsigned fmt, int* err);
// rows [first_level, first_level + levels) of a sibling array = top[j], for all n items
void fill_sib_rows(hipStream_t s, uint8_t* sib, SibLayout lay, unsigned first_level, unsigned levels, uint32_t n,
                   const uint8_t* top, unsigned fmt_out);
// mixed[r] = r < self ? after[r] : before[r]   (device format out, fmt_in in)
void mix_roots(hipStream_t s, const uint8_t* before, const uint8_t* after, uint8_t* mixed, uint32_t n_sub, uint32_t self,
               unsigned fmt_in, int* err);
// top[j] = node (j, (self >> j) ^ 1) of the tree over `mixed` for j < k, then zero[sub_depth + j] for j >= k
void pick_top(hipStream_t s, const uint8_t* levels_buf, uint32_t n_sub, uint32_t self, unsigned k, unsigned levels,
              const uint8_t* zero, unsigned sub_depth, uint8_t* top);

// ---- sharded single-list batch (imt_itree_batch_*) ----
void slot0(hipStream_t s, const uint32_t* time0, uint32_t* slot0_out, uint32_t total);
struct ExtractParams {
    const uint8_t* const* val;   // device array of l0 + 1 device pointers
    const uint32_t* slot;
    const int32_t* sibsrc;
    const uint32_t* node_below;
    size_t stride;
    const uint8_t* tree_nodes;
    const uint64_t* tree_off;
    const uint64_t* tree_len;
    const uint8_t* zero;
    const uint8_t* roots;
    unsigned l0, depth;
    uint32_t ins_begin, ins_count, n_total;
    uint8_t *old_root, *interim_root, *new_root, *low_sib, *new_sib;
    SibLayout lay;
    unsigned fmt_out;
};
void extract(hipStream_t s, const ExtractParams& p);
// time-sliced single list: the (node, value) pairs k_writeback would write, packed behind an atomic counter (at most
// `cap` of them), and their application to a replica's stored level
void pack_writeback(hipStream_t s, const uint8_t* val_l, const uint32_t* from, const uint32_t* node_below, uint32_t total,
                    uint8_t* tree_l, uint8_t* out_vals, uint32_t* out_nodes, uint32_t* counter, uint32_t cap,
                    hipEvent_t done = nullptr);
void apply_packed(hipStream_t s, const uint8_t* vals, const uint32_t* nodes, const uint32_t* counter, uint32_t cap,
                  uint8_t* tree_l, uint64_t len_l);
// every payload of one all-gather applied by one launch (imt_itree_slice_apply_gathered)
struct ApplyJobs {
    struct Job {
        const uint8_t* payload;          // header 128 B | values [cap][32] | node ids [cap]
        int pairs;                       // 1: (node, value) pairs for stored level tree_l; 0: the header's nodes
        uint32_t cap;
        uint8_t* tree_l;
        uint64_t len_l;
        uint8_t *node_in, *node_out, *root;     // header targets (any may be NULL)
    } j[16];
    int n_jobs;
    const uint32_t* poison;              // device-visible word (may be NULL): non-zero = a transport gave up waiting for a
                                         // payload of this gather, nothing is applied (imt_flags.hip)
};
// done (here and in pack_writeback): the event the kernel's own completion signals (hipExtLaunchKernelGGL's stop event) --
// what a hipEventRecord right behind the launch would mark, without the marker packet
void apply_gathered(hipStream_t s, const ApplyJobs& a, hipEvent_t done = nullptr);
void store_top_path(hipStream_t s, const uint8_t* top_path, uint8_t* tree_nodes, const uint64_t* tree_off, unsigned l0,
                    unsigned depth);

}  // namespace launch
}  // namespace imt
