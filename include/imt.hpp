// imt.hpp -- the host side of the hot path in a COMPILED language, above the C ABI of imt.h.
//
// The reference is Rust and this image has no Rust toolchain (bindings/rust/ carries the Rust side as source); this
// header is the same surface for C++17 callers, header-only, names and error behaviour as in the reference:
//
//   imt::Poseidon                pse_poseidon::Poseidon<Fr, 3, 2>::new(8, 57): update / squeeze_and_reset
//                                (src/utils.rs:46-47,96-100; src/indexed_merkle_tree.rs:666-667)
//                                + hash_fix_len_array_trace: the gadget-side witness trace (f1)
//   imt::IndexedMerkleTreeLeaf   src/utils.rs:12-17
//   imt::IndexedMerkleTree       src/utils.rs:5-107: create (= `new`, a keyword here), get_root, get_proof, verify_proof,
//                                the same two error strings (:25, :35)
//   imt::verify_non_inclusion    src/indexed_merkle_tree.rs:127-229   } value level: the mask of IMT_F_* constraints
//   imt::insert_leaf             src/indexed_merkle_tree.rs:231-314   } that do NOT hold (0 = the circuit is satisfied)
//   imt::IndexedTree             the depth-d tree of the reference's tests (update_idx_leaf + rehash + rebuild,
//                                :632-671, :715-735) kept on the GPU: insert_batch returns every insert_leaf input
//
// Every hash runs on the GPU (libimt_hip.so); there is no CPU path: a missing device is imt::Error(IMT_ERR_NO_DEVICE).
// tests/native/reference_tests.cpp re-enacts the reference's own tests on this surface.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "imt.h"

namespace imt {

// bn256::Fr as its canonical little-endian bytes (Fr::to_repr)
struct Fr {
    std::array<uint8_t, 32> le{};
    Fr() = default;
    static Fr from(uint64_t v) {
        Fr r;
        for (int i = 0; i < 8; i++) r.le[i] = (uint8_t)(v >> (8 * i));
        return r;
    }
    static Fr from(bool b) { return from((uint64_t)(b ? 1 : 0)); }
    static Fr zero() { return Fr(); }
    static Fr one() { return from((uint64_t)1); }
    // 64 hex digits, most significant first (how the reference's Debug prints a field element, without the 0x)
    static Fr from_hex(const std::string& hex) {
        if (hex.size() != 64) throw std::invalid_argument("Fr::from_hex: 64 hex digits expected");
        Fr r;
        for (int i = 0; i < 32; i++) r.le[31 - i] = (uint8_t)std::stoul(hex.substr(2 * i, 2), nullptr, 16);
        return r;
    }
    std::string hex() const {
        static const char* d = "0123456789abcdef";
        std::string s(64, '0');
        for (int i = 0; i < 32; i++) {
            s[2 * i] = d[le[31 - i] >> 4];
            s[2 * i + 1] = d[le[31 - i] & 15];
        }
        return s;
    }
    bool is_zero() const {
        for (uint8_t b : le)
            if (b) return false;
        return true;
    }
    uint64_t low_u64() const {
        uint64_t v = 0;
        for (int i = 7; i >= 0; i--) v = (v << 8) | le[i];
        return v;
    }
    friend bool operator==(const Fr& a, const Fr& b) { return a.le == b.le; }
    friend bool operator!=(const Fr& a, const Fr& b) { return !(a == b); }
    friend bool operator<(const Fr& a, const Fr& b) {      // as integers (Fr: Ord compares the canonical value)
        for (int i = 31; i >= 0; i--)
            if (a.le[i] != b.le[i]) return a.le[i] < b.le[i];
        return false;
    }
    friend bool operator>(const Fr& a, const Fr& b) { return b < a; }
};
static_assert(sizeof(Fr) == 32, "Fr rows are passed to the C ABI as they lie");

class Error : public std::runtime_error {
public:
    Error(int code, const std::string& what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }

private:
    int code_;
};

// One imt_ctx (one GPU, one stream).  Context::global() is the process-wide handle the reference-shaped types use,
// the same arrangement as bindings/rust/src/gpu.rs.
class Context {
public:
    explicit Context(int device = 0) {
        int rc = imt_ctx_create(device, &h_);
        if (rc) throw Error(rc, rc == IMT_ERR_NO_DEVICE ? "no HIP device: this library has no CPU path" : "imt_ctx_create failed");
    }
    ~Context() {
        if (h_) imt_ctx_destroy(h_);
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    imt_ctx* get() const { return h_; }
    void check(int rc) const {
        if (rc) throw Error(rc, imt_last_error(h_));
    }
    static Context& global() {
        static Context c(0);
        return c;
    }

private:
    imt_ctx* h_ = nullptr;
};

// f3: what the reference's is_less_than(a, b) (src/indexed_merkle_tree.rs:98-125: range.is_less_than + gate.is_equal per
// 128-bit limb, not x4, and x4, or) assigns for these two field elements: every NEW advice value in assignment order
// (4 (ceil(128 / lookup_bits) + 1) + 27 rows), the row that holds the result, and the rows whose cells a chip must also
// register with the RangeChip's lookup table (the limbs of both range checks).
struct LessThanTrace {
    std::vector<Fr> rows;
    std::vector<uint32_t> lookup_rows;
    uint32_t out_row = 0;
    bool less = false;
};
inline LessThanTrace is_less_than_trace(const Fr& a, const Fr& b, unsigned lookup_bits = 18, Context& ctx = Context::global()) {
    LessThanTrace t;
    const size_t rows = imt_less_than_trace_rows(lookup_bits);
    if (!rows) throw Error(IMT_ERR_RANGE, "is_less_than_trace: 1 <= lookup_bits <= 28");
    t.rows.resize(rows);
    uint8_t lt = 0;
    ctx.check(imt_less_than_trace_batch(ctx.get(), &a, &b, 1, lookup_bits, t.rows.data(), &lt, IMT_FMT_CANONICAL));
    t.less = lt != 0;
    size_t n_cells = 0, n_consts = 0, n_lk = 0;
    ctx.check(imt_less_than_trace_layout(ctx.get(), lookup_bits, nullptr, 0, &n_cells, nullptr, 0, &n_consts, &t.out_row,
                                         IMT_FMT_CANONICAL));
    ctx.check(imt_less_than_lookup_rows(lookup_bits, nullptr, 0, &n_lk));
    t.lookup_rows.resize(n_lk);
    ctx.check(imt_less_than_lookup_rows(lookup_bits, t.lookup_rows.data(), n_lk, &n_lk));
    return t;
}

// The native hasher of the reference's call sites.  The sponge is the reference's: T = 3, RATE = 2, two permutations
// for 2 and for 3 absorbed elements; those are the only arities the reference uses and the only ones offered.
class Poseidon {
public:
    explicit Poseidon(unsigned r_f = 8, unsigned r_p = 57, Context& ctx = Context::global()) : ctx_(&ctx) {
        if (r_f != 8 || r_p != 57) throw Error(IMT_ERR_ARG, "only Poseidon::<Fr, 3, 2>::new(8, 57) is built");
    }
    void update(const Fr* v, size_t n) { buf_.insert(buf_.end(), v, v + n); }
    void update(std::initializer_list<Fr> v) { buf_.insert(buf_.end(), v.begin(), v.end()); }
    void update(const std::vector<Fr>& v) { buf_.insert(buf_.end(), v.begin(), v.end()); }
    Fr squeeze_and_reset() {
        Fr out;
        const size_t n = buf_.size();
        std::vector<Fr> in;
        in.swap(buf_);
        if (n == 2)
            ctx_->check(imt_hash2_batch(ctx_->get(), in.data(), &out, 1, IMT_FMT_CANONICAL));
        else if (n == 3)
            ctx_->check(imt_hash3_batch(ctx_->get(), in.data(), &out, 1, IMT_FMT_CANONICAL));
        else
            throw Error(IMT_ERR_ARG, "squeeze_and_reset after " + std::to_string(n) + " elements: 2 or 3 expected");
        return out;
    }
    // n hashes of `arity` elements each in one launch: what a caller with more than one hash to do should use
    std::vector<Fr> hash_many(const std::vector<Fr>& in, int arity) {
        if ((arity != 2 && arity != 3) || in.size() % (size_t)arity) throw Error(IMT_ERR_ARG, "hash_many: arity 2 or 3");
        std::vector<Fr> out(in.size() / (size_t)arity);
        if (out.empty()) return out;
        ctx_->check(arity == 2 ? imt_hash2_batch(ctx_->get(), in.data(), out.data(), out.size(), IMT_FMT_CANONICAL)
                               : imt_hash3_batch(ctx_->get(), in.data(), out.data(), out.size(), IMT_FMT_CANONICAL));
        return out;
    }
    // f1: what halo2-base's PoseidonHasher::hash_fix_len_array(ctx, gate, inputs) assigns for these inputs
    // (src/indexed_merkle_tree.rs:92,194,271-275,299-303): every NEW advice value in assignment order (1208 rows for two
    // inputs, 1209 for three); rows[out_row] is the hash.  A chip assigns these instead of recomputing the permutations.
    struct Trace {
        std::vector<Fr> rows;
        uint32_t out_row = 0;
        const Fr& output() const { return rows[out_row]; }
    };
    Trace hash_fix_len_array_trace(const std::vector<Fr>& inputs) {
        const int arity = (int)inputs.size();
        if (arity != 2 && arity != 3) throw Error(IMT_ERR_ARG, "hash_fix_len_array_trace: 2 or 3 inputs");
        Trace t;
        t.rows.resize(imt_hash_trace_rows(arity));
        ctx_->check(imt_hash_trace_batch(ctx_->get(), inputs.data(), arity, 1, t.rows.data(), IMT_FMT_CANONICAL));
        size_t n_cells = 0, n_consts = 0;
        ctx_->check(imt_hash_trace_layout(ctx_->get(), arity, nullptr, 0, &n_cells, nullptr, 0, &n_consts, &t.out_row,
                                          IMT_FMT_CANONICAL));
        return t;
    }
    Context& context() const { return *ctx_; }

private:
    Context* ctx_;
    std::vector<Fr> buf_;
};

struct IndexedMerkleTreeLeaf {
    Fr val, next_val, next_idx;
};
static_assert(sizeof(IndexedMerkleTreeLeaf) == 96, "leaf preimages are [3][32] rows");

// src/utils.rs:5-107.  The levels live in HBM; `hash` is held as the reference holds it (and names the context).
class IndexedMerkleTree {
public:
    // IndexedMerkleTree::new (src/utils.rs:20-57), the reference's two Err strings; an even, non-power-of-two
    // length (index panic at :45 in the reference) is IMT_ERR_NOT_POW2
    static IndexedMerkleTree create(Poseidon& hash, const std::vector<Fr>& leaves) {
        imt_tree* t = nullptr;
        Context& c = hash.context();
        int rc = imt_tree_new(c.get(), leaves.data(), leaves.size(), IMT_FMT_CANONICAL, &t);
        if (rc == IMT_ERR_NO_LEAVES) throw Error(rc, "Cannot create Merkle Tree with no leaves");
        if (rc == IMT_ERR_ODD_LEAVES) throw Error(rc, "Leaves must be even");
        c.check(rc);
        return IndexedMerkleTree(hash, t);
    }
    ~IndexedMerkleTree() {
        if (t_) imt_tree_free(t_);
    }
    IndexedMerkleTree(IndexedMerkleTree&& o) noexcept : hash_(o.hash_), t_(o.t_) { o.t_ = nullptr; }
    IndexedMerkleTree& operator=(IndexedMerkleTree&& o) noexcept {
        if (this != &o) {
            if (t_) imt_tree_free(t_);
            hash_ = o.hash_;
            t_ = o.t_;
            o.t_ = nullptr;
        }
        return *this;
    }
    IndexedMerkleTree(const IndexedMerkleTree&) = delete;
    IndexedMerkleTree& operator=(const IndexedMerkleTree&) = delete;

    Fr get_root() const {                                                       // :59-61
        Fr r;
        ctx().check(imt_tree_get_root(t_, &r, IMT_FMT_CANONICAL));
        return r;
    }
    // (proof, proof_helper), helper = 1 iff the node on the path is a left child          :63-85
    std::pair<std::vector<Fr>, std::vector<Fr>> get_proof(size_t index) const {
        const size_t d = imt_tree_num_levels(t_) - 1;
        std::vector<Fr> proof(d), helper(d);
        if (d) ctx().check(imt_tree_get_proof(t_, index, proof.data(), helper.data(), IMT_FMT_CANONICAL));
        return {proof, helper};
    }
    bool verify_proof(const Fr& leaf, size_t index, const Fr& root, const std::vector<Fr>& proof) {   // :87-107
        uint64_t idx = index;
        uint8_t ok = 0;
        ctx().check(imt_verify_proof_batch(ctx().get(), &leaf, &idx, &root, proof.data(), (unsigned)proof.size(), 1, &ok,
                                           IMT_FMT_CANONICAL | IMT_SIB_ITEM_MAJOR));
        return ok != 0;
    }

private:
    IndexedMerkleTree(Poseidon& h, imt_tree* t) : hash_(&h), t_(t) {}
    Context& ctx() const { return hash_->context(); }
    Poseidon* hash_;
    imt_tree* t_;
};

namespace detail {
// helper = 1 <=> left child <=> index bit 0 (src/utils.rs:79).  A helper that is not 0 / 1 is what gate.assert_bit
// rejects (src/indexed_merkle_tree.rs:54): reported as IMT_F_BAD_BIT by the callers.
inline bool helpers_to_index(const std::vector<Fr>& helper, uint64_t& index) {
    index = 0;
    for (size_t l = 0; l < helper.size(); l++) {
        if (helper[l] == Fr::zero())
            index |= (uint64_t)1 << l;
        else if (helper[l] != Fr::one())
            return false;
    }
    return true;
}
inline bool as_bit(const Fr& f, uint8_t& bit) {
    bit = f == Fr::one();
    return bit || f.is_zero();
}
}  // namespace detail

// verify_non_inclusion (src/indexed_merkle_tree.rs:127-229) on its witness values.  Returns the IMT_F_* mask of the
// constraints that fail (0 = satisfied); the reference panics or leaves MockProver unsatisfied in those cases.
inline unsigned verify_non_inclusion(Context& c, const Fr& root, const IndexedMerkleTreeLeaf& low_leaf,
                                     const std::vector<Fr>& low_leaf_proof, const std::vector<Fr>& low_leaf_proof_helper,
                                     const Fr& new_leaf_value, const Fr& is_new_leaf_largest) {
    uint64_t idx;
    uint8_t largest, fail = 0;
    if (low_leaf_proof.size() != low_leaf_proof_helper.size()) throw Error(IMT_ERR_ARG, "proof / helper lengths differ");
    if (!detail::helpers_to_index(low_leaf_proof_helper, idx) || !detail::as_bit(is_new_leaf_largest, largest))
        return IMT_F_BAD_BIT;
    c.check(imt_non_membership_batch(c.get(), &root, &low_leaf, &idx, low_leaf_proof.data(), (unsigned)low_leaf_proof.size(),
                                     &new_leaf_value, &largest, 1, &fail, nullptr, IMT_FMT_CANONICAL | IMT_SIB_ITEM_MAJOR));
    return fail;
}

// insert_leaf (src/indexed_merkle_tree.rs:231-314) on its witness values; as in the reference, new_leaf_index (hashed
// into the rewritten low leaf, :265-269) is not tied to new_leaf_proof_helper (which positions the new slot's path).
inline unsigned insert_leaf(Context& c, const Fr& old_root, const IndexedMerkleTreeLeaf& low_leaf,
                            const std::vector<Fr>& low_leaf_proof, const std::vector<Fr>& low_leaf_proof_helper,
                            const Fr& new_root, const IndexedMerkleTreeLeaf& new_leaf, const Fr& new_leaf_index,
                            const std::vector<Fr>& new_leaf_proof, const std::vector<Fr>& new_leaf_proof_helper,
                            const Fr& is_new_leaf_largest) {
    const size_t d = low_leaf_proof.size();
    if (low_leaf_proof_helper.size() != d || new_leaf_proof.size() != d || new_leaf_proof_helper.size() != d)
        throw Error(IMT_ERR_ARG, "proof / helper lengths differ");
    uint64_t low_idx, new_path_idx;
    uint8_t largest, fail = 0;
    if (!detail::helpers_to_index(low_leaf_proof_helper, low_idx) ||
        !detail::helpers_to_index(new_leaf_proof_helper, new_path_idx) || !detail::as_bit(is_new_leaf_largest, largest))
        return IMT_F_BAD_BIT;
    for (int i = 8; i < 32; i++)
        if (new_leaf_index.le[i]) throw Error(IMT_ERR_RANGE, "new_leaf_index above 2^64");
    const uint64_t new_idx = new_leaf_index.low_u64();
    c.check(imt_insert_witness_batch(c.get(), &old_root, &low_leaf, &low_idx, low_leaf_proof.data(), &new_root, &new_leaf,
                                     &new_idx, &new_path_idx, new_leaf_proof.data(), &largest, (unsigned)d, 1, &fail, nullptr,
                                     IMT_FMT_CANONICAL | IMT_SIB_ITEM_MAJOR));
    return fail;
}

// What the reference's tests do by hand around insert_leaf (update_idx_leaf :632-660, hash_nullifier_pre_images
// :662-671, rebuild + get_proof :715-735), for a whole batch, on the GPU: a depth-d tree whose leaf i is the i-th
// inserted value's {val, next_val, next_idx}, leaf 0 the {0,0,0} sentinel.
class IndexedTree {
public:
    // every argument of insert_leaf for insertion i (proofs as proof[i][level])
    struct Insertions {
        std::vector<uint64_t> low_index;
        std::vector<IndexedMerkleTreeLeaf> low_leaf, new_leaf;
        std::vector<uint8_t> is_largest;
        std::vector<Fr> old_root, interim_root, new_root;
        std::vector<std::vector<Fr>> low_leaf_proof, new_leaf_proof;
        std::vector<uint64_t> new_index;
        static std::vector<Fr> helpers(uint64_t index, size_t depth) {          // src/utils.rs:79
            std::vector<Fr> h(depth);
            for (size_t l = 0; l < depth; l++) h[l] = Fr::from((uint64_t)(((index >> l) & 1) ^ 1));
            return h;
        }
    };
    IndexedTree(Context& c, unsigned depth, uint64_t capacity) : c_(&c), depth_(depth) {
        c.check(imt_itree_new(c.get(), depth, capacity, &t_));
    }
    ~IndexedTree() {
        if (t_) imt_itree_free(t_);
    }
    IndexedTree(const IndexedTree&) = delete;
    IndexedTree& operator=(const IndexedTree&) = delete;
    uint64_t size() const { return imt_itree_size(t_); }
    unsigned depth() const { return depth_; }
    Fr root() {
        Fr r;
        c_->check(imt_itree_root(t_, &r, IMT_FMT_CANONICAL));
        return r;
    }
    // value 0 or a value already present: Error(IMT_ERR_VALUE), the tree is unchanged (the circuit would panic at
    // src/indexed_merkle_tree.rs:190 / stay unsatisfied)
    Insertions insert_batch(const std::vector<Fr>& vals) {
        const size_t n = vals.size(), d = depth_;
        Insertions r;
        if (!n) return r;
        const uint64_t first = size();
        r.low_index.resize(n);
        r.low_leaf.resize(n);
        r.new_leaf.resize(n);
        r.is_largest.resize(n);
        r.old_root.resize(n);
        r.interim_root.resize(n);
        r.new_root.resize(n);
        std::vector<Fr> ls(n * d), ns(n * d);
        imt_insert_out out{r.low_index.data(), r.low_leaf.data(), r.is_largest.data(), r.old_root.data(),
                           r.interim_root.data(), r.new_root.data(), r.new_leaf.data(), ls.data(), ns.data()};
        c_->check(imt_itree_insert_batch(t_, vals.data(), n, &out, IMT_FMT_CANONICAL | IMT_SIB_ITEM_MAJOR));
        r.low_leaf_proof.resize(n);
        r.new_leaf_proof.resize(n);
        r.new_index.resize(n);
        for (size_t i = 0; i < n; i++) {
            r.low_leaf_proof[i].assign(ls.begin() + i * d, ls.begin() + (i + 1) * d);
            r.new_leaf_proof[i].assign(ns.begin() + i * d, ns.begin() + (i + 1) * d);
            r.new_index[i] = first + i;
        }
        return r;
    }
    std::vector<IndexedMerkleTreeLeaf> get_leaves(const std::vector<uint64_t>& index) {
        std::vector<IndexedMerkleTreeLeaf> out(index.size());
        if (!index.empty()) c_->check(imt_itree_get_leaves(t_, index.data(), index.size(), out.data(), IMT_FMT_CANONICAL));
        return out;
    }
    // checkpoint / resume (the reference's serde leaf, src/utils.rs:12-17): every leaf in index order; load() checks the
    // list on the GPU and throws, leaving the tree as it was, if it is not one sorted chain from the sentinel
    std::vector<IndexedMerkleTreeLeaf> snapshot() {
        std::vector<IndexedMerkleTreeLeaf> out(size());
        if (!out.empty()) c_->check(imt_itree_get_leaves(t_, nullptr, out.size(), out.data(), IMT_FMT_CANONICAL));
        return out;
    }
    void load(const std::vector<IndexedMerkleTreeLeaf>& leaves) {
        c_->check(imt_itree_load(t_, leaves.data(), leaves.size(), IMT_FMT_CANONICAL));
    }
    std::vector<Fr> get_proof(uint64_t index) {
        std::vector<Fr> sib(depth_);
        if (depth_) c_->check(imt_itree_get_proof_batch(t_, &index, 1, sib.data(), IMT_FMT_CANONICAL | IMT_SIB_ITEM_MAJOR));
        return sib;
    }
    // the witness of verify_non_inclusion for a value that is NOT in the tree
    struct NonMembership {
        uint64_t low_index;
        IndexedMerkleTreeLeaf low_leaf;
        uint8_t is_largest;
        std::vector<Fr> low_leaf_proof;
    };
    NonMembership non_membership_witness(const Fr& value) {
        NonMembership w;
        w.low_leaf_proof.resize(depth_);
        c_->check(imt_itree_non_membership_witness(t_, &value, 1, &w.low_index, &w.low_leaf, &w.is_largest,
                                                   w.low_leaf_proof.data(), IMT_FMT_CANONICAL | IMT_SIB_ITEM_MAJOR));
        return w;
    }
    imt_itree* get() const { return t_; }

private:
    Context* c_;
    unsigned depth_;
    imt_itree* t_ = nullptr;
};

// ---- several GPUs, the single sorted list: imt_sliced_* --------------------------------------------------------------
// The reference's ONE sorted list (update_idx_leaf, src/indexed_merkle_tree.rs:632-660) on `world` GPUs, bit-exact with
// one: the schedule, its streams and events and the all-gather live in the library; a host makes one call per step.
// This wrapper owns the imt_sliced handle and (optionally) its transport.
class Sliced {
public:
    // all `world` replicas in this process (tests, the one-GPU rehearsal): the in-process transport
    static Sliced local(const std::vector<imt_itree*>& trees, size_t max_slice, int lag = 0) {
        imt_transport* tp = nullptr;
        if (int rc = imt_transport_local_create(&tp)) throw Error(rc, "imt_transport_local_create failed");
        return Sliced(trees, (int)trees.size(), 0, tp, true, max_slice, lag);
    }
    // one rank of a distributed world over a transport the caller made (imt_transport_rccl_create, ..._ipc_create)
    Sliced(imt_itree* tree, int world, int rank, imt_transport* tp, size_t max_slice, int lag = 0)
        : Sliced(std::vector<imt_itree*>{tree}, world, rank, tp, false, max_slice, lag) {}
    Sliced(Sliced&& o) noexcept : w_(o.w_), tp_(o.tp_), own_tp_(o.own_tp_) { o.w_ = nullptr; o.tp_ = nullptr; }
    Sliced(const Sliced&) = delete;
    ~Sliced() {
        if (w_) imt_sliced_destroy(w_);
        if (tp_ && own_tp_) imt_transport_destroy(tp_);
    }
    // vals: all world x n values of the step (device memory); outs[k]: local rank k's witness buffers
    uint64_t step(const void* vals, size_t n, const imt_insert_out* outs, unsigned flags = 0) {
        uint64_t round = 0;
        check(imt_sliced_step(w_, vals, n, outs, flags, &round));
        return round;
    }
    void wait(uint64_t round, int local_rank = 0) { check(imt_sliced_wait(w_, local_rank, round)); }
    void flush() { check(imt_sliced_flush(w_)); }
    imt_sliced_info info() const {
        imt_sliced_info o{};
        imt_sliced_get_info(w_, &o);
        return o;
    }
    // PREP_STREAM, WATCHDOG_MS, TIMING of this world (the others: imt_sliced_set_option(nullptr, ...) before creation)
    void set_option(int option, long value) { check(imt_sliced_set_option(w_, option, value)); }
    // where the world stands, as text: what IMT_ERR_TIMEOUT writes to stderr
    std::string dump() const {
        std::string out((size_t)imt_sliced_dump(w_, nullptr, 0) + 1, '\0');
        imt_sliced_dump(w_, &out[0], out.size());
        out.resize(std::strlen(out.c_str()));
        return out;
    }

private:
    Sliced(const std::vector<imt_itree*>& trees, int world, int first_rank, imt_transport* tp, bool own, size_t max_slice, int lag)
        : tp_(tp), own_tp_(own) {
        const int rc = imt_sliced_create(trees.data(), (int)trees.size(), world, first_rank, tp, max_slice, lag, &w_);
        if (rc) {
            if (own) imt_transport_destroy(tp);
            throw Error(rc, "imt_sliced_create failed");
        }
    }
    void check(int rc) const {
        if (rc) throw Error(rc, imt_sliced_last_error(w_));
    }
    imt_sliced* w_ = nullptr;
    imt_transport* tp_ = nullptr;
    bool own_tp_ = false;
};

}  // namespace imt
