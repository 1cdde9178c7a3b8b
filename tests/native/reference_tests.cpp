// reference_tests.cpp -- the reference's own tests, re-enacted in C++ on include/imt.hpp (the compiled-language host side
// above the C ABI).  Each function follows the test of the same name in /root/reference/src/indexed_merkle_tree.rs
// (lines cited) with the MockProver run replaced by the value-level constraint check of the same function
// (imt::insert_leaf returns the mask of constraints that do not hold).  Every hash runs on the GPU.
// Built and run by tests/test_gpu_parity.py::test_reference_tests_in_cpp, which also compares the printed roots with
// tests/golden/vectors.json (KAT-anchored) -- this program only checks what the reference's tests check.
//
//   g++ -std=c++17 -I include tests/native/reference_tests.cpp -L indexed-merkle-tree-halo2_amd/csrc -limt_hip -o reference_tests
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <tuple>

#include "imt.hpp"

using imt::Fr;
using IMTLeaf = imt::IndexedMerkleTreeLeaf;

#define CHECK(cond)                                                               \
    do {                                                                          \
        if (!(cond)) {                                                            \
            std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

static const char* P_HEX = "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001";

// rng.gen_biguint(254) reduced mod r (:381-386): a 254-bit draw, minus p once if needed
static Fr random_fr(std::mt19937_64& rng) {
    const Fr p = Fr::from_hex(P_HEX);
    Fr a;
    for (int w = 0; w < 4; w++) {
        uint64_t x = rng();
        for (int i = 0; i < 8; i++) a.le[8 * w + i] = (uint8_t)(x >> (8 * i));
    }
    a.le[31] &= 0x3f;
    if (!(a < p)) {
        int borrow = 0;
        for (int i = 0; i < 32; i++) {
            int v = (int)a.le[i] - (int)p.le[i] - borrow;
            borrow = v < 0;
            a.le[i] = (uint8_t)(v + (borrow ? 256 : 0));
        }
    }
    return a;
}

// :805-810
static void test_hash_zero() {
    imt::Poseidon native_hasher(8, 57);
    native_hasher.update({Fr::zero(), Fr::zero(), Fr::zero()});
    const Fr h = native_hasher.squeeze_and_reset();
    std::printf("hash_zero=%s\n", h.hex().c_str());
    // the gadget's side of the same hash (hasher.hash_fix_len_array, :194): the trace a chip would assign ends in it
    const auto t3 = native_hasher.hash_fix_len_array_trace({Fr::zero(), Fr::zero(), Fr::zero()});
    CHECK(t3.rows.size() == 1209 && t3.output() == h);
    native_hasher.update({Fr::from((uint64_t)7), Fr::from((uint64_t)11)});
    const Fr h2 = native_hasher.squeeze_and_reset();
    const auto t2 = native_hasher.hash_fix_len_array_trace({Fr::from((uint64_t)7), Fr::from((uint64_t)11)});
    CHECK(t2.rows.size() == 1208 && t2.output() == h2 && t2.out_row == 1204);
    // f3: the comparison the gadget makes at :180 / :226, as the rows a chip assigns (63 at lookup_bits 18, the reference's
    // tests' :436); the 18 lookup rows are 18-bit limbs
    const auto lt = imt::is_less_than_trace(Fr::from((uint64_t)7), h2);
    CHECK(lt.rows.size() == 63 && lt.less == (Fr::from((uint64_t)7) < h2) && lt.rows[lt.out_row] == Fr::from(lt.less));
    CHECK(lt.lookup_rows.size() == 18);
    for (uint32_t r : lt.lookup_rows) CHECK(lt.rows[r] < Fr::from((uint64_t)1 << 18));
    const auto ge = imt::is_less_than_trace(h2, h2);
    CHECK(!ge.less && ge.rows[ge.out_row] == Fr::zero());
}

// :361-478
static void test_insert_leaf() {
    const size_t tree_size = 8;
    std::vector<Fr> leaves;
    imt::Poseidon native_hasher(8, 57);
    for (size_t i = 0; i < tree_size; i++) {
        native_hasher.update({Fr::from((uint64_t)0), Fr::from((uint64_t)0), Fr::from((uint64_t)0)});
        leaves.push_back(native_hasher.squeeze_and_reset());
    }
    auto tree = imt::IndexedMerkleTree::create(native_hasher, leaves);

    std::mt19937_64 rng(0x494D5401);
    const Fr new_val = random_fr(rng);

    const Fr old_root = tree.get_root();
    const IMTLeaf low_leaf{Fr::from((uint64_t)0), Fr::from((uint64_t)0), Fr::from((uint64_t)0)};
    auto [low_leaf_proof, low_leaf_proof_helper] = tree.get_proof(0);
    CHECK(tree.verify_proof(leaves[0], 0, tree.get_root(), low_leaf_proof) == true);

    const IMTLeaf new_low_leaf{low_leaf.val, new_val, Fr::from((uint64_t)1)};
    native_hasher.update({new_low_leaf.val, new_low_leaf.next_val, new_low_leaf.next_idx});
    leaves[0] = native_hasher.squeeze_and_reset();
    native_hasher.update({new_val, Fr::from((uint64_t)0), Fr::from((uint64_t)0)});
    leaves[1] = native_hasher.squeeze_and_reset();

    tree = imt::IndexedMerkleTree::create(native_hasher, leaves);
    auto [new_leaf_proof, new_leaf_proof_helper] = tree.get_proof(1);
    CHECK(tree.verify_proof(leaves[1], 1, tree.get_root(), new_leaf_proof) == true);

    const Fr new_root = tree.get_root();
    const IMTLeaf new_leaf{new_val, Fr::from((uint64_t)0), Fr::from((uint64_t)0)};
    const Fr new_leaf_index = Fr::from((uint64_t)1);
    const Fr is_new_leaf_largest = Fr::from(true);

    imt::Context& ctx = imt::Context::global();
    CHECK(imt::insert_leaf(ctx, old_root, low_leaf, low_leaf_proof, low_leaf_proof_helper, new_root, new_leaf, new_leaf_index,
                           new_leaf_proof, new_leaf_proof_helper, is_new_leaf_largest) == 0);          // expect_satisfied(true)
    // what MockProver would reject, constraint by constraint
    CHECK(imt::insert_leaf(ctx, old_root, low_leaf, low_leaf_proof, low_leaf_proof_helper, old_root, new_leaf, new_leaf_index,
                           new_leaf_proof, new_leaf_proof_helper, is_new_leaf_largest) == IMT_F_NEW_ROOT);
    CHECK(imt::insert_leaf(ctx, new_root, low_leaf, low_leaf_proof, low_leaf_proof_helper, new_root, new_leaf, new_leaf_index,
                           new_leaf_proof, new_leaf_proof_helper, is_new_leaf_largest) & IMT_F_LOW_IN_ROOT);
    CHECK(imt::insert_leaf(ctx, old_root, low_leaf, low_leaf_proof, low_leaf_proof_helper, new_root, new_leaf, new_leaf_index,
                           new_leaf_proof, new_leaf_proof_helper, Fr::from(false)) & IMT_F_RANGE_PRED);
    CHECK(imt::insert_leaf(ctx, old_root, low_leaf, low_leaf_proof, low_leaf_proof_helper, new_root, new_leaf, new_leaf_index,
                           new_leaf_proof, new_leaf_proof_helper, Fr::from((uint64_t)2)) == IMT_F_BAD_BIT);
    // the verification-only half on the same witness (:127-229)
    CHECK(imt::verify_non_inclusion(ctx, old_root, low_leaf, low_leaf_proof, low_leaf_proof_helper, new_val,
                                    is_new_leaf_largest) == 0);
    std::printf("insert_leaf new_val=%s new_root=%s\n", new_val.hex().c_str(), new_root.hex().c_str());
}

// update_idx_leaf :632-660 -- what it does, on this file's types: the first leaf whose value lies below the new one
// and whose successor lies above it (or is absent) becomes the low leaf
static std::pair<std::vector<IMTLeaf>, size_t> update_idx_leaf(const std::vector<IMTLeaf>& leaves, const Fr& new_val,
                                                               uint64_t new_val_idx) {
    std::vector<IMTLeaf> out = leaves;
    for (size_t i = 0; i < leaves.size(); i++) {
        const IMTLeaf& node = leaves[i];
        const bool first_ever = i == 0 && node.next_val.is_zero();
        const bool between = node.val < new_val && (node.next_val > new_val || node.next_val.is_zero());
        if (first_ever || between) {
            const size_t slot = first_ever ? 1 : (size_t)new_val_idx;
            out[slot].val = new_val;
            if (!first_ever) {
                out[slot].next_val = node.next_val;
                out[slot].next_idx = node.next_idx;
            }
            out[i].next_val = new_val;
            out[i].next_idx = Fr::from((uint64_t)slot);
            return {out, i};
        }
    }
    return {out, 0};
}

// hash_nullifier_pre_images :662-671 (one launch instead of a loop)
static std::vector<Fr> hash_nullifier_pre_images(imt::Poseidon& h, const std::vector<IMTLeaf>& pre) {
    std::vector<Fr> flat;
    for (const IMTLeaf& l : pre) {
        flat.push_back(l.val);
        flat.push_back(l.next_val);
        flat.push_back(l.next_idx);
    }
    return h.hash_many(flat, 3);
}

// :679-803
static void test_insert_leaf_multiple_round() {
    imt::Poseidon native_hasher(8, 57);
    imt::Context& ctx = imt::Context::global();
    const std::vector<Fr> new_vals = {Fr::from((uint64_t)30), Fr::from((uint64_t)10), Fr::from((uint64_t)20),
                                      Fr::from((uint64_t)5),  Fr::from((uint64_t)50), Fr::from((uint64_t)35)};
    std::vector<IMTLeaf> nullifier_tree_preimages(8);
    std::vector<IMTLeaf> old_nullifier_tree_preimages = nullifier_tree_preimages;
    std::vector<Fr> nullifier_tree_leaves = hash_nullifier_pre_images(native_hasher, nullifier_tree_preimages);
    auto tree = imt::IndexedMerkleTree::create(native_hasher, nullifier_tree_leaves);

    // the same six insertions as ONE batch on the GPU tree: must give the same witnesses as the by-hand rounds
    imt::IndexedTree gpu_tree(ctx, 3, 8);
    const auto batch = gpu_tree.insert_batch(new_vals);

    for (size_t round = 0; round < new_vals.size(); round++) {
        const Fr new_val = new_vals[round];
        const Fr old_root = tree.get_root();
        size_t low_leaf_idx;
        std::tie(nullifier_tree_preimages, low_leaf_idx) = update_idx_leaf(nullifier_tree_preimages, new_val, round + 1);
        const IMTLeaf low_leaf = old_nullifier_tree_preimages[low_leaf_idx];
        auto [low_leaf_proof, low_leaf_proof_helper] = tree.get_proof(low_leaf_idx);

        nullifier_tree_leaves = hash_nullifier_pre_images(native_hasher, nullifier_tree_preimages);
        tree = imt::IndexedMerkleTree::create(native_hasher, nullifier_tree_leaves);

        const IMTLeaf new_leaf = nullifier_tree_preimages[round + 1];
        const Fr new_leaf_index = Fr::from((uint64_t)(round + 1));
        auto [new_leaf_proof, new_leaf_proof_helper] = tree.get_proof(round + 1);
        const Fr new_root = tree.get_root();
        const Fr is_new_leaf_largest = Fr::from(nullifier_tree_preimages[round + 1].next_val.is_zero());

        CHECK(imt::insert_leaf(ctx, old_root, low_leaf, low_leaf_proof, low_leaf_proof_helper, new_root, new_leaf,
                               new_leaf_index, new_leaf_proof, new_leaf_proof_helper, is_new_leaf_largest) == 0);
        // the batch path produced exactly these values
        CHECK(batch.low_index[round] == low_leaf_idx && batch.new_index[round] == round + 1);
        CHECK(batch.old_root[round] == old_root && batch.new_root[round] == new_root);
        CHECK(batch.low_leaf[round].val == low_leaf.val && batch.low_leaf[round].next_val == low_leaf.next_val &&
              batch.low_leaf[round].next_idx == low_leaf.next_idx);
        CHECK(batch.new_leaf[round].val == new_leaf.val && batch.new_leaf[round].next_val == new_leaf.next_val &&
              batch.new_leaf[round].next_idx == new_leaf.next_idx);
        CHECK(batch.low_leaf_proof[round] == low_leaf_proof && batch.new_leaf_proof[round] == new_leaf_proof);
        CHECK(imt::IndexedTree::Insertions::helpers(batch.low_index[round], 3) == low_leaf_proof_helper);
        CHECK(Fr::from(batch.is_largest[round] != 0) == is_new_leaf_largest);
        std::printf("round %zu low_leaf_idx=%zu new_root=%s\n", round, low_leaf_idx, new_root.hex().c_str());
        old_nullifier_tree_preimages = nullifier_tree_preimages;
    }
    CHECK(gpu_tree.root() == tree.get_root() && gpu_tree.size() == 7);
    // a value that is not in the tree: its non-membership witness satisfies verify_non_inclusion; one that is does not exist
    const auto w = gpu_tree.non_membership_witness(Fr::from((uint64_t)25));
    CHECK(w.low_leaf.val == Fr::from((uint64_t)20) && w.low_leaf.next_val == Fr::from((uint64_t)30) && !w.is_largest);
    CHECK(imt::verify_non_inclusion(ctx, gpu_tree.root(), w.low_leaf, w.low_leaf_proof,
                                    imt::IndexedTree::Insertions::helpers(w.low_index, 3), Fr::from((uint64_t)25),
                                    Fr::from(false)) == 0);
    CHECK(imt::verify_non_inclusion(ctx, gpu_tree.root(), w.low_leaf, w.low_leaf_proof,
                                    imt::IndexedTree::Insertions::helpers(w.low_index, 3), Fr::from((uint64_t)31),
                                    Fr::from(false)) & IMT_F_RANGE_PRED);
    bool threw = false;
    try {
        gpu_tree.insert_batch({Fr::from((uint64_t)20)});        // already present
    } catch (const imt::Error& e) {
        threw = e.code() == IMT_ERR_VALUE;
    }
    CHECK(threw && gpu_tree.size() == 7);
}

// :597-630: a < b from the 2^128 limbs, the formula of is_less_than (:98-125) -- limbs from imt_split128_batch
static void test_limbs_logic() {
    imt::Context& ctx = imt::Context::global();
    std::mt19937_64 rng(0x494D5403);
    const size_t n = 4096;
    std::vector<Fr> v(2 * n), q(2 * n), r(2 * n);
    for (size_t i = 0; i < 2 * n; i++) v[i] = random_fr(rng);
    for (size_t i = 0; i < 64; i++) {                             // equal high limbs / equal low limbs / equal values
        v[2 * i + 1] = v[2 * i];
        if (i % 3 == 0) v[2 * i + 1].le[0] ^= 1;
        if (i % 3 == 1) v[2 * i + 1].le[20] ^= 1;
    }
    ctx.check(imt_split128_batch(ctx.get(), v.data(), q.data(), r.data(), 2 * n, IMT_FMT_CANONICAL));
    for (size_t i = 0; i < n; i++) {
        const Fr &a_q = q[2 * i], &a_r = r[2 * i], &b_q = q[2 * i + 1], &b_r = r[2 * i + 1];
        for (int k = 16; k < 32; k++) CHECK(a_q.le[k] == 0 && a_r.le[k] == 0);
        CHECK(std::memcmp(a_r.le.data(), v[2 * i].le.data(), 16) == 0 && std::memcmp(a_q.le.data(), v[2 * i].le.data() + 16, 16) == 0);
        const bool lhs = a_q < b_q;
        const bool rhs = (a_q == b_q) && (a_r < b_r);
        CHECK((v[2 * i] < v[2 * i + 1]) == (lhs || rhs));
    }
}

// src/utils.rs:24-36
static void test_tree_errors() {
    imt::Poseidon h;
    const char* msg = "";
    try {
        imt::IndexedMerkleTree::create(h, {});
    } catch (const imt::Error& e) {
        msg = e.code() == IMT_ERR_NO_LEAVES ? "no leaves" : "";
        CHECK(std::string(e.what()) == "Cannot create Merkle Tree with no leaves");
    }
    CHECK(std::string(msg) == "no leaves");
    msg = "";
    try {
        imt::IndexedMerkleTree::create(h, {Fr::from((uint64_t)1), Fr::from((uint64_t)2), Fr::from((uint64_t)3)});
    } catch (const imt::Error& e) {
        msg = "odd";
        CHECK(std::string(e.what()) == "Leaves must be even");
    }
    CHECK(std::string(msg) == "odd");
    auto one = imt::IndexedMerkleTree::create(h, {Fr::from((uint64_t)7)});       // :27-33: a single leaf is its own root
    CHECK(one.get_root() == Fr::from((uint64_t)7) && one.get_proof(0).first.empty());
    bool threw = false;
    try {
        imt::Poseidon bad(8, 56);
    } catch (const imt::Error&) {
        threw = true;
    }
    CHECK(threw);
}

int main() {
    try {
        test_hash_zero();
        test_tree_errors();
        test_insert_leaf();
        test_insert_leaf_multiple_round();
        test_limbs_logic();
    } catch (const imt::Error& e) {
        std::fprintf(stderr, "imt::Error %d: %s\n", e.code(), e.what());
        return 2;
    }
    std::printf("reference tests: ok\n");
    return 0;
}
