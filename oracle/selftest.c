/*
 * selftest.c -- CPU ORACLE (test infrastructure): pins the oracle to the only
 * absolute known-answer the reference holds, the zero-leaf hash literal at
 * src/indexed_merkle_tree.rs:247-250, and to the round structure of
 * test_insert_leaf_multiple_round (:679-803).
 */
#include "imt_oracle.h"
#include <stdio.h>
#include <string.h>

/* the decimal literal of src/indexed_merkle_tree.rs:248, parsed below */
static const char KAT_ZERO_DEC[] =
    "1960587138944869480785025106734196872454309951825657414575195034687326603497";

static void dec_to_le32(uint8_t out[32], const char *s) {
    memset(out, 0, 32);
    for (; *s; s++) {
        unsigned carry = (unsigned)(*s - '0');
        for (int i = 0; i < 32; i++) {
            unsigned v = out[i] * 10u + carry;
            out[i] = (uint8_t)v;
            carry = v >> 8;
        }
    }
}

static void u64le(uint8_t o[32], uint64_t v) {
    memset(o, 0, 32);
    for (int k = 0; k < 8; k++) o[k] = (uint8_t)(v >> (8 * k));
}

int main(void) {
    uint8_t z[32], zero[32] = {0}, KAT_ZERO[32];
    int bad = 0;
    dec_to_le32(KAT_ZERO, KAT_ZERO_DEC);
    orc_hash3(z, zero, zero, zero);
    if (memcmp(z, KAT_ZERO, 32)) { printf("FAIL: H(0,0,0) != reference KAT\n"); bad = 1; }
    {   /* public circomlib known answers: lane 0 of the permutation of [0,1,2] and of [0,0,0] */
        static const char P12[] = "7853200120776062878684798364095072458815029376092732009249414926327459813530";
        static const char P00[] = "14744269619966411208579211824598458697587494354926760081771325075741142829156";
        uint8_t st[3][32], want[32];
        memset(st, 0, sizeof st);
        st[1][0] = 1; st[2][0] = 2;
        orc_permute_bytes(st);
        dec_to_le32(want, P12);
        if (memcmp(st[0], want, 32)) { printf("FAIL: permutation([0,1,2])[0] != circomlib poseidon([1,2])\n"); bad = 1; }
        memset(st, 0, sizeof st);
        orc_permute_bytes(st);
        dec_to_le32(want, P00);
        if (memcmp(st[0], want, 32)) { printf("FAIL: permutation([0,0,0])[0] != circomlib poseidon([0,0])\n"); bad = 1; }
    }
    /* test_insert_leaf_multiple_round: depth 3, values 30,10,20,5,50,35 */
    static const uint64_t vals[6] = {30, 10, 20, 5, 50, 35};
    uint8_t pre[8][3][32], leaves[8][32], old_root[32];
    memset(pre, 0, sizeof pre);
    orc_hash_preimages(&leaves[0][0], &pre[0][0][0], 8);
    orc_tree *tree;
    if (orc_tree_new(&tree, &leaves[0][0], 8)) return 2;
    orc_sparse *sp;
    if (orc_sparse_new(&sp, 3, 8)) return 2;
    for (int round = 0; round < 6; round++) {
        uint8_t v[32], oldpre[8][3][32], lp[3 * 32], lh[3 * 32], np[3 * 32], nh[3 * 32], new_root[32];
        uint64_t low;
        u64le(v, vals[round]);
        orc_tree_get_root(tree, old_root);
        memcpy(oldpre, pre, sizeof pre);
        orc_update_idx_leaf(&pre[0][0][0], 8, v, (uint64_t)round + 1, &low);
        orc_tree_get_proof(tree, low, lp, lh);
        orc_hash_preimages(&leaves[0][0], &pre[0][0][0], 8);
        orc_tree_free(tree);
        if (orc_tree_new(&tree, &leaves[0][0], 8)) return 2;
        orc_tree_get_proof(tree, (size_t)round + 1, np, nh);
        orc_tree_get_root(tree, new_root);
        int largest = 1;
        for (int i = 0; i < 32; i++) if (pre[round + 1][1][i]) largest = 0;
        orc_insert_trace tr;
        int f = orc_insert_leaf(old_root, oldpre[low], lp, lh, new_root, pre[round + 1],
                                (uint64_t)round + 1, np, nh, largest, 3, &tr);
        /* sparse builder must agree with update_idx_leaf + dense rebuild */
        uint64_t slow; int slarg; uint8_t sint[32], snew[32], slp[96], snp[96];
        int rc = orc_sparse_insert(sp, v, &slow, NULL, &slarg, sint, snew, slp, snp);
        int ok = f == 0 && rc == 0 && slow == low && slarg == largest && !memcmp(snew, new_root, 32) &&
                 !memcmp(sint, tr.interim_root, 32) && !memcmp(slp, lp, 96) && !memcmp(snp, np, 96);
        printf("round %d val %llu low %llu largest %d relations=0x%x sparse_rc=%d %s\n", round,
               (unsigned long long)vals[round], (unsigned long long)low, largest, f, rc,
               ok ? "ok" : "FAIL");
        if (!ok) bad = 1;
    }
    orc_tree_free(tree);
    orc_sparse_free(sp);
    printf(bad ? "SELFTEST FAILED\n" : "SELFTEST OK\n");
    return bad;
}
