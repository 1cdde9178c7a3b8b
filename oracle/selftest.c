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
        {   /* f1: every gate of the emitted advice column holds and the output cell is the sponge's hash */
        static uint8_t cells[6000][32], wit[2000][32];
        static orc_trace_cell desc[6000];
        for (int arity = 2; arity <= 3; arity++) {
            uint8_t in[3][32], want[32];
            for (int j = 0; j < 3; j++) u64le(in[j], 0x1234567ULL * (uint64_t)(j + 1) + (uint64_t)arity);
            size_t nc = 0, nw = 0;
            uint32_t out_row = 0;
            if (orc_hash_trace(&in[0][0], arity, &cells[0][0], desc, 6000, &nc, &wit[0][0], 2000, &nw, &out_row)) {
                printf("FAIL: orc_hash_trace\n"); bad = 1; continue;
            }
            orc_hash_var(want, &in[0][0], (size_t)arity);
            if (memcmp(wit[out_row], want, 32)) { printf("FAIL: trace output != sponge (arity %d)\n", arity); bad = 1; }
            size_t gates = 0;
            for (size_t i = 0; i + 3 < nc; i++) {
                if (!desc[i].gate) continue;
                ofr_t a, b, c, d, t;
                ofr_from_bytes(&a, cells[i]); ofr_from_bytes(&b, cells[i + 1]);
                ofr_from_bytes(&c, cells[i + 2]); ofr_from_bytes(&d, cells[i + 3]);
                ofr_mul(&t, &b, &c);
                ofr_add(&t, &t, &a);
                if (!ofr_eq(&t, &d)) { printf("FAIL: gate at cell %zu (arity %d)\n", i, arity); bad = 1; break; }
                gates++;
            }
            printf("trace arity %d: %zu cells, %zu witnesses, %zu gates, out row %u\n", arity, nc, nw, gates, out_row);
            if (nw != (arity == 2 ? 1208u : 1209u) || nc != (arity == 2 ? 4506u : 4509u)) { printf("FAIL: trace size\n"); bad = 1; }
        }
    }
    return bad;
}
