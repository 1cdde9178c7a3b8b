/*
 * fr.c -- CPU ORACLE (test infrastructure): bn256 scalar field, 4x64 Montgomery.
 *
 * Restates halo2curves' bn256::Fr (un-vendored; reached through
 * halo2_base::halo2_proofs::halo2curves at src/indexed_merkle_tree.rs:327 under the
 * alias grumpkin::Fq).  Modulus = the decimal literal at
 * src/indexed_merkle_tree.rs:383.  R, R^2 and -p^-1 are derived from p at init.
 */
#include "imt_oracle.h"
#include <string.h>

typedef unsigned __int128 u128;

static const uint64_t P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL,
                              0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static uint64_t INV;      /* -p^-1 mod 2^64 */
static ofr_t R1, R2;      /* 2^256 mod p, 2^512 mod p (as plain integers) */
static int g_init;

static int geq_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > P[i]) return 1;
        if (a[i] < P[i]) return 0;
    }
    return 1;
}
static void sub_p(uint64_t a[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - P[i] - br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static void dbl_mod(uint64_t a[4]) {
    uint64_t top = a[3] >> 63;
    for (int i = 3; i > 0; i--) a[i] = (a[i] << 1) | (a[i - 1] >> 63);
    a[0] <<= 1;
    if (top || geq_p(a)) sub_p(a);
}

void ofr_init(void) {
    if (g_init) return;
    /* Newton iteration for p^-1 mod 2^64, then negate */
    uint64_t x = 1;
    for (int i = 0; i < 7; i++) x *= 2 - P[0] * x;
    INV = (uint64_t)0 - x;
    uint64_t t[4] = {1, 0, 0, 0};
    for (int i = 0; i < 256; i++) dbl_mod(t);
    memcpy(R1.l, t, sizeof t);
    for (int i = 0; i < 256; i++) dbl_mod(t);
    memcpy(R2.l, t, sizeof t);
    g_init = 1;
}

void ofr_raw_constants(uint64_t p[4], uint64_t r[4], uint64_t r2[4], uint64_t *inv) {
    ofr_init();
    memcpy(p, P, 32);
    memcpy(r, R1.l, 32);
    memcpy(r2, R2.l, 32);
    *inv = INV;
}

void ofr_add(ofr_t *o, const ofr_t *a, const ofr_t *b) {
    uint64_t t[4];
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a->l[i] + b->l[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || geq_p(t)) sub_p(t); /* p < 2^254 so c is never set; kept for clarity */
    memcpy(o->l, t, 32);
}

void ofr_sub(ofr_t *o, const ofr_t *a, const ofr_t *b) {
    uint64_t t[4];
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - br;
        t[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t[i] + P[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(o->l, t, 32);
}

/* Montgomery product a*b*R^-1 mod p, coarsely integrated operand scanning */
void ofr_mul(ofr_t *o, const ofr_t *a, const ofr_t *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * INV;
        c = (u128)m * P[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || geq_p(t)) sub_p(t);
    memcpy(o->l, t, 32);
}

int ofr_from_bytes(ofr_t *out, const uint8_t in[32]) {
    ofr_init();
    ofr_t t;
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int k = 7; k >= 0; k--) v = (v << 8) | in[i * 8 + k];
        t.l[i] = v;
    }
    if (geq_p(t.l)) return ORC_ERR_NONCANONICAL;
    ofr_mul(out, &t, &R2);
    return ORC_OK;
}

void ofr_to_bytes(uint8_t out[32], const ofr_t *a) {
    ofr_t one = {{1, 0, 0, 0}}, t;
    ofr_mul(&t, a, &one);
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 8; k++) out[i * 8 + k] = (uint8_t)(t.l[i] >> (8 * k));
}

void ofr_from_u64(ofr_t *out, uint64_t v) {
    ofr_init();
    ofr_t t = {{v, 0, 0, 0}};
    ofr_mul(out, &t, &R2);
}

int ofr_eq(const ofr_t *a, const ofr_t *b) { return memcmp(a->l, b->l, 32) == 0; }
int ofr_is_zero(const ofr_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }

/* a^(p-2) */
void ofr_inv(ofr_t *o, const ofr_t *a) {
    ofr_init();
    uint64_t e[4];
    memcpy(e, P, 32);
    e[0] -= 2;
    ofr_t acc = R1, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) ofr_mul(&acc, &acc, &base);
        ofr_mul(&base, &base, &base);
    }
    *o = acc;
}

/* Ord for Fr compares canonical integers (used at src/indexed_merkle_tree.rs:647) */
int ofr_cmp_canonical(const ofr_t *a, const ofr_t *b) {
    ofr_t one = {{1, 0, 0, 0}}, x, y;
    ofr_mul(&x, a, &one);
    ofr_mul(&y, b, &one);
    for (int i = 3; i >= 0; i--) {
        if (x.l[i] < y.l[i]) return -1;
        if (x.l[i] > y.l[i]) return 1;
    }
    return 0;
}
