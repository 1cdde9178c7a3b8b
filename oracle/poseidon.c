/*
 * poseidon.c -- CPU ORACLE (test infrastructure): Poseidon T=3, RATE=2, R_F=8, R_P=57
 * over bn256::Fr, in the PLAIN 65-round form, plus the rate-2 sponge.
 *
 * Restates the un-vendored crate pse-poseidon (aerius-labs fork, branch
 * feat/stateless-hash, Cargo.toml:16; no pinned commit): Spec::new(8,57) =
 * Grain-LFSR round constants + Cauchy MDS, Poseidon::{new,update,squeeze_and_reset}.
 * Reference call sites: src/utils.rs:46-47,96-100; src/indexed_merkle_tree.rs:370,
 * 374-375,407-415,510-518,663-668,807-809.  halo2-base's PoseidonHasher::
 * hash_fix_len_array (call sites :92,194,271-275,299-303) yields the same values
 * (the reference's MockProver tests enforce native == circuit).
 * Pinned by the KAT at src/indexed_merkle_tree.rs:247-250 (see selftest.c).
 */
#include "imt_oracle.h"
#include <string.h>

static ofr_t RC[ORC_ROUNDS][3];
static ofr_t MDS[3][3];
static ofr_t CAP0;   /* initial capacity lane: 2^64 */
static ofr_t ONE;
static int g_pinit;

/* ---- Grain LFSR (Poseidon paper, appendix F) ---- */
typedef struct { uint8_t b[80]; int pos; } grain_t;

static int grain_new_bit(grain_t *g) {
    /* b(i+80) = b(i+62)^b(i+51)^b(i+38)^b(i+23)^b(i+13)^b(i) */
    int p = g->pos;
#define GB(k) g->b[(p + (k)) % 80]
    int nb = GB(62) ^ GB(51) ^ GB(38) ^ GB(23) ^ GB(13) ^ GB(0);
#undef GB
    g->b[p] = (uint8_t)nb;
    g->pos = (p + 1) % 80;
    return nb;
}
static void grain_append(grain_t *g, int *n, int width, unsigned v) {
    for (int i = width - 1; i >= 0; i--) g->b[(*n)++] = (v >> i) & 1; /* MSB first */
}
/* pairs: first bit 1 -> output second; first bit 0 -> discard second */
static int grain_next(grain_t *g) {
    while (!grain_new_bit(g)) grain_new_bit(g);
    return grain_new_bit(g);
}
/* 254 filtered bits, most significant first, as a 4x64 little-endian integer */
static void grain_take254(grain_t *g, uint64_t v[4]) {
    memset(v, 0, 32);
    for (int i = 253; i >= 0; i--)
        if (grain_next(g)) v[i / 64] |= (uint64_t)1 << (i % 64);
}
static void limbs_to_bytes(uint8_t out[32], const uint64_t v[4]) {
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 8; k++) out[i * 8 + k] = (uint8_t)(v[i] >> (8 * k));
}
static void grain_field(grain_t *g, ofr_t *out) { /* rejection sampling */
    uint64_t v[4];
    uint8_t by[32];
    for (;;) {
        grain_take254(g, v);
        limbs_to_bytes(by, v);
        if (ofr_from_bytes(out, by) == ORC_OK) return;
    }
}
static void grain_field_norej(grain_t *g, ofr_t *out) { /* from_bytes_wide: value mod p */
    uint64_t v[4], p[4], r[4], r2[4], inv;
    uint8_t by[32];
    ofr_raw_constants(p, r, r2, &inv);
    grain_take254(g, v);
    for (;;) { /* v < 2^254 < 2p: at most one subtraction */
        int ge = 1;
        for (int i = 3; i >= 0; i--) {
            if (v[i] > p[i]) break;
            if (v[i] < p[i]) { ge = 0; break; }
        }
        if (!ge) break;
        unsigned __int128 br = 0;
        for (int i = 0; i < 4; i++) {
            unsigned __int128 d = (unsigned __int128)v[i] - p[i] - br;
            v[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    }
    limbs_to_bytes(by, v);
    ofr_from_bytes(out, by);
}

void orc_poseidon_init(void) {
    if (g_pinit) return;
    ofr_init();
    grain_t g;
    memset(&g, 0, sizeof g);
    int n = 0;
    grain_append(&g, &n, 2, 1);      /* field type: prime */
    grain_append(&g, &n, 4, 0);      /* s-box: x^alpha */
    grain_append(&g, &n, 12, 254);   /* field size in bits */
    grain_append(&g, &n, 12, ORC_T);
    grain_append(&g, &n, 10, ORC_RF);
    grain_append(&g, &n, 10, ORC_RP);
    grain_append(&g, &n, 30, 0x3fffffffu);
    g.pos = 0;
    for (int i = 0; i < 160; i++) grain_new_bit(&g);
    for (int r = 0; r < ORC_ROUNDS; r++)
        for (int i = 0; i < 3; i++) grain_field(&g, &RC[r][i]);
    ofr_t xs[3], ys[3];
    for (int i = 0; i < 3; i++) grain_field_norej(&g, &xs[i]);
    for (int i = 0; i < 3; i++) grain_field_norej(&g, &ys[i]);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { /* Cauchy: 1/(x_i + y_j) */
            ofr_t s;
            ofr_add(&s, &xs[i], &ys[j]);
            ofr_inv(&MDS[i][j], &s);
        }
    /* State::default(): capacity lane = 2^64 */
    ofr_t two32;
    ofr_from_u64(&two32, (uint64_t)1 << 32);
    ofr_mul(&CAP0, &two32, &two32);
    ofr_from_u64(&ONE, 1);
    g_pinit = 1;
}

void orc_poseidon_params(uint8_t rc[ORC_ROUNDS * 3][32], uint8_t mds[9][32]) {
    orc_poseidon_init();
    for (int r = 0; r < ORC_ROUNDS; r++)
        for (int i = 0; i < 3; i++) ofr_to_bytes(rc[r * 3 + i], &RC[r][i]);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) ofr_to_bytes(mds[i * 3 + j], &MDS[i][j]);
}

static void sbox(ofr_t *x) {
    ofr_t x2, x4;
    ofr_mul(&x2, x, x);
    ofr_mul(&x4, &x2, &x2);
    ofr_mul(x, &x4, x);
}

void orc_permute(ofr_t s[3]) {
    orc_poseidon_init();
    for (int r = 0; r < ORC_ROUNDS; r++) {
        for (int i = 0; i < 3; i++) ofr_add(&s[i], &s[i], &RC[r][i]);
        if (r < ORC_RF / 2 || r >= ORC_RF / 2 + ORC_RP) {
            for (int i = 0; i < 3; i++) sbox(&s[i]);
        } else {
            sbox(&s[0]);
        }
        ofr_t n[3];
        for (int i = 0; i < 3; i++) {
            ofr_t acc, t;
            ofr_mul(&acc, &MDS[i][0], &s[0]);
            ofr_mul(&t, &MDS[i][1], &s[1]);
            ofr_add(&acc, &acc, &t);
            ofr_mul(&t, &MDS[i][2], &s[2]);
            ofr_add(&n[i], &acc, &t);
        }
        s[0] = n[0]; s[1] = n[1]; s[2] = n[2];
    }
}

void orc_permute_bytes(uint8_t sb[3][32]) {
    ofr_t s[3];
    for (int i = 0; i < 3; i++) ofr_from_bytes(&s[i], sb[i]);
    orc_permute(s);
    for (int i = 0; i < 3; i++) ofr_to_bytes(sb[i], &s[i]);
}

/* Sponge: state [2^64,0,0]; every full RATE chunk is added to lanes 1..2 and permuted;
 * the leftover (possibly empty) chunk followed by a 1 is added to lanes 1.. and permuted;
 * result = lane 1; state reset afterwards (squeeze_and_reset). */
static void sponge(ofr_t *out, const ofr_t *in, size_t n) {
    orc_poseidon_init();
    ofr_t s[3];
    s[0] = CAP0;
    memset(&s[1], 0, 2 * sizeof(ofr_t));
    size_t i = 0;
    for (; i + 2 <= n; i += 2) {
        ofr_add(&s[1], &s[1], &in[i]);
        ofr_add(&s[2], &s[2], &in[i + 1]);
        orc_permute(s);
    }
    if (i < n) {
        ofr_add(&s[1], &s[1], &in[i]);
        ofr_add(&s[2], &s[2], &ONE);
    } else {
        ofr_add(&s[1], &s[1], &ONE);
    }
    orc_permute(s);
    *out = s[1];
}

void orc_hash2_fr(ofr_t *out, const ofr_t *a, const ofr_t *b) {
    ofr_t in[2] = {*a, *b};
    sponge(out, in, 2);
}
void orc_hash3_fr(ofr_t *out, const ofr_t *a, const ofr_t *b, const ofr_t *c) {
    ofr_t in[3] = {*a, *b, *c};
    sponge(out, in, 3);
}

int orc_hash_var(uint8_t out[32], const uint8_t *in, size_t n) {
    ofr_t buf[16], o;
    if (n > 16) return ORC_ERR_RANGE;
    for (size_t i = 0; i < n; i++)
        if (ofr_from_bytes(&buf[i], in + 32 * i)) return ORC_ERR_NONCANONICAL;
    sponge(&o, buf, n);
    ofr_to_bytes(out, &o);
    return ORC_OK;
}
int orc_hash2(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]) {
    uint8_t in[64];
    memcpy(in, a, 32);
    memcpy(in + 32, b, 32);
    return orc_hash_var(out, in, 2);
}
int orc_hash3(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], const uint8_t c[32]) {
    uint8_t in[96];
    memcpy(in, a, 32);
    memcpy(in + 32, b, 32);
    memcpy(in + 64, c, 32);
    return orc_hash_var(out, in, 3);
}
int orc_hash2_batch(uint8_t *out, const uint8_t *in, size_t n) {
    for (size_t i = 0; i < n; i++) {
        int rc = orc_hash_var(out + 32 * i, in + 64 * i, 2);
        if (rc) return rc;
    }
    return ORC_OK;
}
int orc_hash3_batch(uint8_t *out, const uint8_t *in, size_t n) {
    for (size_t i = 0; i < n; i++) {
        int rc = orc_hash_var(out + 32 * i, in + 96 * i, 3);
        if (rc) return rc;
    }
    return ORC_OK;
}
