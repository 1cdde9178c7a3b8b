// imt_fr_host.hpp -- host-side bn256::Fr arithmetic for the PRODUCT library
// (constant-table generation, boundary conversions, host bookkeeping).
// 4x64-bit Montgomery, R = 2^256.  Independent of oracle/ (which is test-only).
// Field: halo2curves bn256::Fr, the modulus literal at
// /root/reference/src/indexed_merkle_tree.rs:383.
#pragma once
#include <cstdint>
#include <cstring>

namespace imt {

typedef unsigned __int128 u128;

struct HFr {
    uint64_t l[4];
    bool operator==(const HFr& o) const { return std::memcmp(l, o.l, 32) == 0; }
};

struct HField {
    static constexpr uint64_t P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL,
                                      0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    uint64_t inv;   // -p^-1 mod 2^64
    HFr r1, r2;     // R mod p, R^2 mod p
    HFr r3;         // R^3 mod p (Montgomery form of R: one mul converts canonical->Montgomery^2)

    HField() {
        uint64_t x = 1;
        for (int i = 0; i < 7; i++) x *= 2 - P[0] * x;
        inv = 0 - x;
        uint64_t t[4] = {1, 0, 0, 0};
        for (int i = 0; i < 256; i++) dbl(t);
        std::memcpy(r1.l, t, 32);
        for (int i = 0; i < 256; i++) dbl(t);
        std::memcpy(r2.l, t, 32);
        r3 = mul(r2, r2);
    }
    static bool geq_p(const uint64_t a[4]) {
        for (int i = 3; i >= 0; i--) {
            if (a[i] != P[i]) return a[i] > P[i];
        }
        return true;
    }
    static void sub_p(uint64_t a[4]) {
        u128 br = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)a[i] - P[i] - br;
            a[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    }
    static void dbl(uint64_t a[4]) {
        uint64_t top = a[3] >> 63;
        for (int i = 3; i > 0; i--) a[i] = (a[i] << 1) | (a[i - 1] >> 63);
        a[0] <<= 1;
        if (top || geq_p(a)) sub_p(a);
    }
    HFr add(const HFr& a, const HFr& b) const {
        HFr o;
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)a.l[i] + b.l[i];
            o.l[i] = (uint64_t)c;
            c >>= 64;
        }
        if (geq_p(o.l)) sub_p(o.l);
        return o;
    }
    HFr sub(const HFr& a, const HFr& b) const {
        HFr o;
        u128 br = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)a.l[i] - b.l[i] - br;
            o.l[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
        if (br) {
            u128 c = 0;
            for (int i = 0; i < 4; i++) {
                c += (u128)o.l[i] + P[i];
                o.l[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        return o;
    }
    // separated operand scanning: full 512-bit product, then word-by-word reduction
    HFr mul(const HFr& a, const HFr& b) const {
        uint64_t t[9] = {0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)a.l[i] * b.l[j] + t[i + j];
                t[i + j] = (uint64_t)c;
                c >>= 64;
            }
            t[i + 4] = (uint64_t)c;
        }
        for (int i = 0; i < 4; i++) {
            uint64_t m = t[i] * inv;
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)m * P[j] + t[i + j];
                t[i + j] = (uint64_t)c;
                c >>= 64;
            }
            for (int k = i + 4; k < 9 && c; k++) {
                c += t[k];
                t[k] = (uint64_t)c;
                c >>= 64;
            }
        }
        HFr o;
        std::memcpy(o.l, t + 4, 32);
        if (t[8] || geq_p(o.l)) sub_p(o.l);
        return o;
    }
    HFr from_u64(uint64_t v) const { return mul(HFr{{v, 0, 0, 0}}, r2); }
    HFr zero() const { return HFr{{0, 0, 0, 0}}; }
    HFr one() const { return r1; }
    // canonical little-endian bytes -> Montgomery; false if >= p
    bool from_bytes(HFr& out, const uint8_t in[32]) const {
        HFr t;
        std::memcpy(t.l, in, 32);   // little-endian host
        if (geq_p(t.l)) return false;
        out = mul(t, r2);
        return true;
    }
    void to_bytes(uint8_t out[32], const HFr& a) const {
        HFr t = mul(a, HFr{{1, 0, 0, 0}});
        std::memcpy(out, t.l, 32);
    }
    HFr inverse(const HFr& a) const {   // a^(p-2)
        uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};
        HFr acc = r1, base = a;
        for (int i = 0; i < 256; i++) {
            if ((e[i / 64] >> (i % 64)) & 1) acc = mul(acc, base);
            base = mul(base, base);
        }
        return acc;
    }
    bool is_zero(const HFr& a) const { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
};

// canonical 32-byte little-endian integers compared as integers
inline int cmp_le32(const uint8_t* a, const uint8_t* b) {
    const uint64_t* x = reinterpret_cast<const uint64_t*>(a);
    const uint64_t* y = reinterpret_cast<const uint64_t*>(b);
    uint64_t xa[4], ya[4];
    std::memcpy(xa, x, 32);
    std::memcpy(ya, y, 32);
    for (int i = 3; i >= 0; i--) {
        if (xa[i] != ya[i]) return xa[i] < ya[i] ? -1 : 1;
    }
    return 0;
}

}  // namespace imt
