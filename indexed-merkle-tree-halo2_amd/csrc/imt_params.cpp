// imt_params.cpp -- Poseidon T=3, RATE=2, R_F=8, R_P=57 constants for bn256::Fr.
//
// Restates (from the published algorithm; the crate is not vendored in
// /root/reference) what pse-poseidon's Spec::new(8, 57) produces for the call
//   Poseidon::<Fr, 3, 2>::new(8, 57)      /root/reference/src/indexed_merkle_tree.rs:370,663
// and halo2-base's OptimizedPoseidonSpec::new::<8, 57, 0>()   (same file :440-441):
// Grain-LFSR round constants, Cauchy MDS (SECURE_MDS = 0 -> first candidate), then an
// optimised schedule derived here:
//   * partial-round constants of the linear lanes are pushed forward through M into the
//     lane-0 constants and the first constants of the second half of the full rounds;
//   * M = S * N' with N' = diag(1, D) commuting with the partial S-box; every N' is pulled
//     back into the previous round, leaving one dense PRE matrix and 57 sparse ones.
// Any such schedule is value-identical to the plain form; init() checks that.
#include "imt_params.hpp"

namespace imt {

namespace {
struct Grain {
    uint8_t b[80];
    int pos = 0;
    int new_bit() {
        int p = pos;
        auto g = [&](int k) { return b[(p + k) % 80]; };
        int nb = g(62) ^ g(51) ^ g(38) ^ g(23) ^ g(13) ^ g(0);
        b[p] = (uint8_t)nb;
        pos = (p + 1) % 80;
        return nb;
    }
    int next() {   // bit pairs: (1, x) -> emit x; (0, x) -> drop
        while (!new_bit()) new_bit();
        return new_bit();
    }
    void take254(uint8_t le[32]) {   // most significant bit first
        std::memset(le, 0, 32);
        for (int i = 253; i >= 0; i--)
            if (next()) le[i / 8] |= (uint8_t)(1u << (i % 8));
    }
};
}  // namespace

void HostPoseidon::permute_plain(HFr s[3]) const {
    for (int r = 0; r < 65; r++) {
        for (int i = 0; i < 3; i++) s[i] = F.add(s[i], rc[r][i]);
        const int nsb = (r < 4 || r >= 61) ? 3 : 1;
        for (int i = 0; i < nsb; i++) {
            HFr x2 = F.mul(s[i], s[i]), x4 = F.mul(x2, x2);
            s[i] = F.mul(x4, s[i]);
        }
        HFr n[3];
        for (int i = 0; i < 3; i++) {
            n[i] = F.mul(mds[i][0], s[0]);
            n[i] = F.add(n[i], F.mul(mds[i][1], s[1]));
            n[i] = F.add(n[i], F.mul(mds[i][2], s[2]));
        }
        s[0] = n[0]; s[1] = n[1]; s[2] = n[2];
    }
}

void HostPoseidon::permute_opt(HFr s[3]) const {
    auto sbox = [&](HFr& x) {
        HFr x2 = F.mul(x, x), x4 = F.mul(x2, x2);
        x = F.mul(x4, x);
    };
    auto dense = [&](const HFr m[3][3]) {
        HFr n[3];
        for (int i = 0; i < 3; i++) {
            n[i] = F.mul(m[i][0], s[0]);
            n[i] = F.add(n[i], F.mul(m[i][1], s[1]));
            n[i] = F.add(n[i], F.mul(m[i][2], s[2]));
        }
        s[0] = n[0]; s[1] = n[1]; s[2] = n[2];
    };
    for (int r = 0; r < 65; r++) {
        if (r < 4 || r >= 61) {
            const int fr = r < 4 ? r : r - 57;
            for (int i = 0; i < 3; i++) s[i] = F.add(s[i], rc_full[fr][i]);
            for (int i = 0; i < 3; i++) sbox(s[i]);
            dense(r == 3 ? pre : mds);
        } else {
            const int p = r - 4;
            HFr y = F.add(s[0], k_partial[p]);
            sbox(y);
            HFr n0 = F.mul(sp_row[p][0], y);
            n0 = F.add(n0, F.mul(sp_row[p][1], s[1]));
            n0 = F.add(n0, F.mul(sp_row[p][2], s[2]));
            s[1] = F.add(s[1], F.mul(sp_col[p][0], y));
            s[2] = F.add(s[2], F.mul(sp_col[p][1], y));
            s[0] = n0;
        }
    }
}

// The permutation as halo2-base's PoseidonState::permutation / pse-poseidon's Spec::permute run it (no
// absorption here: the caller has added its inputs): s += start[0]; three rounds {x^5 + c, M}; one round
// {x^5 + c, PRE}; 57 rounds {lane 0: x^5 + c; sparse}; three rounds {x^5 + c, M}; one round {x^5, M}.
void HostPoseidon::permute_spec(HFr s[3]) const {
    auto x5c = [&](HFr& x, const HFr& c) {
        HFr x2 = F.mul(x, x), x4 = F.mul(x2, x2);
        x = F.add(F.mul(x4, x), c);
    };
    auto dense = [&](const HFr m[3][3]) {
        HFr n[3];
        for (int i = 0; i < 3; i++) {
            n[i] = F.mul(m[i][0], s[0]);
            n[i] = F.add(n[i], F.mul(m[i][1], s[1]));
            n[i] = F.add(n[i], F.mul(m[i][2], s[2]));
        }
        s[0] = n[0]; s[1] = n[1]; s[2] = n[2];
    };
    for (int i = 0; i < 3; i++) s[i] = F.add(s[i], tr_start[0][i]);
    for (int r = 1; r <= 4; r++) {
        for (int i = 0; i < 3; i++) x5c(s[i], tr_start[r][i]);
        dense(r == 4 ? tr_pre : mds);
    }
    for (int p = 0; p < 57; p++) {
        x5c(s[0], tr_partial[p]);
        HFr n0 = F.mul(tr_row[p][0], s[0]);
        n0 = F.add(n0, F.mul(tr_row[p][1], s[1]));
        n0 = F.add(n0, F.mul(tr_row[p][2], s[2]));
        s[1] = F.add(F.mul(tr_col_hat[p][0], s[0]), s[1]);
        s[2] = F.add(F.mul(tr_col_hat[p][1], s[0]), s[2]);
        s[0] = n0;
    }
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 3; i++) x5c(s[i], r < 3 ? tr_end[r][i] : F.zero());
        dense(mds);
    }
}

// Poseidon::update + squeeze_and_reset for 2 / 3 inputs
HFr HostPoseidon::hash2(const HFr& a, const HFr& b) const {
    HFr s[3] = {cap0, a, b};
    permute_opt(s);
    s[1] = F.add(s[1], F.one());
    permute_opt(s);
    return s[1];
}
HFr HostPoseidon::hash3(const HFr& a, const HFr& b, const HFr& c) const {
    HFr s[3] = {cap0, a, b};
    permute_opt(s);
    s[1] = F.add(s[1], c);
    s[2] = F.add(s[2], F.one());
    permute_opt(s);
    return s[1];
}

bool HostPoseidon::init(std::string& err) {
    // ---- Grain LFSR (Poseidon paper, appendix F) ----
    Grain g;
    int n = 0;
    auto put = [&](int width, unsigned v) {
        for (int i = width - 1; i >= 0; i--) g.b[n++] = (v >> i) & 1;
    };
    put(2, 1); put(4, 0); put(12, 254); put(12, 3); put(10, 8); put(10, 57); put(30, 0x3fffffffu);
    for (int i = 0; i < 160; i++) g.new_bit();
    uint8_t le[32];
    for (int r = 0; r < 65; r++)
        for (int i = 0; i < 3; i++) {
            do { g.take254(le); } while (!F.from_bytes(rc[r][i], le));   // rejection sampling
        }
    HFr xs[3], ys[3];
    auto norej = [&](HFr& o) {   // value mod p (254 bits < 2p: one subtraction at most)
        g.take254(le);
        uint64_t v[4];
        std::memcpy(v, le, 32);
        if (HField::geq_p(v)) HField::sub_p(v);
        std::memcpy(le, v, 32);
        F.from_bytes(o, le);
    };
    for (auto& x : xs) norej(x);
    for (auto& y : ys) norej(y);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) mds[i][j] = F.inverse(F.add(xs[i], ys[j]));
    HFr two32 = F.from_u64(1ULL << 32);
    cap0 = F.mul(two32, two32);

    // ---- constants of the optimised schedule ----
    auto matvec = [&](const HFr m[3][3], const HFr v[3], HFr o[3]) {
        for (int i = 0; i < 3; i++) {
            o[i] = F.mul(m[i][0], v[0]);
            o[i] = F.add(o[i], F.mul(m[i][1], v[1]));
            o[i] = F.add(o[i], F.mul(m[i][2], v[2]));
        }
    };
    for (int r = 0; r < 4; r++)
        for (int i = 0; i < 3; i++) rc_full[r][i] = rc[r][i];
    HFr w[3] = {F.zero(), F.zero(), F.zero()};
    for (int p = 4; p <= 60; p++) {
        HFr v[3], mw[3];
        matvec(mds, w, mw);
        for (int i = 0; i < 3; i++) v[i] = F.add(mw[i], rc[p][i]);
        k_partial[p - 4] = v[0];
        w[0] = F.zero(); w[1] = v[1]; w[2] = v[2];
    }
    {
        HFr mw[3];
        matvec(mds, w, mw);
        for (int i = 0; i < 3; i++) rc_full[4][i] = F.add(rc[61][i], mw[i]);
    }
    for (int r = 62; r < 65; r++)
        for (int i = 0; i < 3; i++) rc_full[r - 57][i] = rc[r][i];

    // ---- sparse factorisation, last partial round first ----
    HFr cur[3][3];
    std::memcpy(cur, mds, sizeof cur);
    for (int p = 60; p >= 4; p--) {
        // cur = [[a, b^T],[c, D]] = [[a, b^T D^-1],[c, I]] * diag(1, D)
        HFr det = F.sub(F.mul(cur[1][1], cur[2][2]), F.mul(cur[1][2], cur[2][1]));
        if (F.is_zero(det)) { err = "singular MDS sub-block"; return false; }
        HFr di = F.inverse(det);
        HFr Dinv[2][2] = {{F.mul(cur[2][2], di), F.mul(F.sub(F.zero(), cur[1][2]), di)},
                          {F.mul(F.sub(F.zero(), cur[2][1]), di), F.mul(cur[1][1], di)}};
        sp_row[p - 4][0] = cur[0][0];
        sp_row[p - 4][1] = F.add(F.mul(cur[0][1], Dinv[0][0]), F.mul(cur[0][2], Dinv[1][0]));
        sp_row[p - 4][2] = F.add(F.mul(cur[0][1], Dinv[0][1]), F.mul(cur[0][2], Dinv[1][1]));
        sp_col[p - 4][0] = cur[1][0];
        sp_col[p - 4][1] = cur[2][0];
        // next (earlier) round's matrix: N' * M
        HFr np[3][3] = {{F.one(), F.zero(), F.zero()},
                        {F.zero(), cur[1][1], cur[1][2]},
                        {F.zero(), cur[2][1], cur[2][2]}};
        HFr nxt[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                nxt[i][j] = F.mul(np[i][0], mds[0][j]);
                nxt[i][j] = F.add(nxt[i][j], F.mul(np[i][1], mds[1][j]));
                nxt[i][j] = F.add(nxt[i][j], F.mul(np[i][2], mds[2][j]));
            }
        std::memcpy(cur, nxt, sizeof cur);
    }
    std::memcpy(pre, cur, sizeof pre);

    // ---- the halo2-base / pse-poseidon spec (Spec::new): needed value by value by the witness trace ----
    {
        typedef HFr M3[3][3];
        auto mat_mul = [&](const M3 a, const M3 b, M3 o) {
            HFr r[3][3];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    r[i][j] = F.mul(a[i][0], b[0][j]);
                    r[i][j] = F.add(r[i][j], F.mul(a[i][1], b[1][j]));
                    r[i][j] = F.add(r[i][j], F.mul(a[i][2], b[2][j]));
                }
            std::memcpy(o, r, sizeof r);
        };
        // M^-1 by the adjugate
        HFr minv[3][3];
        {
            auto cof = [&](int r0, int r1, int c0, int c1) {
                return F.sub(F.mul(mds[r0][c0], mds[r1][c1]), F.mul(mds[r0][c1], mds[r1][c0]));
            };
            HFr adj[3][3];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    const int r0 = (j + 1) % 3, r1 = (j + 2) % 3, c0 = (i + 1) % 3, c1 = (i + 2) % 3;
                    adj[i][j] = cof(r0, r1, c0, c1);       // cyclic indices carry the sign
                }
            HFr det = F.add(F.add(F.mul(mds[0][0], adj[0][0]), F.mul(mds[0][1], adj[1][0])), F.mul(mds[0][2], adj[2][0]));
            if (F.is_zero(det)) { err = "singular MDS"; return false; }
            HFr di = F.inverse(det);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) minv[i][j] = F.mul(adj[i][j], di);
        }
        // calculate_optimized_constants: start[0] = c_0; start[r] = M^-1 c_r; the partial rounds' constants are
        // folded back to front from c_61 (lane 0 stays, lanes 1..2 travel on through M^-1); end[r] = M^-1 c_{62+r}
        for (int i = 0; i < 3; i++) tr_start[0][i] = rc[0][i];
        for (int r = 1; r < 4; r++) matvec(minv, rc[r], tr_start[r]);
        HFr acc[3] = {rc[61][0], rc[61][1], rc[61][2]};
        for (int p = 56; p >= 0; p--) {
            HFr tmp[3];
            matvec(minv, acc, tmp);
            tr_partial[p] = tmp[0];
            acc[0] = rc[4 + p][0];
            acc[1] = F.add(tmp[1], rc[4 + p][1]);
            acc[2] = F.add(tmp[2], rc[4 + p][2]);
        }
        matvec(minv, acc, tr_start[4]);
        for (int r = 0; r < 3; r++) matvec(minv, rc[62 + r], tr_end[r]);
        // calculate_sparse_matrices: A = M^T; 57 x { A = M' M'' (M' = diag(1, hat(A)), M'' = [[A00, A0*],[w_hat, I]],
        // w_hat = hat(A)^-1 (A10, A20)^T); sparse = M''^T; A = M^T M' }, reversed; pre_sparse = A^T
        HFr mt[3][3], a[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) mt[i][j] = a[i][j] = mds[j][i];
        for (int k = 0; k < 57; k++) {
            const int p = 56 - k;
            HFr det = F.sub(F.mul(a[1][1], a[2][2]), F.mul(a[1][2], a[2][1]));
            if (F.is_zero(det)) { err = "singular MDS minor"; return false; }
            HFr di = F.inverse(det);
            HFr hi[2][2] = {{F.mul(a[2][2], di), F.mul(F.sub(F.zero(), a[1][2]), di)},
                            {F.mul(F.sub(F.zero(), a[2][1]), di), F.mul(a[1][1], di)}};
            tr_row[p][0] = a[0][0];
            tr_row[p][1] = F.add(F.mul(hi[0][0], a[1][0]), F.mul(hi[0][1], a[2][0]));
            tr_row[p][2] = F.add(F.mul(hi[1][0], a[1][0]), F.mul(hi[1][1], a[2][0]));
            tr_col_hat[p][0] = a[0][1];
            tr_col_hat[p][1] = a[0][2];
            HFr prime[3][3] = {{F.one(), F.zero(), F.zero()}, {F.zero(), a[1][1], a[1][2]}, {F.zero(), a[2][1], a[2][2]}};
            mat_mul(mt, prime, a);
        }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) tr_pre[i][j] = a[j][i];
    }

    // ---- self-check: optimised == plain ----
    for (uint64_t t = 0; t < 4; t++) {
        HFr a[3] = {F.from_u64(t * 7919), F.from_u64(t * t + 1), F.mul(cap0, F.from_u64(t + 3))};
        HFr b[3] = {a[0], a[1], a[2]};
        permute_plain(a);
        permute_opt(b);
        if (!(a[0] == b[0] && a[1] == b[1] && a[2] == b[2])) {
            err = "optimised Poseidon schedule disagrees with the plain form";
            return false;
        }
        HFr c3[3] = {F.from_u64(t * 7919), F.from_u64(t * t + 1), F.mul(cap0, F.from_u64(t + 3))};
        // permute_spec adds start[0] itself (absorb_with_pre_constants does it in the gadget); the plain form adds
        // c_0 in round 0, and start[0] == c_0
        permute_spec(c3);
        if (!(a[0] == c3[0] && a[1] == c3[1] && a[2] == c3[2])) {
            err = "halo2-base form of the Poseidon schedule disagrees with the plain form";
            return false;
        }
    }
    return true;
}

void HostPoseidon::fill_trace_consts(dev::TraceConsts& tc) const {
    for (int i = 0; i < 3; i++) {
        tc.absorb[0][i] = to_dev(tr_start[0][i]);                                        // two inputs: no padding lane
        tc.absorb[1][i] = to_dev(i == 2 ? F.add(tr_start[0][i], F.one()) : tr_start[0][i]);   // one input: 1 on lane 2
        tc.absorb[2][i] = to_dev(i == 1 ? F.add(tr_start[0][i], F.one()) : tr_start[0][i]);   // no input: 1 on lane 1
    }
    for (int f = 0; f < 8; f++)
        for (int i = 0; i < 3; i++)
            tc.full_c[f][i] = to_dev(f < 4 ? tr_start[f + 1][i] : f < 7 ? tr_end[f - 4][i] : F.zero());
    for (int p = 0; p < 57; p++) {
        tc.partial[p] = to_dev(tr_partial[p]);
        for (int i = 0; i < 3; i++) tc.row[p][i] = to_dev(tr_row[p][i]);
        for (int i = 0; i < 2; i++) tc.col_hat[p][i] = to_dev(tr_col_hat[p][i]);
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            tc.mats[0][i][j] = to_dev(mds[i][j]);
            tc.mats[1][i][j] = to_dev(tr_pre[i][j]);
        }
}

dev::Fe HostPoseidon::int_to_dev_limbs(const uint8_t le[32]) const {
    dev::Fe o;
    for (int i = 0; i < dev::NL; i++) {
        uint32_t v = 0;
        for (int k = 0; k < 29; k++) {
            int bit = 29 * i + k;
            if (bit < 256 && ((le[bit / 8] >> (bit % 8)) & 1)) v |= 1u << k;
        }
        o.v[i] = v;
    }
    return o;
}

dev::Fe HostPoseidon::to_dev(const HFr& x) const {
    // device form = x * 2^261 mod p, as a plain integer split in 29-bit limbs
    HFr two29 = F.from_u64(1ULL << 29);
    HFr t = x;
    for (int i = 0; i < 9; i++) t = F.mul(t, two29);
    uint8_t le[32];
    F.to_bytes(le, t);
    return int_to_dev_limbs(le);
}

void HostPoseidon::fill_consts(dev::PoseidonConsts& pc) const {
    for (int r = 0; r < 8; r++)
        for (int i = 0; i < 3; i++) pc.rc_full[r][i] = to_dev(rc_full[r][i]);
    pc.rc_h2p2[0] = to_dev(rc_full[0][0]);
    pc.rc_h2p2[1] = to_dev(F.add(rc_full[0][1], F.one()));
    pc.rc_h2p2[2] = to_dev(rc_full[0][2]);
    pc.sp_gamma[0] = to_dev(F.zero());
    for (int p = 1; p < 57; p++)
        pc.sp_gamma[p] = to_dev(F.add(F.mul(sp_row[p][1], sp_col[p - 1][0]), F.mul(sp_row[p][2], sp_col[p - 1][1])));
    for (int p = 0; p < 57; p++) {
        pc.k_partial[p] = to_dev(k_partial[p]);
        for (int i = 0; i < 3; i++) pc.sp_row[p][i] = to_dev(sp_row[p][i]);
        for (int i = 0; i < 2; i++) pc.sp_col[p][i] = to_dev(sp_col[p][i]);
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            pc.mats[0][i][j] = to_dev(mds[i][j]);
            pc.mats[1][i][j] = to_dev(pre[i][j]);
        }
    pc.cap0 = to_dev(cap0);
    pc.one = to_dev(F.one());
    // conversions are single Montgomery products (divide by 2^261) on the device
    HFr two29 = F.from_u64(1ULL << 29), r261 = F.one();
    for (int i = 0; i < 9; i++) r261 = F.mul(r261, two29);          // 2^261
    HFr two64 = cap0, two256 = F.mul(F.mul(two64, two64), F.mul(two64, two64));
    uint8_t le[32];
    auto as_int = [&](const HFr& x) { F.to_bytes(le, x); return int_to_dev_limbs(le); };
    pc.from_canon = as_int(F.mul(r261, r261));                                   // x * R^2 / R
    pc.from_mont256 = as_int(F.mul(F.mul(r261, r261), F.inverse(two256)));       // x 2^256 * (R^2/2^256) / R
    pc.to_mont256 = as_int(two256);                                              // x R * 2^256 / R
    std::memset(le, 0, 32); le[0] = 1;
    pc.int_one = int_to_dev_limbs(le);                                           // x R * 1 / R
    HFr z = F.zero();
    pc.zero_leaf = to_dev(hash3(z, z, z));
}

}  // namespace imt
