// imt_params.hpp -- host-side Poseidon parameter generation for the product library.
#pragma once
#include "imt_consts.hpp"
#include "imt_fr_host.hpp"
#include <string>
#include <vector>

namespace imt {

struct HostPoseidon {
    HField F;
    HFr rc[65][3];      // Grain round constants, plain schedule
    HFr mds[3][3];      // Cauchy MDS
    // optimised schedule (see imt_device.hpp::permute)
    HFr rc_full[8][3];
    HFr k_partial[57];
    HFr pre[3][3];
    HFr sp_row[57][3];
    HFr sp_col[57][2];
    HFr cap0;           // 2^64
    // halo2-base / pse-poseidon optimised spec, for the witness trace (f1): derived in init() from rc / mds by
    // the published Spec::new algorithm (calculate_optimized_constants, calculate_sparse_matrices)
    HFr tr_start[5][3], tr_partial[57], tr_end[3][3];
    HFr tr_pre[3][3];
    HFr tr_row[57][3], tr_col_hat[57][2];

    // Generates everything and cross-checks the optimised schedule against the plain
    // 65-round form on fixed states; returns false (with a message) on mismatch.
    bool init(std::string& err);

    void permute_plain(HFr s[3]) const;
    void permute_opt(HFr s[3]) const;
    void permute_spec(HFr s[3]) const;      // the same permutation in the tr_* (halo2-base) convention
    void fill_trace_consts(dev::TraceConsts& tc) const;
    HFr hash2(const HFr& a, const HFr& b) const;
    HFr hash3(const HFr& a, const HFr& b, const HFr& c) const;

    // device image: radix-2^29 limbs, Montgomery R = 2^261
    dev::Fe to_dev(const HFr& x) const;
    dev::Fe int_to_dev_limbs(const uint8_t le[32]) const;   // raw integer -> limbs, no domain change
    void fill_consts(dev::PoseidonConsts& pc) const;
};

}  // namespace imt
