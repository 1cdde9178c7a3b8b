// lane_mont.hip -- ONE Montgomery product spread over the 16 lanes of a DPP row, measured.
//
// The latency of a single call (one verify_proof: 33 hashes x 2 permutations x 203 DEPENDENT products) is the
// 196-instruction chain of one product x ~5 cycles per instruction for a lone wave.  Here a product is spread over a row:
// lane c holds limb c (nine 29-bit limbs in lanes 0..8, Montgomery R = 2^261, as in csrc/imt_device.hpp), four rows per
// wave = four independent products per instruction stream (the three state lanes of a permutation + one spare):
//   T = a * b        column c in lane c: 9 x (row_newbcast of a_i, row_shr:i of b, v_mad_u64_u32)   (+ column 16 in lane 15)
//   two carry-save passes bring every limb below 2^29 + 2^5
//   m = T_lo * n' mod R: 9 broadcasts, 9 mads against per-lane constants n'_(c-i); two passes; lanes >= 9 dropped
//   U = m * p:           9 broadcasts, 9 mads against per-lane constants p_(c-i)  (+ column 16)
//   S = T + U; two passes; the low nine limbs now sum to K * 2^261 with K in {0, 1}: K = "any of them non-zero" (a ballot)
//   result limb j = column 9 + j  (row_shl:9; columns 16 and 17 come from lane 15's top accumulator) + K in limb 0
// The kernel runs `iters` dependent products x <- x * b per row and reports the GPU's own clock per product, next to the
// single-lane product the library uses today (csrc/imt_mont_asm.hpp through imt_device.hpp: 196 instructions), same
// launch shape (one wave), same chain.  tools/lane_mont_check.py checks the row form against Python integers.
//
// Build: hipcc --offload-arch=gfx950 -O3 -I indexed-merkle-tree-halo2_amd/csrc -o tools/microbench/lane_mont tools/microbench/lane_mont.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "imt_device.hpp"

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

static const uint32_t H_P[9] = {0x10000001, 0x1f0fac9f, 0xe5c2450, 0x7d090f3, 0x1585d283, 0x2db40c0, 0xa6e141, 0xe5c2634, 0x30644e};
static const uint32_t H_NP[9] = {0xfffffff, 0x170fac9f, 0x1a446cf0, 0xd0c9698, 0x2391658, 0xc144c83, 0x6cb8e6a, 0x3a1b068, 0x1273f82f};
static const uint32_t H_A[9] = {0x117fd374, 0x1e0f51b7, 0x8cc954f, 0xc82714c, 0x16a3b0d4, 0x1446f350, 0x3d8a09d, 0xbe39f62, 0x17cb76};
static const uint32_t H_B[9] = {0x1814e8a2, 0x12938803, 0x7d96a37, 0x12b39c7e, 0x1e968617, 0x1f43c599, 0x1d14686b, 0x1ad25db9, 0xff508};

constexpr uint32_t M29 = (1u << 29) - 1;

template <int I>
__device__ __forceinline__ uint32_t bcast(uint32_t v) {       // every lane of a row gets lane I's value
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + I, 0xf, 0xf, false);
}
template <int N>
__device__ __forceinline__ uint32_t shr(uint32_t v) {         // lane c gets lane c - N's value, 0 below
    if constexpr (N == 0) return v;
    else return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, true);
}
template <int N>
__device__ __forceinline__ uint32_t shl(uint32_t v) {         // lane c gets lane c + N's value, 0 above
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xf, 0xf, true);
}

// one carry-save pass over the 17 columns (lanes 0..15 in `col`, column 16 in lane 15's `top`): limb = low 29 bits, the
// rest moves one column up
__device__ __forceinline__ void pass64(uint64_t& col, uint64_t& top, bool lane15) {
    const uint32_t lo = (uint32_t)col & M29;
    const uint64_t hi = col >> 29;
    const uint32_t h0 = (uint32_t)hi, h1 = (uint32_t)(hi >> 32);
    const uint64_t in = (uint64_t)shr<1>(h0) | ((uint64_t)shr<1>(h1) << 32);
    col = (uint64_t)lo + in;
    if (lane15) top += hi;
}
__device__ __forceinline__ void pass32(uint64_t& col, uint64_t& top, bool lane15) {      // columns already below 2^61 / 2^29: a 32-bit carry
    const uint32_t lo = (uint32_t)col & M29, hi = (uint32_t)(col >> 29);
    col = (uint64_t)(lo + shr<1>(hi));
    if (lane15) top += hi;
}

struct RowConsts {
    uint32_t np[9], p[9];       // this lane's n'_(c - i) and p_(c - i), 0 outside 0..8
};

// x, y: this lane's limb (0 in lanes 9..15).  Returns this lane's limb of x * y / R mod p (redundant: limbs < 2^29 + 2^5,
// value < 2p + small)
__device__ __forceinline__ uint32_t mont_row(uint32_t x, uint32_t y, const RowConsts& k, unsigned c) {
    const bool lane15 = c == 15;
    uint64_t T = 0, top = 0;
#define TERM(I) T += (uint64_t)bcast<I>(x) * shr<I>(y);
    TERM(0) TERM(1) TERM(2) TERM(3) TERM(4) TERM(5) TERM(6) TERM(7) TERM(8)
#undef TERM
    top = (uint64_t)bcast<8>(x) * shr<7>(y);             // lane 15: a_8 * b_8 = column 16 (other lanes: unused)
    pass64(T, top, lane15);
    pass32(T, top, lane15);
    // m = T_lo * n' mod R
    const uint32_t t = c < 9 ? (uint32_t)T : 0u;
    uint64_t Mc = 0, dummy = 0;
#define TERM(I) Mc += (uint64_t)bcast<I>(t) * k.np[I];
    TERM(0) TERM(1) TERM(2) TERM(3) TERM(4) TERM(5) TERM(6) TERM(7) TERM(8)
#undef TERM
    pass64(Mc, dummy, false);
    pass32(Mc, dummy, false);
    const uint32_t m = c < 9 ? (uint32_t)Mc : 0u;
    // S = T + m * p
    uint64_t S = T;
#define TERM(I) S += (uint64_t)bcast<I>(m) * k.p[I];
    TERM(0) TERM(1) TERM(2) TERM(3) TERM(4) TERM(5) TERM(6) TERM(7) TERM(8)
#undef TERM
    top += (uint64_t)bcast<8>(m) * 0x30644eull;         // lane 15: m_8 * p_8 = column 16 of m * p
    pass64(S, top, lane15);
    pass32(S, top, lane15);
    // the low nine limbs sum to K * 2^261, K in {0, 1}
    const uint64_t nz = __ballot(c < 9 && (uint32_t)S != 0);
    const unsigned row_shift = (threadIdx.x & 48u);
    const uint32_t K = ((uint32_t)(nz >> row_shift) & 0x1ffu) != 0u;
    // result limb j = column 9 + j: lanes 9..15 -> 0..6; columns 16, 17 from lane 15's top
    const uint32_t s = (uint32_t)S;
    uint32_t r = shl<9>(s);
    const uint32_t top_lo = (uint32_t)top & M29, top_hi = (uint32_t)(top >> 29);
    const uint32_t r7 = shl<8>(top_lo), r8 = shl<7>(top_hi);
    r = c == 7 ? r7 : (c == 8 ? r8 : (c < 7 ? r : 0u));
    if (c == 0) r += K;
    return r;
}

__global__ void k_row_chain(const uint32_t* a, const uint32_t* b, const RowConsts* consts, uint32_t* out, uint64_t* cycles, int iters) {
    const unsigned c = threadIdx.x & 15u;
    const RowConsts k = consts[c];
    uint32_t x = c < 9 ? a[c] : 0u, y = c < 9 ? b[c] : 0u;
    const uint64_t t0 = wall_clock64();
    const uint64_t c0 = clock64();
    for (int i = 0; i < iters; i++) x = mont_row(x, y, k, c);
    const uint64_t c1 = clock64();
    const uint64_t t1 = wall_clock64();
    if (threadIdx.x < 16 && c < 9) out[c] = x;
    if (threadIdx.x == 0) {
        cycles[0] = c1 - c0;
        cycles[1] = t1 - t0;
    }
}

// the library's single-lane product, the same chain on every lane of one wave
__global__ void k_lane_chain(const uint32_t* a, const uint32_t* b, uint32_t* out, uint64_t* cycles, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    imt::dev::Fe x, y;
    for (int i = 0; i < 9; i++) {
        x.v[i] = a[i];
        y.v[i] = b[i];
    }
    const uint64_t c0 = clock64();
    const uint64_t t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        imt::dev::Fe r;
        imt::dev::masm::mul_vv_narrow(r, &x, &y);
        x = r;
    }
    const uint64_t t1 = wall_clock64();
    const uint64_t c1 = clock64();
    if (threadIdx.x == 0) {
        for (int i = 0; i < 9; i++) out[i] = x.v[i];
        cycles[0] = c1 - c0;
        cycles[1] = t1 - t0;
    }
#endif
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    CHECK(hipSetDevice(0));
    uint32_t *d_a, *d_b, *d_out;
    uint64_t* d_cyc;
    RowConsts h_k[16], *d_k;
    for (int c = 0; c < 16; c++)
        for (int i = 0; i < 9; i++) {
            const int j = c - i;
            h_k[c].np[i] = (j >= 0 && j <= 8) ? H_NP[j] : 0u;
            h_k[c].p[i] = (j >= 0 && j <= 8) ? H_P[j] : 0u;
        }
    CHECK(hipMalloc((void**)&d_a, 36));
    CHECK(hipMalloc((void**)&d_b, 36));
    CHECK(hipMalloc((void**)&d_out, 36));
    CHECK(hipMalloc((void**)&d_cyc, 16));
    CHECK(hipMalloc((void**)&d_k, sizeof h_k));
    CHECK(hipMemcpy(d_a, H_A, 36, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_b, H_B, 36, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_k, h_k, sizeof h_k, hipMemcpyHostToDevice));
    int khz = 100000;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    uint32_t out[9];
    uint64_t cyc[2];
    for (int rep = 0; rep < 2; rep++) {          // second pass = warm
        hipLaunchKernelGGL(k_row_chain, dim3(1), dim3(64), 0, 0, d_a, d_b, d_k, d_out, d_cyc, iters);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(out, d_out, 36, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(cyc, d_cyc, 16, hipMemcpyDeviceToHost));
    printf("row iters %d limbs", iters);
    for (int i = 0; i < 9; i++) printf(" %x", out[i]);
    printf("\nrow form: %.1f ns per product (%.0f shader clocks), four products per wave in flight\n", cyc[1] * 1e6 / khz / iters, (double)cyc[0] / iters);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_lane_chain, dim3(1), dim3(64), 0, 0, d_a, d_b, d_out, d_cyc, iters);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(out, d_out, 36, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(cyc, d_cyc, 16, hipMemcpyDeviceToHost));
    printf("lane iters %d limbs", iters);
    for (int i = 0; i < 9; i++) printf(" %x", out[i]);
    printf("\nlane form (the library's mul_vv_narrow): %.1f ns per product (%.0f shader clocks)\n", cyc[1] * 1e6 / khz / iters, (double)cyc[0] / iters);
    return 0;
}
