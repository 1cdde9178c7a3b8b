// valu_rates.hip -- instruction-throughput microbenchmark for gfx950 (MI355X).
// Decides which Fr multiplier the Poseidon kernels use: 32-bit v_mad_u64_u32
// limbs vs FP64-FMA 52-bit limbs vs 24-bit multiplies.  Each kernel runs ITER
// iterations of 8 independent dependency chains of ONE instruction (inline asm so
// the compiler cannot substitute another); the host reports wave-instructions per
// cycle per SIMD assuming the 2.4 GHz spec clock, and the raw G-inst/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITER = 4096;

#define KERNEL32(NAME, ASM)                                                        \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned seed) {        \
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
    unsigned a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7; \
    unsigned b = a0 | 1, c = seed * 77 + 12345;                                    \
    for (int i = 0; i < ITER; i++) {                                               \
        asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(b), "v"(c) : "vcc", "s10", "s11");                      \
    }                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
}

#define KERNEL64(NAME, ASM)                                                        \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned seed) {        \
    unsigned long long a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
    unsigned long long a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7; \
    unsigned b = (unsigned)a0 | 1, c = seed * 77 + 12345;                          \
    for (int i = 0; i < ITER; i++) {                                               \
        asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(b), "v"(c) : "vcc");                                    \
    }                                                                              \
    unsigned long long r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)r ^ (unsigned)(r >> 32); \
}

#define KERNELD(NAME, ASM)                                                         \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned seed) {        \
    double a0 = 1.0 + threadIdx.x * 1e-3 + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
    double a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7; \
    double b = 1.0000001, c = 1e-9 * seed;                                         \
    for (int i = 0; i < ITER; i++) {                                               \
        asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(b), "v"(c) : "vcc");                                    \
    }                                                                              \
    double r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)__double_as_longlong(r); \
}

// 32-bit ops
#define A_FMA_F32(x)      "v_fma_f32 " #x ", " #x ", %8, %9\n"
#define A_ADD_U32(x)      "v_add_u32 " #x ", " #x ", %8\n"
#define A_ADD3_U32(x)     "v_add3_u32 " #x ", " #x ", %8, %9\n"
#define A_MUL_LO_U32(x)   "v_mul_lo_u32 " #x ", " #x ", %8\n"
#define A_MUL_HI_U32(x)   "v_mul_hi_u32 " #x ", " #x ", %8\n"
#define A_MAD_U32_U24(x)  "v_mad_u32_u24 " #x ", " #x ", %8, %9\n"
#define A_MUL_U32_U24(x)  "v_mul_u32_u24 " #x ", " #x ", %8\n"
#define A_MUL_HI_U24(x)   "v_mul_hi_u32_u24 " #x ", " #x ", %8\n"
#define A_ADD_CO(x)       "v_add_co_u32 " #x ", vcc, " #x ", %8\n"
#define A_ADDC_CO(x)      "v_addc_co_u32 " #x ", vcc, " #x ", %8, vcc\n"
#define A_ALIGNBIT(x)     "v_alignbit_b32 " #x ", " #x ", %8, 7\n"
#define A_LSHL_OR(x)      "v_lshl_or_b32 " #x ", " #x ", 3, %8\n"
#define A_AND_OR(x)       "v_and_or_b32 " #x ", " #x ", %8, %9\n"
#define A_DOT4_U8(x)      "v_dot4_u32_u8 " #x ", " #x ", %8, %9\n"
#define A_MAD_U32_U16(x)  "v_mad_u32_u16 " #x ", " #x ", %8, %9\n"
#define A_PK_FMA_F32(x)   "v_pk_fma_f32 " #x ", " #x ", " #x ", " #x "\n"
#define A_CNDMASK(x)      "v_cndmask_b32 " #x ", " #x ", %8, vcc\n"
#define A_CNDMASK_E64(x)  "v_cndmask_b32_e64 " #x ", " #x ", %8, s[10:11]\n"
#define A_CNDMASK_ADD(x)  "v_cndmask_b32 " #x ", " #x ", %8, vcc\n v_add_u32 " #x ", " #x ", %9\n"
#define A_CNDMASK_MAD(x)  "v_cndmask_b32 " #x ", " #x ", %8, vcc\n v_mul_lo_u32 " #x ", " #x ", %9\n v_mul_lo_u32 " #x ", " #x ", %9\n v_mul_lo_u32 " #x ", " #x ", %9\n"
#define A_BFI_B32(x)      "v_bfi_b32 " #x ", %8, " #x ", %9\n"
#define A_AND_B32(x)      "v_and_b32 " #x ", " #x ", %8\n"
#define A_OR_B32(x)       "v_or_b32 " #x ", " #x ", %8\n"
#define A_LSHRREV_B32(x)  "v_lshrrev_b32 " #x ", 3, " #x "\n"
#define A_LSHLREV_B32(x)  "v_lshlrev_b32 " #x ", 3, " #x "\n"
#define A_SUB_U32(x)      "v_sub_u32 " #x ", " #x ", %8\n"
#define A_MOV_B32(x)      "v_mov_b32 " #x ", %8\n"
#define A_ASHRREV_I32(x)  "v_ashrrev_i32 " #x ", 3, " #x "\n"
#define A_XOR_B32(x)      "v_xor_b32 " #x ", " #x ", %8\n"
// 64-bit destination ops (x is a VGPR pair)
#define A_MAD_U64_U32(x)  "v_mad_u64_u32 " #x ", vcc, %8, %9, " #x "\n"
#define A_LSHL_ADD_U64(x) "v_lshl_add_u64 " #x ", " #x ", 0, " #x "\n"
#define A_LSHLREV_B64(x)  "v_lshlrev_b64 " #x ", 1, " #x "\n"
#define A_LSHRREV_B64(x)  "v_lshrrev_b64 " #x ", 29, " #x "\n"
#define A_ASHRREV_I64(x)  "v_ashrrev_i64 " #x ", 29, " #x "\n"
#define A_MAD_I64_I32(x)  "v_mad_i64_i32 " #x ", vcc, %8, %9, " #x "\n"
// f64 ops
#define A_FMA_F64(x)      "v_fma_f64 " #x ", " #x ", %8, %9\n"
#define A_ADD_F64(x)      "v_add_f64 " #x ", " #x ", %9\n"
#define A_MUL_F64(x)      "v_mul_f64 " #x ", " #x ", %8\n"

KERNEL32(k_fma_f32, A_FMA_F32)
KERNEL32(k_add_u32, A_ADD_U32)
KERNEL32(k_add3_u32, A_ADD3_U32)
KERNEL32(k_mul_lo_u32, A_MUL_LO_U32)
KERNEL32(k_mul_hi_u32, A_MUL_HI_U32)
KERNEL32(k_mad_u32_u24, A_MAD_U32_U24)
KERNEL32(k_mul_u32_u24, A_MUL_U32_U24)
KERNEL32(k_mul_hi_u24, A_MUL_HI_U24)
KERNEL32(k_add_co, A_ADD_CO)
KERNEL32(k_addc_co, A_ADDC_CO)
KERNEL32(k_alignbit, A_ALIGNBIT)
KERNEL32(k_lshl_or, A_LSHL_OR)
KERNEL32(k_and_or, A_AND_OR)
KERNEL32(k_dot4_u8, A_DOT4_U8)
KERNEL32(k_mad_u32_u16, A_MAD_U32_U16)
KERNEL32(k_cndmask, A_CNDMASK)
KERNEL32(k_cndmask_e64, A_CNDMASK_E64)
KERNEL32(k_cndmask_add, A_CNDMASK_ADD)
KERNEL32(k_cndmask_mad, A_CNDMASK_MAD)
KERNEL32(k_bfi_b32, A_BFI_B32)
KERNEL32(k_and_b32, A_AND_B32)
KERNEL32(k_or_b32, A_OR_B32)
KERNEL32(k_lshrrev_b32, A_LSHRREV_B32)
KERNEL32(k_lshlrev_b32, A_LSHLREV_B32)
KERNEL32(k_sub_u32, A_SUB_U32)
KERNEL32(k_mov_b32, A_MOV_B32)
KERNEL32(k_ashrrev_i32, A_ASHRREV_I32)
KERNEL32(k_xor_b32, A_XOR_B32)
KERNEL64(k_lshrrev_b64, A_LSHRREV_B64)
KERNEL64(k_ashrrev_i64, A_ASHRREV_I64)
KERNEL64(k_mad_i64_i32, A_MAD_I64_I32)
KERNEL64(k_pk_fma_f32, A_PK_FMA_F32)
KERNEL64(k_mad_u64_u32, A_MAD_U64_U32)
KERNEL64(k_lshl_add_u64, A_LSHL_ADD_U64)
KERNEL64(k_lshlrev_b64, A_LSHLREV_B64)
KERNELD(k_fma_f64, A_FMA_F64)
KERNELD(k_add_f64, A_ADD_F64)
KERNELD(k_mul_f64, A_MUL_F64)

// mixed stream: can DFMA and integer ops overlap? (1 DFMA : 1 mad_u64 : 2 add)
__global__ void __launch_bounds__(256) k_mix_fma64_int(unsigned* out, unsigned seed) {
    double d0 = 1.0 + threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
    unsigned u0 = threadIdx.x + seed, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    double b = 1.0000001, c = 1e-9;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %5\n"
            "v_fma_f64 %1, %1, %8, %9\n v_add_u32 %5, %5, %6\n"
            "v_fma_f64 %2, %2, %8, %9\n v_add_u32 %6, %6, %7\n"
            "v_fma_f64 %3, %3, %8, %9\n v_add_u32 %7, %7, %4\n"
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)
            : "v"(b), "v"(c));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)__double_as_longlong(d0 + d1 + d2 + d3) ^ u0 ^ u1 ^ u2 ^ u3;
}
__global__ void __launch_bounds__(256) k_mix_mad64_add(unsigned* out, unsigned seed) {
    unsigned long long d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
    unsigned u0 = threadIdx.x + seed, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    unsigned b = u0 | 1, c = seed + 99;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_addc_co_u32 %4, vcc, 0, %4, vcc\n"
            "v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n"
            "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_addc_co_u32 %6, vcc, 0, %6, vcc\n"
            "v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_addc_co_u32 %7, vcc, 0, %7, vcc\n"
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)
            : "v"(b), "v"(c) : "vcc");
    }
    unsigned long long r = d0 ^ d1 ^ d2 ^ d3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)r ^ (unsigned)(r >> 32) ^ u0 ^ u1 ^ u2 ^ u3;
}


// dependent chains: how many waves per SIMD hide the latency of back-to-back v_mad_u64_u32 on ONE
// accumulator (the single-chain column accumulation of csrc/imt_mont_asm.hpp)?
__global__ void __launch_bounds__(256) k_mad_dep1(unsigned* out, unsigned seed) {
    unsigned long long d0 = threadIdx.x;
    unsigned b = (threadIdx.x + seed) | 1, c = seed + 99;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
            "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
            "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
            "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
            : "+v"(d0) : "v"(b), "v"(c) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)d0 ^ (unsigned)(d0 >> 32);
}
__global__ void __launch_bounds__(256) k_mad_dep2(unsigned* out, unsigned seed) {
    unsigned long long d0 = threadIdx.x, d1 = d0 + 5;
    unsigned b = (threadIdx.x + seed) | 1, c = seed + 99;
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n"
            : "+v"(d0), "+v"(d1) : "v"(b), "v"(c) : "vcc");
    }
    unsigned long long r = d0 ^ d1;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)r ^ (unsigned)(r >> 32);
}
// one column step of the single-chain form: 6 mads, mul_lo, mad, shift -- all dependent
__global__ void __launch_bounds__(256) k_column_dep(unsigned* out, unsigned seed) {
    unsigned b = (threadIdx.x + seed) | 1, c = seed + 99, m = 0, lo = threadIdx.x;
    asm volatile("v_mov_b32 v60, %0\n v_mov_b32 v61, 0\n" : : "v"(lo) : "v60", "v61");
    for (int i = 0; i < ITER; i++) {
        asm volatile(
            "v_mad_u64_u32 v[60:61], vcc, %1, %2, v[60:61]\n v_mad_u64_u32 v[60:61], vcc, %1, %2, v[60:61]\n"
            "v_mad_u64_u32 v[60:61], vcc, %1, %2, v[60:61]\n v_mad_u64_u32 v[60:61], vcc, %1, %2, v[60:61]\n"
            "v_mad_u64_u32 v[60:61], vcc, %1, %2, v[60:61]\n"
            "v_mul_lo_u32 %0, v60, %2\n"
            "v_mad_u64_u32 v[60:61], vcc, %0, %1, v[60:61]\n"
            "v_lshrrev_b64 v[60:61], 29, v[60:61]\n"
            : "+v"(m) : "v"(b), "v"(c) : "vcc", "v60", "v61");
    }
    asm volatile("v_mov_b32 %0, v60\n" : "=v"(lo) : : "v60", "v61");
    out[blockIdx.x * blockDim.x + threadIdx.x] = lo ^ m;
}

typedef void (*kfn)(unsigned*, unsigned);
struct Case { const char* name; kfn fn; int inst_per_iter; };

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int CUS = prop.multiProcessorCount;
    std::vector<Case> cases = {
        {"v_fma_f32", k_fma_f32, 8}, {"v_add_u32", k_add_u32, 8}, {"v_add3_u32", k_add3_u32, 8},
        {"v_mul_lo_u32", k_mul_lo_u32, 8}, {"v_mul_hi_u32", k_mul_hi_u32, 8},
        {"v_mad_u32_u24", k_mad_u32_u24, 8}, {"v_mul_u32_u24", k_mul_u32_u24, 8},
        {"v_mul_hi_u32_u24", k_mul_hi_u24, 8}, {"v_add_co_u32", k_add_co, 8},
        {"v_addc_co_u32", k_addc_co, 8}, {"v_alignbit_b32", k_alignbit, 8},
        {"v_lshl_or_b32", k_lshl_or, 8}, {"v_and_or_b32", k_and_or, 8}, {"v_dot4_u32_u8", k_dot4_u8, 8},
        {"v_pk_fma_f32", k_pk_fma_f32, 8}, {"v_mad_u64_u32", k_mad_u64_u32, 8},
        {"v_lshl_add_u64", k_lshl_add_u64, 8}, {"v_lshlrev_b64", k_lshlrev_b64, 8},
        {"v_cndmask_b32", k_cndmask, 8}, {"v_cndmask_b32_e64 s[10:11]", k_cndmask_e64, 8},
        {"cndmask + add_u32 (pairs)", k_cndmask_add, 16}, {"cndmask + 3 mul_lo (quads)", k_cndmask_mad, 32}, {"v_bfi_b32", k_bfi_b32, 8}, {"v_and_b32", k_and_b32, 8}, {"v_or_b32", k_or_b32, 8}, {"v_xor_b32", k_xor_b32, 8},
        {"v_lshrrev_b32", k_lshrrev_b32, 8}, {"v_lshlrev_b32", k_lshlrev_b32, 8}, {"v_ashrrev_i32", k_ashrrev_i32, 8},
        {"v_sub_u32", k_sub_u32, 8}, {"v_mov_b32", k_mov_b32, 8},
        {"v_lshrrev_b64", k_lshrrev_b64, 8}, {"v_ashrrev_i64", k_ashrrev_i64, 8}, {"v_mad_i64_i32", k_mad_i64_i32, 8},
        {"v_fma_f64", k_fma_f64, 8}, {"v_add_f64", k_add_f64, 8}, {"v_mul_f64", k_mul_f64, 8},
        {"mix 4x(fma_f64+add_u32)", k_mix_fma64_int, 8}, {"mix 4x(mad_u64+addc)", k_mix_mad64_add, 8},
        {"mad_u64 1 dependent chain", k_mad_dep1, 8}, {"mad_u64 2 dependent chains", k_mad_dep2, 8},
        {"column step, dependent", k_column_dep, 8},
    };
    unsigned* out;
    const int wavesPerSimdList[] = {1, 2, 4, 8};
    CHECK(hipMalloc(&out, sizeof(unsigned) * CUS * 8 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-28s", "instruction");
    for (int w : wavesPerSimdList) printf("  w/SIMD=%d: Ginst/s cyc/winst", w);
    printf("\n");
    for (auto& cs : cases) {
        printf("%-28s", cs.name);
        for (int w : wavesPerSimdList) {
            int blocks = CUS * w;  // 256 threads = 4 waves = 1 wave per SIMD per block
            cs.fn<<<blocks, 256>>>(out, 1);  // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) cs.fn<<<blocks, 256>>>(out, r);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            double s = ms * 1e-3 / 5;
            double winst = (double)blocks * 4 * ITER * cs.inst_per_iter;   // wave-instructions
            double lane_inst = winst * 64;
            // cycles per wave-instruction per SIMD at 2.4 GHz
            double cyc = s * 2.4e9 / (winst / (CUS * 4.0));
            printf("  %10.1f %8.2f      ", lane_inst / s * 1e-9, cyc);
        }
        printf("\n");
    }
    return 0;
}
