// trace_store_pattern.hip -- the memory side of k_hash_trace alone: every thread writes `rows` 32-byte rows, row r of
// item i at (r * n + i) * 32 (row-major) or (i * rows + r) * 32 (item-major), with `work` dependent v_mad_u64_u32 between
// two rows (0 = stores only).  Tells how much of the trace kernel's time the store pattern itself can explain.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/trace_store_pattern.hip -o tools/microbench/trace_store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define OK(x)                                                    \
    do {                                                         \
        if ((x) != hipSuccess) {                                 \
            std::fprintf(stderr, "%s failed\n", #x);            \
            std::exit(1);                                        \
        }                                                        \
    } while (0)

typedef unsigned W4 __attribute__((ext_vector_type(4)));

// XCD: workgroups go to the 8 XCDs round-robin; with the remap XCD x works on a contiguous eighth of the items
template <int WORK, bool NT, bool XCD = false>
__global__ void __launch_bounds__(256) k_pattern(W4* out, unsigned n, unsigned rows, int item_major, unsigned seed) {
    const unsigned blk = XCD ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned i = blk * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long acc = seed + i;
    unsigned x = i * 2654435761u + 1;
    for (unsigned r = 0; r < rows; r++) {
#pragma unroll
        for (int k = 0; k < WORK; k++) acc = (unsigned long long)x * (unsigned)acc + (acc >> 29);
        const size_t row = item_major ? (size_t)i * rows + r : (size_t)r * n + i;
        const unsigned lo = (unsigned)acc, hi = (unsigned)(acc >> 32);
        const W4 v0 = {lo, hi, r, i}, v1 = {hi, lo, i, r};
        if (NT) {
            __builtin_nontemporal_store(v0, &out[2 * row]);
            __builtin_nontemporal_store(v1, &out[2 * row + 1]);
        } else {
            out[2 * row] = v0;
            out[2 * row + 1] = v1;
        }
    }
}

// item-major with the four rows of a 128-byte line written together (what staging rows in LDS / registers would do)
template <int WORK>
__global__ void __launch_bounds__(256) k_pattern_lines(W4* out, unsigned n, unsigned rows, unsigned seed) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long acc = seed + i;
    unsigned x = i * 2654435761u + 1;
    for (unsigned r = 0; r < rows; r += 4) {
        W4 v[8];
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int k = 0; k < WORK; k++) acc = (unsigned long long)x * (unsigned)acc + (acc >> 29);
            const unsigned lo = (unsigned)acc, hi = (unsigned)(acc >> 32);
            v[2 * q] = W4{lo, hi, r + q, i};
            v[2 * q + 1] = W4{hi, lo, i, r + q};
        }
        W4* dst = out + 2 * ((size_t)i * rows + r);
#pragma unroll
        for (int q = 0; q < 8; q++) dst[q] = v[q];
    }
}
template <int WORK>
static void run_lines(W4* d, unsigned n, unsigned rows) {
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    k_pattern_lines<WORK><<<(n + 255) / 256, 256>>>(d, n, rows, 1);
    OK(hipDeviceSynchronize());
    OK(hipEventRecord(e0));
    for (int it = 0; it < 3; it++) k_pattern_lines<WORK><<<(n + 255) / 256, 256>>>(d, n, rows, it);
    OK(hipEventRecord(e1));
    OK(hipEventSynchronize(e1));
    float ms;
    OK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    std::printf("n=2^%d rows=%u item-major 128-B lines work=%4d mads/row  %8.3f ms  %7.1f GB/s  %6.1f Mitems/s\n", __builtin_ctz(n),
                rows, WORK, ms, (double)n * rows * 32 / ms / 1e6, n / ms / 1e3);
}

template <int WORK, bool NT, bool XCD = false>
static void run(W4* d, unsigned n, unsigned rows, int item_major) {
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    k_pattern<WORK, NT, XCD><<<(n + 255) / 256, 256>>>(d, n, rows, item_major, 1);
    OK(hipDeviceSynchronize());
    OK(hipEventRecord(e0));
    for (int it = 0; it < 3; it++) k_pattern<WORK, NT, XCD><<<(n + 255) / 256, 256>>>(d, n, rows, item_major, it);
    OK(hipEventRecord(e1));
    OK(hipEventSynchronize(e1));
    float ms;
    OK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double bytes = (double)n * rows * 32;
    std::printf("n=2^%d rows=%u %-10s %s work=%4d mads/row  %8.3f ms  %7.1f GB/s  %6.1f Mitems/s\n", __builtin_ctz(n), rows,
                item_major ? "item-major" : "row-major", NT ? "nontemporal" : XCD ? "xcd-remap  " : "plain      ", WORK, ms, bytes / ms / 1e6, n / ms / 1e3);
}

int main(int argc, char** argv) {
    const unsigned logn = argc > 1 ? atoi(argv[1]) : 18, rows = 1208;
    const unsigned n = 1u << logn;
    W4* d;
    if (hipMalloc(&d, (size_t)n * rows * 32) != hipSuccess) return 1;
    for (int im = 0; im < 2; im++) {
        run<0, false>(d, n, rows, im);
        run<0, true>(d, n, rows, im);
        run<0, false, true>(d, n, rows, im);
        run<64, false>(d, n, rows, im);
        run<128, false, true>(d, n, rows, im);
        run<128, false>(d, n, rows, im);
        run<128, true>(d, n, rows, im);
        run<256, false>(d, n, rows, im);
        run<256, true>(d, n, rows, im);
    }
    run_lines<0>(d, n, rows);
    run_lines<128>(d, n, rows);
    run_lines<256>(d, n, rows);
    OK(hipFree(d));
    return 0;
}
