// barrier_wakeup.hip -- how long after an event completes does a stream that waits for it (on ANOTHER hardware queue) go on?
//
// In the rocprofv3 traces of one rank of eight (profiles/r05_emu8_pools_trace.csv.gz) a round's apply kernel starts 60 - 700
// us after the kernel in front of it on its stream has ended, although everything it waits for (events of other streams)
// completed long before.  A cross-queue hipStreamWaitEvent is a barrier packet; the question here is what a barrier
// packet costs once its signal HAS completed -- alone (the 3.6 us the queue model uses) and while other queues keep the
// device and the command processor busy with back-to-back kernels of 2 048 waves.
//
//   stream X:  k_spin(200 us) -> record E          stream Y:  wait E -> k_stamp
//   latency = stamp - end of the spin (GPU wall clock, 100 MHz)
// Cases: idle device; two / three other streams running long kernels back to back (k_busy: 512 blocks x 256 lanes of
// integer multiply-adds, ~0.9 ms each); Y at normal / high / low priority; a chain of 1 / 3 / 6 such waits in a row (each on
// an event of a different stream), as a round's stream carries per tick.
//
// Build: hipcc --offload-arch=gfx950 -O2 -o barrier_wakeup barrier_wakeup.hip      Run: ./barrier_wakeup
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

__global__ void k_spin(uint64_t ticks, uint64_t* end_stamp) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const uint64_t t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
        *end_stamp = wall_clock64();
    }
}
__global__ void k_stamp(uint64_t* stamp) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *stamp = wall_clock64();
}
// a stand-in for a hash kernel: every lane a dependent chain of 64-bit multiply-adds
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 5))) k_busy(uint32_t iters, uint64_t* sink) {   // at most 5 waves per SIMD, like k_sweep
    uint64_t a = threadIdx.x + 1, b = blockIdx.x + 3;
    for (uint32_t i = 0; i < iters; i++) {
        a = a * 0x9E3779B97F4A7C15ull + b;
        b = b * (uint32_t)a + (a >> 7);
    }
    if (a == 0x1234567 && b == 42) *sink = a;      // never: keeps the loop
}

struct Ctx {
    uint64_t *d, *h;
    std::vector<hipStream_t> load;      // the busy streams
    uint32_t iters = 0;
};

static hipStream_t mk(int prio) {
    hipStream_t s;
    CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
    return s;
}

// median / max latency (us) of `reps` measurements; chain = number of waits in front of the stamp (each on its own event)
static void measure(Ctx& c, const char* label, int y_prio, int x_prio, int n_load, int chain, int reps = 15) {
    hipStream_t y = mk(y_prio);
    std::vector<hipStream_t> xs;
    std::vector<hipEvent_t> ev(chain);
    for (int k = 0; k < chain; k++) {
        xs.push_back(mk(x_prio));
        CHECK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    }
    std::vector<double> lat;
    for (int r = 0; r < reps; r++) {
        CHECK(hipMemset(c.d, 0, 64 * sizeof(uint64_t)));
        CHECK(hipDeviceSynchronize());
        // the load: long kernels back to back on n_load streams (enough to cover the measurement)
        for (int l = 0; l < n_load; l++)
            for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_busy, dim3(512), dim3(256), 0, c.load[l], c.iters, c.d + 40);
        // X_k: spin, record; the LAST spin is the longest, so the stamp's wake-up is counted from ITS end
        for (int k = 0; k < chain; k++) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, xs[k], (uint64_t)(20000 + 2000 * k), c.d + k);
            CHECK(hipEventRecord(ev[k], xs[k]));
        }
        for (int k = 0; k < chain; k++) CHECK(hipStreamWaitEvent(y, ev[k], 0));
        hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, y, c.d + 32);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(c.h, c.d, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost));
        uint64_t last_end = 0;
        for (int k = 0; k < chain; k++) last_end = std::max(last_end, c.h[k]);
        lat.push_back(((double)c.h[32] - (double)last_end) / 100.0);
    }
    std::sort(lat.begin(), lat.end());
    printf("  %-74s median %7.1f us   min %7.1f   max %7.1f\n", label, lat[lat.size() / 2], lat.front(), lat.back());
    fflush(stdout);
    CHECK(hipStreamDestroy(y));
    for (auto s : xs) CHECK(hipStreamDestroy(s));
    for (auto e : ev) CHECK(hipEventDestroy(e));
}

int main() {
    CHECK(hipSetDevice(0));
    Ctx c;
    CHECK(hipMalloc((void**)&c.d, 64 * sizeof(uint64_t)));
    c.h = (uint64_t*)malloc(64 * sizeof(uint64_t));
    int least = 0, greatest = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (int l = 0; l < 3; l++) c.load.push_back(mk(greatest));
    // calibrate k_busy to ~0.9 ms alone
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    c.iters = 20000;
    for (int it = 0; it < 3; it++) {
        CHECK(hipEventRecord(e0, c.load[0]));
        hipLaunchKernelGGL(k_busy, dim3(512), dim3(256), 0, c.load[0], c.iters, c.d + 40);
        CHECK(hipEventRecord(e1, c.load[0]));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        c.iters = (uint32_t)(c.iters * 0.9 / ms);
    }
    printf("k_busy: %u iterations per lane for ~0.9 ms alone (512 blocks x 256 lanes = 2 048 waves); priorities: high %d, low %d\n", c.iters, greatest, least);
    printf("latency from the end of the awaited kernel to the start of the waiter's next kernel (another hardware queue)\n");
    for (int n_load = 0; n_load <= 3; n_load++) {
        printf(" %d other stream(s) running 2 048-wave kernels back to back (HIGH priority)\n", n_load);
        measure(c, "one wait; waiter normal priority, awaited stream normal", 0, 0, n_load, 1);
        measure(c, "one wait; waiter HIGH priority, awaited stream LOW (a round waiting for its gather)", greatest, least, n_load, 1);
        measure(c, "one wait; waiter LOW priority, awaited stream HIGH (a gather waiting for its pack)", least, greatest, n_load, 1);
        measure(c, "three waits in a row; waiter HIGH, awaited streams LOW", greatest, least, n_load, 3);
        measure(c, "six waits in a row; waiter HIGH, awaited streams normal", greatest, 0, n_load, 6);
    }
    return 0;
}
