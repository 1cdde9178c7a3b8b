// ext_launch_event.hip -- can the completion of a KERNEL itself be the event another stream waits for (hipExtLaunchKernelGGL's
// stopEvent), instead of a hipEventRecord behind it (a marker packet of its own on the hardware queue)?  And what does a
// kernel - record - kernel sequence cost against kernel - kernel with the event attached to the first?
//
// Build: hipcc --offload-arch=gfx950 -O2 -o ext_launch_event ext_launch_event.hip      Run: ./ext_launch_event
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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

int main() {
    CHECK(hipSetDevice(0));
    uint64_t *d, h[16];
    CHECK(hipMalloc((void**)&d, sizeof h));
    hipStream_t x, y;
    CHECK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&y, hipStreamNonBlocking));
    for (int timing = 0; timing < 2; timing++) {
        hipEvent_t e;
        CHECK(hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming));
        for (int mode = 0; mode < 2; mode++) {       // 0: kernel, then hipEventRecord; 1: the event attached to the kernel
            std::vector<double> lat, chain;
            for (int r = 0; r < 12; r++) {
                CHECK(hipMemset(d, 0, sizeof h));
                CHECK(hipDeviceSynchronize());
                if (mode == 0) {
                    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, x, (uint64_t)20000, d + 0);
                    CHECK(hipEventRecord(e, x));
                } else {
                    hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, x, nullptr, e, 0, (uint64_t)20000, d + 0);
                    CHECK(hipGetLastError());
                }
                hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, x, d + 1);          // the next kernel of the SAME stream
                CHECK(hipStreamWaitEvent(y, e, 0));
                hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, y, d + 2);          // the waiter on another stream
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
                chain.push_back(((double)h[1] - (double)h[0]) / 100.0);
                lat.push_back(((double)h[2] - (double)h[0]) / 100.0);
            }
            std::sort(lat.begin(), lat.end());
            std::sort(chain.begin(), chain.end());
            printf("%s event, %s: next kernel of the same stream %6.1f us after the spin's end (min %6.1f); waiter on another stream %6.1f us (min %6.1f)%s\n",
                   timing ? "timing " : "no-timing", mode ? "attached to the kernel (hipExtLaunchKernelGGL stopEvent)" : "hipEventRecord behind the kernel        ",
                   chain[chain.size() / 2], chain.front(), lat[lat.size() / 2], lat.front(), lat.front() < 0 ? "   <-- the waiter ran BEFORE the kernel ended: not a dependency" : "");
        }
        CHECK(hipEventDestroy(e));
    }
    return 0;
}
