// queue_map_probe.hip -- which HIP streams share a hardware queue, measured on the device.
//
// The HIP runtime multiplexes streams onto a few in-order hardware (HSA) queues; a stream that shares a queue with a
// busy one stands behind that stream's backlog (tools/queue_sharing_probe.py).  The probe: a one-wave kernel
// spins for 400 us on stream i and writes the time it ended; every other stream j gets a one-lane kernel that writes the
// time it RAN.  stamp(j) >= end(i)  <=>  j's kernel could not start before i's finished  <=>  j and i share a queue.
// All times are the GPU's wall clock (100 MHz), no host timing involved.  This is the prototype of the placement check
// inside imt_sliced_create (csrc/imt_sliced.cpp) and the source of the stream -> queue rule the CPU model of the sliced
// schedule assumes (tests/hwq_model.py).
//
// Build: hipcc --offload-arch=gfx950 -O2 -o queue_map_probe queue_map_probe.hip      Run: ./queue_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
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

// class label per stream: streams with the same label share a hardware queue
static std::vector<int> classes(const std::vector<hipStream_t>& st, uint64_t* d, uint64_t* h) {
    const int n = (int)st.size();
    std::vector<std::vector<int>> share(n, std::vector<int>(n, 0));
    for (int i = 0; i < n; i++) {
        CHECK(hipMemset(d, 0, sizeof(uint64_t) * (n + 1)));
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], (uint64_t)40000, d + n);
        for (int j = 0; j < n; j++)
            if (j != i) hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, st[j], d + j);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h, d, sizeof(uint64_t) * (n + 1), hipMemcpyDeviceToHost));
        for (int j = 0; j < n; j++)
            if (j != i) share[i][j] = h[j] >= h[n];
    }
    std::vector<int> cls(n, -1);
    int next = 0;
    for (int i = 0; i < n; i++) {
        if (cls[i] >= 0) continue;
        cls[i] = next++;
        for (int j = i + 1; j < n; j++)
            if (share[i][j] && share[j][i]) cls[j] = cls[i];
    }
    // consistency: sharing must be an equivalence
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            if (i != j && (share[i][j] != 0) != (cls[i] == cls[j])) printf("  (inconsistent pair %d %d: share %d / %d)\n", i, j, share[i][j], share[j][i]);
    return cls;
}

static void show(const char* what, const std::vector<hipStream_t>& st, const std::vector<std::string>& tag, uint64_t* d, uint64_t* h) {
    std::vector<int> c = classes(st, d, h);
    printf("%s\n  ", what);
    for (size_t i = 0; i < st.size(); i++) printf("%s:%d ", tag[i].c_str(), c[i]);
    printf("\n");
    fflush(stdout);
}

int main() {
    CHECK(hipSetDevice(0));
    uint64_t *d, *h;
    CHECK(hipMalloc((void**)&d, sizeof(uint64_t) * 64));
    h = (uint64_t*)malloc(sizeof(uint64_t) * 64);
    int least = 0, greatest = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    printf("stream priority range: least %d greatest %d; GPU_MAX_HW_QUEUES=%s\n", least, greatest, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)");

    std::vector<hipStream_t> st;
    std::vector<std::string> tag;
    auto add = [&](int prio, const char* t) {
        hipStream_t s;
        CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
        st.push_back(s);
        tag.push_back(std::string(t) + std::to_string(st.size() - 1));
    };
    // A: the null stream + twelve normal-priority streams created one after the other
    st.push_back(nullptr);
    tag.push_back("null");
    for (int i = 0; i < 12; i++) add(0, "n");
    show("A: null stream + 12 normal-priority streams in creation order (label = hardware queue class)", st, tag, d, h);
    // B: four high-priority streams added
    for (int i = 0; i < 4; i++) add(greatest, "h");
    show("B: + 4 high-priority streams", st, tag, d, h);
    // C: four low-priority
    for (int i = 0; i < 4; i++) add(least, "l");
    show("C: + 4 low-priority streams", st, tag, d, h);
    // D: destroy normal streams n2, n3 (positions 2, 3), create two new ones: where do they land?
    CHECK(hipStreamDestroy(st[2]));
    CHECK(hipStreamDestroy(st[3]));
    st.erase(st.begin() + 2, st.begin() + 4);
    tag.erase(tag.begin() + 2, tag.begin() + 4);
    add(0, "N");
    add(0, "N");
    add(0, "N");
    show("D: n2, n3 destroyed, three new normal streams N created", st, tag, d, h);
    // E: the same probe again (is the map stable over time?)
    show("E: the same streams probed again", st, tag, d, h);
    for (auto s : st)
        if (s) hipStreamDestroy(s);
    // F: a fresh set after everything was destroyed
    st.clear();
    tag.clear();
    for (int i = 0; i < 9; i++) add(0, "f");
    show("F: all destroyed, 9 fresh normal streams", st, tag, d, h);
    return 0;
}
