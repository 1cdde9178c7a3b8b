// rccl_streams_probe.hip -- how many streams does an RCCL communicator create for itself, and on which hardware queues?
//
// An RCCL communicator owns two internal streams (`deviceStream`, `hostStream`: the strings are in librccl.so).  In NCCL's
// eager launch path every collective is bracketed by them: the user's stream waits for an event of `deviceStream` before
// the kernel, and `deviceStream` waits for an event recorded on the user's stream behind the kernel.  That second wait is
// a barrier packet on whatever hardware queue `deviceStream` shares -- and it stays there until the collective has
// completed, i.e. until every rank has reached it.  If that queue is also a round stream's, the round's next hash kernels
// stand behind every collective of that communicator ("Hardware queues").  With one rank RCCL short-cuts a
// collective to a copy, so the waits themselves cannot be watched on a one-GPU box; what CAN be measured is how many
// streams ncclCommInitRank creates in the normal-priority pool and where they land: the runtime gives a new stream the
// queue with the fewest streams, so the queues that the NEXT streams avoid are the ones RCCL's streams took.
//
// Build: hipcc --offload-arch=gfx950 -O2 -o rccl_streams_probe rccl_streams_probe.hip -lrccl     Run: ./rccl_streams_probe [comms]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
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
#define NCHECK(x)                                                                      \
    do {                                                                               \
        ncclResult_t r_ = (x);                                                         \
        if (r_ != ncclSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_));                   \
            exit(3);                                                                   \
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

static std::vector<int> classes(const std::vector<hipStream_t>& st, uint64_t* d, uint64_t* h) {
    const int n = (int)st.size();
    std::vector<std::vector<int>> share(n, std::vector<int>(n, 0));
    for (int i = 0; i < n; i++) {
        CHECK(hipMemset(d, 0, sizeof(uint64_t) * (n + 1)));
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], (uint64_t)30000, d + n);
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
    return cls;
}

int main(int argc, char** argv) {
    const int n_comms = argc > 1 ? atoi(argv[1]) : 2;
    CHECK(hipSetDevice(0));
    uint64_t *d, *h;
    CHECK(hipMalloc((void**)&d, sizeof(uint64_t) * 128));
    h = (uint64_t*)malloc(sizeof(uint64_t) * 128);
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    const int nq = q ? atoi(q) : 4;
    printf("GPU_MAX_HW_QUEUES=%s\n", q ? q : "(unset: 4)");
    std::vector<hipStream_t> st;
    std::vector<std::string> tag;
    auto add = [&](int prio, const std::string& t) {
        hipStream_t s;
        CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
        st.push_back(s);
        tag.push_back(t);
    };
    auto show = [&](const char* what) {
        std::vector<int> c = classes(st, d, h);
        printf("%s\n  ", what);
        for (size_t i = 0; i < st.size(); i++) printf("%s:%d ", tag[i].c_str(), c[i]);
        printf("\n");
        fflush(stdout);
    };
    // one marker stream per hardware queue of the normal pool (a fresh process hands them out in a fixed order)
    for (int i = 0; i < nq; i++) add(0, "m" + std::to_string(i));
    show("markers (one per queue of the normal-priority pool)");
    int least = 0, greatest = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    std::vector<ncclComm_t> comms;
    for (int c = 0; c < n_comms; c++) {
        ncclUniqueId id;
        NCHECK(ncclGetUniqueId(&id));
        ncclComm_t comm;
        NCHECK(ncclCommInitRank(&comm, 1, id, 0));
        comms.push_back(comm);
        // where do the next nq streams of each pool land?  the queues they reach LAST are the ones the communicator's streams took
        for (int i = 0; i < nq; i++) add(0, "c" + std::to_string(c) + "n" + std::to_string(i));
        char what[160];
        snprintf(what, sizeof what, "after communicator %d: %d more normal-priority streams, in creation order", c, nq);
        show(what);
    }
    // the other pools: four high- and four low-priority streams (a pool RCCL had used would show a shifted order)
    for (int i = 0; i < 4; i++) add(greatest, "h" + std::to_string(i));
    for (int i = 0; i < 4; i++) add(least, "l" + std::to_string(i));
    show("+ 4 high- and 4 low-priority streams");
    for (auto c : comms) ncclCommDestroy(c);
    for (auto s : st) hipStreamDestroy(s);
    return 0;
}
