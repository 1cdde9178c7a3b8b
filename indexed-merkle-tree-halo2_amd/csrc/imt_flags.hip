// imt_flags.hip -- stream-ordered flags between PROCESSES (the IPC transport of imt_sliced_*, imt_sliced.cpp).
//
// A flag is a 64-bit counter in host memory both processes have mapped and registered with HIP (a POSIX shared-memory
// page: uncached for the GPU, so coherent without any cache-scope reasoning).  k_flag_set publishes "everything before
// me on this stream is done, for the k-th time"; k_flag_wait holds its stream until every flag of a set has reached k.
// Counters only grow, so a wait never depends on the order in which the two hosts issue their work.  Every wave of
// k_flag_wait reaches an exit: a flag that does not arrive within the time limit sets an error bit and the kernel
// returns (the host reports it: a peer died or hangs), so the grid always drains.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "imt_flags.hpp"

namespace {

__global__ void k_flag_set(uint64_t* flag, uint64_t value) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The error word is STICKY and doubles as a poison: once a wait has given up, the payload it waited for is not there, so
// everything that would consume or acknowledge it -- the copies out of the peers' buffers, the `copied` flag that lets
// the peers overwrite theirs, the apply of the gathered payloads (k_apply_gathered reads the same word) -- must not run.
// Every later wait returns at once.  The host sees the word at its next call (imt_sliced_step / _wait / _flush) and the
// world refuses to go on.
__device__ inline bool poisoned(const uint32_t* poison) {
    return poison && __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
}

__global__ void k_flag_set_checked(uint64_t* flag, uint64_t value, const uint32_t* poison) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && !poisoned(poison)) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_flag_wait(imt::launch::FlagWait w) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (poisoned(w.err)) return;
    const uint64_t t0 = wall_clock64();
    for (int i = 0; i < w.n; i++) {
        while (__hip_atomic_load(w.flag[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < w.value) {
            if (wall_clock64() - t0 > w.timeout_ticks) {      // the exit every wave reaches
                __hip_atomic_fetch_or(w.err, 1u << (i & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
            __builtin_amdgcn_s_sleep(64);
        }
    }
}

// payload copy out of a peer's memory.  hipMemcpyAsync from an IPC-mapped pointer blocks the host until the stream has
// drained (measured: 1.1 ms per call behind a waiting kernel, profiles/r04 notes), so the copy is a kernel of its own:
// 16 bytes per lane and step, grid-stride, sizes are multiples of 16.
__global__ void k_copy16(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16, const uint32_t* poison) {
    if (poisoned(poison)) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ void k_copy16_multi(imt::launch::CopyJobs jobs, size_t n16, const uint32_t* poison) {
    if (poisoned(poison)) return;
    uint4* __restrict__ dst = (uint4*)jobs.dst[blockIdx.y];
    const uint4* __restrict__ src = (const uint4*)jobs.src[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ---- which streams share a hardware queue (imt_sliced.cpp: QueueProbe) ----
// k_spin holds its stream's hardware queue for `ticks` of the 100 MHz wall clock and writes when it ended; k_stamp
// writes when it RAN.  A stamp taken on another stream that is not earlier than the spin's end could not start before
// the spin had finished: the two streams share an in-order hardware queue.  One wave each, bounded, GPU clock only.
__global__ void k_spin(uint64_t ticks, uint64_t* end_stamp) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    *end_stamp = wall_clock64();
}
__global__ void k_stamp(uint64_t* stamp) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *stamp = wall_clock64();
}

}  // namespace

namespace imt {
namespace launch {

void copy16(hipStream_t s, void* dst, const void* src, size_t bytes, const uint32_t* poison) {
    const size_t n16 = bytes / 16;
    if (!n16) return;
    const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, 512);
    hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, s, (uint4*)dst, (const uint4*)src, n16, poison);
}

void copy16_multi(hipStream_t s, const CopyJobs& jobs, size_t bytes, const uint32_t* poison) {
    const size_t n16 = bytes / 16;
    if (!n16 || jobs.n <= 0) return;
    // enough loads in flight per peer to keep a link busy (64 blocks x 256 lanes x 16 B = 256 KB), no more: the copy shares
    // the device with the hash kernels
    const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, 64);
    hipLaunchKernelGGL(k_copy16_multi, dim3(blocks, (unsigned)jobs.n), dim3(256), 0, s, jobs, n16, poison);
}

void flag_set_checked(hipStream_t s, uint64_t* flag, uint64_t value, const uint32_t* poison) {
    hipLaunchKernelGGL(k_flag_set_checked, dim3(1), dim3(64), 0, s, flag, value, poison);
}

void spin(hipStream_t s, uint64_t ticks, uint64_t* end_stamp) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ticks, end_stamp); }
void stamp(hipStream_t s, uint64_t* out) { hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, out); }

void flag_set(hipStream_t s, uint64_t* flag, uint64_t value) { hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(64), 0, s, flag, value); }

void flag_wait(hipStream_t s, const FlagWait& w) {
    if (w.n > 0) hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, s, w);
}

}  // namespace launch
}  // namespace imt
