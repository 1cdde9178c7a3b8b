// emu_gather.hip -- a collective that LOOKS like RCCL's to the GPU, for tools/rank_emulation.py.
//
// The rank emulation so far modelled an all-gather as a one-wave sleep + world blit copies.  A real ncclAllGather is ONE
// kernel of several workgroups (a channel each, 256-512 lanes) that has to get wave slots on the device -- beside two
// resident 2 048-wave hash kernels, from a LOW-priority queue -- and then sits there polling flags until its peers' data
// has come over the links.  This kernel has that shape: `wgs` workgroups of 512 lanes; workgroup 0's first lane plays
// the peers (it raises a flag in memory once  wait_ticks  of the 100 MHz wall clock have passed since IT started: the
// modelled  latency + bytes / link rate ), every other workgroup's first lane polls the flag (sleeping between polls,
// the other lanes at the barrier, as RCCL's do), then all of them copy: slot k of recv = this rank's own payload (what
// the emulation always did: memory-safe for the apply kernel, costs what a real slot costs).
//
// Exit condition: every polling lane leaves after EMU_POLL_CAP_TICKS (1 s) whatever the flag says; workgroup 0 is
// dispatched first, so the flag is raised unless the device is wedged anyway.
//
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/microbench/libemu_gather.so tools/microbench/emu_gather.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

struct EmuSlot {
    unsigned int flag, done;
    unsigned long long t_first, t_last;      // wall clock when workgroup 0 started / when the LAST workgroup started
};
struct EmuStat {
    unsigned int spread_ticks, wgs;          // t_last - t_first of one collective: how long until all its workgroups had wave slots
};
constexpr int EMU_SLOTS = 4096;
constexpr unsigned long long EMU_POLL_CAP_TICKS = 100000000ull;      // 1 s of the 100 MHz wall clock

__global__ __launch_bounds__(512) void k_emu_gather(uint4* __restrict__ recv, const uint4* __restrict__ send, size_t n16, int world,
                                                    unsigned long long wait_ticks, EmuSlot* slot, EmuStat* stat) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        if (blockIdx.x == 0) __hip_atomic_store(&slot->t_first, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicMax(&slot->t_last, t0);
        if (blockIdx.x == 0) {
            while (wall_clock64() - t0 < wait_ticks) __builtin_amdgcn_s_sleep(16);
            __hip_atomic_store(&slot->flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(&slot->flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0 && wall_clock64() - t0 < EMU_POLL_CAP_TICKS)
                __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = send[i];
        for (int k = 0; k < world; k++) recv[(size_t)k * n16 + i] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {          // the last workgroup out gives the slot back
        if (__hip_atomic_fetch_add(&slot->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            const unsigned long long a = __hip_atomic_load(&slot->t_first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long b = __hip_atomic_load(&slot->t_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stat->spread_ticks = b > a ? (unsigned int)(b - a) : 0u;
            stat->wgs = gridDim.x;
            __hip_atomic_store(&slot->t_last, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&slot->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&slot->flag, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static EmuSlot* g_slots = nullptr;
static EmuStat* g_stats = nullptr;
static unsigned g_next = 0;
constexpr unsigned EMU_STATS = 1u << 16;

extern "C" int emu_gather_init(void) {
    if (g_slots) return 0;
    if (hipMalloc((void**)&g_slots, sizeof(EmuSlot) * EMU_SLOTS) != hipSuccess) return -1;
    if (hipMalloc((void**)&g_stats, sizeof(EmuStat) * EMU_STATS) != hipSuccess) return -1;
    if (hipMemset(g_stats, 0, sizeof(EmuStat) * EMU_STATS) != hipSuccess) return -1;
    return hipMemset(g_slots, 0, sizeof(EmuSlot) * EMU_SLOTS) == hipSuccess ? 0 : -1;
}

// recv[k * nbytes, (k + 1) * nbytes) = send[0, nbytes) for k < world once wait_us have passed; nbytes a multiple of 16
extern "C" int emu_gather(void* stream, void* recv, const void* send, size_t nbytes, int world, double wait_us, int wgs) {
    if (!g_slots || (nbytes & 15) || wgs < 1 || wgs > 256) return -1;
    const unsigned call = g_next++;
    EmuSlot* slot = g_slots + (call % EMU_SLOTS);
    hipLaunchKernelGGL(k_emu_gather, dim3(wgs), dim3(512), 0, (hipStream_t)stream, (uint4*)recv, (const uint4*)send, nbytes / 16, world,
                       (unsigned long long)(wait_us * 100.0), slot, g_stats + (call % EMU_STATS));
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// after a device sync: out[i] = microseconds between the start of workgroup 0 and the start of the LAST workgroup of call
// first + i (how long the collective's kernel waited for all its wave slots); returns the number of calls made so far
extern "C" unsigned emu_gather_spreads(unsigned first, unsigned n, float* out) {
    if (!g_stats || n > EMU_STATS) return g_next;
    EmuStat* h = new EmuStat[EMU_STATS];
    if (hipMemcpy(h, g_stats, sizeof(EmuStat) * EMU_STATS, hipMemcpyDeviceToHost) == hipSuccess)
        for (unsigned i = 0; i < n; i++) out[i] = h[(first + i) % EMU_STATS].spread_ticks / 100.0f;
    delete[] h;
    return g_next;
}
