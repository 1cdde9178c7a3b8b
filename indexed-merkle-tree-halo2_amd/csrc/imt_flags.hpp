// imt_flags.hpp -- launchers of imt_flags.hip: cross-process, stream-ordered flags in shared host memory.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>

namespace imt {
namespace launch {

constexpr int FLAG_WAIT_MAX = 16;
struct FlagWait {
    const uint64_t* flag[FLAG_WAIT_MAX];   // device-visible addresses of the counters
    int n;
    uint64_t value;                        // wait until every counter >= value
    uint64_t timeout_ticks;                // wall_clock64 ticks (100 MHz) before giving up
    uint32_t* err;                         // device-visible word: bit i set = flag i did not arrive
};
void flag_set(hipStream_t s, uint64_t* flag, uint64_t value);
void flag_wait(hipStream_t s, const FlagWait& w);
// bytes: a multiple of 16; both pointers 16-byte aligned device-visible memory (a peer's IPC-mapped buffer included)
// poison (may be null): a device-visible word; the copy / the store is skipped when it is non-zero (the sticky error word
// of a transport whose wait has given up -- imt_flags.hip)
void copy16(hipStream_t s, void* dst, const void* src, size_t bytes, const uint32_t* poison = nullptr);
// the payloads of SEVERAL peers in one launch (blockIdx.y = peer): every peer is read over its own xGMI link, so the
// links work side by side instead of one after the other (seven copies of 4.7 MB at N = 8: one link time, not seven)
struct CopyJobs {
    void* dst[FLAG_WAIT_MAX];
    const void* src[FLAG_WAIT_MAX];
    int n;
};
void copy16_multi(hipStream_t s, const CopyJobs& jobs, size_t bytes, const uint32_t* poison);
void flag_set_checked(hipStream_t s, uint64_t* flag, uint64_t value, const uint32_t* poison);
// the hardware-queue probe: hold stream s for `ticks` of the 100 MHz wall clock and write the end time / write the time
// the kernel ran (one wave each)
void spin(hipStream_t s, uint64_t ticks, uint64_t* end_stamp);
void stamp(hipStream_t s, uint64_t* out);

}  // namespace launch
}  // namespace imt
