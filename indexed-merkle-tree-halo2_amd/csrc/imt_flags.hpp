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
void copy16(hipStream_t s, void* dst, const void* src, size_t bytes);

}  // namespace launch
}  // namespace imt
