// imt_ctx.hpp -- internal definition of the C-ABI handles (imt_capi.cpp, imt_itree.cpp).
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>
#include "../../include/imt.h"
#include "imt_launch.hpp"
#include "imt_params.hpp"

#define IMT_MAX_DEPTH 64

struct imt_ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    imt::HostPoseidon hp;
    int* d_err = nullptr;             // device word: kernels OR 1 into it on a non-canonical input
    uint8_t* d_zero = nullptr;        // Z[0..IMT_MAX_DEPTH] in device format
    std::string last_error;
    uint32_t coop_max_events = 16384;   // IMT_OPT_COOP_MAX_EVENTS: launches up to this size use the quad-per-hash kernel
    struct Scratch { void* p = nullptr; size_t cap = 0; };
    std::vector<Scratch> scratch;     // grow-only staging buffers, indexed by slot
    // optional kernel timing (imt_profile_*)
    bool profiling = false;
    struct ProfPair { hipEvent_t a, b; int cls; };
    std::vector<ProfPair> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[IMT_PROF_CLASSES] = {0};
    double prof_n[IMT_PROF_CLASSES] = {0};
    hipEvent_t prof_event();
    // RAII-less helpers: begin returns an index into prof_pending (or -1 when off)
    int prof_begin(int cls, hipStream_t on = nullptr);
    void prof_end(int idx, hipStream_t on = nullptr);
    std::vector<hipStream_t> side_streams;   // internal streams imt_ctx_sync must also drain

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        last_error = buf;
        return code;
    }
    int hip_fail(hipError_t e, const char* what) {
        return fail(IMT_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    }
    // device scratch of at least `bytes` in slot `slot`; nullptr on failure (last_error set)
    void* dev_scratch(size_t slot, size_t bytes);
    // give a slot's buffer back if it has grown beyond `keep_below` bytes (a one-off bulk call: imt_itree_load); the
    // stream must have drained (the callers sync before they return)
    void trim_scratch(size_t slot, size_t keep_below);
    int set_device();
    // zero the error word / read it back (synchronises the stream)
    int clear_err();
    int sync_and_check();
};

#define IMT_HIP(ctx, call)                                   \
    do {                                                     \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess) return (ctx)->hip_fail(e__, #call); \
    } while (0)

struct imt_tree {
    imt_ctx* ctx = nullptr;
    size_t n_leaves = 0, n_levels = 0;
    uint8_t* d_nodes = nullptr;          // all levels, bottom-up, device format
    uint64_t* d_off = nullptr;           // [n_levels]
    uint64_t* d_len = nullptr;           // [n_levels]
    std::vector<uint64_t> h_off, h_len;
};
