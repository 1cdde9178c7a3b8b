// imt_sliced_rccl.cpp -- the RCCL transport of imt_sliced_* (include/imt.h): ncclAllGather over xGMI, called by the
// library itself on its own communicators and streams (no torch, no host-language collective in between).  One
// communicator per round in flight (up to IMT_SLICED_ROUNDS): a communicator serialises its collectives, and the gathers
// of overlapping rounds must not queue behind each other -- each waits for ITS round's hash kernel first.
//
// RCCL is bound at RUN time (dlopen), not at link time: a process holds one RCCL, and when the host has already
// mapped one (PyTorch-ROCm bundles its own librccl.so next to its own HIP runtime) the library must use THAT copy rather
// than drag /opt/rocm's in beside it; a host without any gets /opt/rocm's through this library's RUNPATH; and the
// single-GPU entry points keep working on a box with no RCCL at all.  The calls are RCCL's C API, nothing in between.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>
#include <new>
#include "imt_sliced_transport.hpp"

using namespace imt::sliced;

namespace {

struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string error, path;
    bool ok = false;
};

const RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so"})          // the copy the process already holds, if any
            if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!h)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) {
            api.error = std::string("RCCL not found (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : "");
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(h, n);
            if (!p && api.error.empty()) api.error = std::string("RCCL lacks ") + n;
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.GetVersion = (decltype(api.GetVersion))sym("ncclGetVersion");
        Dl_info info;
        if (api.AllGather && dladdr((void*)api.AllGather, &info) && info.dli_fname) api.path = info.dli_fname;
        api.ok = api.error.empty();
    });
    return api;
}

struct RcclTransport : Transport {
    imt_ctx* ctx = nullptr;
    imt_transport* handle = nullptr;
    ncclComm_t comms[ROUNDS] = {};
    int n_comms = 0;
    bool owned = false;

    int fail(ncclResult_t r, const char* what) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s: %s", what, rccl().GetErrorString(r));
        if (ctx) ctx->last_error = buf;
        if (handle) handle->error = buf;
        return IMT_ERR_HIP;
    }
    ~RcclTransport() override {
        if (owned)
            for (int i = 0; i < n_comms; i++)
                if (comms[i]) rccl().CommDestroy(comms[i]);
    }
    int all_gather(Rank& rk, int slot, int r, size_t bytes, Stream st) override {
        const int i = rk.at(slot, r);
        const ncclResult_t res = rccl().AllGather(rk.send[i], rk.recv[i], bytes, ncclUint8, comms[slot % n_comms], (hipStream_t)st);
        return res == ncclSuccess ? IMT_OK : fail(res, "ncclAllGather");
    }
    int channels() const override { return n_comms; }
    int small_gather(const void* send, void* recv, size_t bytes, Stream st) override {
        const ncclResult_t res = rccl().AllGather(send, recv, bytes, ncclUint8, comms[0], (hipStream_t)st);
        return res == ncclSuccess ? IMT_OK : fail(res, "ncclAllGather");
    }
};

}  // namespace

extern "C" {

int imt_rccl_get_unique_id(void* id) {
    static_assert(sizeof(ncclUniqueId) == IMT_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!id) return IMT_ERR_ARG;
    if (!rccl().ok) return IMT_ERR_NO_DEVICE;
    return rccl().GetUniqueId((ncclUniqueId*)id) == ncclSuccess ? IMT_OK : IMT_ERR_HIP;
}

int imt_transport_rccl_create(imt_ctx* ctx, const void* unique_ids, int n_comms, int world, int rank, imt_transport** out) {
    if (!ctx || !out) return IMT_ERR_ARG;
    *out = nullptr;
    if (!unique_ids || n_comms < 1 || n_comms > ROUNDS || world < 1 || rank < 0 || rank >= world)
        return ctx->fail(IMT_ERR_ARG, "imt_transport_rccl_create: 1 <= n_comms <= %d, 0 <= rank < world", ROUNDS);
    if (!rccl().ok) return ctx->fail(IMT_ERR_NO_DEVICE, "%s", rccl().error.c_str());
    int rc = ctx->set_device();
    if (rc) return rc;
    RcclTransport* t = new (std::nothrow) RcclTransport();
    if (!t) return ctx->fail(IMT_ERR_ALLOC, "out of host memory");
    t->ctx = ctx;
    t->owned = true;
    for (int i = 0; i < n_comms; i++) {
        ncclUniqueId id;
        memcpy(&id, (const uint8_t*)unique_ids + (size_t)i * IMT_RCCL_UNIQUE_ID_BYTES, sizeof id);
        const ncclResult_t res = rccl().CommInitRank(&t->comms[i], world, id, rank);
        if (res != ncclSuccess) {
            rc = t->fail(res, "ncclCommInitRank");
            delete t;
            return rc;
        }
        t->n_comms = i + 1;
    }
    // One small all-gather per communicator, waited for: RCCL sets a communicator's connections up at its FIRST
    // collective (hundreds of milliseconds), and with one communicator per step in flight the last of them would
    // otherwise meet that inside somebody's timed region; it also fails here, loudly, if the ranks cannot reach
    // each other.
    {
        uint8_t* warm = nullptr;
        const size_t wb = 256;
        hipError_t e = hipMalloc((void**)&warm, wb * ((size_t)world + 1));
        if (e == hipSuccess) e = hipMemsetAsync(warm, 0, wb * ((size_t)world + 1), ctx->stream);
        ncclResult_t res = ncclSuccess;
        for (int i = 0; e == hipSuccess && res == ncclSuccess && i < n_comms; i++)
            res = rccl().AllGather(warm, warm + wb, wb, ncclUint8, t->comms[i], ctx->stream);
        if (e == hipSuccess && res == ncclSuccess) e = hipStreamSynchronize(ctx->stream);
        if (warm) hipFree(warm);
        if (e != hipSuccess || res != ncclSuccess) {
            rc = res != ncclSuccess ? t->fail(res, "ncclAllGather (warm-up)") : ctx->hip_fail(e, "RCCL warm-up");
            delete t;
            return rc;
        }
    }
    *out = imt_transport_wrap(t, ctx);
    if (*out) t->handle = *out;
    return *out ? IMT_OK : IMT_ERR_ALLOC;
}

int imt_transport_rccl_adopt(void* const* nccl_comms, int n_comms, imt_transport** out) {
    if (!out) return IMT_ERR_ARG;
    *out = nullptr;
    if (!nccl_comms || n_comms < 1 || n_comms > ROUNDS) return IMT_ERR_ARG;
    if (!rccl().ok) return IMT_ERR_NO_DEVICE;
    RcclTransport* t = new (std::nothrow) RcclTransport();
    if (!t) return IMT_ERR_ALLOC;
    for (int i = 0; i < n_comms; i++) {
        if (!nccl_comms[i]) {
            delete t;
            return IMT_ERR_ARG;
        }
        t->comms[i] = (ncclComm_t)nccl_comms[i];
    }
    t->n_comms = n_comms;
    *out = imt_transport_wrap(t, nullptr);
    if (*out) t->handle = *out;
    return *out ? IMT_OK : IMT_ERR_ALLOC;
}

// which RCCL the library is bound to ("" before the first RCCL call or if none was found), and its version code
const char* imt_rccl_library(int* version_out) {
    const RcclApi& a = rccl();
    int v = 0;
    if (a.ok && a.GetVersion) a.GetVersion(&v);
    if (version_out) *version_out = v;
    return a.ok ? a.path.c_str() : a.error.c_str();
}

}  // extern "C"
