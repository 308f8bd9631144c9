// rccl_api.hpp -- RCCL bound at run time (dlopen) the first time a communicator is asked for, so that the library has no
// load-time dependency on it and single-GPU users never load it. Shared by the straight-element solver (sw2d_device.hip)
// and the curved one (sw2d_curved_device.hip).
#pragma once
#include "../host/capi_internal.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstdlib>
#include <mutex>
#include <string>

namespace bdg_rccl {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// The table is published only once all nine entry points are bound: a library that lacks one (a BDG_RCCL_LIBRARY without,
// say, ncclAllReduce) makes EVERY call throw, instead of the first one throwing and the later ones jumping through a
// half-filled table. First calls from two threads are serialised by the mutex.
inline RcclApi& rccl() {
    static RcclApi api;
    static std::mutex lock;
    std::lock_guard<std::mutex> hold(lock);
    if (api.handle) return api;
    RcclApi local;
    // BDG_RCCL_LIBRARY: another library with the same nine entry points (the tests substitute a file-based
    // transport so that several ranks can share the one GPU of a test box, which RCCL itself refuses).
    const char* names[] = {std::getenv("BDG_RCCL_LIBRARY"), "librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        local.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (local.handle) break;
    }
    if (!local.handle) {
        const char* why = dlerror();
        throw bdg_detail::hip_error(std::string("cannot load RCCL: ") + (why ? why : "no library name given"));
    }
    auto sym = [&](const char* name) {
        void* p = dlsym(local.handle, name);
        if (!p) {
            dlclose(local.handle);
            throw bdg_detail::hip_error(std::string("RCCL symbol missing: ") + name);
        }
        return p;
    };
    local.GetUniqueId = reinterpret_cast<decltype(local.GetUniqueId)>(sym("ncclGetUniqueId"));
    local.CommInitRank = reinterpret_cast<decltype(local.CommInitRank)>(sym("ncclCommInitRank"));
    local.CommDestroy = reinterpret_cast<decltype(local.CommDestroy)>(sym("ncclCommDestroy"));
    local.GroupStart = reinterpret_cast<decltype(local.GroupStart)>(sym("ncclGroupStart"));
    local.GroupEnd = reinterpret_cast<decltype(local.GroupEnd)>(sym("ncclGroupEnd"));
    local.Send = reinterpret_cast<decltype(local.Send)>(sym("ncclSend"));
    local.Recv = reinterpret_cast<decltype(local.Recv)>(sym("ncclRecv"));
    local.AllReduce = reinterpret_cast<decltype(local.AllReduce)>(sym("ncclAllReduce"));
    local.GetErrorString = reinterpret_cast<decltype(local.GetErrorString)>(sym("ncclGetErrorString"));
    api = local;
    return api;
}

inline void ncclCheck(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw bdg_detail::hip_error(std::string(what) + ": " + rccl().GetErrorString(r));
}

} // namespace bdg_rccl
