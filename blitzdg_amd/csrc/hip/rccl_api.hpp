// rccl_api.hpp -- RCCL bound at run time (dlopen) the first time a communicator is asked for, so that the library has no
// load-time dependency on it and single-GPU users never load it. Shared by the straight-element solver (sw2d_device.hip)
// and the curved one (sw2d_curved_device.hip).
#pragma once
#include "../host/capi_internal.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstdlib>
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

inline RcclApi& rccl() {
    static RcclApi api;
    if (api.handle) return api;
    // BDG_RCCL_LIBRARY: another library with the same nine entry points (the tests substitute a file-based
    // transport so that several ranks can share the one GPU of a test box, which RCCL itself refuses).
    const char* names[] = {std::getenv("BDG_RCCL_LIBRARY"), "librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
    }
    if (!api.handle) throw bdg_detail::hip_error(std::string("cannot load RCCL: ") + dlerror());
    auto sym = [&](const char* name) {
        void* p = dlsym(api.handle, name);
        if (!p) throw bdg_detail::hip_error(std::string("RCCL symbol missing: ") + name);
        return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    return api;
}

inline void ncclCheck(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw bdg_detail::hip_error(std::string(what) + ": " + rccl().GetErrorString(r));
}

} // namespace bdg_rccl
