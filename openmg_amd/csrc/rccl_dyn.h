// RCCL, resolved at run time so that libopenmg_hip.so loads on hosts without it (dist.hip defines g_rccl).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <string>

#include "common.h"

namespace omg {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;

    template <typename F>
    void sym(F &fn, const char *name) {
        fn = reinterpret_cast<F>(dlsym(handle, name));
        if (!fn) throw Error(OMG_ERR_UNSUPPORTED, std::string("RCCL symbol missing: ") + name);
    }
    void load() {
        if (handle) return;
        // OMG_RCCL_LIB: the library to take the eleven symbols from instead (RTLD_LOCAL: its names must not shadow a
        // real librccl PyTorch has loaded).  tests/fake_rccl builds one that lets ranks SHARING a GPU run these call
        // sites; a path that does not load is an error, never a silent fall back to the real library.
        if (const char *over = getenv("OMG_RCCL_LIB"); over && *over) {
            handle = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            if (!handle) throw Error(OMG_ERR_UNSUPPORTED, std::string("OMG_RCCL_LIB: cannot load ") + over + ": " + dlerror());
        }
        // an already-loaded copy (e.g. the one PyTorch brought) wins; then the ROCm install
        for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (handle) break;
            handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!handle) throw Error(OMG_ERR_UNSUPPORTED, std::string("cannot load librccl: ") + dlerror());
        sym(GetUniqueId, "ncclGetUniqueId");
        sym(CommInitRank, "ncclCommInitRank");
        sym(CommDestroy, "ncclCommDestroy");
        sym(CommCount, "ncclCommCount");
        sym(GetErrorString, "ncclGetErrorString");
        sym(GroupStart, "ncclGroupStart");
        sym(GroupEnd, "ncclGroupEnd");
        sym(Send, "ncclSend");
        sym(Recv, "ncclRecv");
        sym(AllGather, "ncclAllGather");
        sym(AllReduce, "ncclAllReduce");
    }
};
extern Rccl g_rccl;

#define OMG_NCCL(call)                                                                        \
    do {                                                                                      \
        ncclResult_t r_ = (call);                                                             \
        if (r_ != ncclSuccess)                                                                \
            throw omg::Error(OMG_ERR_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

template <typename V> struct NcclType;
template <> struct NcclType<double> { static constexpr ncclDataType_t value = ncclDouble; };
template <> struct NcclType<float> { static constexpr ncclDataType_t value = ncclFloat; };


}  // namespace omg
