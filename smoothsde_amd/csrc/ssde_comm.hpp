// ssde_comm.hpp -- RCCL entry points of the engine, bound lazily.
//
// The library does not link librccl: a one-GPU host (the reference's usual case, one R process on one device) should
// not pay for loading it.  The first multi-GPU use dlopen()s "librccl.so.1" -- the soname resolves to the copy the
// process already holds (PyTorch ships one) or to the ROCm installation through this library's RUNPATH -- and binds
// the eight calls the engine makes.  Types are taken from <rccl/rccl.h>; nothing else of RCCL is used.
#ifndef SSDE_COMM_HPP
#define SSDE_COMM_HPP

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <string>

namespace ssde_engine {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;

    bool load() {
        if (lib) return true;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* nm : names) {
            lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("librccl not loadable: ") + dlerror(); return false; }
        bool ok = true;
        auto bind = [&](auto& fp, const char* sym) {
            fp = reinterpret_cast<std::decay_t<decltype(fp)>>(dlsym(lib, sym));
            if (!fp) { ok = false; err = std::string("librccl lacks ") + sym; }
        };
        bind(GetUniqueId, "ncclGetUniqueId");
        bind(CommInitRank, "ncclCommInitRank");
        bind(CommInitAll, "ncclCommInitAll");
        bind(CommDestroy, "ncclCommDestroy");
        bind(CommCount, "ncclCommCount");
        bind(AllReduce, "ncclAllReduce");
        bind(GroupStart, "ncclGroupStart");
        bind(GroupEnd, "ncclGroupEnd");
        bind(GetErrorString, "ncclGetErrorString");
        if (!ok) { dlclose(lib); lib = nullptr; }
        return ok;
    }
};

RcclApi& rccl();   // process-wide instance (ssde_engine_dist.hip)

}  // namespace ssde_engine
#endif
