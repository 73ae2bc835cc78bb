// Multi-GPU statistics exchange over RCCL for hosts without torch.distributed.
//
// The reference is one process with one env and has no collective anywhere; here envs are sharded over one process per GPU
// and the ONLY exchange is the per-episode statistics block [E, 17] float64 (get_statistics, utils.py:12-123).  These entry
// points run that all-gather on the handle's own stream, ordered after the statistics kernel and overlapping nothing else.
// librccl is opened at first use (dlopen, local scope): the step path has no link-time dependency on it, and a process
// that already carries a copy of RCCL (PyTorch bundles one) keeps using its own.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

static RcclApi *rccl_api() {
    static RcclApi api;
    if (api.lib || !api.err.empty()) return &api;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
    }
    if (!api.lib) { api.err = std::string("cannot open librccl: ") + dlerror(); return &api; }
#define SYM(field, sym) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, sym)); \
    if (!api.field) { api.err = std::string("librccl lacks ") + sym; api.lib = nullptr; return &api; }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    return &api;
}

struct CommState {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0;
    double *d_send = nullptr;   // [E, 17] this rank's statistics
    int send_envs = 0;
    long long gathers = 0;      // all-gathers issued
    int checked_envs = -1;      // the env count every rank was verified to share (-1: not verified yet)
};
