// lv_comm.hip — the one exchange step of the multi-GPU path (SURVEY §8 row e): scan points shard across the
// GPUs of a node (one process per GPU), the map is replicated, and each measurement pass all-reduces the
// 96-double H^T H / H^T h record.  The collective is RCCL's (over xGMI), issued from here directly on the
// context stream, so a whole iterated update is enqueued without a host round trip per pass.
//
// librccl is bound at run time (dlopen of the path the caller names — for a torch process that is the copy
// torch bundles, which is already loaded and must not be duplicated), through the five classic NCCL entry
// points whose signatures have been stable since NCCL 2.0.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>

#include "lv_host.hpp"

namespace lv {

namespace {
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(void* id) = nullptr;
    int (*CommInitRank)(void** comm, int nranks, UniqueId128 id, int rank) = nullptr;  // id by value (128-byte struct)
    int (*AllReduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t sendcount, int dtype, void* comm, hipStream_t stream) = nullptr;   // optional
    int (*CommDestroy)(void* comm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    char path[1024] = {0};
};
RcclApi g_rccl;

constexpr int kNcclFloat64 = 8;  // ncclDouble
constexpr int kNcclSum = 0;

int bind_rccl(const char* library) {
    const char* lib = (library && library[0]) ? library : "librccl.so.1";
    if (g_rccl.handle) {
        if (std::strcmp(g_rccl.path, lib) != 0) {
            set_error("RCCL already bound from %s (asked for %s)", g_rccl.path, lib);
            return LV_ESTATE;
        }
        return LV_OK;
    }
    void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { set_error("dlopen(%s): %s", lib, dlerror()); return LV_ENODEV; }
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy) {
        set_error("%s does not export the NCCL entry points", lib);
        return LV_ENODEV;
    }
    std::snprintf(a.path, sizeof(a.path), "%s", lib);
    g_rccl = a;
    return LV_OK;
}

const char* rccl_error(int code) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "?"; }
}  // namespace

int comm_unique_id(const char* library, void* id128) {
    int rc = bind_rccl(library);
    if (rc) return rc;
    const int e = g_rccl.GetUniqueId(id128);
    if (e) { set_error("ncclGetUniqueId: %s", rccl_error(e)); return LV_EHIP; }
    return LV_OK;
}

int comm_init(const char* library, const void* id128, int rank, int world, void** comm_out) {
    int rc = bind_rccl(library);
    if (rc) return rc;
    UniqueId128 id;
    std::memcpy(&id, id128, sizeof(id));
    void* comm = nullptr;
    const int e = g_rccl.CommInitRank(&comm, world, id, rank);
    if (e) { set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, rccl_error(e)); return LV_EHIP; }
    *comm_out = comm;
    return LV_OK;
}

int comm_destroy(void* comm) {
    if (!comm || !g_rccl.CommDestroy) return LV_OK;
    const int e = g_rccl.CommDestroy(comm);
    if (e) { set_error("ncclCommDestroy: %s", rccl_error(e)); return LV_EHIP; }
    return LV_OK;
}

// in-place sum of the 96-double record over all ranks, ordered on `stream` like a kernel
int comm_allreduce_record(void* comm, double* record, hipStream_t stream) {
    const int e = g_rccl.AllReduce(record, record, (size_t)SUMS_LEN, kNcclFloat64, kNcclSum, comm, stream);
    if (e) { set_error("ncclAllReduce: %s", rccl_error(e)); return LV_EHIP; }
    return LV_OK;
}

// The one-launch-per-pass form of the multi-GPU update: every rank's pass_kernel leaves its workgroup partials in ITS slot
// of `buf` (world slots of count_per_rank doubles); this gathers all slots on every rank, in place, ordered on `stream`.
// Every rank's next launch then folds all ranks' partials itself, in the same fixed order: identical states without a
// broadcast, and no reduce / solve kernels between the passes.
bool comm_has_allgather() { return g_rccl.AllGather != nullptr; }
int comm_allgather_inplace(void* comm, double* buf, size_t count_per_rank, int rank, hipStream_t stream) {
    if (!g_rccl.AllGather) { set_error("librccl does not export ncclAllGather"); return LV_ENODEV; }
    const int e = g_rccl.AllGather(buf + (size_t)rank * count_per_rank, buf, count_per_rank, kNcclFloat64, comm, stream);
    if (e) { set_error("ncclAllGather: %s", rccl_error(e)); return LV_EHIP; }
    return LV_OK;
}

}  // namespace lv
