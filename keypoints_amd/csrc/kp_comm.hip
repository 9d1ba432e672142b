// kp_comm_*: the gradient exchange of the data-parallel train step as plain C-ABI calls on RCCL ("no global
// state except an RCCL communicator handle (kp_comm_init/destroy/allreduce)"; 8e: one all-reduce per flat gradient arena,
// optional bf16 transport, broadcast of parameters / BatchNorm buffers from rank 0).
// RCCL is bound at RUN time (dlopen of the librccl the process already holds -- torch ships one -- or the system's), so the
// library loads, and every other entry point works, on a box without RCCL.  Collectives are enqueued on the caller's
// stream; nothing here synchronises with the host except communicator creation / destruction.
#include <dlfcn.h>
#include <mutex>
#include <string.h>
#include "kp_common.h"

namespace {

// the few RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes; enums by value)
struct UniqueId { char internal[128]; };
typedef void* Comm;
constexpr int kSum = 0, kAvg = 4, kF16 = 6, kF32 = 7, kBf16 = 9;

struct Api {
    void* lib = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommCount)(Comm, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Api& api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        // RTLD_NOLOAD first: the copy the process already mapped (torch's librccl.so), so that there is ONE RCCL per process
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) if (!a.lib) a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names) if (!a.lib) a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!a.lib) a.lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!a.lib) return;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
        a.CommCount = (decltype(a.CommCount))dlsym(a.lib, "ncclCommCount");
        a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
        a.Broadcast = (decltype(a.Broadcast))dlsym(a.lib, "ncclBroadcast");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.Broadcast;
    });
    return a;
}

int fail(const char* what, int rc) {
    const Api& a = api();
    kp_set_error("%s: RCCL error %d (%s)", what, rc, a.GetErrorString ? a.GetErrorString(rc) : "?");
    return KP_E_LAUNCH;
}

int nccl_type(int dtype) { return dtype == KP_F32 ? kF32 : dtype == KP_BF16 ? kBf16 : dtype == KP_F16 ? kF16 : -1; }

}  // namespace

extern "C" {

int kp_comm_available(void) { return api().ok ? 1 : 0; }

int kp_comm_unique_id(void* id128) {
    KP_REQUIRE(id128, "kp_comm_unique_id: null pointer");
    KP_REQUIRE(api().ok, "kp_comm_unique_id: no RCCL library in this process (librccl.so)");
    UniqueId id;
    const int rc = api().GetUniqueId(&id);
    if (rc) return fail("ncclGetUniqueId", rc);
    memcpy(id128, id.internal, sizeof(id.internal));
    return KP_OK;
}

int kp_comm_init(const void* id128, int rank, int world, void** comm) {
    KP_REQUIRE(id128 && comm && world > 0 && rank >= 0 && rank < world, "kp_comm_init: bad arguments (rank %d of %d)", rank, world);
    KP_REQUIRE(api().ok, "kp_comm_init: no RCCL library in this process (librccl.so)");
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    Comm c = nullptr;
    const int rc = api().CommInitRank(&c, world, id, rank);   // binds the communicator to the CURRENT HIP device
    if (rc) return fail("ncclCommInitRank", rc);
    *comm = c;
    return KP_OK;
}

int kp_comm_allreduce(void* comm, void* buf, size_t count, int dtype, int average, void* stream) {
    KP_REQUIRE(comm && (buf || count == 0), "kp_comm_allreduce: null pointer");
    const int t = nccl_type(dtype);
    KP_REQUIRE(t >= 0, "kp_comm_allreduce: bad dtype %d", dtype);
    if (count == 0) return KP_OK;
    const int rc = api().AllReduce(buf, buf, count, t, average ? kAvg : kSum, (Comm)comm, (hipStream_t)stream);
    return rc ? fail("ncclAllReduce", rc) : KP_OK;
}

int kp_comm_broadcast(void* comm, void* buf, size_t count, int dtype, int root, void* stream) {
    KP_REQUIRE(comm && (buf || count == 0) && root >= 0, "kp_comm_broadcast: bad arguments");
    const int t = nccl_type(dtype);
    KP_REQUIRE(t >= 0, "kp_comm_broadcast: bad dtype %d", dtype);
    if (count == 0) return KP_OK;
    const int rc = api().Broadcast(buf, buf, count, t, root, (Comm)comm, (hipStream_t)stream);
    return rc ? fail("ncclBroadcast", rc) : KP_OK;
}

int kp_comm_count(void* comm, int* ranks) {
    KP_REQUIRE(comm && ranks, "kp_comm_count: null pointer");
    KP_REQUIRE(api().ok && api().CommCount, "kp_comm_count: no RCCL library in this process");
    const int rc = api().CommCount((Comm)comm, ranks);   // what the COMMUNICATOR says, not what the launcher was told
    return rc ? fail("ncclCommCount", rc) : KP_OK;
}

int kp_comm_destroy(void* comm) {
    if (!comm) return KP_OK;
    KP_REQUIRE(api().ok, "kp_comm_destroy: no RCCL library in this process");
    const int rc = api().CommDestroy((Comm)comm);
    return rc ? fail("ncclCommDestroy", rc) : KP_OK;
}

}  // extern "C"
