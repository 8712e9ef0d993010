// comm.hip - RCCL called directly (C ABI: sp_comm_*), for the collectives of the train step that sit on the critical path.
//
// Why not torch.distributed for these: ProcessGroupNCCL runs every collective on its own stream and fences it with events on both sides -
// five host calls through Python per message (~40 us of host time).  The train step has 104 SyncBatchNorm messages (one [2C] sum per
// BatchNorm layer and direction, ddp...:89-90) in a chain of dependent launches: nothing can run under them, so their own stream buys
// nothing, and 104 x 40 us makes the 6.3 ms bf16 step host-bound (bench.py --sync-bn-latency-us: + 3.3 ms at zero latency).  Here a
// message is ONE call that enqueues ncclAllReduce on the caller's stream - stream order is the dependency.  The gradient buckets keep
// their own stream (they do overlap with backward) and can use either path.
//
// librccl is resolved at run time (dlopen) so that the library still loads - and everything single-GPU still runs - where RCCL is absent.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "sp_common.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;
    decltype(&ncclCommUserRank) comm_user_rank = nullptr;
    decltype(&ncclCommCuDevice) comm_device = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool tried = false;
};

Rccl& rccl() {
    static Rccl r;
    if (r.tried) return r;
    r.tried = true;
    // a copy that is already in the process (torch ships one and loads it with its distributed backend) wins: one RCCL per process
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!r.handle)
        for (const char* n : names)
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!r.handle) return r;
    r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(r.handle, "ncclAllReduce"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.handle, "ncclGetErrorString"));
    r.comm_count = reinterpret_cast<decltype(r.comm_count)>(dlsym(r.handle, "ncclCommCount"));
    r.comm_user_rank = reinterpret_cast<decltype(r.comm_user_rank)>(dlsym(r.handle, "ncclCommUserRank"));
    r.comm_device = reinterpret_cast<decltype(r.comm_device)>(dlsym(r.handle, "ncclCommCuDevice"));
    if (!(r.get_unique_id && r.comm_init_rank && r.all_reduce && r.comm_destroy && r.error_string && r.comm_count && r.comm_user_rank &&
          r.comm_device))
        r.handle = nullptr;
    return r;
}

int fail(const char* what, ncclResult_t rc) {
    sp_set_error("%s: %s", what, rccl().error_string ? rccl().error_string(rc) : "RCCL error");
    return SP_ELAUNCH;
}

}  // namespace

#define SP_NEED_RCCL()                                                                                      \
    do {                                                                                                    \
        if (!rccl().handle) {                                                                               \
            sp_set_error("librccl.so not found (dlopen): the sp_comm_* entry points need RCCL");            \
            return SP_ELAUNCH;                                                                              \
        }                                                                                                   \
    } while (0)

extern "C" int sp_comm_available(void) { return rccl().handle ? 1 : 0; }

extern "C" int sp_comm_unique_id(void* id128) {
    SP_REQUIRE(id128, "sp_comm_unique_id: null pointer");
    SP_NEED_RCCL();
    static_assert(sizeof(ncclUniqueId) == 128, "the C ABI hands the id over as 128 bytes");
    ncclUniqueId id;
    const ncclResult_t rc = rccl().get_unique_id(&id);
    if (rc != ncclSuccess) return fail("ncclGetUniqueId", rc);
    memcpy(id128, &id, sizeof(id));
    return SP_OK;
}

extern "C" int sp_comm_create(const void* id128, int world, int rank, void** comm) {
    SP_REQUIRE(id128 && comm && world >= 1 && rank >= 0 && rank < world, "sp_comm_create: bad argument");
    SP_NEED_RCCL();
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t rc = rccl().comm_init_rank(&c, world, id, rank);     // collective: every rank of the group calls it (current device)
    if (rc != ncclSuccess) return fail("ncclCommInitRank", rc);
    *comm = c;
    return SP_OK;
}

extern "C" int sp_comm_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream) {
    SP_REQUIRE(comm && buf && n > 0, "sp_comm_allreduce_sum_f32: bad argument");
    SP_NEED_RCCL();
    const ncclResult_t rc = rccl().all_reduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
    if (rc != ncclSuccess) return fail("ncclAllReduce", rc);
    return SP_OK;
}

// The SyncBatchNorm forward message is (sum, sum of squares) per channel in fp64 (train.hip: bn_sums_from_conv_kernel): summed as
// ncclFloat64.  (Round 3 sent these buffers through the f32 entry point with the fp64 element count - wrong for world > 1.)
extern "C" int sp_comm_allreduce_sum_f64(void* comm, double* buf, int64_t n, void* stream) {
    SP_REQUIRE(comm && buf && n > 0, "sp_comm_allreduce_sum_f64: bad argument");
    SP_NEED_RCCL();
    const ncclResult_t rc = rccl().all_reduce(buf, buf, (size_t)n, ncclFloat64, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
    if (rc != ncclSuccess) return fail("ncclAllReduce", rc);
    return SP_OK;
}

// What RCCL itself says about a communicator: ranks in it, this process's rank, the HIP device it was created on (bench.py prints
// these per rank so that an N-GPU line proves N ranks met).
extern "C" int sp_comm_info(void* comm, int* world, int* rank, int* device) {
    SP_REQUIRE(comm && world && rank && device, "sp_comm_info: null pointer");
    SP_NEED_RCCL();
    ncclResult_t rc = rccl().comm_count((ncclComm_t)comm, world);
    if (rc != ncclSuccess) return fail("ncclCommCount", rc);
    rc = rccl().comm_user_rank((ncclComm_t)comm, rank);
    if (rc != ncclSuccess) return fail("ncclCommUserRank", rc);
    rc = rccl().comm_device((ncclComm_t)comm, device);
    if (rc != ncclSuccess) return fail("ncclCommCuDevice", rc);
    return SP_OK;
}

extern "C" int sp_comm_destroy(void* comm) {
    if (!comm) return SP_OK;
    SP_NEED_RCCL();
    const ncclResult_t rc = rccl().comm_destroy((ncclComm_t)comm);
    if (rc != ncclSuccess) return fail("ncclCommDestroy", rc);
    return SP_OK;
}
