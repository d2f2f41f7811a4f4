// The only inter-GPU exchange of the path: an all-gather of the Result tensor over RCCL (xGMI).
// librccl.so is opened lazily so that single-GPU use never pays for (or depends on) it.
// Types, enumerators and prototypes come from <rccl/rccl.h> (compile time); only the symbols are resolved with dlsym.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "pvhip_common.h"

using namespace pvhip;

namespace {

static_assert(sizeof(ncclUniqueId) == PVHIP_UNIQUE_ID_BYTES, "pvhip.h promises a unique id of NCCL_UNIQUE_ID_BYTES bytes");
typedef ncclUniqueId UniqueId;
typedef ncclComm_t   Comm;

typedef decltype(&ncclGetUniqueId)    GetUniqueIdFn;
typedef decltype(&ncclCommInitRank)   CommInitRankFn;
typedef decltype(&ncclAllGather)      AllGatherFn;
typedef decltype(&ncclCommDestroy)    CommDestroyFn;
typedef decltype(&ncclCommCount)      CommCountFn;
typedef decltype(&ncclGetErrorString) GetErrorStringFn;

struct Rccl {
    void*            handle = nullptr;
    GetUniqueIdFn    get_unique_id = nullptr;
    CommInitRankFn   comm_init_rank = nullptr;
    AllGatherFn      all_gather = nullptr;
    CommDestroyFn    comm_destroy = nullptr;
    CommCountFn      comm_count = nullptr;
    GetErrorStringFn get_error_string = nullptr;
    Comm             comm = nullptr;
    int              rank = 0, world = 1;
};
Rccl& rccl() {
    static Rccl r;
    return r;
}

int load_rccl() {
    Rccl& r = rccl();
    if (r.handle != nullptr) return PVHIP_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) {
        r.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (r.handle != nullptr) break;
    }
    if (r.handle == nullptr) return fail(PVHIP_ECOMM, "cannot dlopen librccl.so: %s", dlerror());
    r.get_unique_id    = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank   = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
    r.all_gather       = (AllGatherFn)dlsym(r.handle, "ncclAllGather");
    r.comm_destroy     = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
    r.comm_count       = (CommCountFn)dlsym(r.handle, "ncclCommCount");
    r.get_error_string = (GetErrorStringFn)dlsym(r.handle, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy || !r.comm_count)
        return fail(PVHIP_ECOMM, "librccl.so lacks a required symbol");
    return PVHIP_OK;
}

int rccl_fail(const char* what, int code) {
    Rccl& r = rccl();
    return fail(PVHIP_ECOMM, "%s -> RCCL error %d (%s)", what, code,
                r.get_error_string ? r.get_error_string((ncclResult_t)code) : "?");
}

}  // namespace

extern "C" {

int pvhip_comm_unique_id(void* unique_id_out) {
    PVHIP_CHECK_ARG(unique_id_out != nullptr);
    int rc = load_rccl();
    if (rc) return rc;
    UniqueId id;
    int      e = rccl().get_unique_id(&id);
    if (e != 0) return rccl_fail("ncclGetUniqueId", e);
    memcpy(unique_id_out, id.internal, PVHIP_UNIQUE_ID_BYTES);
    return PVHIP_OK;
}

int pvhip_comm_init(const void* unique_id, int rank, int world) {
    PVHIP_REQUIRE_INIT();
    PVHIP_CHECK_ARG(unique_id != nullptr && world >= 1 && rank >= 0 && rank < world);
    int rc = load_rccl();
    if (rc) return rc;
    Rccl& r = rccl();
    if (r.comm != nullptr) return fail(PVHIP_EINVAL, "pvhip_comm_init: communicator already initialised");
    UniqueId id;
    memcpy(id.internal, unique_id, PVHIP_UNIQUE_ID_BYTES);
    int e = r.comm_init_rank(&r.comm, world, id, rank);
    if (e != 0) {
        r.comm = nullptr;
        return rccl_fail("ncclCommInitRank", e);
    }
    r.rank  = rank;
    r.world = world;
    return PVHIP_OK;
}

int pvhip_comm_allgather_f32(const float* send, float* recv, size_t count_per_rank) {
    PVHIP_REQUIRE_INIT();
    Rccl& r = rccl();
    if (r.comm == nullptr) return fail(PVHIP_ECOMM, "pvhip_comm_allgather_f32: communicator not initialised");
    if (count_per_rank == 0) return PVHIP_OK;
    PVHIP_CHECK_ARG(send != nullptr && recv != nullptr);
    int e = r.all_gather(send, recv, count_per_rank, ncclFloat32, r.comm, state().stream);
    if (e != 0) return rccl_fail("ncclAllGather", e);
    return PVHIP_OK;
}

int pvhip_comm_ranks(int* count) {
    PVHIP_CHECK_ARG(count != nullptr);
    Rccl& r = rccl();
    *count = 0;
    if (r.comm == nullptr) return fail(PVHIP_ECOMM, "pvhip_comm_ranks: communicator not initialised");
    int e = r.comm_count(r.comm, count);
    if (e != 0) return rccl_fail("ncclCommCount", e);
    return PVHIP_OK;
}

int pvhip_comm_destroy(void) {
    Rccl& r = rccl();
    if (r.comm == nullptr) return PVHIP_OK;
    if (state().ready) (void)hipStreamSynchronize(state().stream);
    int e  = r.comm_destroy(r.comm);
    r.comm = nullptr;
    if (e != 0) return rccl_fail("ncclCommDestroy", e);
    return PVHIP_OK;
}

}  // extern "C"
