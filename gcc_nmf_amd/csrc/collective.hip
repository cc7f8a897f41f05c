// RCCL binding of the shared-dictionary all-reduce hook (include/gccnmf_hip.h: gccnmf_allreduce_fn).
// The exchange is ONE ncclAllReduce(sum, f32) of [num (Fp*Kp) || den (Kp)] per KL-NMF iteration (the W update of
// gccNMF/gccNMFFunctions.py:77 summed over every rank's columns) -- 2.1 MB at K = 1024, latency-bound on xGMI, so it
// stays one fused buffer on the compute stream (the next W.H depends on it: nothing to overlap with).
//
// librccl is bound with dlopen at first use: a process that already holds a copy (PyTorch ships its own librccl.so with the
// same soname) keeps using that one instance; a host without RCCL can still load libgccnmf_hip.so for single-GPU work.
#include <dlfcn.h>
#include <mutex>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// building without the RCCL headers: the handful of declarations this file binds at run time (ABI of librccl.so.1 / nccl.h)
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream);
}
#endif
#include "common.h"
#include "../../include/gccnmf_hip.h"

namespace {
struct RcclApi {
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    bool ok = false;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);           // the instance this process already uses
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        api.get_unique_id = (decltype(api.get_unique_id))dlsym(h, "ncclGetUniqueId");
        api.comm_init_rank = (decltype(api.comm_init_rank))dlsym(h, "ncclCommInitRank");
        api.comm_destroy = (decltype(api.comm_destroy))dlsym(h, "ncclCommDestroy");
        api.all_reduce = (decltype(api.all_reduce))dlsym(h, "ncclAllReduce");
        api.ok = api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_reduce;
    });
    return api;
}
static_assert(sizeof(ncclUniqueId) == GCCNMF_RCCL_UNIQUE_ID_BYTES, "unique id size");
}  // namespace

extern "C" {

int gccnmf_rccl_available(void) {
    GCCNMF_ENTER(); return rccl().ok ? 1 : 0; }

int gccnmf_rccl_unique_id(char* id_bytes) {
    GCCNMF_ENTER();
    if (!id_bytes) return GCCNMF_ERR_ARG;
    if (!rccl().ok) return GCCNMF_ERR_COLLECTIVE;
    ncclUniqueId id;
    if (rccl().get_unique_id(&id) != ncclSuccess) return GCCNMF_ERR_COLLECTIVE;
    for (int i = 0; i < GCCNMF_RCCL_UNIQUE_ID_BYTES; ++i) id_bytes[i] = id.internal[i];
    return GCCNMF_OK;
}

int gccnmf_rccl_comm_init(const char* id_bytes, int world_size, int rank, void** comm) {
    GCCNMF_ENTER();
    if (!id_bytes || !comm || world_size < 1 || rank < 0 || rank >= world_size) return GCCNMF_ERR_ARG;
    if (!rccl().ok) return GCCNMF_ERR_COLLECTIVE;
    ncclUniqueId id;
    for (int i = 0; i < GCCNMF_RCCL_UNIQUE_ID_BYTES; ++i) id.internal[i] = id_bytes[i];
    ncclComm_t c = nullptr;
    if (rccl().comm_init_rank(&c, world_size, id, rank) != ncclSuccess) return GCCNMF_ERR_COLLECTIVE;
    *comm = (void*)c;
    return GCCNMF_OK;
}

int gccnmf_rccl_comm_destroy(void* comm) {
    GCCNMF_ENTER();
    if (!comm) return GCCNMF_ERR_ARG;
    if (!rccl().ok) return GCCNMF_ERR_COLLECTIVE;
    return rccl().comm_destroy((ncclComm_t)comm) == ncclSuccess ? GCCNMF_OK : GCCNMF_ERR_COLLECTIVE;
}

int gccnmf_rccl_allreduce(void* comm, float* buf, long count, void* stream) {
    GCCNMF_ENTER();
    if (!comm || !buf || count < 0) return GCCNMF_ERR_ARG;
    if (!rccl().ok) return GCCNMF_ERR_COLLECTIVE;
    return rccl().all_reduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream) == ncclSuccess
               ? GCCNMF_OK
               : GCCNMF_ERR_COLLECTIVE;
}

gccnmf_allreduce_fn gccnmf_rccl_allreduce_hook(void) {
    GCCNMF_ENTER(); return &gccnmf_rccl_allreduce; }

}  // extern "C"
