// libpcr.so -- the one exchange step of the multi-GPU loop: sum of the 17 partials over the ranks.
//
// One process per GPU; clouds and indices replicated; samples sharded; per iteration ONE all-reduce of 136 bytes,
// enqueued on the context's stream between the fused step and the solve, so the device-resident loop
// (pcr_fit_*) never involves the host.  RCCL is bound at run time (dlopen): libpcr.so has no link-time
// dependency on it, and inside a PyTorch process the copy PyTorch already loaded is the one that gets used.
// bench.py --gpus N uses exactly this (pcr_allreduce_partials on the kernels' stream; torch.distributed only carries the
// unique id to the ranks).  Callers that already own a communicator (torch.distributed's "nccl" backend IS RCCL on
// ROCm) can use that instead: all-reduce pcr_fit_partials_d() themselves -- bench.py's agreed fallback when RCCL cannot
// be bound on every rank, and what its gloo rehearsal on a one-GPU box does.
#include <dlfcn.h>
#include <string.h>
#include "pcr_internal.h"

namespace {

// the handful of RCCL declarations needed (rccl.h: ncclUniqueId :43, ncclCommInitRank :220, ncclAllReduce :611)
struct UniqueId {
    char internal[128];
};
typedef void *Comm;
typedef int (*GetUniqueIdFn)(UniqueId *);
typedef int (*CommInitRankFn)(Comm *, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void *, void *, size_t, int, int, Comm, hipStream_t);
typedef const char *(*GetErrorStringFn)(int);
constexpr int kNcclDouble = 8, kNcclSum = 0;

struct Rccl {
    void *handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr;
    GetErrorStringFn get_error_string = nullptr;
};

Rccl *rccl() {
    static Rccl R;
    static bool tried = false;
    if (tried) return R.handle ? &R : nullptr;
    tried = true;
    // RTLD_NOLOAD first: reuse a copy that is already in the process (PyTorch ships its own)
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names)
        if (!R.handle) R.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char *n : names)
        if (!R.handle) R.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!R.handle) {
        pcr::set_last_error("RCCL is not available: %s", dlerror());
        return nullptr;
    }
    R.get_unique_id = (GetUniqueIdFn)dlsym(R.handle, "ncclGetUniqueId");
    R.comm_init_rank = (CommInitRankFn)dlsym(R.handle, "ncclCommInitRank");
    R.comm_destroy = (CommDestroyFn)dlsym(R.handle, "ncclCommDestroy");
    R.all_reduce = (AllReduceFn)dlsym(R.handle, "ncclAllReduce");
    R.get_error_string = (GetErrorStringFn)dlsym(R.handle, "ncclGetErrorString");
    if (!R.get_unique_id || !R.comm_init_rank || !R.comm_destroy || !R.all_reduce) {
        pcr::set_last_error("RCCL library lacks an expected entry point");
        dlclose(R.handle);
        R.handle = nullptr;
        return nullptr;
    }
    return &R;
}

int fail(Rccl *R, const char *what, int code) {
    pcr::set_last_error("%s failed: %s", what, R->get_error_string ? R->get_error_string(code) : "RCCL error");
    return PCR_E_HIP;
}

}  // namespace

struct pcr_comm {
    pcr_ctx *ctx;
    Comm comm;
    int rank, world;
};

extern "C" {

int pcr_comm_unique_id(char id[PCR_COMM_ID_BYTES]) {
    PCR_REQUIRE(id, "pcr_comm_unique_id: NULL argument");
    Rccl *R = rccl();
    if (!R) return PCR_E_UNSUPPORTED;
    UniqueId u;
    const int rc = R->get_unique_id(&u);
    if (rc != 0) return fail(R, "ncclGetUniqueId", rc);
    memcpy(id, u.internal, sizeof u.internal);
    return PCR_OK;
}

int pcr_comm_create(pcr_ctx *ctx, int rank, int world, const char id[PCR_COMM_ID_BYTES], pcr_comm **out) {
    PCR_REQUIRE(ctx && id && out, "pcr_comm_create: NULL argument");
    PCR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "pcr_comm_create: rank %d of %d", rank, world);
    *out = nullptr;
    Rccl *R = rccl();
    if (!R) return PCR_E_UNSUPPORTED;
    PCR_HIP(hipSetDevice(ctx->device));
    UniqueId u;
    memcpy(u.internal, id, sizeof u.internal);
    pcr_comm *c = new pcr_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    const int rc = R->comm_init_rank(&c->comm, world, u, rank);
    if (rc != 0) {
        delete c;
        return fail(R, "ncclCommInitRank", rc);
    }
    *out = c;
    return PCR_OK;
}

int pcr_allreduce_partials(pcr_comm *comm, double *partials_d) {
    PCR_REQUIRE(comm && partials_d, "pcr_allreduce_partials: NULL argument");
    Rccl *R = rccl();
    if (!R) return PCR_E_UNSUPPORTED;
    PCR_HIP(hipSetDevice(comm->ctx->device));
    const int rc = R->all_reduce(partials_d, partials_d, PCR_NUM_PARTIALS, kNcclDouble, kNcclSum, comm->comm, comm->ctx->stream);
    if (rc != 0) return fail(R, "ncclAllReduce", rc);
    return PCR_OK;
}

int pcr_reduce17_with_comm(void *comm, double *buf17_d) { return pcr_allreduce_partials((pcr_comm *)comm, buf17_d); }

int pcr_comm_destroy(pcr_comm *comm) {
    if (!comm) return PCR_OK;
    Rccl *R = rccl();
    if (R) {
        hipSetDevice(comm->ctx->device);
        hipStreamSynchronize(comm->ctx->stream);
        R->comm_destroy(comm->comm);
    }
    delete comm;
    return PCR_OK;
}

}  // extern "C"
