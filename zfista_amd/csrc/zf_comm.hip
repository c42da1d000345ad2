// zf_comm.hip - RCCL inside the library ("context create(device id, optional RCCL
// rank/world/unique-id)"; 8e: the exchanges C1 / C2 of a sharded decision vector).
//
// A zf_comm wraps one ncclComm_t (RCCL over xGMI, one rank per GPU).  Attached to a solver
// (zf_solver_set_comm) it lets zf_solver_enqueue_init / zf_solver_enqueue_steps issue the packed
// all-gathers of a sharded step themselves, on the solver's stream, between the trial and the
// decide kernel: a multi-rank pass needs no host code per pass (round 1 drove it from a Python loop
// through torch.distributed).  The rank-ordered summation of the gathered packs stays where it was
// (zf_decide_kernel, zf_sum_parts_kernel): all ranks take bitwise-identical decisions.
//
// librccl is loaded with dlopen at the first zf_comm call - the library itself has no link-time
// dependency on it and single-GPU users never touch it.  Inside a process that already loaded
// RCCL (PyTorch does) the same soname resolves to that copy.
#include <dlfcn.h>

#include "zf_common.h"

namespace {
typedef void* nccl_comm_t;
struct nccl_unique_id {
    char internal[128];
};
struct rccl_api {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
rccl_api g_rccl;

int rccl_load() {
    if (g_rccl.handle) return ZF_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return zf_fail(ZF_ERR_STATE, "zf_comm: cannot load librccl (%s)", dlerror());
    g_rccl.GetUniqueId = (int (*)(nccl_unique_id*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(nccl_comm_t*, int, nccl_unique_id, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(nccl_comm_t))dlsym(h, "ncclCommDestroy");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather)
        return zf_fail(ZF_ERR_STATE, "zf_comm: librccl lacks an expected symbol%s");
    g_rccl.handle = h;
    return ZF_OK;
}
int rccl_check(int rc, const char* what) {
    if (rc == 0) return ZF_OK;
    return zf_fail(ZF_ERR_STATE, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
}
}  // namespace

struct zf_comm {
    nccl_comm_t comm = nullptr;
    int rank = 0, world = 1;
};

// 128 bytes that rank 0 creates and every rank of the communicator must receive (over any host
// channel: a file, MPI, torch.distributed, a socket) before zf_comm_create
extern "C" int zf_comm_unique_id(void* id128) {
    ZF_REQUIRE(id128, "zf_comm_unique_id: null argument");
    int rc = rccl_load();
    if (rc) return rc;
    return rccl_check(g_rccl.GetUniqueId(static_cast<nccl_unique_id*>(id128)), "ncclGetUniqueId");
}

// collective over the `world` ranks: every rank calls it with the same id, its own rank, and with
// its GPU current (zf_set_device / hipSetDevice)
extern "C" int zf_comm_create(zf_comm** out, int32_t rank, int32_t world, const void* id128) {
    ZF_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world, "zf_comm_create: bad argument");
    int rc = rccl_load();
    if (rc) return rc;
    zf_comm* c = new (std::nothrow) zf_comm();
    if (!c) return zf_fail(ZF_ERR_ARG, "zf_comm_create: out of host memory");
    nccl_unique_id id;
    memcpy(&id, id128, sizeof(id));
    rc = rccl_check(g_rccl.CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
    if (rc) {
        delete c;
        return rc;
    }
    c->rank = rank;
    c->world = world;
    *out = c;
    return ZF_OK;
}

extern "C" int zf_comm_destroy(zf_comm* c) {
    if (!c) return ZF_OK;
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return ZF_OK;
}

extern "C" int zf_comm_info(zf_comm* c, int32_t* rank, int32_t* world) {
    ZF_REQUIRE(c && rank && world, "zf_comm_info: null argument");
    *rank = c->rank;
    *world = c->world;
    return ZF_OK;
}

// recv (world x count doubles, rank-major) <- send (count doubles) of every rank, on `stream`
extern "C" int zf_comm_all_gather(zf_comm* c, const double* send_dev, double* recv_dev, int64_t count, void* stream) {
    ZF_REQUIRE(c && send_dev && recv_dev && count >= 0, "zf_comm_all_gather: bad argument");
    return rccl_check(g_rccl.AllGather(send_dev, recv_dev, (size_t)count, /*ncclDouble*/ 8, c->comm, (hipStream_t)stream),
                      "ncclAllGather");
}
