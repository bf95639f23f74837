// Incumbent exchange of the sharded frontier through the C-ABI (include/gomilp_lp.h: gomilp_comm_*,
// gomilp_incumbent_allreduce): ONE RCCL all-reduce(min) per wave over xGMI.  Reference: the incumbent is the only state
// the solveWorker goroutines share (/root/reference/tree.go:207-263: checkSolution compares every candidate with
// `incumbent.z`, :228-230); with the frontier sharded over GPUs (one process per GPU) that comparison needs the global
// minimum — and, for determinism, the smallest child index that attains it (the node the reference's FIFO order meets first).
//
// Every rank contributes a table of 2 * world doubles: (z, index) in its own slots, +Inf elsewhere; min-reduce hands every
// rank the whole table in one collective (16 * world bytes: latency-bound), the lexicographic minimum is taken locally.
// RCCL is loaded with dlopen at first use, so the library itself has no link-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/gomilp_lp.h"

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.h) break;
        }
        if (!r.h) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.h, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.h, "ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.h, "ncclAllReduce"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
    });
    return r;
}

}  // namespace

struct gomilp_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    double *d_tab = nullptr, *h_tab = nullptr;   // 2 * world doubles: device table, pinned mirror
    std::mutex mu;
};

extern "C" {

int gomilp_comm_unique_id(char *id_out) {
    if (!id_out) return GOMILP_ERR_BAD_SHAPE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GOMILP_ERR_DEVICE;
    Rccl &r = rccl();
    if (!r.ok) return GOMILP_ERR_DEVICE;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return GOMILP_ERR_DEVICE;
    memcpy(id_out, id.internal, GOMILP_COMM_ID_BYTES);
    return GOMILP_OK;
}

gomilp_comm *gomilp_comm_create(int rank, int world, const char *id, int device, int *status) {
    auto fail = [&](int code) -> gomilp_comm * { if (status) *status = code; return nullptr; };
    if (world < 1 || rank < 0 || rank >= world || !id) return fail(GOMILP_ERR_BAD_SHAPE);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GOMILP_ERR_DEVICE);
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev) return fail(GOMILP_ERR_DEVICE);
    Rccl &r = rccl();
    if (!r.ok) return fail(GOMILP_ERR_DEVICE);
    if (hipSetDevice(device) != hipSuccess) return fail(GOMILP_ERR_DEVICE);
    gomilp_comm *c = new gomilp_comm;
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId uid;
    memcpy(uid.internal, id, GOMILP_COMM_ID_BYTES);
    const size_t bytes = (size_t)2 * world * sizeof(double);
    if (r.CommInitRank(&c->comm, world, uid, rank) != ncclSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->d_tab), bytes) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&c->h_tab), bytes, hipHostMallocDefault) != hipSuccess) {
        gomilp_comm_destroy(c);
        return fail(GOMILP_ERR_DEVICE);
    }
    if (status) *status = GOMILP_OK;
    return c;
}

void gomilp_comm_destroy(gomilp_comm *c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm) rccl().CommDestroy(c->comm);
    if (c->d_tab) hipFree(c->d_tab);
    if (c->h_tab) hipHostFree(c->h_tab);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int gomilp_comm_rank(const gomilp_comm *c) { return c ? c->rank : -1; }
int gomilp_comm_world(const gomilp_comm *c) { return c ? c->world : 0; }

int gomilp_incumbent_allreduce(gomilp_comm *c, double local_z, int64_t local_index, double *global_z, int64_t *global_index) {
    if (!c || !global_z || !global_index || local_index < 0 || local_index > GOMILP_NO_INCUMBENT) return GOMILP_ERR_BAD_SHAPE;
    std::lock_guard<std::mutex> g(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) return GOMILP_ERR_DEVICE;
    const int W = c->world;
    const bool has = !(local_z != local_z) && local_z < INFINITY && local_index < GOMILP_NO_INCUMBENT;   // NaN / +Inf / no index: nothing to offer
    for (int r = 0; r < 2 * W; r++) c->h_tab[r] = INFINITY;
    if (has) { c->h_tab[2 * c->rank] = local_z; c->h_tab[2 * c->rank + 1] = (double)local_index; }   // child indices are exact in a double
    const size_t bytes = (size_t)2 * W * sizeof(double);
    if (hipMemcpyAsync(c->d_tab, c->h_tab, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return GOMILP_ERR_DEVICE;
    if (rccl().AllReduce(c->d_tab, c->d_tab, (size_t)2 * W, ncclFloat64, ncclMin, c->comm, c->stream) != ncclSuccess) return GOMILP_ERR_DEVICE;
    if (hipMemcpyAsync(c->h_tab, c->d_tab, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return GOMILP_ERR_DEVICE;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return GOMILP_ERR_DEVICE;
    gomilp_incumbent_pick(c->h_tab, W, global_z, global_index);
    return GOMILP_OK;
}

// the host logic of the exchange, separately callable (and testable without a GPU): lexicographic minimum of a table of
// (z, index) pairs, +Inf = no candidate
void gomilp_incumbent_pick(const double *table, int world, double *global_z, int64_t *global_index) {
    double bz = INFINITY;
    int64_t bi = GOMILP_NO_INCUMBENT;
    for (int r = 0; r < world; r++) {
        const double z = table[2 * r], fi = table[2 * r + 1];
        if (!(z < INFINITY) || !(fi < INFINITY)) continue;
        const int64_t i = (int64_t)fi;
        if (z < bz || (z == bz && i < bi)) { bz = z; bi = i; }
    }
    *global_z = bz;
    *global_index = bi;
}

}  // extern "C"
