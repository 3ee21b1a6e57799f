// comm.hip -- the multi-GPU step behind the C ABI (no reference counterpart: zvdb is single-node).
//
// One process per GPU, the index sharded by vector-id range (ids are dense insertion order, hnsw.zig:77): every rank answers
// every query on its shard, ONE all-gather of the packed per-shard top-k over RCCL / xGMI (nq k 12 + nq 4 bytes per rank:
// latency-bound, single hop), then every rank merges the G lists by (distance, id) -- all enqueued on the CALLER's stream, no
// host round trip.  A Zig / C caller shards through these entry points alone (zvdb_amd/sharded.py is a thin caller of them).
//
// RCCL is bound at run time (dlopen of librccl.so.1, the library torch.distributed's "nccl" backend maps too -- one copy per
// process): an index user that never shards pays nothing and needs no RCCL installed.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "common.h"

namespace zvdb {
namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x.lib) break;
        }
        if (!x.lib) return x;
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.lib, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.lib, "ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.lib, "ncclCommDestroy"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.lib, "ncclAllGather"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.lib, "ncclGetErrorString"));
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllGather && x.GetErrorString;
        return x;
    }();
    return r;
}

int need_rccl() {
    if (rccl().ok) return ZVDB_OK;
    set_error("RCCL is not available (dlopen librccl.so.1: %s)", rccl().lib ? "symbols missing" : dlerror());
    return ZVDB_ERR_UNSUPPORTED;
}

#define ZV_NCCL(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess) {                                                                   \
            zvdb::set_error("%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(_r), __FILE__, __LINE__); \
            return ZVDB_ERR_HIP;                                                                   \
        }                                                                                          \
    } while (0)

uint64_t pack_bytes(uint32_t nq, uint32_t k) { return round_up64((uint64_t)nq * k * 12 + (uint64_t)nq * 4, 256); }

}  // namespace
}  // namespace zvdb

struct zvdb_hip_comm {
    ncclComm_t comm = nullptr;
    uint32_t rank = 0, world = 1;
    int device = 0;
};

using namespace zvdb;

extern "C" {

int zvdb_hip_comm_unique_id(void *out_id) {
    if (!out_id) { set_error("null argument"); return ZVDB_ERR_INVALID; }
    ZV_CHECK(need_rccl());
    static_assert(sizeof(ncclUniqueId) == ZVDB_COMM_ID_BYTES, "ZVDB_COMM_ID_BYTES");
    ncclUniqueId id;
    ZV_NCCL(rccl().GetUniqueId(&id));
    std::memcpy(out_id, &id, sizeof(id));
    return ZVDB_OK;
}

int zvdb_hip_comm_create(uint32_t rank, uint32_t world, const void *unique_id, int32_t device, zvdb_hip_comm **out) {
    if (!unique_id || !out || world == 0 || rank >= world) { set_error("comm_create: bad arguments (rank %u of %u)", rank, world); return ZVDB_ERR_INVALID; }
    ZV_CHECK(need_rccl());
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device"); return ZVDB_ERR_NO_DEVICE; }
    if (device < 0) ZV_HIP(hipGetDevice(&device));
    ZV_HIP(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    zvdb_hip_comm *c = new zvdb_hip_comm;
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclResult_t r = rccl().CommInitRank(&c->comm, (int)world, id, (int)rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank failed: %s", rccl().GetErrorString(r));
        delete c;
        return ZVDB_ERR_HIP;
    }
    *out = c;
    return ZVDB_OK;
}

void zvdb_hip_comm_destroy(zvdb_hip_comm *c) {
    if (!c) return;
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    delete c;
}

int zvdb_hip_comm_info(zvdb_hip_comm *c, uint32_t *out_rank, uint32_t *out_world) {
    if (!c) { set_error("null argument"); return ZVDB_ERR_INVALID; }
    if (out_rank) *out_rank = c->rank;
    if (out_world) *out_world = c->world;
    return ZVDB_OK;
}

// workspace of the sharded calls: [this rank's packed result][world packed results, gathered][the local search's own workspace]
int zvdb_hip_sharded_workspace(zvdb_hip_index *h, zvdb_hip_comm *c, uint32_t nq, uint32_t k, uint32_t mode, uint64_t *out_bytes) {
    if (!h || !c || !out_bytes) { set_error("null argument"); return ZVDB_ERR_INVALID; }
    uint64_t local = 0;
    if (mode != ZVDB_SHARDED_GRAPH) ZV_CHECK(zvdb_hip_bruteforce_workspace(h, nq, k, mode, &local));
    *out_bytes = pack_bytes(nq, k) * (1 + (uint64_t)c->world) + round_up64(local, 256);
    return ZVDB_OK;
}

static int gather_and_merge(zvdb_hip_comm *c, unsigned char *ws, uint32_t nq, uint32_t k, void *d_ids, void *d_dist, void *d_count, void *stream) {
    const uint64_t pb = pack_bytes(nq, k);
    ZV_NCCL(rccl().AllGather(ws, ws + pb, pb, ncclUint8, c->comm, static_cast<hipStream_t>(stream)));
    return zvdb_hip_merge_topk_packed_dev(ws + pb, pb, c->world, nq, k, d_ids, d_dist, d_count, stream);
}

int zvdb_hip_search_bruteforce_sharded_dev(zvdb_hip_index *h, zvdb_hip_comm *c, const void *d_queries, uint32_t nq, uint32_t dim, uint32_t k,
                                           uint32_t mode, uint64_t id_offset, void *d_workspace, uint64_t workspace_bytes, void *d_ids,
                                           void *d_dist, void *d_count, void *stream) {
    if (!h || !c || !d_workspace) { set_error("null argument"); return ZVDB_ERR_INVALID; }
    uint64_t need = 0;
    ZV_CHECK(zvdb_hip_sharded_workspace(h, c, nq, k, mode, &need));
    if (workspace_bytes < need) { set_error("sharded workspace too small: need %llu bytes", (unsigned long long)need); return ZVDB_ERR_INVALID; }
    unsigned char *ws = static_cast<unsigned char *>(d_workspace);
    const uint64_t pb = pack_bytes(nq, k);
    unsigned char *local = ws + pb * (1 + (uint64_t)c->world);
    ZV_CHECK(zvdb_hip_search_bruteforce_dev(h, d_queries, nq, dim, k, mode, id_offset, local, workspace_bytes - pb * (1 + (uint64_t)c->world), ws,
                                            ws + (size_t)nq * k * 8, ws + (size_t)nq * k * 12, stream));
    return gather_and_merge(c, ws, nq, k, d_ids, d_dist, d_count, stream);
}

int zvdb_hip_search_graph_sharded_dev(zvdb_hip_index *h, zvdb_hip_comm *c, const void *d_queries, uint32_t nq, uint32_t dim, uint32_t k,
                                      uint32_t ef, uint64_t entry, uint64_t id_offset, void *d_workspace, uint64_t workspace_bytes, void *d_ids,
                                      void *d_dist, void *d_count, void *d_stats, void *stream) {
    if (!h || !c || !d_workspace) { set_error("null argument"); return ZVDB_ERR_INVALID; }
    const uint64_t pb = pack_bytes(nq, k);
    if (workspace_bytes < pb * (1 + (uint64_t)c->world)) { set_error("sharded workspace too small"); return ZVDB_ERR_INVALID; }
    unsigned char *ws = static_cast<unsigned char *>(d_workspace);
    ZV_CHECK(zvdb_hip_search_graph_dev(h, d_queries, nq, dim, k, ef, entry, id_offset, ws, ws + (size_t)nq * k * 8, ws + (size_t)nq * k * 12, d_stats, stream));
    return gather_and_merge(c, ws, nq, k, d_ids, d_dist, d_count, stream);
}

}  // extern "C"
