// comm.hip -- the multi-GPU step behind the C ABI (no reference counterpart: zvdb is single-node).
//
// One process per GPU, the index sharded by vector-id range (ids are dense insertion order, hnsw.zig:77): every rank answers
// every query on its shard, ONE all-gather of the packed per-shard top-k over RCCL / xGMI (nq k 12 + nq 4 bytes per rank:
// latency-bound, single hop), then every rank merges the G lists by (distance, id) -- all enqueued on the CALLER's stream, no
// host round trip.  A Zig / C caller shards through these entry points alone (zvdb_amd/sharded.py is a thin caller of them).
//
// RCCL is bound at run time (dlopen of librccl.so.1, the library torch.distributed's "nccl" backend maps too -- one copy per
// process): an index user that never shards pays nothing and needs no librccl at RUN time.  (Building this file needs RCCL's
// header for the types below; ROCm ships it.)
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <string>

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
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;      // (optional: comm_query answers from the handle without them)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    bool ok = false;
    std::string load_error;  // dlerror() of the failed dlopen, kept once (a second dlerror() returns NULL)
};

Rccl &rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x.lib) break;
        }
        if (!x.lib) {
            const char *e = dlerror();
            x.load_error = e ? e : "dlopen failed";
            return x;
        }
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.lib, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.lib, "ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.lib, "ncclCommDestroy"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.lib, "ncclAllGather"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.lib, "ncclGetErrorString"));
        x.CommCount = reinterpret_cast<decltype(x.CommCount)>(dlsym(x.lib, "ncclCommCount"));
        x.CommUserRank = reinterpret_cast<decltype(x.CommUserRank)>(dlsym(x.lib, "ncclCommUserRank"));
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllGather && x.GetErrorString;
        return x;
    }();
    return r;
}

int need_rccl() {
    if (rccl().ok) return ZVDB_OK;
    set_error("RCCL is not available (dlopen librccl.so.1: %s)", rccl().lib ? "symbols missing" : rccl().load_error.c_str());
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
    // all-gather timing (zvdb_hip_comm_timing_*): HIP event pairs around ncclAllGather on the caller's stream, a bounded ring
    static constexpr int kTimingSlots = 64;
    int timing_on = 0;
    uint32_t timing_n = 0;
    hipEvent_t ev0[kTimingSlots] = {}, ev1[kTimingSlots] = {};
};

using namespace zvdb;

namespace {
// puts the calling thread back on the device it was on
struct DeviceGuard {
    int prev;
    explicit DeviceGuard(int p) : prev(p) {}
    ~DeviceGuard() { (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
}  // namespace

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

// Everything zvdb_hip_comm_create can refuse WITHOUT the other ranks: a host calls this on every rank and agrees on the outcome
// over its own channel before any rank enters the collective create (a rank that fails alone would leave the others waiting in
// ncclCommInitRank).  Not collective; changes no state.
int zvdb_hip_comm_precheck(int32_t device) {
    ZV_CHECK(need_rccl());
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device"); return ZVDB_ERR_NO_DEVICE; }
    if (device >= ndev) { set_error("comm: device %d of %d", device, ndev); return ZVDB_ERR_INVALID; }
    return ZVDB_OK;
}

int zvdb_hip_comm_create(uint32_t rank, uint32_t world, const void *unique_id, int32_t device, zvdb_hip_comm **out) {
    if (!unique_id || !out || world == 0 || rank >= world) { set_error("comm_create: bad arguments (rank %u of %u)", rank, world); return ZVDB_ERR_INVALID; }
    ZV_CHECK(zvdb_hip_comm_precheck(device));
    int prev = 0;
    ZV_HIP(hipGetDevice(&prev));
    if (device < 0) device = prev;
    DeviceGuard guard(prev);  // the calling thread's current device is the caller's business: restored on every way out
    ZV_HIP(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    zvdb_hip_comm *c = new (std::nothrow) zvdb_hip_comm;
    if (!c) { set_error("comm_create: out of memory"); return ZVDB_ERR_OOM; }
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
    for (int i = 0; i < zvdb_hip_comm::kTimingSlots; ++i) {
        if (c->ev0[i]) (void)hipEventDestroy(c->ev0[i]);
        if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]);
    }
    delete c;
}

int zvdb_hip_comm_info(zvdb_hip_comm *c, uint32_t *out_rank, uint32_t *out_world) {
    if (!c) { set_error("null argument"); return ZVDB_ERR_INVALID; }
    if (out_rank) *out_rank = c->rank;
    if (out_world) *out_world = c->world;
    return ZVDB_OK;
}

// What the communicator itself says (ncclCommUserRank / ncclCommCount: the ranks RCCL has joined, not the numbers the caller
// passed in) and the PCI bus id of the device it lives on -- a bench line gathers these over all ranks as its evidence that N
// ranks on N distinct GPUs took part.
int zvdb_hip_comm_query(zvdb_hip_comm *c, uint32_t *out_rccl_rank, uint32_t *out_rccl_ranks, char *out_bus_id, uint32_t bus_id_bytes) {
    if (!c) { set_error("null argument"); return ZVDB_ERR_INVALID; }
    int r = (int)c->rank, n = (int)c->world;
    if (rccl().CommUserRank) ZV_NCCL(rccl().CommUserRank(c->comm, &r));
    if (rccl().CommCount) ZV_NCCL(rccl().CommCount(c->comm, &n));
    if (out_rccl_rank) *out_rccl_rank = (uint32_t)r;
    if (out_rccl_ranks) *out_rccl_ranks = (uint32_t)n;
    if (out_bus_id && bus_id_bytes) {
        out_bus_id[0] = 0;
        ZV_HIP(hipDeviceGetPCIBusId(out_bus_id, (int)bus_id_bytes, c->device));
    }
    return ZVDB_OK;
}

// All-gather timing: on = 1 starts a new measurement, 2 resumes, 0 pauses (samples kept) -- the convention of zvdb_hip_timing_enable.
int zvdb_hip_comm_timing_enable(zvdb_hip_comm *c, int on) {
    if (!c) { set_error("null argument"); return ZVDB_ERR_INVALID; }
    if (on == 1) c->timing_n = 0;
    c->timing_on = on ? 1 : 0;
    return ZVDB_OK;
}

int zvdb_hip_comm_timing_read(zvdb_hip_comm *c, double *out_us_total, uint32_t *out_samples) {
    if (!c) { set_error("null argument"); return ZVDB_ERR_INVALID; }
    double us = 0.0;
    const uint32_t n = c->timing_n < (uint32_t)zvdb_hip_comm::kTimingSlots ? c->timing_n : (uint32_t)zvdb_hip_comm::kTimingSlots;
    for (uint32_t i = 0; i < n; ++i) {
        float ms = 0.f;
        ZV_HIP(hipEventSynchronize(c->ev1[i]));
        ZV_HIP(hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]));
        us += (double)ms * 1e3;
    }
    if (out_us_total) *out_us_total = us;
    if (out_samples) *out_samples = n;
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
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool timed = c->timing_on && c->timing_n < (uint32_t)zvdb_hip_comm::kTimingSlots;
    const uint32_t slot = c->timing_n;
    if (timed) {
        if (!c->ev0[slot]) {  // (events belong to the communicator's device, whatever device the calling thread is on)
            int prev = 0;
            ZV_HIP(hipGetDevice(&prev));
            DeviceGuard guard(prev);
            ZV_HIP(hipSetDevice(c->device));
            ZV_HIP(hipEventCreate(&c->ev0[slot]));
            ZV_HIP(hipEventCreate(&c->ev1[slot]));
        }
        ZV_HIP(hipEventRecord(c->ev0[slot], s));
    }
    ZV_NCCL(rccl().AllGather(ws, ws + pb, pb, ncclUint8, c->comm, s));
    if (timed) {
        ZV_HIP(hipEventRecord(c->ev1[slot], s));
        c->timing_n = slot + 1;
    }
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
