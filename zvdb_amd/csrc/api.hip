// api.hip -- extern "C" entry points of libzvdb_hip.so (declared in include/zvdb_hip.h)
// plus the host-side plumbing they share: error string, staging buffers, capacity growth.
#include <cstdarg>
#include <cstring>
#include <string>
#include <algorithm>

#include "common.h"
#include <sys/stat.h>

#include <sys/types.h>
#include <map>
#include <unistd.h>

namespace zvdb {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Tuning table: the A/B and test switches of the kernels' launch code (DESIGN.md "Runtime switches") are entries of a
// process-wide name -> value table that the CALLER fills (zvdb_hip_set_tuning); an embedded library does not read its
// configuration from the process environment.  Only a diagnostic build (-DZVDB_TUNING_FROM_ENV) falls back to getenv for a name
// that was never set.  Every switch is read when a call is made (no value is latched at first use).
namespace {
std::mutex g_tune_mu;
std::map<std::string, std::string> &tune_map() {
    static std::map<std::string, std::string> m;
    return m;
}
}  // namespace
const char *tuning(const char *name) {
    static thread_local std::string hold[8];
    static thread_local unsigned slot = 0;
    {
        std::lock_guard<std::mutex> g(g_tune_mu);
        auto it = tune_map().find(name);
        if (it != tune_map().end()) {
            std::string &h = hold[slot++ & 7];
            h = it->second;
            return h.c_str();
        }
    }
#ifdef ZVDB_TUNING_FROM_ENV
    return getenv(name);
#else
    return nullptr;
#endif
}

int ensure_pin(zvdb_hip_index *h, size_t bytes) {
    if (h->pin.bytes >= bytes) return ZVDB_OK;
    if (h->pin.ptr) {
        ZV_HIP(hipStreamSynchronize(h->stream));
        ZV_HIP(hipHostFree(h->pin.ptr));
        h->pin.ptr = nullptr;
        h->pin.bytes = 0;
    }
    const size_t nb = std::max<size_t>(round_up64(bytes, 4096), 1 << 16);
    ZV_HIP(hipHostMalloc(&h->pin.ptr, nb, hipHostMallocDefault));
    h->pin.bytes = nb;
    return ZVDB_OK;
}

int ensure_ins_block(zvdb_hip_index *h, size_t bytes) {
    if (h->ins_bytes >= bytes) return ZVDB_OK;
    if (h->ins_host) { ZV_HIP(hipStreamSynchronize(h->stream)); ZV_HIP(hipHostFree(h->ins_host)); h->ins_host = nullptr; h->ins_bytes = 0; }
    const size_t nb = round_up64(bytes, 4096);
    // fine-grained (coherent): the host polls a sequence word the kernel stores at system scope; with coarse-grained
    // memory that store need not be visible before the kernel ends and every call would burn its whole spin budget
    ZV_HIP(hipHostMalloc(&h->ins_host, nb, hipHostMallocMapped | hipHostMallocCoherent));
    if (!h->ins_host) { set_error("hipHostMalloc returned no block"); return ZVDB_ERR_OOM; }
    ZV_HIP(hipHostGetDevicePointer(&h->ins_dev, h->ins_host, 0));
    h->ins_bytes = nb;
    return ZVDB_OK;
}

int ensure_scratch(zvdb_hip_index *h, size_t bytes) {
    if (h->scratch_bytes >= bytes) return ZVDB_OK;
    if (h->scratch) {
        ZV_HIP(hipStreamSynchronize(h->stream));
        ZV_HIP(hipFree(h->scratch));
        h->scratch = nullptr;
        h->scratch_bytes = 0;
    }
    const size_t nb = std::max<size_t>(round_up64(bytes, 4096), 1 << 16);
    ZV_HIP(hipMalloc(&h->scratch, nb));
    h->scratch_bytes = nb;
    return ZVDB_OK;
}

template <typename T>
static int regrow(T **p, uint64_t old_elems, uint64_t new_elems, hipStream_t s) {
    T *np = nullptr;
    ZV_HIP(hipMalloc(reinterpret_cast<void **>(&np), new_elems * sizeof(T)));
    ZV_HIP(hipMemsetAsync(np, 0, new_elems * sizeof(T), s));
    if (*p && old_elems)
        ZV_HIP(hipMemcpyAsync(np, *p, old_elems * sizeof(T), hipMemcpyDeviceToDevice, s));
    ZV_HIP(hipStreamSynchronize(s));
    if (*p) ZV_HIP(hipFree(*p));
    *p = np;
    return ZVDB_OK;
}

int ensure_capacity(zvdb_hip_index *h, uint64_t want) {
    if (want <= h->cap) return ZVDB_OK;
    if (want > 0xFFFFFFF0ull) {
        set_error("more than 2^32 rows per shard are not supported (ids are u32 on device)");
        return ZVDB_ERR_UNSUPPORTED;
    }
    const uint64_t ncap = std::max<uint64_t>({want, h->cap * 2, 1024});
    const bool bf16_rows = (h->flags & ZVDB_FLAG_BF16_ROWS) != 0;  // bf16 is THE row storage: no fp32 container
    if (!bf16_rows) ZV_CHECK(regrow(&h->rows, h->count * h->ld, ncap * h->ld, h->stream));
    ZV_CHECK(regrow(&h->norm, h->count, ncap, h->stream));
    if ((h->flags & ZVDB_FLAG_BF16_MIRROR) || bf16_rows) {
        ZV_CHECK(regrow(&h->rows_h, h->count * h->ldh, ncap * h->ldh, h->stream));
        ZV_CHECK(regrow(&h->norm_h, h->count, ncap, h->stream));
    }
    ZV_CHECK(regrow(&h->adj, h->count * h->gstride, ncap * h->gstride, h->stream));
    ZV_CHECK(regrow(&h->deg, h->count, ncap, h->stream));
    ZV_CHECK(regrow(&h->lvl, h->count, ncap, h->stream));
    ZV_CHECK(regrow(&h->slot, h->count, ncap, h->stream));
    h->cap = ncap;
    return ZVDB_OK;
}

// The bulk graph's upper layers describe the index only while its rows and lists are the ones the build left: rows added
// afterwards would be unreachable from the descent, lists replaced by upload_graph / set_neighbors / build_knn_graph are not the
// ones the upper layers were selected over.  Those calls drop the flag; added rows are caught by the count.
static bool bulk_graph_valid(const zvdb_hip_index *h) { return h->bulk_graph && h->bulk_count == h->count; }

// tie_buf is shared by every compat search on this index, whatever stream it is launched on: a search waits for the
// previous one's last kernel before it touches the flags, and leaves its own marker behind
static int tie_begin(zvdb_hip_index *h, hipStream_t s) {
    if (!h->tie_event) ZV_HIP(hipEventCreateWithFlags(&h->tie_event, hipEventDisableTiming));
    if (h->tie_event_set) ZV_HIP(hipStreamWaitEvent(s, h->tie_event, 0));
    return ZVDB_OK;
}
static int tie_end(zvdb_hip_index *h, hipStream_t s) {
    ZV_HIP(hipEventRecord(h->tie_event, s));
    h->tie_event_set = true;
    return ZVDB_OK;
}

static int ensure_tie(zvdb_hip_index *h, uint64_t nq) {
    if (nq <= h->tie_cap) return ZVDB_OK;
    if (h->tie_buf) { ZV_HIP(hipDeviceSynchronize()); ZV_HIP(hipFree(h->tie_buf)); h->tie_buf = nullptr; h->tie_cap = 0; }
    const uint64_t ncap = std::max<uint64_t>(nq, 4096);
    ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->tie_buf), ncap * sizeof(uint32_t)));
    h->tie_cap = ncap;
    return ZVDB_OK;
}

// upper-layer lists (one slot per (node, layer >= 1)); new slots are zeroed (empty lists)
static int ensure_upper(zvdb_hip_index *h, uint64_t want) {
    if (want <= h->up_cap) return ZVDB_OK;
    const uint64_t ncap = std::max<uint64_t>({want, h->up_cap * 2, 1024});
    ZV_CHECK(regrow(&h->adj_up, h->up_count * h->gstride, ncap * h->gstride, h->stream));
    ZV_CHECK(regrow(&h->deg_up, h->up_count, ncap, h->stream));
    h->up_cap = ncap;
    return ZVDB_OK;
}

void timing_begin(zvdb_hip_index *h, hipStream_t s) {
    if (!h->timing) return;
    if (h->events_used == h->events.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        h->events.emplace_back(a, b);
    }
    (void)hipEventRecord(h->events[h->events_used].first, s);
}

void timing_end(zvdb_hip_index *h, hipStream_t s) {
    if (!h->timing || h->events_used >= h->events.size()) return;
    (void)hipEventRecord(h->events[h->events_used].second, s);
    h->events_used += 1;
}

static int set_dim(zvdb_hip_index *h, uint32_t dim) {
    if (dim == 0) {
        set_error("dimension 0");
        return ZVDB_ERR_INVALID;
    }
    if (h->dim == 0) {
        h->dim = dim;
        h->ld = round_up(dim, 32);
        h->ldh = round_up(dim, 64);
    } else if (h->dim != dim) {
        set_error("Mismatched dimensions in distance calculation (index %u, got %u)", h->dim, dim);
        return ZVDB_ERR_DIM_MISMATCH;
    }
    return ZVDB_OK;
}

static int use_device(zvdb_hip_index *h) {
    ZV_HIP(hipSetDevice(h->device));
    return ZVDB_OK;
}

// Every host-api call ends with its work complete, except the fused insert, which returns while the new row's norm /
// mirror kernel is still queued on h->stream.  Calls on h->stream are ordered behind it; calls that launch on a CALLER's
// stream (the *_dev entry points) settle it first.
static int settle_own_stream(zvdb_hip_index *h) {
    if (h->own_stream_busy) {
        ZV_HIP(hipStreamSynchronize(h->stream));
        h->own_stream_busy = false;
    }
    return ZVDB_OK;
}

// rows [first, first+n) are in place in h->rows: normalise (cosine), norms, bf16 mirror (fp32-row indexes)
static int finish_rows(zvdb_hip_index *h, uint64_t first, uint64_t n, hipStream_t s) {
    if (h->metric == ZVDB_METRIC_COSINE)
        ZV_CHECK(launch_normalize_rows(h->rows + first * h->ld, h->dim, h->ld, n, s));
    ZV_CHECK(launch_finalize_rows(h, first, n, s));
    return ZVDB_OK;
}

// device pointer to row `owner` as an fp32 query [dim] (distances_from, shrinkConnections' owner): the row itself for fp32
// rows, a widened copy in a small per-index buffer for bf16 rows (valid until the next call; stream order)
static int owner_query(zvdb_hip_index *h, uint64_t owner, const float **out) {
    if (!(h->flags & ZVDB_FLAG_BF16_ROWS)) { *out = h->rows + owner * h->ld; return ZVDB_OK; }
    if (!h->qrow_f32) ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->qrow_f32), (size_t)std::max(h->ldh, 64u) * sizeof(float)));
    ZV_CHECK(launch_rows_to_f32(h, owner, 1, h->qrow_f32, h->stream));
    *out = h->qrow_f32;
    return ZVDB_OK;
}

// Appends n rows at ids [count, count + n).  `fill(dst, r0, cnt)` writes rows [r0, r0 + cnt) of the incoming batch as fp32 with
// row stride h->ld (zero padded) into dst on stream s.  fp32-row indexes: dst = the rows' final place, one call.
// ZVDB_FLAG_BF16_ROWS: the rows pass through a bounded fp32 staging chunk (<= 256 MiB) -- normalised there under cosine --
// and are stored as bf16 (RNE) only; the staging chunk is the only fp32 copy that ever exists.
template <class Fill>
static int add_rows(zvdb_hip_index *h, uint64_t n, hipStream_t s, Fill fill) {
    const uint64_t first = h->count;
    if (!(h->flags & ZVDB_FLAG_BF16_ROWS)) {
        ZV_CHECK(fill(h->rows + first * h->ld, (uint64_t)0, n));
        return finish_rows(h, first, n, s);
    }
    const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(n, (256ull << 20) / ((uint64_t)h->ld * sizeof(float))));
    if (h->stage_rows < chunk) {
        if (h->stage_f32) { ZV_HIP(hipStreamSynchronize(s)); ZV_HIP(hipFree(h->stage_f32)); h->stage_f32 = nullptr; h->stage_rows = 0; }
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->stage_f32), chunk * h->ld * sizeof(float)));
        h->stage_rows = chunk;
    }
    for (uint64_t r0 = 0; r0 < n; r0 += chunk) {
        const uint64_t cnt = std::min(chunk, n - r0);
        ZV_HIP(hipMemsetAsync(h->stage_f32, 0, cnt * h->ld * sizeof(float), s));
        ZV_CHECK(fill(h->stage_f32, r0, cnt));
        if (h->metric == ZVDB_METRIC_COSINE) ZV_CHECK(launch_normalize_rows(h->stage_f32, h->dim, h->ld, cnt, s));
        ZV_CHECK(launch_store_rows_bf16(h, h->stage_f32, first + r0, cnt, s));
    }
    return ZVDB_OK;
}

static int stage_queries(zvdb_hip_index *h, const float *queries, uint32_t nq, float *d_q) {
    const size_t qb = (size_t)nq * h->dim * sizeof(float);
    ZV_CHECK(ensure_pin(h, qb));
    std::memcpy(h->pin.ptr, queries, qb);
    ZV_HIP(hipMemcpyAsync(d_q, h->pin.ptr, qb, hipMemcpyHostToDevice, h->stream));
    return ZVDB_OK;
}

}  // namespace zvdb

using namespace zvdb;

#define ZV_ARG(cond, msg)              \
    do {                               \
        if (!(cond)) {                 \
            zvdb::set_error("%s", msg); \
            return ZVDB_ERR_INVALID;   \
        }                              \
    } while (0)

extern "C" {

const char *zvdb_hip_last_error(void) { return g_err; }
int zvdb_hip_abi_version(void) { return ZVDB_HIP_ABI_VERSION; }

int zvdb_hip_device_count(int *out_count) {
    ZV_ARG(out_count, "null out_count");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *out_count = n;
    return ZVDB_OK;
}

int zvdb_hip_create(const zvdb_hip_config *cfg, zvdb_hip_index **out) {
    ZV_ARG(cfg && out, "null argument");
    ZV_ARG(cfg->abi_version == ZVDB_HIP_ABI_VERSION, "ABI version mismatch");
    ZV_ARG(cfg->metric <= ZVDB_METRIC_COSINE, "unknown metric");
    ZV_ARG(cfg->m >= 1 && cfg->m <= 4096, "m out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: libzvdb_hip has no CPU fallback");
        return ZVDB_ERR_NO_DEVICE;
    }
    int dev = cfg->device;
    if (dev < 0) ZV_HIP(hipGetDevice(&dev));
    ZV_ARG(dev < ndev, "device ordinal out of range");
    ZV_HIP(hipSetDevice(dev));
    zvdb_hip_index *h = new (std::nothrow) zvdb_hip_index();
    if (!h) return ZVDB_ERR_OOM;
    h->device = dev;
    {
        int cus = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        h->cu_count = (uint32_t)std::max(cus, 1);
    }
    h->metric = cfg->metric;
    h->m = cfg->m;
    h->ef_construction = cfg->ef_construction;
    h->gstride = cfg->graph_stride ? cfg->graph_stride : cfg->m;
    h->flags = cfg->flags;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipStreamCreate failed");
        delete h;
        return ZVDB_ERR_HIP;
    }
    if (cfg->dim) {
        int rc = set_dim(h, cfg->dim);
        if (rc == ZVDB_OK && cfg->capacity) rc = ensure_capacity(h, cfg->capacity);
        if (rc != ZVDB_OK) {
            zvdb_hip_destroy(h);
            return rc;
        }
    }
    *out = h;
    return ZVDB_OK;
}

void zvdb_hip_destroy(zvdb_hip_index *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->rows);
    (void)hipFree(h->rows_h);
    (void)hipFree(h->norm);
    (void)hipFree(h->norm_h);
    (void)hipFree(h->adj);
    (void)hipFree(h->deg);
    (void)hipFree(h->lvl);
    (void)hipFree(h->slot);
    (void)hipFree(h->adj_up);
    (void)hipFree(h->deg_up);
    (void)hipFree(h->tie_buf);
    if (h->tie_event) (void)hipEventDestroy(h->tie_event);
    (void)hipFree(h->scratch);
    (void)hipFree(h->rows_q);
    (void)hipFree(h->norm_q);
    (void)hipFree(h->rows_p);
    (void)hipFree(h->norm_p);
    if (h->guess_flag_host) (void)hipHostFree(h->guess_flag_host);
    if (h->cert_host) (void)hipHostFree(h->cert_host);
    if (h->cert_dev) (void)hipFree(h->cert_dev);
    if (h->bal_host) (void)hipHostFree(h->bal_host);
    if (h->ef_hint_host) (void)hipHostFree(h->ef_hint_host);
    (void)hipFree(h->ef_hint_dev);
    (void)hipFree(h->rows_ff);
    (void)hipFree(h->norm_ff);
    (void)hipFree(h->kstat);
    (void)hipFree(h->kstat_acc);
    (void)hipFree(h->ef_visit_buf);
    (void)hipFree(h->ef_net_buf);
    if (h->ef_event) (void)hipEventDestroy(h->ef_event);
    if (h->cert_event) (void)hipEventDestroy(h->cert_event);
    (void)hipFree(h->spec_wpos0);
    (void)hipFree(h->spec_wposU);
    if (h->ins_host) (void)hipHostFree(h->ins_host);
    (void)hipFree(h->rows_pf);
    (void)hipFree(h->rows_qf);
    (void)hipFree(h->rows_hf);
    (void)hipFree(h->stage_f32);
    (void)hipFree(h->qrow_f32);
    (void)hipFree(h->norm_hf);
    (void)hipFree(h->norm_qf);
    (void)hipFree(h->norm_pf);
    (void)hipFree(h->q_range);
    if (h->pin.ptr) (void)hipHostFree(h->pin.ptr);
    for (auto &e : h->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int zvdb_hip_count(zvdb_hip_index *h, uint64_t *out_count) {
    ZV_ARG(h && out_count, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    *out_count = h->count;
    return ZVDB_OK;
}

int zvdb_hip_dim(zvdb_hip_index *h, uint32_t *out_dim) {
    ZV_ARG(h && out_dim, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    *out_dim = h->dim;
    return ZVDB_OK;
}

int zvdb_hip_add(zvdb_hip_index *h, const float *rows, uint64_t n, uint32_t dim, uint64_t *first_id) {
    ZV_ARG(h && (rows || n == 0), "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    ZV_CHECK(set_dim(h, dim));
    if (first_id) *first_id = h->count;
    if (n == 0) return ZVDB_OK;
    ZV_CHECK(ensure_capacity(h, h->count + n));
    ZV_CHECK(add_rows(h, n, h->stream, [&](float *dst, uint64_t b0, uint64_t cnt) -> int {
        // chunked through the pinned buffer so the copy is a real async DMA
        const uint64_t chunk_rows = std::max<uint64_t>(1, (8u << 20) / (dim * sizeof(float)));
        for (uint64_t r0 = 0; r0 < cnt; r0 += chunk_rows) {
            const uint64_t nr = std::min(chunk_rows, cnt - r0);
            const size_t bytes = nr * dim * sizeof(float);
            ZV_CHECK(ensure_pin(h, bytes));
            std::memcpy(h->pin.ptr, rows + (b0 + r0) * dim, bytes);
            ZV_HIP(hipMemcpy2DAsync(dst + r0 * h->ld, h->ld * sizeof(float), h->pin.ptr, dim * sizeof(float),
                                    dim * sizeof(float), nr, hipMemcpyHostToDevice, h->stream));
            ZV_HIP(hipStreamSynchronize(h->stream));
        }
        return ZVDB_OK;
    }));
    ZV_HIP(hipStreamSynchronize(h->stream));
    h->count += n;
    return ZVDB_OK;
}

int zvdb_hip_add_dev(zvdb_hip_index *h, const void *d_rows, uint64_t n, uint32_t dim, void *stream,
                     uint64_t *first_id) {
    ZV_ARG(h && (d_rows || n == 0), "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    ZV_CHECK(settle_own_stream(h));
    ZV_CHECK(set_dim(h, dim));
    if (first_id) *first_id = h->count;
    if (n == 0) return ZVDB_OK;
    ZV_CHECK(ensure_capacity(h, h->count + n));
    hipStream_t s = static_cast<hipStream_t>(stream);
    ZV_CHECK(add_rows(h, n, s, [&](float *dst, uint64_t b0, uint64_t cnt) -> int {
        return launch_pack_rows(static_cast<const float *>(d_rows) + b0 * dim, dim, dst, h->ld, cnt, s);
    }));
    ZV_HIP(hipStreamSynchronize(s));
    h->count += n;
    return ZVDB_OK;
}

int zvdb_hip_add_synthetic(zvdb_hip_index *h, uint64_t n, uint32_t dim, uint64_t seed,
                           uint64_t first_row, uint64_t *first_id) {
    ZV_ARG(h, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    ZV_CHECK(set_dim(h, dim));
    if (first_id) *first_id = h->count;
    if (n == 0) return ZVDB_OK;
    ZV_CHECK(ensure_capacity(h, h->count + n));
    ZV_CHECK(add_rows(h, n, h->stream, [&](float *dst, uint64_t b0, uint64_t cnt) -> int {
        return launch_fill_synthetic(dst, cnt, dim, h->ld, seed, first_row + b0, h->stream);
    }));
    ZV_HIP(hipStreamSynchronize(h->stream));
    h->count += n;
    return ZVDB_OK;
}

int zvdb_hip_fill_synthetic_dev(void *d_out, uint64_t nrows, uint32_t dim, uint64_t seed,
                                uint64_t first_row, void *stream) {
    ZV_ARG(d_out || nrows == 0, "null argument");
    ZV_ARG(dim % 4 == 0, "fill_synthetic_dev needs dim % 4 == 0 (dense float4 rows)");
    return launch_fill_synthetic(static_cast<float *>(d_out), nrows, dim, dim, seed, first_row,
                                 static_cast<hipStream_t>(stream));
}

int zvdb_hip_get_vector(zvdb_hip_index *h, uint64_t id, float *out, uint32_t dim) {
    ZV_ARG(h && out, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    if (id >= h->count) {
        set_error("NodeNotFound: id %llu >= count %llu", (unsigned long long)id,
                  (unsigned long long)h->count);
        return ZVDB_ERR_NODE_NOT_FOUND;
    }
    if (dim != h->dim) {
        set_error("Mismatched dimensions (index %u, got %u)", h->dim, dim);
        return ZVDB_ERR_DIM_MISMATCH;
    }
    if (h->flags & ZVDB_FLAG_BF16_ROWS) {  // bf16 storage: widen on the host (exact)
        std::vector<uint16_t> tmp(dim);
        ZV_HIP(hipMemcpyAsync(tmp.data(), h->rows_h + id * h->ldh, dim * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream));
        ZV_HIP(hipStreamSynchronize(h->stream));
        for (uint32_t i = 0; i < dim; ++i) {
            const uint32_t u = (uint32_t)tmp[i] << 16;
            std::memcpy(out + i, &u, 4);
        }
        return ZVDB_OK;
    }
    ZV_HIP(hipMemcpyAsync(out, h->rows + id * h->ld, dim * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    ZV_HIP(hipStreamSynchronize(h->stream));
    return ZVDB_OK;
}

int zvdb_hip_set_neighbors(zvdb_hip_index *h, uint64_t node, const uint64_t *ids, uint32_t n) {
    ZV_ARG(h && (ids || n == 0), "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    if (node >= h->count) {
        set_error("NodeNotFound: node %llu", (unsigned long long)node);
        return ZVDB_ERR_NODE_NOT_FOUND;
    }
    if (n > h->gstride) {
        set_error("neighbour list of %u exceeds the device stride %u", n, h->gstride);
        return ZVDB_ERR_INVALID;
    }
    ZV_CHECK(ensure_pin(h, (h->gstride + 1) * sizeof(uint32_t)));
    uint32_t *p = static_cast<uint32_t *>(h->pin.ptr);
    for (uint32_t i = 0; i < n; ++i) {
        if (ids[i] >= h->count) {
            set_error("NodeNotFound: neighbour %llu", (unsigned long long)ids[i]);
            return ZVDB_ERR_NODE_NOT_FOUND;
        }
        p[i] = (uint32_t)ids[i];
    }
    p[h->gstride] = n;
    h->bulk_graph = false;  // (only now: a rejected call changes nothing, so it must not cost the index its descent either)
    if (n)
        ZV_HIP(hipMemcpyAsync(h->adj + node * h->gstride, p, n * sizeof(uint32_t), hipMemcpyHostToDevice,
                              h->stream));
    ZV_HIP(hipMemcpyAsync(h->deg + node, p + h->gstride, sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    ZV_HIP(hipStreamSynchronize(h->stream));
    return ZVDB_OK;
}

int zvdb_hip_upload_graph(zvdb_hip_index *h, const uint32_t *adj, const uint32_t *deg, uint64_t n,
                          uint32_t stride) {
    ZV_ARG(h && adj && deg, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    ZV_ARG(n <= h->count, "graph has more nodes than the index has rows");
    ZV_ARG(stride <= h->gstride, "graph stride exceeds the device stride");
    for (uint64_t i = 0; i < n; ++i) {
        ZV_ARG(deg[i] <= stride, "degree exceeds stride");
        for (uint32_t j = 0; j < deg[i]; ++j)
            if (adj[i * stride + j] >= h->count) {
                set_error("NodeNotFound: neighbour %u of node %llu", adj[i * stride + j],
                          (unsigned long long)i);
                return ZVDB_ERR_NODE_NOT_FOUND;
            }
    }
    h->bulk_graph = false;  // (after the validation: a rejected call leaves the index as it was)
    ef_hint_reset(h, h->stream);
    ZV_HIP(hipMemcpy2D(h->adj, h->gstride * sizeof(uint32_t), adj, stride * sizeof(uint32_t),
                       stride * sizeof(uint32_t), n, hipMemcpyHostToDevice));
    ZV_HIP(hipMemcpy(h->deg, deg, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    return ZVDB_OK;
}

int zvdb_hip_get_neighbors(zvdb_hip_index *h, uint64_t node, uint64_t *out_ids, uint32_t cap,
                           uint32_t *out_n) {
    ZV_ARG(h && out_n, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    if (node >= h->count) {
        set_error("NodeNotFound: node %llu", (unsigned long long)node);
        return ZVDB_ERR_NODE_NOT_FOUND;
    }
    std::vector<uint32_t> tmp(h->gstride + 1);
    ZV_HIP(hipMemcpy(tmp.data(), h->adj + node * h->gstride, h->gstride * sizeof(uint32_t),
                     hipMemcpyDeviceToHost));
    ZV_HIP(hipMemcpy(&tmp[h->gstride], h->deg + node, sizeof(uint32_t), hipMemcpyDeviceToHost));
    const uint32_t n = tmp[h->gstride];
    *out_n = n;
    for (uint32_t i = 0; i < n && i < cap; ++i) out_ids[i] = tmp[i];
    return ZVDB_OK;
}

// shared body of distances / distances_from
static int distances_impl(zvdb_hip_index *h, const float *host_query, uint64_t owner,
                          const uint64_t *ids, uint32_t n, float *out) {
    if (n == 0) return ZVDB_OK;
    if (!host_query && n <= 1024) {
        // insert's upper layers and shrinks call this once per step with a handful of ids: ids and distances travel through
        // host-mapped memory and completion is polled (no copy-engine transfers, no stream-synchronise wake-up)
        const size_t ib = round_up64((size_t)n * 4, 64), need2 = 2 * ib + 64;
        ZV_CHECK(ensure_ins_block(h, need2));
        unsigned char *hb = static_cast<unsigned char *>(h->ins_host), *db = static_cast<unsigned char *>(h->ins_dev);
        uint32_t *m_ids = reinterpret_cast<uint32_t *>(hb);
        for (uint32_t i = 0; i < n; ++i) {
            if (ids[i] >= h->count) {
                set_error("NodeNotFound: id %llu", (unsigned long long)ids[i]);
                return ZVDB_ERR_NODE_NOT_FOUND;
            }
            m_ids[i] = (uint32_t)ids[i];
        }
        const uint32_t seq = ++h->ins_seq ? h->ins_seq : ++h->ins_seq;
        volatile uint32_t *flag = reinterpret_cast<volatile uint32_t *>(hb + 2 * ib);
        *flag = 0;
        const float *d_owner = nullptr;
        ZV_CHECK(owner_query(h, owner, &d_owner));
        ZV_CHECK(launch_exact_distances(h, d_owner, reinterpret_cast<const uint32_t *>(db), n,
                                        reinterpret_cast<float *>(db + ib), h->stream));
        ZV_CHECK(launch_publish(nullptr, nullptr, 0, reinterpret_cast<uint32_t *>(db + 2 * ib), seq, h->stream));
        for (int spin = 0; spin < 40000 && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; ++spin) __builtin_ia32_pause();
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) ZV_HIP(hipStreamSynchronize(h->stream));
        std::memcpy(out, hb + ib, (size_t)n * 4);
        return ZVDB_OK;
    }
    const size_t qb = (size_t)h->ld * sizeof(float);
    const size_t need = qb + (size_t)n * 8;
    ZV_CHECK(ensure_scratch(h, need));
    ZV_CHECK(ensure_pin(h, need));
    unsigned char *pin = static_cast<unsigned char *>(h->pin.ptr);
    unsigned char *scr = static_cast<unsigned char *>(h->scratch);
    uint32_t *p_ids = reinterpret_cast<uint32_t *>(pin + qb);
    for (uint32_t i = 0; i < n; ++i) {
        if (ids[i] >= h->count) {
            set_error("NodeNotFound: id %llu", (unsigned long long)ids[i]);
            return ZVDB_ERR_NODE_NOT_FOUND;
        }
        p_ids[i] = (uint32_t)ids[i];
    }
    const float *d_query;
    size_t h2d_off = qb, h2d_bytes = (size_t)n * 4;
    if (host_query) {
        std::memset(pin, 0, qb);
        std::memcpy(pin, host_query, h->dim * sizeof(float));
        d_query = reinterpret_cast<const float *>(scr);
        h2d_off = 0;
        h2d_bytes += qb;
    } else {
        ZV_CHECK(owner_query(h, owner, &d_query));
    }
    ZV_HIP(hipMemcpyAsync(scr + h2d_off, pin + h2d_off, h2d_bytes, hipMemcpyHostToDevice, h->stream));
    uint32_t *d_ids = reinterpret_cast<uint32_t *>(scr + qb);
    float *d_out = reinterpret_cast<float *>(scr + qb + (size_t)n * 4);
    ZV_CHECK(launch_exact_distances(h, d_query, d_ids, n, d_out, h->stream));
    float *p_out = reinterpret_cast<float *>(pin + qb + (size_t)n * 4);
    ZV_HIP(hipMemcpyAsync(p_out, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    ZV_HIP(hipStreamSynchronize(h->stream));
    std::memcpy(out, p_out, (size_t)n * 4);
    return ZVDB_OK;
}

int zvdb_hip_distances(zvdb_hip_index *h, const float *query, uint32_t dim, const uint64_t *ids,
                       uint32_t n, float *out) {
    ZV_ARG(h && query && (ids || n == 0) && (out || n == 0), "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    if (dim != h->dim) {
        set_error("Mismatched dimensions in distance calculation (index %u, got %u)", h->dim, dim);
        return ZVDB_ERR_DIM_MISMATCH;
    }
    return distances_impl(h, query, 0, ids, n, out);
}

int zvdb_hip_distances_from(zvdb_hip_index *h, uint64_t owner, const uint64_t *ids, uint32_t n,
                            float *out) {
    ZV_ARG(h && (ids || n == 0) && (out || n == 0), "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    if (owner >= h->count) {
        set_error("NodeNotFound: owner %llu", (unsigned long long)owner);
        return ZVDB_ERR_NODE_NOT_FOUND;
    }
    return distances_impl(h, nullptr, owner, ids, n, out);
}

int zvdb_hip_greedy_layer0(zvdb_hip_index *h, uint64_t owner, uint64_t *ep, float *ep_dist,
                           uint64_t *out_evals) {
    ZV_ARG(h && ep && ep_dist, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    if (owner >= h->count || *ep >= h->count) {
        set_error("NodeNotFound: owner %llu / ep %llu", (unsigned long long)owner, (unsigned long long)*ep);
        return ZVDB_ERR_NODE_NOT_FOUND;
    }
    ZV_CHECK(ensure_scratch(h, 64));
    ZV_CHECK(ensure_pin(h, 64));
    struct Io { uint32_t ep; float dist; unsigned long long evals; };
    Io *p = static_cast<Io *>(h->pin.ptr);
    p->ep = (uint32_t)*ep;
    p->dist = *ep_dist;
    p->evals = 0;
    Io *d = static_cast<Io *>(h->scratch);
    ZV_HIP(hipMemcpyAsync(d, p, sizeof(Io), hipMemcpyHostToDevice, h->stream));
    ZV_CHECK(launch_greedy_layer0(h, (uint32_t)owner, &d->ep, &d->dist, &d->evals, h->stream));
    ZV_HIP(hipMemcpyAsync(p, d, sizeof(Io), hipMemcpyDeviceToHost, h->stream));
    ZV_HIP(hipStreamSynchronize(h->stream));
    *ep = p->ep;
    *ep_dist = p->dist;
    if (out_evals) *out_evals = p->evals;
    return ZVDB_OK;
}

// insert's device work for layer 0 in one launch and one synchronisation: append the point (hnsw.zig:77-82), then -- unless
// it is the first node -- :86, :88-104 at layer 0 and connect(id, ep, 0) (:106-108, :119-141 with :143-170) on the device
// mirror.  See k_insert_layer0.
int zvdb_hip_insert_layer0(zvdb_hip_index *h, const float *point, uint32_t dim, uint64_t entry, uint64_t *out_id,
                           uint64_t *out_ep, float *out_ep_dist, uint64_t *out_ep_list, uint32_t *out_ep_n,
                           uint32_t *out_owner_n, uint64_t *out_evals) {
    ZV_ARG(h && point, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    ZV_CHECK(set_dim(h, dim));
    const bool first = entry == ~0ull;
    if (!first && entry >= h->count) {
        set_error("NodeNotFound: entry %llu", (unsigned long long)entry);
        return ZVDB_ERR_NODE_NOT_FOUND;
    }
    ZV_ARG(first || (out_ep && out_ep_dist && out_ep_list && out_ep_n), "null argument");
    ZV_CHECK(ensure_capacity(h, h->count + 1));
    const uint64_t id = h->count;
    const size_t row_bytes = round_up64((uint64_t)dim * sizeof(float), 256), ob = insert_l0_out_bytes(h);
    ZV_CHECK(ensure_ins_block(h, row_bytes + ob));  // host-mapped [row | result] block
    std::memcpy(h->ins_host, point, (size_t)dim * sizeof(float));
    // plain fp32 L2 rows (the reference's only case): the insert kernel copies the point itself and the row's norm / mirror
    // entries are computed BEHIND it in the stream (nothing on the insert path reads them): one launch on the critical path
    const bool fast = !first && h->metric == ZVDB_METRIC_L2SQ && !(h->flags & ZVDB_FLAG_BF16_ROWS) && (size_t)dim * 4 <= 96 * 1024;
    if (!fast) {
        ZV_CHECK(add_rows(h, 1, h->stream, [&](float *dst, uint64_t, uint64_t) -> int {
            return launch_copy_row(static_cast<const float *>(h->ins_dev), dst, dim, h->stream);
        }));
    }
    unsigned char *res = static_cast<unsigned char *>(h->ins_host) + row_bytes;
    if (!first) {
        const uint32_t seq = ++h->ins_seq ? h->ins_seq : ++h->ins_seq;  // never 0 (a fresh block reads 0)
        volatile uint32_t *flag = &reinterpret_cast<InsertL0Out *>(res)->seq;
        *flag = 0;
        ZV_CHECK(launch_insert_layer0(h, (uint32_t)id, (uint32_t)entry, static_cast<unsigned char *>(h->ins_dev) + row_bytes, seq,
                                      fast ? static_cast<const float *>(h->ins_dev) : nullptr, h->stream));
        if (fast) {
            ZV_CHECK(finish_rows(h, id, 1, h->stream));
            h->own_stream_busy = true;
        }
        // the kernel's last store is the sequence number: poll it (a stream synchronise costs a ~10 us wake-up per insert);
        // bounded -- after ~200 us, or on any doubt, fall back to the synchronise, which also surfaces launch errors
        for (int spin = 0; spin < 20000 && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; ++spin) __builtin_ia32_pause();
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) ZV_HIP(hipStreamSynchronize(h->stream));
    } else {
        ZV_HIP(hipStreamSynchronize(h->stream));
    }
    h->count += 1;
    if (out_id) *out_id = id;
    if (first) return ZVDB_OK;
    const InsertL0Out *hd = reinterpret_cast<const InsertL0Out *>(res);
    if (hd->error) {
        set_error("layer-0 list exceeds the device stride %u (m = %u)", h->gstride, h->m);
        return ZVDB_ERR_INVALID;
    }
    const uint32_t *lst = reinterpret_cast<const uint32_t *>(res + sizeof(InsertL0Out));
    *out_ep = hd->ep;
    *out_ep_dist = hd->ep_dist;
    *out_ep_n = hd->ep_n;
    for (uint32_t i = 0; i < hd->ep_n; ++i) out_ep_list[i] = lst[i];
    if (out_owner_n) *out_owner_n = hd->owner_n;
    if (out_evals) *out_evals = hd->evals;
    return ZVDB_OK;
}

// n inserts' layer-0 work inside ONE launch: the wave of k_insert_layer0 runs them one after the other, so the result is
// that of n zvdb_hip_insert_layer0 calls (every insert sees the lists its predecessors left) without their launches and
// host round trips.  Plain fp32 L2 rows only (the reference's case); other configurations loop over the single call.
int zvdb_hip_insert_batch_layer0(zvdb_hip_index *h, const float *points, uint32_t dim, uint64_t n, uint64_t entry,
                                 uint64_t *out_first_id, uint64_t *out_ep, float *out_ep_dist, uint64_t *out_ep_lists,
                                 uint32_t *out_ep_n, uint32_t *out_owner_n) {
    ZV_ARG(h && (points || n == 0) && out_ep && out_ep_dist && out_ep_lists && out_ep_n, "null argument");
    if (n == 0) return ZVDB_OK;
    uint32_t m = 0;
    bool fast = false;
    {
        std::lock_guard<std::mutex> g(h->mu);
        ZV_CHECK(use_device(h));
        ZV_CHECK(set_dim(h, dim));
        m = h->m;
        if (entry >= h->count) {
            set_error("NodeNotFound: entry %llu", (unsigned long long)entry);
            return ZVDB_ERR_NODE_NOT_FOUND;
        }
        fast = h->metric == ZVDB_METRIC_L2SQ && !(h->flags & ZVDB_FLAG_BF16_ROWS) && (size_t)dim * 4 <= 96 * 1024;
        if (fast) {
            if (out_first_id) *out_first_id = h->count;
            const size_t ob = insert_l0_out_bytes(h);
            const uint64_t chunk_max = 8192;
            for (uint64_t c0 = 0; c0 < n; c0 += chunk_max) {
                const uint64_t nc = std::min<uint64_t>(chunk_max, n - c0);
                ZV_CHECK(ensure_capacity(h, h->count + nc));
                const size_t pb = round_up64(nc * dim * sizeof(float), 256), rb = nc * ob;
                ZV_CHECK(ensure_pin(h, pb + rb));
                ZV_CHECK(ensure_scratch(h, pb + rb));
                unsigned char *pin = static_cast<unsigned char *>(h->pin.ptr), *scr = static_cast<unsigned char *>(h->scratch);
                std::memcpy(pin, points + c0 * dim, nc * dim * sizeof(float));
                ZV_HIP(hipMemcpyAsync(scr, pin, nc * dim * sizeof(float), hipMemcpyHostToDevice, h->stream));
                const uint64_t id0 = h->count;
                ZV_CHECK(launch_insert_layer0(h, (uint32_t)id0, (uint32_t)entry, scr + pb, 1u, reinterpret_cast<const float *>(scr),
                                              h->stream, (uint32_t)nc, (uint32_t)ob));
                ZV_CHECK(finish_rows(h, id0, nc, h->stream));
                ZV_HIP(hipMemcpyAsync(pin + pb, scr + pb, rb, hipMemcpyDeviceToHost, h->stream));
                ZV_HIP(hipStreamSynchronize(h->stream));
                h->count += nc;
                for (uint64_t i = 0; i < nc; ++i) {
                    const InsertL0Out *hd = reinterpret_cast<const InsertL0Out *>(pin + pb + i * ob);
                    if (hd->error) {
                        set_error("layer-0 list exceeds the device stride %u (m = %u)", h->gstride, h->m);
                        return ZVDB_ERR_INVALID;
                    }
                    const uint32_t *lst = reinterpret_cast<const uint32_t *>(hd + 1);
                    out_ep[c0 + i] = hd->ep;
                    out_ep_dist[c0 + i] = hd->ep_dist;
                    out_ep_n[c0 + i] = hd->ep_n;
                    for (uint32_t j = 0; j < hd->ep_n; ++j) out_ep_lists[(c0 + i) * m + j] = lst[j];
                    if (out_owner_n) out_owner_n[c0 + i] = hd->owner_n;
                }
            }
            return ZVDB_OK;
        }
    }
    for (uint64_t i = 0; i < n; ++i) {  // (cosine / dot / bf16-valued rows: one call per insert)
        uint64_t id = 0;
        uint32_t on = 0;
        ZV_CHECK(zvdb_hip_insert_layer0(h, points + i * dim, dim, entry, &id, out_ep + i, out_ep_dist + i, out_ep_lists + i * m,
                                        out_ep_n + i, &on, nullptr));
        if (i == 0 && out_first_id) *out_first_id = id;
        if (out_owner_n) out_owner_n[i] = on;
    }
    return ZVDB_OK;
}

// The whole of insert (hnsw.zig:73-117), every layer, for n points in one launch per chunk: see k_insert_full.
int zvdb_hip_insert_batch(zvdb_hip_index *h, const float *points, uint32_t dim, uint64_t n, const uint32_t *levels,
                          uint64_t *out_first_id, uint32_t *out_nrec, uint32_t *out_recs, uint64_t recs_cap) {
    ZV_ARG(h && (points || n == 0) && (levels || n == 0) && out_nrec && out_recs, "null argument");
    if (n == 0) return ZVDB_OK;
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    ZV_CHECK(set_dim(h, dim));
    if (h->metric != ZVDB_METRIC_L2SQ || (h->flags & ZVDB_FLAG_BF16_ROWS) || (size_t)dim * 4 > 96 * 1024 || h->m > h->gstride) {
        set_error("zvdb_hip_insert_batch serves plain fp32 L2 rows with m <= graph_stride (the reference's case)");
        return ZVDB_ERR_UNSUPPORTED;
    }
    for (uint64_t i = 0; i < n; ++i) ZV_ARG(levels[i] <= 31, "level above the reference's cap of 31 (hnsw.zig:177)");
    if (out_first_id) *out_first_id = h->count;
    const uint32_t m = h->m, rw = 4 + h->gstride, ow = 4 + m;
    uint64_t done = 0, rec_out = 0;
    // chunks of more than 4 points run as speculative parallel rounds (exact_kernels.hip: k_insert_spec); ZVDB_INSERT_SPEC=0: one wave, in order
    const char *spec_env = tuning("ZVDB_INSERT_SPEC");
    // (list ids carry two flag bits: shards beyond 2^30 rows or upper-layer slots fall back to the in-order kernel)
    const bool spec = !(spec_env && spec_env[0] == '0') && m >= 1 && h->count + n < (1ull << 30) && h->up_count + 32ull * n < (1ull << 30);
    const uint32_t spec_bmax = 1024;
    const size_t spec_rsb = round_up64((size_t)spec_bmax * insert_spec_rmax() * 16, 256), spec_rcb = round_up64((size_t)spec_bmax * 8, 256);  // read counts [1024] + deferred-connect masks [1024]
    const size_t spec_extra = spec ? spec_rsb + spec_rcb + 256 : 0;
    {   // Everything that can fail for want of memory is reserved for ALL n points before the first launch, so that a
        // failure leaves the index as it was (the reference's insert leaves a consistent index on error: errdefer,
        // hnsw.zig:80): a later chunk must not fail after earlier chunks advanced count and the device lists.
        uint64_t ups = 0, recs_all = 0, rec_chunk_max = 0, rec_chunk = 0;
        uint32_t ml = h->max_level;
        for (uint64_t i = 0; i < n; ++i) {
            if (i == 0 && h->count == 0) { ml = levels[0]; ups += levels[0]; continue; }
            const uint64_t pos = i - (h->count == 0 ? 1 : 0);
            if (pos % 8192 == 0) rec_chunk = 0;
            const uint64_t r = std::min(levels[i], ml) + 1;
            recs_all += r;
            rec_chunk += r;
            rec_chunk_max = std::max(rec_chunk_max, rec_chunk);
            ups += levels[i];
            ml = std::max(ml, levels[i]);
        }
        if (recs_all > recs_cap) {
            set_error("record buffer too small: need %llu records", (unsigned long long)recs_all);
            return ZVDB_ERR_INVALID;
        }
        ZV_CHECK(ensure_capacity(h, h->count + n));
        ZV_CHECK(ensure_upper(h, h->up_count + ups));
        const uint64_t ncm = std::min<uint64_t>(8192, n);
        const size_t need = round_up64(ncm * dim * sizeof(float), 256) + round_up64(ncm * sizeof(InsertParam), 256) +
                            round_up64(rec_chunk_max * rw * 4, 256) + 64 + spec_extra;
        if (n > 4) { ZV_CHECK(ensure_pin(h, need)); ZV_CHECK(ensure_scratch(h, need)); }
        else ZV_CHECK(ensure_ins_block(h, need));
        ZV_CHECK(ensure_pin(h, (size_t)dim * 4 + 64));
    }
    if (h->count == 0) {  // the first node: nothing to scan or connect to (:110-112)
        ZV_CHECK(ensure_capacity(h, 1));
        ZV_CHECK(ensure_upper(h, levels[0]));
        ZV_CHECK(ensure_pin(h, (size_t)dim * 4 + 64));
        std::memcpy(h->pin.ptr, points, (size_t)dim * 4);
        uint32_t *w = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(h->pin.ptr) + round_up64((size_t)dim * 4, 8));
        w[0] = levels[0];
        w[1] = 0;
        ZV_HIP(hipMemcpyAsync(h->rows, h->pin.ptr, (size_t)dim * 4, hipMemcpyHostToDevice, h->stream));
        ZV_HIP(hipMemcpyAsync(h->lvl, w, 4, hipMemcpyHostToDevice, h->stream));
        ZV_HIP(hipMemcpyAsync(h->slot, w + 1, 4, hipMemcpyHostToDevice, h->stream));
        ZV_CHECK(finish_rows(h, 0, 1, h->stream));
        ZV_HIP(hipStreamSynchronize(h->stream));
        h->count = 1;
        h->max_level = levels[0];
        h->up_count = levels[0];
        out_nrec[0] = 0;
        done = 1;
    }
    const uint64_t chunk_max = 8192;
    while (done < n) {
        const uint64_t nc = std::min<uint64_t>(chunk_max, n - done);
        // host-side schedule of the chunk: slots, record offsets, max_level before each insert
        std::vector<InsertParam> prm(nc);
        uint64_t up = h->up_count, totrec = 0;
        uint32_t maxlev = h->max_level;
        for (uint64_t b = 0; b < nc; ++b) {
            const uint32_t level = levels[done + b];
            prm[b] = {level, (uint32_t)up, (uint32_t)totrec, maxlev};
            up += level;
            totrec += std::min(level, maxlev) + 1;
            maxlev = std::max(maxlev, level);
        }
        if (rec_out + totrec > recs_cap) {
            set_error("record buffer too small: need %llu records", (unsigned long long)(rec_out + totrec));
            return ZVDB_ERR_INVALID;
        }
        ZV_CHECK(ensure_capacity(h, h->count + nc));
        ZV_CHECK(ensure_upper(h, up));
        const size_t pb = round_up64(nc * dim * sizeof(float), 256), qb = round_up64(nc * sizeof(InsertParam), 256),
                     rb = round_up64(totrec * rw * 4, 256);
        const uint64_t id0 = h->count;
        const unsigned char *res = nullptr;
        bool spec_looped = false;
        uint64_t spec_dbg_rounds = 0, spec_dbg_launched = 0;
        if (nc <= 4) {  // a call or two per point: everything through host-mapped memory, completion polled
            ZV_CHECK(ensure_ins_block(h, pb + qb + rb + 64));
            unsigned char *hb = static_cast<unsigned char *>(h->ins_host), *db = static_cast<unsigned char *>(h->ins_dev);
            std::memcpy(hb, points + done * dim, nc * dim * sizeof(float));
            std::memcpy(hb + pb, prm.data(), nc * sizeof(InsertParam));
            const uint32_t seq = ++h->ins_seq ? h->ins_seq : ++h->ins_seq;
            volatile uint32_t *flag = reinterpret_cast<volatile uint32_t *>(hb + pb + qb + rb);
            *flag = 0;
            ZV_CHECK(launch_insert_full(h, (uint32_t)id0, (uint32_t)nc, reinterpret_cast<const float *>(db),
                                        reinterpret_cast<const InsertParam *>(db + pb), reinterpret_cast<uint32_t *>(db + pb + qb),
                                        reinterpret_cast<uint32_t *>(db + pb + qb + rb), seq, h->stream));
            ZV_CHECK(finish_rows(h, id0, nc, h->stream));
            h->own_stream_busy = true;
            for (int spin = 0; spin < 40000 && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; ++spin) __builtin_ia32_pause();
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) ZV_HIP(hipStreamSynchronize(h->stream));
            res = hb + pb + qb;
        } else {
            ZV_CHECK(ensure_pin(h, pb + qb + rb + spec_extra));
            ZV_CHECK(ensure_scratch(h, pb + qb + rb + spec_extra));
            unsigned char *pin = static_cast<unsigned char *>(h->pin.ptr), *scr = static_cast<unsigned char *>(h->scratch);
            std::memcpy(pin, points + done * dim, nc * dim * sizeof(float));
            std::memcpy(pin + pb, prm.data(), nc * sizeof(InsertParam));
            ZV_HIP(hipMemcpyAsync(scr, pin, pb + qb, hipMemcpyHostToDevice, h->stream));
            if (!spec) {
                ZV_CHECK(launch_insert_full(h, (uint32_t)id0, (uint32_t)nc, reinterpret_cast<const float *>(scr),
                                            reinterpret_cast<const InsertParam *>(scr + pb), reinterpret_cast<uint32_t *>(scr + pb + qb),
                                            nullptr, 0u, h->stream));
            } else {
                // per-list write stamps for every list the index can hold (sized once per call: all capacity was reserved above)
                if (h->spec_cap0 < h->cap) {
                    (void)hipFree(h->spec_wpos0);
                    h->spec_wpos0 = nullptr; h->spec_cap0 = 0;
                    ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->spec_wpos0), h->cap * 8));
                    ZV_HIP(hipMemsetAsync(h->spec_wpos0, 0xFF, h->cap * 8, h->stream));
                    h->spec_cap0 = h->cap;
                }
                if (h->spec_capU < std::max<uint64_t>(h->up_cap, 1)) {
                    (void)hipFree(h->spec_wposU);
                    h->spec_wposU = nullptr; h->spec_capU = 0;
                    const uint64_t cu = std::max<uint64_t>(h->up_cap, 1);
                    ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->spec_wposU), cu * 8));
                    ZV_HIP(hipMemsetAsync(h->spec_wposU, 0xFF, cu * 8, h->stream));
                    h->spec_capU = cu;
                }
                uint32_t *d_rset = reinterpret_cast<uint32_t *>(scr + pb + qb + rb), *d_rcnt = reinterpret_cast<uint32_t *>(scr + pb + qb + rb + spec_rsb);
                uint32_t *d_cmin = reinterpret_cast<uint32_t *>(scr + pb + qb + rb + spec_rsb + spec_rcb);
                volatile uint32_t *h_cmin = reinterpret_cast<volatile uint32_t *>(pin + pb + qb + rb);
                uint64_t spec_rounds = 0, spec_launched = 0;
                // one cooperative launch runs all rounds of the chunk (ZVDB_INSERT_SPEC=rounds: one round per three launches + a host sync)
                bool looped = false;
                if (!(spec_env && spec_env[0] == 'r')) {
                    uint32_t lb = 48;  // (measured 200k x 128: 32 -> 204k, 48 -> 257k, 64 -> 252k, 96 -> 219k inserts/s)
                    if (const char *be = tuning("ZVDB_INSERT_SPEC_B")) lb = (uint32_t)std::max(1, std::min((int)spec_bmax, atoi(be)));
                    const int rc = launch_insert_spec_loop(h, (uint32_t)id0, (uint32_t)nc, lb, reinterpret_cast<const float *>(scr),
                                                           reinterpret_cast<const InsertParam *>(scr + pb), reinterpret_cast<uint32_t *>(scr + pb + qb),
                                                           h->spec_wpos0, h->spec_wposU, d_rset, d_rcnt, d_cmin, h->spec_epoch + 1, h->stream);
                    if (rc == ZVDB_OK) {
                        looped = true;
                        ZV_HIP(hipMemcpyAsync(const_cast<uint32_t *>(h_cmin), d_cmin, 52, hipMemcpyDeviceToHost, h->stream));  // read after the sync below
                        spec_launched = lb;
                    } else {
                        (void)hipGetLastError();  // (no cooperative launch here: the round-by-round form below)
                    }
                }
                uint64_t pos = 0;
                uint32_t bcur = 64;
                while (!looped && pos < nc) {
                    const uint32_t B = (uint32_t)std::min<uint64_t>(bcur, nc - pos);
                    ZV_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_cmin), (int)B, 1, h->stream));
                    h->spec_epoch += 1;
                    ZV_CHECK(launch_insert_spec_round(h, (uint32_t)(id0 + pos), B, reinterpret_cast<const float *>(scr) + pos * dim,
                                                      reinterpret_cast<const InsertParam *>(scr + pb) + pos, reinterpret_cast<uint32_t *>(scr + pb + qb),
                                                      h->spec_wpos0, h->spec_wposU, h->spec_epoch, d_rset, d_rcnt, d_cmin, h->stream));
                    ZV_HIP(hipMemcpyAsync(const_cast<uint32_t *>(h_cmin), d_cmin, 4, hipMemcpyDeviceToHost, h->stream));
                    ZV_HIP(hipStreamSynchronize(h->stream));
                    const uint32_t c = *h_cmin;
                    if (c == 0 || c > B) { set_error("speculative insert: bad commit count %u of %u", c, B); return ZVDB_ERR_HIP; }
                    pos += c;
                    bcur = std::max(32u, std::min(spec_bmax, 2 * c));
                    spec_rounds += 1;
                    spec_launched += B;
                }
                spec_looped = looped;
                spec_dbg_rounds = spec_rounds; spec_dbg_launched = spec_launched;
            }
            ZV_CHECK(finish_rows(h, id0, nc, h->stream));
            ZV_HIP(hipMemcpyAsync(pin + pb + qb, scr + pb + qb, totrec * rw * 4, hipMemcpyDeviceToHost, h->stream));
            ZV_HIP(hipStreamSynchronize(h->stream));
            res = pin + pb + qb;
            if (spec_looped) {  // the persistent loop reports the rounds (= epochs) it used
                const volatile uint32_t *ctl_h = reinterpret_cast<const volatile uint32_t *>(pin + pb + qb + rb);
                if (ctl_h[5] != 0) { set_error("speculative insert: the round loop's grid barrier timed out"); return ZVDB_ERR_HIP; }
                const uint32_t rounds = ctl_h[4];
                h->spec_epoch += rounds;
                spec_dbg_rounds = rounds;
                spec_dbg_launched = spec_dbg_launched * rounds;
                if (tuning("ZVDB_INSERT_SPEC_DEBUG"))
                    fprintf(stderr, "[zvdb insert spec] rounds stopped by: a scanned list with two earlier writers %u, re-scanned list leads elsewhere %u, appended point closer %u, read set overflow %u; deferred connects redone %u\n",
                            ctl_h[8], ctl_h[9], ctl_h[10], ctl_h[11], ctl_h[12]);
            }
            if (spec && tuning("ZVDB_INSERT_SPEC_DEBUG"))
                fprintf(stderr, "[zvdb insert spec] chunk of %llu points at id %llu: %llu rounds%s, %llu point-slots launched (%.1f committed per round)\n",
                        (unsigned long long)nc, (unsigned long long)id0, (unsigned long long)spec_dbg_rounds, spec_looped ? " in one cooperative launch" : "",
                        (unsigned long long)spec_dbg_launched, (double)nc / (double)std::max<uint64_t>(spec_dbg_rounds, 1));
        }
        h->count += nc;
        h->up_count = up;
        h->max_level = maxlev;
        const uint32_t *rr = reinterpret_cast<const uint32_t *>(res);
        for (uint64_t b = 0; b < nc; ++b) {
            const uint32_t nrec = std::min(prm[b].level, prm[b].maxlev_pre) + 1;
            out_nrec[done + b] = nrec;
            for (uint32_t l = 0; l < nrec; ++l) {
                const uint32_t *src = rr + (size_t)(prm[b].rec_off + l) * rw;
                uint32_t *dstp = out_recs + (rec_out + prm[b].rec_off + l) * ow;
                dstp[0] = src[0]; dstp[1] = src[1]; dstp[2] = src[2]; dstp[3] = src[3];
                for (uint32_t j = 0; j < std::min(src[2], m); ++j) dstp[4 + j] = src[4 + j];
            }
        }
        rec_out += totrec;
        done += nc;
    }
    return ZVDB_OK;
}

static int check_query_args(zvdb_hip_index *h, uint32_t dim, uint32_t k) {
    if (h->count > 0 && dim != h->dim) {
        set_error("Mismatched dimensions in distance calculation (index %u, got %u)", h->dim, dim);
        return ZVDB_ERR_DIM_MISMATCH;
    }
    if (k > (1u << 20)) {
        set_error("k too large");
        return ZVDB_ERR_INVALID;
    }
    return ZVDB_OK;
}

static void fill_empty(uint32_t nq, uint32_t k, uint64_t *ids, float *dist, uint32_t *cnt) {
    for (size_t i = 0; i < (size_t)nq * k; ++i) {
        ids[i] = ~0ull;
        dist[i] = __builtin_inff();
    }
    for (uint32_t i = 0; i < nq; ++i) cnt[i] = 0;
}

int zvdb_hip_search_graph_dev(zvdb_hip_index *h, const void *d_queries, uint32_t nq, uint32_t dim,
                              uint32_t k, uint32_t ef, uint64_t entry, uint64_t id_offset,
                              void *d_ids, void *d_dist, void *d_count, void *d_stats, void *stream) {
    ZV_ARG(h && d_queries && d_ids && d_dist && d_count, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    ZV_CHECK(settle_own_stream(h));
    ZV_CHECK(check_query_args(h, dim, k));
    ZV_ARG(h->count > 0 && k > 0, "search_graph_dev needs a non-empty index and k > 0");
    const bool descend = entry == ZVDB_ENTRY_DESCEND;
    if (descend) {
        if (ef == 0 || !bulk_graph_valid(h)) {
            set_error("ZVDB_ENTRY_DESCEND needs ef > 0 and a graph built by zvdb_hip_build_graph over the index as it is now (no rows added, no lists replaced since)");
            return ZVDB_ERR_INVALID;
        }
        entry = h->bulk_top;
    }
    if (entry >= h->count) {
        set_error("NodeNotFound: entry %llu", (unsigned long long)entry);
        return ZVDB_ERR_NODE_NOT_FOUND;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ef == 0 || descend) {  // (tie_buf doubles as the per-query entry buffer of the descent)
        ZV_CHECK(ensure_tie(h, nq));
        ZV_CHECK(tie_begin(h, s));
    }
    timing_begin(h, s);
    int rc;
    if (ef == 0)
        rc = launch_search_compat(h, static_cast<const float *>(d_queries), nq, k, (uint32_t)entry,
                                  id_offset, static_cast<uint64_t *>(d_ids), static_cast<float *>(d_dist),
                                  static_cast<uint32_t *>(d_count),
                                  static_cast<unsigned long long *>(d_stats), s, h->tie_buf);
    else {
        rc = launch_search_ef(h, static_cast<const float *>(d_queries), nq, k, ef, (uint32_t)entry,
                              id_offset, static_cast<uint64_t *>(d_ids), static_cast<float *>(d_dist),
                              static_cast<uint32_t *>(d_count),
                              static_cast<unsigned long long *>(d_stats), s, descend ? h->tie_buf : nullptr, false, descend);
    }
    timing_end(h, s);
    if ((ef == 0 || descend) && rc == ZVDB_OK) ZV_CHECK(tie_end(h, s));
    return rc;
}

int zvdb_hip_search_graph_reserve(zvdb_hip_index *h, uint32_t nq, uint32_t k, uint32_t ef) {
    ZV_ARG(h, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    ZV_ARG(nq > 0 && k > 0 && h->dim > 0, "search_graph_reserve needs nq > 0, k > 0 and a dimensioned index");
    ZV_CHECK(ensure_tie(h, nq));  // compat flags / descent entries
    if (ef > 0)
        ZV_CHECK(launch_search_ef(h, nullptr, nq, k, ef, 0u, 0ull, nullptr, nullptr, nullptr, nullptr, h->stream, nullptr, true));
    return ZVDB_OK;
}

int zvdb_hip_search_graph(zvdb_hip_index *h, const float *queries, uint32_t nq, uint32_t dim,
                          uint32_t k, uint32_t ef, uint64_t entry, uint64_t *out_ids, float *out_dist,
                          uint32_t *out_count, zvdb_hip_search_stats *stats) {
    ZV_ARG(h && (queries || nq == 0) && out_ids && out_dist && out_count, "null argument");
    if (stats) stats->dist_evals = stats->pops = 0;
    if (nq == 0) return ZVDB_OK;
    {
        std::lock_guard<std::mutex> g(h->mu);
        ZV_CHECK(use_device(h));
        ZV_CHECK(check_query_args(h, dim, k));
        if (h->count == 0 || k == 0) {  // hnsw.zig:201: no entry point -> empty result, no error
            fill_empty(nq, k, out_ids, out_dist, out_count);
            return ZVDB_OK;
        }
        const size_t qb = round_up64((size_t)nq * dim * 4, 256);
        const size_t ib = round_up64((size_t)nq * k * 8, 256);
        const size_t db = round_up64((size_t)nq * k * 4, 256);
        const size_t cb = round_up64((size_t)nq * 4, 256);
        const size_t total = qb + ib + db + cb + 64;
        if (nq == 1 && ef == 0 && total <= (1u << 20)) {
            // The reference's calling pattern -- ONE query per search() (hnsw.zig:194) -- is all latency: the query and the
            // results travel through host-mapped memory (the kernel stages the query in LDS once and writes each result
            // once), no copy-engine transfers, and the host polls a sequence number published behind the search kernel
            // instead of paying a stream-synchronise wake-up.
            if (entry >= h->count) {
                set_error("NodeNotFound: entry %llu", (unsigned long long)entry);
                return ZVDB_ERR_NODE_NOT_FOUND;
            }
            ZV_CHECK(ensure_ins_block(h, total + 64));
            ZV_CHECK(ensure_scratch(h, 64));
            ZV_CHECK(ensure_tie(h, 1));
            ZV_CHECK(tie_begin(h, h->stream));
            unsigned char *hb = static_cast<unsigned char *>(h->ins_host), *dbs = static_cast<unsigned char *>(h->ins_dev);
            std::memcpy(hb, queries, (size_t)dim * 4);
            const uint32_t seq = ++h->ins_seq ? h->ins_seq : ++h->ins_seq;
            volatile uint32_t *flag = reinterpret_cast<volatile uint32_t *>(hb + total);
            *flag = 0;
            // ONE launch (exact_kernels.hip: k_search_compat_single): counters cleared, search, counters and the sequence word published
            ZV_CHECK(launch_search_compat_single(h, reinterpret_cast<const float *>(dbs), k, (uint32_t)entry, reinterpret_cast<uint64_t *>(dbs + qb),
                                                 reinterpret_cast<float *>(dbs + qb + ib), reinterpret_cast<uint32_t *>(dbs + qb + ib + db),
                                                 static_cast<unsigned long long *>(h->scratch), h->tie_buf,
                                                 reinterpret_cast<unsigned long long *>(dbs + qb + ib + db + cb), reinterpret_cast<uint32_t *>(dbs + total), seq,
                                                 h->stream));
            ZV_CHECK(tie_end(h, h->stream));
            for (int spin = 0; spin < 40000 && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; ++spin) __builtin_ia32_pause();
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) ZV_HIP(hipStreamSynchronize(h->stream));
            std::memcpy(out_ids, hb + qb, (size_t)k * 8);
            std::memcpy(out_dist, hb + qb + ib, (size_t)k * 4);
            std::memcpy(out_count, hb + qb + ib + db, 4);
            const unsigned long long *st = reinterpret_cast<const unsigned long long *>(hb + qb + ib + db + cb);
            if (stats) {
                stats->dist_evals = st[0];
                stats->pops = st[1];
            }
            return ZVDB_OK;
        }
        ZV_CHECK(ensure_scratch(h, total));
        ZV_CHECK(ensure_pin(h, total));
        unsigned char *scr = static_cast<unsigned char *>(h->scratch);
        ZV_CHECK(stage_queries(h, queries, nq, reinterpret_cast<float *>(scr)));
        ZV_HIP(hipMemsetAsync(scr + qb + ib + db + cb, 0, 64, h->stream));
        // unlock-free inner call: we already hold the mutex, so call the launchers directly
        const bool descend = entry == ZVDB_ENTRY_DESCEND;
        if (descend) {
            if (ef == 0 || !bulk_graph_valid(h)) {
                set_error("ZVDB_ENTRY_DESCEND needs ef > 0 and a graph built by zvdb_hip_build_graph over the index as it is now (no rows added, no lists replaced since)");
                return ZVDB_ERR_INVALID;
            }
            entry = h->bulk_top;
        }
        if (entry >= h->count) {
            set_error("NodeNotFound: entry %llu", (unsigned long long)entry);
            return ZVDB_ERR_NODE_NOT_FOUND;
        }
        uint64_t *d_ids = reinterpret_cast<uint64_t *>(scr + qb);
        float *d_dist = reinterpret_cast<float *>(scr + qb + ib);
        uint32_t *d_cnt = reinterpret_cast<uint32_t *>(scr + qb + ib + db);
        unsigned long long *d_st = reinterpret_cast<unsigned long long *>(scr + qb + ib + db + cb);
        if (ef == 0 || descend) {
            ZV_CHECK(ensure_tie(h, nq));
            ZV_CHECK(tie_begin(h, h->stream));
        }
        int rc = ef == 0 ? launch_search_compat(h, reinterpret_cast<float *>(scr), nq, k, (uint32_t)entry, 0,
                                                d_ids, d_dist, d_cnt, d_st, h->stream, h->tie_buf)
                         : launch_search_ef(h, reinterpret_cast<float *>(scr), nq, k, ef, (uint32_t)entry,
                                            0, d_ids, d_dist, d_cnt, d_st, h->stream, descend ? h->tie_buf : nullptr, false, descend);
        ZV_CHECK(rc);
        if (ef == 0 || descend) ZV_CHECK(tie_end(h, h->stream));
        unsigned char *pin = static_cast<unsigned char *>(h->pin.ptr);
        ZV_HIP(hipMemcpyAsync(pin + qb, scr + qb, ib + db + cb + 64, hipMemcpyDeviceToHost, h->stream));
        ZV_HIP(hipStreamSynchronize(h->stream));
        std::memcpy(out_ids, pin + qb, (size_t)nq * k * 8);
        std::memcpy(out_dist, pin + qb + ib, (size_t)nq * k * 4);
        std::memcpy(out_count, pin + qb + ib + db, (size_t)nq * 4);
        const unsigned long long *st = reinterpret_cast<const unsigned long long *>(pin + qb + ib + db + cb);
        if (stats) {
            stats->dist_evals = st[0];
            stats->pops = st[1];
        }
        if (st[2] != 0) {
            set_error("ef-mode visited set exhausted for %llu queries (more nodes visited than the table sized from ef * stride holds)", st[2]);
            return ZVDB_ERR_UNSUPPORTED;
        }
    }
    return ZVDB_OK;
}

int zvdb_hip_bruteforce_workspace(zvdb_hip_index *h, uint32_t nq, uint32_t k, uint32_t mode,
                                  uint64_t *out_bytes) {
    ZV_ARG(h && out_bytes, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_ARG(mode <= ZVDB_BF_FP4, "unknown brute-force mode");
    *out_bytes = bruteforce_workspace_bytes(h, nq, k, mode);
    return ZVDB_OK;
}

int zvdb_hip_search_bruteforce_dev(zvdb_hip_index *h, const void *d_queries, uint32_t nq, uint32_t dim,
                                   uint32_t k, uint32_t mode, uint64_t id_offset, void *d_workspace,
                                   uint64_t workspace_bytes, void *d_ids, void *d_dist, void *d_count,
                                   void *stream) {
    ZV_ARG(h && d_queries && d_ids && d_dist && d_count && d_workspace, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    ZV_CHECK(settle_own_stream(h));
    ZV_CHECK(check_query_args(h, dim, k));
    ZV_ARG(mode <= ZVDB_BF_FP4, "unknown brute-force mode");
    ZV_ARG(h->count > 0 && k > 0, "search_bruteforce_dev needs a non-empty index and k > 0");
    return launch_bruteforce(h, static_cast<const float *>(d_queries), nq, k, mode, id_offset, d_workspace,
                             workspace_bytes, static_cast<uint64_t *>(d_ids), static_cast<float *>(d_dist),
                             static_cast<uint32_t *>(d_count), static_cast<hipStream_t>(stream));
}

int zvdb_hip_search_bruteforce(zvdb_hip_index *h, const float *queries, uint32_t nq, uint32_t dim,
                               uint32_t k, uint32_t mode, uint64_t *out_ids, float *out_dist,
                               uint32_t *out_count) {
    ZV_ARG(h && (queries || nq == 0) && out_ids && out_dist && out_count, "null argument");
    if (nq == 0) return ZVDB_OK;
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    ZV_CHECK(check_query_args(h, dim, k));
    ZV_ARG(mode <= ZVDB_BF_FP4, "unknown brute-force mode");
    if (h->count == 0 || k == 0) {
        fill_empty(nq, k, out_ids, out_dist, out_count);
        return ZVDB_OK;
    }
    const size_t qb = round_up64((size_t)nq * dim * 4, 256);
    const size_t ib = round_up64((size_t)nq * k * 8, 256);
    const size_t db = round_up64((size_t)nq * k * 4, 256);
    const size_t cb = round_up64((size_t)nq * 4, 256);
    const uint64_t wsb = bruteforce_workspace_bytes(h, nq, k, mode);
    const size_t total = qb + ib + db + cb + wsb;
    ZV_CHECK(ensure_scratch(h, total));
    ZV_CHECK(ensure_pin(h, qb + ib + db + cb));
    unsigned char *scr = static_cast<unsigned char *>(h->scratch);
    ZV_CHECK(stage_queries(h, queries, nq, reinterpret_cast<float *>(scr)));
    uint64_t *d_ids = reinterpret_cast<uint64_t *>(scr + qb);
    float *d_dist = reinterpret_cast<float *>(scr + qb + ib);
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(scr + qb + ib + db);
    ZV_CHECK(launch_bruteforce(h, reinterpret_cast<float *>(scr), nq, k, mode, 0, scr + qb + ib + db + cb, wsb,
                               d_ids, d_dist, d_cnt, h->stream));
    unsigned char *pin = static_cast<unsigned char *>(h->pin.ptr);
    ZV_HIP(hipMemcpyAsync(pin + qb, scr + qb, ib + db + cb, hipMemcpyDeviceToHost, h->stream));
    ZV_HIP(hipStreamSynchronize(h->stream));
    std::memcpy(out_ids, pin + qb, (size_t)nq * k * 8);
    std::memcpy(out_dist, pin + qb + ib, (size_t)nq * k * 4);
    std::memcpy(out_count, pin + qb + ib + db, (size_t)nq * 4);
    return ZVDB_OK;
}

int zvdb_hip_build_knn_graph(zvdb_hip_index *h, uint32_t degree, uint32_t mode) {
    ZV_ARG(h, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    ZV_ARG(mode <= ZVDB_BF_FP4, "unknown brute-force mode");
    ZV_ARG(degree >= 1 && degree <= h->gstride, "degree must be in [1, graph_stride]");
    if (h->count == 0) return ZVDB_OK;
    h->bulk_graph = false;
    ef_hint_reset(h, h->stream);
    const uint32_t k = degree + 1;  // the row itself is (one of) its own nearest
    const uint32_t B = 1024;        // rows per search batch
    CertPause no_cert(h);           // (common.h: the builders' kNN passes run uncertified)
    const size_t ib = round_up64((size_t)B * k * 8, 256), db = round_up64((size_t)B * k * 4, 256);
    const size_t cb = round_up64((size_t)B * 4, 256), qb = round_up64((size_t)B * h->dim * 4, 256);
    const uint64_t wsb = bruteforce_workspace_bytes(h, B, k, mode);
    ZV_CHECK(ensure_scratch(h, qb + ib + db + cb + wsb));
    unsigned char *scr = static_cast<unsigned char *>(h->scratch);
    float *d_q = reinterpret_cast<float *>(scr);
    uint64_t *d_ids = reinterpret_cast<uint64_t *>(scr + qb);
    float *d_dist = reinterpret_cast<float *>(scr + qb + ib);
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(scr + qb + ib + db);
    for (uint64_t r0 = 0; r0 < h->count; r0 += B) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(B, h->count - r0);
        // queries = rows [r0, r0+n) as dense [n][dim]
        ZV_CHECK(launch_rows_to_f32(h, r0, n, d_q, h->stream));
        ZV_CHECK(launch_bruteforce(h, d_q, n, k, mode, 0, scr + qb + ib + db + cb, wsb, d_ids, d_dist, d_cnt, h->stream));
        ZV_CHECK(launch_results_to_adj(h, d_ids, d_cnt, k, r0, n, degree, h->stream));
    }
    ZV_HIP(hipStreamSynchronize(h->stream));
    return ZVDB_OK;
}

int zvdb_hip_build_graph(zvdb_hip_index *h, uint32_t M, uint32_t ef_construction, uint32_t mode, uint64_t level_seed) {
    ZV_ARG(h, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    ZV_CHECK(settle_own_stream(h));
    ZV_ARG(mode <= ZVDB_BF_FP4, "unknown brute-force mode");
    ZV_ARG(M >= 1 && M <= 32 && 2 * M <= h->gstride, "M must be in [1, 32] and 2 M <= graph_stride (layer 0 keeps up to 2 M links)");
    ZV_ARG(ef_construction >= 1, "ef_construction must be >= 1");
    if (h->count > 0xFFFFFFF0ull) { set_error("shard too large"); return ZVDB_ERR_UNSUPPORTED; }
    if (h->count == 0) return ZVDB_OK;
    ef_hint_reset(h, h->stream);
    if (h->count == 1) {
        ZV_HIP(hipMemsetAsync(h->deg, 0, 4, h->stream));
        ZV_HIP(hipStreamSynchronize(h->stream));
    }
    // The build writes levels, slots and lists in place (a second copy of a 10M-row graph is what it cannot afford): a build that
    // fails midway (e.g. out of memory in an upper-layer sub-index) must not leave half of the new graph under the old header.
    // It leaves the index WITHOUT a graph: empty lists on every layer, no levels -- rows untouched, every search well defined.
    h->bulk_graph = false;
    const int rc = build_bulk_graph(h, M, ef_construction, mode, level_seed, ensure_upper);
    if (rc != ZVDB_OK) {
        const std::string why = zvdb_hip_last_error();
        (void)hipStreamSynchronize(h->stream);
        (void)hipMemsetAsync(h->deg, 0, h->count * 4, h->stream);
        if (h->lvl) (void)hipMemsetAsync(h->lvl, 0, h->count * 4, h->stream);
        if (h->slot) (void)hipMemsetAsync(h->slot, 0, h->count * 4, h->stream);
        (void)hipStreamSynchronize(h->stream);
        h->up_count = 0;
        h->max_level = 0;
        set_error("%s (the index is left without a graph)", why.c_str());
    }
    return rc;
}


// ---- on-disk index format (absent in the reference, which is memory-only) -------------------------------
// One flat little-endian file: a 256-byte header, then the device arrays exactly as they sit in HBM, each 256-byte aligned and
// in this order -- rows (fp32 [count][ld], or bf16 [count][ldh] with ZVDB_FLAG_BF16_ROWS), norm [count], adj [count][gstride],
// deg [count], lvl [count], slot [count], adj_up [up_count][gstride], deg_up [up_count].  Lazily built mirrors are NOT stored
// (they are functions of the rows and rebuild on first use); the optional row-major bf16 mirror is recomputed on load.
// The layout is the upload layout: a loader may mmap the file and hipMemcpy each section as is.
namespace {
struct ZvdbFileHeader {
    char magic[8];  // "ZVDBHIP1"
    uint32_t version, dim, ld, ldh, metric, m, ef_construction, gstride, flags, max_level, bulk_graph, bulk_top;
    uint64_t count, up_count;
    uint64_t off_rows, off_norm, off_adj, off_deg, off_lvl, off_slot, off_adj_up, off_deg_up, file_bytes;
    unsigned char pad[256 - 8 - 12 * 4 - 11 * 8];
};
static_assert(sizeof(ZvdbFileHeader) == 256, "header size");

int copy_section_out(zvdb_hip_index *h, FILE *f, const void *dev, uint64_t bytes, uint64_t off) {
    if (fseeko(f, (off_t)off, SEEK_SET) != 0) { set_error("seek failed"); return ZVDB_ERR_INVALID; }
    const uint64_t chunk = 64ull << 20;
    ZV_CHECK(ensure_pin(h, (size_t)std::min<uint64_t>(chunk, std::max<uint64_t>(bytes, 1))));
    for (uint64_t o = 0; o < bytes; o += chunk) {
        const uint64_t nb = std::min(chunk, bytes - o);
        ZV_HIP(hipMemcpyAsync(h->pin.ptr, static_cast<const unsigned char *>(dev) + o, nb, hipMemcpyDeviceToHost, h->stream));
        ZV_HIP(hipStreamSynchronize(h->stream));
        if (fwrite(h->pin.ptr, 1, nb, f) != nb) { set_error("short write"); return ZVDB_ERR_INVALID; }
    }
    return ZVDB_OK;
}
int copy_section_in(zvdb_hip_index *h, FILE *f, void *dev, uint64_t bytes, uint64_t off) {
    if (fseeko(f, (off_t)off, SEEK_SET) != 0) { set_error("seek failed"); return ZVDB_ERR_INVALID; }
    const uint64_t chunk = 64ull << 20;
    ZV_CHECK(ensure_pin(h, (size_t)std::min<uint64_t>(chunk, std::max<uint64_t>(bytes, 1))));
    for (uint64_t o = 0; o < bytes; o += chunk) {
        const uint64_t nb = std::min(chunk, bytes - o);
        if (fread(h->pin.ptr, 1, nb, f) != nb) { set_error("short read: truncated index file"); return ZVDB_ERR_INVALID; }
        ZV_HIP(hipMemcpyAsync(static_cast<unsigned char *>(dev) + o, h->pin.ptr, nb, hipMemcpyHostToDevice, h->stream));
        ZV_HIP(hipStreamSynchronize(h->stream));
    }
    return ZVDB_OK;
}
}  // namespace

int zvdb_hip_save(zvdb_hip_index *h, const char *path) {
    ZV_ARG(h && path, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    ZV_CHECK(settle_own_stream(h));
    ZV_HIP(hipStreamSynchronize(h->stream));
    const bool hb = (h->flags & ZVDB_FLAG_BF16_ROWS) != 0;
    ZvdbFileHeader hd{};
    std::memcpy(hd.magic, "ZVDBHIP1", 8);
    hd.version = 1; hd.dim = h->dim; hd.ld = h->ld; hd.ldh = h->ldh; hd.metric = h->metric; hd.m = h->m;
    hd.ef_construction = h->ef_construction; hd.gstride = h->gstride; hd.flags = h->flags; hd.max_level = h->max_level;
    hd.bulk_graph = bulk_graph_valid(h) ? 1u : 0u;  /* (a bulk graph that no longer covers every row is saved as what it is: lists without a descent) */ hd.bulk_top = h->bulk_top; hd.count = h->count; hd.up_count = h->up_count;
    const uint64_t n = h->count, rowb = hb ? (uint64_t)h->ldh * 2 : (uint64_t)h->ld * 4;
    uint64_t off = 256;
    auto place = [&](uint64_t bytes) { const uint64_t o = off; off = round_up64(off + bytes, 256); return o; };
    hd.off_rows = place(n * rowb); hd.off_norm = place(n * 4); hd.off_adj = place(n * h->gstride * 4); hd.off_deg = place(n * 4);
    hd.off_lvl = place(n * 4); hd.off_slot = place(n * 4); hd.off_adj_up = place(h->up_count * h->gstride * 4); hd.off_deg_up = place(h->up_count * 4);
    hd.file_bytes = off;
    FILE *f = fopen(path, "wb");
    if (!f) { set_error("cannot open %s for writing", path); return ZVDB_ERR_INVALID; }
    int rc = ZVDB_OK;
    if (fwrite(&hd, 1, sizeof(hd), f) != sizeof(hd)) { set_error("short write"); rc = ZVDB_ERR_INVALID; }
    if (rc == ZVDB_OK && n) {
        rc = copy_section_out(h, f, hb ? (const void *)h->rows_h : (const void *)h->rows, n * rowb, hd.off_rows);
        if (rc == ZVDB_OK) rc = copy_section_out(h, f, h->norm, n * 4, hd.off_norm);
        if (rc == ZVDB_OK) rc = copy_section_out(h, f, h->adj, n * h->gstride * 4, hd.off_adj);
        if (rc == ZVDB_OK) rc = copy_section_out(h, f, h->deg, n * 4, hd.off_deg);
        if (rc == ZVDB_OK) rc = copy_section_out(h, f, h->lvl, n * 4, hd.off_lvl);
        if (rc == ZVDB_OK) rc = copy_section_out(h, f, h->slot, n * 4, hd.off_slot);
        if (rc == ZVDB_OK && h->up_count) rc = copy_section_out(h, f, h->adj_up, h->up_count * h->gstride * 4, hd.off_adj_up);
        if (rc == ZVDB_OK && h->up_count) rc = copy_section_out(h, f, h->deg_up, h->up_count * 4, hd.off_deg_up);
    }
    if (rc == ZVDB_OK && ftruncate(fileno(f), (off_t)hd.file_bytes) != 0) { set_error("cannot size %s", path); rc = ZVDB_ERR_INVALID; }
    if (fclose(f) != 0 && rc == ZVDB_OK) { set_error("close failed"); rc = ZVDB_ERR_INVALID; }
    return rc;
}

int zvdb_hip_load(const char *path, int32_t device, zvdb_hip_index **out) {
    ZV_ARG(path && out, "null argument");
    FILE *f = fopen(path, "rb");
    if (!f) { set_error("cannot open %s", path); return ZVDB_ERR_INVALID; }
    ZvdbFileHeader hd{};
    if (fread(&hd, 1, sizeof(hd), f) != sizeof(hd) || std::memcmp(hd.magic, "ZVDBHIP1", 8) != 0 || hd.version != 1) {
        fclose(f);
        set_error("%s is not a zvdb-hip index file (version 1)", path);
        return ZVDB_ERR_INVALID;
    }
    // The file is input from outside: nothing in it is trusted.  (1) Header fields against their domains; (2) the section table
    // against the ONE layout the fields imply (the writer's arithmetic, overflow-checked) and against the file's real size;
    // (3) after the upload, one kernel over every index the graph kernels follow without a bounds test (k_validate_graph).
    {
        auto bad = [&](const char *what) { fclose(f); set_error("%s: corrupt index file (%s)", path, what); return ZVDB_ERR_INVALID; };
        if (hd.flags & ~(ZVDB_FLAG_BF16_MIRROR | ZVDB_FLAG_BF16_ROWS)) return bad("unknown flags");
        if (hd.metric > ZVDB_METRIC_COSINE) return bad("metric");
        if (hd.dim > (1u << 20) || hd.ld > (1u << 21) || hd.ldh > (1u << 21)) return bad("dimension");
        if (hd.gstride == 0 || hd.gstride > 4096 || hd.m == 0 || hd.m > 4096) return bad("m / graph stride");  // (zvdb_hip_create's domains; m may exceed the stride)
        if (hd.count > 0xFFFFFFF0ull || hd.up_count > 0xFFFFFFF0ull) return bad("row count");
        if (hd.count == 0 && hd.up_count != 0) return bad("upper layers without rows");
        if (hd.count && hd.dim == 0) return bad("rows without a dimension");
        if (hd.max_level > 31) return bad("max_level");  // hnsw.zig:172-180 caps the level at 31
        if (hd.bulk_graph > 1 || (hd.count ? hd.bulk_top >= hd.count : hd.bulk_top != 0)) return bad("entry node");
        const bool hbf = (hd.flags & ZVDB_FLAG_BF16_ROWS) != 0;
        const unsigned __int128 lim = (unsigned __int128)1 << 62;
        unsigned __int128 off = 256;
        bool ok = true;
        auto place = [&](unsigned __int128 bytes, uint64_t got) {
            if (bytes > lim || off > lim) { ok = false; return; }
            if ((uint64_t)off != got) ok = false;
            off = (off + bytes + 255) / 256 * 256;
        };
        const unsigned __int128 n = hd.count, up = hd.up_count, rowb = hbf ? (unsigned __int128)hd.ldh * 2 : (unsigned __int128)hd.ld * 4;
        place(n * rowb, hd.off_rows); place(n * 4, hd.off_norm); place(n * hd.gstride * 4, hd.off_adj); place(n * 4, hd.off_deg);
        place(n * 4, hd.off_lvl); place(n * 4, hd.off_slot); place(up * hd.gstride * 4, hd.off_adj_up); place(up * 4, hd.off_deg_up);
        if (!ok || off > lim || (uint64_t)off != hd.file_bytes) return bad("section table");
        struct stat st{};
        if (fstat(fileno(f), &st) != 0 || (uint64_t)st.st_size != hd.file_bytes) return bad("file size differs from the header's");
    }
    zvdb_hip_config cfg{};
    cfg.abi_version = ZVDB_HIP_ABI_VERSION; cfg.dim = hd.dim; cfg.metric = hd.metric; cfg.m = hd.m; cfg.ef_construction = hd.ef_construction;
    cfg.device = device; cfg.flags = hd.flags; cfg.graph_stride = hd.gstride; cfg.capacity = hd.count;
    zvdb_hip_index *h = nullptr;
    int rc = zvdb_hip_create(&cfg, &h);
    if (rc != ZVDB_OK) { fclose(f); return rc; }
    {
        std::lock_guard<std::mutex> g(h->mu);
        const bool hb = (h->flags & ZVDB_FLAG_BF16_ROWS) != 0;
        const uint64_t n = hd.count, rowb = hb ? (uint64_t)h->ldh * 2 : (uint64_t)h->ld * 4;
        if ((hd.dim && (hd.ld != h->ld || hd.ldh != h->ldh)) || hd.gstride != h->gstride) { set_error("index file geometry mismatch"); rc = ZVDB_ERR_INVALID; }
        if (rc == ZVDB_OK && n) {
            rc = ensure_capacity(h, n);
            if (rc == ZVDB_OK && hd.up_count) rc = ensure_upper(h, hd.up_count);
            if (rc == ZVDB_OK) rc = copy_section_in(h, f, hb ? (void *)h->rows_h : (void *)h->rows, n * rowb, hd.off_rows);
            if (rc == ZVDB_OK) rc = copy_section_in(h, f, h->norm, n * 4, hd.off_norm);
            if (rc == ZVDB_OK) rc = copy_section_in(h, f, h->adj, n * h->gstride * 4, hd.off_adj);
            if (rc == ZVDB_OK) rc = copy_section_in(h, f, h->deg, n * 4, hd.off_deg);
            if (rc == ZVDB_OK) rc = copy_section_in(h, f, h->lvl, n * 4, hd.off_lvl);
            if (rc == ZVDB_OK) rc = copy_section_in(h, f, h->slot, n * 4, hd.off_slot);
            if (rc == ZVDB_OK && hd.up_count) rc = copy_section_in(h, f, h->adj_up, hd.up_count * h->gstride * 4, hd.off_adj_up);
            if (rc == ZVDB_OK && hd.up_count) rc = copy_section_in(h, f, h->deg_up, hd.up_count * 4, hd.off_deg_up);
            // the rows' padding [dim, ld) is zeroed (the exact kernels walk whole chunks and rely on it) and every norm is recomputed
            // from the rows: neither is taken from the file on trust
            if (rc == ZVDB_OK) rc = launch_sanitize_loaded_rows(h, n, h->stream);
            if (rc == ZVDB_OK && !hb) rc = launch_finalize_rows(h, 0, n, h->stream);  // norms + the optional row-major bf16 mirror, from the rows
            if (rc == ZVDB_OK) {  // (3) the graph's indices
                uint32_t *d_bad = nullptr, bad = 0;
                if (hipMalloc(reinterpret_cast<void **>(&d_bad), 4) != hipSuccess) rc = ZVDB_ERR_OOM;
                if (rc == ZVDB_OK && hipMemsetAsync(d_bad, 0, 4, h->stream) != hipSuccess) rc = ZVDB_ERR_HIP;
                if (rc == ZVDB_OK) rc = launch_validate_graph(h, n, hd.up_count, hd.max_level, d_bad, h->stream);
                if (rc == ZVDB_OK && hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = ZVDB_ERR_HIP;
                if (rc == ZVDB_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = ZVDB_ERR_HIP;
                (void)hipFree(d_bad);
                if (rc == ZVDB_OK && bad) {
                    set_error("%s: corrupt index file (graph:%s%s%s%s)", path, bad & 1 ? " degree > stride" : "", bad & 2 ? " neighbour id out of range" : "",
                              bad & 4 ? " level > max_level" : "", bad & 8 ? " upper-layer slot out of range" : "");
                    rc = ZVDB_ERR_INVALID;
                }
            }
            if (rc == ZVDB_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = ZVDB_ERR_HIP;
        }
        if (rc == ZVDB_OK) {
            h->count = n; h->up_count = hd.up_count; h->max_level = hd.max_level; h->bulk_graph = hd.bulk_graph != 0; h->bulk_top = hd.bulk_top; h->bulk_count = n;
        }
    }
    fclose(f);
    if (rc != ZVDB_OK) { zvdb_hip_destroy(h); return rc; }
    *out = h;
    return ZVDB_OK;
}

int zvdb_hip_storage_info(zvdb_hip_index *h, uint64_t *out_row_bytes, uint64_t *out_resident_bytes) {
    ZV_ARG(h, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    const bool hb = (h->flags & ZVDB_FLAG_BF16_ROWS) != 0;
    if (out_row_bytes) *out_row_bytes = hb ? (uint64_t)h->ldh * 2 : (uint64_t)h->ld * 4;
    if (out_resident_bytes) {
        uint64_t b = 0;
        if (h->rows) b += h->cap * h->ld * 4;
        if (h->rows_h) b += h->cap * h->ldh * 2;
        if (h->norm) b += h->cap * 4;
        if (h->norm_h) b += h->cap * 4;
        if (h->adj) b += h->cap * h->gstride * 4 + h->cap * 4;
        if (h->lvl) b += h->cap * 8;
        if (h->adj_up) b += h->up_cap * h->gstride * 4 + h->up_cap * 4;
        if (h->rows_q) b += h->cap_q * h->ldq + h->cap_q * 4;
        if (h->rows_qf) b += h->cap_qf * i8s_ksteps(h->dim) * 32 + h->cap_qf * 4;
        if (h->rows_p) b += h->cap_p * h->ldp + h->cap_p * 4;
        if (h->rows_pf) b += h->cap_pf * fp4s_ksteps(h->dim) * 32 + h->cap_pf * 4;
        if (h->rows_hf) b += h->cap_hf * bf16s_ksteps(h->dim) * 32 + (h->cap_hf + 64) * 4;
        if (h->rows_ff) b += h->cap_ff * f32s_ksteps(h->dim) * 32 + h->cap_ff * 4;
        if (h->kstat) { const uint64_t D = f32s_ksteps(h->dim) * 8; b += (2 + 2 * D + D * D) * 4 + (3 + 2 * D + D * D) * 8; }  // (moments behind k_rq_guess)
        b += h->ef_visit_bytes + (h->ef_net_list_words + h->ef_net_table_words) * 4 + (h->spec_cap0 + h->spec_capU) * 8;  // ef-search net / global visited tables, speculative-insert stamps
        *out_resident_bytes = b;
    }
    return ZVDB_OK;
}

int zvdb_hip_graph_info(zvdb_hip_index *h, uint64_t *out_entry, uint32_t *out_max_level) {
    ZV_ARG(h, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    if (out_entry) *out_entry = bulk_graph_valid(h) ? h->bulk_top : 0;
    if (out_max_level) *out_max_level = h->max_level;
    return ZVDB_OK;
}

int zvdb_hip_get_neighbors_layer(zvdb_hip_index *h, uint64_t node, uint32_t layer, uint64_t *out_ids, uint32_t cap, uint32_t *out_n) {
    ZV_ARG(h && out_n, "null argument");
    if (layer == 0) return zvdb_hip_get_neighbors(h, node, out_ids, cap, out_n);
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    if (node >= h->count || !h->lvl) { set_error("NodeNotFound: %llu", (unsigned long long)node); return ZVDB_ERR_NODE_NOT_FOUND; }
    ZV_HIP(hipStreamSynchronize(h->stream));
    uint32_t lv = 0, sl = 0;
    ZV_HIP(hipMemcpy(&lv, h->lvl + node, 4, hipMemcpyDeviceToHost));
    if (layer > lv) { set_error("node %llu has no layer %u", (unsigned long long)node, layer); return ZVDB_ERR_NODE_NOT_FOUND; }
    ZV_HIP(hipMemcpy(&sl, h->slot + node, 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> tmp(h->gstride);
    uint32_t n = 0;
    const size_t row = (size_t)sl + layer - 1;
    ZV_HIP(hipMemcpy(&n, h->deg_up + row, 4, hipMemcpyDeviceToHost));
    n = std::min(n, h->gstride);
    ZV_HIP(hipMemcpy(tmp.data(), h->adj_up + row * h->gstride, (size_t)n * 4, hipMemcpyDeviceToHost));
    *out_n = n;
    for (uint32_t i = 0; i < n && i < cap; ++i) out_ids[i] = tmp[i];
    return ZVDB_OK;
}

int zvdb_hip_merge_topk_dev(const void *d_ids, const void *d_dist, const void *d_count, uint32_t G,
                            uint32_t nq, uint32_t k, void *d_out_ids, void *d_out_dist,
                            void *d_out_count, void *stream) {
    ZV_ARG(d_ids && d_dist && d_count && d_out_ids && d_out_dist && d_out_count, "null argument");
    ZV_ARG(G >= 1, "G must be >= 1");
    return launch_merge_topk(static_cast<const uint64_t *>(d_ids), static_cast<const float *>(d_dist),
                             static_cast<const uint32_t *>(d_count), G, nq, k,
                             static_cast<uint64_t *>(d_out_ids), static_cast<float *>(d_out_dist),
                             static_cast<uint32_t *>(d_out_count), static_cast<hipStream_t>(stream));
}

int zvdb_hip_merge_topk_packed_dev(const void *d_packed, uint64_t rank_stride_bytes, uint32_t G, uint32_t nq,
                                   uint32_t k, void *d_out_ids, void *d_out_dist, void *d_out_count, void *stream) {
    ZV_ARG(d_packed && d_out_ids && d_out_dist && d_out_count, "null argument");
    ZV_ARG(G >= 1, "G must be >= 1");
    ZV_ARG(rank_stride_bytes >= (uint64_t)nq * k * 12 + (uint64_t)nq * 4 && rank_stride_bytes % 8 == 0,
           "rank stride too small / misaligned");
    const unsigned char *b = static_cast<const unsigned char *>(d_packed);
    return launch_merge_topk_strided(reinterpret_cast<const uint64_t *>(b), rank_stride_bytes,
                                     reinterpret_cast<const float *>(b + (size_t)nq * k * 8), rank_stride_bytes,
                                     reinterpret_cast<const uint32_t *>(b + (size_t)nq * k * 12), rank_stride_bytes, G,
                                     nq, k, static_cast<uint64_t *>(d_out_ids), static_cast<float *>(d_out_dist),
                                     static_cast<uint32_t *>(d_out_count), static_cast<hipStream_t>(stream));
}

int zvdb_hip_index_set_screen_survivors(zvdb_hip_index *h, uint32_t survivors) {
    ZV_ARG(h, "null argument");
    ZV_ARG(survivors == 0 || survivors == 64 || survivors == 128 || survivors == 256, "survivors must be 0, 64, 128 or 256");
    std::lock_guard<std::mutex> g(h->mu);
    h->screen_survivors = survivors;
    return ZVDB_OK;
}

int zvdb_hip_set_tuning(const char *name, const char *value) {
    if (!name || strncmp(name, "ZVDB_", 5) != 0) { set_error("tuning names start with ZVDB_"); return ZVDB_ERR_INVALID; }
    std::lock_guard<std::mutex> g(g_tune_mu);
    if (value) tune_map()[name] = value;
    else tune_map().erase(name);
    return ZVDB_OK;
}

const char *zvdb_hip_get_tuning(const char *name) { return name ? tuning(name) : nullptr; }

int zvdb_hip_index_set_option(zvdb_hip_index *h, uint32_t option, uint64_t value) {
    ZV_ARG(h, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    switch (option) {
    case ZVDB_OPT_CERTIFY:
        ZV_ARG(value <= 1, "ZVDB_OPT_CERTIFY takes 0 or 1");
        h->cert_on = value != 0;
        return ZVDB_OK;
    case ZVDB_OPT_CERT_MIN_LEVEL:
        ZV_ARG(value <= 3, "ZVDB_OPT_CERT_MIN_LEVEL takes 0 .. 3");
        h->cert_level_min = (uint32_t)value;
        return ZVDB_OK;
    case ZVDB_OPT_EF_SEARCH_WIDTH:
        ZV_ARG(value >= 1 && value <= 4, "ZVDB_OPT_EF_SEARCH_WIDTH takes 1 .. 4");
        h->ef_width = (uint32_t)value;
        ef_hint_reset(h, h->stream);  // (the visited-set fills recorded so far belong to the other width)
        return ZVDB_OK;
    case ZVDB_OPT_CERT_FORCE_REPAIR:
        ZV_ARG(value <= 1, "ZVDB_OPT_CERT_FORCE_REPAIR takes 0 or 1");
        h->cert_force = value != 0;
        return ZVDB_OK;
    default:
        break;
    }
    set_error("unknown option %u", option);
    return ZVDB_ERR_INVALID;
}

int zvdb_hip_bruteforce_cert_info(zvdb_hip_index *h, zvdb_hip_cert_info *out) {
    ZV_ARG(h && out, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    out->queries = h->cert_queries;
    out->batches = h->cert_batches;
    out->repaired = h->cert_host ? (uint64_t)__atomic_load_n(h->cert_host + 1, __ATOMIC_RELAXED) : 0;
    out->level = std::max(h->cert_level, h->cert_level_min);
    out->enabled = h->cert_on ? 1u : 0u;
    out->rerank_rows = h->cert_host ? (uint64_t)__atomic_load_n(h->cert_host + 2, __ATOMIC_RELAXED) : 0;
    for (int i = 0; i < 5; ++i) out->rerank_cycles[i] = h->cert_host ? __atomic_load_n(h->cert_host + 3 + i, __ATOMIC_RELAXED) : 0;
    out->guess_misses = h->cert_host ? __atomic_load_n(h->cert_host + 10, __ATOMIC_RELAXED) : 0;
    out->bound_checked = h->cert_host ? __atomic_load_n(h->cert_host + 8, __ATOMIC_RELAXED) : 0;
    out->bound_violations = h->cert_host ? __atomic_load_n(h->cert_host + 9, __ATOMIC_RELAXED) : 0;
    return ZVDB_OK;
}

int zvdb_hip_index_set_shard_count(zvdb_hip_index *h, uint32_t shards) {
    ZV_ARG(h, "null argument");
    ZV_ARG(shards >= 1, "shard count must be >= 1");
    std::lock_guard<std::mutex> g(h->mu);
    h->shard_count = shards;
    return ZVDB_OK;
}

int zvdb_hip_timing_enable(zvdb_hip_index *h, int enable) {
    ZV_ARG(h, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    h->timing = enable != 0;
    if (enable == 1) h->events_used = 0;  // 1: start a new measurement; 2: resume (sampled timing); 0: pause, keep the samples
    return ZVDB_OK;
}

int zvdb_hip_timing_read(zvdb_hip_index *h, double *out_total_ms, uint32_t *out_launches) {
    ZV_ARG(h && out_total_ms && out_launches, "null argument");
    std::lock_guard<std::mutex> g(h->mu);
    ZV_CHECK(use_device(h));
    double tot = 0.0;
    for (size_t i = 0; i < h->events_used; ++i) {
        ZV_HIP(hipEventSynchronize(h->events[i].second));
        float ms = 0.0f;
        ZV_HIP(hipEventElapsedTime(&ms, h->events[i].first, h->events[i].second));
        tot += ms;
    }
    *out_total_ms = tot;
    *out_launches = (uint32_t)h->events_used;
    h->events_used = 0;
    return ZVDB_OK;
}

}  // extern "C"
