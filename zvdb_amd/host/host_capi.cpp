// host_capi.cpp -- C wrappers over zvdb::HNSW<float> (hnsw.hpp) so that Python (ctypes) tests and
// tools drive the SAME host-side bookkeeping a Zig/C++ caller would.  Not part of the drop-in ABI
// (that is include/zvdb_hip.h); built as libzvdb_host.so and linked against libzvdb_hip.so.
#include <cstring>
#include <string>

#include "hnsw.hpp"

namespace {
thread_local std::string g_err;
using H = zvdb::HNSW<float>;

template <typename F>
int guard(F &&f) {
    try {
        f();
        return 0;
    } catch (const zvdb::Panic &e) {
        g_err = e.what();
        return 100;  // reference @panic (hnsw.zig:183-185)
    } catch (const zvdb::Error &e) {
        g_err = e.what();
        return e.code;
    } catch (const std::bad_alloc &) {
        g_err = "OutOfMemory";
        return ZVDB_ERR_OOM;
    } catch (const std::exception &e) {
        g_err = e.what();
        return ZVDB_ERR_INVALID;
    }
}
}  // namespace

extern "C" {

const char *zvdb_host_last_error(void) { return g_err.c_str(); }

void *zvdb_host_hnsw_init(size_t m, size_t ef_construction, int device, uint32_t flags) {
    return new (std::nothrow) H(m, ef_construction, device, flags);
}

void zvdb_host_hnsw_deinit(void *p) { delete static_cast<H *>(p); }

int zvdb_host_hnsw_insert(void *p, const float *point, size_t len, int level) {
    return guard([&] {
        H *h = static_cast<H *>(p);
        if (level >= 0) h->set_next_level(level);
        h->insert(point, len);
    });
}

int zvdb_host_hnsw_insert_many(void *p, const float *points, size_t n, size_t len, const int *levels) {
    return guard([&] { static_cast<H *>(p)->insert_many(points, n, len, levels); });
}

int zvdb_host_hnsw_search(void *p, const float *query, size_t len, size_t k, uint64_t *out_ids,
                          float *out_dist, size_t *out_n, uint64_t *out_evals, uint64_t *out_pops) {
    return guard([&] {
        H *h = static_cast<H *>(p);
        auto res = h->search(query, len, k);
        *out_n = res.size();
        for (size_t i = 0; i < res.size(); ++i) {
            out_ids[i] = res[i].id;
            out_dist[i] = h->last_distances()[i];
        }
        if (out_evals) *out_evals = h->last_stats().dist_evals;
        if (out_pops) *out_pops = h->last_stats().pops;
    });
}

int zvdb_host_hnsw_search_many(void *p, const float *queries, size_t nq, size_t len, size_t k, uint64_t *out_ids, float *out_dist,
                               uint32_t *out_count) {
    return guard([&] { static_cast<H *>(p)->search_many(queries, nq, len, k, out_ids, out_dist, out_count); });
}

size_t zvdb_host_hnsw_count(void *p) { return static_cast<H *>(p)->count(); }
size_t zvdb_host_hnsw_dim(void *p) { return static_cast<H *>(p)->dim(); }
size_t zvdb_host_hnsw_max_level(void *p) { return static_cast<H *>(p)->max_level; }
long long zvdb_host_hnsw_entry_point(void *p) {
    H *h = static_cast<H *>(p);
    return h->entry_point ? (long long)*h->entry_point : -1;
}
void *zvdb_host_hnsw_handle(void *p) { return static_cast<H *>(p)->handle(); }

// connections[layer] of node id; returns -1 if the node lacks that layer
long long zvdb_host_hnsw_connections(void *p, size_t id, size_t layer, uint64_t *out, size_t cap) {
    H *h = static_cast<H *>(p);
    if (id >= h->count()) return -2;
    const auto &c = h->connections(id);
    if (layer >= c.size()) return -1;
    for (size_t i = 0; i < c[layer].size() && i < cap; ++i) out[i] = c[layer][i];
    return (long long)c[layer].size();
}

}  // extern "C"
