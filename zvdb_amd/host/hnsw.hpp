// hnsw.hpp -- C++ host-side mirror of zvdb's `HNSW(T)` (reference src/hnsw.zig:8-247) over the
// C ABI of libzvdb_hip.so (include/zvdb_hip.h).
//
// This is what `hnsw_hip.zig` would be if a Zig toolchain existed in the build image (it does
// not: SURVEY.md 8c; the Zig binding is shown in INTEGRATION.md).  Exactly as north_star asks,
// the GRAPH BOOKKEEPING stays on the host -- id assignment, level draw, per-layer adjacency
// lists, connect/shrink list surgery, the global mutex, Node copies in results -- while every
// distance evaluation, neighbour-list scan and top-k selection is a HIP kernel behind the ABI:
//
//   reference call site                          ABI call
//   hnsw.zig:24-26  Node.init vector copy    ->  zvdb_hip_add
//   hnsw.zig:86     entry distance           ->  zvdb_hip_distances_from
//   hnsw.zig:88-104 greedy scan, layer 0     ->  zvdb_hip_greedy_layer0   (device mirror of layer 0)
//   hnsw.zig:88-104 greedy scan, layers >= 1 ->  zvdb_hip_distances_from  (lists live on the host)
//   hnsw.zig:160-161 shrink comparator       ->  zvdb_hip_distances_from  (+ stable insertion sort here)
//   hnsw.zig:128-140 list updates at layer 0 ->  zvdb_hip_set_neighbors   (host -> device mirror)
//   hnsw.zig:194-236 search                  ->  zvdb_hip_search_graph (ef = 0: reference-compat)
//
// Same names, argument meaning and error behaviour as the reference: init(m, ef_construction)
// declares no dimension (hnsw.zig:52); insert copies the point (:24-26); search returns Node
// copies whose `point`/`connections` alias index memory (:213-214,:235) and may return fewer
// than k (test_hnsw.zig:122-125); a dimension mismatch is a panic, not an error (:183-185).
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/zvdb_hip.h"

namespace zvdb {

// Zig error set members the reference can return (hnsw.zig:120-121 + allocator failures)
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};
// the reference @panic's on mismatched dimensions (hnsw.zig:183-185)
struct Panic : std::logic_error {
    using std::logic_error::logic_error;
};

template <typename T>
class HNSW;  // the GPU path is fp32 only; HNSW(i32)/HNSW(f64) stay on the CPU

template <>
class HNSW<float> {
  public:
    // mirrors `const Node = struct { id, point, connections, mutex }` (hnsw.zig:12-17);
    // search returns by-value copies whose slices alias index-owned memory.
    struct Node {
        size_t id = 0;
        const float *point = nullptr;  // point.ptr
        size_t point_len = 0;          // point.len
        const std::vector<std::vector<size_t>> *connections = nullptr;
    };

    // fields readable by callers, as in the reference (hnsw.zig:44-50)
    std::optional<size_t> entry_point;
    size_t max_level = 0;
    size_t m;
    size_t ef_construction;

    // HNSW(T).init(allocator, m, ef_construction) -- hnsw.zig:52-62.  No device work happens
    // here (the reference init is infallible and allocation-free); the device index is created
    // at the first insert, when the dimension becomes known.
    HNSW(size_t m_, size_t ef_construction_, int device = -1, uint32_t flags = 0)
        : m(m_), ef_construction(ef_construction_), device_(device), flags_(flags),
          rng_(std::random_device{}()) {}
    HNSW(const HNSW &) = delete;
    HNSW &operator=(const HNSW &) = delete;
    ~HNSW() { deinit(); }

    // hnsw.zig:64-71
    void deinit() {
        if (h_) zvdb_hip_destroy(h_);
        h_ = nullptr;
        points_.clear();
        conns_.clear();
        entry_point.reset();
        max_level = 0;
    }

    // `hnsw.nodes.count()` (test_hnsw.zig:198)
    size_t count() const {
        std::lock_guard<std::mutex> g(mu_);
        return points_.size();
    }
    size_t dim() const { return dim_; }
    zvdb_hip_index *handle() { return h_; }

    // test hook: force the level of the NEXT insert (randomLevel is unobservable through search,
    // SURVEY 0.4, but full-graph parity tests want identical upper layers)
    void set_next_level(int level) { forced_level_ = level; }

    // insert -- hnsw.zig:73-117
    void insert(const float *point, size_t len) {
        std::lock_guard<std::mutex> g(mu_);                                   // :74-75
        insert_locked(point, len);
    }

    // n inserts, same result as n insert() calls in this order (extension: the reference has no bulk call).  The layer-0
    // work of a whole chunk runs inside ONE device launch (zvdb_hip_insert_batch_layer0); the host bookkeeping (levels, the
    // layer-0 lists the device settled on, the upper layers) follows in the same order.  levels: optional forced levels.
    void insert_many(const float *points, size_t n, size_t len, const int *levels = nullptr) {
        std::lock_guard<std::mutex> g(mu_);
        if (n && !h_) create_device_index(len);  // (settles which of the two insert paths this index uses)
        if (device_layers_) {
            insert_full(points, n, len, levels);
            return;
        }
        size_t i = 0;
        if (n && !entry_point) {  // the first node of an index has nothing to scan or connect to
            if (levels) forced_level_ = levels[0];
            insert_locked(points, len);
            i = 1;
        }
        if (i < n && len != dim_) throw Panic("Mismatched dimensions in distance calculation");
        const size_t chunk_max = 8192;
        while (i < n) {
            const size_t nc = std::min(chunk_max, n - i);
            b_ep_.resize(nc); b_dist_.resize(nc); b_lists_.resize(nc * (m ? m : 1)); b_epn_.resize(nc); b_on_.resize(nc);
            uint64_t first = 0;
            check(zvdb_hip_insert_batch_layer0(h_, points + i * len, (uint32_t)len, nc, *entry_point, &first, b_ep_.data(),
                                               b_dist_.data(), b_lists_.data(), b_epn_.data(), b_on_.data()));
            if (first != points_.size()) throw Error(ZVDB_ERR_INVALID, "device id out of step with host id");
            for (size_t j = 0; j < nc; ++j) {
                const size_t level = levels ? (size_t)levels[i + j] : random_level();
                adopt(points + (i + j) * len, len, level, b_ep_[j], b_dist_[j], b_lists_.data() + j * m, b_epn_[j], b_on_[j]);
            }
            i += nc;
        }
    }

    // false: the step-by-step binding (layer 0 fused on the device, upper layers a host loop over distances_from);
    // true (default; ZVDB_HOST_INSERT=steps selects the other): the whole insert on the device, zvdb_hip_insert_batch.
    // Fixed at the first insert: an index is fed through one of the two only.
    void use_device_layers(bool on) { if (!h_) device_layers_ = on; }

  private:
    // the whole insert on the device: levels drawn here (the binding owns the RNG, hnsw.zig:172-180), one call for the n
    // points, then the host mirror adopts, per point and layer, the lists the device settled on
    void insert_full(const float *points, size_t n, size_t len, const int *levels) {
        if (n == 0) return;
        if (!h_) create_device_index(len);
        if (len != dim_) throw Panic("Mismatched dimensions in distance calculation");
        lv_.resize(n);
        uint64_t total = 0;
        for (size_t i = 0; i < n; ++i) {
            size_t level = levels ? (size_t)levels[i] : random_level();
            if (i == 0 && forced_level_ >= 0) level = (size_t)forced_level_;
            lv_[i] = (uint32_t)level;
            total += level + 1;
        }
        forced_level_ = -1;
        const size_t rw = 4 + m;
        nrec_.resize(n);
        recs_.resize(total * rw);
        uint64_t first = 0;
        check(zvdb_hip_insert_batch(h_, points, (uint32_t)len, n, lv_.data(), &first, nrec_.data(), recs_.data(), total));
        if (first != points_.size()) throw Error(ZVDB_ERR_INVALID, "device id out of step with host id");
        const uint32_t *r = recs_.data();
        for (size_t i = 0; i < n; ++i) {
            const size_t id = points_.size(), level = lv_[i];                  // :77-78
            std::unique_ptr<float[]> owned(new float[len ? len : 1]);
            std::memcpy(owned.get(), points + i * len, len * sizeof(float));
            points_.push_back(std::move(owned));                               // :82
            conns_.emplace_back(level + 1);                                    // :19-23
            if (entry_point) {                                                 // :84
                for (uint32_t l = 0; l < nrec_[i]; ++l, r += rw) {             // layers 0..min(level, max_level) (:106)
                    const size_t ep = r[0];
                    conns_[id][l].assign(r[1] ? 1 : 0, ep);
                    if (r[3]) conns_[ep][l].assign(r + 4, r + 4 + r[2]);
                }
            } else {
                entry_point = id;                                              // :110-112
            }
            if (level > max_level) max_level = level;                          // :114-116
        }
    }

    void insert_locked(const float *point, size_t len) {
        if (!h_) create_device_index(len);  // (settles which of the two insert paths this index uses)
        if (device_layers_) {
            insert_full(point, 1, len, nullptr);
            return;
        }
        if (!h_) create_device_index(len);
        if (len != dim_) throw Panic("Mismatched dimensions in distance calculation");
        const size_t id = points_.size();                                      // :77
        const size_t level = forced_level_ >= 0 ? (size_t)forced_level_ : random_level();  // :78
        forced_level_ = -1;
        // ONE launch + one synchronisation for everything layer 0 needs: Node.init copy (:24-26), distance to the entry
        // point (:86), the while(changed) scan at layer 0 (:90-104) and connect(id, ep, 0) (:106-108) incl. both
        // shrinkConnections, on the device mirror; the lists it settled on are adopted below
        uint64_t dev_id = 0, ep = entry_point ? *entry_point : ~0ull;          // :85
        float curr_dist = 0.0f;
        uint32_t ep_n = 0, owner_n = 0;
        ids_.resize(m + 1);
        check(zvdb_hip_insert_layer0(h_, point, (uint32_t)len, ep, &dev_id, &ep, &curr_dist, ids_.data(), &ep_n, &owner_n,
                                     nullptr));
        if (dev_id != id) throw Error(ZVDB_ERR_INVALID, "device id out of step with host id");
        l0_.assign(ids_.begin(), ids_.begin() + ep_n);
        adopt(point, len, level, ep, curr_dist, l0_.data(), ep_n, owner_n);
    }

    // host bookkeeping of one insert whose layer-0 work the device has done: the owned copy of the point, the node's
    // lists, the layer-0 lists the device settled on, then the upper layers (hnsw.zig:88-108 for layer >= 1) and :110-116
    void adopt(const float *point, size_t len, size_t level, uint64_t ep, float curr_dist, const uint64_t *ep_list,
               uint32_t ep_n, uint32_t owner_n) {
        const size_t id = points_.size();
        std::unique_ptr<float[]> owned(new float[len ? len : 1]);
        std::memcpy(owned.get(), point, len * sizeof(float));
        points_.push_back(std::move(owned));                                   // :82
        conns_.emplace_back(level + 1);                                        // :19-23

        if (entry_point) {                                                     // :84
            for (size_t layer = 0; layer <= max_level; ++layer) {              // :88 (pre-insert max_level)
                if (layer == 0) {
                    // host mirror of what the device did: id's list = [ep] (shrunk to m), ep's list as returned
                    conns_[id][0].assign(owner_n ? 1 : 0, (size_t)ep);
                    conns_[(size_t)ep][0].assign(ep_list, ep_list + ep_n);
                    continue;
                } else {
                    bool changed = true;
                    while (changed) {                                          // :90
                        changed = false;
                        const auto &cur = conns_[ep];                          // :92
                        if (layer < cur.size() && !cur[layer].empty()) {       // :93
                            const std::vector<size_t> &lst = cur[layer];
                            ids_.assign(lst.begin(), lst.end());
                            dist_.resize(lst.size());
                            check(zvdb_hip_distances_from(h_, id, ids_.data(), (uint32_t)ids_.size(),
                                                          dist_.data()));      // :96 (whole list, :94)
                            for (size_t i = 0; i < lst.size(); ++i) {
                                if (dist_[i] < curr_dist) {                    // :97 strict
                                    ep = ids_[i];
                                    curr_dist = dist_[i];
                                    changed = true;
                                }
                            }
                        }
                    }
                }
                if (layer <= level) connect(id, (size_t)ep, layer);            // :106-108
            }
        } else {
            entry_point = id;                                                  // :110-112
        }
        if (level > max_level) max_level = level;                              // :114-116
    }

  public:
    void insert(const std::vector<float> &p) { insert(p.data(), p.size()); }

    // search -- hnsw.zig:194-236.  Caller owns the returned vector (the reference returns a
    // caller-owned slice, :235); elements alias index memory until deinit/mutation.
    std::vector<Node> search(const float *query, size_t len, size_t k) {
        std::lock_guard<std::mutex> g(mu_);                                    // :195-196
        std::vector<Node> result;
        if (!entry_point || k == 0) return result;                             // :201 / loop never runs
        if (len != dim_) throw Panic("Mismatched dimensions in distance calculation");
        out_ids_.resize(k);
        out_dist_.resize(k);
        uint32_t cnt = 0;
        check(zvdb_hip_search_graph(h_, query, 1, (uint32_t)len, (uint32_t)k, /*ef=*/0,
                                    /*entry=*/*entry_point, out_ids_.data(), out_dist_.data(), &cnt,
                                    &last_stats_));
        result.reserve(cnt);
        for (uint32_t i = 0; i < cnt; ++i) {
            Node n;
            n.id = (size_t)out_ids_[i];
            n.point = points_[n.id].get();
            n.point_len = dim_;
            n.connections = &conns_[n.id];
            result.push_back(n);
        }
        last_dist_.assign(out_dist_.begin(), out_dist_.begin() + cnt);
        return result;
    }
    std::vector<Node> search(const std::vector<float> &q, size_t k) { return search(q.data(), q.size(), k); }

    // nq searches in ONE device launch (extension: the reference has no batch call, hnsw.zig:194 takes one query): the same
    // results as nq search() calls -- ids and distances only, no Node copies (callers that want the points read them through
    // point(id)).  out_ids / out_dist: nq x k, row-major, unused slots id = UINT64_MAX / +inf; out_count[q] <= k results of
    // query q (fewer than k are legal, test_hnsw.zig:122-125).  Where the device wins: one launch answers the whole batch
    // (millions of queries per second) while a search() call pays a launch + a synchronisation per query.
    void search_many(const float *queries, size_t nq, size_t len, size_t k, uint64_t *out_ids, float *out_dist, uint32_t *out_count) {
        std::lock_guard<std::mutex> g(mu_);                                    // :195-196
        for (size_t q = 0; q < nq; ++q) out_count[q] = 0;
        if (!entry_point || k == 0 || nq == 0) {                               // :201: an empty index answers nothing
            for (size_t i = 0; i < nq * k; ++i) { out_ids[i] = ~0ull; out_dist[i] = std::numeric_limits<float>::infinity(); }
            return;
        }
        if (len != dim_) throw Panic("Mismatched dimensions in distance calculation");
        check(zvdb_hip_search_graph(h_, queries, (uint32_t)nq, (uint32_t)len, (uint32_t)k, /*ef=*/0, /*entry=*/*entry_point,
                                    out_ids, out_dist, out_count, &last_stats_));
    }
    // the owned copy of a point (what a search result's Node.point aliases)
    const float *point(size_t id) const { return id < points_.size() ? points_[id].get() : nullptr; }

    // distances of the last search's results (the reference recomputes them in its tests)
    const std::vector<float> &last_distances() const { return last_dist_; }
    const zvdb_hip_search_stats &last_stats() const { return last_stats_; }
    const std::vector<std::vector<size_t>> &connections(size_t id) const { return conns_[id]; }

  private:
    void check(int rc) {
        if (rc == ZVDB_OK) return;
        const std::string msg = zvdb_hip_last_error();
        if (rc == ZVDB_ERR_DIM_MISMATCH) throw Panic(msg);
        throw Error(rc, msg);
    }

    void create_device_index(size_t len) {
        zvdb_hip_config cfg{};
        cfg.abi_version = ZVDB_HIP_ABI_VERSION;
        cfg.dim = (uint32_t)len;
        cfg.metric = ZVDB_METRIC_L2SQ;  // the only reference metric (hnsw.zig:182-192)
        cfg.m = (uint32_t)m;
        cfg.ef_construction = (uint32_t)ef_construction;
        cfg.device = device_;
        cfg.flags = flags_;
        cfg.graph_stride = (uint32_t)m;  // m caps every layer incl. 0 (hnsw.zig:146)
        check(zvdb_hip_create(&cfg, &h_));
        dim_ = len;
        // zvdb_hip_insert_batch serves plain fp32 rows that fit a wave's LDS; anything else goes step by step
        if ((flags_ & ZVDB_FLAG_BF16_ROWS) || len * sizeof(float) > 96 * 1024) device_layers_ = false;
    }

    // randomLevel -- hnsw.zig:172-180: geometric(p = 0.5), capped at 31
    size_t random_level() {
        size_t level = 0;
        std::uniform_real_distribution<float> u(0.0f, 1.0f);
        while (level < 31 && u(rng_) < 0.5f) level += 1;
        return level;
    }

    // connect -- hnsw.zig:119-141
    void connect(size_t source, size_t target, size_t level) {
        if (source >= conns_.size() || target >= conns_.size())
            throw Error(ZVDB_ERR_NODE_NOT_FOUND, "NodeNotFound");              // :120-121
        auto &s = conns_[source];
        auto &t = conns_[target];
        if (level < s.size()) s[level].push_back(target);                      // :128-130
        if (level < t.size()) t[level].push_back(source);                      // :131-133
        if (level < s.size()) shrink_connections(source, level);               // :135-137
        if (level < t.size()) shrink_connections(target, level);               // :138-140
        // (layer 0 never comes through here: zvdb_hip_insert_layer0 connects it on the device mirror)
    }

    // shrinkConnections -- hnsw.zig:143-170
    void shrink_connections(size_t node_id, size_t level) {
        auto &lst = conns_[node_id][level];
        if (lst.size() <= m) return;                                           // :146
        ids_.assign(lst.begin(), lst.end());                                   // :148-150
        dist_.resize(ids_.size());
        // the comparator's distance(owner, candidate) (:160-161), one batched kernel launch
        check(zvdb_hip_distances_from(h_, node_id, ids_.data(), (uint32_t)ids_.size(), dist_.data()));
        // std.sort.insertion (:166): stable; sorting (id, dist) pairs on the precomputed values
        // is identical to recomputing both distances in every comparison.
        for (size_t i = 1; i < ids_.size(); ++i) {
            size_t j = i;
            while (j > 0 && dist_[j] < dist_[j - 1]) {                         // :162 strict
                std::swap(dist_[j], dist_[j - 1]);
                std::swap(ids_[j], ids_[j - 1]);
                --j;
            }
        }
        lst.resize(m);                                                         // :168
        for (size_t i = 0; i < m; ++i) lst[i] = (size_t)ids_[i];               // :169
    }

    mutable std::mutex mu_;  // hnsw.zig:50
    zvdb_hip_index *h_ = nullptr;
    int device_;
    uint32_t flags_;
    size_t dim_ = 0;
    int forced_level_ = -1;
    std::mt19937_64 rng_;
    std::vector<std::unique_ptr<float[]>> points_;           // Node.point (host copy: results alias it)
    std::vector<std::vector<std::vector<size_t>>> conns_;    // Node.connections
    std::vector<uint64_t> ids_, out_ids_, l0_, b_ep_, b_lists_;
    std::vector<uint32_t> b_epn_, b_on_, lv_, nrec_, recs_;
    bool device_layers_ = [] { const char *e = zvdb_hip_get_tuning("ZVDB_HOST_INSERT"); return !(e && e[0] == 's'); }();
    std::vector<float> b_dist_;
    std::vector<float> dist_, out_dist_, last_dist_;
    zvdb_hip_search_stats last_stats_{};
};

}  // namespace zvdb
