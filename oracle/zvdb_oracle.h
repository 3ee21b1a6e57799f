/*
 * zvdb_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A from-scratch C restatement of the allisoneer/zvdb hot path (reference
 * src/hnsw.zig) used ONLY as the checker for the HIP path:
 *   - tests/                       (parity + known answers)
 *   - __graft_entry__.smoke()      (one small GPU-vs-oracle check)
 *   - bench.py's cpu_baseline leg  (timed CPU baseline, kind = "port")
 * Nothing under zvdb_amd/ may include, link or call this file.
 *
 * PARITY PIN STATUS
 *   The reference is Zig (>= 0.13, build.zig.zon:9); no zig toolchain exists in
 *   the build image or on the GPU box, so the reference cannot be compiled or
 *   run here.  The oracle is pinned by the known answers the reference's own
 *   tests hold (src/test_hnsw.zig:24-41, :43-53, :55-68, :104-126, and the
 *   properties of :70-102, :154-209, :275-317) -- see tests/test_oracle_pins.py.
 *   Two things are "parity unpinned" and say so wherever they are used:
 *     (1) the pop order of EXACTLY equal fp32 distances, which depends on the
 *         un-vendored Zig std.PriorityQueue internals (restated below from the
 *         published Zig 0.13 std/priority_queue.zig algorithm);
 *     (2) every extension north_star names that has no reference symbol:
 *         brute-force top-k, ef-mode beam search, dot/cosine metrics, bf16.
 */
#ifndef ZVDB_ORACLE_H
#define ZVDB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a1: distance (src/hnsw.zig:182-192) -------------------------------- */
/* strict left-to-right fp32: diff = a[i]-b[i]; sum += diff*diff; no sqrt.   */
float orc_distance_f32(const float *a, const float *b, size_t d);
double orc_distance_f64(const double *a, const double *b, size_t d);
int32_t orc_distance_i32(const int32_t *a, const int32_t *b, size_t d);
/* extensions (no reference symbol; parity unpinned): rank keys, smaller=closer */
float orc_neg_dot_f32(const float *a, const float *b, size_t d); /* -(sum a*b) sequential */

/* ---- synthetic data: U[0,1) fp32, counter based -------------------------- */
/* element i of stream `seed` = (mix64(seed, i) >> 40) * 2^-24.  Matches the  */
/* DISTRIBUTION of std.crypto.random.float(f32) used by                       */
/* benchmarks/shared_benchmarks.zig:53-59 (the reference RNG is unseedable).  */
uint64_t orc_mix64(uint64_t seed, uint64_t idx);
void orc_fill_uniform(uint64_t seed, uint64_t first_elem, uint64_t count, float *out);
/* round-to-nearest-even fp32 -> bf16 -> fp32 (extension helper) */
float orc_bf16_round(float x);

/* ---- HNSW(f32) index (src/hnsw.zig:8-247) -------------------------------- */
typedef struct orc_hnsw orc_hnsw;

/* init (hnsw.zig:52-62): infallible, no dimension declared. `level_seed`
 * seeds the oracle's own geometric level generator (hnsw.zig:172-180 uses an
 * unseedable CSPRNG; levels are unobservable through search). */
orc_hnsw *orc_hnsw_create(size_t m, size_t ef_construction, uint64_t level_seed);
void orc_hnsw_destroy(orc_hnsw *h); /* deinit (hnsw.zig:64-71) */

/* insert (hnsw.zig:73-117).  level < 0 -> draw randomLevel().  Returns 0, or
 * -1 on out-of-memory, -2 on dimension mismatch (the reference @panic's,
 * hnsw.zig:183-185). */
int orc_hnsw_insert(orc_hnsw *h, const float *point, size_t d, int level);

/* search (hnsw.zig:194-236): returns number of results (<= k; may be < k,
 * test_hnsw.zig:122-125), ids/dists ascending by distance (stable w.r.t. pop
 * order). out arrays need k entries. */
size_t orc_hnsw_search(orc_hnsw *h, const float *query, size_t d, size_t k,
                       uint64_t *out_ids, float *out_dist);
/* nq such searches in one call without touching the counters: read-only on the index, so several threads may run it on one index
 * (the query-parallel CPU baseline of bench.py -- beyond the reference, which serialises searches on one mutex, hnsw.zig:195-196) */
void orc_hnsw_search_many(const orc_hnsw *h, const float *queries, size_t nq, size_t d, size_t k,
                          uint64_t *out_ids, float *out_dist, uint64_t *out_n);

/* 16-way blocked squared-L2 used by the ef-mode TRAVERSAL (extension): the 4-float chunk c of the
 * rows goes to partial sum (c mod 16), each partial accumulates its chunks in increasing order
 * (x,y,z,w sequentially, un-contracted); the 16 partials are then combined by the butterfly
 * p[l] = p[l] + p[l ^ off] for off = 8, 4, 2, 1.  This is the order a 16-lane group of a wave64
 * produces with coalesced 16-byte loads; it differs from a1 only in rounding. */
float orc_distance_blocked16_f32(const float *a, const float *b, size_t d);

/* ef-mode beam search over layer 0 (EXTENSION, parity unpinned by the
 * reference; this function IS the specification the HIP kernel must match).
 * Traversal distances are orc_distance_blocked16_f32; the returned k entries are the first k of W,
 * re-evaluated with the exact a1 distance and ordered by (exact distance, id):
 *   W = list of <= ef (dist,id) entries kept sorted ascending by (dist,id);
 *   seed W with node `entry`; visited = {entry};
 *   repeat: c = first entry of W not yet expanded; none -> stop; mark expanded;
 *     for each layer-0 neighbour n of c in list order, n not visited:
 *        visited += n; dn = distance(query, n);
 *        if |W| < ef or (dn,n) < last(W): insert keeping order, drop last if |W| > ef.
 *   result = first min(k,|W|) entries of W.                                   */
size_t orc_hnsw_search_ef(orc_hnsw *h, const float *query, size_t d, size_t k,
                          size_t ef, uint64_t *out_ids, float *out_dist);
/* the same beam search from a chosen entry node under a chosen metric: 0 = squared L2 (above), 1 = inner product --
 * traversal distance -(16-way blocked sum of a*b) (orc_neg_dot_blocked16_f32), final order by the sequential
 * orc_neg_dot_f32 (cosine = inner product over rows and queries normalised by the caller). */
float orc_neg_dot_blocked16_f32(const float *a, const float *b, size_t d);
size_t orc_hnsw_search_ef_ex(orc_hnsw *h, const float *query, size_t d, size_t k, size_t ef, int metric, size_t entry,
                             uint64_t *out_ids, float *out_dist);
/* ... with a SEARCH WIDTH w (1 .. 8; the specification of ZVDB_OPT_EF_SEARCH_WIDTH): the step "c = first entry of W not yet
 * expanded" becomes "c_1 .. c_j = the first j <= w entries of W not yet expanded (none -> stop); mark ALL of them expanded
 * (j pops)", then the neighbour lists of c_1, ..., c_j are walked one after the other exactly as above (visited test, distance,
 * insertion into W).  A node picked in a step stays picked whatever its step's insertions do to W.  w = 1 is the search above;
 * w > 1 gives a GPU w lists' worth of row fetches per dependent step (the reference has no counterpart: hnsw.zig:211-224 pops
 * one node at a time). */
size_t orc_hnsw_search_ef_w(orc_hnsw *h, const float *query, size_t d, size_t k, size_t ef, int metric, size_t entry, size_t width,
                            uint64_t *out_ids, float *out_dist);

size_t orc_hnsw_count(const orc_hnsw *h);
size_t orc_hnsw_dim(const orc_hnsw *h);
size_t orc_hnsw_max_level(const orc_hnsw *h);
int orc_hnsw_node_level(const orc_hnsw *h, size_t id);
/* copies connections[layer] of node id into out (cap entries); returns length,
 * or (size_t)-1 if the node lacks that layer. */
size_t orc_hnsw_neighbors(const orc_hnsw *h, size_t id, size_t layer, uint64_t *out, size_t cap);
const float *orc_hnsw_point(const orc_hnsw *h, size_t id);
/* replace connections[layer] of node id (test helper for hand-made graphs) */
int orc_hnsw_set_neighbors(orc_hnsw *h, size_t id, size_t layer, const uint64_t *ids, size_t n);

/* counters (E = distance evaluations on the search frontier path
 * hnsw.zig:208,:219 -- NOT the comparator re-evaluations of :230; P = pops). */
typedef struct {
    uint64_t search_dist_evals; /* hnsw.zig:208 + :219 */
    uint64_t search_pops;       /* hnsw.zig:212 */
    uint64_t sort_dist_evals;   /* hnsw.zig:230 comparator (2 per compare) */
    uint64_t insert_dist_evals; /* hnsw.zig:86,:96 */
    uint64_t shrink_dist_evals; /* hnsw.zig:160-161 */
} orc_counters;
void orc_hnsw_counters(const orc_hnsw *h, orc_counters *out);
void orc_hnsw_reset_counters(orc_hnsw *h);

/* ---- bulk graph construction + hierarchical entry descent (EXTENSION; SURVEY 8f rank 1 + 3; parity unpinned by the
 * reference, whose insert links one edge per layer and never reads ef_construction: hnsw.zig:59,:106-108).  These
 * functions are the specification of graph_build.hip and k_descend; semantics are spelled out at their definitions. */
uint32_t orc_bulk_level(uint64_t seed, uint64_t id);
size_t orc_graph_select(const float *V, size_t d, size_t ld, int metric, const uint64_t *cand_ids, const float *cand_dist,
                        size_t ncand, size_t cap, uint64_t *out_ids, float *out_dist);
void orc_graph_build_layer0(const float *V, size_t N, size_t d, size_t ld, int metric, const uint64_t *cand_ids,
                            const float *cand_dist, size_t ncand, size_t M, size_t M0, size_t cmax, uint32_t *adj, uint32_t *deg);
size_t orc_hnsw_descend(orc_hnsw *h, const float *query, size_t d, int metric, size_t top);

/* ---- a10: exact brute-force top-k (EXTENSION, parity unpinned) ------------ */
/* For each query the k smallest a1 distances over rows [0,N), ascending,
 * ties by lower id.  metric: 0 = L2^2 (a1), 1 = -dot, 2 = cosine (rows and
 * queries are assumed pre-normalised; rank key = -dot).  Returns count per
 * query = min(k,N) in out_count (may be NULL).                              */
void orc_bruteforce_topk(const float *V, size_t N, size_t d, size_t ld,
                         const float *Q, size_t nq, size_t k, int metric,
                         uint64_t *out_ids, float *out_dist, uint32_t *out_count);

/* k-way merge of G per-shard top-k lists (C1 all-gather consumer): inputs
 * ids[g][nq][k], dist[g][nq][k], count[g][nq]; output top-k by (dist,id). */
void orc_merge_topk(const uint64_t *ids, const float *dist, const uint32_t *count,
                    size_t G, size_t nq, size_t k,
                    uint64_t *out_ids, float *out_dist, uint32_t *out_count);

#ifdef __cplusplus
}
#endif
#endif
