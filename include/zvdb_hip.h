/*
 * zvdb_hip.h -- C ABI of libzvdb_hip.so: the MI355X (gfx950) distance /
 * neighbour-scan / top-k engine behind zvdb's HNSW(T) insert/search API.
 *
 * The reference (allisoneer/zvdb) has no FFI: its boundary is the Zig generic
 * type re-exported at src/zvdb.zig:1 (HNSW(T), src/hnsw.zig:8).  These entry
 * points are what a thin `hnsw_hip.zig` wrapper binds with `extern "c"` (see
 * INTEGRATION.md) -- host-side graph bookkeeping (ids, adjacency lists, level
 * draw, mutex, Node copies) stays on the host; every distance evaluation,
 * neighbour-list scan and top-k selection runs in hand-written HIP kernels.
 *
 * Conventions
 *   - every call returns an int status (ZVDB_OK = 0); zvdb_hip_last_error()
 *     returns a thread-local message for the last failing call;
 *   - the caller owns every in/out buffer; no callbacks; plain pointers/sizes;
 *   - ids cross the ABI as uint64_t (Zig usize, hnsw.zig:13) and are u32 on
 *     the device;
 *   - a handle carries its own mutex: calls may come from any thread
 *     (mirrors the global mutex of hnsw.zig:74-75,:195-196);
 *   - "host" entry points take host pointers and synchronise before returning;
 *     "_dev" entry points take DEVICE pointers + a hipStream_t (as void*) and
 *     only enqueue work: they are what bench.py and the multi-GPU path call
 *     with torch-owned device memory.  Index-owned scratch they need is
 *     allocated the FIRST time a call of a given shape needs it (hipMalloc, and a
 *     device synchronise when a smaller buffer is replaced) -- the brute-force
 *     workspace is the caller's (zvdb_hip_bruteforce_workspace), the graph
 *     search's scratch can be taken ahead of time with
 *     zvdb_hip_search_graph_reserve; after that a _dev call allocates nothing
 *     and does not synchronise;
 *   - without a HIP device every entry point that needs one fails with
 *     ZVDB_ERR_NO_DEVICE.  There is no CPU fallback.
 */
#ifndef ZVDB_HIP_H
#define ZVDB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZVDB_HIP_ABI_VERSION 1

enum zvdb_status {
    ZVDB_OK = 0,
    ZVDB_ERR_INVALID = 1,        /* bad argument */
    ZVDB_ERR_DIM_MISMATCH = 2,   /* reference @panic, hnsw.zig:183-185 */
    ZVDB_ERR_OOM = 3,            /* error.OutOfMemory */
    ZVDB_ERR_NODE_NOT_FOUND = 4, /* error.NodeNotFound, hnsw.zig:120-121 */
    ZVDB_ERR_HIP = 5,            /* HIP runtime failure */
    ZVDB_ERR_UNSUPPORTED = 6,
    ZVDB_ERR_NO_DEVICE = 7
};

enum zvdb_metric {
    ZVDB_METRIC_L2SQ = 0,  /* hnsw.zig:182-192 (the only reference metric) */
    ZVDB_METRIC_DOT = 1,   /* extension: rank by -q.v (parity unpinned) */
    ZVDB_METRIC_COSINE = 2 /* extension: rows normalised at add; rank by -q.v */
};

/* brute-force contraction modes (a10; no reference symbol) */
enum zvdb_bf_mode {
    /* The first three are exact AND certified (ZVDB_OPT_CERTIFY below): per query either a proof that no row the screen
     * left out can enter the top-k, or an exact scan of all rows in the reference's arithmetic. */
    ZVDB_BF_F32 = 0,  /* fp32 MFMA (v_mfma_f32_32x32x2_f32) screen + exact fp32 re-rank */
    ZVDB_BF_BF16 = 1, /* bf16 MFMA screen over the bf16 mirror + exact fp32 re-rank */
    ZVDB_BF_I8 = 2,   /* int8 MFMA (v_mfma_i32_32x32x32_i8) screen over a scalar-quantised mirror
                         (global affine range for L2, symmetric for dot/cosine; built lazily from
                         the fp32 rows) + exact fp32 re-rank */
    ZVDB_BF_FP4 = 3   /* fp4 (OCP E2M1) MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales) screen over
                         a 4-bit scalar-quantised mirror (same global range, 2 elements per byte; built lazily)
                         + exact fp32 re-rank of k' = 64 survivors.  APPROXIMATE: a true neighbour whose 4-bit
                         key is not among the 64 best is lost (recall is measured, not guaranteed); returned
                         distances are still the reference's exact arithmetic */
};

typedef struct zvdb_hip_index zvdb_hip_index; /* opaque */

typedef struct zvdb_hip_config {
    uint32_t abi_version;     /* ZVDB_HIP_ABI_VERSION */
    uint32_t dim;             /* 0 = take from the first add (hnsw.zig:52 declares none) */
    uint32_t metric;          /* enum zvdb_metric */
    uint32_t m;               /* hnsw.zig:48: cap of EVERY layer's list incl. layer 0 (:146) */
    uint32_t ef_construction; /* hnsw.zig:49: stored, never read by the reference */
    int32_t device;           /* HIP device ordinal; -1 = current device */
    uint32_t flags;           /* ZVDB_FLAG_* */
    uint32_t graph_stride;    /* layer-0 adjacency stride on device; 0 = m */
    uint64_t capacity;        /* rows to reserve up front (0 = grow geometrically) */
} zvdb_hip_config;

#define ZVDB_FLAG_BF16_MIRROR 1u /* keep a bf16 copy of the rows for ZVDB_BF_BF16 */
/* bf16 row STORAGE ("bf16 = storage type with fp32 accumulate"): every element is rounded
 * to bf16 (round-to-nearest-even) when it is added -- after the cosine normalisation -- and the index keeps ONLY the bf16
 * rows (2 bytes per element: half the HBM footprint and half the bytes every gather-bound kernel moves; there is no fp32
 * container).  get_vector widens exactly; every exact distance and every re-rank is the reference's fp32 arithmetic
 * (hnsw.zig:182-192), in its order, over the widened values -- bit-identical to an fp32 index fed bf16-representable rows.
 * The bf16 brute-force screen is lossless on such an index and also serves ZVDB_BF_F32.  Queries are never rounded.
 * zvdb_hip_insert_batch (the fused reference insert) serves fp32 rows only; the step-by-step insert calls work. */
#define ZVDB_FLAG_BF16_ROWS 2u

typedef struct zvdb_hip_search_stats {
    uint64_t dist_evals; /* E of SURVEY 8d: hnsw.zig:208 + :219 evaluations */
    uint64_t pops;       /* P: hnsw.zig:212 */
} zvdb_hip_search_stats;

/* ---- library ------------------------------------------------------------ */
const char *zvdb_hip_last_error(void);
int zvdb_hip_abi_version(void);
int zvdb_hip_device_count(int *out_count);

/* ---- lifecycle: HNSW(T).init / deinit (hnsw.zig:52-62, :64-71) ----------- */
int zvdb_hip_create(const zvdb_hip_config *cfg, zvdb_hip_index **out);
void zvdb_hip_destroy(zvdb_hip_index *h);

/* nodes.count() (hnsw.zig:77, test_hnsw.zig:198) and the implied dimension */
int zvdb_hip_count(zvdb_hip_index *h, uint64_t *out_count);
int zvdb_hip_dim(zvdb_hip_index *h, uint32_t *out_dim);

/* ---- rows: the vector copy of Node.init (hnsw.zig:24-26) ----------------- */
/* Append n rows of `dim` floats (host memory); ids are dense from count()
 * (hnsw.zig:77).  *first_id receives the id of rows[0].  dim must match
 * (ZVDB_ERR_DIM_MISMATCH mirrors the @panic of hnsw.zig:183-185). */
int zvdb_hip_add(zvdb_hip_index *h, const float *rows, uint64_t n, uint32_t dim, uint64_t *first_id);
/* Same with rows already in device memory (row-major, stride dim floats). */
int zvdb_hip_add_dev(zvdb_hip_index *h, const void *d_rows, uint64_t n, uint32_t dim,
                     void *stream, uint64_t *first_id);
/* Append n rows of the build's seeded synthetic U[0,1) stream (distribution of
 * shared_benchmarks.zig:53-59), generated on the device: element e of the
 * stream is (mix64(seed, e) >> 40) * 2^-24, rows [first_row, first_row+n). */
int zvdb_hip_add_synthetic(zvdb_hip_index *h, uint64_t n, uint32_t dim, uint64_t seed,
                           uint64_t first_row, uint64_t *first_id);
/* Fill a DEVICE buffer with rows of the same stream (queries for bench). */
int zvdb_hip_fill_synthetic_dev(void *d_out, uint64_t nrows, uint32_t dim, uint64_t seed,
                                uint64_t first_row, void *stream);
/* Copy row `id` back (Node.point readers: test_hnsw.zig:40,:67,:98,:306). */
int zvdb_hip_get_vector(zvdb_hip_index *h, uint64_t id, float *out, uint32_t dim);

/* ---- graph mirror: connections[0] of every node (hnsw.zig:15,:216) -------- */
/* Replace layer-0 list of `node` (host bookkeeping -> device mirror).  Only
 * layer 0 is mirrored: search reads nothing else (hnsw.zig:216). */
int zvdb_hip_set_neighbors(zvdb_hip_index *h, uint64_t node, const uint64_t *ids, uint32_t n);
/* Bulk upload: adj[n][stride] u32 + deg[n] u32 for nodes [0,n). */
int zvdb_hip_upload_graph(zvdb_hip_index *h, const uint32_t *adj, const uint32_t *deg,
                          uint64_t n, uint32_t stride);
int zvdb_hip_get_neighbors(zvdb_hip_index *h, uint64_t node, uint64_t *out_ids, uint32_t cap,
                           uint32_t *out_n);
/* Batched GPU construction of the layer-0 graph (NOT reference-parity: the
 * reference insert links one edge per layer, hnsw.zig:106-108).  Every node gets its `degree`
 * exact nearest neighbours (by the reference distance; ties by lower id; self excluded), found
 * with the brute-force MFMA path in batches of rows; lists are stored nearest first.  degree <=
 * graph_stride.  `mode` = enum zvdb_bf_mode of the screen (final order is exact either way). */
int zvdb_hip_build_knn_graph(zvdb_hip_index *h, uint32_t degree, uint32_t mode);
/* Bulk construction of a recall-competent HNSW graph on the device (SURVEY 8f rank 1 + 3; NOT reference parity: the
 * reference insert links one edge per layer and never reads ef_construction, hnsw.zig:59,:106-108 -- this call gives the
 * reference's parameters the meaning they have in HNSW proper).  Specification: the oracle (orc_bulk_level,
 * orc_graph_select, orc_graph_build_layer0), matched bit for bit:
 *   levels   level(i) = trailing one-bits of mix64(level_seed, i), cap 31 (hnsw.zig:172-180's distribution);
 *   layer 0  candidates = each node's min(ef_construction, 96) exact nearest neighbours; forward links by the HNSW
 *            neighbour-selection heuristic with keep-pruned fill (cap M); reverse links added; the union is cut to
 *            2 M links with the same heuristic (graph_stride >= 2 M; M <= 32);
 *   layer L  the M exact nearest among the nodes whose level reaches L;
 *   entry    the first node of the highest level (zvdb_hip_graph_info).
 * `mode` = enum zvdb_bf_mode of the screen behind the nearest-neighbour lists: the exact screens (ZVDB_BF_F32 / _BF16 / _I8)
 * give the oracle's graph (list for list at every tested size; the builder's passes run WITHOUT the exactness certificate
 * -- k = 65 .. 97 leaves it too few survivors -- so on very large wide shards the _BF16 / _I8 lists can differ from the
 * exact nearest neighbours in a few entries: DESIGN 3 C); ZVDB_BF_FP4 (approximate candidates, exact re-rank) builds wide shards faster -- beyond 768
 * dimensions the exact screens run on the staged kernels: 12.5M x 1536 bf16 rows build in 226 s instead of 335 and answer the
 * same recall (profiles/r3/).
 * Replaces every layer of the index's graph.  Search it with zvdb_hip_search_graph(ef > 0, entry = ZVDB_ENTRY_DESCEND):
 * a greedy descent through the upper layers picks each query's layer-0 entry, then the ef-mode beam search runs. */
int zvdb_hip_build_graph(zvdb_hip_index *h, uint32_t M, uint32_t ef_construction, uint32_t mode, uint64_t level_seed);
int zvdb_hip_graph_info(zvdb_hip_index *h, uint64_t *out_entry, uint32_t *out_max_level);
/* On-disk index format (the reference is memory-only): one flat little-endian file -- a 256-byte header, then
 * the device arrays as they sit in HBM (rows as stored: fp32 or bf16; norms; every layer's adjacency; levels), each section
 * 256-byte aligned, so a loader may mmap the file and upload the sections as they are.  Lazily built mirrors are not stored.
 * zvdb_hip_load creates a new index (device = HIP ordinal, -1 = current) holding exactly what was saved: the same rows, graph
 * (reference-compat or bulk, incl. its entry) and therefore the same search results. */
int zvdb_hip_save(zvdb_hip_index *h, const char *path);
int zvdb_hip_load(const char *path, int32_t device, zvdb_hip_index **out);
/* HBM accounting: bytes of one stored row (fp32: 4 ld; ZVDB_FLAG_BF16_ROWS: 2 ldh) and the bytes this index holds on the device
 * right now (row storage, norms, adjacency of every layer, and whichever lazily built mirrors exist). */
int zvdb_hip_storage_info(zvdb_hip_index *h, uint64_t *out_row_bytes, uint64_t *out_resident_bytes);
/* connections[layer] of a node on the device (layer 0: zvdb_hip_get_neighbors); NODE_NOT_FOUND if the node lacks the layer */
int zvdb_hip_get_neighbors_layer(zvdb_hip_index *h, uint64_t node, uint32_t layer, uint64_t *out_ids, uint32_t cap, uint32_t *out_n);
/* `entry` value of the graph searches: per-query entry = hierarchical descent from the bulk graph's top node */
#define ZVDB_ENTRY_DESCEND 0xFFFFFFFFFFFFFFFFull

/* ---- a1/a2/a3 seam: distance(query, rows[ids]) (hnsw.zig:86,:96,:160-161) -- */
/* out[i] = distance(query, row ids[i]) in EXACT reference arithmetic (strict
 * sequential fp32 sub/mul/add, no FMA).  `query` is host memory of dim floats. */
int zvdb_hip_distances(zvdb_hip_index *h, const float *query, uint32_t dim,
                       const uint64_t *ids, uint32_t n, float *out);
/* Same with the query being row `owner` of the index (insert's new point,
 * shrinkConnections' owner). */
int zvdb_hip_distances_from(zvdb_hip_index *h, uint64_t owner, const uint64_t *ids, uint32_t n,
                            float *out);
/* a2: the whole layer-0 greedy descent of insert (hnsw.zig:88-104 at layer 0)
 * for query row `owner`, starting at (*ep, *ep_dist); returns the final pair
 * and the number of distance evaluations. */
/* insert's device work for layer 0, fused into ONE launch and one synchronisation (the separate calls above cost a
 * synchronous launch each): appends `point` (hnsw.zig:77-82; id = count), then, unless entry == UINT64_MAX (first node),
 * distance(point, entry) (:86), the greedy scan at layer 0 (:88-104) and connect(id, ep, 0) (:106-108, :119-141) with
 * shrinkConnections (:143-170) of both lists, on the device mirror of layer 0.  Outputs what the host mirror adopts:
 * the arg-min ep and its distance (the ep the upper layers continue from), ep's list after the connect (out_ep_list:
 * room for m ids) and the length of the new node's own list ([ep] unless m == 0). */
int zvdb_hip_insert_layer0(zvdb_hip_index *h, const float *point, uint32_t dim, uint64_t entry, uint64_t *out_id,
                           uint64_t *out_ep, float *out_ep_dist, uint64_t *out_ep_list, uint32_t *out_ep_n,
                           uint32_t *out_owner_n, uint64_t *out_evals);
/* n of the above in ONE launch (extension: the reference inserts one point per call): the inserts run one after the
 * other on the device, each seeing the lists its predecessors left -- the result of n single calls, without their
 * launches and host round trips.  Needs a non-empty index (entry = its entry point).  Outputs are arrays of n
 * (out_ep_lists: n x m ids, row i holds out_ep_n[i] entries). */
int zvdb_hip_insert_batch_layer0(zvdb_hip_index *h, const float *points, uint32_t dim, uint64_t n, uint64_t entry,
                                 uint64_t *out_first_id, uint64_t *out_ep, float *out_ep_dist, uint64_t *out_ep_lists,
                                 uint32_t *out_ep_n, uint32_t *out_owner_n);
/* The WHOLE of insert (hnsw.zig:73-117), every layer, for n points per call: the device holds each node's level and all
 * of its lists, and one launch per chunk of 8,192 points runs the inserts -- append the point, then per layer
 * 0..max_level the greedy scan (:90-104) and, up to the point's own level, connect (:106-108) with both
 * shrinkConnections -- each insert seeing what its predecessors left: the graph of n reference inserts in this order.
 * (Chunks of more than 4 points run as speculative parallel rounds: a round's points are inserted concurrently against
 * the committed graph, everything each one read is validated against what the earlier ones of the round wrote, the valid
 * prefix commits in order and the rest runs again -- the same graph and the same records, bit for bit.)
 * levels[i] = the level randomLevel() (:172-180) drew for point i (the caller draws it: the binding owns the RNG).
 * The entry point is the first node (:84, :111).  For the host mirror: out_nrec[i] records for point i, one per layer
 * 0..min(level, max_level before it), each 4 + m words {ep, owner_n, ep_n, ep_has_layer, ep's list after the connect}:
 * connections[layer] of the new node becomes [ep] (empty if owner_n == 0), and, if ep_has_layer, that of ep becomes the
 * list.  recs_cap (in records) >= sum(levels[i] + 1).  An index fed through this call must be fed ONLY through it
 * (the step-by-step calls above keep upper layers on the host).  Plain fp32 L2 rows, m <= graph_stride. */
int zvdb_hip_insert_batch(zvdb_hip_index *h, const float *points, uint32_t dim, uint64_t n, const uint32_t *levels,
                          uint64_t *out_first_id, uint32_t *out_nrec, uint32_t *out_recs, uint64_t recs_cap);
int zvdb_hip_greedy_layer0(zvdb_hip_index *h, uint64_t owner, uint64_t *ep, float *ep_dist,
                           uint64_t *out_evals);

/* ---- a5/a6/a7: search (hnsw.zig:194-236) ---------------------------------- */
/* Batched graph search over the device mirror of layer 0, entry point = node
 * `entry` (the reference never moves it off node 0: hnsw.zig:84,:110-112).
 *   ef == 0 : reference-compat mode -- first k pops of a best-first frontier,
 *             then stable sort (bit-exact ids/distances vs the oracle);
 *   ef  > 0 : ef-mode beam search (extension; spec in oracle/zvdb_oracle.h).  The per-query visited
 *             set lives in LDS, sized from the fills the index has seen for this ef (a query that
 *             outgrows it is re-run by a second launch with the safe size; beyond ef = 256 its
 *             entries are 21 bits wide behind a bijection of the ids -- exact membership, three
 *             to a 64-bit word: twice the queries per CU at ef = 512 / 1,024); when the safe size
 *             does not fit (ef * stride beyond ~12k) that launch keeps the visited sets in an
 *             index-owned device buffer of up to 2 GiB, allocated on first need; otherwise a last
 *             16-workgroup launch serves the rare query that fills even the safe LDS table from a
 *             small index-owned pool of global tables (16 tables of >= 16 ef stride slots: 4 MiB at
 *             ef <= 256 / stride 16, 32 MiB at ef = 1024 / stride 32).  Searches of one index that
 *             are launched on different streams are ordered among themselves (they share that
 *             scratch); searches of different indexes are independent.
 * Outputs: out_ids[nq*k], out_dist[nq*k], out_count[nq] (count may be < k:
 * test_hnsw.zig:122-125; unused slots are id = UINT64_MAX, dist = +inf).
 * stats may be NULL. */
int zvdb_hip_search_graph(zvdb_hip_index *h, const float *queries, uint32_t nq, uint32_t dim,
                          uint32_t k, uint32_t ef, uint64_t entry,
                          uint64_t *out_ids, float *out_dist, uint32_t *out_count,
                          zvdb_hip_search_stats *stats);

/* ---- a10: brute-force Q x V^T + fused top-k (no reference symbol) --------- */
/* For each query the k smallest distances over all rows, ascending, ties by
 * lower id; final distances are recomputed in exact reference arithmetic.
 * ZVDB_BF_F32 / _BF16 / _I8 are CERTIFIED (ZVDB_OPT_CERTIFY, on by default): the ids are proven to be the exact
 * top-k per query, or recomputed by an exact scan inside the same call.
 * k <= 112 on every route and dimension (k <= 48, 56 with ZVDB_BF_F32, keep
 * k' = 64 survivors per query; larger k -- the approximate ZVDB_BF_FP4 screen
 * from k > 32 -- keep k' = 128); k <= 240 on the stream routes (ZVDB_BF_I8 and
 * ZVDB_BF_BF16 with dim <= 768, ZVDB_BF_F32 with dim <= 128 (<= 256 for batches of
 * more than 128 queries), ZVDB_BF_FP4 with dim <= 1536: k' = 256, the fp4 screen
 * from k > 64); else ZVDB_ERR_UNSUPPORTED. */
int zvdb_hip_search_bruteforce(zvdb_hip_index *h, const float *queries, uint32_t nq, uint32_t dim,
                               uint32_t k, uint32_t mode,
                               uint64_t *out_ids, float *out_dist, uint32_t *out_count);

/* ---- device-pointer variants (bench / multi-GPU; enqueue only) ------------ */
/* Workspace sizing for the _dev calls (bytes); allocate once, reuse. */
int zvdb_hip_bruteforce_workspace(zvdb_hip_index *h, uint32_t nq, uint32_t k, uint32_t mode,
                                  uint64_t *out_bytes);
/* d_queries: nq x dim fp32 row-major; outputs on device: d_ids u64[nq*k] (ids
 * already offset by id_offset: the shard's first global id), d_dist f32[nq*k],
 * d_count u32[nq]. */
int zvdb_hip_search_bruteforce_dev(zvdb_hip_index *h, const void *d_queries, uint32_t nq,
                                   uint32_t dim, uint32_t k, uint32_t mode, uint64_t id_offset,
                                   void *d_workspace, uint64_t workspace_bytes,
                                   void *d_ids, void *d_dist, void *d_count, void *stream);
/* The *_dev entry points take device pointers and launch on the CALLER's stream without a host round trip: results are
 * ready in stream order.  The handle's mutex covers the host side only -- a caller that mutates the index (add / insert)
 * while searches it launched are still running on other streams orders the two itself (the host-pointer calls always
 * return with their device work complete).
 * d_stats: NULL (no counters), or device memory for THREE unsigned 64-bit words that the CALLER zeroes before the launch
 * and the kernels add to: [0] distance evaluations E, [1] frontier pops P (the figures of SURVEY 8d's roofline,
 * bytes = E d sizeof(elt) + P M0 4), [2] ef mode only: queries whose visited set overflowed (their result is not to
 * spec; the host-pointer entry point turns a non-zero count into an error). */
int zvdb_hip_search_graph_dev(zvdb_hip_index *h, const void *d_queries, uint32_t nq, uint32_t dim,
                              uint32_t k, uint32_t ef, uint64_t entry, uint64_t id_offset,
                              void *d_ids, void *d_dist, void *d_count, void *d_stats,
                              void *stream);
/* Takes, ahead of time, every index-owned buffer a zvdb_hip_search_graph_dev call with up to nq queries, this k and this
 * ef (0 = reference-compat) allocates on first need -- so that the searches themselves only enqueue kernels (no hipMalloc,
 * no synchronise: e.g. under stream capture).  Buffers only grow; reserving again with smaller sizes is a no-op. */
int zvdb_hip_search_graph_reserve(zvdb_hip_index *h, uint32_t nq, uint32_t k, uint32_t ef);
/* C1 consumer: merge G all-gathered per-shard lists ([G][nq][k] ids/dist,
 * [G][nq] counts, device memory) into the global top-k by (dist, id). */
int zvdb_hip_merge_topk_dev(const void *d_ids, const void *d_dist, const void *d_count,
                            uint32_t G, uint32_t nq, uint32_t k,
                            void *d_out_ids, void *d_out_dist, void *d_out_count, void *stream);

/* Declares that this index holds ONE of `shards` id-range shards whose per-shard results are merged by
 * zvdb_hip_merge_topk*_dev (no reference counterpart: zvdb is single-node).  The approximate 4-bit screen
 * (ZVDB_BF_FP4, k <= 32) then keeps this shard's SHARE of the 64 survivors per query instead of all 64 --
 * ceil(64/G + 6 sqrt(64 (1/G)(1 - 1/G))), at least k + 8: a true neighbour inside the global 64 best screen keys is
 * lost only if its shard holds more than that many of them (a 6-sigma event for uniformly spread ids) -- which
 * shortens the fixed per-batch work (appends, merge, exact re-rank) that bounds strong scaling.  Exact screens are
 * unaffected.  shards = 1 (default) restores the full budget. */
int zvdb_hip_index_set_shard_count(zvdb_hip_index *h, uint32_t shards);

/* Survivors of the approximate ZVDB_BF_FP4 screen that are re-ranked exactly, per query: 0 = chosen from k (64 for
 * k <= 32, 128 for k <= 64, 256 above), or at least 64 / 128 / 256.  More survivors buy recall on data whose neighbours
 * differ by less than the 4-bit step resolves (e.g. tight clusters far apart), for a longer re-rank; recall is measured,
 * never guaranteed, on this route -- ZVDB_BF_I8 / _BF16 / _F32 are the exact ones. */
int zvdb_hip_index_set_screen_survivors(zvdb_hip_index *h, uint32_t survivors);

/* Index options (behaviour that used to hang on environment variables lives here: an embedded library is configured by its
 * caller, not by the process environment). */
enum zvdb_option {
    /* 1 (default) / 0: certify the results of the exact brute-force screens (ZVDB_BF_F32 / _BF16 / _I8).  With it every query's
     * result is PROVEN to be the k smallest reference distances (hnsw.zig:182-192; ties by lower id) or recomputed: the re-rank
     * compares the k-th exact distance with the k'-th survivor's screen key through a bound on everything the screen rounds
     * (bf16 operands: residual norms measured when the mirror is built; int8 codes likewise; MFMA accumulation: one ulp per
     * step in any order; the reference sum's own rounding) -- zvdb_amd/csrc/bf_cert.h states the inequality.  A query the
     * bound cannot certify is recomputed inside the same call by an exact scan of every row in the reference's arithmetic
     * (no screen involved), and the index keeps more survivors (64, 128, 256) for the following batches so that the proof
     * goes through without it.  Without it the exact screens return the true top-k "whenever it is among the k' best screen keys",
     * unchecked.  ZVDB_BF_FP4 is approximate by design and never certified. */
    ZVDB_OPT_CERTIFY = 1,
    /* 0 .. 3: survivors kept per query from the first batch on (0: as few as the route allows and adapt; 1: 64; 2: 128; 3: 256) */
    ZVDB_OPT_CERT_MIN_LEVEL = 2,
    /* 1: every query is sent through the exact scan whatever the certificate says (tests of the repair path; results are the
     * same by definition, only slower) */
    ZVDB_OPT_CERT_FORCE_REPAIR = 3,
    /* 1 (default) .. 4: SEARCH WIDTH of the ef-mode beam search -- the number of unexpanded beam entries taken per step: all of them
     * are marked expanded, then their neighbour lists are read TOGETHER (width lists' worth of row fetches per dependent step
     * instead of one list's: the search is a chain of dependent steps, so on a GPU its rate is bytes in flight per step).
     * Specification: orc_hnsw_search_ef_w in oracle/zvdb_oracle.h, matched bit for bit (ids, distances, E / P counters); width 1
     * is the plain beam search.  Recall at equal ef is the same to within noise; E grows by a few per cent (nodes that a
     * step's own insertions would have pushed past position ef are expanded anyway).  No reference counterpart (hnsw.zig:211-224
     * pops one node at a time); ef = 0 (reference-compat search) ignores it. */
    ZVDB_OPT_EF_SEARCH_WIDTH = 4
};
int zvdb_hip_index_set_option(zvdb_hip_index *h, uint32_t option, uint64_t value);

/* Tuning switches: the A/B and test knobs of the launch code (kernel variants, forced thresholds, visited-table sizes, ...: the
 * table "Runtime switches" in DESIGN.md lists them) are named string values in a process-wide table that the CALLER fills.
 * They select between implementations with IDENTICAL results (tests hold them to that) -- none is needed for normal use.
 * The library never reads the process environment (a diagnostic build with -DZVDB_TUNING_FROM_ENV falls back to getenv for
 * names that were never set).  value NULL removes the entry.  Switches are read when a call is made. */
int zvdb_hip_set_tuning(const char *name, const char *value);
const char *zvdb_hip_get_tuning(const char *name); /* NULL if unset; valid until a few further calls on this thread */

typedef struct zvdb_hip_cert_info {
    uint64_t queries;  /* queries that went through the certificate */
    uint64_t batches;
    uint64_t repaired; /* of them: recomputed by the exact scan (read without synchronising: launches still running are not counted) */
    uint32_t level;    /* survivors currently kept: 0 = the route's minimum (32 for k <= 16, else 64), 1 = 64, 2 = 128, 3 = 256 */
    uint32_t enabled;
    uint64_t rerank_rows; /* diagnostic (tuning ZVDB_CERT_DEBUG=1): candidate rows the re-rank evaluated exactly, modulo 2^32 */
    uint32_t rerank_cycles[5]; /* diagnostic: 100-MHz ticks of every 64th query's re-rank, summed per phase: query constants, first
                                  (staged) batch, further candidates, final bound, ranking + output */
    uint32_t guess_misses; /* queries whose guessed threshold was too tight (repaired by the exact scan when a batch has <= 32 of them -- those also
                              count under `repaired` -- else by one gated re-scan of the corpus); they do not raise `level` */
    uint32_t bound_checked;    /* diagnostic (tuning ZVDB_CERT_DEBUG=3): evaluated candidates whose screen key was held against the bound of */
    uint32_t bound_violations; /* their own exact distance, and how many exceeded it (must be 0: the inequality of bf_cert.h on real hardware) */
} zvdb_hip_cert_info;
int zvdb_hip_bruteforce_cert_info(zvdb_hip_index *h, zvdb_hip_cert_info *out);

/* Same merge over ONE all-gathered byte buffer: rank g's chunk starts at g * rank_stride_bytes and
 * holds ids u64[nq*k] at offset 0, dist f32[nq*k] at nq*k*8, count u32[nq] at nq*k*12 (so a rank
 * sends its three result arrays with a single collective). */
int zvdb_hip_merge_topk_packed_dev(const void *d_packed, uint64_t rank_stride_bytes, uint32_t G,
                                   uint32_t nq, uint32_t k, void *d_out_ids, void *d_out_dist,
                                   void *d_out_count, void *stream);

/* ---- the multi-GPU step behind the ABI (the reference is single-node: no counterpart) -------------------
 * One process per GPU, the index sharded by vector-id range (ids are dense insertion order, hnsw.zig:77; a shard's first
 * global id is its id_offset).  A communicator wraps one RCCL rank: rank 0 obtains the unique id (zvdb_hip_comm_unique_id)
 * and hands its ZVDB_COMM_ID_BYTES bytes to the other ranks by any channel it has (a file, a socket, MPI, torch.distributed);
 * every rank then calls zvdb_hip_comm_create (collective: returns once all `world` ranks have joined).  RCCL is bound at run
 * time (librccl.so.1, xGMI inside a node); ZVDB_ERR_UNSUPPORTED if it cannot be loaded.
 * The sharded searches run the local search on this rank's shard, ONE all-gather of the packed per-shard top-k
 * (nq k 12 + nq 4 bytes per rank) and the k-way merge by (distance, id) -- everything enqueued on the caller's stream, no host
 * round trip; every rank ends with the same global top-k in d_ids / d_dist / d_count (layouts of the unsharded *_dev calls).
 * Exactness: the union of exact per-shard top-k lists contains the exact global top-k, so the brute-force routes stay exact
 * (and certified) when sharded; the graph route's recall is the per-shard recall.
 * A communicator serves one call at a time (RCCL's rule for a rank): callers that search from several threads use one
 * communicator per thread or serialise the sharded calls themselves. */
typedef struct zvdb_hip_comm zvdb_hip_comm; /* opaque */
#define ZVDB_COMM_ID_BYTES 128
#define ZVDB_SHARDED_GRAPH 0xFFFFFFFFu /* `mode` of zvdb_hip_sharded_workspace for the graph route */
int zvdb_hip_comm_unique_id(void *out_id /* ZVDB_COMM_ID_BYTES bytes */);
int zvdb_hip_comm_create(uint32_t rank, uint32_t world, const void *unique_id, int32_t device /* -1 = current */, zvdb_hip_comm **out);
void zvdb_hip_comm_destroy(zvdb_hip_comm *c);
int zvdb_hip_comm_info(zvdb_hip_comm *c, uint32_t *out_rank, uint32_t *out_world);
/* Everything zvdb_hip_comm_create can refuse on ONE rank (RCCL not loadable, no device, bad device index), without the collective:
 * call it on every rank and agree on the outcome over the host's own channel BEFORE any rank enters zvdb_hip_comm_create -- a
 * rank that failed alone would leave the others waiting in ncclCommInitRank. */
int zvdb_hip_comm_precheck(int32_t device /* -1 = current */);
/* What RCCL itself reports for this communicator (ncclCommUserRank / ncclCommCount) and the PCI bus id ("0000:05:00.0") of its
 * device: gathered over the ranks, the evidence that N ranks on N distinct GPUs joined. */
int zvdb_hip_comm_query(zvdb_hip_comm *c, uint32_t *out_rccl_rank, uint32_t *out_rccl_ranks, char *out_bus_id, uint32_t bus_id_bytes);
/* HIP events around the ncclAllGather of the sharded calls, on the caller's stream (up to 64 samples): enable 1 = new measurement,
 * 2 = resume, 0 = pause; read synchronises on the recorded events and returns their summed microseconds. */
int zvdb_hip_comm_timing_enable(zvdb_hip_comm *c, int on);
int zvdb_hip_comm_timing_read(zvdb_hip_comm *c, double *out_us_total, uint32_t *out_samples);
/* bytes of device workspace the sharded calls need (packed send + receive buffers + the local search's own workspace) */
int zvdb_hip_sharded_workspace(zvdb_hip_index *h, zvdb_hip_comm *c, uint32_t nq, uint32_t k, uint32_t mode, uint64_t *out_bytes);
int zvdb_hip_search_bruteforce_sharded_dev(zvdb_hip_index *h, zvdb_hip_comm *c, const void *d_queries, uint32_t nq, uint32_t dim,
                                           uint32_t k, uint32_t mode, uint64_t id_offset, void *d_workspace, uint64_t workspace_bytes,
                                           void *d_ids, void *d_dist, void *d_count, void *stream);
int zvdb_hip_search_graph_sharded_dev(zvdb_hip_index *h, zvdb_hip_comm *c, const void *d_queries, uint32_t nq, uint32_t dim,
                                      uint32_t k, uint32_t ef, uint64_t entry, uint64_t id_offset, void *d_workspace,
                                      uint64_t workspace_bytes, void *d_ids, void *d_dist, void *d_count, void *d_stats, void *stream);

/* Kernel timing hook for bench.py's roofline object: the library records HIP
 * events around the dominant kernel of the last *_dev brute-force/graph call
 * on the stream it was launched on; this returns the elapsed ms of the last
 * `n` recorded launches (synchronises on the events).
 * enable: 1 = start a new measurement, 2 = resume (sampled timing: an event
 * pair costs two ~6 us bubbles in the launch chain), 0 = pause (samples kept). */
int zvdb_hip_timing_enable(zvdb_hip_index *h, int enable);
int zvdb_hip_timing_read(zvdb_hip_index *h, double *out_total_ms, uint32_t *out_launches);

#ifdef __cplusplus
}
#endif
#endif /* ZVDB_HIP_H */
