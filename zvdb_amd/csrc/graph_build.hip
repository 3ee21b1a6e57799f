// graph_build.hip -- bulk construction of a recall-competent HNSW graph on the device.
//
// NOT reference parity: the reference insert links ONE edge per layer (hnsw.zig:106-108), keeps the entry point at node 0
// (:84, :110-112) and never reads ef_construction (:59) -- a graph whose recall@10 is ~0.04.  This builder keeps
// the reference's parameters (m, ef_construction, the level distribution of randomLevel :172-180) and gives them the
// meaning they have in HNSW proper; its specification is the oracle (oracle/zvdb_oracle.c: orc_bulk_level,
// orc_graph_select, orc_graph_build_layer0, build_bulk_graph in oracle/__init__.py), matched bit for bit (tested).
//
//   levels   level(i) = trailing one-bits of mix64(seed, i), cap 31          (each further level with probability 1/2)
//   layer 0  F[u] = the ef_construction exact nearest neighbours of u         (brute-force MFMA route, exact re-rank)
//            S[u] = select(u, F[u], cap M)                                    forward links: HNSW's neighbour heuristic
//            C[v] = S[v] u { u : v in S[u] }  sorted by (distance, id), first cmax   + reverse links
//            L0[v] = select(v, C[v], cap M0 = 2 M)                             degree cap with the same heuristic
//   layer L  the M exact nearest among the nodes whose level reaches L         (upper layers: navigation only)
//   entry    the first node of the highest level
//
// select(base, candidates ascending by (d(base, c), c), cap): walk the candidates while fewer than cap are kept; c is kept
// iff d(base, c) < d(c, s) for every s kept so far (strict), else set aside; the set-aside ones then fill up to cap in order.
// Every distance is the reference arithmetic (hnsw.zig:182-192: exact_dist, one lane walks one row sequentially), so
// the comparisons -- and with them the graph -- are those of the oracle.  One wave per node; a candidate row is staged in
// LDS, the kept rows (<= 64, one per lane) are gathered from L2 / HBM: a gather-bound kernel (bytes ~ n_cand x n_kept x row).
#include "common.h"
#include "wave_select.h"
#include "exact_dist.h"

#pragma clang fp contract(off)

namespace zvdb {

namespace {

constexpr uint32_t GB_CMAX = 128;  // candidates entering the final selection (S[v] + nearest reverse links)

// brute-force results of rows [r0, r0 + n) -> candidate lists (self dropped; at most pool entries, nearest first)
__global__ __launch_bounds__(256) void k_gb_cands(const uint64_t *__restrict__ ids, const float *__restrict__ dist, const uint32_t *__restrict__ cnt,
                                                  uint32_t k, uint64_t r0, uint32_t n, uint32_t pool, uint64_t id_bias /* ids are local + bias */,
                                                  uint32_t *__restrict__ c_ids, float *__restrict__ c_dist, uint32_t *__restrict__ c_n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t self = r0 + i;
    uint32_t w = 0;
    const uint32_t c = min(cnt[i], k);
    for (uint32_t j = 0; j < c && w < pool; ++j) {
        const uint64_t id = ids[(size_t)i * k + j] - id_bias;
        if (id != self) {
            c_ids[self * pool + w] = (uint32_t)id;
            c_dist[self * pool + w] = dist[(size_t)i * k + j];
            ++w;
        }
    }
    c_n[self] = w;
}

// select() of the header on candidates held in LDS (cid / cdist [n], ascending by (distance, id)); the result (ids, distances)
// goes to kid / kdist [cap] in LDS; returns its length (wave-uniform).  q = LDS staging of one row (dim floats).
template <typename RT>
__device__ __forceinline__ uint32_t gb_select(const RT *rows, uint32_t ld, uint32_t dim, int metric, const uint32_t *cid, const float *cdist,
                                              uint32_t n, uint32_t cap, uint32_t *kid, float *kdist, uint32_t *aside, float *q, int lane) {
    uint32_t nk = 0, na = 0;
    for (uint32_t i = 0; i < n && nk < cap; ++i) {
        const uint32_t c = cid[i];
        const float dc = cdist[i];
        bool bad = false;
        if (nk > 0) {
            __syncthreads();  // (one wave per block: orders the LDS staging against the previous step's reads)
            const RT *crow = rows + (size_t)c * ld;
            for (uint32_t e = lane; e < dim; e += kWave) q[e] = row_elem(crow, e);
            __syncthreads();
            if ((uint32_t)lane < nk) {
                const float dcs = exact_dist_rt(metric, q, rows + (size_t)kid[lane] * ld, dim);
                bad = !(dc < dcs);
            }
        }
        if (__ballot(bad) == 0) {
            if (lane == 0) { kid[nk] = c; kdist[nk] = dc; }
            nk += 1;
        } else {
            if (lane == 0) aside[na] = i;
            na += 1;
        }
        __syncthreads();
    }
    for (uint32_t i = 0; i < na && nk < cap; ++i) {
        if (lane == 0) { const uint32_t a = aside[i]; kid[nk] = cid[a]; kdist[nk] = cdist[a]; }
        nk += 1;
    }
    __syncthreads();
    return nk;
}

// forward links S[u] = select(u, F[u], M); counts the reverse links every target will receive
template <typename RT>
__global__ __launch_bounds__(64) void k_gb_select_fwd(const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric, uint32_t N,
                                                      const uint32_t *__restrict__ c_ids, const float *__restrict__ c_dist,
                                                      const uint32_t *__restrict__ c_n, uint32_t pool, uint32_t M,
                                                      uint32_t *__restrict__ s_ids, float *__restrict__ s_dist, uint32_t *__restrict__ s_n,
                                                      uint32_t *__restrict__ r_cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *q = reinterpret_cast<float *>(smem);                      // [dim_pad]
    const uint32_t dim_pad = (dim + 3) & ~3u;
    uint32_t *cid = reinterpret_cast<uint32_t *>(q + dim_pad);       // [GB_CMAX]
    float *cdist = reinterpret_cast<float *>(cid + GB_CMAX);
    uint32_t *aside = reinterpret_cast<uint32_t *>(cdist + GB_CMAX);
    uint32_t *kid = aside + GB_CMAX;                                 // [64]
    float *kdist = reinterpret_cast<float *>(kid + 64);
    const int lane = threadIdx.x;
    const uint32_t u = blockIdx.x;
    if (u >= N) return;
    const uint32_t n = min(c_n[u], pool);
    for (uint32_t i = lane; i < n; i += kWave) { cid[i] = c_ids[(size_t)u * pool + i]; cdist[i] = c_dist[(size_t)u * pool + i]; }
    __syncthreads();
    const uint32_t nk = gb_select(rows, ld, dim, metric, cid, cdist, n, M, kid, kdist, aside, q, lane);
    if ((uint32_t)lane < nk) {
        s_ids[(size_t)u * M + lane] = kid[lane];
        s_dist[(size_t)u * M + lane] = kdist[lane];
        atomicAdd(r_cnt + kid[lane], 1u);
    }
    if (lane == 0) s_n[u] = nk;
}

// reverse links into CSR slots (order within a node's slots is arbitrary: the final selection sorts by (distance, id))
__global__ __launch_bounds__(256) void k_gb_fill_rev(uint32_t N, uint32_t M, const uint32_t *__restrict__ s_ids, const float *__restrict__ s_dist,
                                                     const uint32_t *__restrict__ s_n, const uint64_t *__restrict__ r_off, uint32_t *__restrict__ r_cur,
                                                     uint32_t *__restrict__ r_src, float *__restrict__ r_dist) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t u = t / M;
    const uint32_t j = (uint32_t)(t % M);
    if (u >= N || j >= s_n[u]) return;
    const uint32_t v = s_ids[u * M + j];
    const uint32_t pos = atomicAdd(r_cur + v, 1u);
    r_src[r_off[v] + pos] = (uint32_t)u;
    r_dist[r_off[v] + pos] = s_dist[u * M + j];
}

// L0[v] = select(v, first cmax of (S[v] u reverse links) by (distance, id), M0)
template <typename RT>
__global__ __launch_bounds__(64) void k_gb_select_final(const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric, uint32_t N, uint32_t M,
                                                        uint32_t M0, const uint32_t *__restrict__ s_ids, const float *__restrict__ s_dist,
                                                        const uint32_t *__restrict__ s_n, const uint64_t *__restrict__ r_off,
                                                        const uint32_t *__restrict__ r_src, const float *__restrict__ r_dist,
                                                        uint32_t *__restrict__ adj, uint32_t *__restrict__ deg, uint32_t gstride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *q = reinterpret_cast<float *>(smem);
    const uint32_t dim_pad = (dim + 3) & ~3u;
    uint32_t *cid = reinterpret_cast<uint32_t *>(q + dim_pad);
    float *cdist = reinterpret_cast<float *>(cid + GB_CMAX);
    uint32_t *aside = reinterpret_cast<uint32_t *>(cdist + GB_CMAX);
    uint32_t *kid = aside + GB_CMAX;
    float *kdist = reinterpret_cast<float *>(kid + 64);
    uint32_t *sown = reinterpret_cast<uint32_t *>(kdist + 64);       // [64] S[v] ids (duplicate test)
    uint64_t *pool = reinterpret_cast<uint64_t *>(sown + 64);        // [GB_CMAX] selected keys
    uint32_t *hist = reinterpret_cast<uint32_t *>(pool + GB_CMAX);   // [256]
    const int lane = threadIdx.x;
    const uint32_t v = blockIdx.x;
    if (v >= N) return;
    const uint32_t ns = min(s_n[v], M);
    const uint64_t ro = r_off[v];
    const uint32_t nr = (uint32_t)(r_off[v + 1] - ro);
    if ((uint32_t)lane < ns) sown[lane] = s_ids[(size_t)v * M + lane];
    __syncthreads();
    const uint32_t nall = ns + nr, E = (nall + kWave - 1) / kWave;
    auto key_at = [&](uint32_t e) -> uint64_t {  // (f32_key(distance) << 32) | id ; ~0: nothing / a duplicate of S[v]
        const uint32_t i = e * kWave + lane;
        if (i < ns) return ((uint64_t)f32_key(s_dist[(size_t)v * M + i]) << 32) | sown[i];
        if (i >= nall) return ~0ull;
        const uint32_t src = r_src[ro + (i - ns)];
        for (uint32_t t = 0; t < ns; ++t)
            if (sown[t] == src) return ~0ull;
        return ((uint64_t)f32_key(r_dist[ro + (i - ns)]) << 32) | src;
    };
    uint32_t nvl = 0;
    for (uint32_t e = 0; e < E; ++e) nvl += key_at(e) != ~0ull ? 1u : 0u;
    uint32_t nv = nvl;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nv += __shfl_xor(nv, off);
    uint64_t T = ~0ull - 1;
    if (nv > GB_CMAX) T = wave_kth_key<0>(key_at, E, GB_CMAX, hist, lane);
    // compact the keys <= T (at most cmax of them: keys are distinct) into LDS
    uint32_t base = 0;
    for (uint32_t e = 0; e < E; ++e) {
        const uint64_t k = key_at(e);
        const bool selv = k <= T;
        const unsigned long long mk = __ballot(selv);
        const uint32_t pos = base + (uint32_t)__popcll(mk & ((1ull << lane) - 1));
        if (selv && pos < GB_CMAX) pool[pos] = k;
        base += (uint32_t)__popcll(mk);
    }
    const uint32_t n = min(base, GB_CMAX);
    __syncthreads();
    // ascending order by rank counting
    for (uint32_t i = lane; i < n; i += kWave) {
        const uint64_t k = pool[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += pool[j] < k ? 1u : 0u;
        cid[rank] = (uint32_t)k;
        cdist[rank] = f32_unkey((uint32_t)(k >> 32));
    }
    __syncthreads();
    const uint32_t nk = gb_select(rows, ld, dim, metric, cid, cdist, n, M0, kid, kdist, aside, q, lane);
    if ((uint32_t)lane < nk) adj[(size_t)v * gstride + lane] = kid[lane];
    if (lane == 0) deg[v] = nk;
}

// rows of a layer's nodes -> dense [n][dim] (the layer is searched as an index of its own)
// (nodes == nullptr: rows [first, first + n) in place -- the queries of a self-kNN batch, widened to fp32)
template <typename RT>
__global__ __launch_bounds__(256) void k_gb_gather_rows(const RT *__restrict__ rows, uint32_t ld, uint32_t dim, const uint32_t *__restrict__ nodes,
                                                        uint64_t first, uint64_t n, float *__restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t r = t / dim;
    if (r >= n) return;
    const uint32_t c = (uint32_t)(t % dim);
    out[r * dim + c] = row_elem(rows + (size_t)(nodes ? nodes[r] : first + r) * ld, c);
}

// upper layer L: search results of the layer's rows [r0, r0 + n) (LOCAL ids) -> the nodes' lists at layer L
__global__ __launch_bounds__(256) void k_gb_upper_write(const uint64_t *__restrict__ ids, const uint32_t *__restrict__ cnt, uint32_t k, uint64_t r0,
                                                        uint32_t n, uint32_t M, const uint32_t *__restrict__ nodes, const uint32_t *__restrict__ slot,
                                                        uint32_t layer, uint32_t *__restrict__ adj_up, uint32_t *__restrict__ deg_up, uint32_t gstride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t self = r0 + i;
    const uint32_t node = nodes[self];
    const size_t row = (size_t)slot[node] + layer - 1;
    uint32_t w = 0;
    const uint32_t c = min(cnt[i], k);
    for (uint32_t j = 0; j < c && w < M; ++j) {
        const uint64_t id = ids[(size_t)i * k + j];
        if (id != self) adj_up[row * gstride + w++] = nodes[id];
    }
    deg_up[row] = w;
}

struct DevBuf {  // scoped device allocation
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 256));
        if (e != hipSuccess) { set_error("hipMalloc(%zu) failed in build_graph: %s", bytes, hipGetErrorString(e)); p = nullptr; return ZVDB_ERR_OOM; }
        return ZVDB_OK;
    }
    template <class T> T *as() { return static_cast<T *>(p); }
};

size_t gb_lds_bytes(uint32_t dim) {
    const size_t dim_pad = (dim + 3) & ~3u;
    return dim_pad * 4 + GB_CMAX * 12 + 64 * 8 + 64 * 4 + GB_CMAX * 8 + 256 * 4;
}

// exact kNN (self excluded) of every row of `src` among its own rows, in batches; consume(d_ids, d_dist, d_cnt, k, r0, n)
template <class F>
int gb_self_knn(zvdb_hip_index *src, uint32_t k, uint32_t mode, F consume) {
    const uint32_t B = 1024;
    CertPause no_cert(src);  // (common.h: the builders' kNN passes run uncertified)
    const size_t ib = round_up64((size_t)B * k * 8, 256), db = round_up64((size_t)B * k * 4, 256);
    const size_t cb = round_up64((size_t)B * 4, 256), qb = round_up64((size_t)B * src->dim * 4, 256);
    const uint64_t wsb = bruteforce_workspace_bytes(src, B, k, mode);
    ZV_CHECK(ensure_scratch(src, qb + ib + db + cb + wsb));
    unsigned char *scr = static_cast<unsigned char *>(src->scratch);
    float *d_q = reinterpret_cast<float *>(scr);
    uint64_t *d_ids = reinterpret_cast<uint64_t *>(scr + qb);
    float *d_dist = reinterpret_cast<float *>(scr + qb + ib);
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(scr + qb + ib + db);
    for (uint64_t r0 = 0; r0 < src->count; r0 += B) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(B, src->count - r0);
        ZV_CHECK(launch_rows_to_f32(src, r0, n, d_q, src->stream));
        ZV_CHECK(launch_bruteforce(src, d_q, n, k, mode, 0, scr + qb + ib + db + cb, wsb, d_ids, d_dist, d_cnt, src->stream));
        ZV_CHECK(consume(d_ids, d_dist, d_cnt, k, r0, n));
    }
    return ZVDB_OK;
}

}  // namespace

// rows [first, first + n) of the index as dense fp32 [n][dim] (queries of the self-kNN passes: graph builders)
int launch_rows_to_f32(const zvdb_hip_index *h, uint64_t first, uint64_t n, float *d_out, hipStream_t s) {
    if (n == 0) return ZVDB_OK;
    return rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        const uint64_t threads = n * h->dim;
        hipLaunchKernelGGL(k_gb_gather_rows<row_elem_t<decltype(rows)>>, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, rows, ld, h->dim,
                           (const uint32_t *)nullptr, first, n, d_out);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    });
}

uint32_t bulk_level(uint64_t seed, uint64_t id) {  // oracle: orc_bulk_level
    const uint64_t x = mix64(seed, id);
    uint32_t level = 0;
    while (level < 31 && ((x >> level) & 1ull)) ++level;
    return level;
}

// the body of zvdb_hip_build_graph (api.hip holds the mutex and has validated the arguments)
int build_bulk_graph(zvdb_hip_index *h, uint32_t M, uint32_t ef_construction, uint32_t mode, uint64_t level_seed,
                     int (*ensure_upper_fn)(zvdb_hip_index *, uint64_t)) {
    const uint64_t N = h->count;
    hipStream_t s = h->stream;
    const uint32_t M0 = 2 * M;
    const uint32_t pool = (uint32_t)std::min<uint64_t>(std::min<uint32_t>(ef_construction, 96u), N - 1);
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;

    // ---- levels (host: the binding owns the RNG, hnsw.zig:172-180) and upper-layer slots ------------------------------
    std::vector<uint32_t> lv(N), sl(N);
    uint64_t up = 0;
    uint32_t maxlev = 0, top = 0;
    for (uint64_t i = 0; i < N; ++i) {
        lv[i] = bulk_level(level_seed, i);
        sl[i] = (uint32_t)up;
        up += lv[i];
        if (lv[i] > maxlev) { maxlev = lv[i]; top = (uint32_t)i; }
    }
    ZV_CHECK(ensure_upper_fn(h, up));
    ZV_HIP(hipMemcpyAsync(h->lvl, lv.data(), N * 4, hipMemcpyHostToDevice, s));
    ZV_HIP(hipMemcpyAsync(h->slot, sl.data(), N * 4, hipMemcpyHostToDevice, s));
    if (up) ZV_HIP(hipMemsetAsync(h->deg_up, 0, up * 4, s));
    ZV_HIP(hipStreamSynchronize(s));

    // ---- layer 0 ----------------------------------------------------------------------------------------------------
    DevBuf c_ids, c_dist, c_n, s_ids, s_dist, s_n, r_cnt, r_off, r_src, r_dist;
    ZV_CHECK(c_ids.alloc(N * pool * 4));
    ZV_CHECK(c_dist.alloc(N * pool * 4));
    ZV_CHECK(c_n.alloc(N * 4));
    ZV_CHECK(s_ids.alloc(N * M * 4));
    ZV_CHECK(s_dist.alloc(N * M * 4));
    ZV_CHECK(s_n.alloc(N * 4));
    ZV_CHECK(r_cnt.alloc(N * 4));
    ZV_CHECK(r_off.alloc((N + 1) * 8));
    ZV_HIP(hipMemsetAsync(r_cnt.p, 0, N * 4, s));
    if (pool > 0) {
        ZV_CHECK(gb_self_knn(h, pool + 1, mode, [&](uint64_t *d_ids, float *d_dist, uint32_t *d_cnt, uint32_t k, uint64_t r0, uint32_t n) -> int {
            hipLaunchKernelGGL(k_gb_cands, dim3((n + 255) / 256), dim3(256), 0, s, d_ids, d_dist, d_cnt, k, r0, n, pool, 0ull, c_ids.as<uint32_t>(),
                               c_dist.as<float>(), c_n.as<uint32_t>());
            ZV_HIP(hipGetLastError());
            return ZVDB_OK;
        }));
    } else {
        ZV_HIP(hipMemsetAsync(c_n.p, 0, N * 4, s));
    }
    const size_t lds = gb_lds_bytes(h->dim);
    ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = row_elem_t<decltype(rows)>;
        if (lds > 48 * 1024) {
            ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gb_select_fwd<RT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gb_select_final<RT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        hipLaunchKernelGGL(k_gb_select_fwd<RT>, dim3((uint32_t)N), dim3(64), lds, s, rows, ld, h->dim, metric, (uint32_t)N, c_ids.as<uint32_t>(),
                           c_dist.as<float>(), c_n.as<uint32_t>(), std::max(pool, 1u), M, s_ids.as<uint32_t>(), s_dist.as<float>(), s_n.as<uint32_t>(),
                           r_cnt.as<uint32_t>());
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    }));
    // CSR offsets of the reverse links (host scan: N words each way)
    std::vector<uint32_t> rc(N);
    ZV_HIP(hipMemcpyAsync(rc.data(), r_cnt.p, N * 4, hipMemcpyDeviceToHost, s));
    ZV_HIP(hipStreamSynchronize(s));
    std::vector<uint64_t> ro(N + 1);
    ro[0] = 0;
    for (uint64_t i = 0; i < N; ++i) ro[i + 1] = ro[i] + rc[i];
    const uint64_t total = ro[N];
    ZV_CHECK(r_src.alloc(total * 4));
    ZV_CHECK(r_dist.alloc(total * 4));
    ZV_HIP(hipMemcpyAsync(r_off.p, ro.data(), (N + 1) * 8, hipMemcpyHostToDevice, s));
    ZV_HIP(hipMemsetAsync(r_cnt.p, 0, N * 4, s));  // reused as the fill cursor
    {
        const uint64_t threads = N * M;
        hipLaunchKernelGGL(k_gb_fill_rev, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, (uint32_t)N, M, s_ids.as<uint32_t>(),
                           s_dist.as<float>(), s_n.as<uint32_t>(), r_off.as<uint64_t>(), r_cnt.as<uint32_t>(), r_src.as<uint32_t>(), r_dist.as<float>());
        ZV_HIP(hipGetLastError());
    }
    ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        hipLaunchKernelGGL(k_gb_select_final<row_elem_t<decltype(rows)>>, dim3((uint32_t)N), dim3(64), lds, s, rows, ld, h->dim, metric, (uint32_t)N, M,
                           M0, s_ids.as<uint32_t>(), s_dist.as<float>(), s_n.as<uint32_t>(), r_off.as<uint64_t>(), r_src.as<uint32_t>(),
                           r_dist.as<float>(), h->adj, h->deg, h->gstride);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    }));
    ZV_HIP(hipStreamSynchronize(s));

    // ---- upper layers: each one an index of its own over the gathered rows of its nodes ---------------------------------
    for (uint32_t layer = 1; layer <= maxlev; ++layer) {
        std::vector<uint32_t> nodes;
        for (uint64_t i = 0; i < N; ++i)
            if (lv[i] >= layer) nodes.push_back((uint32_t)i);
        const uint64_t n = nodes.size();
        if (n < 2) continue;  // a single node: its list stays empty (deg_up was zeroed)
        DevBuf d_nodes, d_rows;
        ZV_CHECK(d_nodes.alloc(n * 4));
        ZV_CHECK(d_rows.alloc(n * h->dim * 4));
        ZV_HIP(hipMemcpyAsync(d_nodes.p, nodes.data(), n * 4, hipMemcpyHostToDevice, s));
        {
            const uint64_t threads = n * h->dim;
            ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
                hipLaunchKernelGGL(k_gb_gather_rows<row_elem_t<decltype(rows)>>, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, rows, ld, h->dim,
                                   d_nodes.as<uint32_t>(), 0ull, n, d_rows.as<float>());
                ZV_HIP(hipGetLastError());
                return ZVDB_OK;
            }));
        }
        ZV_HIP(hipStreamSynchronize(s));
        zvdb_hip_config cfg{};
        cfg.abi_version = ZVDB_HIP_ABI_VERSION;
        cfg.dim = h->dim;
        cfg.metric = metric;  // (cosine rows are already normalised: the layer index must not normalise again)
        cfg.m = h->m;
        cfg.ef_construction = h->ef_construction;
        cfg.device = h->device;
        cfg.flags = h->flags & (ZVDB_FLAG_BF16_MIRROR | ZVDB_FLAG_BF16_ROWS);  // (bf16 rows: the gathered values are bf16-representable, re-rounding is the identity)
        cfg.graph_stride = 1;
        cfg.capacity = n;
        zvdb_hip_index *sub = nullptr;
        ZV_CHECK(zvdb_hip_create(&cfg, &sub));
        int rc2 = zvdb_hip_add_dev(sub, d_rows.p, n, h->dim, nullptr, nullptr);
        const uint32_t kk = (uint32_t)std::min<uint64_t>(M + 1, n);
        if (rc2 == ZVDB_OK) {
            std::lock_guard<std::mutex> g2(sub->mu);
            rc2 = gb_self_knn(sub, kk, mode, [&](uint64_t *d_ids, float *, uint32_t *d_cnt, uint32_t k, uint64_t r0, uint32_t nb) -> int {
                hipLaunchKernelGGL(k_gb_upper_write, dim3((nb + 255) / 256), dim3(256), 0, sub->stream, d_ids, d_cnt, k, r0, nb, M, d_nodes.as<uint32_t>(),
                                   h->slot, layer, h->adj_up, h->deg_up, h->gstride);
                ZV_HIP(hipGetLastError());
                return ZVDB_OK;
            });
            if (rc2 == ZVDB_OK && hipStreamSynchronize(sub->stream) != hipSuccess) rc2 = ZVDB_ERR_HIP;
        }
        zvdb_hip_destroy(sub);
        ZV_CHECK(rc2);
    }
    h->max_level = maxlev;
    h->up_count = up;
    h->bulk_top = top;
    h->bulk_count = N;
    h->bulk_graph = true;
    return ZVDB_OK;
}

}  // namespace zvdb
