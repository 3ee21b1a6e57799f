// exact_kernels.hip -- kernels that must reproduce the reference arithmetic bit for bit.
//
// Reference arithmetic (src/hnsw.zig:182-192): sum += (a[i]-b[i])*(a[i]-b[i]), strict
// left-to-right fp32, every sub/mul/add rounded separately (Zig never contracts).  This file
// is compiled with -ffp-contract=off and additionally pins the pragma below, so hipcc's default
// -ffp-contract=fast-honor-pragmas cannot fuse the mul+add into v_fma/v_fmac.
//
// Kernels (all wave64, gfx950):
//   k_exact_distances   K1/K5  a1 at the call sites hnsw.zig:86,:96,:160-161
//   k_greedy_layer0     a2     insert's steepest-descent scan at layer 0 (hnsw.zig:88-104)
//   k_search_compat     a5/a6  search (hnsw.zig:194-236): one wave per query
//   k_search_ef         ext.   ef-mode beam search (spec: oracle/zvdb_oracle.h)
//   k_rerank            K5     exact re-rank + (dist,id) top-k of brute-force candidates
#include "common.h"

#pragma clang fp contract(off)

namespace zvdb {

// ---- a1 ---------------------------------------------------------------------
// One lane walks one whole row sequentially (the ONLY order that is bit-exact);
// parallelism comes from different lanes taking different rows.  q may live in
// LDS (broadcast reads) or global memory (uniform address).
template <int METRIC>
__device__ __forceinline__ float exact_dist(const float *__restrict__ q,
                                            const float *__restrict__ row, uint32_t dim) {
    float sum = 0.0f;
    const uint32_t d4 = dim & ~3u;
    uint32_t i = 0;
#pragma unroll 4
    for (; i < d4; i += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(row + i);
        const float4 a = *reinterpret_cast<const float4 *>(q + i);
        if (METRIC == ZVDB_METRIC_L2SQ) {
            float t;
            t = a.x - v.x; t = t * t; sum = sum + t;
            t = a.y - v.y; t = t * t; sum = sum + t;
            t = a.z - v.z; t = t * t; sum = sum + t;
            t = a.w - v.w; t = t * t; sum = sum + t;
        } else {
            float t;
            t = a.x * v.x; sum = sum + t;
            t = a.y * v.y; sum = sum + t;
            t = a.z * v.z; sum = sum + t;
            t = a.w * v.w; sum = sum + t;
        }
    }
    for (; i < dim; ++i) {
        if (METRIC == ZVDB_METRIC_L2SQ) {
            float t = q[i] - row[i];
            t = t * t;
            sum = sum + t;
        } else {
            const float t = q[i] * row[i];
            sum = sum + t;
        }
    }
    return METRIC == ZVDB_METRIC_L2SQ ? sum : -sum;
}

__device__ __forceinline__ float exact_dist_rt(int metric, const float *q, const float *row,
                                               uint32_t dim) {
    return metric == ZVDB_METRIC_L2SQ ? exact_dist<ZVDB_METRIC_L2SQ>(q, row, dim)
                                      : exact_dist<ZVDB_METRIC_DOT>(q, row, dim);
}

__device__ __forceinline__ bool lex_less(float da, uint32_t ia, float db, uint32_t ib) {
    return da < db || (da == db && ia < ib);
}

// ---- K1: out[i] = distance(query, rows[ids[i]]) ------------------------------
__global__ __launch_bounds__(256) void k_exact_distances(const float *__restrict__ rows, uint32_t ld,
                                                         uint32_t dim, int metric,
                                                         const float *__restrict__ query,
                                                         const uint32_t *__restrict__ ids, uint32_t n,
                                                         float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *row = rows + (size_t)ids[i] * ld;
    out[i] = exact_dist_rt(metric, query, row, dim);
}

int launch_exact_distances(const zvdb_hip_index *h, const float *d_query, const uint32_t *d_ids,
                           uint32_t n, float *d_out, hipStream_t s) {
    if (n == 0) return ZVDB_OK;
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    hipLaunchKernelGGL(k_exact_distances, dim3((n + 255) / 256), dim3(256), 0, s, h->rows, h->ld,
                       h->dim, metric, d_query, d_ids, n, d_out);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// ---- a2: layer-0 greedy descent of insert (hnsw.zig:88-104) --------------------
// One wave.  Each while-iteration scans the WHOLE list of the node that is current at
// the top of the iteration (hnsw.zig:92-94); the running best uses strict '<' so the
// list's arg-min wins and the earliest entry wins ties (hnsw.zig:97).
__global__ __launch_bounds__(64) void k_greedy_layer0(const float *__restrict__ rows, uint32_t ld,
                                                      uint32_t dim, int metric,
                                                      const uint32_t *__restrict__ adj,
                                                      const uint32_t *__restrict__ deg,
                                                      uint32_t gstride, uint32_t owner,
                                                      uint32_t *ep_io, float *dist_io,
                                                      unsigned long long *evals_out) {
    const int lane = threadIdx.x;
    const float *q = rows + (size_t)owner * ld;
    uint32_t ep = *ep_io;
    float cur = *dist_io;
    unsigned long long evals = 0;
    bool changed = true;
    while (changed) {
        changed = false;
        const uint32_t dg = deg[ep];
        float best_d = cur;
        uint32_t best_j = 0xFFFFFFFFu;  // list position of the running best, none yet
        uint32_t best_id = ep;
        for (uint32_t base = 0; base < dg; base += kWave) {
            const uint32_t j = base + lane;
            float d = 0.0f;
            uint32_t nb = 0;
            const bool act = j < dg;
            if (act) {
                nb = adj[(size_t)ep * gstride + j];
                d = exact_dist_rt(metric, q, rows + (size_t)nb * ld, dim);
            }
            // wave arg-min over (d, j) of active lanes that beat best_d strictly
            float md = act ? d : __builtin_inff();
            uint32_t mj = act ? j : 0xFFFFFFFFu;
            uint32_t mid = nb;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float od = __shfl_xor(md, off);
                const uint32_t oj = __shfl_xor(mj, off);
                const uint32_t oid = __shfl_xor(mid, off);
                if (od < md || (od == md && oj < mj)) { md = od; mj = oj; mid = oid; }
            }
            if (mj != 0xFFFFFFFFu && md < best_d) { best_d = md; best_j = mj; best_id = mid; }
        }
        evals += dg;
        if (best_j != 0xFFFFFFFFu) { ep = best_id; cur = best_d; changed = true; }
    }
    if (lane == 0) {
        *ep_io = ep;
        *dist_io = cur;
        *evals_out = evals;
    }
}

int launch_greedy_layer0(const zvdb_hip_index *h, uint32_t owner, uint32_t *d_ep_io,
                         float *d_dist_io, unsigned long long *d_evals, hipStream_t s) {
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    hipLaunchKernelGGL(k_greedy_layer0, dim3(1), dim3(64), 0, s, h->rows, h->ld, h->dim, metric,
                       h->adj, h->deg, h->gstride, owner, d_ep_io, d_dist_io, d_evals);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// ---- LDS-resident per-query state helpers (one wave owns them) ------------------
// Visited set: open addressing, key = id + 1 (0 = empty), linear probing.
__device__ __forceinline__ uint32_t hash_slot(uint32_t id, uint32_t bits) {
    return (id * 2654435761u) >> (32 - bits);
}
__device__ __forceinline__ bool hash_contains(const uint32_t *tab, uint32_t bits, uint32_t id) {
    const uint32_t mask = (1u << bits) - 1;
    uint32_t s = hash_slot(id, bits);
    for (;;) {
        const uint32_t v = tab[s];
        if (v == 0) return false;
        if (v == id + 1) return true;
        s = (s + 1) & mask;
    }
}
// wave-uniform insert (every lane executes it with the same id)
__device__ __forceinline__ void hash_insert_uniform(uint32_t *tab, uint32_t bits, uint32_t id) {
    const uint32_t mask = (1u << bits) - 1;
    uint32_t s = hash_slot(id, bits);
    for (;;) {
        const uint32_t v = tab[s];
        if (v == 0) { tab[s] = id + 1; return; }
        if (v == id + 1) return;
        s = (s + 1) & mask;
    }
}
// per-lane concurrent insert; returns true if THIS lane inserted the id (first owner)
__device__ __forceinline__ bool hash_insert_atomic(uint32_t *tab, uint32_t bits, uint32_t id) {
    const uint32_t mask = (1u << bits) - 1;
    uint32_t s = hash_slot(id, bits);
    for (;;) {
        const uint32_t old = atomicCAS(&tab[s], 0u, id + 1);
        if (old == 0) return true;
        if (old == id + 1) return false;
        s = (s + 1) & mask;
    }
}

// Frontier: Zig std.PriorityQueue (std/priority_queue.zig, Zig >= 0.13) restated on two LDS
// arrays; executed wave-uniformly (every lane performs the same reads and writes) so the
// pop order -- including the order of exactly equal distances -- is the oracle's.
struct LdsHeap {
    float *d;
    uint32_t *i;
    uint32_t n;
};
__device__ __forceinline__ void heap_add(LdsHeap &hp, uint32_t id, float dist) {
    uint32_t child = hp.n;
    hp.n += 1;
    while (child > 0) {
        const uint32_t parent = (child - 1) >> 1;
        const float pd = hp.d[parent];
        if (!(dist < pd)) break;
        hp.d[child] = pd;
        hp.i[child] = hp.i[parent];
        child = parent;
    }
    hp.d[child] = dist;
    hp.i[child] = id;
}
__device__ __forceinline__ void heap_remove(LdsHeap &hp, uint32_t &id, float &dist) {
    id = hp.i[0];
    dist = hp.d[0];
    const float ld_ = hp.d[hp.n - 1];
    const uint32_t li = hp.i[hp.n - 1];
    hp.n -= 1;
    if (hp.n == 0) return;
    uint32_t idx = 0;
    for (;;) {
        uint32_t lesser = (idx * 2) | 1;
        if (!(lesser < hp.n)) break;
        const uint32_t next = lesser + 1;
        float cd = hp.d[lesser];
        if (next < hp.n) {
            const float nd = hp.d[next];
            if (nd < cd) { lesser = next; cd = nd; }
        }
        if (ld_ < cd) break;
        hp.d[idx] = cd;
        hp.i[idx] = hp.i[lesser];
        idx = lesser;
    }
    hp.d[idx] = ld_;
    hp.i[idx] = li;
}

// ---- a5/a6: reference-compat search (hnsw.zig:194-236) -------------------------
__global__ __launch_bounds__(64) void k_search_compat(
    const float *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
    const uint32_t *__restrict__ adj, const uint32_t *__restrict__ deg, uint32_t gstride,
    uint32_t N, const float *__restrict__ queries, uint32_t k, uint32_t entry, uint64_t id_offset,
    uint64_t *__restrict__ out_ids, float *__restrict__ out_dist, uint32_t *__restrict__ out_count,
    unsigned long long *__restrict__ stats, uint32_t dim_pad, uint32_t heap_cap, uint32_t hash_bits,
    uint32_t rcap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *q = reinterpret_cast<float *>(smem);
    float *heap_d = q + dim_pad;
    uint32_t *heap_i = reinterpret_cast<uint32_t *>(heap_d + heap_cap);
    uint32_t *hash = heap_i + heap_cap;
    float *res_d = reinterpret_cast<float *>(hash + (1u << hash_bits));
    uint32_t *res_i = reinterpret_cast<uint32_t *>(res_d + rcap);
    float *nb_d = reinterpret_cast<float *>(res_i + rcap);
    uint32_t *nb_i = reinterpret_cast<uint32_t *>(nb_d + kWave);

    const int lane = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    const float *qg = queries + (size_t)qi * dim;
    for (uint32_t i = lane; i < dim_pad; i += kWave) q[i] = i < dim ? qg[i] : 0.0f;
    for (uint32_t i = lane; i < (1u << hash_bits); i += kWave) hash[i] = 0;
    __syncthreads();

    uint32_t nres = 0;
    unsigned long long evals = 0, pops = 0;
    if (N > 0 && k > 0) {  // hnsw.zig:201: no entry point -> empty result
        LdsHeap hp{heap_d, heap_i, 0};
        const float d0 = exact_dist_rt(metric, q, rows + (size_t)entry * ld, dim);  // :208
        evals = 1;
        heap_add(hp, entry, d0);
        hash_insert_uniform(hash, hash_bits, entry);                                 // :209
        __syncthreads();
        while (hp.n > 0 && nres < k) {                                               // :211
            uint32_t cid;
            float cd;
            heap_remove(hp, cid, cd);                                                // :212
            cid = __builtin_amdgcn_readfirstlane(cid);
            pops += 1;
            res_i[nres] = cid;                                                       // :213-214
            res_d[nres] = cd;
            nres += 1;
            const uint32_t dg = deg[cid];                                            // :216 layer 0
            for (uint32_t base = 0; base < dg; base += kWave) {
                const uint32_t j = base + lane;
                const uint32_t cnt = min(dg - base, (uint32_t)kWave);
                __syncthreads();
                if (j < dg) {
                    const uint32_t nb = adj[(size_t)cid * gstride + j];
                    nb_i[lane] = nb;
                    if (!hash_contains(hash, hash_bits, nb))                         // :217
                        nb_d[lane] = exact_dist_rt(metric, q, rows + (size_t)nb * ld, dim);  // :219
                }
                __syncthreads();
                for (uint32_t jj = 0; jj < cnt; ++jj) {  // list order (:216) decides heap layout
                    const uint32_t nb = nb_i[jj];
                    if (!hash_contains(hash, hash_bits, nb)) {
                        heap_add(hp, nb, nb_d[jj]);                                   // :220
                        hash_insert_uniform(hash, hash_bits, nb);                     // :221
                        evals += 1;
                    }
                }
            }
        }
        __syncthreads();
        // stable insertion sort by distance (hnsw.zig:227-233); the comparator's recomputed
        // distance is bit-identical to the stored one.
        for (uint32_t i = 1; i < nres; ++i) {
            uint32_t j = i;
            while (j > 0) {
                const float a = res_d[j], b = res_d[j - 1];
                if (!(a < b)) break;
                const uint32_t ia = res_i[j], ib = res_i[j - 1];
                res_d[j] = b; res_d[j - 1] = a;
                res_i[j] = ib; res_i[j - 1] = ia;
                --j;
            }
        }
        __syncthreads();
    }
    for (uint32_t i = lane; i < k; i += kWave) {
        const bool ok = i < nres;
        out_ids[(size_t)qi * k + i] = ok ? (uint64_t)res_i[i] + id_offset : ~0ull;
        out_dist[(size_t)qi * k + i] = ok ? res_d[i] : __builtin_inff();
    }
    if (lane == 0) {
        out_count[qi] = nres;
        if (stats) {
            atomicAdd(&stats[0], evals);
            atomicAdd(&stats[1], pops);
        }
    }
}

static int set_dyn_lds(const void *fn, size_t bytes) {
    if (bytes > 160 * 1024) {
        set_error("per-query search state needs %zu B of LDS (> 160 KiB): k*m too large", bytes);
        return ZVDB_ERR_UNSUPPORTED;
    }
    if (bytes > 48 * 1024) {
        ZV_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    }
    return ZVDB_OK;
}

static uint32_t bits_for(uint64_t entries) {  // table with >= 2x entries, at least 64 slots
    uint32_t b = 6;
    while ((1ull << b) < entries * 2) ++b;
    return b;
}

int launch_search_compat(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t k,
                         uint32_t entry, uint64_t id_offset, uint64_t *d_ids, float *d_dist,
                         uint32_t *d_count, unsigned long long *d_stats, hipStream_t s) {
    if (nq == 0) return ZVDB_OK;
    const uint64_t N = h->count;
    const uint32_t rcap = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(k, N));
    const uint64_t bound = std::max<uint64_t>(1, std::min<uint64_t>(N, 1 + (uint64_t)rcap * h->gstride));
    const uint32_t heap_cap = (uint32_t)bound;
    const uint32_t hash_bits = bits_for(bound);
    const uint32_t dim_pad = round_up(std::max(h->dim, 1u), 4);
    const size_t lds = 4ull * (dim_pad + 2ull * heap_cap + (1ull << hash_bits) + 2ull * rcap + 2 * kWave);
    ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_search_compat), lds));
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    hipLaunchKernelGGL(k_search_compat, dim3(nq), dim3(64), lds, s, h->rows, h->ld, h->dim, metric,
                       h->adj, h->deg, h->gstride, (uint32_t)N, d_queries, k, entry, id_offset, d_ids,
                       d_dist, d_count, d_stats, dim_pad, heap_cap, hash_bits, rcap);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// ---- ef-mode beam search (extension; specification: oracle/zvdb_oracle.h) --------
// W = sorted (dist,id) list of <= ef entries with an "expanded" flag, LDS resident.
__global__ __launch_bounds__(64) void k_search_ef(
    const float *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
    const uint32_t *__restrict__ adj, const uint32_t *__restrict__ deg, uint32_t gstride,
    uint32_t N, const float *__restrict__ queries, uint32_t k, uint32_t ef, uint32_t entry,
    uint64_t id_offset, uint64_t *__restrict__ out_ids, float *__restrict__ out_dist,
    uint32_t *__restrict__ out_count, unsigned long long *__restrict__ stats, uint32_t dim_pad,
    uint32_t hash_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *q = reinterpret_cast<float *>(smem);
    float *wd = q + dim_pad;
    uint32_t *wi = reinterpret_cast<uint32_t *>(wd + ef + 1);
    uint32_t *wx = wi + ef + 1;
    uint32_t *hash = wx + ef + 1;
    float *nb_d = reinterpret_cast<float *>(hash + (1u << hash_bits));
    uint32_t *nb_i = reinterpret_cast<uint32_t *>(nb_d + kWave);

    const int lane = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    const float *qg = queries + (size_t)qi * dim;
    for (uint32_t i = lane; i < dim_pad; i += kWave) q[i] = i < dim ? qg[i] : 0.0f;
    for (uint32_t i = lane; i < (1u << hash_bits); i += kWave) hash[i] = 0;
    __syncthreads();

    uint32_t wn = 0;
    unsigned long long evals = 0, pops = 0;
    const uint32_t hash_limit = (1u << hash_bits) - (1u << (hash_bits - 2));  // 75 % load cap
    uint32_t hash_used = 0;
    bool overflow = false;
    if (N > 0 && k > 0) {
        const float d0 = exact_dist_rt(metric, q, rows + (size_t)entry * ld, dim);
        if (lane == 0) { wd[0] = d0; wi[0] = entry; wx[0] = 0; }
        wn = 1;
        evals = 1;
        hash_insert_uniform(hash, hash_bits, entry);
        hash_used = 1;
        __syncthreads();
        for (;;) {
            // first entry of W not yet expanded
            uint32_t c = 0xFFFFFFFFu;
            for (uint32_t base = 0; base < wn && c == 0xFFFFFFFFu; base += kWave) {
                const uint32_t i = base + lane;
                const bool open = i < wn && wx[i] == 0;
                const unsigned long long b = __ballot(open);
                if (b) c = base + (uint32_t)__builtin_ctzll(b);
            }
            if (c == 0xFFFFFFFFu) break;
            const uint32_t cid = wi[c];
            __syncthreads();
            if (lane == 0) wx[c] = 1;
            pops += 1;
            const uint32_t dg = deg[cid];
            for (uint32_t base = 0; base < dg; base += kWave) {
                const uint32_t j = base + lane;
                const uint32_t cnt = min(dg - base, (uint32_t)kWave);
                __syncthreads();
                bool fresh = false;
                uint32_t nb = 0;
                if (hash_used + cnt >= hash_limit) overflow = true;  // wave-uniform
                if (j < dg && !overflow) {
                    nb = adj[(size_t)cid * gstride + j];
                    fresh = hash_insert_atomic(hash, hash_bits, nb);
                }
                float dn = 0.0f;
                if (fresh) dn = exact_dist_rt(metric, q, rows + (size_t)nb * ld, dim);
                const unsigned long long fb = __ballot(fresh);
                nb_i[lane] = nb;
                nb_d[lane] = dn;
                hash_used += (uint32_t)__popcll(fb);
                evals += (unsigned long long)__popcll(fb);
                __syncthreads();
                // merge the fresh candidates into W one at a time (set semantics: the result
                // is the best ef of the union, independent of the insertion order)
                unsigned long long todo = fb;
                while (todo) {
                    const int jj = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const float cd = nb_d[jj];
                    const uint32_t ci = nb_i[jj];
                    if (wn == ef && !lex_less(cd, ci, wd[wn - 1], wi[wn - 1])) continue;
                    // position = number of entries lexicographically smaller
                    uint32_t pos = 0;
                    for (uint32_t b2 = 0; b2 < wn; b2 += kWave) {
                        const uint32_t i = b2 + lane;
                        const bool lt = i < wn && lex_less(wd[i], wi[i], cd, ci);
                        pos += (uint32_t)__popcll(__ballot(lt));
                    }
                    const uint32_t last = wn < ef ? wn : ef - 1;  // index that receives the shifted tail
                    // shift [pos, last) one slot right, top chunk first
                    for (int b2 = (int)((last - pos + kWave - 1) / kWave) - 1; b2 >= 0; --b2) {
                        const uint32_t i = pos + (uint32_t)b2 * kWave + lane;
                        float td = 0.0f;
                        uint32_t ti = 0, tx = 0;
                        const bool mv = i < last;
                        if (mv) { td = wd[i]; ti = wi[i]; tx = wx[i]; }
                        __syncthreads();
                        if (mv) { wd[i + 1] = td; wi[i + 1] = ti; wx[i + 1] = tx; }
                        __syncthreads();
                    }
                    if (lane == 0) { wd[pos] = cd; wi[pos] = ci; wx[pos] = 0; }
                    if (wn < ef) wn += 1;
                    __syncthreads();
                }
            }
        }
        __syncthreads();
    }
    const uint32_t nres = wn < k ? wn : k;
    for (uint32_t i = lane; i < k; i += kWave) {
        const bool ok = i < nres;
        out_ids[(size_t)qi * k + i] = ok ? (uint64_t)wi[i] + id_offset : ~0ull;
        out_dist[(size_t)qi * k + i] = ok ? wd[i] : __builtin_inff();
    }
    if (lane == 0) {
        out_count[qi] = nres;
        if (stats) {
            atomicAdd(&stats[0], evals);
            atomicAdd(&stats[1], pops);
            if (overflow) atomicAdd(&stats[2], 1ull);  // visited set exhausted: result not to spec
        }
    }
}

int launch_search_ef(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t k,
                     uint32_t ef, uint32_t entry, uint64_t id_offset, uint64_t *d_ids,
                     float *d_dist, uint32_t *d_count, unsigned long long *d_stats, hipStream_t s) {
    if (nq == 0) return ZVDB_OK;
    if (ef < k) ef = k;
    const uint64_t N = h->count;
    const uint32_t dim_pad = round_up(std::max(h->dim, 1u), 4);
    // visited set: every node is visited at most once.  Measured evaluations per query are ~1.1 x
    // ef * stride; the table takes >= 3 x ef * stride slots (75 % load cap => room for 2.2 x), because its
    // LDS footprint sets the waves per CU (the first version reserved 8 x and ran ONE wave per CU).
    // A query that still exhausts it is flagged in stats[2] (the host entry point reports it).
    uint64_t want = std::min<uint64_t>(std::max<uint64_t>(N, 1), (uint64_t)ef * h->gstride * 3 / 2 + 64);
    uint32_t hash_bits = bits_for(want);
    for (;;) {
        const size_t lds = 4ull * (dim_pad + 3ull * (ef + 1) + (1ull << hash_bits) + 2 * kWave);
        if (lds <= 160 * 1024 || hash_bits <= 8) break;
        --hash_bits;
    }
    const size_t lds = 4ull * (dim_pad + 3ull * (ef + 1) + (1ull << hash_bits) + 2 * kWave);
    ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_search_ef), lds));
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    hipLaunchKernelGGL(k_search_ef, dim3(nq), dim3(64), lds, s, h->rows, h->ld, h->dim, metric, h->adj,
                       h->deg, h->gstride, (uint32_t)N, d_queries, k, ef, entry, id_offset, d_ids,
                       d_dist, d_count, d_stats, dim_pad, hash_bits);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// ---- K5: exact re-rank of brute-force candidates ---------------------------------
// cand[q][kc] (u32 local ids, 0xFFFFFFFF = empty).  One wave per query: every lane takes
// candidates lane, lane+64, ... and computes the reference-exact distance; the final order
// is (exact dist, id) ascending, duplicates dropped.
__global__ __launch_bounds__(64) void k_rerank(const float *__restrict__ rows, uint32_t ld,
                                               uint32_t dim, int metric,
                                               const float *__restrict__ queries,
                                               const uint32_t *__restrict__ cand, uint32_t kc,
                                               uint32_t k, uint64_t id_offset,
                                               uint64_t *__restrict__ out_ids,
                                               float *__restrict__ out_dist,
                                               uint32_t *__restrict__ out_count, uint32_t dim_pad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *q = reinterpret_cast<float *>(smem);
    float *cd = q + dim_pad;
    uint32_t *ci = reinterpret_cast<uint32_t *>(cd + kc);
    const int lane = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    const float *qg = queries + (size_t)qi * dim;
    for (uint32_t i = lane; i < dim_pad; i += kWave) q[i] = i < dim ? qg[i] : 0.0f;
    __syncthreads();
    for (uint32_t i = lane; i < kc; i += kWave) ci[i] = cand[(size_t)qi * kc + i];
    __syncthreads();
    // drop duplicate ids (keep the first copy), then evaluate the survivors exactly
    for (uint32_t base = 0; base < kc; base += kWave) {
        const uint32_t i = base + lane;
        bool dup = false;
        if (i < kc) {
            const uint32_t ii = ci[i];
            if (ii != 0xFFFFFFFFu)
                for (uint32_t j = 0; j < i; ++j)
                    if (ci[j] == ii) { dup = true; break; }
        }
        __syncthreads();
        if (dup) ci[i] = 0xFFFFFFFFu;  // only later entries are cleared: earlier tests stay valid
        __syncthreads();
    }
    for (uint32_t i = lane; i < kc; i += kWave) {
        const uint32_t id = ci[i];
        cd[i] = id == 0xFFFFFFFFu ? __builtin_inff()
                                  : exact_dist_rt(metric, q, rows + (size_t)id * ld, dim);
    }
    __syncthreads();
    uint32_t total = 0;
    for (uint32_t base = 0; base < kc; base += kWave) {
        const uint32_t i = base + lane;
        const bool valid = i < kc && ci[i] != 0xFFFFFFFFu;
        uint32_t rank = 0;
        if (valid) {
            const float di = cd[i];
            const uint32_t ii = ci[i];
            for (uint32_t j = 0; j < kc; ++j) {
                const uint32_t ij = ci[j];
                if (ij != 0xFFFFFFFFu && lex_less(cd[j], ij, di, ii)) ++rank;
            }
            if (rank < k) {
                out_ids[(size_t)qi * k + rank] = (uint64_t)ii + id_offset;
                out_dist[(size_t)qi * k + rank] = di;
            }
        }
        total += (uint32_t)__popcll(__ballot(valid));
    }
    const uint32_t nres = total < k ? total : k;
    for (uint32_t i = nres + lane; i < k; i += kWave) {
        out_ids[(size_t)qi * k + i] = ~0ull;
        out_dist[(size_t)qi * k + i] = __builtin_inff();
    }
    if (lane == 0) out_count[qi] = nres;
}

int launch_rerank(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, const uint32_t *d_cand,
                  uint32_t kc, uint32_t k, uint64_t id_offset, uint64_t *d_ids, float *d_dist,
                  uint32_t *d_count, hipStream_t s) {
    if (nq == 0) return ZVDB_OK;
    const uint32_t dim_pad = round_up(std::max(h->dim, 1u), 4);
    const size_t lds = 4ull * (dim_pad + 2ull * kc);
    ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_rerank), lds));
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    hipLaunchKernelGGL(k_rerank, dim3(nq), dim3(64), lds, s, h->rows, h->ld, h->dim, metric, d_queries,
                       d_cand, kc, k, id_offset, d_ids, d_dist, d_count, dim_pad);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

}  // namespace zvdb
