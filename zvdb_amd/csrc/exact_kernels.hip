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
//   k_insert_layer0 / k_insert_full      a2-a4  the insert (hnsw.zig:73-117), one wave, n points per launch in order
//   k_insert_spec[_loop] / _check / _commit     the same insert for a round of points speculatively in parallel (bulk calls)
//   k_search_compat[_fast]      a5/a6  search (hnsw.zig:194-236): one wave per query
//   k_search_ef_reg / k_search_ef       ext.   ef-mode beam search (spec: oracle/zvdb_oracle.h): beam in registers / in LDS
//   k_descend           ext.   entry descent through the bulk graph's upper layers
//   k_rerank[_staged]   K5     exact re-rank + (dist,id) top-k of brute-force candidates
#include <cstring>
#include "common.h"
#include "wave_select.h"
#include "exact_dist.h"
#include "bf_cert.h"

#pragma clang fp contract(off)

namespace zvdb {

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

// ---- K1: out[i] = distance(query, rows[ids[i]]) ------------------------------
template <typename RT>
__global__ __launch_bounds__(256) void k_exact_distances(const RT *__restrict__ rows, uint32_t ld,
                                                         uint32_t dim, int metric,
                                                         const float *__restrict__ query,
                                                         const uint32_t *__restrict__ ids, uint32_t n,
                                                         float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RT *row = rows + (size_t)ids[i] * ld;
    out[i] = exact_dist_rt(metric, query, row, dim);
}

int launch_exact_distances(const zvdb_hip_index *h, const float *d_query, const uint32_t *d_ids,
                           uint32_t n, float *d_out, hipStream_t s) {
    if (n == 0) return ZVDB_OK;
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    return rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = std::remove_cv_t<std::remove_pointer_t<decltype(rows)>>;
        hipLaunchKernelGGL(k_exact_distances<RT>, dim3((n + 255) / 256), dim3(256), 0, s, rows, ld, h->dim, metric, d_query, d_ids, n, d_out);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    });
}

// ---- a2: layer-0 greedy descent of insert (hnsw.zig:88-104) --------------------
// One wave.  Each while-iteration scans the WHOLE list of the node that is current at
// the top of the iteration (hnsw.zig:92-94); the running best uses strict '<' so the
// list's arg-min wins and the earliest entry wins ties (hnsw.zig:97).
template <typename RT>
__global__ __launch_bounds__(64) void k_greedy_layer0(const RT *__restrict__ rows, uint32_t ld,
                                                      uint32_t dim, int metric,
                                                      const uint32_t *__restrict__ adj,
                                                      const uint32_t *__restrict__ deg,
                                                      uint32_t gstride, uint32_t owner,
                                                      uint32_t *ep_io, float *dist_io,
                                                      unsigned long long *evals_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g0[];
    const int lane = threadIdx.x;
    float *q = reinterpret_cast<float *>(smem_g0);  // the owner's row, widened to fp32 ([dim], 16-byte aligned)
    for (uint32_t i = lane; i < dim; i += kWave) q[i] = row_elem(rows + (size_t)owner * ld, i);
    __syncthreads();
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

// ---- a2 + a4 + a3 at layer 0, fused (one wave, one launch per insert) -----------------------------------------------
// hnsw.zig:86 (distance of the new point to the entry point), :88-104 at layer 0 (the greedy scan above), :106-108 ->
// connect(owner, ep, 0) :119-141: append each node to the other's layer-0 list, then shrinkConnections :143-170 of both
// (only when a list exceeds m: distances to its owner, one lane per entry; the reference's stable insertion sort with a
// strict '<' is reproduced by ranks -- rank(i) = #{d_j < d_i} + #{j < i : d_j == d_i}; the first m stay).  All on the
// device mirror of layer 0; `out` receives what the host mirror must adopt.

// (`rows` is read and written -- the new row is copied in and read back -- hence one plain pointer: see k_insert_full)
// RT = stored row element.  bf16 rows (ZVDB_FLAG_BF16_ROWS): the point is already in its row (point_src == nullptr); a row used
// as the QUERY of a distance (the new point; ep as the owner of its shrink) is widened into LDS first.
template <typename RT>
__global__ __launch_bounds__(64) void k_insert_layer0(RT *rows, uint32_t ld, uint32_t dim, int metric,
                                                      uint32_t *adj, uint32_t *deg, uint32_t gstride,
                                                      uint32_t m, uint32_t first_owner, uint32_t ep_in, unsigned char *__restrict__ out_base,
                                                      uint32_t seq, const float *__restrict__ point_src /* optional: [n][dim] new points */,
                                                      uint32_t n_batch, uint32_t out_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t cap = (gstride + 1 + 3) & ~3u;           // a list holds at most gstride entries before the append (padded: ql is 16-byte aligned)
    uint32_t *lst = reinterpret_cast<uint32_t *>(smem);     // [cap]
    float *dst = reinterpret_cast<float *>(lst + cap);      // [cap]
    uint32_t *srt = reinterpret_cast<uint32_t *>(dst + cap);  // [cap]
    float *ql = reinterpret_cast<float *>(srt + cap);       // [dim] (with point_src, or bf16 rows)
    float *qo = ql + ((dim + 3) & ~3u);                     // [dim] (bf16 rows: the other node's row as a query)
    // A BATCH is n_batch inserts executed one after the other by this one wave (insert i + 1 sees the lists insert i left:
    // the same result as n_batch separate calls, without their launches).  The lists are re-read through L2 (agent-scope
    // loads) because this CU's vector L1 may still hold what an earlier iteration read.
    auto ld_u32 = [](const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (uint32_t b = 0; b < n_batch; ++b) {
    const uint32_t owner = first_owner + b;
    InsertL0Out *out = reinterpret_cast<InsertL0Out *>(out_base + (size_t)b * out_stride);
    uint32_t *out_ep_list = reinterpret_cast<uint32_t *>(out + 1);
    uint32_t *out_owner_list = out_ep_list + gstride;
    const float *q = ql;
    if constexpr (std::is_same_v<RT, float>) {
        q = rows + (size_t)owner * ld;
        if (point_src) {  // Node.init's copy (:24-26) done here: the point goes to its row AND to LDS, which the walks below read
            for (uint32_t i = lane; i < dim; i += kWave) {
                const float v = point_src[(size_t)b * dim + i];
                rows[(size_t)owner * ld + i] = v;
                ql[i] = v;
            }
            __syncthreads();
            q = ql;
        }
    } else {
        for (uint32_t i = lane; i < dim; i += kWave) ql[i] = row_elem(rows + (size_t)owner * ld, i);
        __syncthreads();
    }
    // :86 -- one lane walks the row (the only bit-exact order), the wave adopts its value
    float cur = 0.0f;
    if (lane == 0) cur = exact_dist_rt(metric, q, rows + (size_t)ep_in * ld, dim);
    cur = __shfl(cur, 0);
    uint32_t ep = ep_in;
    unsigned long long evals = 0;
    bool changed = true;
    while (changed) {  // :90-104 (same scan as k_greedy_layer0)
        changed = false;
        const uint32_t dg = min(ld_u32(deg + ep), gstride);
        float best_d = cur;
        uint32_t best_j = 0xFFFFFFFFu, best_id = ep;
        for (uint32_t base = 0; base < dg; base += kWave) {
            const uint32_t j = base + lane;
            float d = 0.0f;
            uint32_t nb = 0;
            const bool act = j < dg;
            if (act) {
                nb = ld_u32(adj + (size_t)ep * gstride + j);
                d = exact_dist_rt(metric, q, rows + (size_t)nb * ld, dim);
            }
            float md = act ? d : __builtin_inff();
            uint32_t mj = act ? j : 0xFFFFFFFFu, mid = nb;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float od = __shfl_xor(md, off);
                const uint32_t oj = __shfl_xor(mj, off), oid = __shfl_xor(mid, off);
                if (od < md || (od == md && oj < mj)) { md = od; mj = oj; mid = oid; }
            }
            if (mj != 0xFFFFFFFFu && md < best_d) { best_d = md; best_j = mj; best_id = mid; }
        }
        evals += dg;
        if (best_j != 0xFFFFFFFFu) { ep = best_id; cur = best_d; changed = true; }
    }
    // connect(owner, ep, 0): lists are read before either is written (the two nodes differ)
    uint32_t err = 0;
    auto link = [&](uint32_t node, uint32_t other, uint32_t *out_list) -> uint32_t {
        const uint32_t n0 = min(ld_u32(deg + node), gstride);
        for (uint32_t i = lane; i < n0; i += kWave) lst[i] = ld_u32(adj + (size_t)node * gstride + i);
        if (lane == 0) lst[n0] = other;                      // :128-133 append
        uint32_t n = n0 + 1;
        __syncthreads();
        if (n > m) {                                          // :146 shrink
            const float *o = q;
            if (node != owner) {
                if constexpr (std::is_same_v<RT, float>) o = rows + (size_t)node * ld;
                else {
                    for (uint32_t i = lane; i < dim; i += kWave) qo[i] = row_elem(rows + (size_t)node * ld, i);
                    __syncthreads();
                    o = qo;
                }
            }
            for (uint32_t i = lane; i < n; i += kWave) dst[i] = exact_dist_rt(metric, o, rows + (size_t)lst[i] * ld, dim);  // :160-161
            __syncthreads();
            for (uint32_t i = lane; i < n; i += kWave) {
                const float di = dst[i];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < n; ++j) {
                    const float dj = dst[j];
                    rank += (dj < di || (dj == di && j < i)) ? 1u : 0u;
                }
                srt[rank] = lst[i];
            }
            __syncthreads();
            n = m;                                            // :168-169
            for (uint32_t i = lane; i < n; i += kWave) lst[i] = srt[i];
            __syncthreads();
        }
        if (n > gstride) { err = 1; n = gstride; }
        for (uint32_t i = lane; i < n; i += kWave) {
            adj[(size_t)node * gstride + i] = lst[i];
            out_list[i] = lst[i];
        }
        if (lane == 0) deg[node] = n;
        __syncthreads();
        return n;
    };
    const uint32_t owner_n = link(owner, ep, out_owner_list);
    const uint32_t ep_n = link(ep, owner, out_ep_list);
    if (lane == 0) {
        out->ep = ep;
        out->ep_dist = cur;
        out->owner_n = owner_n;
        out->ep_n = ep_n;
        out->evals = evals;
        out->error = err;
    }
    __threadfence();  // the row and both lists are in L2 before the next insert of the batch (or anybody else) reads them
    __syncthreads();
    }
    __threadfence_system();  // `out` may be host-mapped memory: everything above is visible to the host before the sequence number
    __syncthreads();
    if (lane == 0) {
        InsertL0Out *last = reinterpret_cast<InsertL0Out *>(out_base + (size_t)(n_batch - 1) * out_stride);
        __hip_atomic_store(&last->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- the whole insert, every layer, n inserts per launch (hnsw.zig:73-117) -----------------------------------------
// As k_insert_layer0, with the upper layers on the device too: per layer 0..max_level (ascending, :88) the greedy scan
// over connections[layer] of the current node (:90-104; a node has the layer iff its level reaches it, :93) and, while
// layer <= the new node's level, connect (:106-108) with both shrinks.  Per (insert, layer <= min(level, max_level)) one
// record {ep, owner_n, ep_n, ep_has_layer, ep's list} tells the host mirror what to adopt.  Parameters per insert are
// computed by the host (level, first upper-layer slot, record offset, max_level before the insert).
// `rows` is read AND written (Node.init's copy of the new row is read back by ep's shrink and by later inserts of the batch):
// one plain pointer, no restrict / const, so the compiler may not treat the loads as invariant.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_insert_full(float *rows, uint32_t ld, uint32_t dim, int metric,
                                                    uint32_t *adj, uint32_t *deg, uint32_t *adj_up, uint32_t *deg_up,
                                                    uint32_t *lvl, uint32_t *slot, uint32_t gstride, uint32_t m,
                                                    uint32_t first_owner, uint32_t n_batch, const float *__restrict__ points,
                                                    const InsertParam *__restrict__ params,
                                                    uint32_t *__restrict__ recs, uint32_t *seq_dst, uint32_t seq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t cap = (gstride + 1 + 3) & ~3u;             // (list arrays padded to 16 bytes: q is read by ds_read_b128)
    uint32_t *lst = reinterpret_cast<uint32_t *>(smem);       // [cap]
    float *dst = reinterpret_cast<float *>(lst + cap);        // [cap]
    uint32_t *srt = reinterpret_cast<uint32_t *>(dst + cap);  // [cap]
    float *q = reinterpret_cast<float *>(srt + cap);          // [dim], 16-byte aligned
    const uint32_t rec_words = 4 + gstride;
    auto ld_u32 = [](const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (uint32_t b = 0; b < n_batch; ++b) {
        const uint32_t owner = first_owner + b;
        const uint32_t level = params[b].level, my_slot = params[b].slot, rec0 = params[b].rec_off, maxlev = params[b].maxlev_pre;
        if (lane == 0) { lvl[owner] = level; slot[owner] = my_slot; }
        for (uint32_t i = lane; i < dim; i += kWave) {  // Node.init's copy (:24-26): to the row and to LDS
            const float v = points[(size_t)b * dim + i];
            rows[(size_t)owner * ld + i] = v;
            q[i] = v;
        }
        __syncthreads();
        // list of `node` at `layer` (nullptr: the node does not reach the layer)
        // (the level / slot of the current ep are kept in registers: most upper layers of most inserts are empty, and
        // asking memory for lvl[ep] once per layer would cost more than the layers' work)
        uint32_t ep_cached = 0xFFFFFFFFu, ep_lvl = 0, ep_slot = 0;
        auto list_of = [&](uint32_t node, uint32_t layer, uint32_t *&dg) -> uint32_t * {
            if (layer == 0) { dg = deg + node; return adj + (size_t)node * gstride; }
            uint32_t nl, ns;
            if (node == owner) { nl = level; ns = my_slot; }
            else {
                if (node != ep_cached) { ep_cached = node; ep_lvl = ld_u32(lvl + node); ep_slot = ep_lvl ? ld_u32(slot + node) : 0u; }
                nl = ep_lvl; ns = ep_slot;
            }
            if (nl < layer) return nullptr;
            const uint32_t sl = ns + layer - 1;
            dg = deg_up + sl;
            return adj_up + (size_t)sl * gstride;
        };
        uint32_t ep = 0;  // the entry point is the first node, always (hnsw.zig:84, :111)
        float cur = 0.0f;
        if (lane == 0) cur = exact_dist_rt(metric, q, rows + (size_t)ep * ld, dim);  // :86
        cur = __shfl(cur, 0);
        for (uint32_t layer = 0; layer <= maxlev; ++layer) {  // :88
            bool changed = true;
            while (changed) {  // :90-104
                changed = false;
                uint32_t *dgp = nullptr;
                const uint32_t *al = list_of(ep, layer, dgp);
                if (!al) break;
                // the degree and the first 64 list slots are requested TOGETHER (both depend on ep only): one memory round
                // trip per hop instead of two before the rows can be fetched (slots past the degree hold stale or zero ids)
                const uint32_t nb0 = (uint32_t)lane < gstride ? ld_u32(al + lane) : 0u;
                const uint32_t dg = min(ld_u32(dgp), gstride);
                float best_d = cur;
                uint32_t best_j = 0xFFFFFFFFu, best_id = ep;
                for (uint32_t base = 0; base < dg; base += kWave) {
                    const uint32_t j = base + lane;
                    float d = 0.0f;
                    uint32_t nb = 0;
                    const bool act = j < dg;
                    if (act) {
                        nb = base == 0 ? nb0 : ld_u32(al + j);
                        d = exact_dist_rt(metric, q, rows + (size_t)nb * ld, dim);
                    }
                    float md = act ? d : __builtin_inff();
                    uint32_t mj = act ? j : 0xFFFFFFFFu, mid = nb;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
                        const float od = __shfl_xor(md, off);
                        const uint32_t oj = __shfl_xor(mj, off), oid = __shfl_xor(mid, off);
                        if (od < md || (od == md && oj < mj)) { md = od; mj = oj; mid = oid; }
                    }
                    if (mj != 0xFFFFFFFFu && md < best_d) { best_d = md; best_j = mj; best_id = mid; }
                }
                if (best_j != 0xFFFFFFFFu) { ep = best_id; cur = best_d; changed = true; }
            }
            if (layer > level) continue;  // :106 -- above the new node's level only the scan runs
            uint32_t *rec = recs + (size_t)(rec0 + layer) * rec_words;
            // connect(owner, ep, layer) :119-141
            auto link = [&](uint32_t node, uint32_t other, uint32_t *out_list) -> uint32_t {
                uint32_t *dgp = nullptr;
                uint32_t *al = list_of(node, layer, dgp);
                if (!al) return 0xFFFFFFFFu;                       // :128 / :131 the node lacks the layer
                const uint32_t n0 = min(ld_u32(dgp), gstride);
                for (uint32_t i = lane; i < n0; i += kWave) lst[i] = ld_u32(al + i);
                if (lane == 0) lst[n0] = other;                    // append
                uint32_t n = n0 + 1;
                __syncthreads();
                if (n > m) {                                        // shrinkConnections :143-170
                    const float *o = (node == owner) ? q : rows + (size_t)node * ld;
                    for (uint32_t i = lane; i < n; i += kWave) dst[i] = exact_dist_rt(metric, o, rows + (size_t)lst[i] * ld, dim);
                    __syncthreads();
                    for (uint32_t i = lane; i < n; i += kWave) {
                        const float di = dst[i];
                        uint32_t rank = 0;
                        for (uint32_t j = 0; j < n; ++j) {
                            const float dj = dst[j];
                            rank += (dj < di || (dj == di && j < i)) ? 1u : 0u;
                        }
                        srt[rank] = lst[i];
                    }
                    __syncthreads();
                    n = m;
                    for (uint32_t i = lane; i < n; i += kWave) lst[i] = srt[i];
                    __syncthreads();
                }
                if (n > gstride) n = gstride;  // (m <= gstride is checked on the host)
                for (uint32_t i = lane; i < n; i += kWave) {
                    al[i] = lst[i];
                    if (out_list) out_list[i] = lst[i];
                }
                if (lane == 0) *dgp = n;
                __syncthreads();
                return n;
            };
            const uint32_t owner_n = link(owner, ep, nullptr);
            const uint32_t ep_n = link(ep, owner, rec + 4);
            if (lane == 0) {
                rec[0] = ep;
                rec[1] = owner_n;
                rec[2] = ep_n == 0xFFFFFFFFu ? 0u : ep_n;
                rec[3] = ep_n == 0xFFFFFFFFu ? 0u : 1u;
            }
            __threadfence();  // both lists are in L2 before the next layer / insert reads them
            __syncthreads();
        }
        __threadfence();
        __syncthreads();
    }
    __threadfence_system();  // recs may be host-mapped memory
    __syncthreads();
    if (lane == 0 && seq_dst) __hip_atomic_store(seq_dst, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int launch_insert_full(const zvdb_hip_index *h, uint32_t first_owner, uint32_t n, const float *points, const InsertParam *params,
                       uint32_t *recs, uint32_t *seq_dst, uint32_t seq, hipStream_t s) {
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    const size_t lds = 3ull * ((h->gstride + 1 + 3) & ~3u) * sizeof(uint32_t) + (size_t)h->dim * sizeof(float);
    if (lds > 48 * 1024) ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_insert_full), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_insert_full, dim3(1), dim3(64), lds, s, h->rows, h->ld, h->dim, metric, h->adj, h->deg, h->adj_up, h->deg_up,
                       h->lvl, h->slot, h->gstride, h->m, first_owner, n, points, params, recs, seq_dst, seq);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// ---- the whole insert, SPECULATIVELY in parallel (same results as k_insert_full, insert by insert) --------------------------------
// The reference's insert is one chain of dependent steps per point, and points must go in one after the other (each sees the
// graph the previous ones left): on a GPU that is ~15 us of latency per point, a third of what one CPU core needs.  But what
// insert i READS of the shared graph is small -- the lists along its greedy paths (one per hop) -- and what it WRITES to it is
// smaller: one existing list per layer <= its level (its final ep's).  So a round takes the next B points and
//   k_insert_spec    one wave per point, all against the graph as committed so far: the full insert (:73-117) with the list
//                    writes held back as records, the set R of list ids it read kept, and every list it would write stamped
//                    with its position in the round (wpos[list] = min position, tagged with the round's epoch);
//   k_insert_check   point j is valid iff what it read is what a sequential run would have shown it, or would have led it the
//                    same way: for every list of R_j stamped this round by a position j' < j -- a list it only SCANNED, to which
//                    j' was merely appended (no shrink), is harmless unless j' is strictly closer to j's point than the best the
//                    scan ended with (the appended entry is scanned last, strict '<', so nothing else about the hop changes: one
//                    extra exact distance decides); the list it CONNECTED to, or a list that was shrunk (reordered), is a conflict.
//                    c = the first invalid position;
//   k_insert_commit  points < c write their lists (pairwise disjoint: two writers of one list conflict, the later one read it).
// The host advances by c >= 1 (the first point of a round is always valid) and runs the next round from there; the points past c
// are simply computed again.  (The reference's graphs are chains -- mean degree 1.6, ~30 hops per insert -- so the paths of a
// round share most of their lists: stamp-only validation committed ~8 points per round at N = 100k, the scan rule above many more.)
// Records, levels, slots, rows: exactly those of k_insert_full -- the host mirror and the tests do not see the difference.
constexpr uint32_t kSpecRMax = 224;  // read-set capacity (lists read by one insert); more: the point is valid only as the first of a round
__device__ __forceinline__ unsigned long long *spec_wpos(unsigned long long *wpos0, unsigned long long *wposU, uint32_t lid) {
    return (lid & 0x80000000u) ? wposU + (lid & 0x7FFFFFFFu) : wpos0 + lid;
}
// COH = true reads past the L1 (agent-scope atomic loads).  Not needed by the round loop below -- its grid barriers fence at agent
// scope -- and kept only as the switch to flip if a platform's fences ever turn out weaker than the memory model says.
template <bool COH> __device__ __forceinline__ uint32_t ldc(const uint32_t *p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH> __device__ __forceinline__ unsigned long long ldc64(const unsigned long long *p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
// stamp of a list: [round key : 32 | second-smallest writer position : 11 (0x7FF none) | smallest : 11 | the smallest one shrank it : 1]
__device__ __forceinline__ void spec_stamp(unsigned long long *p, uint32_t key, uint32_t b, uint32_t shrunk) {
    unsigned long long old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        uint32_t lo;
        if ((uint32_t)(old >> 32) != key) {
            lo = (0x7FFu << 12) | (b << 1) | shrunk;  // first writer of this round (whatever an older round left is void)
        } else {
            uint32_t m1 = ((uint32_t)old >> 1) & 0x7FFu, s1 = (uint32_t)old & 1u, m2 = ((uint32_t)old >> 12) & 0x7FFu;
            if (b < m1) { m2 = m1; m1 = b; s1 = shrunk; }
            else if (b != m1 && b < m2) m2 = b;
            lo = (m2 << 12) | (m1 << 1) | s1;
        }
        const unsigned long long nv = ((unsigned long long)key << 32) | lo;
        if (nv == old) return;
        const unsigned long long seen = atomicCAS(p, old, nv);
        if (seen == old) return;
        old = seen;
    }
}
struct SpecArgs {
    float *rows; uint32_t ld, dim; int metric;
    uint32_t *adj, *deg, *adj_up, *deg_up, *lvl, *slot; uint32_t gstride, m;
    uint32_t first_owner; const float *points; const InsertParam *params; uint32_t *recs;
    unsigned long long *wpos0, *wposU; uint32_t *rset, *rcnt, *defer;
};

template <bool COH>
__device__ __forceinline__ void spec_body(const SpecArgs &A, uint32_t b, uint32_t epoch_key, unsigned char *smem) {
    float *rows = A.rows;
    const uint32_t ld = A.ld, dim = A.dim, gstride = A.gstride, m = A.m;
    const int metric = A.metric;
    const int lane = threadIdx.x;
    const uint32_t cap = (gstride + 1 + 3) & ~3u;
    uint32_t *lst = reinterpret_cast<uint32_t *>(smem);       // [cap]
    float *dst = reinterpret_cast<float *>(lst + cap);        // [cap]
    uint32_t *srt = reinterpret_cast<uint32_t *>(dst + cap);  // [cap]
    uint32_t *rs = srt + cap;                                 // [kSpecRMax] list ids read
    float *rbd = reinterpret_cast<float *>(rs + kSpecRMax);   // [kSpecRMax] best distance each scan ended with
    uint32_t *rid = reinterpret_cast<uint32_t *>(rbd + kSpecRMax);  // [kSpecRMax] the node it moved to (0xFFFFFFFF: it stayed)
    uint32_t *rly = rid + kSpecRMax;                          // [kSpecRMax] the list's layer
    float *q = reinterpret_cast<float *>(rly + kSpecRMax);    // [dim], 16-byte aligned (cap and kSpecRMax are multiples of 4)
    const uint32_t rec_words = 4 + gstride;
    const uint32_t owner = A.first_owner + b;
    const uint32_t level = A.params[b].level, my_slot = A.params[b].slot, rec0 = A.params[b].rec_off, maxlev = A.params[b].maxlev_pre;
    if (lane == 0) { A.lvl[owner] = level; A.slot[owner] = my_slot; A.defer[b] = 0u; }  // (nobody reaches `owner` before it is committed)
    for (uint32_t i = lane; i < dim; i += kWave) {  // Node.init's copy (:24-26): to the row and to LDS
        const float v = A.points[(size_t)b * dim + i];
        rows[(size_t)owner * ld + i] = v;
        q[i] = v;
    }
    __syncthreads();
    uint32_t rn = 0;  // (wave-uniform)
    uint32_t ep_cached = 0xFFFFFFFFu, ep_lvl = 0, ep_slot = 0;
    // list of an EXISTING node at `layer` (nullptr: the node does not reach the layer); lid = its id in the stamp tables
    auto list_of = [&](uint32_t node, uint32_t layer, const uint32_t *&dg, uint32_t &lid) -> const uint32_t * {
        if (layer == 0) { dg = A.deg + node; lid = node; return A.adj + (size_t)node * gstride; }
        if (node != ep_cached) { ep_cached = node; ep_lvl = ldc<COH>(A.lvl + node); ep_slot = ep_lvl ? ldc<COH>(A.slot + node) : 0u; }
        if (ep_lvl < layer) return nullptr;
        const uint32_t sl = ep_slot + layer - 1;
        dg = A.deg_up + sl;
        lid = 0x80000000u | sl;
        return A.adj_up + (size_t)sl * gstride;
    };
    auto note_read = [&](uint32_t lid, float best_after_scan, uint32_t moved_to, uint32_t layer) {  // lid | 0x40000000: read to connect
        if (rn < kSpecRMax) { if (lane == 0) { rs[rn] = lid; rbd[rn] = best_after_scan; rid[rn] = moved_to; rly[rn] = layer; } }
        rn += 1;  // (past the capacity: counted, the check treats the point as unverifiable)
    };
    uint32_t ep = 0;  // the entry point is the first node, always (hnsw.zig:84, :111)
    float cur = 0.0f;
    if (lane == 0) cur = exact_dist_rt(metric, q, rows + (size_t)ep * ld, dim);  // :86
    cur = __shfl(cur, 0);
    for (uint32_t layer = 0; layer <= maxlev; ++layer) {  // :88
        bool changed = true;
        while (changed) {  // :90-104
            changed = false;
            const uint32_t *dgp = nullptr;
            uint32_t lid = 0;
            const uint32_t *al = list_of(ep, layer, dgp, lid);
            if (!al) break;
            const uint32_t nb0 = (uint32_t)lane < gstride ? ldc<COH>(al + lane) : 0u;
            const uint32_t dg = min(ldc<COH>(dgp), gstride);
            float best_d = cur;
            uint32_t best_j = 0xFFFFFFFFu, best_id = ep;
            for (uint32_t base = 0; base < dg; base += kWave) {
                const uint32_t j = base + lane;
                float d = 0.0f;
                uint32_t nb = 0;
                const bool act = j < dg;
                if (act) {
                    nb = base == 0 ? nb0 : ldc<COH>(al + j);
                    d = exact_dist_rt(metric, q, rows + (size_t)nb * ld, dim);
                }
                float md = act ? d : __builtin_inff();
                uint32_t mj = act ? j : 0xFFFFFFFFu, mid = nb;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float od = __shfl_xor(md, off);
                    const uint32_t oj = __shfl_xor(mj, off), oid = __shfl_xor(mid, off);
                    if (od < md || (od == md && oj < mj)) { md = od; mj = oj; mid = oid; }
                }
                if (mj != 0xFFFFFFFFu && md < best_d) { best_d = md; best_j = mj; best_id = mid; }
            }
            note_read(lid, best_d, best_j != 0xFFFFFFFFu ? best_id : 0xFFFFFFFFu, layer);
            if (best_j != 0xFFFFFFFFu) { ep = best_id; cur = best_d; changed = true; }
        }
        if (layer > level) continue;  // :106 -- above the new node's level only the scan runs
        uint32_t *rec = A.recs + (size_t)(rec0 + layer) * rec_words;
        // connect(owner, ep, layer) :119-141.  The owner's list is new: [ep] (one entry, no shrink).  ep's list, if ep has the layer:
        // its committed list + owner, shrunk (:143-170) when it exceeds m -- kept in the record, written by the commit step.
        const uint32_t *dgp = nullptr;
        uint32_t lid = 0;
        const uint32_t *al = list_of(ep, layer, dgp, lid);
        uint32_t ep_n = 0xFFFFFFFFu;
        if (al) {
            note_read(lid | 0x40000000u, 0.0f, 0xFFFFFFFFu, layer);
            const uint32_t n0 = min(ldc<COH>(dgp), gstride);
            for (uint32_t i = lane; i < n0; i += kWave) lst[i] = ldc<COH>(al + i);
            if (lane == 0) lst[n0] = owner;  // append
            uint32_t n = n0 + 1;
            const bool shrunk = n > m;
            __syncthreads();
            if (n > m) {  // shrinkConnections :143-170
                const float *o = rows + (size_t)ep * ld;
                for (uint32_t i = lane; i < n; i += kWave)
                    dst[i] = lst[i] == owner ? exact_dist_rt(metric, o, q, dim) : exact_dist_rt(metric, o, rows + (size_t)lst[i] * ld, dim);
                __syncthreads();
                for (uint32_t i = lane; i < n; i += kWave) {
                    const float di = dst[i];
                    uint32_t rank = 0;
                    for (uint32_t j = 0; j < n; ++j) {
                        const float dj = dst[j];
                        rank += (dj < di || (dj == di && j < i)) ? 1u : 0u;
                    }
                    srt[rank] = lst[i];
                }
                __syncthreads();
                n = m;
                for (uint32_t i = lane; i < n; i += kWave) lst[i] = srt[i];
                __syncthreads();
            }
            if (n > gstride) n = gstride;
            for (uint32_t i = lane; i < n; i += kWave) rec[4 + i] = lst[i];
            ep_n = n;
            if (lane == 0) spec_stamp(spec_wpos(A.wpos0, A.wposU, lid), epoch_key, b, shrunk ? 1u : 0u);
            __syncthreads();
        }
        if (lane == 0) {
            rec[0] = ep;
            rec[1] = 1u;  // the owner's list: [ep]
            rec[2] = ep_n == 0xFFFFFFFFu ? 0u : ep_n;
            rec[3] = ep_n == 0xFFFFFFFFu ? 0u : 1u;
        }
    }
    __syncthreads();
    for (uint32_t i = lane; i < min(rn, kSpecRMax); i += kWave) {
        uint32_t *e = A.rset + ((size_t)b * kSpecRMax + i) * 4;
        e[0] = rs[i];
        e[1] = __float_as_uint(rbd[i]);
        e[2] = rid[i];
        e[3] = rly[i];
    }
    if (lane == 0) A.rcnt[b] = rn;
}

// is position b of the round one a sequential run would have led another way?  (wave-uniform result; see k_insert_spec)
// A list b only scanned and that an earlier point SHRANK (reordered) is scanned again here, as the earlier point's record leaves it:
// the hop is unchanged iff it ends on the same node at the same distance (or still finds nothing closer) -- every insert scans node
// 0's list and one in ten ends there and shrinks it, so without this the rounds stop at the first such point.
// HUB (the cooperative loop): a point whose CONNECT list an earlier point of the round wrote too -- both end on the same node, the
// reference's graphs are full of such hubs -- is not a conflict: its trajectory is checked like any other (the scan of that list
// included), only its record of the list is void; the bit of that layer is set in defer[b], the commit step leaves the list
// alone, and after it ONE workgroup redoes the deferred connects in position order on the lists as they then are (k_insert_spec_loop).
// What a scan can still not be checked against is a list with TWO earlier writers (the second one's list exists only after the commit).
template <bool COH, bool HUB>
__device__ __forceinline__ bool check_body(const SpecArgs &A, uint32_t b, uint32_t epoch_key, uint32_t *why = nullptr, uint32_t *any_deferred = nullptr) {
    const int lane = threadIdx.x;
    if (b == 0) return false;  // the first point of a round read the committed graph only
    const uint32_t n = ldc<COH>(A.rcnt + b);
    bool bad = n > kSpecRMax;
    uint32_t cause = bad ? 8u : 0u;  // (diagnostic: 1 connect list, 2 a re-scanned list leads elsewhere, 4 beaten by an appended point, 8 read-set overflow)
    const float *myrow = A.rows + (size_t)(A.first_owner + b) * A.ld;
    const uint32_t rec_words = 4 + A.gstride;
    for (uint32_t i0 = 0; i0 < min(n, kSpecRMax); i0 += kWave) {
        const uint32_t i = i0 + lane;
        bool rescan = false;
        uint32_t wb = 0, bid = 0, layer = 0;
        float bd = 0.0f;
        if (i < min(n, kSpecRMax)) {
            const uint32_t *e = A.rset + ((size_t)b * kSpecRMax + i) * 4;
            const uint32_t r = ldc<COH>(e);
            const unsigned long long v = ldc64<COH>(spec_wpos(A.wpos0, A.wposU, r & 0xBFFFFFFFu));
            wb = ((uint32_t)v >> 1) & 0x7FFu;
            const uint32_t wb2 = ((uint32_t)v >> 12) & 0x7FFu;
            if ((uint32_t)(v >> 32) == epoch_key && wb < b) {  // written this round by an earlier point
                bd = __uint_as_float(ldc<COH>(e + 1));
                if (r & 0x40000000u) {  // my connect list
                    if constexpr (HUB) { atomicOr(A.defer + b, 1u << ldc<COH>(e + 3)); if (any_deferred) atomicOr(any_deferred, 1u); }
                    else { bad = true; cause |= 1u; }
                }
                else if (wb2 < b) { bad = true; cause |= 1u; }  // scanned, and two earlier points wrote it
                else if ((uint32_t)v & 1u) { rescan = true; bid = ldc<COH>(e + 2); layer = ldc<COH>(e + 3); }  // shrunk: scan it again
                else {  // the writer was appended at its end: the hop is unchanged unless the writer beats the scan's best
                    const float d = exact_dist_rt(A.metric, myrow, A.rows + (size_t)(A.first_owner + wb) * A.ld, A.dim);
                    if (d < bd) { bad = true; cause |= 4u; }
                }
            }
        }
        unsigned long long todo = __ballot(rescan);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t wbl = (uint32_t)__shfl((int)wb, l), bidl = (uint32_t)__shfl((int)bid, l), lyl = (uint32_t)__shfl((int)layer, l);
            const float bdl = __shfl(bd, l);
            const uint32_t *rec = A.recs + (size_t)(A.params[wbl].rec_off + lyl) * rec_words;  // the list as the earlier point leaves it
            const uint32_t ln = min(ldc<COH>(rec + 2), A.gstride);
            float md = __builtin_inff();
            uint32_t mj = 0xFFFFFFFFu, mid = 0;
            for (uint32_t base = 0; base < ln; base += kWave) {  // (the scan of spec_body: strict '<', the earliest of equal minima)
                const uint32_t j = base + lane;
                float d = __builtin_inff();
                uint32_t x = 0, jj = 0xFFFFFFFFu;
                if (j < ln) {
                    x = ldc<COH>(rec + 4 + j);
                    d = exact_dist_rt(A.metric, myrow, A.rows + (size_t)x * A.ld, A.dim);
                    jj = j;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float od = __shfl_xor(d, off);
                    const uint32_t oj = __shfl_xor(jj, off), ox = __shfl_xor(x, off);
                    if (od < d || (od == d && oj < jj)) { d = od; jj = oj; x = ox; }
                }
                if (jj != 0xFFFFFFFFu && d < md) { md = d; mj = jj; mid = x; }
            }
            const bool same = bidl == 0xFFFFFFFFu ? !(mj != 0xFFFFFFFFu && md < bdl) : (mj != 0xFFFFFFFFu && mid == bidl && md == bdl);
            if (!same) { bad = true; cause |= 2u; }
        }
    }
    if (why) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cause |= __shfl_xor(cause, off);
        *why = cause;
    }
    return __ballot(bad) != 0;
}

// position b writes what it recorded: ep's list and degree per layer (where ep has the layer), its own list [ep]
template <bool COH, bool HUB>
__device__ __forceinline__ void commit_body(const SpecArgs &A, uint32_t b) {
    const int lane = threadIdx.x;
    const uint32_t gstride = A.gstride;
    const uint32_t deferred = HUB ? ldc<COH>(A.defer + b) : 0u;
    const uint32_t owner = A.first_owner + b, level = A.params[b].level, my_slot = A.params[b].slot, rec0 = A.params[b].rec_off, maxlev = A.params[b].maxlev_pre;
    const uint32_t rec_words = 4 + gstride;
    for (uint32_t layer = 0; layer <= min(level, maxlev); ++layer) {
        const uint32_t *rec = A.recs + (size_t)(rec0 + layer) * rec_words;
        const uint32_t ep = ldc<COH>(rec + 0);
        uint32_t *ol = layer == 0 ? A.adj + (size_t)owner * gstride : A.adj_up + (size_t)(my_slot + layer - 1) * gstride;
        uint32_t *od = layer == 0 ? A.deg + owner : A.deg_up + (my_slot + layer - 1);
        if (lane == 0) { ol[0] = ep; *od = 1u; }
        if (ldc<COH>(rec + 3) && !((deferred >> layer) & 1u)) {
            const uint32_t n = ldc<COH>(rec + 2);
            const uint32_t es = layer == 0 ? 0u : ldc<COH>(A.slot + ep) + layer - 1;
            uint32_t *el = layer == 0 ? A.adj + (size_t)ep * gstride : A.adj_up + (size_t)es * gstride;
            uint32_t *ed = layer == 0 ? A.deg + ep : A.deg_up + es;
            for (uint32_t i = lane; i < n; i += kWave) el[i] = ldc<COH>(rec + 4 + i);
            if (lane == 0) *ed = n;
        }
    }
}

// a deferred connect of position b at `layer`, on ep's list AS IT IS NOW: append, shrink (:143-170), write it and put it in b's record
__device__ __forceinline__ void spec_merge_one(const SpecArgs &A, uint32_t b, uint32_t layer, unsigned char *smem) {
    const int lane = threadIdx.x;
    const uint32_t gstride = A.gstride, m = A.m, cap = (gstride + 1 + 3) & ~3u;
    uint32_t *lst = reinterpret_cast<uint32_t *>(smem);
    float *dst = reinterpret_cast<float *>(lst + cap);
    uint32_t *srt = reinterpret_cast<uint32_t *>(dst + cap);
    const uint32_t owner = A.first_owner + b;
    uint32_t *rec = A.recs + (size_t)(A.params[b].rec_off + layer) * (4 + gstride);
    const uint32_t ep = ldc<false>(rec);
    const uint32_t es = layer == 0 ? 0u : ldc<false>(A.slot + ep) + layer - 1;
    uint32_t *el = layer == 0 ? A.adj + (size_t)ep * gstride : A.adj_up + (size_t)es * gstride;
    uint32_t *ed = layer == 0 ? A.deg + ep : A.deg_up + es;
    const uint32_t n0 = min(ldc<false>(ed), gstride);
    __syncthreads();
    for (uint32_t i = lane; i < n0; i += kWave) lst[i] = ldc<false>(el + i);
    if (lane == 0) lst[n0] = owner;
    uint32_t n = n0 + 1;
    __syncthreads();
    if (n > m) {
        const float *o = A.rows + (size_t)ep * A.ld;
        for (uint32_t i = lane; i < n; i += kWave) dst[i] = exact_dist_rt(A.metric, o, A.rows + (size_t)lst[i] * A.ld, A.dim);
        __syncthreads();
        for (uint32_t i = lane; i < n; i += kWave) {
            const float di = dst[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; ++j) {
                const float dj = dst[j];
                rank += (dj < di || (dj == di && j < i)) ? 1u : 0u;
            }
            srt[rank] = lst[i];
        }
        __syncthreads();
        n = m;
        for (uint32_t i = lane; i < n; i += kWave) lst[i] = srt[i];
        __syncthreads();
    }
    if (n > gstride) n = gstride;
    for (uint32_t i = lane; i < n; i += kWave) { el[i] = lst[i]; rec[4 + i] = lst[i]; }
    if (lane == 0) { *ed = n; rec[2] = n; }
    __threadfence();  // the next deferred connect may be on the same list
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_insert_spec(SpecArgs A, uint32_t epoch_key) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    spec_body<false>(A, blockIdx.x, epoch_key, smem);
}
// c = first position of the round that a sequential run would have led another way; cmin starts at B
__global__ __launch_bounds__(64) void k_insert_check(SpecArgs A, uint32_t epoch_key, uint32_t *cmin) {
    if (check_body<false, false>(A, blockIdx.x, epoch_key) && threadIdx.x == 0) atomicMin(cmin, blockIdx.x);
}
__global__ __launch_bounds__(64) void k_insert_commit(SpecArgs A, const uint32_t *__restrict__ cmin) {
    if (blockIdx.x < *cmin) commit_body<false, false>(A, blockIdx.x);
}

// The rounds of a whole chunk in ONE cooperative launch of B workgroups (one wave each): speculate | check | commit with a grid
// barrier between the phases (and, in rounds that deferred a connect, a fourth phase: workgroup 0 redoes them in order), the next
// round starting at the first point not committed -- no host round trip per round (the three-launch form above costs ~80 us per
// round, of which the GPU works ~35).  ctl: [0] barrier counter (zero at launch), [1], [2] the rounds' commit counts (even / odd
// rounds, preset to B), [3] first epoch, [4] out: rounds, [5] barrier time-out, [6], [7] "a connect was deferred" (even / odd
// rounds), [8..11] what stopped the rounds, [12] deferred connects redone.
// (returns false if the other workgroups never arrived -- ~10 s of polling: every workgroup then leaves the loop instead of
// spinning for ever, ctl[5] tells the host; a cooperative launch keeps them co-resident, so this is a guard, not a path)
__device__ __forceinline__ bool spec_grid_barrier(uint32_t *ctl, uint32_t nblocks, uint32_t &gen) {
    __shared__ uint32_t ok_s;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        gen += 1;
        atomicAdd(ctl, 1u);
        uint32_t spins = 0, ok = 1;
        while (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * nblocks) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 25) || __hip_atomic_load(ctl + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(ctl + 5, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        __threadfence();
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}
__global__ __launch_bounds__(64) void k_insert_spec_loop(SpecArgs A0, uint32_t n_chunk, uint32_t *ctl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t B = gridDim.x, b = blockIdx.x;
    const uint32_t epoch0 = __hip_atomic_load(ctl + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t gen = 0, pos = 0, round = 0;
    while (pos < n_chunk) {  // (pos, round: the same in every workgroup)
        SpecArgs A = A0;
        A.first_owner = A0.first_owner + pos;
        A.points = A0.points + (size_t)pos * A0.dim;
        A.params = A0.params + pos;
        // (recs offsets are chunk-wide: params[].rec_off; rset / rcnt are per position of the round)
        const uint32_t nb = min(B, n_chunk - pos);
        const uint32_t key = 0xFFFFFFFFu - (epoch0 + round);
        uint32_t *cm = ctl + 1 + (round & 1u);
        // (plain loads throughout: every phase starts behind a grid barrier whose agent-scope fences -- release before the arrival,
        // acquire after the last one -- make the other workgroups' writes of the previous phase visible, across XCDs too)
        if (b < nb) spec_body<false>(A, b, key, smem);
        if (b == 0 && threadIdx.x == 0) {
            __hip_atomic_store(ctl + 1 + ((round + 1) & 1u), B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctl + 6 + ((round + 1) & 1u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!spec_grid_barrier(ctl, B, gen)) return;
        uint32_t why = 0;
        const bool bad = b < nb && check_body<false, true>(A, b, key, &why, ctl + 6 + (round & 1u));
        if (bad && threadIdx.x == 0) atomicMin(cm, b);
        if (!spec_grid_barrier(ctl, B, gen)) return;
        const uint32_t c = min(__hip_atomic_load(cm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), nb);
        if (bad && b == c && threadIdx.x == 0) {  // diagnostic: why the round stopped here (ctl[8 + cause bit])
            for (uint32_t bit = 0; bit < 4; ++bit)
                if (why & (1u << bit)) atomicAdd(ctl + 8 + bit, 1u);
        }
        const bool any_deferred = __hip_atomic_load(ctl + 6 + (round & 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (b < c) commit_body<false, true>(A, b);
        if (!spec_grid_barrier(ctl, B, gen)) return;
        if (any_deferred) {  // (the same in every workgroup) the deferred connects of the committed points, in position order
            if (b == 0) {
                uint32_t redone = 0;
                for (uint32_t j = 1; j < c; ++j) {
                    uint32_t dm = A.defer[j];
                    while (dm) {
                        const uint32_t layer = (uint32_t)__builtin_ctz(dm);
                        dm &= dm - 1;
                        spec_merge_one(A, j, layer, smem);
                        redone += 1;
                    }
                }
                if (threadIdx.x == 0 && redone) atomicAdd(ctl + 12, redone);
            }
            if (!spec_grid_barrier(ctl, B, gen)) return;
        }
        pos += c;  // c >= 1: position 0 is always valid
        round += 1;
    }
    if (b == 0 && threadIdx.x == 0) {
        __hip_atomic_store(ctl + 3, epoch0 + round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ctl[4] = round;
    }
}

static SpecArgs make_spec_args(const zvdb_hip_index *h, uint32_t first_owner, const float *points, const InsertParam *params, uint32_t *recs,
                               unsigned long long *wpos0, unsigned long long *wposU, uint32_t *rset, uint32_t *rcnt) {
    SpecArgs A;
    A.rows = h->rows; A.ld = h->ld; A.dim = h->dim; A.metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    A.adj = h->adj; A.deg = h->deg; A.adj_up = h->adj_up; A.deg_up = h->deg_up; A.lvl = h->lvl; A.slot = h->slot; A.gstride = h->gstride; A.m = h->m;
    A.first_owner = first_owner; A.points = points; A.params = params; A.recs = recs;
    A.wpos0 = wpos0; A.wposU = wposU; A.rset = rset; A.rcnt = rcnt; A.defer = rcnt + 1024;  // (rcnt and defer share one block: [1024] each)
    return A;
}
static size_t spec_lds_bytes(const zvdb_hip_index *h) {
    return (3ull * ((h->gstride + 1 + 3) & ~3u) + 4 * kSpecRMax) * sizeof(uint32_t) + (size_t)h->dim * sizeof(float);
}

int launch_insert_spec_round(const zvdb_hip_index *h, uint32_t first_owner, uint32_t B, const float *points, const InsertParam *params,
                             uint32_t *recs, unsigned long long *wpos0, unsigned long long *wposU, uint32_t epoch, uint32_t *rset,
                             uint32_t *rcnt, uint32_t *cmin, hipStream_t s) {
    const size_t lds = spec_lds_bytes(h);
    ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_insert_spec), lds));
    const uint32_t key = 0xFFFFFFFFu - epoch;
    const SpecArgs A = make_spec_args(h, first_owner, points, params, recs, wpos0, wposU, rset, rcnt);
    hipLaunchKernelGGL(k_insert_spec, dim3(B), dim3(64), lds, s, A, key);
    hipLaunchKernelGGL(k_insert_check, dim3(B), dim3(64), 0, s, A, key, cmin);
    hipLaunchKernelGGL(k_insert_commit, dim3(B), dim3(64), 0, s, A, cmin);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// the whole chunk in one cooperative launch; ctl = 8 device words (see k_insert_spec_loop): set here; after it ran, ctl[4] = rounds used
int launch_insert_spec_loop(const zvdb_hip_index *h, uint32_t first_owner, uint32_t n_chunk, uint32_t B, const float *points,
                            const InsertParam *params, uint32_t *recs, unsigned long long *wpos0, unsigned long long *wposU, uint32_t *rset,
                            uint32_t *rcnt, uint32_t *ctl, uint32_t epoch, hipStream_t s) {
    const size_t lds = spec_lds_bytes(h);
    ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_insert_spec_loop), lds));
    const uint32_t init[13] = {0u, B, B, epoch, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    ZV_HIP(hipMemcpyAsync(ctl, init, sizeof(init), hipMemcpyHostToDevice, s));
    SpecArgs A = make_spec_args(h, first_owner, points, params, recs, wpos0, wposU, rset, rcnt);
    void *args[] = {&A, &n_chunk, &ctl};
    ZV_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_insert_spec_loop), dim3(B), dim3(64), args, (unsigned int)lds, s));
    return ZVDB_OK;
}
uint32_t insert_spec_rmax() { return kSpecRMax; }

__global__ __launch_bounds__(256) void k_copy_row(const float *__restrict__ src, float *__restrict__ dst, uint32_t dim) {
    for (uint32_t i = threadIdx.x; i < dim; i += 256) dst[i] = src[i];
}

int launch_copy_row(const float *src, float *dst, uint32_t dim, hipStream_t s) {
    hipLaunchKernelGGL(k_copy_row, dim3(1), dim3(256), 0, s, src, dst, dim);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

__global__ __launch_bounds__(64) void k_publish(unsigned long long *__restrict__ dst, const unsigned long long *__restrict__ src,
                                                uint32_t n, uint32_t *__restrict__ seq_dst, uint32_t seq) {
    for (uint32_t i = threadIdx.x; i < n; i += 64) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(seq_dst, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int launch_publish(unsigned long long *dst, const unsigned long long *src, uint32_t n, uint32_t *seq_dst, uint32_t seq, hipStream_t s) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, dst, src, n, seq_dst, seq);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

size_t insert_l0_out_bytes(const zvdb_hip_index *h) { return round_up64(sizeof(InsertL0Out) + 2ull * h->gstride * sizeof(uint32_t), 8); }

int launch_insert_layer0(const zvdb_hip_index *h, uint32_t owner, uint32_t ep, void *d_out, uint32_t seq, const float *point_src,
                         hipStream_t s, uint32_t n_batch, uint32_t out_stride) {
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    const size_t lists = 3ull * ((h->gstride + 1 + 3) & ~3u) * sizeof(uint32_t), dimb = (size_t)((h->dim + 3) & ~3u) * sizeof(float);
    if (h->flags & ZVDB_FLAG_BF16_ROWS) {
        if (point_src) { set_error("fused point copy needs fp32 rows"); return ZVDB_ERR_INVALID; }
        const size_t lds = lists + 2 * dimb;
        ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_insert_layer0<bf16_t>), lds));
        hipLaunchKernelGGL(k_insert_layer0<bf16_t>, dim3(1), dim3(64), lds, s, reinterpret_cast<bf16_t *>(h->rows_h), h->ldh, h->dim, metric, h->adj,
                           h->deg, h->gstride, h->m, owner, ep, static_cast<unsigned char *>(d_out), seq, point_src, n_batch,
                           out_stride ? out_stride : (uint32_t)insert_l0_out_bytes(h));
    } else {
        const size_t lds = lists + (point_src ? dimb : 0);
        ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_insert_layer0<float>), lds));
        hipLaunchKernelGGL(k_insert_layer0<float>, dim3(1), dim3(64), lds, s, h->rows, h->ld, h->dim, metric, h->adj, h->deg, h->gstride,
                           h->m, owner, ep, static_cast<unsigned char *>(d_out), seq, point_src, n_batch,
                           out_stride ? out_stride : (uint32_t)insert_l0_out_bytes(h));
    }
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

int launch_greedy_layer0(const zvdb_hip_index *h, uint32_t owner, uint32_t *d_ep_io,
                         float *d_dist_io, unsigned long long *d_evals, hipStream_t s) {
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    return rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = std::remove_cv_t<std::remove_pointer_t<decltype(rows)>>;
        const size_t lds = (size_t)round_up(std::max(h->dim, 1u), 4) * sizeof(float);
        ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_greedy_layer0<RT>), lds));
        hipLaunchKernelGGL(k_greedy_layer0<RT>, dim3(1), dim3(64), lds, s, rows, ld, h->dim, metric, h->adj, h->deg, h->gstride, owner, d_ep_io,
                           d_dist_io, d_evals);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    });
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

// The same table with ANY slot count (k_search_ef_reg: the LDS a query's table takes sets the queries per CU, and a power of two
// wastes up to half of it): home slot = high half of hash x size, probing wraps at size.
__device__ __forceinline__ uint32_t hashn_slot(uint32_t id, uint32_t size) { return __umulhi(id * 2654435761u, size); }
__device__ __forceinline__ void hashn_insert_uniform(uint32_t *tab, uint32_t size, uint32_t id) {
    uint32_t s = hashn_slot(id, size);
    for (;;) {
        const uint32_t v = tab[s];
        if (v == 0) { tab[s] = id + 1; return; }
        if (v == id + 1) return;
        s = s + 1 == size ? 0 : s + 1;
    }
}
__device__ __forceinline__ bool hashn_insert_atomic(uint32_t *tab, uint32_t size, uint32_t id) {
    uint32_t s = hashn_slot(id, size);
    for (;;) {
        const uint32_t old = atomicCAS(&tab[s], 0u, id + 1);
        if (old == 0) return true;
        if (old == id + 1) return false;
        s = s + 1 == size ? 0 : s + 1;
    }
}

// Visited set of the LARGE beams (k_search_ef_reg with 8 / 16 beam blocks: ef > 256), r4: 21-bit entries, three to a 64-bit word.
// The table's LDS decides how many queries a CU holds (a third of HBM at ef = 512 with two, an eighth at ef = 1,024 with one), and a
// 32-bit slot spends a third of its bits on what the slot's position already says.  A node id is first sent through a BIJECTION
// of [0, 2^b) (b = bits of the row count: multiply by an odd constant, xor with the own right shift, twice); the low r = max(0, b -
// 20) bits of the image pick one of 2^r regions of the table, the remaining <= 20 bits (+ 1; 0 = empty) are what is stored, and an
// entry never leaves its region (bucketised linear probing over the region's words, wrapping inside it).  (region, stored bits)
// give the image and the image gives the id, so membership is EXACT -- no fingerprint collisions -- at 21.3 bits per entry.  A region
// that fills up (cw words, all three fields taken) reports 2: the caller hands the query to the next pass like any full table.
// Fields of a word fill in the order 0, 1, 2 and are never cleared; lanes of the wave meet in one 64-bit compare-and-swap.
__device__ __forceinline__ uint32_t hp_mix(uint32_t id, uint32_t bmask, uint32_t half) {
    uint32_t h = (id * 0x9E3779B1u) & bmask;
    h ^= h >> half;
    h = (h * 0x85EBCA6Bu) & bmask;
    h ^= h >> half;
    return h;
}
// 0: the id was there; 1: THIS lane put it in; 2: its region is full
__device__ __forceinline__ uint32_t hp_insert_atomic(unsigned long long *tab, uint32_t cw, uint32_t r, uint32_t bmask, uint32_t half, uint32_t id) {
    const uint32_t h = hp_mix(id, bmask, half);
    const uint32_t quo = h >> r;  // < 2^20
    const unsigned long long tag = (unsigned long long)quo + 1ull;
    unsigned long long *reg = tab + (size_t)(h & ((1u << r) - 1u)) * cw;
    uint32_t w = __umulhi(quo << 12, cw);  // home word: quo cw / 2^20
    for (uint32_t n = 0; n < cw; ++n) {
        unsigned long long v = reg[w];
        for (;;) {
            const uint32_t f0 = (uint32_t)v & 0x1FFFFFu, f1 = (uint32_t)(v >> 21) & 0x1FFFFFu, f2 = (uint32_t)(v >> 42) & 0x1FFFFFu;
            if (f0 == (uint32_t)tag || f1 == (uint32_t)tag || f2 == (uint32_t)tag) return 0;
            const int sh = f0 == 0 ? 0 : f1 == 0 ? 21 : f2 == 0 ? 42 : -1;
            if (sh < 0) break;  // the word is full: on to the next
            const unsigned long long old = atomicCAS(&reg[w], v, v | (tag << sh));
            if (old == v) return 1;
            v = old;  // another lane's entry went into this word meanwhile: look again
        }
        w = w + 1 == cw ? 0 : w + 1;
    }
    return 2;
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
template <typename RT>
__device__ __forceinline__ void compat_heap_body(
    const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
    const uint32_t *__restrict__ adj, const uint32_t *__restrict__ deg, uint32_t gstride,
    uint32_t N, const float *__restrict__ queries, uint32_t k, uint32_t entry, uint64_t id_offset,
    uint64_t *__restrict__ out_ids, float *__restrict__ out_dist, uint32_t *__restrict__ out_count,
    unsigned long long *__restrict__ stats, uint32_t dim_pad, uint32_t heap_cap, uint32_t hash_bits,
    uint32_t rcap, uint32_t qi_arg, unsigned char *smem) {
    float *q = reinterpret_cast<float *>(smem);
    float *heap_d = q + dim_pad;
    uint32_t *heap_i = reinterpret_cast<uint32_t *>(heap_d + heap_cap);
    uint32_t *hash = heap_i + heap_cap;
    float *res_d = reinterpret_cast<float *>(hash + (1u << hash_bits));
    uint32_t *res_i = reinterpret_cast<uint32_t *>(res_d + rcap);
    float *nb_d = reinterpret_cast<float *>(res_i + rcap);
    uint32_t *nb_i = reinterpret_cast<uint32_t *>(nb_d + kWave);

    const int lane = threadIdx.x;
    const uint32_t qi = qi_arg;
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
    }
    // stable insertion sort by distance (hnsw.zig:227-233; the comparator's recomputed distance is bit-identical to the
    // stored one), done by ranks -- #{d_j < d_i} + #{j < i : d_j == d_i} is the position a stable sort gives entry i
    for (uint32_t i = lane; i < k; i += kWave) {
        out_ids[(size_t)qi * k + i] = ~0ull;
        out_dist[(size_t)qi * k + i] = __builtin_inff();
    }
    __syncthreads();
    for (uint32_t i = lane; i < nres; i += kWave) {
        const float di = res_d[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nres; ++j) {
            const float dj = res_d[j];
            rank += (dj < di || (dj == di && j < i)) ? 1u : 0u;
        }
        out_ids[(size_t)qi * k + rank] = (uint64_t)res_i[i] + id_offset;
        out_dist[(size_t)qi * k + rank] = di;
    }
    if (lane == 0) {
        out_count[qi] = nres;
        if (stats) {
            atomicAdd(&stats[0], evals);
            atomicAdd(&stats[1], pops);
        }
    }
}

// The same search with the frontier as an UNORDERED set: add = one LDS write per neighbour (all neighbours of a pop in
// parallel), pop = wave arg-min.  A binary heap's remove-min returns THE minimum whenever it is unique, so the pop sequence,
// the results and the evaluation counts are those of the PriorityQueue emulation above -- the heap's layout only decides
// among EQUAL minima.  When the minimum is not unique at a pop, this kernel gives the query up (tie[qi] = 1) and the exact
// emulation runs it instead (gated launch right behind this one): identical results either way, ~10x fewer dependent LDS
// round trips in the common case.
template <typename RT>
__device__ __forceinline__ bool compat_fast_body(
    const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
    const uint32_t *__restrict__ adj, const uint32_t *__restrict__ deg, uint32_t gstride,
    uint32_t N, const float *__restrict__ queries, uint32_t k, uint32_t entry, uint64_t id_offset,
    uint64_t *__restrict__ out_ids, float *__restrict__ out_dist, uint32_t *__restrict__ out_count,
    unsigned long long *__restrict__ stats, uint32_t dim_pad, uint32_t heap_cap, uint32_t hash_bits,
    uint32_t rcap, uint32_t *__restrict__ tie, uint32_t qi_arg, unsigned char *smem) {
    float *q = reinterpret_cast<float *>(smem);
    float *fr_d = q + dim_pad;
    uint32_t *fr_i = reinterpret_cast<uint32_t *>(fr_d + heap_cap);
    uint32_t *hash = fr_i + heap_cap;
    float *res_d = reinterpret_cast<float *>(hash + (1u << hash_bits));
    uint32_t *res_i = reinterpret_cast<uint32_t *>(res_d + rcap);

    const int lane = threadIdx.x;
    const uint32_t qi = qi_arg;
    const float *qg = queries + (size_t)qi * dim;
    for (uint32_t i = lane; i < dim_pad; i += kWave) q[i] = i < dim ? qg[i] : 0.0f;
    for (uint32_t i = lane; i < (1u << hash_bits); i += kWave) hash[i] = 0;
    __syncthreads();

    uint32_t nres = 0, n = 0;
    unsigned long long evals = 0, pops = 0;
    bool gave_up = false;
    if (N > 0 && k > 0) {
        float d0 = 0.0f;
        if (lane == 0) {
            d0 = exact_dist_rt(metric, q, rows + (size_t)entry * ld, dim);  // :208
            fr_d[0] = d0;
            fr_i[0] = entry;
            hash_insert_atomic(hash, hash_bits, entry);                     // :209
        }
        n = 1;
        evals = 1;
        __syncthreads();
        while (n > 0 && nres < k) {                                         // :211
            // pop: the minimum and how many entries hold it
            float bd = __builtin_inff();
            uint32_t bp = 0xFFFFFFFFu;
            for (uint32_t i = lane; i < n; i += kWave) {
                const float d = fr_d[i];
                if (d < bd || bp == 0xFFFFFFFFu) { bd = d; bp = i; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float od = __shfl_xor(bd, off);
                const uint32_t op = __shfl_xor(bp, off);
                if (op != 0xFFFFFFFFu && (bp == 0xFFFFFFFFu || od < bd || (od == bd && op < bp))) { bd = od; bp = op; }
            }
            uint32_t same = 0;
            for (uint32_t i = lane; i < n; i += kWave) same += fr_d[i] == bd ? 1u : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) same += __shfl_xor(same, off);
            if (same != 1) { gave_up = true; break; }  // equal minima (or a NaN): the heap's layout decides -> exact emulation
            const uint32_t cid = fr_i[bp];
            __syncthreads();
            if (lane == 0) {
                fr_d[bp] = fr_d[n - 1];
                fr_i[bp] = fr_i[n - 1];
                res_i[nres] = cid;                                          // :213-214
                res_d[nres] = bd;
            }
            n -= 1;
            nres += 1;
            pops += 1;
            __syncthreads();
            // (the degree and the first 64 list slots are requested together: both depend on cid only)
            const uint32_t nb0 = (uint32_t)lane < gstride ? adj[(size_t)cid * gstride + lane] : 0u;
            const uint32_t dg = deg[cid];                                   // :216 layer 0
            for (uint32_t base = 0; base < dg; base += kWave) {
                const uint32_t j = base + lane;
                bool fresh = false;
                uint32_t nb = 0;
                float d = 0.0f;
                if (j < dg) {
                    nb = base == 0 ? nb0 : adj[(size_t)cid * gstride + j];
                    if (!hash_contains(hash, hash_bits, nb)) {              // :217
                        d = exact_dist_rt(metric, q, rows + (size_t)nb * ld, dim);  // :219
                        fresh = hash_insert_atomic(hash, hash_bits, nb);    // :221 (one owner per id)
                    }
                }
                const unsigned long long fm = __ballot(fresh);
                if (fresh) {
                    const uint32_t pos = n + (uint32_t)__popcll(fm & ((1ull << lane) - 1));
                    fr_d[pos] = d;                                          // :220
                    fr_i[pos] = nb;
                }
                n += (uint32_t)__popcll(fm);
                evals += (unsigned long long)__popcll(fm);
                __syncthreads();
            }
        }
    }
    if (gave_up) {
        if (lane == 0) tie[qi] = 1;
        return false;
    }
    if (lane == 0) tie[qi] = 0;
    __syncthreads();
    // stable sort by distance (hnsw.zig:227-233) by ranks: #{d_j < d_i} + #{j < i : d_j == d_i}
    for (uint32_t i = lane; i < k; i += kWave) {
        out_ids[(size_t)qi * k + i] = ~0ull;
        out_dist[(size_t)qi * k + i] = __builtin_inff();
    }
    __syncthreads();
    for (uint32_t i = lane; i < nres; i += kWave) {
        const float di = res_d[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nres; ++j) {
            const float dj = res_d[j];
            rank += (dj < di || (dj == di && j < i)) ? 1u : 0u;
        }
        out_ids[(size_t)qi * k + rank] = (uint64_t)res_i[i] + id_offset;
        out_dist[(size_t)qi * k + rank] = di;
    }
    if (lane == 0) {
        out_count[qi] = nres;
        if (stats) {
            atomicAdd(&stats[0], evals);
            atomicAdd(&stats[1], pops);
        }
    }
    return true;
}


template <typename RT>
__global__ __launch_bounds__(64) void k_search_compat(
    const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
    const uint32_t *__restrict__ adj, const uint32_t *__restrict__ deg, uint32_t gstride,
    uint32_t N, const float *__restrict__ queries, uint32_t k, uint32_t entry, uint64_t id_offset,
    uint64_t *__restrict__ out_ids, float *__restrict__ out_dist, uint32_t *__restrict__ out_count,
    unsigned long long *__restrict__ stats, uint32_t dim_pad, uint32_t heap_cap, uint32_t hash_bits,
    uint32_t rcap, const uint32_t *__restrict__ gate /* optional: run only the queries the fast kernel gave up on */) {
    if (gate && gate[blockIdx.x] == 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    compat_heap_body<RT>(rows, ld, dim, metric, adj, deg, gstride, N, queries, k, entry, id_offset, out_ids, out_dist, out_count, stats, dim_pad,
                         heap_cap, hash_bits, rcap, blockIdx.x, smem);
}
template <typename RT>
__global__ __launch_bounds__(64) void k_search_compat_fast(
    const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
    const uint32_t *__restrict__ adj, const uint32_t *__restrict__ deg, uint32_t gstride,
    uint32_t N, const float *__restrict__ queries, uint32_t k, uint32_t entry, uint64_t id_offset,
    uint64_t *__restrict__ out_ids, float *__restrict__ out_dist, uint32_t *__restrict__ out_count,
    unsigned long long *__restrict__ stats, uint32_t dim_pad, uint32_t heap_cap, uint32_t hash_bits,
    uint32_t rcap, uint32_t *__restrict__ tie) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    compat_fast_body<RT>(rows, ld, dim, metric, adj, deg, gstride, N, queries, k, entry, id_offset, out_ids, out_dist, out_count, stats, dim_pad,
                         heap_cap, hash_bits, rcap, tie, blockIdx.x, smem);
}
// ONE query per call (the reference's calling pattern, hnsw.zig:194 / shared_benchmarks.zig:104-109), ONE launch: what the batched
// route does with a memset, the unordered-frontier kernel, the gated heap emulation and a publish kernel -- four launches and an
// event pair, ~15 us of a 57-us call (profiles/r4/per_call_*) -- in one wave: clear the counters, run the unordered-frontier search,
// fall back to the PriorityQueue emulation if a pop's minimum was not unique, copy the counters next to the results (host-mapped
// memory) and release the sequence word the host polls.
// (amdgpu_waves_per_eu(1, 1): ONE wave runs on the whole device -- without it hipcc schedules for occupancy, keeps the kernel at 75
// VGPRs and serialises the 16 row loads a lane has in flight: 52 -> 42 us per call, r4)
template <typename RT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_search_compat_single(
    const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
    const uint32_t *__restrict__ adj, const uint32_t *__restrict__ deg, uint32_t gstride,
    uint32_t N, const float *__restrict__ queries, uint32_t k, uint32_t entry,
    uint64_t *__restrict__ out_ids, float *__restrict__ out_dist, uint32_t *__restrict__ out_count,
    unsigned long long *__restrict__ stats /* device: 2 words */, uint32_t dim_pad, uint32_t heap_cap, uint32_t hash_bits,
    uint32_t rcap, uint32_t *__restrict__ tie /* device: 1 word */, int fast_ok, unsigned long long *__restrict__ host_stats,
    uint32_t *__restrict__ seq_dst, uint32_t seq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (threadIdx.x == 0) {
        __hip_atomic_store(&stats[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&stats[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(tie, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (one wave: its stores and atomics reach the L2 in program order, and a barrier waits for them -- no device-scope fence, which
    // on this part writes the XCD's L2 back: three of them cost this kernel 11 us)
    __syncthreads();
    bool done = false;  // (wave-uniform)
    if (fast_ok)
        done = compat_fast_body<RT>(rows, ld, dim, metric, adj, deg, gstride, N, queries, k, entry, 0ull, out_ids, out_dist, out_count, stats, dim_pad,
                                    heap_cap, hash_bits, rcap, tie, 0u, smem);
    __syncthreads();
    if (!done) {
        compat_heap_body<RT>(rows, ld, dim, metric, adj, deg, gstride, N, queries, k, entry, 0ull, out_ids, out_dist, out_count, stats, dim_pad,
                             heap_cap, hash_bits, rcap, 0u, smem);
    }
    __syncthreads();
    if (threadIdx.x < 2) host_stats[threadIdx.x] = __hip_atomic_load(&stats[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(seq_dst, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static uint32_t bits_for(uint64_t entries) {  // table with >= 2x entries, at least 64 slots
    uint32_t b = 6;
    while ((1ull << b) < entries * 2) ++b;
    return b;
}

int launch_search_compat(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t k,
                         uint32_t entry, uint64_t id_offset, uint64_t *d_ids, float *d_dist,
                         uint32_t *d_count, unsigned long long *d_stats, hipStream_t s, uint32_t *d_tie) {
    if (nq == 0) return ZVDB_OK;
    const uint64_t N = h->count;
    const uint32_t rcap = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(k, N));
    const uint64_t bound = std::max<uint64_t>(1, std::min<uint64_t>(N, 1 + (uint64_t)rcap * h->gstride));
    const uint32_t heap_cap = (uint32_t)bound;
    const uint32_t hash_bits = bits_for(bound);
    const uint32_t dim_pad = round_up(std::max(h->dim, 1u), 4);
    const size_t lds = 4ull * (dim_pad + 2ull * heap_cap + (1ull << hash_bits) + 2ull * rcap + 2 * kWave);
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    const bool fast_off = [] { const char *e = tuning("ZVDB_COMPAT_SEARCH"); return e && e[0] == 'h'; }();  // "heap": emulation only
    return rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = std::remove_cv_t<std::remove_pointer_t<decltype(rows)>>;
        ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_search_compat<RT>), lds));
        uint32_t *tie = d_tie;
        if (tie && !fast_off) {  // unordered-frontier kernel first; the exact emulation serves the queries it gave up on
            ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_search_compat_fast<RT>), lds));
            hipLaunchKernelGGL(k_search_compat_fast<RT>, dim3(nq), dim3(64), lds, s, rows, ld, h->dim, metric,
                               h->adj, h->deg, h->gstride, (uint32_t)N, d_queries, k, entry, id_offset, d_ids,
                               d_dist, d_count, d_stats, dim_pad, heap_cap, hash_bits, rcap, tie);
        } else {
            tie = nullptr;
        }
        hipLaunchKernelGGL(k_search_compat<RT>, dim3(nq), dim3(64), lds, s, rows, ld, h->dim, metric,
                           h->adj, h->deg, h->gstride, (uint32_t)N, d_queries, k, entry, id_offset, d_ids,
                           d_dist, d_count, d_stats, dim_pad, heap_cap, hash_bits, rcap, tie);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    });
}

int launch_search_compat_single(const zvdb_hip_index *h, const float *d_query, uint32_t k, uint32_t entry, uint64_t *d_ids, float *d_dist,
                                uint32_t *d_count, unsigned long long *d_stats, uint32_t *d_tie, unsigned long long *host_stats, uint32_t *seq_dst,
                                uint32_t seq, hipStream_t s) {
    const uint64_t N = h->count;
    const uint32_t rcap = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(k, N));
    const uint64_t bound = std::max<uint64_t>(1, std::min<uint64_t>(N, 1 + (uint64_t)rcap * h->gstride));
    const uint32_t heap_cap = (uint32_t)bound;
    const uint32_t hash_bits = bits_for(bound);
    const uint32_t dim_pad = round_up(std::max(h->dim, 1u), 4);
    const size_t lds = 4ull * (dim_pad + 2ull * heap_cap + (1ull << hash_bits) + 2ull * rcap + 2 * kWave);
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    const bool fast_off = [] { const char *e = tuning("ZVDB_COMPAT_SEARCH"); return e && e[0] == 'h'; }();
    return rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = std::remove_cv_t<std::remove_pointer_t<decltype(rows)>>;
        ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_search_compat_single<RT>), lds));
        hipLaunchKernelGGL(k_search_compat_single<RT>, dim3(1), dim3(64), lds, s, rows, ld, h->dim, metric, h->adj, h->deg, h->gstride, (uint32_t)N,
                           d_query, k, entry, d_ids, d_dist, d_count, d_stats, dim_pad, heap_cap, hash_bits, rcap, d_tie, fast_off ? 0 : 1, host_stats,
                           seq_dst, seq);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    });
}

// ---- traversal distances of one list window (shared by both ef kernels) ---------------------------------------------------
// nb_d[j] <- blocked16 distance (oracle: orc_distance_blocked16_f32) between q (LDS, zero padded) and row nb_i[j] for every
// j < cnt (the fresh neighbours, compacted).  WPQ waves of a workgroup share the window: a step covers G * U * WPQ rows (G
// lane groups per wave, U rows per group), wave w takes rows r0 + (u WPQ + w) G + g.
//   fp32 rows: G = 4 groups of 16 lanes; lane t takes the 16-byte chunks t, t + 16, ... of its row, accumulates them in
//              increasing order (un-contracted), the 16 partials meet in the xor butterfly 8, 4, 2, 1.
//   bf16 rows: G = 8 groups of EIGHT lanes with 16-byte loads.  Lane a takes bytes 128 m + 16 a of the row (m = 0, 1, ...):
//              the 8-byte chunks 2a + 16m and 2a + 1 + 16m, i.e. exactly the chunks of the blocked16 partial sums 2a and
//              2a + 1 in increasing order -- the lane carries BOTH partials as two sequential chains, and the butterfly
//              over partial indices becomes lane xor 4, 2, 1 on each chain followed by low + high.  Same values as the
//              16-lane walk, half the load requests per row (rows are zero padded to a multiple of 128 bytes).
// All loads of a super-step (the whole row at d = 768) are issued before the first use, branch free, so the counted waits
// let the sums of piece i start while pieces i+1.. are still in flight.
template <typename RT, int WPQ, int U>
__device__ __forceinline__ void ef_window_distances(const RT *__restrict__ rows, uint32_t ld, const float *q, uint32_t nchunks, int metric,
                                                    uint32_t cnt, const uint32_t *nb_i, float *nb_d, int lane, int wave) {
    if constexpr (sizeof(RT) == 2) {
        constexpr int G = 8;
        const int g8 = lane >> 3, a8 = lane & 7;
        const uint32_t nm = ld / 64;  // 128-byte lines per row
        for (uint32_t r0 = 0; r0 < cnt; r0 += G * U * WPQ) {
            if (r0 + (uint32_t)wave * G >= cnt) break;  // (wave-uniform: nothing for this wave in the step)
            const bf16_t *rp[U];
            bool fr[U];
            float slo[U], shi[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t j = r0 + (uint32_t)(u * WPQ + wave) * G + g8;
                fr[u] = j < cnt;
                rp[u] = rows + (size_t)(fr[u] ? nb_i[j] : 0u) * ld + 8 * a8;
                slo[u] = 0.0f;
                shi[u] = 0.0f;
            }
            // MM lines per row and super-step; MM follows the row length -- 768-d rows are ONE super-step of 12.  When MM divides
            // the line count (every power-of-two length and 768 / 1536) the walk needs no masks at all: the caller pads q with
            // zeros to the rows' padded length, the rows' own padding is zero, so the tail adds +0.  Other lengths take the
            // MASKED walk: a line past the row's end re-reads the last one and its terms are forced to +0.
            auto walk = [&](auto mm_tag, auto masked_tag) {
                constexpr int MM = decltype(mm_tag)::value;
                constexpr bool MASKED = decltype(masked_tag)::value;
                for (uint32_t m0 = 0; m0 < nm; m0 += MM) {
                    uint4 vv[MM][U];
#pragma unroll
                    for (int i = 0; i < MM; ++i) {
                        const uint32_t me = MASKED ? min(m0 + i, nm - 1) : m0 + i;
#pragma unroll
                        for (int u = 0; u < U; ++u) vv[i][u] = *reinterpret_cast<const uint4 *>(rp[u] + 64 * me);
                    }
#pragma unroll
                    for (int i = 0; i < MM; ++i) {
                        const uint32_t cl = 2 * a8 + 16 * (m0 + i);
                        const bool okm = !MASKED || m0 + i < nm;  // (q is padded to whole lines: only a re-read line needs the mask)
                        const float4 al = *reinterpret_cast<const float4 *>(q + 4 * (MASKED ? min(cl, 16 * nm - 2) : cl));
                        const float4 ah = *reinterpret_cast<const float4 *>(q + 4 * (MASKED ? min(cl, 16 * nm - 2) : cl) + 4);
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const uint4 w = vv[i][u];
                            float l0 = bf16_lo(w.x), l1 = bf16_hi(w.x), l2 = bf16_lo(w.y), l3 = bf16_hi(w.y);
                            float h0 = bf16_lo(w.z), h1 = bf16_hi(w.z), h2 = bf16_lo(w.w), h3 = bf16_hi(w.w);
                            if (metric == ZVDB_METRIC_L2SQ) {
                                l0 = al.x - l0; l0 = l0 * l0; l1 = al.y - l1; l1 = l1 * l1;
                                l2 = al.z - l2; l2 = l2 * l2; l3 = al.w - l3; l3 = l3 * l3;
                                h0 = ah.x - h0; h0 = h0 * h0; h1 = ah.y - h1; h1 = h1 * h1;
                                h2 = ah.z - h2; h2 = h2 * h2; h3 = ah.w - h3; h3 = h3 * h3;
                            } else {
                                l0 = al.x * l0; l1 = al.y * l1; l2 = al.z * l2; l3 = al.w * l3;
                                h0 = ah.x * h0; h1 = ah.y * h1; h2 = ah.z * h2; h3 = ah.w * h3;
                            }
                            if (!okm) { l0 = 0.0f; l1 = 0.0f; l2 = 0.0f; l3 = 0.0f; h0 = 0.0f; h1 = 0.0f; h2 = 0.0f; h3 = 0.0f; }
                            slo[u] = slo[u] + l0; slo[u] = slo[u] + l1; slo[u] = slo[u] + l2; slo[u] = slo[u] + l3;
                            shi[u] = shi[u] + h0; shi[u] = shi[u] + h1; shi[u] = shi[u] + h2; shi[u] = shi[u] + h3;
                        }
                    }
                }
            };
            using std::integral_constant;
            if (nm % 12 == 0) walk(integral_constant<int, 12>{}, std::false_type{});
            else if (nm % 8 == 0) walk(integral_constant<int, 8>{}, std::false_type{});
            else if (nm == 4) walk(integral_constant<int, 4>{}, std::false_type{});
            else if (nm == 2) walk(integral_constant<int, 2>{}, std::false_type{});
            else if (nm == 1) walk(integral_constant<int, 1>{}, std::false_type{});
            else walk(integral_constant<int, 6>{}, std::true_type{});
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float sl = slo[u], sh = shi[u];
#pragma unroll
                for (int off = 4; off > 0; off >>= 1) { sl = sl + __shfl_xor(sl, off); sh = sh + __shfl_xor(sh, off); }
                const float sv = sl + sh;
                if (fr[u] && a8 == 0) nb_d[r0 + (uint32_t)(u * WPQ + wave) * G + g8] = metric == ZVDB_METRIC_L2SQ ? sv : -sv;
            }
        }
    } else {
        constexpr int G = 4;
        const int g = lane >> 4, t = lane & 15;
        for (uint32_t r0 = 0; r0 < cnt; r0 += G * U * WPQ) {
            if (r0 + (uint32_t)wave * G >= cnt) break;
            const RT *rp[U];
            bool fr[U];
            float sacc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t j = r0 + (uint32_t)(u * WPQ + wave) * G + g;
                fr[u] = j < cnt;
                rp[u] = rows + (size_t)(fr[u] ? nb_i[j] : 0u) * ld;
                sacc[u] = 0.0f;
            }
            // super-steps of 64 * SS chunks (a 768-d row is ONE super-step at SS = 3; measured with U = 4, one wave per query, 1M x 768,
            // ef = 128, batch 1,024 / 4,096 / 16,384: SS = 3 637k / 741k / 807k QPS, SS = 1 654k / 660k / 729k); a group without a row re-reads row 0
            constexpr int SS = 3;
            const uint32_t lastc = ld / 4 - 1;
            for (uint32_t c0 = 0; c0 < nchunks; c0 += 64 * SS) {
                float4 vv[4 * SS][U];
#pragma unroll
                for (int i = 0; i < 4 * SS; ++i) {
                    const uint32_t ce = min(c0 + t + 16 * i, lastc);
#pragma unroll
                    for (int u = 0; u < U; ++u) vv[i][u] = row_load4(rp[u] + 4 * ce);
                }
#pragma unroll
                for (int i = 0; i < 4 * SS; ++i) {
                    const uint32_t c = c0 + t + 16 * i;
                    const bool okc = c < nchunks;
                    const float4 a = *reinterpret_cast<const float4 *>(q + 4 * min(c, nchunks - 1));
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float4 v = vv[i][u];
                        float w0, w1, w2, w3;
                        if (metric == ZVDB_METRIC_L2SQ) {
                            w0 = a.x - v.x; w0 = w0 * w0; w1 = a.y - v.y; w1 = w1 * w1;
                            w2 = a.z - v.z; w2 = w2 * w2; w3 = a.w - v.w; w3 = w3 * w3;
                        } else {
                            w0 = a.x * v.x; w1 = a.y * v.y; w2 = a.z * v.z; w3 = a.w * v.w;
                        }
                        if (!okc) { w0 = 0.0f; w1 = 0.0f; w2 = 0.0f; w3 = 0.0f; }  // +0 leaves the sum unchanged
                        sacc[u] = sacc[u] + w0; sacc[u] = sacc[u] + w1;
                        sacc[u] = sacc[u] + w2; sacc[u] = sacc[u] + w3;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float sv = sacc[u];
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) sv = sv + __shfl_xor(sv, off);
                if (fr[u] && t == 0) nb_d[r0 + (uint32_t)(u * WPQ + wave) * G + g] = metric == ZVDB_METRIC_L2SQ ? sv : -sv;
            }
        }
    }
}

// ---- ef-mode beam search (extension; specification: oracle/zvdb_oracle.h) --------
// W = sorted (dist,id) list of <= ef entries with an "expanded" flag, LDS resident.
//
// Traversal distances use the oracle's 16-way BLOCKED order (orc_distance_blocked16_f32): a 16-lane
// group of the wave walks one neighbour row with coalesced 16-byte loads (lane t takes the 4-float
// chunks t, t+16, ...; each lane accumulates its chunks sequentially, un-contracted), then the 16
// partials are combined by the xor butterfly 8,4,2,1.  Four neighbour rows are in flight per wave
// step instead of one row per lane walking 3 KB alone: same bytes, coalesced, 4x fewer dependent
// loads per lane.  The k results are re-evaluated with the exact reference arithmetic (a1).
// Beam W: UNSORTED set of <= ef 64-bit keys ((monotone fp32 key << 32) | id) + an "expanded" flag per slot,
// double buffered in LDS.  The oracle's sorted-list formulation is equivalent to set semantics:
//   next node to expand = minimum key among unexpanded entries          (one wave min-reduction)
//   after an expansion   W = best ef of (W u fresh candidates)           (one radix select per expansion,
//                                                                         only when a candidate beats the worst)
// so the per-candidate sorted insertions (two barriers per 64 slots shifted) of the first version are gone.
// PROF = 1 (ZVDB_EF_PROF=1, diagnostic): per-phase shader-cycle sums of wave 0's lane 0 go to g_ef_prof (results unchanged)
// Visited-set sizing in two passes.  The LDS hash of a query sets the waves per CU, and a size that is SAFE (1.5 ef stride
// entries) is 3 - 5 x what searches really touch (0.3 - 1.1 ef degree, data dependent): at ef 256 / stride 32 the safe table
// is 128 KB -- one wave per CU.  So the index remembers the largest fill it has seen per ef bucket (fill_hint, copied to
// the host asynchronously after every launch); once known, the batch first runs with a table of 2.5 x that (kEfTight: a
// query that fills it stops, marks its count kEfAgain and writes nothing else) and a second launch with the safe table
// serves exactly the marked queries (kEfRetry).  Results and counters are those of a single safe pass.
// pass: kEfSingle all queries, a full table is an error (stats[2]); kEfTight all queries, a query that fills its table marks
// itself (count = kEfAgain) for the next launch; kEfRetry only the marked, full table = error; kEfRetryMark only the marked, marks again.
constexpr uint32_t kEfSingle = 0, kEfTight = 1, kEfRetry = 2, kEfRetryMark = 3, kEfAgain = 0xFFFFFFFFu;
__device__ unsigned long long g_ef_prof[10];
// GH: the visited set lives in GLOBAL memory (ghash: one pre-zeroed table of 2^hash_bits slots per query of the launch, query
// index = block + qoff) -- the safe pass of searches whose safe table does not fit the LDS (ef * stride beyond ~12k)
// The net's work list (index-owned, zeroed by the host before the last LDS pass of a call): [0] = queries appended, [1] = pop
// cursor of the net launch, [2 ...] = query indices.  The LAST LDS pass of a call appends every query it marks; the net launch is
// kEfNetBlocks workgroups that pop queries until the list is empty, each on its own global visited table.
constexpr uint32_t kEfNetBlocks = 16;
__device__ __forceinline__ void ef_mark_again(uint32_t *out_count, uint32_t qi, uint32_t *net_list) {  // (one lane)
    out_count[qi] = kEfAgain;
    if (net_list) net_list[2 + atomicAdd(net_list, 1u)] = qi;
}

// one query of the LDS-beam search; hash_g: the query's visited table when GH (pre-zeroed), unused otherwise
template <typename RT, int PROF, bool GH>
__device__ __forceinline__ void ef_search_one(
    const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
    const uint32_t *__restrict__ adj, const uint32_t *__restrict__ deg, uint32_t gstride,
    uint32_t N, const float *__restrict__ queries, uint32_t k, uint32_t ef, uint32_t entry,
    uint64_t id_offset, uint64_t *__restrict__ out_ids, float *__restrict__ out_dist,
    uint32_t *__restrict__ out_count, unsigned long long *__restrict__ stats, uint32_t dim_pad,
    uint32_t hash_bits, uint32_t efp, const uint32_t *__restrict__ entries, uint32_t pass, uint32_t *__restrict__ fill_hint,
    uint32_t *hash_g, uint32_t qi, uint32_t *net_list, unsigned char *smem, uint32_t width) {
    if ((pass == kEfRetry || pass == kEfRetryMark) && out_count[qi] != kEfAgain) return;  // served by an earlier pass
    if (entries) entry = entries[qi];
    float *q = reinterpret_cast<float *>(smem);
    uint64_t *wk = reinterpret_cast<uint64_t *>(q + dim_pad);          // [2][efp]
    uint32_t *wx = reinterpret_cast<uint32_t *>(wk + 2 * efp);         // [2][efp] expanded flags
    uint32_t *hist = wx + 2 * efp;                                     // [256] select scratch
    uint32_t *lds_hash = hist + 256;
    float *nb_d = reinterpret_cast<float *>(lds_hash + (GH ? 0u : (1u << hash_bits)));
    uint32_t *nb_i = reinterpret_cast<uint32_t *>(nb_d + kWave);
    const int lane = threadIdx.x;
    const int t = lane & 15;
    uint32_t *hash;  // (one address space per instantiation: LDS atomics or global atomics)
    if constexpr (GH) hash = hash_g;
    else hash = lds_hash;
    const float *qg = queries + (size_t)qi * dim;
    for (uint32_t i = lane; i < dim_pad; i += kWave) q[i] = i < dim ? qg[i] : 0.0f;
    if constexpr (!GH)
        for (uint32_t i = lane; i < (1u << hash_bits); i += kWave) hash[i] = 0;
    __syncthreads();
    const uint32_t nchunks = (dim + 3) / 4;

    uint32_t wn = 0, cur = 0;
    uint64_t worst = ~0ull;  // max key of W, valid when wn == ef
    unsigned long long evals = 0, pops = 0;
    unsigned long long pf[6] = {0, 0, 0, 0, 0, 0}, pt = 0;
    auto stamp = [&](int ph) { if constexpr (PROF) { const unsigned long long n = __builtin_amdgcn_s_memtime(); pf[ph] += n - pt; pt = n; } };
    if constexpr (PROF) pt = __builtin_amdgcn_s_memtime();
    const uint32_t hash_limit = (1u << hash_bits) - (1u << (hash_bits - 2));  // 75 % load cap
    uint32_t hash_used = 0;
    bool overflow = false;
    if (N > 0 && k > 0) {
        const RT *erow = rows + (size_t)entry * ld;
        const float d0 = metric == ZVDB_METRIC_L2SQ ? blocked16_dist<ZVDB_METRIC_L2SQ>(q, erow, nchunks, t)
                                                    : blocked16_dist<ZVDB_METRIC_DOT>(q, erow, nchunks, t);
        if (lane == 0) { wk[0] = ((uint64_t)f32_key(d0) << 32) | entry; wx[0] = 0; }
        wn = 1;
        if (ef == 1) worst = ((uint64_t)f32_key(d0) << 32) | entry;  // the entry alone fills the beam (found by the two-kernel fuzz: the first
                                                                      // candidate of an ef = 1 search was compared against an unset worst and lost)
        evals = 1;
        hash_insert_uniform(hash, hash_bits, entry);
        hash_used = 1;
        __syncthreads();
        for (;;) {
            uint64_t *W = wk + cur * efp;
            uint32_t *X = wx + cur * efp;
            // next to expand: the smallest unexpanded key -- with a search width (orc_hnsw_search_ef_w) the `width` smallest, all marked
            // expanded before any list is read; their lists are then walked one after the other (the spec's own order)
            uint32_t cids[4] = {0, 0, 0, 0}, nsel = 0;
            for (uint32_t w = 0; w < width && w < 4; ++w) {
                uint64_t lm = ~0ull;
                uint32_t li = 0;  // slot of this lane's smallest unexpanded key
                for (uint32_t i = lane; i < wn; i += kWave)
                    if (X[i] == 0 && W[i] < lm) { lm = W[i]; li = i; }
                const uint64_t kbest = wave_min64(lm);
                if (kbest == ~0ull) break;
                if (lm == kbest) X[li] = 1;  // (keys are distinct: one lane)
                cids[w] = (uint32_t)kbest;
                nsel = w + 1;
                __syncthreads();
            }
            if (nsel == 0) break;
            pops += nsel;
            stamp(0);  // next-to-expand scan
            for (uint32_t sidx = 0; sidx < nsel; ++sidx) {
            const uint32_t cid = cids[sidx];
            // the degree and the first 64 list slots are requested TOGETHER (both depend on cid only): one memory round trip per
            // pop instead of two (slots past the degree hold stale or zero ids and are masked below)
            const uint32_t nb_first = (uint32_t)lane < gstride ? adj[(size_t)cid * gstride + lane] : 0u;
            const uint32_t dg_all = deg[cid];  // lists longer than 64 are walked 64 entries at a time
            for (uint32_t lbase = 0; lbase < dg_all; lbase += kWave) {
                const uint32_t cnt = min(dg_all - lbase, (uint32_t)kWave);
                __syncthreads();
                if (hash_used + cnt >= hash_limit) overflow = true;  // wave-uniform
                if (overflow && (pass == kEfTight || pass == kEfRetryMark)) {  // the launch behind this one serves the query
                    if (lane == 0) ef_mark_again(out_count, qi, net_list);
                    return;
                }
                // (1) one lane per neighbour: visited test-and-set; fb = fresh list positions
                uint32_t nbl = 0;
                bool fresh1 = false;
                if ((uint32_t)lane < cnt && !overflow) {
                    nbl = lbase == 0 ? nb_first : adj[(size_t)cid * gstride + lbase + lane];
                    fresh1 = hash_insert_atomic(hash, hash_bits, nbl);
                }
                // the fresh neighbours move to the front of nb_i (list order kept): the distance passes below take 16 rows each,
                // and ~15 of a 32-entry list are fresh in the steady state -- one pass instead of two for most pops
                const unsigned long long fb0 = __ballot(fresh1);
                if (fresh1) nb_i[__popcll(fb0 & ((1ull << lane) - 1ull))] = nbl;
                const uint32_t nf = (uint32_t)__popcll(fb0);
                const unsigned long long fb = nf == kWave ? ~0ull : (1ull << nf) - 1ull;
                __syncthreads();
                stamp(1);  // degree + list load + visited test-and-set
                if (nf == 0) continue;
                ef_window_distances<RT, 1, sizeof(RT) == 2 ? 2 : 4>(rows, ld, q, nchunks, metric, nf, nb_i, nb_d, lane, 0);
                hash_used += (uint32_t)__popcll(fb);
                evals += (unsigned long long)__popcll(fb);
                __syncthreads();
                stamp(2);  // blocked distances of the fresh neighbours
                // (3) W = best ef of (W u fresh candidates): lane j holds candidate j
                uint64_t ck = ~0ull;
                if ((fb >> lane) & 1ull) ck = ((uint64_t)f32_key(nb_d[lane]) << 32) | nb_i[lane];
                const bool pass = ck != ~0ull && (wn < ef || ck < worst);
                const unsigned long long pm = __ballot(pass);
                const uint32_t na = (uint32_t)__popcll(pm);
                if (na == 0) { stamp(3); continue; }
                W = wk + cur * efp;
                X = wx + cur * efp;
                if (wn + na <= ef) {  // room: append
                    const uint32_t pos = wn + (uint32_t)__popcll(pm & ((1ull << lane) - 1));
                    if (pass) { W[pos] = ck; X[pos] = 0; }
                    wn += na;
                    __syncthreads();
                    if (wn == ef) {
                        uint64_t mx = 0;
                        for (uint32_t i = lane; i < wn; i += kWave) mx = umax64(mx, W[i]);
                        worst = wave_max64(mx);
                    }
                } else if (na <= 8) {
                    // few candidates (the steady state of a search: W is full and most neighbours lose against its worst):
                    // the oracle's own sequential rule, candidate by candidate in list order -- while there is room append,
                    // else replace the worst entry if the candidate beats it and find the new worst.  Same set as the select
                    // below at ~400 cycles per accepted candidate instead of one 5,000-cycle radix select per pop.
                    unsigned long long todo = pm;
                    while (todo) {
                        const int l = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        const uint64_t ckl = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ck >> 32), l) << 32) |
                                             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ck, l);
                        bool wrote = false;
                        if (wn < ef) {
                            if (lane == 0) { W[wn] = ckl; X[wn] = 0; }
                            wn += 1;
                            wrote = true;
                        } else if (ckl < worst) {
                            for (uint32_t i = lane; i < wn; i += kWave)
                                if (W[i] == worst) { W[i] = ckl; X[i] = 0; }  // (keys are distinct: exactly one slot)
                            wrote = true;
                        }
                        if (wrote && wn == ef) {
                            __syncthreads();
                            uint64_t mx = 0;
                            for (uint32_t i = lane; i < wn; i += kWave) mx = umax64(mx, W[i]);
                            worst = wave_max64(mx);
                        }
                    }
                    __syncthreads();
                } else {  // select the ef best of the union, compact into the other buffer
                    const uint32_t Ew = (wn + kWave - 1) / kWave;
                    const uint64_t ckp = pass ? ck : ~0ull;
                    const uint64_t T = wave_kth_key<0>(
                        [&](uint32_t e) -> uint64_t {
                            if (e == Ew) return ckp;
                            const uint32_t i = e * kWave + lane;
                            return i < wn ? W[i] : ~0ull;
                        },
                        Ew + 1, ef, hist, lane);
                    uint64_t *W2 = wk + (cur ^ 1) * efp;
                    uint32_t *X2 = wx + (cur ^ 1) * efp;
                    uint32_t base = 0;
                    for (uint32_t e = 0; e <= Ew; ++e) {
                        const uint32_t i = e * kWave + lane;
                        const uint64_t kk = e == Ew ? ckp : (i < wn ? W[i] : ~0ull);
                        const uint32_t xf = (e == Ew || i >= wn) ? 0u : X[i];
                        const bool keep = kk <= T;
                        const unsigned long long m = __ballot(keep);
                        const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
                        if (keep) { W2[pos] = kk; X2[pos] = xf; }
                        base += (uint32_t)__popcll(m);
                    }
                    wn = ef;
                    cur ^= 1;
                    worst = T;
                    __syncthreads();
                }
                stamp(3);  // beam update (append / radix select)
            }
            }
        }
        __syncthreads();
    }
    stamp(4);
    // the k smallest keys of W, re-evaluated with the exact reference arithmetic, ordered by (exact distance, id)
    const uint32_t nres = wn < k ? wn : k;
    uint64_t *W = wk + cur * efp;
    uint64_t *R = wk + (cur ^ 1) * efp;  // result ids (low 32 bits) gathered here
    float *RD = reinterpret_cast<float *>(wx);
    if (nres > 0) {
        uint64_t T = ~0ull - 1;
        if (wn > nres) {
            const uint32_t Ew = (wn + kWave - 1) / kWave;
            T = wave_kth_key<0>([&](uint32_t e) -> uint64_t { const uint32_t i = e * kWave + lane; return i < wn ? W[i] : ~0ull; },
                                Ew, nres, hist, lane);
        }
        uint32_t base = 0;
        for (uint32_t e = 0; e * kWave < wn; ++e) {
            const uint32_t i = e * kWave + lane;
            const uint64_t kk = i < wn ? W[i] : ~0ull;
            const bool keep = kk <= T;
            const unsigned long long m = __ballot(keep);
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
            if (keep) R[pos] = kk;
            base += (uint32_t)__popcll(m);
        }
        __syncthreads();
        for (uint32_t i = lane; i < nres; i += kWave)
            RD[i] = exact_dist_rt(metric, q, rows + (size_t)(uint32_t)R[i] * ld, dim);
        __syncthreads();
        for (uint32_t i = lane; i < nres; i += kWave) {
            const float di = RD[i];
            const uint32_t ii = (uint32_t)R[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < nres; ++j)
                if (lex_less(RD[j], (uint32_t)R[j], di, ii)) ++rank;
            out_ids[(size_t)qi * k + rank] = (uint64_t)ii + id_offset;
            out_dist[(size_t)qi * k + rank] = di;
        }
    }
    for (uint32_t i = nres + lane; i < k; i += kWave) {
        out_ids[(size_t)qi * k + i] = ~0ull;
        out_dist[(size_t)qi * k + i] = __builtin_inff();
    }
    if (lane == 0) {
        out_count[qi] = nres;
        if (fill_hint) atomicMax(fill_hint, hash_used);
        if (stats) {
            atomicAdd(&stats[0], evals);
            atomicAdd(&stats[1], pops);
            if (overflow) atomicAdd(&stats[2], 1ull);  // visited set exhausted: result not to spec
        }
        if constexpr (PROF) {
            stamp(5);  // final k: exact re-evaluation + order
            for (int i = 0; i < 6; ++i) atomicAdd(&g_ef_prof[i], pf[i]);
            atomicAdd(&g_ef_prof[6], pops);
            atomicAdd(&g_ef_prof[7], 1ull);
        }
    }
}

// launch forms of ef_search_one.  LDS tables (GH = false): block = query.  GH, net_list == nullptr: the global safe pass, block =
// query (+ qoff), table = block (pre-zeroed by the host).  GH, net_list != nullptr: the NET -- each of the kEfNetBlocks workgroups
// owns one table and serves queries popped from the work list (clearing the table per query) until the list is empty.
template <typename RT, int PROF = 0, bool GH = false>
__global__ __launch_bounds__(64) void k_search_ef(
    const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
    const uint32_t *__restrict__ adj, const uint32_t *__restrict__ deg, uint32_t gstride,
    uint32_t N, const float *__restrict__ queries, uint32_t k, uint32_t ef, uint32_t entry,
    uint64_t id_offset, uint64_t *__restrict__ out_ids, float *__restrict__ out_dist,
    uint32_t *__restrict__ out_count, unsigned long long *__restrict__ stats, uint32_t dim_pad,
    uint32_t hash_bits, uint32_t efp, const uint32_t *__restrict__ entries /* optional: per-query entry (k_descend) */,
    uint32_t pass, uint32_t *__restrict__ fill_hint, uint32_t *ghash, uint32_t qoff,
    uint32_t *net_list /* LDS passes: marks are appended here (may be null); GH: non-null = the net launch */, uint32_t width /* nodes expanded per step */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (GH) {
        uint32_t *table = ghash + ((size_t)blockIdx.x << hash_bits);
        if (net_list) {
            const uint32_t total = __hip_atomic_load(net_list, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                uint32_t i = 0;
                if (threadIdx.x == 0) i = atomicAdd(net_list + 1, 1u);
                i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
                if (i >= total) return;
                const uint32_t qi = net_list[2 + i];
                for (uint32_t j = threadIdx.x; j < (1u << hash_bits); j += kWave) table[j] = 0;
                __threadfence();
                __syncthreads();
                ef_search_one<RT, PROF, true>(rows, ld, dim, metric, adj, deg, gstride, N, queries, k, ef, entry, id_offset, out_ids, out_dist, out_count,
                                              stats, dim_pad, hash_bits, efp, entries, pass, fill_hint, table, qi, nullptr, smem, width);
                __syncthreads();
            }
        }
        ef_search_one<RT, PROF, true>(rows, ld, dim, metric, adj, deg, gstride, N, queries, k, ef, entry, id_offset, out_ids, out_dist, out_count, stats,
                                      dim_pad, hash_bits, efp, entries, pass, fill_hint, table, blockIdx.x + qoff, nullptr, smem, width);
    } else {
        ef_search_one<RT, PROF, false>(rows, ld, dim, metric, adj, deg, gstride, N, queries, k, ef, entry, id_offset, out_ids, out_dist, out_count, stats,
                                       dim_pad, hash_bits, efp, entries, pass, fill_hint, nullptr, blockIdx.x + qoff, net_list, smem, width);
    }
}

// ---- ef-mode beam search, beam in REGISTERS (ef <= 64 EPL) ---------------------------------------------------------------------
// Same search, same results as k_search_ef above (keys are distinct, so "the ef best of the union" is one set however it is
// held); what changes is where W lives.  The profile of the LDS-set kernel (ZVDB_EF_PROF=1, 1M x 768, ef = 128) put a third
// of every pop into W bookkeeping: ~1,400 cycles to find the next node (LDS scan + a 64-bit wave reduction) and ~3,400 to
// merge the candidates (one reduction per accepted candidate to find the new worst, or a radix select).  Here W is the
// oracle's own formulation again -- a list SORTED by key -- spread over the wave's registers: slot s sits in lane s % 64 of
// wkr[s / 64], the "expanded" flags are one 64-bit scalar mask per block.  Then
//   next node to expand  = first zero bit of the flag masks                       (scalar ff0 + two readlanes)
//   worst entry          = slot ef - 1                                            (two readlanes)
//   insert a candidate   = position by ballot/popcount of (key < candidate), then every block from there on moves up one
//                          lane with a wave_shr:1 DPP move (the block's last entry carries into lane 0 of the next), the
//                          flag masks shift alongside on the scalar unit; the entry pushed past slot ef - 1 is the old worst
// -- no LDS traffic, no barrier, no reduction: ~50 instructions per accepted candidate.  The first k slots are the result.
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}
__device__ __forceinline__ uint64_t wave_shr1_64(uint64_t v) {  // lane i <- lane i - 1 (lane 0: unspecified, overwritten by the caller)
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}

// WPQ waves per query (a workgroup of 64 WPQ threads): wave 0 runs the search exactly as described; the distance phase of a
// window -- 60 - 70 % of a pop -- is shared by all WPQ waves (each takes its rows of the step: ef_window_distances), two
// workgroup barriers per window (window published in ctl[0] / distances done).  Same results, same counters; what it buys is
// latency: a 1,024-query batch is one round of resident workgroups either way, its time is the slowest query's chain of pops.
constexpr uint32_t kEfDone = 0xFFFFFFFFu;  // ctl[0]: no further window (a window has 1 .. 64 rows)
template <typename RT, int EPL, int WPQ = 1, int PROF = 0, bool WIDE = false>
__global__ __launch_bounds__(64 * WPQ, WPQ == 8 ? 1 : WPQ == 4 ? (EPL >= 8 ? 2 : 4) : WPQ == 2 ? 2 : 1) void k_search_ef_reg(
    const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
    const uint32_t *__restrict__ adj, const uint32_t *__restrict__ deg, uint32_t gstride,
    uint32_t N, const float *__restrict__ queries, uint32_t k, uint32_t ef, uint32_t entry,
    uint64_t id_offset, uint64_t *__restrict__ out_ids, float *__restrict__ out_dist,
    uint32_t *__restrict__ out_count, unsigned long long *__restrict__ stats, uint32_t dim_pad,
    uint32_t hash_size /* slots of the visited table: any count >= 256 */, uint32_t kpad, const uint32_t *__restrict__ entries /* optional: per-query entry (k_descend) */,
    uint32_t pass, uint32_t *__restrict__ fill_hint, uint32_t * /* ghash: k_search_ef only */, uint32_t /* qoff */,
    uint32_t *net_list /* marks are appended here (the last LDS pass of a call; else null) */,
    uint32_t width_mode /* nodes expanded per step: 1 .. 4; large beams (EPL >= 8) also: r << 8 | b << 16 | packed visited table << 24 (hp_insert_atomic) */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t width = EPL >= 8 ? (width_mode & 0xFFu) : width_mode;
    const bool hp_on = EPL >= 8 && ((width_mode >> 24) & 1u) != 0;
    const uint32_t hp_r = (width_mode >> 8) & 0xFFu, hp_b = (width_mode >> 16) & 0xFFu;
    const uint32_t hp_bmask = hp_b >= 32 ? ~0u : (1u << hp_b) - 1u, hp_half = hp_b / 2 > 1 ? hp_b / 2 : 1u;
    const uint32_t hp_cw = (hash_size / 2) >> hp_r;  // 64-bit words per region
    if ((pass == kEfRetry || pass == kEfRetryMark) && out_count[blockIdx.x] != kEfAgain) return;  // served by an earlier pass
    if (entries) entry = entries[blockIdx.x];
    float *q = reinterpret_cast<float *>(smem);
    uint64_t *Wt = reinterpret_cast<uint64_t *>(q + dim_pad);    // [64 EPL] merge scatter buffer: keys ...
    uint32_t *Ft = reinterpret_cast<uint32_t *>(Wt + kWave * EPL);  // [64 EPL] ... and expanded flags
    uint32_t *hash = Ft + kWave * EPL;
    float *nb_d = reinterpret_cast<float *>(hash + hash_size);
    uint32_t *nb_i = reinterpret_cast<uint32_t *>(nb_d + kWave);
    uint32_t *R = nb_i + kWave;                          // [kpad] result ids
    float *RD = reinterpret_cast<float *>(R + kpad);     // [kpad] their exact distances
    uint32_t *ctl = reinterpret_cast<uint32_t *>(RD + kpad);  // [4] window hand-over to the helper waves (WPQ > 1)
    constexpr bool LMERGE = EPL >= 8;  // (at 4 blocks -- ef = 256 -- the gather merge measured the same as the merge by ranks: A/B on one box) large beams: Wt / Ft always hold the beam and a window merges by gather (merge_lds below)
    uint64_t *Cs = reinterpret_cast<uint64_t *>(ctl + 4);  // LMERGE: [64] the window's candidates in key order ...
    uint32_t *Mb = reinterpret_cast<uint32_t *>(Cs + kWave);  // ... and [2 EPL] one bit per beam slot a candidate lands in

    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int t = lane & 15;
    const uint32_t qi = blockIdx.x;
    const float *qg = queries + (size_t)qi * dim;
    for (uint32_t i = threadIdx.x; i < dim_pad; i += kWave * WPQ) q[i] = i < dim ? qg[i] : 0.0f;
    for (uint32_t i = threadIdx.x; i < hash_size; i += kWave * WPQ) hash[i] = 0;
    __syncthreads();
    const uint32_t nchunks = (dim + 3) / 4;
    constexpr int U = sizeof(RT) == 2 ? (WPQ == 1 ? 2 : 1) : (WPQ >= 4 ? 1 : 4 / WPQ);  // rows per lane group and step: 16 rows per step (bf16 rows, WPQ = 4: 32; two rows per group at WPQ = 2 measured 24 % SLOWER with a search width, r4)
    if constexpr (WPQ > 1) {
        if (wave != 0) {  // helper waves: their share of every published window's distances
            for (;;) {
                __syncthreads();  // (A) window published
                const uint32_t nfh = ctl[0];
                if (nfh == kEfDone) return;
                ef_window_distances<RT, WPQ, U>(rows, ld, q, nchunks, metric, nfh, nb_i, nb_d, lane, wave);
                __syncthreads();  // (B) distances done
            }
        }
    }
    // wave 0 from here on; sync_w orders its own LDS traffic (a workgroup barrier when it is the only wave)
    auto sync_w = [&]() {
        if constexpr (WPQ == 1) __syncthreads();
        else __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto release_helpers = [&]() {
        if constexpr (WPQ > 1) {
            if (lane == 0) ctl[0] = kEfDone;
            __syncthreads();  // (A)
        }
    };

    uint64_t wkr[EPL];          // sorted beam: slot s = lane s % 64 of wkr[s / 64]; slots >= wn are meaningless
    uint64_t xm[EPL];           // expanded flags (wave-uniform)
#pragma unroll
    for (int e = 0; e < EPL; ++e) { wkr[e] = ~0ull; xm[e] = 0; }
    uint32_t wn = 0;
    uint64_t worst = ~0ull;     // key of slot ef - 1, valid when wn == ef
    unsigned long long evals = 0, pops = 0;
    unsigned long long pf[6] = {0, 0, 0, 0, 0, 0}, pt = 0;
    auto stamp = [&](int ph) { if constexpr (PROF) { const unsigned long long n = __builtin_amdgcn_s_memtime(); pf[ph] += n - pt; pt = n; } };
    if constexpr (PROF) pt = __builtin_amdgcn_s_memtime();
    uint32_t hash_limit = hash_size - hash_size / 8;  // 87.5 % load cap (the tight table is sized for <= 74 % at the largest fill seen)
    if constexpr (EPL >= 8) {
        if (hp_on) { const uint32_t cap = (3u * hp_cw) << hp_r; hash_limit = cap - cap / 8; }
    }
    uint32_t hash_used = 0;
    bool overflow = false;
    // visited test-and-set of one lane's neighbour: true = fresh.  Large beams on the packed table: `full` = its region is full
    auto visit = [&](uint32_t id, bool &full) -> bool {
        if constexpr (EPL >= 8) {
            if (hp_on) {
                const uint32_t v = hp_insert_atomic(reinterpret_cast<unsigned long long *>(hash), hp_cw, hp_r, hp_bmask, hp_half, id);
                full = v == 2;
                return v == 1;
            }
        }
        return hashn_insert_atomic(hash, hash_size, id);
    };

    uint32_t pre_id = 0xFFFFFFFFu, pre_list = 0;  // speculative list prefetch (ids are < N <= 2^32 - 1)
    unsigned long long pre_hits = 0, list_wait = 0;
    auto insert_sorted = [&](uint64_t ckl) {  // ckl wave-uniform, distinct from every key held; if the beam is full: ckl < worst
        uint32_t p = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) p += (uint32_t)__popcll(__ballot((uint32_t)(e * kWave + lane) < wn && wkr[e] < ckl));
        const uint32_t ep = p >> 6, lp = p & 63;
        uint64_t carry = 0, carrybit = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            if ((uint32_t)e >= ep) {  // (wave-uniform)
                const bool first = (uint32_t)e == ep;
                const uint64_t last = readlane64(wkr[e], kWave - 1);
                const uint64_t sh = wave_shr1_64(wkr[e]);
                const uint32_t thr = first ? lp : 0u;
                const uint64_t in0 = first ? ckl : carry;
                wkr[e] = (uint32_t)lane < thr ? wkr[e] : ((uint32_t)lane == thr ? in0 : sh);
                const uint64_t lowmask = first ? ((1ull << lp) - 1ull) : 0ull;
                const uint64_t top = xm[e] >> 63;
                xm[e] = (xm[e] & lowmask) | ((xm[e] & ~lowmask) << 1) | (first ? 0ull : carrybit);
                carry = last;
                carrybit = top;
            }
        }
        if (wn < ef) wn += 1;
        if (wn == ef) {
#pragma unroll
            for (int e = 0; e < EPL; ++e)
                if ((ef - 1) >> 6 == (uint32_t)e) worst = readlane64(wkr[e], (int)((ef - 1) & 63));
        }
    };

    // Large beams (EPL >= 8: ef > 256).  The merge by ranks further down costs ~6 instructions per (candidate, block) pair -- at 16
    // blocks and the 20-40 candidates a search-width window accepts, more than the window's row fetches (ZVDB_EF_PROF: the beam
    // update is a sixth of a pop at ef = 128, a third at ef = 512).  Here the LDS copy of the beam is kept current (Wt: keys in slot
    // order, Ft: expanded flags, also set when a slot is picked) and a window's candidates merge in O(blocks + candidates):
    //   candidate lane: #beam entries below it by BINARY SEARCH in Wt (<= 11 dependent LDS reads, all candidates at once), + its
    //   rank among the candidates = its slot in the merged beam; it sets that slot's bit in Mb and leaves its key in Cs[rank];
    //   every beam slot p then GATHERS: with c = #candidate bits below p (a popcount prefix), p holds candidate Cs[c] if its own
    //   bit is set, else the old entry p - c -- one LDS read per slot; the registers take the result, then refresh Wt / Ft.
    // Same beam as the sorted insertion of each candidate in turn (keys are distinct), so results and counters do not change.
    auto merge_lds = [&](uint64_t ck, bool cpass, unsigned long long todo, uint32_t na) {
        if ((uint32_t)lane < 2u * EPL) Mb[lane] = 0;
        uint32_t lo = 0, hi = cpass ? wn : 0u;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (Wt[mid] < ck) lo = mid + 1; else hi = mid;
        }
        uint32_t crank = 0;
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            crank += readlane64(ck, l) < ck ? 1u : 0u;
        }
        sync_w();  // (Mb cleared before the bits go in)
        if (cpass) {
            const uint32_t pos = lo + crank;
            Cs[crank] = ck;
            if (pos < ef) atomicOr(&Mb[pos >> 5], 1u << (pos & 31));
        }
        const uint32_t wn_new = min(wn + na, ef);
        sync_w();
        uint32_t cbase = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const uint64_t mw = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)Mb[2 * e + 1]) << 32) |
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)Mb[2 * e]);
            const uint32_t p = (uint32_t)(e * kWave + lane);
            const uint32_t c = cbase + (uint32_t)__popcll(mw & ((1ull << lane) - 1ull));
            const bool isc = (mw >> lane) & 1ull;
            const bool live = p < wn_new;
            uint32_t f = 0;
            if (live) {
                wkr[e] = isc ? Cs[c] : Wt[p - c];
                f = isc ? 0u : Ft[p - c];
            }
            xm[e] = __ballot(live && f != 0);
            cbase += (uint32_t)__popcll(mw);
        }
        wn = wn_new;
        sync_w();  // (every gather done before the slots are overwritten)
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const uint32_t p = (uint32_t)(e * kWave + lane);
            if (p < wn) { Wt[p] = wkr[e]; Ft[p] = (uint32_t)(xm[e] >> lane) & 1u; }
        }
        if (wn == ef) {
#pragma unroll
            for (int e = 0; e < EPL; ++e)
                if ((ef - 1) >> 6 == (uint32_t)e) worst = readlane64(wkr[e], (int)((ef - 1) & 63));
        }
    };

    if (N > 0 && k > 0) {
        const RT *erow = rows + (size_t)entry * ld;
        const float d0 = metric == ZVDB_METRIC_L2SQ ? blocked16_dist<ZVDB_METRIC_L2SQ>(q, erow, nchunks, t)
                                                    : blocked16_dist<ZVDB_METRIC_DOT>(q, erow, nchunks, t);
        insert_sorted(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)f32_key(d0)) << 32) | entry);
        if constexpr (LMERGE) {
            if (lane == 0) { Wt[0] = wkr[0]; Ft[0] = 0; }
        }
        evals = 1;
        if constexpr (EPL >= 8) {
            bool full0 = false;
            (void)visit(entry, full0);  // (every lane the same id: one of them puts it in)
        } else {
            hashn_insert_uniform(hash, hash_size, entry);
        }
        hash_used = 1;
        sync_w();
        if constexpr (!WIDE) {  // width 1: the loop of rounds 2-3, untouched -- every restructuring that also served a search width
                                // (a shared lambda with two call sites: +17 % time on bf16 rows; a window state machine: +21 %) cost it
        for (;;) {
                // next to expand: the first unexpanded slot
                uint32_t slot = kWave * EPL;
    #pragma unroll
                for (int e = EPL - 1; e >= 0; --e)
                    if (~xm[e] != 0ull) slot = (uint32_t)e * kWave + (uint32_t)__builtin_ctzll(~xm[e]);
                if (slot >= wn) break;
                uint64_t kbest = 0;
    #pragma unroll
                for (int e = 0; e < EPL; ++e)
                    if (slot >> 6 == (uint32_t)e) { kbest = readlane64(wkr[e], (int)(slot & 63)); xm[e] |= 1ull << (slot & 63); }
                if constexpr (LMERGE) {
                    if (lane == 0) Ft[slot] = 1;
                }
                const uint32_t cid = (uint32_t)kbest;
                pops += 1;
                stamp(0);  // next-to-expand
                // one vector load brings the first list slots AND the degree (lane 63 reads deg[cid] when the stride leaves it free) --
                // or brought them already: the list of the NEXT unexpanded slot is requested speculatively a whole distance phase
                // ahead (below); it is this pop's node unless a candidate of the previous pop went in before it
                uint32_t nb_first, dg_all;  // lists longer than 64 are walked 64 entries at a time
                if (cid == pre_id) {
                    nb_first = pre_list;
                    if constexpr (PROF) pre_hits += 1;
                } else {
                    nb_first = *((uint32_t)lane < gstride ? adj + (size_t)cid * gstride + lane : deg + cid);
                }
                dg_all = gstride < kWave ? (uint32_t)__builtin_amdgcn_readlane((int)nb_first, kWave - 1) : deg[cid];
                pre_id = 0xFFFFFFFFu;
                if constexpr (PROF) { const unsigned long long n = __builtin_amdgcn_s_memtime(); list_wait += n - pt + (dg_all & 0u); }
                for (uint32_t lbase = 0; lbase < dg_all; lbase += kWave) {
                    const uint32_t cnt = min(dg_all - lbase, (uint32_t)kWave);
                    sync_w();
                    if (hash_used + cnt >= hash_limit) overflow = true;  // wave-uniform
                    if (overflow && (pass == kEfTight || pass == kEfRetryMark)) {  // the launch behind this one serves the query
                        if (lane == 0) ef_mark_again(out_count, qi, net_list);
                        release_helpers();
                        return;
                    }
                    uint32_t nbl = 0;
                    bool fresh1 = false;
                    if constexpr (EPL >= 8) {
                        bool full1 = false;
                        if ((uint32_t)lane < cnt && !overflow) {
                            nbl = lbase == 0 ? nb_first : adj[(size_t)cid * gstride + lbase + lane];
                            fresh1 = visit(nbl, full1);
                        }
                        if (__ballot(full1)) {  // a region of the packed table is full: the launch behind this one serves the query
                            if (pass == kEfTight || pass == kEfRetryMark) {
                                if (lane == 0) ef_mark_again(out_count, qi, net_list);
                                release_helpers();
                                return;
                            }
                            overflow = true;
                        }
                    } else
                    if ((uint32_t)lane < cnt && !overflow) {
                        nbl = lbase == 0 ? nb_first : adj[(size_t)cid * gstride + lbase + lane];
                        fresh1 = hashn_insert_atomic(hash, hash_size, nbl);
                    }
                    // the fresh neighbours move to the front of nb_i (list order kept): the distance passes below take 16 rows each,
                    // and ~15 of a 32-entry list are fresh in the steady state -- one pass instead of two for most pops
                    const unsigned long long fb0 = __ballot(fresh1);
                    if (fresh1) nb_i[__popcll(fb0 & ((1ull << lane) - 1ull))] = nbl;
                    const uint32_t nf = (uint32_t)__popcll(fb0);
                    const unsigned long long fb = nf == kWave ? ~0ull : (1ull << nf) - 1ull;
                    if (WPQ > 1 && nf != 0) {
                        if (lane == 0) ctl[0] = nf;
                        __syncthreads();  // (A) window published to the helper waves
                    } else {
                        sync_w();
                    }
                    if (lbase == 0 && gstride < kWave) {  // speculative list request for the next pop, in flight during the distance phase (47 % of
                                                          // pops use it at 1M x 768, ef = 128: +1.7 % QPS on bf16 rows, +0.7 % on fp32, A/B on one box)
                        uint32_t slot2 = kWave * EPL;
    #pragma unroll
                        for (int e = EPL - 1; e >= 0; --e)
                            if (~xm[e] != 0ull) slot2 = (uint32_t)e * kWave + (uint32_t)__builtin_ctzll(~xm[e]);
                        if (slot2 < wn) {
    #pragma unroll
                            for (int e = 0; e < EPL; ++e)
                                if (slot2 >> 6 == (uint32_t)e) pre_id = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wkr[e], (int)(slot2 & 63));
                            pre_list = *((uint32_t)lane < gstride ? adj + (size_t)pre_id * gstride + lane : deg + pre_id);
                        }
                    }
                    stamp(1);  // degree + list load + visited test-and-set
                    if (nf == 0) continue;
                    ef_window_distances<RT, WPQ, U>(rows, ld, q, nchunks, metric, nf, nb_i, nb_d, lane, 0);
                    hash_used += nf;
                    evals += nf;
                    __syncthreads();  // (B) distances done
                    stamp(2);  // blocked distances of the fresh neighbours
                    uint64_t ck = ~0ull;
                    if ((fb >> lane) & 1ull) ck = ((uint64_t)f32_key(nb_d[lane]) << 32) | nb_i[lane];
                    const bool pass = ck != ~0ull && (wn < ef || ck < worst);
                    unsigned long long todo = __ballot(pass);
                    const uint32_t na = (uint32_t)__popcll(todo);
                    if constexpr (LMERGE) {
                        if (na != 0) merge_lds(ck, pass, todo, na);
                    } else if (na == 1) {
                        insert_sorted(readlane64(ck, __builtin_ctzll(todo)));
                    } else if (na > 1) {
                        // several candidates: ONE merge by ranks instead of na insertions.  new slot of a W entry = its slot + #(candidates
                        // below it); new slot of a candidate = #(W entries below it) + #(candidates below it) -- per candidate two readlanes,
                        // a compare + add per block and per candidate lane, and a ballot/popcount per block, ~16 instructions against ~70
                        // for an insertion; then one scatter through LDS (slots >= ef fall off: the best ef of the union stay).
                        uint32_t shift[EPL], crank = 0, wless = 0;
    #pragma unroll
                        for (int e = 0; e < EPL; ++e) shift[e] = 0;
                        while (todo) {
                            const int l = __builtin_ctzll(todo);
                            todo &= todo - 1;
                            const uint64_t ckl = readlane64(ck, l);
                            uint32_t below = 0;
    #pragma unroll
                            for (int e = 0; e < EPL; ++e) {
                                const bool valid = (uint32_t)(e * kWave + lane) < wn;
                                shift[e] += (valid && ckl < wkr[e]) ? 1u : 0u;
                                below += (uint32_t)__popcll(__ballot(valid && wkr[e] < ckl));
                            }
                            crank += ckl < ck ? 1u : 0u;
                            if (lane == l) wless = below;
                        }
    #pragma unroll
                        for (int e = 0; e < EPL; ++e) {
                            const uint32_t slot = (uint32_t)(e * kWave + lane), np = slot + shift[e];
                            if (slot < wn && np < ef) { Wt[np] = wkr[e]; Ft[np] = (uint32_t)(xm[e] >> lane) & 1u; }
                        }
                        if (pass && wless + crank < ef) { Wt[wless + crank] = ck; Ft[wless + crank] = 0; }
                        wn = min(wn + na, ef);
                        sync_w();
    #pragma unroll
                        for (int e = 0; e < EPL; ++e) {
                            const uint32_t slot = (uint32_t)(e * kWave + lane);
                            const bool live = slot < wn;
                            if (live) wkr[e] = Wt[slot];
                            xm[e] = __ballot(live && Ft[slot] != 0);
                        }
                        if (wn == ef) {
    #pragma unroll
                            for (int e = 0; e < EPL; ++e)
                                if ((ef - 1) >> 6 == (uint32_t)e) worst = readlane64(wkr[e], (int)((ef - 1) & 63));
                        }
                    }
                    stamp(3);  // beam update
                }
            }
} else {
        for (;;) {
                // next to expand: the first unexpanded slots of W -- one (the search of orc_hnsw_search_ef_ex) or, with a search width
                // (ZVDB_OPT_EF_SEARCH_WIDTH; spec: orc_hnsw_search_ef_w), up to `width` of them, ALL marked expanded before any of
                // their lists is read
                uint32_t cids[4] = {0, 0, 0, 0}, nsel = 0;
    #pragma unroll
                for (uint32_t w = 0; w < 4; ++w) {
                    if (w >= width) break;
                    uint32_t slot = kWave * EPL;
    #pragma unroll
                    for (int e = EPL - 1; e >= 0; --e)
                        if (~xm[e] != 0ull) slot = (uint32_t)e * kWave + (uint32_t)__builtin_ctzll(~xm[e]);
                    if (slot >= wn) break;
                    uint64_t kb = 0;
    #pragma unroll
                    for (int e = 0; e < EPL; ++e)
                        if (slot >> 6 == (uint32_t)e) { kb = readlane64(wkr[e], (int)(slot & 63)); xm[e] |= 1ull << (slot & 63); }
                    if constexpr (LMERGE) {
                        if (lane == 0) Ft[slot] = 1;
                    }
                    cids[w] = (uint32_t)kb;
                    nsel = w + 1;
                }
                if (nsel == 0) break;
                pops += nsel;
                stamp(0);  // next-to-expand
                // one window: <= 64 list entries (one lane each) -> visited test-and-set, the fresh ones compacted, their distances (shared
                // with the helper waves), the beam update.  false: the visited table is full and a later launch serves the query.
                auto window = [&](bool valid, uint32_t nbl, uint32_t cnt, bool prefetch_ok) -> bool {
                    sync_w();
                    if (hash_used + cnt >= hash_limit) overflow = true;  // wave-uniform
                    if (overflow && (pass == kEfTight || pass == kEfRetryMark)) {  // the launch behind this one serves the query
                        if (lane == 0) ef_mark_again(out_count, qi, net_list);
                        release_helpers();
                        return false;
                    }
                    bool fresh1 = false, full1 = false;
                    if (valid && !overflow) fresh1 = visit(nbl, full1);
                    if constexpr (EPL >= 8) {
                        if (__ballot(full1)) {  // a region of the packed table is full: the launch behind this one serves the query
                            if (pass == kEfTight || pass == kEfRetryMark) {
                                if (lane == 0) ef_mark_again(out_count, qi, net_list);
                                release_helpers();
                                return false;
                            }
                            overflow = true;
                        }
                    }
                    // the fresh neighbours move to the front of nb_i (list order kept): the distance passes below take 16 rows each,
                    // and ~15 of a 32-entry list are fresh in the steady state -- one pass instead of two for most pops
                    const unsigned long long fb0 = __ballot(fresh1);
                    if (fresh1) nb_i[__popcll(fb0 & ((1ull << lane) - 1ull))] = nbl;
                    const uint32_t nf = (uint32_t)__popcll(fb0);
                    const unsigned long long fb = nf == kWave ? ~0ull : (1ull << nf) - 1ull;
                    if (WPQ > 1 && nf != 0) {
                        if (lane == 0) ctl[0] = nf;
                        __syncthreads();  // (A) window published to the helper waves
                    } else {
                        sync_w();
                    }
                    if (prefetch_ok && gstride < kWave) {  // speculative list request for the next pop, in flight during the distance phase (47 % of
                                                          // pops use it at 1M x 768, ef = 128: +1.7 % QPS on bf16 rows, +0.7 % on fp32, A/B on one box)
                        uint32_t slot2 = kWave * EPL;
    #pragma unroll
                        for (int e = EPL - 1; e >= 0; --e)
                            if (~xm[e] != 0ull) slot2 = (uint32_t)e * kWave + (uint32_t)__builtin_ctzll(~xm[e]);
                        if (slot2 < wn) {
    #pragma unroll
                            for (int e = 0; e < EPL; ++e)
                                if (slot2 >> 6 == (uint32_t)e) pre_id = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wkr[e], (int)(slot2 & 63));
                            pre_list = *((uint32_t)lane < gstride ? adj + (size_t)pre_id * gstride + lane : deg + pre_id);
                        }
                    }
                    stamp(1);  // degree + list load + visited test-and-set
                    if (nf == 0) return true;
                    ef_window_distances<RT, WPQ, U>(rows, ld, q, nchunks, metric, nf, nb_i, nb_d, lane, 0);
                    hash_used += nf;
                    evals += nf;
                    __syncthreads();  // (B) distances done
                    stamp(2);  // blocked distances of the fresh neighbours
                    uint64_t ck = ~0ull;
                    if ((fb >> lane) & 1ull) ck = ((uint64_t)f32_key(nb_d[lane]) << 32) | nb_i[lane];
                    const bool pass = ck != ~0ull && (wn < ef || ck < worst);
                    unsigned long long todo = __ballot(pass);
                    const uint32_t na = (uint32_t)__popcll(todo);
                    if constexpr (LMERGE) {
                        if (na != 0) merge_lds(ck, pass, todo, na);
                    } else if (na == 1) {
                        insert_sorted(readlane64(ck, __builtin_ctzll(todo)));
                    } else if (na > 1) {
                        // several candidates: ONE merge by ranks instead of na insertions.  new slot of a W entry = its slot + #(candidates
                        // below it); new slot of a candidate = #(W entries below it) + #(candidates below it) -- per candidate two readlanes,
                        // a compare + add per block and per candidate lane, and a ballot/popcount per block, ~16 instructions against ~70
                        // for an insertion; then one scatter through LDS (slots >= ef fall off: the best ef of the union stay).
                        uint32_t shift[EPL], crank = 0, wless = 0;
    #pragma unroll
                        for (int e = 0; e < EPL; ++e) shift[e] = 0;
                        while (todo) {
                            const int l = __builtin_ctzll(todo);
                            todo &= todo - 1;
                            const uint64_t ckl = readlane64(ck, l);
                            uint32_t below = 0;
    #pragma unroll
                            for (int e = 0; e < EPL; ++e) {
                                const bool valid = (uint32_t)(e * kWave + lane) < wn;
                                shift[e] += (valid && ckl < wkr[e]) ? 1u : 0u;
                                below += (uint32_t)__popcll(__ballot(valid && wkr[e] < ckl));
                            }
                            crank += ckl < ck ? 1u : 0u;
                            if (lane == l) wless = below;
                        }
    #pragma unroll
                        for (int e = 0; e < EPL; ++e) {
                            const uint32_t slot = (uint32_t)(e * kWave + lane), np = slot + shift[e];
                            if (slot < wn && np < ef) { Wt[np] = wkr[e]; Ft[np] = (uint32_t)(xm[e] >> lane) & 1u; }
                        }
                        if (pass && wless + crank < ef) { Wt[wless + crank] = ck; Ft[wless + crank] = 0; }
                        wn = min(wn + na, ef);
                        sync_w();
    #pragma unroll
                        for (int e = 0; e < EPL; ++e) {
                            const uint32_t slot = (uint32_t)(e * kWave + lane);
                            const bool live = slot < wn;
                            if (live) wkr[e] = Wt[slot];
                            xm[e] = __ballot(live && Ft[slot] != 0);
                        }
                        if (wn == ef) {
    #pragma unroll
                            for (int e = 0; e < EPL; ++e)
                                if ((ef - 1) >> 6 == (uint32_t)e) worst = readlane64(wkr[e], (int)((ef - 1) & 63));
                        }
                    }
                    stamp(3);  // beam update
                    return true;
                };
                const uint32_t per = (uint32_t)kWave / max(gstride, 1u);  // whole lists that fit one window
                if (nsel > 1 && per > 1) {
                    // the lists of the picked nodes side by side in a window (lane = list w, entry j), `per` lists per window: that many
                    // lists' worth of row fetches per dependent step.  The beam is a set, so merging the candidates of several lists
                    // together equals the spec's list-by-list insertion.
                    pre_id = 0xFFFFFFFFu;
                    for (uint32_t s0 = 0; s0 < nsel; s0 += per) {
                        const uint32_t nl = min(per, nsel - s0);
                        const uint32_t w = (uint32_t)lane / gstride, j = (uint32_t)lane - w * gstride;
                        const bool live = w < nl;
                        const uint32_t iw = s0 + w;
                        const uint32_t cw = iw == 0 ? cids[0] : iw == 1 ? cids[1] : iw == 2 ? cids[2] : cids[3];
                        uint32_t nbl = 0, dgw = 0;
                        if (live) { nbl = adj[(size_t)cw * gstride + j]; dgw = min(deg[cw], gstride); }
                        const bool valid = live && j < dgw;
                        const uint32_t cnt = (uint32_t)__popcll(__ballot(valid));
                        if (cnt != 0 && !window(valid, nbl, cnt, false)) return;
                    }
                } else {
                    for (uint32_t s = 0; s < nsel; ++s) {
                        const uint32_t cid = cids[s];
                        // one vector load brings the first list slots AND the degree (lane 63 reads deg[cid] when the stride leaves it free) --
                        // or brought them already: the list of the NEXT unexpanded slot is requested speculatively a whole distance phase
                        // ahead (in `window`); it is this pop's node unless a candidate of the previous pop went in before it
                        uint32_t nb_first, dg_all;  // lists longer than 64 are walked 64 entries at a time
                        if (cid == pre_id) {
                            nb_first = pre_list;
                            if constexpr (PROF) pre_hits += 1;
                        } else {
                            nb_first = *((uint32_t)lane < gstride ? adj + (size_t)cid * gstride + lane : deg + cid);
                        }
                        dg_all = gstride < kWave ? (uint32_t)__builtin_amdgcn_readlane((int)nb_first, kWave - 1) : deg[cid];
                        pre_id = 0xFFFFFFFFu;
                        if constexpr (PROF) { const unsigned long long n = __builtin_amdgcn_s_memtime(); list_wait += n - pt + (dg_all & 0u); }
                        for (uint32_t lbase = 0; lbase < dg_all; lbase += kWave) {
                            const uint32_t cnt = min(dg_all - lbase, (uint32_t)kWave);
                            const bool valid = (uint32_t)lane < cnt;
                            const uint32_t nbl = !valid ? 0u : lbase == 0 ? nb_first : adj[(size_t)cid * gstride + lbase + lane];
                            if (!window(valid, nbl, cnt, lbase == 0 && width == 1)) return;
                        }
                    }
                }
            }
}
        sync_w();
    }
    release_helpers();
    stamp(4);
    // the first k slots of W, re-evaluated with the exact reference arithmetic, ordered by (exact distance, id)
    const uint32_t nres = wn < k ? wn : k;
    if (nres > 0) {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            if ((uint32_t)(e * kWave + lane) < nres) R[e * kWave + lane] = (uint32_t)wkr[e];
        sync_w();
        for (uint32_t i = lane; i < nres; i += kWave)
            RD[i] = exact_dist_rt(metric, q, rows + (size_t)R[i] * ld, dim);
        sync_w();
        for (uint32_t i = lane; i < nres; i += kWave) {
            const float di = RD[i];
            const uint32_t ii = R[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < nres; ++j)
                if (lex_less(RD[j], R[j], di, ii)) ++rank;
            out_ids[(size_t)qi * k + rank] = (uint64_t)ii + id_offset;
            out_dist[(size_t)qi * k + rank] = di;
        }
    }
    for (uint32_t i = nres + lane; i < k; i += kWave) {
        out_ids[(size_t)qi * k + i] = ~0ull;
        out_dist[(size_t)qi * k + i] = __builtin_inff();
    }
    if (lane == 0) {
        out_count[qi] = nres;
        if (fill_hint) atomicMax(fill_hint, hash_used);
        if (stats) {
            atomicAdd(&stats[0], evals);
            atomicAdd(&stats[1], pops);
            if (overflow) atomicAdd(&stats[2], 1ull);  // visited set exhausted: result not to spec
        }
        if constexpr (PROF) {
            stamp(5);  // final k: exact re-evaluation + order
            for (int i = 0; i < 6; ++i) atomicAdd(&g_ef_prof[i], pf[i]);
            atomicAdd(&g_ef_prof[6], pops);
            atomicAdd(&g_ef_prof[8], pre_hits);
            atomicAdd(&g_ef_prof[9], list_wait);
            atomicAdd(&g_ef_prof[7], 1ull);
        }
    }
}

// ---- hierarchical entry descent of the bulk graph (extension; specification: orc_hnsw_descend) -----------------------------
// One workgroup of WPQ waves per query: from the graph's top node down the layers max_level .. 1; on each layer repeat { scan
// the WHOLE list of the current node in list order with the blocked traversal distance, keep the strict arg-min (the earliest
// entry wins ties), move there } until nothing is closer (hnsw.zig:90-104's greedy step, run top-down, on the device's
// upper-layer lists: node's layer L list = row slot[node] + L - 1 of adj_up).  The reference's level distribution (each further
// level with probability 1/2, hnsw.zig:172-180) makes this ~20 layers at 1M nodes, i.e. a chain of ~40 dependent hops: a hop
// evaluates its whole list (16 rows) in ONE step shared by the waves (ef_window_distances), then every wave takes the same
// (distance, position) minimum from LDS -- the first version walked 4 rows per step on one wave and cost more than an ef = 32 search.
template <typename RT, int WPQ>
__global__ __launch_bounds__(64 * WPQ) void k_descend(const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
                                                      const uint32_t *__restrict__ lvl, const uint32_t *__restrict__ slot,
                                                      const uint32_t *__restrict__ adj_up, const uint32_t *__restrict__ deg_up, uint32_t gstride,
                                                      const float *__restrict__ queries, uint32_t top, uint32_t *__restrict__ out_entry,
                                                      unsigned long long *__restrict__ stats, uint32_t dim_pad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *q = reinterpret_cast<float *>(smem);
    float *nb_d = q + dim_pad;                                  // [64]
    uint32_t *nb_i = reinterpret_cast<uint32_t *>(nb_d + kWave);  // [64]
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int t = lane & 15;
    const uint32_t qi = blockIdx.x;
    const float *qg = queries + (size_t)qi * dim;
    for (uint32_t i = threadIdx.x; i < dim_pad; i += kWave * WPQ) q[i] = i < dim ? qg[i] : 0.0f;
    __syncthreads();
    const uint32_t nchunks = (dim + 3) / 4;
    constexpr int U = sizeof(RT) == 2 ? (WPQ == 1 ? 2 : 1) : 4 / WPQ;  // 16 rows per step
    // every wave keeps the same (cur, dcur): all decisions below derive from LDS / global values the waves share
    uint32_t cur = top;
    float dcur = metric == ZVDB_METRIC_L2SQ ? blocked16_dist<ZVDB_METRIC_L2SQ>(q, rows + (size_t)cur * ld, nchunks, t)
                                            : blocked16_dist<ZVDB_METRIC_DOT>(q, rows + (size_t)cur * ld, nchunks, t);
    dcur = __shfl(dcur, 0);
    unsigned long long evals = 1;
    for (uint32_t layer = lvl[top]; layer >= 1; --layer) {
        for (;;) {
            if (lvl[cur] < layer) break;
            const size_t row = (size_t)slot[cur] + layer - 1;
            const uint32_t dg = min(deg_up[row], gstride);
            uint32_t best = cur;
            float dbest = dcur;
            for (uint32_t base = 0; base < dg; base += kWave) {
                const uint32_t cnt = min(dg - base, (uint32_t)kWave);
                __syncthreads();  // (the previous window's readers are done)
                if (wave == 0 && (uint32_t)lane < cnt) nb_i[lane] = adj_up[row * gstride + base + lane];
                __syncthreads();
                ef_window_distances<RT, WPQ, U>(rows, ld, q, nchunks, metric, cnt, nb_i, nb_d, lane, wave);
                __syncthreads();
                // list order: the smallest (distance, position); strict '<' against the running best keeps the earliest of equal minima
                const uint64_t key = (uint32_t)lane < cnt ? ((uint64_t)f32_key(nb_d[lane]) << 32) | (uint32_t)lane : ~0ull;
                const uint32_t jm = (uint32_t)wave_min64(key);
                const float dm = nb_d[jm];
                if (dm < dbest) { dbest = dm; best = nb_i[jm]; }
            }
            evals += dg;
            if (best == cur) break;
            cur = best;
            dcur = dbest;
        }
    }
    if (threadIdx.x == 0) {
        out_entry[qi] = cur;
        if (stats) atomicAdd(&stats[0], evals);
    }
}

int launch_descend(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t top, uint32_t *d_entries,
                   unsigned long long *d_stats, hipStream_t s) {
    if (nq == 0) return ZVDB_OK;
    // q in LDS: zero padded to the rows' padded length for bf16 rows (the unmasked 8-lane walk reads whole 128-byte lines)
    const uint32_t dim_pad = (h->flags & ZVDB_FLAG_BF16_ROWS) ? std::max(h->ldh, 64u) : round_up(std::max(h->dim, 1u), 4);
    const size_t lds = 4ull * dim_pad + 8ull * kWave;
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    return rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = std::remove_cv_t<std::remove_pointer_t<decltype(rows)>>;
        constexpr int WPQ = sizeof(RT) == 2 ? 2 : 4;
        ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_descend<RT, WPQ>), lds));
        hipLaunchKernelGGL((k_descend<RT, WPQ>), dim3(nq), dim3(64 * WPQ), lds, s, rows, ld, h->dim, metric, h->lvl, h->slot, h->adj_up, h->deg_up,
                           h->gstride, d_queries, top, d_entries, d_stats, dim_pad);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    });
}

// ---- index-owned scratch of the ef search (all mutable members of the index; callers hold h->mu) -------------------------------
static int ef_scratch_begin(const zvdb_hip_index *h, hipStream_t s) {
    if (!h->ef_event) ZV_HIP(hipEventCreateWithFlags(&h->ef_event, hipEventDisableTiming));
    if (h->ef_event_set) ZV_HIP(hipStreamWaitEvent(s, h->ef_event, 0));
    return ZVDB_OK;
}
static int ef_scratch_end(const zvdb_hip_index *h, hipStream_t s) {
    ZV_HIP(hipEventRecord(h->ef_event, s));
    h->ef_event_set = true;
    return ZVDB_OK;
}
// (re)allocation frees a buffer other streams may still be reading: wait for the whole device first (rare: sizes only grow)
static int ef_reserve_visit(const zvdb_hip_index *h, size_t bytes) {
    if (h->ef_visit_bytes >= bytes) return ZVDB_OK;
    if (h->ef_visit_buf) { ZV_HIP(hipDeviceSynchronize()); (void)hipFree(h->ef_visit_buf); h->ef_visit_buf = nullptr; h->ef_visit_bytes = 0; }
    ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->ef_visit_buf), bytes));
    h->ef_visit_bytes = bytes;
    return ZVDB_OK;
}
// table of the net: >= 16 ef stride slots (a straggler that filled the safe LDS table of 1.5 ef stride)
static uint32_t ef_net_bits(const zvdb_hip_index *h, uint32_t ef) {
    const uint64_t N = std::max(h->count, h->cap);  // (from the capacity: rows added later must not make a reserved search reallocate)
    return bits_for(std::min<uint64_t>(std::max<uint64_t>(N, 1), std::max<uint64_t>((uint64_t)ef * h->gstride * 16, 65536)));
}
// the net's buffer: [list: 2 + nq words, padded to 256 B][kEfNetBlocks tables of 4 << bits bytes] -- 4 MiB at ef <= 256 / stride
// 16, 32 MiB at ef = 1024 / stride 32 (before: 256 tables, up to 1 GiB, taken by the first search whether or not it needed them)
static int ef_reserve_net(const zvdb_hip_index *h, uint32_t nq, uint32_t ef) {
    const size_t list_words = round_up64((uint64_t)nq + 2, 64);
    const size_t table_words = (size_t)kEfNetBlocks << ef_net_bits(h, ef);
    if (h->ef_net_list_words >= list_words && h->ef_net_table_words >= table_words) return ZVDB_OK;
    const size_t lw = std::max(list_words, h->ef_net_list_words), tw = std::max(table_words, h->ef_net_table_words);
    if (h->ef_net_buf) { ZV_HIP(hipDeviceSynchronize()); (void)hipFree(h->ef_net_buf); h->ef_net_buf = nullptr; h->ef_net_list_words = h->ef_net_table_words = 0; }
    ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->ef_net_buf), (lw + tw) * sizeof(uint32_t)));
    h->ef_net_list_words = lw;
    h->ef_net_table_words = tw;
    return ZVDB_OK;
}

// waves per query of the register-beam kernel (epl: its beam blocks).  Measured at 1M x 768, M = 16, ef = 128, batch 1,024 / 4,096 /
// 16,384 (QPS): fp32 rows 1 wave 775k / 901k / 993k, 2 waves 863k / 982k / 1.09M, 4 waves 894k / 1.03M / 1.11M; bf16 rows 1.00M /
// 1.19M / 1.30M, 2 waves 1.23M / 1.54M / 1.67M, 4 waves 0.69M (8 rows per wave step: half the waves idle on 16-entry lists; with the
// 16-lane walk 0.94M / 1.20M / 1.29M).  ZVDB_EF_WPQ = 1 / 2 / 4 / 8: experiment / tests, read per call.
// Large beams (epl 8, 16: ef > 256) with a search width: the visited table leaves room for one or two queries per CU (lds_g: what the
// LDS allows, 0 = unknown), so the idle wave slots join the query's distance phases instead -- 4 waves per query at two queries per
// CU, 8 at one (1M x 768, M = 16, width 4, 2 / 4 / 8 waves: ef = 512 bf16 rows 234k / 290k / 189k QPS, fp32 rows 181k / 235k / 168k;
// ef = 1,024 bf16 rows 57.5k / 70.3k / 77.3k, fp32 rows 48.3k / 58.6k / 71.4k; profiles/r4/ef_large_probe_*).
static uint32_t ef_waves_per_query(const zvdb_hip_index *h, int epl, uint32_t width, uint32_t lds_g) {
    uint32_t wq = (h->flags & ZVDB_FLAG_BF16_ROWS) ? 2u : 4u;
    bool forced = false;
    if (const char *we = tuning("ZVDB_EF_WPQ")) { wq = (uint32_t)atoi(we); forced = true; }
    if (wq != 1 && wq != 2 && wq != 4 && wq != 8) wq = 1;
    if (epl < 8 && wq == 8) wq = 4;                       // (not instantiated)
    if (epl >= 8 && width <= 1 && wq >= 4) wq = 2;        // the wide workgroups exist for the search-width loop only
    if (epl >= 8 && width > 1 && !forced) wq = lds_g == 1 ? 8u : lds_g == 2 ? 4u : 2u;  // (lds_g: queries per CU of the two-wave plan)
    return wq;
}

int launch_search_ef(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t k,
                     uint32_t ef, uint32_t entry, uint64_t id_offset, uint64_t *d_ids,
                     float *d_dist, uint32_t *d_count, unsigned long long *d_stats, hipStream_t s, const uint32_t *d_entries,
                     bool reserve_only) {
    if (nq == 0) return ZVDB_OK;
    const uint32_t width = std::max(1u, std::min(4u, h->ef_width));  // ZVDB_OPT_EF_SEARCH_WIDTH (spec: orc_hnsw_search_ef_w)
    if (ef < k) ef = k;
    const uint64_t N = h->count;
    // q in LDS: zero padded to the rows' padded length for bf16 rows (the unmasked 8-lane walk reads whole 128-byte lines)
    const uint32_t dim_pad = (h->flags & ZVDB_FLAG_BF16_ROWS) ? std::max(h->ldh, 64u) : round_up(std::max(h->dim, 1u), 4);
    // visited set: every node is visited at most once.  Measured evaluations per query are ~1.1 x
    // ef * stride; the table takes >= 3 x ef * stride slots (75 % load cap => room for 2.2 x), because its
    // LDS footprint sets the waves per CU (the first version reserved 8 x and ran ONE wave per CU).
    // A query that still exhausts it is flagged in stats[2] (the host entry point reports it).
    uint64_t want = std::min<uint64_t>(std::max<uint64_t>(N, 1), (uint64_t)ef * h->gstride * 3 / 2 + 64);
    uint32_t hash_bits = bits_for(want);  // the SAFE table
    if (const char *hb_env = tuning("ZVDB_EF_HASH_BITS")) hash_bits = (uint32_t)std::max(8, std::min(15, atoi(hb_env)));  // (experiment)
    const uint32_t efp = round_up(ef, kWave), kpad = round_up(std::max(k, 1u), kWave);
    // beam in registers up to ef = 1024 (k_search_ef_reg), in LDS beyond (k_search_ef); ZVDB_EF_BEAM=lds forces the LDS kernel
    const char *beam_env = tuning("ZVDB_EF_BEAM");  // (read per call: the parity test switches it)
    const bool beam_lds = beam_env && beam_env[0] == 'l';
    const int epl = beam_lds ? 0 : ef <= 128 ? 2 : ef <= 256 ? 4 : ef <= 512 ? 8 : ef <= 1024 ? 16 : 0;
    auto lds_slots = [&](uint64_t slots) -> size_t {  // LDS of a query with a visited table of `slots` entries
        if (epl) return 4ull * dim_pad + 12ull * kWave * epl + 4ull * slots + 8ull * kWave + 8ull * kpad + 16 + (epl >= 8 ? 8ull * kWave + 8ull * epl : 0);  // (+ Cs, Mb of merge_lds)
        return 4ull * dim_pad + 2ull * efp * 8 + 2ull * efp * 4 + 1024 + 4ull * slots + 8ull * kWave;
    };
    auto lds_bytes = [&](uint32_t hb) -> size_t { return lds_slots(1ull << hb); };
    // A safe table that leaves fewer than two workgroups per CU (ef * stride beyond ~12k) does not go to the LDS at all: the safe
    // pass then keeps the visited sets in GLOBAL memory (k_search_ef<GH>: a pre-zeroed table of >= 8 ef stride slots per query in
    // an index-owned buffer, the batch in chunks of <= 2 GiB) -- slower per pop, but it serves only what the tight LDS pass
    // handed back (or the first batch of an ef bucket, which establishes the fill), and nothing overflows: before, such searches
    // ran one wave per CU on a table clamped to the LDS and failed with "visited set exhausted" beyond it.
    const char *gv_env = tuning("ZVDB_EF_GLOBAL_VISITED");  // "1" / "0": force / forbid the global safe pass (tests; read per call)
    uint32_t lds_cap_bits = 15;
    while (lds_bytes(lds_cap_bits) > 160 * 1024 && lds_cap_bits > 8) --lds_cap_bits;
    const bool safe_global = gv_env ? gv_env[0] == '1' : lds_bytes(hash_bits) > 80 * 1024;
    if (!safe_global) hash_bits = std::min(hash_bits, lds_cap_bits);
    uint32_t gbits = bits_for(std::min<uint64_t>(std::max<uint64_t>(N, 1), (uint64_t)ef * h->gstride * 4 + 64));
    // two-pass sizing (see kEfTight): the largest visited-set fill this index has seen for searches of this ef bucket
    if (!h->ef_hint_dev) {
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->ef_hint_dev), 64));
        ZV_HIP(hipMemsetAsync(h->ef_hint_dev, 0, 64, s));
        ZV_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->ef_hint_host), 64, hipHostMallocDefault));
        std::memset(h->ef_hint_host, 0, 64);
    }
    uint32_t bucket = 0;
    while ((1u << bucket) < ef && bucket < 15) ++bucket;
    const uint32_t seen = __atomic_load_n(&h->ef_hint_host[bucket], __ATOMIC_RELAXED);
    const char *tp_env = tuning("ZVDB_EF_TWO_PASS");  // "0": always the safe table; "t<bits>": force the tight table (tests; read per call)
    const uint32_t safe_rank = safe_global ? 32u : hash_bits;  // (a global safe table is "larger" than any LDS one)
    uint32_t tight_bits = safe_rank;
    if (tp_env && tp_env[0] == 't') {
        tight_bits = std::min(std::min(safe_rank, lds_cap_bits), (uint32_t)std::max(8, atoi(tp_env + 1)));
    } else if (seen && !(tp_env && tp_env[0] == '0')) {
        tight_bits = 8;
        while ((1ull << tight_bits) < (uint64_t)seen * 5 / 2 + 128) ++tight_bits;  // <= 40 % load at the largest fill seen
        if (tight_bits > lds_cap_bits) tight_bits = safe_rank;                      // (does not fit the LDS: no tight pass)
        tight_bits = std::min(tight_bits, safe_rank);
    }
    // Register-beam kernel: the tight table takes ANY slot count, chosen for the queries per CU it allows.  A query is one chain of
    // dependent pops (latency), so throughput = queries in flight x a rate per query, and a batch runs in ceil(nq / (CUs x queries per
    // CU)) rounds: 1,024 queries at ef = 256 on 1536-d bf16 rows ran as two rounds of 512 on 64-KB (2^14-slot) tables that were 22 %
    // full.  Rule: the MOST queries per CU (up to what the kernel's VGPRs allow) whose table still holds the largest fill seen at
    // <= 74 % load (cap 87.5 %: beyond it the safe pass serves the query, as before), never more slots than 2.5 x that fill.
    uint32_t tight_slots = 0;
    uint32_t wq_sel = epl ? ef_waves_per_query(h, epl, width, 0) : 1u;  // waves per query of the launch
    // large beams keep the visited set in 21-bit entries, three to a 64-bit word (hp_insert_atomic): 1.5 entries per 32-bit unit of LDS
    // where the plain table has one.  hp_b: bits of an id; its low hp_r bits (after a bijection) pick the region.  Needs regions of >=
    // 16 words (beyond ~10^8 rows at these table sizes they get too small: the plain table then); ZVDB_EF_PACKED_VISITED=0 / 1: tests
    uint32_t hp_b = 1;
    while (hp_b < 32 && (1ull << hp_b) < std::max<uint64_t>(N, 2)) ++hp_b;
    uint32_t hp_r = hp_b > 20 ? hp_b - 20 : 0;
    const char *pk_env = tuning("ZVDB_EF_PACKED_VISITED");
    const char *pr_env = tuning("ZVDB_EF_PACKED_R");  // tests: MORE regions than the row count needs (small regions fill up: the retry path)
    if (pr_env) hp_r = std::min(hp_b, std::max(hp_r, (uint32_t)atoi(pr_env)));
    auto packed_fits = [&](uint64_t units) { return ((units / 2) >> hp_r) >= (pr_env ? 1u : 16u); };
    bool packed = epl >= 8 && !(pk_env && pk_env[0] == '0') && (safe_global || packed_fits(1ull << hash_bits));
    if (epl && tp_env && tp_env[0] == 's') {  // "s<slots>": force a tight table of that many slots (tests: overflow + retry on a non-power-of-two table)
        tight_slots = (uint32_t)std::max(64, std::min(1 << 15, atoi(tp_env + 1)));
        if (lds_slots(tight_slots) > 160 * 1024) tight_slots = 0;
        if (tight_slots) tight_bits = 0;
    } else if (epl && !tp_env) {
        // (no fill seen yet for this ef on this index: the first batch assumes ef * stride / 2 -- about what searches visit --
        // instead of running whole on the safe table; a query that visits more is served by the safe pass like any other overflow)
        const uint32_t seen_eff = seen ? seen : (uint32_t)std::min<uint64_t>(std::max<uint64_t>(N, 1), (uint64_t)ef * h->gstride / 2);
        // (entries wanted: smin .. spref; in 32-bit units of LDS that is 2/3 of it on the packed table)
        const uint64_t smin_e = (uint64_t)seen_eff * 27 / 20 + 64, spref_e = (uint64_t)seen_eff * 5 / 2 + 128;
        packed = packed && packed_fits(smin_e * 2 / 3);
        const uint64_t smin = packed ? (smin_e * 2 + 2) / 3 : smin_e, spref = packed ? round_up64((spref_e * 2 + 2) / 3, 64) : spref_e;
        auto fit_of = [&](uint32_t g) -> uint64_t {  // 32-bit units a table may take with g queries per CU (0: the fill does not fit)
            const size_t per = (160u * 1024u) / g;
            if (per <= lds_slots(0) + 4 * smin) return 0;
            const uint64_t fit = ((per - lds_slots(0)) / 4) & ~63ull;
            return fit >= smin ? fit : 0;
        };
        static const uint32_t ncu = [] { int v = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return (uint32_t)std::max(v, 1); }();
        // queries per CU under a register limit of vlim workgroups: the MOST whose table holds the fill, and of those the FEWEST that
        // take no more rounds for this batch (a larger table probes less); 0: the fill does not fit the LDS
        auto plan = [&](uint32_t vlim) -> uint32_t {
            uint32_t gmax = 0;
            for (uint32_t g = vlim; g >= 1 && !gmax; --g) if (fit_of(g)) gmax = g;
            if (!gmax) return 0;
            const uint32_t rounds = (nq + ncu * gmax - 1) / (ncu * gmax);
            uint32_t g = gmax;
            while (g > 1 && fit_of(g - 1) && (nq + ncu * (g - 1) - 1) / (ncu * (g - 1)) == rounds) --g;
            return g;
        };
        // workgroups per CU the kernels' registers allow: 2 waves x <= 160 VGPRs (EPL <= 4): 6; 2 x 172 .. 236 (EPL 8, 16), 4 x 128, 1 x 256+: 4;
        // the wide workgroups of the large beams (EPL 8, 16 at 4 / 8 waves of <= 256 VGPRs): 2 / 1 -- taken (ef_waves_per_query) when
        // the two-wave plan would leave the CU with two queries or one anyway
        const uint32_t wq = wq_sel = ef_waves_per_query(h, epl, width, epl >= 8 ? plan(4) : 0);
        const uint32_t vlim = wq == 8 ? 1u : (wq == 4 && epl >= 8) ? 2u : (wq == 2 && epl <= 4) ? 6u : 4u;
        if (const uint32_t g = plan(vlim)) tight_slots = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(fit_of(g), spref), 1ull << 15);
        if (tight_slots && !safe_global && tight_slots >= (1u << hash_bits)) tight_slots = 0;  // (the safe table is no larger: one pass)
        if (tight_slots) tight_bits = 0;  // (marks "sized in slots" below)
    }
    const bool two_pass = tight_slots ? true : tight_bits < safe_rank;
    if (packed && two_pass && !packed_fits(tight_slots ? tight_slots : 1ull << tight_bits)) packed = false;  // (forced tiny tables of the tests)
    const uint32_t width_mode = epl >= 8 ? (width | hp_r << 8 | hp_b << 16 | (packed ? 1u << 24 : 0u)) : width;  // k_search_ef_reg's last argument
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    const bool prof = [] { const char *e = tuning("ZVDB_EF_PROF"); return e && e[0] == '1'; }();
    return rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = std::remove_cv_t<std::remove_pointer_t<decltype(rows)>>;
        uint32_t wpq_launch = 1;
        auto go1 = [&](auto kern, uint32_t wparam, uint32_t hb, uint32_t pass, uint32_t *net_list) -> int {
            // hb: table size as a power of two; the tight pass of the register-beam kernel may carry a slot count instead (hb == 0)
            const uint32_t slots = hb ? (1u << hb) : tight_slots;
            const size_t lds = lds_slots(slots);
            if (epl) hb = slots;  // (k_search_ef_reg takes slots, k_search_ef bits)
            ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(kern), lds));
            hipLaunchKernelGGL(kern, dim3(nq), dim3(64 * wpq_launch), lds, s, rows, ld, h->dim, metric, h->adj, h->deg, h->gstride, (uint32_t)N,
                               d_queries, k, ef, entry, id_offset, d_ids, d_dist, d_count, d_stats, dim_pad, hb, wparam, d_entries,
                               pass, h->ef_hint_dev + bucket, static_cast<uint32_t *>(nullptr), 0u, net_list, epl ? width_mode : width);
            ZV_HIP(hipGetLastError());
            return ZVDB_OK;
        };
        auto go_global = [&](uint32_t pass) -> int {  // the safe pass with the visited sets in global memory
            const size_t tb = 4ull << gbits;
            const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(nq, (2ull << 30) / tb));
            ZV_CHECK(ef_reserve_visit(h, chunk * tb));
            const size_t lds = 4ull * dim_pad + 2ull * efp * 8 + 2ull * efp * 4 + 1024 + 8ull * kWave;
            ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_search_ef<RT, 0, true>), lds));
            for (uint64_t q0 = 0; q0 < nq; q0 += chunk) {
                const uint32_t nc = (uint32_t)std::min<uint64_t>(chunk, nq - q0);
                ZV_HIP(hipMemsetAsync(h->ef_visit_buf, 0, (size_t)nc * tb, s));
                hipLaunchKernelGGL((k_search_ef<RT, 0, true>), dim3(nc), dim3(64), lds, s, rows, ld, h->dim, metric, h->adj, h->deg, h->gstride, (uint32_t)N,
                                   d_queries, k, ef, entry, id_offset, d_ids, d_dist, d_count, d_stats, dim_pad, gbits, efp, d_entries,
                                   pass, h->ef_hint_dev + bucket, h->ef_visit_buf, (uint32_t)q0, static_cast<uint32_t *>(nullptr), width);
                ZV_HIP(hipGetLastError());
            }
            return ZVDB_OK;
        };
        // the net under the LDS passes: a query that fills even the safe LDS table (small ef: the way from the entry node to the
        // query's region alone can visit more than 2 ef stride nodes -- one query in ~5,000 at ef = 16 on a 500k-row shard, found by
        // tools/soak_graph.py) is appended to the index's work list by the last LDS pass and served here, visited set in global
        // memory: kEfNetBlocks workgroups, one table each, pop the list until it is empty (an empty list costs one 16-block launch)
        auto go_net = [&]() -> int {
            const size_t lds = 4ull * dim_pad + 2ull * efp * 8 + 2ull * efp * 4 + 1024 + 8ull * kWave;
            ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_search_ef<RT, 0, true>), lds));
            hipLaunchKernelGGL((k_search_ef<RT, 0, true>), dim3(kEfNetBlocks), dim3(64), lds, s, rows, ld, h->dim, metric, h->adj, h->deg, h->gstride,
                               (uint32_t)N, d_queries, k, ef, entry, id_offset, d_ids, d_dist, d_count, d_stats, dim_pad, ef_net_bits(h, ef), efp, d_entries,
                               (uint32_t)kEfRetry, h->ef_hint_dev + bucket, h->ef_net_buf + h->ef_net_list_words, 0u, h->ef_net_buf, width);
            ZV_HIP(hipGetLastError());
            return ZVDB_OK;
        };
        // Index-owned scratch (global visited tables, the net's list and tables) is shared by every ef search on this index whatever
        // stream it is launched on: a search waits for the previous user's last launch (ef_scratch_begin) and records its own
        // (ef_scratch_end) -- the chain tie_buf has (api.hip).  h->mu serialises the host-side enqueue only.
        auto go = [&](auto kern, uint32_t wparam) -> int {
            if (safe_global) {
                // (sized from the CAPACITY: rows added after zvdb_hip_search_graph_reserve must not make a search reallocate)
                const size_t tb = 4ull << bits_for(std::min<uint64_t>(std::max<uint64_t>(std::max<uint64_t>(N, h->cap), 1), (uint64_t)ef * h->gstride * 4 + 64));
                ZV_CHECK(ef_reserve_visit(h, std::max<uint64_t>(1, std::min<uint64_t>(nq, (2ull << 30) / tb)) * tb));
            } else {
                ZV_CHECK(ef_reserve_net(h, nq, ef));
            }
            if (reserve_only) return ZVDB_OK;  // (zvdb_hip_search_graph_reserve: everything a launch of this shape allocates)
            ZV_CHECK(ef_scratch_begin(h, s));
            if (two_pass) ZV_CHECK(go1(kern, wparam, tight_bits, kEfTight, nullptr));
            if (safe_global) {
                ZV_CHECK(go_global(two_pass ? kEfRetry : kEfSingle));
            } else {
                ZV_HIP(hipMemsetAsync(h->ef_net_buf, 0, 8, s));  // list count + pop cursor
                ZV_CHECK(go1(kern, wparam, hash_bits, two_pass ? kEfRetryMark : kEfTight, h->ef_net_buf));
                ZV_CHECK(go_net());
            }
            ZV_HIP(hipMemcpyAsync(h->ef_hint_host, h->ef_hint_dev, 64, hipMemcpyDeviceToHost, s));  // read by a LATER call, never waited for
            ZV_CHECK(ef_scratch_end(h, s));
            return ZVDB_OK;
        };
        if (prof) {  // diagnostic build of the same kernel: per-phase cycles, printed after a synchronise
            unsigned long long zero[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, got[10];
            ZV_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ef_prof), zero, sizeof(zero)));
            ZV_CHECK(epl == 2 ? go(k_search_ef_reg<RT, 2, 1, 1>, kpad) : epl == 4 ? go(k_search_ef_reg<RT, 4, 1, 1>, kpad)
                     : epl == 8 ? go(k_search_ef_reg<RT, 8, 1, 1>, kpad) : epl == 16 ? go(k_search_ef_reg<RT, 16, 1, 0>, kpad) : go(k_search_ef<RT, 1>, efp));
            ZV_HIP(hipStreamSynchronize(s));
            ZV_HIP(hipMemcpyFromSymbol(got, HIP_SYMBOL(g_ef_prof), sizeof(got)));
            const double nw = (double)std::max<unsigned long long>(got[7], 1), np = (double)std::max<unsigned long long>(got[6], 1);
            fprintf(stderr, "[zvdb ef prof] %llu queries, %.1f pops each; shader cycles per POP: next-scan %.0f, list+visited %.0f, distances %.0f, beam update %.0f; "
                    "per QUERY: tail %.0f, final-k %.0f, total %.0f; list prefetch hits %.2f, list wait %.0f / pop; visited table %u units%s, %u waves per query (safe 2^%u%s)\n", got[7], np / nw,
                    got[0] / np, got[1] / np, got[2] / np, got[3] / np, got[4] / nw, got[5] / nw, (got[0] + got[1] + got[2] + got[3] + got[4] + got[5]) / nw,
                    got[8] / np, got[9] / np, two_pass ? (tight_slots ? tight_slots : 1u << tight_bits) : 1u << (safe_global ? gbits : hash_bits), packed ? " of 1.5 packed entries" : "", wq_sel, safe_global ? gbits : hash_bits, safe_global ? " global" : "");
            return ZVDB_OK;
        }
        const uint32_t wpq = wq_sel;
        wpq_launch = wpq;
        if (width > 1) {  // the search-width form of the loop (its own instantiations: the width-1 loop stays as it was)
            if (wpq == 8) return epl == 8 ? go(k_search_ef_reg<RT, 8, 8, 0, true>, kpad) : go(k_search_ef_reg<RT, 16, 8, 0, true>, kpad);
            if (wpq == 4 && epl >= 8) return epl == 8 ? go(k_search_ef_reg<RT, 8, 4, 0, true>, kpad) : go(k_search_ef_reg<RT, 16, 4, 0, true>, kpad);
            if (wpq == 4) return epl == 2 ? go(k_search_ef_reg<RT, 2, 4, 0, true>, kpad) : go(k_search_ef_reg<RT, 4, 4, 0, true>, kpad);
            if (wpq == 2) return epl == 2 ? go(k_search_ef_reg<RT, 2, 2, 0, true>, kpad) : epl == 4 ? go(k_search_ef_reg<RT, 4, 2, 0, true>, kpad)
                                 : epl == 8 ? go(k_search_ef_reg<RT, 8, 2, 0, true>, kpad) : go(k_search_ef_reg<RT, 16, 2, 0, true>, kpad);
            return epl == 2 ? go(k_search_ef_reg<RT, 2, 1, 0, true>, kpad) : epl == 4 ? go(k_search_ef_reg<RT, 4, 1, 0, true>, kpad)
                   : epl == 8 ? go(k_search_ef_reg<RT, 8, 1, 0, true>, kpad) : epl == 16 ? go(k_search_ef_reg<RT, 16, 1, 0, true>, kpad) : go(k_search_ef<RT>, efp);
        }
        if (wpq == 4) return epl == 2 ? go(k_search_ef_reg<RT, 2, 4>, kpad) : go(k_search_ef_reg<RT, 4, 4>, kpad);
        if (wpq == 2) return epl == 2 ? go(k_search_ef_reg<RT, 2, 2>, kpad) : epl == 4 ? go(k_search_ef_reg<RT, 4, 2>, kpad)
                             : epl == 8 ? go(k_search_ef_reg<RT, 8, 2>, kpad) : go(k_search_ef_reg<RT, 16, 2>, kpad);
        return epl == 2 ? go(k_search_ef_reg<RT, 2>, kpad) : epl == 4 ? go(k_search_ef_reg<RT, 4>, kpad)
               : epl == 8 ? go(k_search_ef_reg<RT, 8>, kpad) : epl == 16 ? go(k_search_ef_reg<RT, 16>, kpad) : go(k_search_ef<RT>, efp);
    });
}

// the graph changed wholesale (upload / bulk build / load): forget the visited-set fills seen on the old one
void ef_hint_reset(const zvdb_hip_index *h, hipStream_t s) {
    if (!h->ef_hint_dev) return;
    (void)hipMemsetAsync(h->ef_hint_dev, 0, 64, s);
    (void)hipStreamSynchronize(s);  // (an in-flight copy of the old values must not land after the reset)
    std::memset(h->ef_hint_host, 0, 64);
}

// ---- K5: exact re-rank of brute-force candidates ---------------------------------
// cand[q][kc] (u32 local ids, 0xFFFFFFFF = empty).  One wave per query: every lane takes
// candidates lane, lane+64, ... and computes the reference-exact distance; the final order
// is (exact dist, id) ascending, duplicates dropped.
// With a certificate (cert.kth != null; bf_cert.h) the lists arrive sorted by screen key and are evaluated 64 at a time: after
// each round the k-th exact distance so far gives the largest screen key a row of the top-k can carry, candidates above it are
// not evaluated, and the query is certified when that key lies below the k'-th survivor's -- else it goes on the repair list.
__device__ __forceinline__ void cert_verdict(const CertArgs &cert, uint32_t qi, uint32_t ks, int lane, uint32_t rows_evaluated = 0) {
    if (lane != 0) return;
    if (cert.dbg_rows && cert.dbg_mode == 1) __hip_atomic_fetch_add(cert.dbg_rows, rows_evaluated, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t kth = cert.kth[qi];
    if (!cert.force && (kth == 0xFFFFFFFFu || ks < kth)) return;  // every row is a survivor / no row left out can reach the top-k
    const uint32_t slot = atomicAdd(cert.nflag, 1u);
    cert.flist[slot] = qi;
    if (cert.host_flag) __hip_atomic_store(cert.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// number of valid entries of the LDS lists [0, n) (n a multiple of 4, the arrays 16-byte aligned) that precede (di, ii) in
// (distance, id) order.  Four entries per 16-byte LDS read, four reads in flight: the plain one-entry loop is a chain of dependent
// LDS round trips (~120 cycles each) -- at 128 candidates the ranking loops of a re-rank cost more than its row fetches
// (profiles/r4/: 140 us against 33 us).  Entries without a candidate carry id ~0 (and distance +inf).
__device__ __forceinline__ uint32_t lex_rank4(const float *cd, const uint32_t *ci, uint32_t n, float di, uint32_t ii) {
    uint32_t rank = 0;
#pragma unroll 4
    for (uint32_t j = 0; j < n; j += 4) {
        const float4 d = *reinterpret_cast<const float4 *>(cd + j);
        const uint4 c = *reinterpret_cast<const uint4 *>(ci + j);
        rank += (c.x != 0xFFFFFFFFu && lex_less(d.x, c.x, di, ii)) ? 1u : 0u;
        rank += (c.y != 0xFFFFFFFFu && lex_less(d.y, c.y, di, ii)) ? 1u : 0u;
        rank += (c.z != 0xFFFFFFFFu && lex_less(d.z, c.z, di, ii)) ? 1u : 0u;
        rank += (c.w != 0xFFFFFFFFu && lex_less(d.w, c.w, di, ii)) ? 1u : 0u;
    }
    return rank;
}
// k-th smallest (dist, id) among the valid entries [0, n) of the LDS lists (wave-collective; +inf if fewer than k are valid)
__device__ __forceinline__ float cert_kth_dist(const float *cd, const uint32_t *ci, uint32_t n, uint32_t k, float *slot, int lane) {
    if (lane == 0) *slot = __builtin_inff();
    __syncthreads();
    for (uint32_t i = lane; i < n; i += kWave) {
        const uint32_t ii = ci[i];
        if (ii == 0xFFFFFFFFu) continue;
        const float di = cd[i];
        if (lex_rank4(cd, ci, n, di, ii) == k - 1) *slot = di;
    }
    __syncthreads();
    return *slot;
}

template <typename RT>
__global__ __launch_bounds__(64) void k_rerank(const RT *__restrict__ rows, uint32_t ld,
                                               uint32_t dim, int metric,
                                               const float *__restrict__ queries,
                                               const uint32_t *__restrict__ cand, uint32_t kc,
                                               uint32_t k, uint64_t id_offset,
                                               uint64_t *__restrict__ out_ids,
                                               float *__restrict__ out_dist,
                                               uint32_t *__restrict__ out_count, uint32_t dim_pad, CertArgs cert) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *q = reinterpret_cast<float *>(smem);
    float *cd = q + dim_pad;
    uint32_t *ci = reinterpret_cast<uint32_t *>(cd + kc);
    uint32_t *ck = ci + kc;  // [kc] screen keys (certificate only)
    float *dk_slot = reinterpret_cast<float *>(ck + kc);
    const int lane = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    const float *qg = queries + (size_t)qi * dim;
    for (uint32_t i = lane; i < dim_pad; i += kWave) q[i] = i < dim ? qg[i] : 0.0f;
    __syncthreads();
    for (uint32_t i = lane; i < kc; i += kWave) {
        ci[i] = cand[(size_t)qi * kc + i];
        if (cert.kth) ck[i] = cert.keys[(size_t)qi * kc + i];
    }
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
    if (cert.kth) {
        const CertQ cq = cert_query_consts(cert, q, dim, metric, lane);
        uint32_t ks = 0xFFFFFFFEu;
        for (uint32_t base = 0; base < kc; base += kWave) {
            const uint32_t i = base + lane;
            const uint32_t id = i < kc ? ci[i] : 0xFFFFFFFFu;
            const bool need = id != 0xFFFFFFFFu && (base == 0 || ck[i] <= ks);
            if (i < kc) {
                cd[i] = need ? exact_dist_rt(metric, q, rows + (size_t)id * ld, dim) : __builtin_inff();
                if (!need) ci[i] = 0xFFFFFFFFu;  // provably outside the top-k: not evaluated, not ranked
            }
            if (__ballot(need) == 0) {  // (sorted lists: nothing further can be needed)
                for (uint32_t j = base + kWave + lane; j < kc; j += kWave) { ci[j] = 0xFFFFFFFFu; cd[j] = __builtin_inff(); }
                break;
            }
            const float dk = cert_kth_dist(cd, ci, min(base + kWave, kc), k, dk_slot, lane);
            ks = cert_safe_key(cert, cq, dk, dim, metric);
        }
        __syncthreads();
        cert_verdict(cert, qi, ks, lane);
    } else {
        for (uint32_t i = lane; i < kc; i += kWave) {
            const uint32_t id = ci[i];
            cd[i] = id == 0xFFFFFFFFu ? __builtin_inff()
                                      : exact_dist_rt(metric, q, rows + (size_t)id * ld, dim);
        }
    }
    __syncthreads();
    uint32_t total = 0;
    for (uint32_t base = 0; base < kc; base += kWave) {
        const uint32_t i = base + lane;
        const bool valid = i < kc && ci[i] != 0xFFFFFFFFu;
        if (valid) {
            const float di = cd[i];
            const uint32_t ii = ci[i];
            uint32_t rank = 0;
            if ((kc & 3u) == 0 && (dim_pad & 3u) == 0) {
                rank = lex_rank4(cd, ci, kc, di, ii);
            } else {
                for (uint32_t j = 0; j < kc; ++j) {
                    const uint32_t ij = ci[j];
                    if (ij != 0xFFFFFFFFu && lex_less(cd[j], ij, di, ii)) ++rank;
                }
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

// K5 (staged): the same exact re-rank for kc = 64 candidates, with the rows brought in by LDS-DMA.
// k_rerank lets every lane walk its own 3-KB row with 16-byte loads: each wave-instruction touches 64 different
// cache lines (201 MB per 1024-query batch came in at 0.9 TB/s).  Here the wave gathers 256-byte chunks of all 64
// rows per step -- `global_load_lds_dwordx4`, 16 lanes per row (4 rows x 2 full lines per instruction), no VGPR
// round trip -- into a [64 rows][256 B] LDS image, double buffered, and every lane then walks ITS row's chunk out of
// LDS.  LDS-DMA writes lane-linear, so the bank swizzle is applied to the SOURCE piece: position p of row r holds
// piece p ^ (r & 15) and lane i reads position k ^ (i & 15) (conflict-free ds_read_b128).  The arithmetic is the
// reference's (hnsw.zig:182-192): one lane, strictly ascending k, sub / mul / add, no contraction.
typedef const __attribute__((address_space(1))) void *rr_gptr_t;
typedef __attribute__((address_space(3))) void *rr_lptr_t;
// LDS-DMA by inline asm (per-lane 64-bit source address, wave-uniform LDS destination through M0).  With the builtin, hipcc's
// waitcnt pass puts s_waitcnt vmcnt(0) in front of EVERY LDS read of the walk below -- it cannot prove the reads disjoint from
// the DMA of the next chunk, which is in flight by design -- so fetch and walk ran strictly one after the other (r2 measured
// "deeper staging changes nothing" for exactly this reason).  The kernel waits for its chunks itself.
__device__ __forceinline__ void rr_dma16(const unsigned char *src, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
}
// KC = 64 / 128 / 256 candidates per query, evaluated in batches of SR rows through the same two stage buffers.  RT = stored row
// element.  CHB = bytes of a row fetched per step (256 / 512 / 1024): one LDS-DMA instruction moves 1 KiB = 1024 / CHB rows x CHB
// contiguous bytes, so a larger CHB reads longer runs of a row at once (a 16-byte piece holds 4 fp32 / 8 bf16 elements).  SR =
// rows staged per batch: 64, or 32 with CHB = 512 (the same 32 KiB of LDS, 4 workgroups per CU) when the lists hold at most 32
// candidates or arrive sorted by screen key with a certificate (bf_cert.h) -- then a batch is evaluated only as far as the bound
// of the batches before it leaves candidates that could still enter the top-k (on the benchmark's data: the first 32 of 128).
// Measured in the headline step (1,024
// queries x 32 candidates x 3 KiB rows; rocprofv3 stats of `bench.py --mode bf16`): round 2's kernel 71.6 us; with the waits and the walk fixed (below) 41.7 us
// (256 B x 64 rows), 40.3 us (512 B x 32 rows), 61.5 us (1 KiB x 32 rows: half the workgroups per CU); four stage buffers with three
// chunks in flight 43.3 us -- what is left is the rate of 100 MB of scattered 256 .. 512-byte pieces (~3 TB/s), not round trips.
template <int KC, typename RT, int CHB = 256, int SR = 64>
__global__ __launch_bounds__(64) void k_rerank_staged(const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
                                                      const float *__restrict__ queries, const uint32_t *__restrict__ cand,
                                                      uint32_t k, uint64_t id_offset, uint64_t *__restrict__ out_ids,
                                                      float *__restrict__ out_dist, uint32_t *__restrict__ out_count,
                                                      uint32_t dim_pad /* multiple of a chunk's elements */, CertArgs cert) {
    constexpr uint32_t kc = KC;
    constexpr int NE = KC / 64;                   // list entries per lane
    constexpr int NB = KC / SR;                   // batches
    constexpr int LPR = CHB / 16;                 // lanes per row of a DMA instruction = 16-byte pieces per chunk
    constexpr int RPI = 64 / LPR;                 // rows per DMA instruction
    constexpr int NINS = SR / RPI;                // DMA instructions per chunk
    constexpr int RR_CH = CHB / (int)sizeof(RT);  // elements per chunk
    constexpr int EPP = 16 / (int)sizeof(RT);     // elements per 16-byte piece
    constexpr uint32_t BUF = (uint32_t)SR * CHB;  // bytes of one stage buffer
    static_assert(SR == 64 || SR == 32, "staged rows per batch");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *stage = smem;                                              // [2][SR rows][CHB]
    float *q = reinterpret_cast<float *>(smem + 2 * BUF);                     // [dim_pad], zero padded
    float *cd = q + dim_pad;
    uint32_t *ci = reinterpret_cast<uint32_t *>(cd + kc);
    uint32_t *ck = ci + kc;  // [kc] screen keys (certificate only)
    float *dk_slot = reinterpret_cast<float *>(ck + kc);
    const int lane = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    const float *qg = queries + (size_t)qi * dim;
    for (uint32_t i = lane; i < dim_pad; i += kWave) q[i] = i < dim ? qg[i] : 0.0f;
#pragma unroll
    for (int bt = 0; bt < NE; ++bt) {
        ci[lane + 64 * bt] = cand[(size_t)qi * kc + lane + 64 * bt];
        cd[lane + 64 * bt] = __builtin_inff();
        if (cert.kth) ck[lane + 64 * bt] = cert.keys[(size_t)qi * kc + lane + 64 * bt];
    }
    __syncthreads();
    {   // drop duplicate ids (keep the first copy); lists sorted by (key, id) -- the certificate's -- hold the copies of a row side by side
        bool dup[NE];
#pragma unroll
        for (int bt = 0; bt < NE; ++bt) {
            dup[bt] = false;
            const int me = lane + 64 * bt;
            const uint32_t ii = ci[me];
            if (ii != 0xFFFFFFFFu) {
                if (cert.kth) dup[bt] = me > 0 && ci[me - 1] == ii;
                else
                    for (int j = 0; j < me; ++j)
                        if (ci[j] == ii) { dup[bt] = true; break; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int bt = 0; bt < NE; ++bt)
            if (dup[bt]) ci[lane + 64 * bt] = 0xFFFFFFFFu;
        __syncthreads();
    }
    CertQ cq{};
    uint32_t ks = 0xFFFFFFFEu;  // largest screen key a row of the top-k can carry, by what has been evaluated so far
    unsigned long long tstamp[6] = {0, 0, 0, 0, 0, 0};
    const bool prof = cert.dbg_rows && cert.dbg_mode == 2 && (qi & 255u) == 0;  // (tuning ZVDB_CERT_DEBUG=2: cycle stamps of every 256th query)
    if (prof) tstamp[0] = __builtin_amdgcn_s_memtime();
    if (cert.kth) cq = cert_query_consts(cert, q, dim, metric, lane);
    if (prof) tstamp[1] = __builtin_amdgcn_s_memtime();
    const uint32_t nch = (ld * (uint32_t)sizeof(RT) + CHB - 1) / CHB;
    const uint32_t row_bytes = ld * (uint32_t)sizeof(RT);
    const uint32_t stage_lds = (uint32_t)(size_t)(rr_lptr_t)stage;
    // With a certificate only the FIRST batch is staged: what the bound leaves of the other candidates (on the benchmark's data ~3 of
    // 96 on average, up to ~70 for the densest neighbourhoods of a batch) is walked straight from global memory, one lane per row,
    // 64 rows at a time (below) -- a staged batch is a chain of dependent chunk fetches whatever it holds, and a 1,024-query launch
    // is ONE round of workgroups, so its time is the slowest query's: three staged batches made it 140 us, profiles/r4/.
    const int nb_staged = cert.kth ? 1 : NB;
#pragma unroll 1
    for (int bt = 0; bt < nb_staged; ++bt) {
        // DMA side: instruction t fills rows RPI t .. RPI t + RPI - 1 of this batch; this lane serves row RPI t + lane / LPR, piece
        // lane % LPR -- of the SOURCE piece (lane % LPR) ^ (row % LPR): the swizzle that makes the walk's reads conflict-free
        // slots past the last candidate of the batch that is needed (lists are compacted to the front: a shard that keeps fewer than
        // 64 survivors leaves a ~0 tail; with a certificate: candidates above the bound) are not fetched at all: whole DMA
        // instructions drop out, wave-uniformly
        uint32_t my_id = lane < SR ? ci[SR * bt + lane] : 0xFFFFFFFFu;
        if (cert.kth && bt > 0 && my_id != 0xFFFFFFFFu && ck[SR * bt + lane] > ks) {  // provably outside the top-k
            my_id = 0xFFFFFFFFu;
            ci[SR * bt + lane] = 0xFFFFFFFFu;
        }
        const unsigned long long vmask = __ballot(my_id != 0xFFFFFFFFu);
        if (vmask == 0) {
            if (!cert.kth) continue;
            // sorted lists: nothing further can be needed
            for (uint32_t j = (uint32_t)SR * (bt + 1) + lane; j < kc; j += kWave) ci[j] = 0xFFFFFFFFu;
            break;
        }
        __syncthreads();  // (the cleared entries are seen by the DMA lanes below)
        const int nvalid = 64 - __builtin_clzll(vmask);
        const int nt_used = min((nvalid + RPI - 1) / RPI, NINS);
        auto issue = [&](uint32_t c, uint32_t buf) {
#pragma unroll 8
            for (int t = 0; t < NINS; ++t) {
                if (t >= nt_used) break;
                const uint32_t r = (uint32_t)RPI * t + ((uint32_t)lane / LPR);
                const uint32_t id = ci[SR * bt + r];
                const unsigned char *src_row = reinterpret_cast<const unsigned char *>(rows + (size_t)(id == 0xFFFFFFFFu ? 0u : id) * ld);
                // the row's last chunk may be short (ld is a multiple of 32 floats): stay inside the row
                const uint32_t off = min(c * (uint32_t)CHB + ((((uint32_t)lane % LPR) ^ (r % LPR)) << 4), row_bytes - 16);
                rr_dma16(src_row + off, (uint32_t)__builtin_amdgcn_readfirstlane((int)(stage_lds + buf * BUF + t * 1024)));
            }
        };
        float sum = 0.0f;
        issue(0, 0);
        for (uint32_t c = 0; c < nch; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk c has landed (single wave: no barrier needed)
            if (c + 1 < nch) issue(c + 1, (c + 1) & 1);
            const unsigned char *mine = stage + (c & 1) * BUF + (lane & (SR - 1)) * CHB;
            const uint32_t k0 = c * RR_CH;
            // Elements [dim, ld) are zero in the row (padding) and in q: they add +0.0, which changes nothing -- so a chunk that lies
            // inside the row is walked whole, without a test per element; only the row's last, short chunk (ld is a multiple of 32
            // fp32 / 64 bf16 elements, not of a chunk) stops at the row's end, a whole piece at a time.  The sum stays the
            // reference's: one element after the other, each operation rounded (this file is built without contraction).
            auto walk = [&](auto l2c, auto fullc) {
                constexpr bool L2 = decltype(l2c)::value, FULL = decltype(fullc)::value;
#pragma unroll
                for (int p = 0; p < LPR; ++p) {
                    const uint32_t kk = k0 + EPP * p;
                    if constexpr (!FULL) { if (kk >= ld) break; }
                    float vv[EPP], aa[EPP];
                    if constexpr (sizeof(RT) == 4) {
                        const float4 v = *reinterpret_cast<const float4 *>(mine + ((p ^ (lane & (LPR - 1))) << 4));
                        vv[0] = v.x; vv[1] = v.y; vv[2] = v.z; vv[3] = v.w;
                    } else {
                        const uint4 w = *reinterpret_cast<const uint4 *>(mine + ((p ^ (lane & (LPR - 1))) << 4));
                        vv[0] = bf16_lo(w.x); vv[1] = bf16_hi(w.x); vv[2] = bf16_lo(w.y); vv[3] = bf16_hi(w.y);
                        vv[4] = bf16_lo(w.z); vv[5] = bf16_hi(w.z); vv[6] = bf16_lo(w.w); vv[7] = bf16_hi(w.w);
                    }
#pragma unroll
                    for (int e4 = 0; e4 < EPP; e4 += 4) {
                        const float4 a = *reinterpret_cast<const float4 *>(q + kk + e4);
                        aa[e4] = a.x; aa[e4 + 1] = a.y; aa[e4 + 2] = a.z; aa[e4 + 3] = a.w;
                    }
#pragma unroll
                    for (int e = 0; e < EPP; ++e) {
                        if constexpr (L2) {
                            float t = aa[e] - vv[e];
                            t = t * t;
                            sum = sum + t;
                        } else {
                            const float t = aa[e] * vv[e];
                            sum = sum + t;
                        }
                    }
                }
            };
            const bool full = k0 + RR_CH <= ld;
            if (metric == ZVDB_METRIC_L2SQ) { if (full) walk(std::true_type{}, std::true_type{}); else walk(std::true_type{}, std::false_type{}); }
            else { if (full) walk(std::false_type{}, std::true_type{}); else walk(std::false_type{}, std::false_type{}); }
        }
        if (lane < SR) cd[SR * bt + lane] = my_id == 0xFFFFFFFFu ? __builtin_inff() : (metric == ZVDB_METRIC_L2SQ ? sum : -sum);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my LDS reads of the stage are done before the next batch refills it
    }
    __syncthreads();
    if (prof) tstamp[2] = __builtin_amdgcn_s_memtime();
    uint32_t n_hi = kc;  // entries [n_hi, kc) hold no evaluated candidate (sorted lists: what is evaluated is, but for holes, a prefix)
    if (cert.kth) {
        n_hi = SR;
        for (uint32_t base = SR; base < kc; base += kWave) {
            const float dk0 = cert_kth_dist(cd, ci, n_hi, k, dk_slot, lane);
            ks = cert_safe_key(cert, cq, dk0, dim, metric);
            const uint32_t i = base + lane;
            const uint32_t id = i < kc ? ci[i] : 0xFFFFFFFFu;
            const bool need = id != 0xFFFFFFFFu && ck[i] <= ks;
            if (i < kc && !need) ci[i] = 0xFFFFFFFFu;  // provably outside the top-k: not evaluated, not ranked
            if (__ballot(need) == 0) {  // sorted lists: nothing further can be needed
                for (uint32_t j = base + kWave + lane; j < kc; j += kWave) ci[j] = 0xFFFFFFFFu;
                break;
            }
            if (need) cd[i] = exact_dist_rt(metric, q, rows + (size_t)id * ld, dim);
            const unsigned long long nm = __ballot(need);
            n_hi = min(kc, (base + 64u - (uint32_t)__builtin_clzll(nm) + 3u) & ~3u);
            __syncthreads();
        }
        __syncthreads();
        if (prof) tstamp[3] = __builtin_amdgcn_s_memtime();
        const float dk = cert_kth_dist(cd, ci, n_hi, k, dk_slot, lane);
        ks = cert_safe_key(cert, cq, dk, dim, metric);
        if (prof) tstamp[4] = __builtin_amdgcn_s_memtime();
        if (cert.dbg_rows && cert.dbg_mode == 3) {
            // self-check of the inequality on the rows this query DID evaluate: a row with exact distance d carries a screen key of at
            // most key_safe(d) -- the statement the certificate applies to the rows it did not evaluate.  [6] += checked, [7] += violations
            uint32_t chk = 0, bad = 0;
#pragma unroll
            for (int bt = 0; bt < NE; ++bt) {
                const uint32_t i = 64 * bt + lane;
                if (ci[i] != 0xFFFFFFFFu) {
                    chk += 1;
                    bad += ck[i] > cert_safe_key(cert, cq, cd[i], dim, metric) ? 1u : 0u;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { chk += __shfl_xor(chk, off); bad += __shfl_xor(bad, off); }
            if (lane == 0) {
                __hip_atomic_fetch_add(cert.dbg_rows + 6, chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_fetch_add(cert.dbg_rows + 7, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        uint32_t nev = 0;
        if (cert.dbg_rows && cert.dbg_mode == 1) {
#pragma unroll
            for (int bt = 0; bt < NE; ++bt) nev += (uint32_t)__popcll(__ballot(ci[64 * bt + lane] != 0xFFFFFFFFu));
        }
        cert_verdict(cert, qi, ks, lane, nev);
    }
    uint32_t total = 0;
#pragma unroll
    for (int bt = 0; bt < NE; ++bt) {
        const uint32_t my_id = ci[64 * bt + lane];
        const bool valid = my_id != 0xFFFFFFFFu;
        if (valid) {
            const float di = cd[64 * bt + lane];
            const uint32_t rank = lex_rank4(cd, ci, n_hi, di, my_id);
            if (rank < k) {
                out_ids[(size_t)qi * k + rank] = (uint64_t)my_id + id_offset;
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
    if (prof && lane == 0) {
        tstamp[5] = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 5; ++i)
            __hip_atomic_fetch_add(cert.dbg_rows + 1 + i, (uint32_t)(tstamp[i + 1] - tstamp[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int launch_rerank(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, const uint32_t *d_cand,
                  uint32_t kc, uint32_t k, uint64_t id_offset, uint64_t *d_ids, float *d_dist,
                  uint32_t *d_count, hipStream_t s, uint32_t kp_valid, const CertArgs *cert_in) {
    if (nq == 0) return ZVDB_OK;
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    const bool staged_off = [] { const char *e = tuning("ZVDB_RERANK"); return e && e[0] == 'v' && e[1] == '1'; }();
    const CertArgs cert = cert_in ? *cert_in : CertArgs{};
    return rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = std::remove_cv_t<std::remove_pointer_t<decltype(rows)>>;
        if ((kc == 64 || kc == 128 || kc == 256) && ld * sizeof(RT) >= 256 && !staged_off) {  // rows gathered through LDS-DMA (see k_rerank_staged)
            auto go_geo = [&](auto kern, uint32_t chb, uint32_t sr) -> int {
                const uint32_t dp = round_up(h->dim, chb / (uint32_t)sizeof(RT));  // q is read a whole chunk at a time
                const size_t lds2 = 2ull * sr * chb + 4ull * (dp + 3ull * kc) + 16;
                ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(kern), lds2));
                hipLaunchKernelGGL(kern, dim3(nq), dim3(64), lds2, s, rows, ld, h->dim, metric, d_queries, d_cand, k, id_offset, d_ids, d_dist,
                                   d_count, dp, cert);
                ZV_HIP(hipGetLastError());
                return ZVDB_OK;
            };
            auto go = [&](auto kern) -> int { return go_geo(kern, 256, 64); };
            // lists of at most 32 candidates (the caller says so: k <= 16 on the exact screens), or sorted lists with a certificate
            // (evaluated only as far as the bound requires): 32 staged rows, longer runs per row
            const bool sr32 = ld * sizeof(RT) >= 1024 && ((kc == 64 && kp_valid && kp_valid <= 32) || (cert.kth && k <= 24));
            if (sr32 && kc == 64) return go_geo(k_rerank_staged<64, RT, 512, 32>, 512, 32);
            if (sr32 && kc == 128) return go_geo(k_rerank_staged<128, RT, 512, 32>, 512, 32);
            if (sr32 && kc == 256) return go_geo(k_rerank_staged<256, RT, 512, 32>, 512, 32);
            if (kc == 64) return go(k_rerank_staged<64, RT>);
            if (kc == 128) return go(k_rerank_staged<128, RT>);
            return go(k_rerank_staged<256, RT>);
        }
        const uint32_t dim_pad = round_up(std::max(h->dim, 1u), 4);
        const size_t lds = 4ull * (dim_pad + 3ull * kc) + 16;
        ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(k_rerank<RT>), lds));
        hipLaunchKernelGGL(k_rerank<RT>, dim3(nq), dim3(64), lds, s, rows, ld, h->dim, metric, d_queries, d_cand, kc, k, id_offset, d_ids, d_dist,
                           d_count, dim_pad, cert);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    });
}

}  // namespace zvdb
