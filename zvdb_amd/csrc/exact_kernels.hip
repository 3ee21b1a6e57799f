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
// (the ef-mode beam search and the entry descent: ef_search.hip; the exact re-rank of brute-force candidates: rerank.hip; the
// visited sets they share: lds_state.h)
#include <cstring>
#include "common.h"
#include "wave_select.h"
#include "exact_dist.h"
#include "lds_state.h"

#pragma clang fp contract(off)

namespace zvdb {


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
            float md;  // wave arg-min over (d, j) of the active lanes
            uint32_t mj, mid;
            wave_argmin_dj(act, d, j, nb, md, mj, mid);
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
            float md;  // wave arg-min over (d, j) of the active lanes
            uint32_t mj, mid;
            wave_argmin_dj(act, d, j, nb, md, mj, mid);
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
                    float md;  // wave arg-min over (d, j) of the active lanes
                    uint32_t mj, mid;
                    wave_argmin_dj(act, d, j, nb, md, mj, mid);
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
                float md;  // wave arg-min over (d, j) of the active lanes
                uint32_t mj, mid;
                wave_argmin_dj(act, d, j, nb, md, mj, mid);
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
template <typename RT, bool LAT = false>
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
            // pop: the minimum and how many entries hold it.  LAT (the one-query-per-call kernel, one wave on the device): a DPP reduction
            // of (order key, position) and ballots -- the __shfl_xor butterflies are 18 dependent ds_bpermute round trips, 2,700 shader
            // cycles of a 10,000-cycle pop (1,270 on DPP).  The batched kernel keeps the butterflies: with thousands of waves the round
            // trips are hidden and the DPP form's extra vector instructions cost it 4-7 % (A/B on one box, 100k x 128 / 768, r4)
            float bd = __builtin_inff();
            uint32_t bp = 0xFFFFFFFFu, same = 0;
            if constexpr (LAT) {
                // ONE pass over the frontier: every lane keeps the two smallest (key, position) words of its entries; the minimum of the
                // firsts is the pop, and it is unique iff no other first and no second carries its key (a lane's third entry cannot tie
                // unless its second does).  Equal order keys = equal distances (-0 = +0); a NaN minimum gives the query up as before.
                uint64_t k1 = ~0ull, k2 = ~0ull;
                for (uint32_t i = lane; i < n; i += kWave) {
                    const uint64_t key = ((uint64_t)dist_order_key(fr_d[i]) << 32) | i;
                    if (key < k1) { k2 = k1; k1 = key; }
                    else if (key < k2) k2 = key;
                }
                const uint64_t bk = wave_min64(k1);
                const uint32_t bh = (uint32_t)(bk >> 32);
                bp = (uint32_t)bk;
                const bool tie = (k1 != bk && (uint32_t)(k1 >> 32) == bh) || (k2 != ~0ull && (uint32_t)(k2 >> 32) == bh);
                same = (__ballot(tie) != 0 || bh == 0xFFFFFFFFu) ? 2u : 1u;
            } else {
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
                for (uint32_t i = lane; i < n; i += kWave) same += fr_d[i] == bd ? 1u : 0u;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) same += __shfl_xor(same, off);
            }
            if (same != 1) { gave_up = true; break; }  // equal minima (or a NaN): the heap's layout decides -> exact emulation
            const uint32_t cid = fr_i[bp];
            if constexpr (LAT) bd = fr_d[bp];  // (only lane 0 uses it: the load overlaps the list request below)
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
            // (the degree and the first 64 list slots are requested together: both depend on cid only.  r5, measured and removed: every
            // frontier entry carrying its node's list -- requested when the node enters the frontier, beside its row, so that a pop
            // reads its list from LDS: 33 -> 46 us per call (profiles/r5/per_call_list_prefetch_ab.txt).  A wave's loads retire in
            // issue order, so the 16 cold list requests of a step sit in front of the row walk's loads; the lists of the nodes
            // that ARE popped -- the entry point's neighbourhood, the same for every query -- were L2 hits all along.)
            const uint32_t nb0 = (uint32_t)lane < gstride ? adj[(size_t)cid * gstride + lane] : 0u;
            const uint32_t dg = deg[cid];                                   // :216 layer 0
            for (uint32_t base = 0; base < dg; base += kWave) {
                const uint32_t j = base + lane;
                bool fresh = false;
                uint32_t nb = 0;
                float d = 0.0f;
                if (j < dg) {
                    nb = base == 0 ? nb0 : adj[(size_t)cid * gstride + j];
                    // :217 + :221 in one test-and-set (one owner per id; marking the node before its distance instead of after changes nothing
                    // another lane could see: a second occurrence of the id in the list loses either way)
                    fresh = hash_insert_atomic(hash, hash_bits, nb);
                    if (fresh) d = exact_dist_rt(metric, q, rows + (size_t)nb * ld, dim);  // :219
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
        done = compat_fast_body<RT, true>(rows, ld, dim, metric, adj, deg, gstride, N, queries, k, entry, 0ull, out_ids, out_dist, out_count, stats, dim_pad,
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

}  // namespace zvdb
