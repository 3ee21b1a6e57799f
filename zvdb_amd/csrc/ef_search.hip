// ef_search.hip -- the ef-mode beam search over the device graph (extension; specification: oracle/zvdb_oracle.h,
// orc_hnsw_search_ef_ex / orc_hnsw_search_ef_w) and the hierarchical entry descent of the bulk graph (orc_hnsw_descend):
//   k_search_ef_reg     beam in registers (ef <= 1,024), 1 / 2 / 4 / 8 waves per query, search width, two-pass visited sizing
//   k_search_ef         beam in LDS (beyond, and the retry / net passes with visited sets in global memory)
//   k_descend           entry descent through the upper layers
// Traversal distances are the oracle's 16-way blocked sums, un-contracted (-ffp-contract=off + the pragma below); the final k are
// re-evaluated with the reference's sequential arithmetic (exact_dist.h).
#include <cstring>
#include <string>
#include "common.h"
#include "wave_select.h"
#include "exact_dist.h"
#include "lds_state.h"

#pragma clang fp contract(off)

namespace zvdb {

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
                const float sl = group_sum_lane0<8>(slo[u]), sh = group_sum_lane0<8>(shi[u]);  // (the butterfly 4, 2, 1 as the group's lane 0 sees it)
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
                const float sv = group_sum_lane0<16>(sacc[u]);  // (the butterfly 8, 4, 2, 1 as the group's lane 0 sees it)
                if (fr[u] && t == 0) nb_d[r0 + (uint32_t)(u * WPQ + wave) * G + g] = metric == ZVDB_METRIC_L2SQ ? sv : -sv;
            }
        }
    }
}

// ---- hierarchical entry descent of the bulk graph: the per-query body (k_descend and the first pass of k_search_ef_reg) --------
// (specification: orc_hnsw_descend; the kernel's comment further down describes the algorithm)
// the descent of one query by the WPQ waves of its workgroup (every wave runs the same control flow on the same LDS / global values);
// q is in LDS already; returns the layer-0 entry, adds the distance evaluations to `evals`
template <typename RT, int WPQ, int U>
__device__ __forceinline__ uint32_t descend_body(const RT *__restrict__ rows, uint32_t ld, int metric, const uint32_t *__restrict__ lvl,
                                                 const uint32_t *__restrict__ slot, const uint32_t *__restrict__ adj_up,
                                                 const uint32_t *__restrict__ deg_up, uint32_t gstride, uint32_t top, const float *q, uint32_t nchunks,
                                                 float *nb_d, uint32_t *nb_i, int lane, int wave, unsigned long long &evals) {
    const int t = lane & 15;
    // every wave keeps the same (cur, dcur): all decisions below derive from LDS / global values the waves share
    uint32_t cur = top;
    float dcur = metric == ZVDB_METRIC_L2SQ ? blocked16_dist<ZVDB_METRIC_L2SQ>(q, rows + (size_t)cur * ld, nchunks, t)
                                            : blocked16_dist<ZVDB_METRIC_DOT>(q, rows + (size_t)cur * ld, nchunks, t);
    evals += 1;
    // A hop is a link of the query's chain.  The first version read lvl[cur] -> slot[cur] -> deg_up[row] -> the list -> the rows: five
    // dependent memory round trips, 3.75 us per hop, 150 us per 1,024-query batch at 1M rows -- a fifth of an ef = 128 search (r4).
    // Now the degree and the first 64 list entries are requested TOGETHER, and the level / slot of every neighbour travel with its
    // row (the next hop starts from one of them): list -> rows, two round trips per hop.  Same hops, same evaluations, same entry.
    uint32_t cur_lvl = lvl[top], cur_slot = slot[top];
    for (uint32_t layer = cur_lvl; layer >= 1; --layer) {
        for (;;) {
            if (cur_lvl < layer) break;
            const size_t row = (size_t)cur_slot + layer - 1;
            const uint32_t first = (uint32_t)lane < gstride ? adj_up[row * gstride + lane] : 0u;  // (slots past the degree: stale or zero ids, not used)
            const uint32_t dg = min(deg_up[row], gstride);
            uint32_t best = cur, best_lvl = cur_lvl, best_slot = cur_slot;
            float dbest = dcur;
            for (uint32_t base = 0; base < dg; base += kWave) {
                const uint32_t cnt = min(dg - base, (uint32_t)kWave);
                __syncthreads();  // (the previous window's readers are done)
                uint32_t nbv = 0, nl = 0, ns = 0;
                if ((uint32_t)lane < cnt) {
                    nbv = base == 0 ? first : adj_up[row * gstride + base + lane];
                    nl = lvl[nbv];
                    ns = slot[nbv];
                    if (wave == 0) nb_i[lane] = nbv;
                }
                __syncthreads();
                ef_window_distances<RT, WPQ, U>(rows, ld, q, nchunks, metric, cnt, nb_i, nb_d, lane, wave);
                __syncthreads();
                // list order: the smallest (distance, position); strict '<' against the running best keeps the earliest of equal minima
                const uint64_t key = (uint32_t)lane < cnt ? ((uint64_t)f32_key(nb_d[lane]) << 32) | (uint32_t)lane : ~0ull;
                const uint32_t jm = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wave_min64(key));
                const float dm = nb_d[jm];
                if (dm < dbest) {
                    dbest = dm;
                    best = (uint32_t)__builtin_amdgcn_readlane((int)nbv, (int)jm);
                    best_lvl = (uint32_t)__builtin_amdgcn_readlane((int)nl, (int)jm);
                    best_slot = (uint32_t)__builtin_amdgcn_readlane((int)ns, (int)jm);
                }
            }
            evals += dg;
            if (best == cur) break;
            cur = best;
            dcur = dbest;
            cur_lvl = best_lvl;
            cur_slot = best_slot;
        }
    }
    __syncthreads();  // (the last window's readers are done before the caller reuses nb_i / nb_d)
    return cur;
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
// The entry descent fused into the first pass of the register-beam kernel (r4): as two launches a batch paid the slowest query's
// descent plus the slowest query's search; in one kernel a query starts its search when ITS descent ends.  on = 0: the entry comes
// from the kernel's `entry` / `entries` arguments as before (retry passes read what the first pass left in entries_out).
struct EfDescend {
    const uint32_t *lvl, *slot, *adj_up, *deg_up;
    uint32_t *entries_out;
    uint32_t top, on;
};
__device__ __forceinline__ void ef_mark_again(uint32_t *out_count, uint32_t qi, uint32_t *net_list) {  // (one lane)
    out_count[qi] = kEfAgain;
    if (net_list) net_list[3 + atomicAdd(net_list, 1u)] = qi;
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
    uint32_t *net_list /* LDS passes: marks are appended here (may be null); GH: non-null = the net launch */, uint32_t width /* nodes expanded per step */,
    EfDescend /* (k_search_ef_reg's: this kernel always takes its entries from `entries`) */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (GH) {
        uint32_t *table = ghash + ((size_t)blockIdx.x << hash_bits);
        if (net_list) {
            const uint32_t total = __hip_atomic_load(net_list, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                uint32_t i = 0;
                if (threadIdx.x == 0) i = atomicAdd(net_list + 1, 1u);
                i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
                if (i >= total) {
                    // the list is empty: the LAST workgroup to find it so leaves the three header words zero for the next call (every other
                    // workgroup has read its final cursor by then; the next call's kernels start behind this one) -- no zeroing launch per call
                    if (threadIdx.x == 0 && atomicAdd(net_list + 2, 1u) == gridDim.x - 1) {
                        __hip_atomic_store(net_list, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(net_list + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(net_list + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    return;
                }
                const uint32_t qi = net_list[3 + i];
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
constexpr uint32_t kEfFinal = 0xFFFFFFFEu;  // ctl[0]: the search is over -- share the exact re-evaluation of the final k (ctl[1] results), then leave
template <typename RT, int EPL, int WPQ = 1, int PROF = 0, bool WIDE = false>
#ifndef ZVDB_EF_WPE2
#define ZVDB_EF_WPE2 2   /* waves per SIMD the two-wave workgroups are compiled for.  (EPL <= 4 ? 3 : 2) -- a cap of 168 VGPRs, six resident queries per CU instead of four at EPL 4 -- measured 0 .. 6 % SLOWER at batch 1,024 and 4,096 (profiles/r5/ef_vgpr_cap_ab.txt) */
#endif
__global__ __launch_bounds__(64 * WPQ, WPQ == 8 ? 1 : WPQ == 4 ? (EPL >= 8 ? 2 : 4) : WPQ == 2 ? ZVDB_EF_WPE2 : 1) void k_search_ef_reg(
    const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
    const uint32_t *__restrict__ adj, const uint32_t *__restrict__ deg, uint32_t gstride,
    uint32_t N, const float *__restrict__ queries, uint32_t k, uint32_t ef, uint32_t entry,
    uint64_t id_offset, uint64_t *__restrict__ out_ids, float *__restrict__ out_dist,
    uint32_t *__restrict__ out_count, unsigned long long *__restrict__ stats, uint32_t dim_pad,
    uint32_t hash_size /* slots of the visited table: any count >= 256 */, uint32_t kpad, const uint32_t *__restrict__ entries /* optional: per-query entry (k_descend) */,
    uint32_t pass, uint32_t *__restrict__ fill_hint, uint32_t * /* ghash: k_search_ef only */, uint32_t /* qoff */,
    uint32_t *net_list /* marks are appended here (the last LDS pass of a call; else null) */,
    uint32_t width_mode /* nodes expanded per step: 1 .. 4; large beams (EPL >= 8) also: r << 8 | b << 16 | packed visited table << 24 (hp_insert_atomic) */,
    EfDescend desc /* on: this (first) pass runs the entry descent itself */) {
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
    float *nb_d = reinterpret_cast<float *>(hash + hash_size);   // [2 x 64] (a search-width step evaluates both windows' fresh entries in one distance phase)
    uint32_t *nb_i = reinterpret_cast<uint32_t *>(nb_d + 2 * kWave);
    uint32_t *R = nb_i + 2 * kWave;                      // [kpad] result ids
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
    constexpr int U = sizeof(RT) == 2 ? (WPQ == 1 ? 2 : 1) : (WPQ >= 4 ? 1 : 4 / WPQ);  // rows per lane group and step: 16 rows per step (bf16 rows, WPQ = 4: 32; two rows per group at WPQ = 2 measured 24 % SLOWER with a search width, r4; r6, with one distance phase per step: 25 - 92 spilled registers, -24 .. -42 %)
    if (desc.on) {  // the entry descent of this query, by all waves of the workgroup (wave-uniform branch); later passes find the entry in entries_out
        unsigned long long dev = 0;
        entry = descend_body<RT, WPQ, U>(rows, ld, metric, desc.lvl, desc.slot, desc.adj_up, desc.deg_up, gstride, desc.top, q, nchunks, nb_d, nb_i, lane, wave, dev);
        if (threadIdx.x == 0) {
            desc.entries_out[qi] = entry;
            if (stats) atomicAdd(&stats[0], dev);
        }
    }
    if constexpr (WPQ > 1) {
        if (wave != 0) {  // helper waves: their share of every published window's distances
            for (;;) {
                __syncthreads();  // (A) window published
                const uint32_t nfh = ctl[0];
                if (nfh == kEfDone) return;
                if (nfh == kEfFinal) {  // the final k: this wave's share of the exact re-evaluation (below), then out
                    const uint32_t nresh = ctl[1];
                    for (uint32_t i = (uint32_t)lane * WPQ + (uint32_t)wave; i < nresh; i += kWave * WPQ)
                        RD[i] = exact_dist_rt(metric, q, rows + (size_t)R[i] * ld, dim);
                    __syncthreads();  // (B)
                    return;
                }
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
                // beam update with one window's candidates (ck: one key per lane, ~0 = none): the best ef of (W u candidates) stay
                auto merge_cands = [&](uint64_t ck) {
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
                };
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
                    merge_cands(ck);
                    return true;
                };
                const uint32_t per = (uint32_t)kWave / max(gstride, 1u);  // whole lists that fit one window
                if (nsel > 1 && per > 1) {
                    // the lists of the picked nodes side by side in a window (lane = list w, entry j), `per` lists per window: that many
                    // lists' worth of row fetches per dependent step.  The beam is a set, so merging the candidates of several lists
                    // together equals the spec's list-by-list insertion.
                    pre_id = 0xFFFFFFFFu;
                    // (r6) ALL the step's lists are requested before its first window runs: the picks are known (and marked) up front, so
                    // the second window's list fetch -- a dependent random read that used to sit between the first window's merge and its own
                    // visited pass -- is in flight during the first window's distance phase.  At most two windows (per >= 2, nsel <= 4).
                    // (Measured and dropped, profiles/r6/ef_fused_step_ab.txt: requesting the NEXT step's likely lists ahead of the distance phase
                    // as well -- the width-1 loop's speculation -- -6 %; widths up to 8 (four windows per phase): nothing at batch 1,024, -8 % at
                    // batch 4,096 for the extra LDS.)
                    const uint32_t w = (uint32_t)lane / gstride, j = (uint32_t)lane - w * gstride;
                    uint32_t nblv[2] = {0, 0}, dgv[2] = {0, 0};
                    const uint32_t nwin = (nsel + per - 1) / per;
#pragma unroll
                    for (uint32_t wi = 0; wi < 2; ++wi) {
                        const uint32_t s0 = wi * per;
                        if (s0 >= nsel) break;
                        const uint32_t nl = min(per, nsel - s0);
                        const bool live = w < nl;
                        const uint32_t iw = s0 + w;
                        const uint32_t cw = iw == 0 ? cids[0] : iw == 1 ? cids[1] : iw == 2 ? cids[2] : cids[3];
                        if (live) { nblv[wi] = adj[(size_t)cw * gstride + j]; dgv[wi] = min(deg[cw], gstride); }
                    }
                    // ONE distance phase per step: both windows' lists go through the visited test-and-set first (it does not depend on the
                    // beam), their fresh entries sit side by side in nb_i, one phase evaluates them all (one ramp-up of the row gathers and
                    // one barrier pair with the helper waves per step instead of two; 16-row passes are filled across the windows), then
                    // the windows' candidates merge in list order -- window 1's acceptance test sees the beam window 0's merge left, as in
                    // the spec's list-by-list insertion.  E / P and results unchanged (tests: the oracle's orc_hnsw_search_ef_w).
                    uint32_t nfw[2] = {0, 0}, tot = 0;
#pragma unroll
                    for (uint32_t wi = 0; wi < 2; ++wi) {
                        if (wi >= nwin) break;
                        const uint32_t nl = min(per, nsel - wi * per);
                        const bool valid = w < nl && j < dgv[wi];
                        const uint32_t cnt = (uint32_t)__popcll(__ballot(valid));
                        if (cnt == 0) continue;
                        sync_w();
                        if (hash_used + cnt >= hash_limit) overflow = true;  // wave-uniform
                        if (overflow && (pass == kEfTight || pass == kEfRetryMark)) {  // the launch behind this one serves the query
                            if (lane == 0) ef_mark_again(out_count, qi, net_list);
                            release_helpers();
                            return;
                        }
                        bool fresh1 = false, full1 = false;
                        if (valid && !overflow) fresh1 = visit(nblv[wi], full1);
                        if constexpr (EPL >= 8) {
                            if (__ballot(full1)) {  // a region of the packed table is full: the launch behind this one serves the query
                                if (pass == kEfTight || pass == kEfRetryMark) {
                                    if (lane == 0) ef_mark_again(out_count, qi, net_list);
                                    release_helpers();
                                    return;
                                }
                                overflow = true;
                            }
                        }
                        const unsigned long long fb0 = __ballot(fresh1);
                        if (fresh1) nb_i[tot + (uint32_t)__popcll(fb0 & ((1ull << lane) - 1ull))] = nblv[wi];
                        const uint32_t nf = (uint32_t)__popcll(fb0);
                        nfw[wi] = nf;
                        tot += nf;
                        hash_used += nf;
                        evals += nf;
                    }
                    if (WPQ > 1 && tot != 0) {
                        if (lane == 0) ctl[0] = tot;
                        __syncthreads();  // (A) the step's entries published to the helper waves
                    } else {
                        sync_w();
                    }
                    stamp(1);  // list loads + visited test-and-set
                    if (tot != 0) {
                        ef_window_distances<RT, WPQ, U>(rows, ld, q, nchunks, metric, tot, nb_i, nb_d, lane, 0);
                        __syncthreads();  // (B) distances done
                        stamp(2);  // blocked distances of the fresh neighbours
                        uint32_t off = 0;
#pragma unroll
                        for (uint32_t wi = 0; wi < 2; ++wi) {
                            if (wi >= nwin) break;
                            const uint32_t nf = nfw[wi];
                            if (nf == 0) continue;
                            uint64_t ck = ~0ull;
                            if ((uint32_t)lane < nf) ck = ((uint64_t)f32_key(nb_d[off + lane]) << 32) | nb_i[off + lane];
                            merge_cands(ck);
                            off += nf;
                        }
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
    stamp(4);
    // the first k slots of W, re-evaluated with the exact reference arithmetic, ordered by (exact distance, id).  One lane walks one
    // row in the reference's order; the helper waves take their share of the rows before they leave (k = 100 on 3-KB rows was one wave
    // walking two rounds of rows: 0.1 ms of a 2.4-ms search, r3)
    const uint32_t nres = wn < k ? wn : k;
    if (nres > 0) {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            if ((uint32_t)(e * kWave + lane) < nres) R[e * kWave + lane] = (uint32_t)wkr[e];
    }
    // (up to 32 results stay with wave 0: for k = 10 the hand-over's barrier costs what the split saves -- A/B on one box; k = 100: +1-3 %)
    const bool share = WPQ > 1 && nres > 32;  // wave-uniform
    if constexpr (WPQ > 1) {
        if (lane == 0) { ctl[0] = share ? kEfFinal : kEfDone; ctl[1] = nres; }
        __syncthreads();  // (A) R and the hand-over are published
    } else {
        sync_w();
    }
    if (nres > 0) {
        const uint32_t step = share ? (uint32_t)WPQ : 1u;
        for (uint32_t i = (uint32_t)lane * step; i < nres; i += kWave * step)  // (wave 0's share, or all)
            RD[i] = exact_dist_rt(metric, q, rows + (size_t)R[i] * ld, dim);
        if (share) __syncthreads();  // (B) every share is in RD
        else sync_w();
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
    const uint32_t qi = blockIdx.x;
    const float *qg = queries + (size_t)qi * dim;
    for (uint32_t i = threadIdx.x; i < dim_pad; i += kWave * WPQ) q[i] = i < dim ? qg[i] : 0.0f;
    __syncthreads();
    const uint32_t nchunks = (dim + 3) / 4;
    constexpr int U = sizeof(RT) == 2 ? (WPQ == 1 ? 2 : 1) : 4 / WPQ;  // 16 rows per step
    unsigned long long evals = 0;
    const uint32_t cur = descend_body<RT, WPQ, U>(rows, ld, metric, lvl, slot, adj_up, deg_up, gstride, top, q, nchunks, nb_d, nb_i, lane, wave, evals);
    if (threadIdx.x == 0) {
        out_entry[qi] = cur;
        if (stats) atomicAdd(&stats[0], evals);
    }
}

// A few words zeroed by a launch of our own instead of an 8-byte hipMemsetAsync (a blit kernel of the runtime's: the same 4 us, one library less in the path).
__global__ void k_zero_words(uint32_t *p, uint32_t n) {
    if (threadIdx.x < n) p[threadIdx.x] = 0;
}
static int zero_words(uint32_t *p, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, s, p, n);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// the net's header words are zero when a call begins: the previous call's net launch left them so -- unless that call failed between its
// launches (dirty): then they are zeroed here
static int ef_net_begin(const zvdb_hip_index *h, hipStream_t s) {
    if (h->ef_net_dirty) ZV_CHECK(zero_words(h->ef_net_buf, 3, s));
    h->ef_net_dirty = true;
    return ZVDB_OK;
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
    // (a search on the SAME stream as the last one is ordered behind it by the stream itself: no barrier packet for the stream's own event.
    //  Measured with the change above: ef = 128 kernel-timed 0.671 -> 0.682 of HBM at batch 1,024, one box; the ~20-us hole rocprofv3 shows in
    //  front of every call is the profiler's: host-timed calls are 13 us longer than their kernels, 8.5 of them the two small launches)
    if (h->ef_event_set && h->ef_event_stream != s) ZV_HIP(hipStreamWaitEvent(s, h->ef_event, 0));
    return ZVDB_OK;
}
static int ef_scratch_end(const zvdb_hip_index *h, hipStream_t s) {
    ZV_HIP(hipEventRecord(h->ef_event, s));
    h->ef_event_set = true;
    h->ef_event_stream = s;
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
// the net's buffer: [list: 3 header words (count, pop cursor, workgroups done) + nq, padded to 256 B][kEfNetBlocks tables of 4 << bits bytes] -- 4 MiB at ef <= 256 / stride
// 16, 32 MiB at ef = 1024 / stride 32 (before: 256 tables, up to 1 GiB, taken by the first search whether or not it needed them)
static int ef_reserve_net(const zvdb_hip_index *h, uint32_t nq, uint32_t ef) {
    const size_t list_words = round_up64((uint64_t)nq + 3, 64);
    const size_t table_words = (size_t)kEfNetBlocks << ef_net_bits(h, ef);
    if (h->ef_net_list_words >= list_words && h->ef_net_table_words >= table_words) return ZVDB_OK;
    const size_t lw = std::max(list_words, h->ef_net_list_words), tw = std::max(table_words, h->ef_net_table_words);
    if (h->ef_net_buf) { ZV_HIP(hipDeviceSynchronize()); (void)hipFree(h->ef_net_buf); h->ef_net_buf = nullptr; h->ef_net_list_words = h->ef_net_table_words = 0; }
    ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->ef_net_buf), (lw + tw) * sizeof(uint32_t)));
    ZV_HIP(hipMemset(h->ef_net_buf, 0, 16));   // the header words: zero from here on between calls (the net launch leaves them so)
    h->ef_net_dirty = false;
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
    if (wq != 1 && wq != 2 && wq != 4 && wq != 8) wq = 1;   // (r5: THREE waves -- 12 per CU at 168 VGPRs, no spills -- measured 0 .. 19 % slower than two on bf16 rows, profiles/r5/ef_three_waves_ab.txt)
    if (epl < 8 && wq == 8) wq = 4;                       // (not instantiated)
    if (epl >= 8 && width <= 1 && wq >= 4) wq = 2;        // the wide workgroups exist for the search-width loop only
    if (epl >= 8 && width > 1 && !forced) wq = lds_g == 1 ? 8u : lds_g == 2 ? 4u : 2u;  // (lds_g: queries per CU of the two-wave plan)
    return wq;
}

int launch_search_ef(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t k,
                     uint32_t ef, uint32_t entry, uint64_t id_offset, uint64_t *d_ids,
                     float *d_dist, uint32_t *d_count, unsigned long long *d_stats, hipStream_t s, uint32_t *d_entries,
                     bool reserve_only, bool descend) {
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
    // (switch values that are consulted again further down are COPIED here: tuning() hands out pointers into a small per-thread ring
    // that later lookups overwrite)
    std::string beam_s, gv_s, tp_s;
    auto tuning_copy = [](const char *name, std::string &hold) -> const char * {
        const char *e = tuning(name);
        if (!e) return nullptr;
        hold = e;
        return hold.c_str();
    };
    const char *beam_env = tuning_copy("ZVDB_EF_BEAM", beam_s);  // (read per call: the parity test switches it)
    const bool beam_lds = beam_env && beam_env[0] == 'l';
    const int epl = beam_lds ? 0 : ef <= 128 ? 2 : ef <= 256 ? 4 : ef <= 512 ? 8 : ef <= 1024 ? 16 : 0;
    auto lds_slots = [&](uint64_t slots) -> size_t {  // LDS of a query with a visited table of `slots` entries
        if (epl) return 4ull * dim_pad + 12ull * kWave * epl + 4ull * slots + 16ull * kWave + 8ull * kpad + 16 + (epl >= 8 ? 8ull * kWave + 8ull * epl : 0);  // (+ Cs, Mb of merge_lds)
        return 4ull * dim_pad + 2ull * efp * 8 + 2ull * efp * 4 + 1024 + 4ull * slots + 8ull * kWave;
    };
    auto lds_bytes = [&](uint32_t hb) -> size_t { return lds_slots(1ull << hb); };
    // A safe table that leaves fewer than two workgroups per CU (ef * stride beyond ~12k) does not go to the LDS at all: the safe
    // pass then keeps the visited sets in GLOBAL memory (k_search_ef<GH>: a pre-zeroed table of >= 8 ef stride slots per query in
    // an index-owned buffer, the batch in chunks of <= 2 GiB) -- slower per pop, but it serves only what the tight LDS pass
    // handed back (or the first batch of an ef bucket, which establishes the fill), and nothing overflows: before, such searches
    // ran one wave per CU on a table clamped to the LDS and failed with "visited set exhausted" beyond it.
    const char *gv_env = tuning_copy("ZVDB_EF_GLOBAL_VISITED", gv_s);  // "1" / "0": force / forbid the global safe pass (tests; read per call)
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
    const char *tp_env = tuning_copy("ZVDB_EF_TWO_PASS", tp_s);  // "0": always the safe table; "t<bits>": force the tight table (tests; read per call)
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
        const uint32_t ncu = std::max(h->cu_count, 1u);  // (of the index's own device, read at create)
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
        // the entry descent (descend: `entry` is the graph's top node, d_entries the per-query entry buffer): inside the FIRST pass when
        // that pass is the register-beam kernel's (EfDescend), else as its own launch ahead of the search
        const char *fd_env = tuning("ZVDB_EF_FUSE_DESCENT");  // "0": the descent as its own launch (A/B, tests)
        const bool fuse_descent = descend && epl != 0 && (two_pass || !safe_global) && !prof && !reserve_only && !(fd_env && fd_env[0] == '0');
        if (descend && !fuse_descent && !reserve_only) ZV_CHECK(launch_descend(h, d_queries, nq, entry, d_entries, d_stats, s));
        bool first_launch = true;
        auto go1 = [&](auto kern, uint32_t wparam, uint32_t hb, uint32_t pass, uint32_t *net_list) -> int {
            EfDescend desc{};
            if (fuse_descent && first_launch) desc = EfDescend{h->lvl, h->slot, h->adj_up, h->deg_up, d_entries, entry, 1u};
            first_launch = false;
            // hb: table size as a power of two; the tight pass of the register-beam kernel may carry a slot count instead (hb == 0)
            const uint32_t slots = hb ? (1u << hb) : tight_slots;
            const size_t lds = lds_slots(slots);
            if (epl) hb = slots;  // (k_search_ef_reg takes slots, k_search_ef bits)
            ZV_CHECK(set_dyn_lds(reinterpret_cast<const void *>(kern), lds));
            hipLaunchKernelGGL(kern, dim3(nq), dim3(64 * wpq_launch), lds, s, rows, ld, h->dim, metric, h->adj, h->deg, h->gstride, (uint32_t)N,
                               d_queries, k, ef, entry, id_offset, d_ids, d_dist, d_count, d_stats, dim_pad, hb, wparam, d_entries,
                               pass, h->ef_hint_dev + bucket, static_cast<uint32_t *>(nullptr), 0u, net_list, epl ? width_mode : width, desc);
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
                                   pass, h->ef_hint_dev + bucket, h->ef_visit_buf, (uint32_t)q0, static_cast<uint32_t *>(nullptr), width, EfDescend{});
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
                               (uint32_t)kEfRetry, h->ef_hint_dev + bucket, h->ef_net_buf + h->ef_net_list_words, 0u, h->ef_net_buf, width, EfDescend{});
            ZV_HIP(hipGetLastError());
            h->ef_net_dirty = false;   // (enqueued: it leaves the header words zero)
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
            // Once the fill of this ef bucket is known and the index has a few searches behind it, the tight table is sized for the largest
            // fill seen (<= 74 % load) and a query that still fills it is an outlier: it goes STRAIGHT to the net (16 workgroups on global
            // tables) -- the safe LDS pass between the two, an empty 5-us launch in practice, is left out.  ZVDB_EF_DIRECT_NET=0 / 1: tests.
            const char *dn_env = tuning("ZVDB_EF_DIRECT_NET");
            const bool direct_net = two_pass && !safe_global && (dn_env ? dn_env[0] == '1' : (seen != 0 && h->ef_hint_calls >= 16 && !tp_env));
            if (direct_net) {
                ZV_CHECK(ef_net_begin(h, s));
                ZV_CHECK(go1(kern, wparam, tight_bits, kEfTight, h->ef_net_buf));
                ZV_CHECK(go_net());
            } else {
            if (two_pass) ZV_CHECK(go1(kern, wparam, tight_bits, kEfTight, nullptr));
            if (safe_global) {
                ZV_CHECK(go_global(two_pass ? kEfRetry : kEfSingle));
            } else {
                ZV_CHECK(ef_net_begin(h, s));
                ZV_CHECK(go1(kern, wparam, hash_bits, two_pass ? kEfRetryMark : kEfTight, h->ef_net_buf));
                ZV_CHECK(go_net());
            }
            }
            if (h->ef_hint_calls < 16 || (h->ef_hint_calls & 7u) == 0)
                ZV_HIP(hipMemcpyAsync(h->ef_hint_host, h->ef_hint_dev, 64, hipMemcpyDeviceToHost, s));  // read by a LATER call, never waited for
            h->ef_hint_calls += 1;
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
    h->ef_hint_calls = 0;
    // (an in-flight copy of the old values must not land after the reset -- and the search that makes it may run on a CALLER's stream:
    // wait for the last search's event, not only for this stream)
    if (h->ef_event_set) (void)hipEventSynchronize(h->ef_event);
    (void)hipMemsetAsync(h->ef_hint_dev, 0, 64, s);
    (void)hipStreamSynchronize(s);
    std::memset(h->ef_hint_host, 0, 64);
}

}  // namespace zvdb
