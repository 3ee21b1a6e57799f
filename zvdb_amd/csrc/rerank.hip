// rerank.hip -- K5: exact re-rank of the brute-force screens' candidates with the reference's sequential fp32 arithmetic
// (hnsw.zig:182-192; -ffp-contract=off + the pragma below), (distance, id) top-k, and the per-query verdict of the exactness
// certificate (bf_cert.h): k_rerank (any list size), k_rerank_staged (LDS-DMA staged gather).
#include <cstring>
#include "common.h"
#include "wave_select.h"
#include "exact_dist.h"
#include "bf_cert.h"
#include "lds_state.h"

#pragma clang fp contract(off)

namespace zvdb {

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
    // kth == 0: not a failed proof but a guess miss handed over by the verifying merge (k_bf_merge_sel vmode 3) -- it loosens the
    // next guesses through its own flag and says nothing about the survivors the proof needs: the level stays where it is
    if (cert.host_flag && kth != 0u) __hip_atomic_store(cert.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
    // (the exact scan's finished-block counter is zero when its launch behind this one starts, whatever an earlier launch left: the merges
    // that normally reset it are one place, this is the other -- nothing between the two reads it)
    if (cert.kth && qi == 0 && threadIdx.x == 0) cert.nflag[1] = 0;
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
// (amdgpu_waves_per_eu(1, 1): a 1,024-query launch is one wave per SIMD -- the LDS stage allows four workgroups per CU -- and its time is
// the slowest wave's chain; told so, hipcc schedules for that instead of for occupancy: 79 -> 63 us per certified batch, A/B of four
// builds on one box, r4.  48 loads in flight per lane for the further candidates measured 6 % SLOWER on top of it.)
template <int KC, typename RT, int CHB = 256, int SR = 64>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_rerank_staged(const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric,
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
    // (the exact scan's finished-block counter is zero when its launch behind this one starts, whatever an earlier launch left: the merges
    // that normally reset it are one place, this is the other -- nothing between the two reads it)
    if (cert.kth && qi == 0 && threadIdx.x == 0) cert.nflag[1] = 0;
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
