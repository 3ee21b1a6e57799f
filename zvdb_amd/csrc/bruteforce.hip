// bruteforce.hip -- K4: batched brute-force Q x V^T on MFMA with a fused, never-materialised
// per-query top-k' selection, followed by a k-way merge of the per-stream lists and (in
// exact_kernels.hip) the exact fp32 re-rank.  No reference symbol exists for this path (a10):
// north_star names it; it is the recall-guaranteed route and the ground-truth generator.
//
//   screen distance   d'(q,v) = |v|^2 - 2 q.v   (the |q|^2 term is constant per query)
//   q.v               fp32 MFMA v_mfma_f32_32x32x2_f32 (exact fma chain) on the fp32 rows, or
//                     bf16 MFMA v_mfma_f32_32x32x16_bf16 on the bf16 mirror (fp32 accumulate)
//   selection         per (query tile, corpus stream) workgroup: running threshold tau[q] =
//                     64th best screen distance seen so far by this workgroup; only d' < tau is
//                     appended (LDS atomic slot + 8-byte global store); a buffer that crosses
//                     its high-water mark is pruned to its exact best 64 by one wave (bitonic
//                     sort / merge on 64-bit (dist,id) keys in registers).  Streams scan rows
//                     in increasing id order, so '<' implements "ties by lower id" exactly.
//
// Work decomposition (DESIGN.md "K4"): 128 x 128 output tile per step, 4 waves as 2(M) x 2(N),
// each wave 64 x 64 = 2 x 2 MFMA 32x32 fragments (64 accumulator VGPRs); K is walked in
// 128-byte slices of every row (32 fp32 or 64 bf16 elements), register-staged global -> LDS,
// double buffered, one barrier per slice; 2 workgroups per CU hide the epilogue / prune of one
// behind the MFMAs of the other.  Block id -> (stream, query tile) keeps all query tiles of a
// stream on one XCD so each 128-row corpus slab is fetched from HBM once per XCD L2.
#include "common.h"

namespace zvdb {

namespace {

constexpr int BM = 128, BN = 128;   // output tile: queries x corpus rows
constexpr int KB = 128;             // bytes of every row consumed per K slice
constexpr int KP = 64;              // k' : candidates kept per (query, stream) and re-ranked
constexpr int HW = 192;             // prune when a buffer holds more than this
constexpr int CAP = HW + BN;        // appends of one tile can never overflow
constexpr int TILE_BYTES = BM * KB; // 16 KiB per operand per stage

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
    return (uint64_t)__shfl((unsigned long long)v, src);
}
__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int m) {
    return (uint64_t)__shfl_xor((unsigned long long)v, m);
}
__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b) { return a < b ? b : a; }

// ascending bitonic sort of 64 keys, one per lane
__device__ __forceinline__ uint64_t wave_sort64(uint64_t key, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint64_t o = shfl_xor64(key, j);
            const bool up = (lane & k) == 0;
            const bool lower = (lane & j) == 0;
            key = (lower == up) ? umin64(key, o) : umax64(key, o);
        }
    }
    return key;
}
// best, nw: ascending; returns the 64 smallest of the union, ascending
__device__ __forceinline__ uint64_t wave_merge64(uint64_t best, uint64_t nw, int lane) {
    uint64_t m = umin64(best, shfl64(nw, 63 - lane));
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const uint64_t o = shfl_xor64(m, j);
        m = (lane & j) == 0 ? umin64(m, o) : umax64(m, o);
    }
    return m;
}

__device__ __forceinline__ uint64_t make_key(float d, uint32_t id) {
    return ((uint64_t)f32_key(d) << 32) | id;
}

// Reduce row `row`'s candidate buffer to its exact best KP (sorted, at the front).
__device__ __forceinline__ uint64_t prune_row(uint64_t *__restrict__ buf, uint32_t *cnt, uint32_t *nsorted,
                                              float *tau, int row, int lane) {
    const uint32_t n = min(cnt[row], (uint32_t)CAP);
    const uint32_t ns = nsorted[row];
    uint64_t best = ~0ull;
    for (uint32_t base = 0; base < n; base += kWave) {
        const uint32_t i = base + lane;
        uint64_t key = i < n ? buf[i] : ~0ull;
        if (!(base == 0 && ns == kWave)) key = wave_sort64(key, lane);
        best = base == 0 ? key : wave_merge64(best, key, lane);
    }
    if (n > 0 && best != ~0ull) buf[lane] = best;
    const uint32_t kept = min(n, (uint32_t)KP);
    const uint64_t last = shfl64(best, KP - 1);
    if (lane == 0) {
        cnt[row] = kept;
        nsorted[row] = kept == KP ? KP : 0;  // a short list is re-sorted next time (it is cheap)
        tau[row] = kept == KP ? f32_unkey((uint32_t)(last >> 32)) : __builtin_inff();
    }
    return best;
}

template <bool BF16>
__global__ __launch_bounds__(256, 2) void k_bf_scan(
    const unsigned char *__restrict__ A, const unsigned char *__restrict__ B, uint32_t row_bytes,
    uint32_t KT, const float *__restrict__ vnorm, uint32_t N, uint32_t nq, float alpha,
    uint32_t QT, uint32_t S, uint32_t NT, uint64_t *__restrict__ cand,
    uint64_t *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *tiles = smem;                                   // [2 stages][A | B] 64 KiB
    float *tau = reinterpret_cast<float *>(smem + 4 * TILE_BYTES); // [BM]
    uint32_t *cnt = reinterpret_cast<uint32_t *>(tau + BM);        // [BM]
    uint32_t *nsorted = cnt + BM;                                  // [BM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // block -> (stream s, query tile qt): all QT tiles of a stream share (blockIdx % 8), i.e.
    // one XCD under round-robin placement (speed only, never correctness).
    const uint32_t b = blockIdx.x;
    const uint32_t x = b & 7, j = b >> 3;
    const uint32_t qt = j % QT;
    const uint32_t s = (j / QT) * 8 + x;
    const uint32_t per = (NT + S - 1) / S;
    const uint32_t t_begin = min(s * per, NT), t_end = min(t_begin + per, NT);

    uint64_t *mybuf = cand + (size_t)b * BM * CAP;
    if (tid < BM) {
        tau[tid] = (qt * BM + tid) < nq ? __builtin_inff() : -__builtin_inff();
        cnt[tid] = 0;
        nsorted[tid] = 0;
    }

    // staging map: chunk c = tid + 256*i -> row c>>3, 16-byte chunk c&7 (8 lanes = one 128-B line)
    const unsigned char *a_base = A + (size_t)qt * BM * row_bytes;  // wave-uniform
    uint32_t a_off[4], lds_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t c = tid + 256 * i;
        const uint32_t row = c >> 3, ch = c & 7;
        a_off[i] = row * row_bytes + ch * 16;
        lds_off[i] = row * KB + ((ch ^ ((row >> 1) & 7)) << 4);
    }
    // fragment read map: lane reads 16 B of row (l & 31), chunk 2*cc + (l >> 5), swizzled
    const uint32_t fr = lane & 31, fh = lane >> 5;

    for (uint32_t t = t_begin; t < t_end; ++t) {
        const uint32_t col0 = t * BN;
        const unsigned char *b_base = B + (size_t)col0 * row_bytes;  // wave-uniform
        uint32_t b_off[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t c = tid + 256 * i;
            const uint32_t row = min((uint32_t)(c >> 3), N - 1 - col0);  // clamp: masked in the epilogue
            b_off[i] = row * row_bytes + (c & 7) * 16;
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

        u32x4 ra[4], rb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const u32x4 *>(a_base + a_off[i]);
            rb[i] = *reinterpret_cast<const u32x4 *>(b_base + b_off[i]);
        }
        __syncthreads();  // previous tile's readers are done with stage 0 / state arrays settled
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4 *>(tiles + lds_off[i]) = ra[i];
            *reinterpret_cast<u32x4 *>(tiles + TILE_BYTES + lds_off[i]) = rb[i];
        }
        __syncthreads();

        for (uint32_t kt = 0; kt < KT; ++kt) {
            const unsigned char *st = tiles + (kt & 1) * 2 * TILE_BYTES;
            const bool more = kt + 1 < KT;
            if (more) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ra[i] = *reinterpret_cast<const u32x4 *>(a_base + (size_t)(kt + 1) * KB + a_off[i]);
                    rb[i] = *reinterpret_cast<const u32x4 *>(b_base + (size_t)(kt + 1) * KB + b_off[i]);
                }
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                u32x4 fa[2], fb[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const uint32_t row = wm * 64 + mi * 32 + fr;
                    fa[mi] = *reinterpret_cast<const u32x4 *>(
                        st + row * KB + (((2 * cc + fh) ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const uint32_t row = wn * 64 + ni * 32 + fr;
                    fb[ni] = *reinterpret_cast<const u32x4 *>(
                        st + TILE_BYTES + row * KB + (((2 * cc + fh) ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        if (BF16) {
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, fa[mi]), __builtin_bit_cast(bf16x8, fb[ni]),
                                acc[mi][ni], 0, 0, 0);
                        } else {
                            const f32x4 x = __builtin_bit_cast(f32x4, fa[mi]);
                            const f32x4 y = __builtin_bit_cast(f32x4, fb[ni]);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.x, acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.y, acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, y.z, acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, y.w, acc[mi][ni], 0, 0, 0);
                        }
                    }
            }
            if (more) {
                unsigned char *nx = tiles + ((kt + 1) & 1) * 2 * TILE_BYTES;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *reinterpret_cast<u32x4 *>(nx + lds_off[i]) = ra[i];
                    *reinterpret_cast<u32x4 *>(nx + TILE_BYTES + lds_off[i]) = rb[i];
                }
            }
            __syncthreads();
        }

        // ---- fused selection epilogue -------------------------------------------------
        // C layout (32x32): column = lane & 31 (corpus row), row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        {
            const uint32_t qb = wm * 64 + 4 * fh;  // lane's first query row; (mi, r) add constants
            const float *tau_l = tau + qb;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const uint32_t col = col0 + wn * 64 + ni * 32 + fr;
                const float vn = col < N ? (vnorm ? vnorm[col] : 0.0f) : __builtin_inff();
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qc = mi * 32 + (r & 3) + 8 * (r >> 2);  // compile-time constant
                        const float d = __builtin_fmaf(alpha, acc[mi][ni][r], vn);
                        if (d < tau_l[qc]) {
                            // rare path: keep its address arithmetic inside the branch (an opaque
                            // copy of the row index stops LICM from hoisting 32 address pairs)
                            uint32_t qrow = qb + qc;
                            asm volatile("" : "+v"(qrow));
                            const uint32_t slot = atomicAdd(&cnt[qrow], 1u);
                            if (slot < (uint32_t)CAP) mybuf[(size_t)qrow * CAP + slot] = make_key(d, col);
                        }
                    }
                }
            }
        }
        const int over = __syncthreads_or(tid < BM && cnt[tid] > (uint32_t)HW);
        if (over) {
            for (int row = wave; row < BM; row += 4)
                if (cnt[row] > (uint32_t)HW) prune_row(mybuf + (size_t)row * CAP, cnt, nsorted, tau, row, lane);
        }
    }

    // ---- stream end: exact best KP per query -> partial[q][s][KP] ------------------------
    __syncthreads();
    for (int row = wave; row < BM; row += 4) {
        const uint32_t q = qt * BM + row;
        if (q >= nq) continue;
        uint64_t best = ~0ull;
        if (cnt[row] > 0) best = prune_row(mybuf + (size_t)row * CAP, cnt, nsorted, tau, row, lane);
        partial[((size_t)q * S + s) * KP + lane] = best;
    }
}

// Merge the S sorted per-stream lists of each query into the global best KP (ids only; the
// exact distances are recomputed by k_rerank).  One wave per query.
__global__ __launch_bounds__(64) void k_bf_merge(const uint64_t *__restrict__ partial, uint32_t S,
                                                 uint32_t *__restrict__ cand_ids) {
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint64_t *p = partial + (size_t)q * S * KP;
    uint64_t best = p[lane];
    for (uint32_t s = 1; s < S; ++s) {
        const uint64_t nw = p[(size_t)s * KP + lane];
        if (__builtin_amdgcn_readfirstlane((uint32_t)(nw >> 32)) == 0xFFFFFFFFu &&
            __builtin_amdgcn_readfirstlane((uint32_t)nw) == 0xFFFFFFFFu)
            continue;  // empty list (its smallest key is the sentinel)
        best = wave_merge64(best, nw, lane);
    }
    cand_ids[(size_t)q * KP + lane] = best == ~0ull ? 0xFFFFFFFFu : (uint32_t)best;
}

// queries -> operand layout of the scan: [nq_pad][row_bytes], zero padded; bf16 RNE for the
// bf16 screen.
__global__ __launch_bounds__(256) void k_bf_prep_queries(const float *__restrict__ q, uint32_t nq,
                                                         uint32_t dim, unsigned char *__restrict__ out,
                                                         uint32_t nq_pad, uint32_t row_elems, int bf16) {
    const uint64_t total = (uint64_t)nq_pad * row_elems;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(t / row_elems), c = (uint32_t)(t - (uint64_t)r * row_elems);
        const float v = (r < nq && c < dim) ? q[(size_t)r * dim + c] : 0.0f;
        if (bf16) {
            uint32_t u = __float_as_uint(v);
            u += 0x7FFFu + ((u >> 16) & 1u);
            reinterpret_cast<uint16_t *>(out)[t] = (uint16_t)(u >> 16);
        } else {
            reinterpret_cast<float *>(out)[t] = v;
        }
    }
}

struct BfPlan {
    uint32_t nq_pad, QT, S, NT, KT, row_bytes;
    uint64_t off_q, off_cand, off_partial, off_ids, total;
};

BfPlan make_plan(const zvdb_hip_index *h, uint32_t nq, uint32_t mode) {
    BfPlan p{};
    p.nq_pad = round_up(std::max(nq, 1u), BM);
    p.QT = p.nq_pad / BM;
    p.NT = (uint32_t)((h->count + BN - 1) / BN);
    p.row_bytes = mode == ZVDB_BF_BF16 ? h->ldh * 2 : h->ld * 4;
    p.KT = p.row_bytes / KB;
    // 2 workgroups per CU x 256 CUs resident; streams in multiples of 8 (XCD groups)
    const uint32_t resident = 512;
    uint32_t S = std::max(8u, (resident / p.QT) / 8 * 8);
    while (S > 8 && S > p.NT) S -= 8;  // do not create streams with no tiles
    p.S = S;
    uint64_t off = 0;
    p.off_q = off;
    off += round_up64((uint64_t)p.nq_pad * p.row_bytes, 256);
    p.off_cand = off;
    off += (uint64_t)p.QT * p.S * BM * CAP * 8;
    p.off_partial = off;
    off += (uint64_t)p.nq_pad * p.S * KP * 8;
    p.off_ids = off;
    off += (uint64_t)p.nq_pad * KP * 4;
    p.total = off;
    return p;
}

}  // namespace

uint64_t bruteforce_workspace_bytes(const zvdb_hip_index *h, uint32_t nq, uint32_t k, uint32_t mode) {
    (void)k;
    if (h->dim == 0) return 256;
    return make_plan(h, nq, mode).total;
}

int launch_bruteforce(zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t k,
                      uint32_t mode, uint64_t id_offset, void *d_ws, uint64_t ws_bytes,
                      uint64_t *d_ids, float *d_dist, uint32_t *d_count, hipStream_t s) {
    if (nq == 0) return ZVDB_OK;
    // the screen keeps k' = 64 candidates per query; the exact re-rank needs a margin over k to
    // absorb the screen's rounding (bf16 inputs: wider margin).  TODO(next round): 2 keys/lane.
    const uint32_t kmax = mode == ZVDB_BF_BF16 ? KP - 16 : KP - 8;
    if (k > kmax) {
        set_error("brute force supports k <= %u in this mode (k' = %d candidates are re-ranked exactly)", kmax, KP);
        return ZVDB_ERR_UNSUPPORTED;
    }
    if (mode == ZVDB_BF_BF16 && !(h->flags & ZVDB_FLAG_BF16_MIRROR)) {
        set_error("ZVDB_BF_BF16 needs an index created with ZVDB_FLAG_BF16_MIRROR");
        return ZVDB_ERR_INVALID;
    }
    if (h->count > 0xFFFFFFFFull - BN) {
        set_error("shard too large");
        return ZVDB_ERR_UNSUPPORTED;
    }
    const BfPlan p = make_plan(h, nq, mode);
    if (ws_bytes < p.total) {
        set_error("workspace too small: need %llu bytes, got %llu", (unsigned long long)p.total,
                  (unsigned long long)ws_bytes);
        return ZVDB_ERR_INVALID;
    }
    unsigned char *ws = static_cast<unsigned char *>(d_ws);
    unsigned char *q_op = ws + p.off_q;
    uint64_t *cand = reinterpret_cast<uint64_t *>(ws + p.off_cand);
    uint64_t *partial = reinterpret_cast<uint64_t *>(ws + p.off_partial);
    uint32_t *cand_ids = reinterpret_cast<uint32_t *>(ws + p.off_ids);

    const bool bf16 = mode == ZVDB_BF_BF16;
    const uint32_t row_elems = bf16 ? h->ldh : h->ld;
    {
        const uint64_t total = (uint64_t)p.nq_pad * row_elems;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 2048);
        hipLaunchKernelGGL(k_bf_prep_queries, dim3(grid), dim3(256), 0, s, d_queries, nq, h->dim, q_op,
                           p.nq_pad, row_elems, bf16 ? 1 : 0);
    }
    // screen key: L2 -> |v|^2 - 2 q.v ; dot / cosine -> -q.v (norm array unused: alpha only)
    const bool l2 = h->metric == ZVDB_METRIC_L2SQ;
    const float alpha = l2 ? -2.0f : -1.0f;
    const float *vn = l2 ? (bf16 ? h->norm_h : h->norm) : nullptr;
    const size_t lds = 4 * TILE_BYTES + 3 * BM * 4;
    const uint32_t grid = p.QT * p.S;
    timing_begin(h, s);
    if (bf16) {
        ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bf_scan<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_bf_scan<true>, dim3(grid), dim3(256), lds, s,
                           q_op, reinterpret_cast<const unsigned char *>(h->rows_h), p.row_bytes, p.KT, vn,
                           (uint32_t)h->count, nq, alpha, p.QT, p.S, p.NT, cand, partial);
    } else {
        ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bf_scan<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_bf_scan<false>, dim3(grid), dim3(256), lds, s,
                           q_op, reinterpret_cast<const unsigned char *>(h->rows), p.row_bytes, p.KT, vn,
                           (uint32_t)h->count, nq, alpha, p.QT, p.S, p.NT, cand, partial);
    }
    timing_end(h, s);
    ZV_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_bf_merge, dim3(nq), dim3(64), 0, s, partial, p.S, cand_ids);
    ZV_HIP(hipGetLastError());
    return launch_rerank(h, d_queries, nq, cand_ids, KP, k, id_offset, d_ids, d_dist, d_count, s);
}

}  // namespace zvdb
