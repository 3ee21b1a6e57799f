// exact_scan.hip -- the repair pass behind the certificate of the brute-force screens (bf_cert.h): for every query on the
// list, the k smallest distances over ALL rows of the shard, each evaluated in the reference's arithmetic (hnsw.zig:182-192:
// one lane walks one row, strictly ascending, sub / mul / add rounded separately -- this file is built with
// -ffp-contract=off), ties by lower id.  No screen, no mirror, no MFMA: what it returns is the definition of a10.
//
//   k_exact_scan        grid (B row ranges, Y query-group workers) x 256 threads: a lane owns a row of the tile and walks it once
//                       for XS_QB queries held in LDS; (dist, id) words that do not exceed the query's bound (start: the k-th result
//                       the screen route returned -- real rows, so an upper bound of the true k-th) go to an LDS buffer, pruned by an
//                       exact radix select when it fills; each block leaves its <= k best per query
//   k_exact_scan_merge  one wave per listed query: the B partial lists -> exact top-k -> the query's result slots
// Both read the list's length on the device: with an empty list (the common case) they return at once.
// Cost when it runs: N dim 3 vector operations per listed query (0.7 ms at 10M x 768) on top of one pass over the rows.
#include "common.h"
#include "wave_select.h"
#include "exact_dist.h"

#pragma clang fp contract(off)

namespace zvdb {

namespace {

// XS_QB (template): queries per pass over a row range -- the rows are read once per XS_QB listed queries; 8, or 2 for rows so wide that 8
// queries do not fit the LDS
constexpr int XS_CAP = 512;   // buffered (dist, id) words per query and block
constexpr int XS_T = 256;     // rows per tile = threads per block
constexpr uint32_t XS_POOL = 8192;  // B k <= this many words in the merge

// compact the <= XS_CAP words of buf to their k smallest (k <= 256) and return the k-th; one wave
__device__ __forceinline__ uint64_t xs_prune(uint64_t *buf, uint32_t n, uint32_t k, uint32_t *hist, uint64_t *tmp, int lane) {
    const uint32_t E = (n + kWave - 1) / kWave;
    const uint64_t T = wave_kth_key<0>([&](uint32_t e) { const uint32_t i = e * kWave + lane; return i < n ? buf[i] : ~0ull; }, E, k, hist, lane);
    uint32_t base = 0;
    for (uint32_t e = 0; e < E; ++e) {
        const uint32_t i = e * kWave + lane;
        const uint64_t w = i < n ? buf[i] : ~0ull;
        const bool sel = w <= T;
        const unsigned long long m = __ballot(sel);
        const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (sel && pos < 256u) tmp[pos] = w;
        base += (uint32_t)__popcll(m);
    }
    // (one wave: its LDS operations complete in order, the copy below sees every store above)
    for (uint32_t i = lane; i < k; i += kWave) buf[i] = tmp[i];
    return T;
}

template <typename RT, int XS_QB>
__global__ __launch_bounds__(XS_T) void k_exact_scan(const RT *__restrict__ rows, uint32_t ld, uint32_t dim, int metric, uint32_t N,
                                                     const float *__restrict__ queries, const uint32_t *__restrict__ nflag,
                                                     const uint32_t *__restrict__ flist, uint32_t k, uint64_t id_offset,
                                                     const uint64_t *__restrict__ cur_ids, const float *__restrict__ cur_dist,
                                                     const uint32_t *__restrict__ cur_count, uint64_t *__restrict__ partial,
                                                     uint32_t rows_per_block, uint32_t dim_pad) {
    const uint32_t nf = *nflag;
    if (nf == 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qs = reinterpret_cast<float *>(smem);                                   // [XS_QB][dim_pad]
    uint64_t *buf = reinterpret_cast<uint64_t *>(qs + (size_t)XS_QB * dim_pad);    // [XS_QB][XS_CAP]
    uint64_t *tmp = buf + XS_QB * XS_CAP;                                          // [4 waves][256]
    uint64_t *thr = tmp + 4 * 256;                                                 // [XS_QB]
    uint32_t *cnt = reinterpret_cast<uint32_t *>(thr + XS_QB);                     // [XS_QB]
    uint32_t *hist = cnt + XS_QB;                                                  // [4 waves][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t B = gridDim.x, b = blockIdx.x;
    const uint32_t r_begin = b * rows_per_block, r_end = min(N, r_begin + rows_per_block);
    for (uint32_t g0 = blockIdx.y * XS_QB; g0 < nf; g0 += gridDim.y * XS_QB) {
        const uint32_t nqg = min((uint32_t)XS_QB, nf - g0);
        __syncthreads();
        for (uint32_t i = tid; i < XS_QB * dim_pad; i += XS_T) {
            const uint32_t j = i / dim_pad, c = i - j * dim_pad;
            qs[i] = (j < nqg && c < dim) ? queries[(size_t)flist[g0 + j] * dim + c] : 0.0f;
        }
        if (tid < XS_QB) {
            uint64_t t = ~0ull - 1;  // no bound
            if ((uint32_t)tid < nqg) {
                const uint32_t qi = flist[g0 + tid];
                // (the re-rank writes a zero dot / cosine distance as -0.0 = -(+0.0); the words below are built from d + 0.0f, one key per
                // value, so the bound is normalised the same way -- else every row at distance 0 would fall outside it)
                if (cur_count[qi] >= k) t = ((uint64_t)f32_key(cur_dist[(size_t)qi * k + k - 1] + 0.0f) << 32) | (uint32_t)(cur_ids[(size_t)qi * k + k - 1] - id_offset);
            }
            thr[tid] = t;
            cnt[tid] = 0;
        }
        __syncthreads();
        for (uint32_t r0 = r_begin; r0 < r_end; r0 += XS_T) {
            const uint32_t row = r0 + tid;
            if (row < r_end) {
                const RT *v = rows + (size_t)row * ld;
                float sum[XS_QB];
#pragma unroll
                for (int j = 0; j < XS_QB; ++j) sum[j] = 0.0f;
                const uint32_t d4 = dim & ~3u;
                uint32_t i = 0;
#pragma unroll 4
                for (; i < d4; i += 4) {
                    const float4 x = row_load4(v + i);
#pragma unroll
                    for (int j = 0; j < XS_QB; ++j) {
                        const float4 a = *reinterpret_cast<const float4 *>(qs + (size_t)j * dim_pad + i);
                        if (metric == ZVDB_METRIC_L2SQ) {
                            float t = a.x - x.x; t = t * t; sum[j] = sum[j] + t;
                            t = a.y - x.y; t = t * t; sum[j] = sum[j] + t;
                            t = a.z - x.z; t = t * t; sum[j] = sum[j] + t;
                            t = a.w - x.w; t = t * t; sum[j] = sum[j] + t;
                        } else {
                            float t = a.x * x.x; sum[j] = sum[j] + t;
                            t = a.y * x.y; sum[j] = sum[j] + t;
                            t = a.z * x.z; sum[j] = sum[j] + t;
                            t = a.w * x.w; sum[j] = sum[j] + t;
                        }
                    }
                }
                for (; i < dim; ++i) {
                    const float x = row_elem(v, i);
#pragma unroll
                    for (int j = 0; j < XS_QB; ++j) {
                        const float a = qs[(size_t)j * dim_pad + i];
                        if (metric == ZVDB_METRIC_L2SQ) { float t = a - x; t = t * t; sum[j] = sum[j] + t; }
                        else { const float t = a * x; sum[j] = sum[j] + t; }
                    }
                }
#pragma unroll
                for (int j = 0; j < XS_QB; ++j) {
                    if ((uint32_t)j >= nqg) break;
                    const float d = (metric == ZVDB_METRIC_L2SQ ? sum[j] : -sum[j]) + 0.0f;  // (-0 -> +0: one key per value)
                    const uint64_t w = ((uint64_t)f32_key(d) << 32) | row;
                    if (w <= thr[j]) {
                        const uint32_t slot = atomicAdd(&cnt[j], 1u);
                        buf[(size_t)j * XS_CAP + slot] = w;  // (slot < XS_CAP: a tile adds <= 256 to a buffer pruned to <= 256)
                    }
                }
            }
            __syncthreads();
            for (uint32_t j = (uint32_t)wave; j < nqg; j += XS_T / kWave) {  // wave w prunes queries w, w + 4, ...
                if (cnt[j] > (uint32_t)(XS_CAP - XS_T)) {
                    const uint64_t T = xs_prune(buf + (size_t)j * XS_CAP, cnt[j], k, hist + wave * 256, tmp + wave * 256, lane);
                    if (lane == 0) { cnt[j] = k; thr[j] = T; }
                }
            }
            __syncthreads();
        }
        for (uint32_t j = (uint32_t)wave; j < nqg; j += XS_T / kWave) {
            uint32_t n = cnt[j];
            if (n > k) { xs_prune(buf + (size_t)j * XS_CAP, n, k, hist + wave * 256, tmp + wave * 256, lane); n = k; }
            uint64_t *out = partial + ((size_t)(g0 + j) * B + b) * k;
            for (uint32_t i = lane; i < k; i += kWave) out[i] = i < n ? buf[(size_t)j * XS_CAP + i] : ~0ull;
        }
    }
}

__global__ __launch_bounds__(64) void k_exact_scan_merge(const uint32_t *__restrict__ nflag, const uint32_t *__restrict__ flist, const uint64_t *__restrict__ partial,
                                                         uint32_t B, uint32_t k, int metric, uint64_t id_offset, uint64_t *__restrict__ out_ids,
                                                         float *__restrict__ out_dist, uint32_t *__restrict__ out_count, uint32_t *host_repaired) {
    const uint32_t f = blockIdx.x;
    if (f >= *nflag) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
    uint64_t *sel = reinterpret_cast<uint64_t *>(smem + SEL_HIST_BYTES);  // [256]
    uint64_t *pool = sel + 256;                                           // [B k]
    const int lane = threadIdx.x;
    const uint32_t n = B * k;
    const uint64_t *p = partial + (size_t)f * n;
    uint32_t nv_l = 0;
    for (uint32_t i = lane; i < n; i += kWave) {
        const uint64_t w = p[i];
        pool[i] = w;
        nv_l += w != ~0ull ? 1u : 0u;
    }
    uint32_t nv = nv_l;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nv += __shfl_xor(nv, off);
    __syncthreads();
    const uint32_t nres = min(nv, k);
    const uint32_t qi = flist[f];
    if (nres > 0) {
        const uint32_t E = (n + kWave - 1) / kWave;
        const uint64_t T = wave_kth_key<0>([&](uint32_t e) { const uint32_t i = e * kWave + lane; return i < n ? pool[i] : ~0ull; }, E, nres, hist, lane);
        uint32_t base = 0;
        for (uint32_t e = 0; e < E; ++e) {
            const uint32_t i = e * kWave + lane;
            const uint64_t w = i < n ? pool[i] : ~0ull;
            const bool s = w <= T;
            const unsigned long long m = __ballot(s);
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
            if (s && pos < 256u) sel[pos] = w;
            base += (uint32_t)__popcll(m);
        }
        __syncthreads();
        for (uint32_t i = lane; i < nres; i += kWave) {  // (dist, id) words are distinct: the rank is the position
            const uint64_t w = sel[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < nres; ++j) rank += sel[j] < w ? 1u : 0u;
            out_ids[(size_t)qi * k + rank] = (uint64_t)(uint32_t)w + id_offset;
            // a zero dot / cosine distance is -(+0.0) = -0.0 in the reference's arithmetic (the oracle and the re-rank write that);
            // the words carry +0.0 so that a value has one key
            const float dv = f32_unkey((uint32_t)(w >> 32));
            out_dist[(size_t)qi * k + rank] = (metric != ZVDB_METRIC_L2SQ && dv == 0.0f) ? -0.0f : dv;
        }
    }
    for (uint32_t i = nres + lane; i < k; i += kWave) {
        out_ids[(size_t)qi * k + i] = ~0ull;
        out_dist[(size_t)qi * k + i] = __builtin_inff();
    }
    if (lane == 0) {
        out_count[qi] = nres;
        if (host_repaired) __hip_atomic_fetch_add(host_repaired, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// row ranges (= partial lists per listed query): B k <= XS_POOL words in the merge, and nq B k words of scratch kept under 64 MiB for large
// batches (down to 16 ranges: the grid's second dimension -- query groups -- carries the parallelism then)
uint32_t xs_blocks(uint32_t nq, uint32_t k) {
    k = std::max(k, 1u);
    const uint32_t by_pool = std::max(1u, std::min(1024u, XS_POOL / k));
    const uint64_t by_mem = (64ull << 20) / (8ull * k * std::max(nq, 1u));
    return std::max<uint32_t>(std::min<uint32_t>(by_pool, 16u), (uint32_t)std::min<uint64_t>(by_pool, by_mem));
}

}  // namespace

uint64_t exact_scan_scratch_bytes(uint32_t nq, uint32_t k) { return (uint64_t)nq * xs_blocks(nq, k) * k * 8; }

int launch_exact_scan(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t k, uint64_t id_offset, const uint32_t *nflag,
                      const uint32_t *flist, void *scratch, uint64_t scratch_bytes, uint64_t *d_ids, float *d_dist, uint32_t *d_count,
                      uint32_t *host_repaired, hipStream_t s) {
    if (nq == 0 || k == 0 || h->count == 0) return ZVDB_OK;
    if (k > 256) { set_error("exact scan: k <= 256"); return ZVDB_ERR_UNSUPPORTED; }
    const uint32_t N = (uint32_t)h->count;
    uint32_t B = xs_blocks(nq, k);
    B = std::min<uint32_t>(B, (N + XS_T - 1) / XS_T);
    if (scratch_bytes < (uint64_t)nq * B * k * 8) { set_error("exact scan: scratch too small"); return ZVDB_ERR_INVALID; }
    const uint32_t rpb = round_up((N + B - 1) / B, XS_T);
    B = (N + rpb - 1) / rpb;
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    const uint32_t dim_pad = round_up(std::max(h->dim, 1u), 4);
    auto lds_for = [&](size_t qb) { return qb * dim_pad * 4 + qb * XS_CAP * 8 + 4 * 256 * 8 + qb * 8 + qb * 4 + 4 * 256 * 4 + 16; };
    const uint32_t QB = lds_for(8) <= 128 * 1024 ? 8u : 2u;
    const size_t lds = lds_for(QB);
    if (lds > 160 * 1024) { set_error("exact scan: dimension too large for the query tile"); return ZVDB_ERR_UNSUPPORTED; }
    const uint32_t Y = std::max(1u, std::min((nq + QB - 1) / QB, std::max(1u, 2048u / B)));
    ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = row_elem_t<decltype(rows)>;
        auto go = [&](auto kern) -> int {
            if (lds > 64 * 1024) ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(B, Y), dim3(XS_T), lds, s, rows, ld, h->dim, metric, N, d_queries, nflag, flist, k, id_offset, d_ids,
                               d_dist, d_count, reinterpret_cast<uint64_t *>(scratch), rpb, dim_pad);
            ZV_HIP(hipGetLastError());
            return ZVDB_OK;
        };
        return QB == 8 ? go(k_exact_scan<RT, 8>) : go(k_exact_scan<RT, 2>);
    }));
    const size_t mlds = SEL_HIST_BYTES + 256 * 8 + (size_t)B * k * 8;
    if (mlds > 64 * 1024) ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_exact_scan_merge), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
    hipLaunchKernelGGL(k_exact_scan_merge, dim3(nq), dim3(64), mlds, s, nflag, flist, reinterpret_cast<const uint64_t *>(scratch), B, k, metric, id_offset, d_ids,
                       d_dist, d_count, host_repaired);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

}  // namespace zvdb
