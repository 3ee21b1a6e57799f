// exact_scan.hip -- the repair pass behind the certificate of the brute-force screens (bf_cert.h): for every query on the
// list, the k smallest distances over ALL rows of the shard, each evaluated in the reference's arithmetic (hnsw.zig:182-192:
// one lane walks one row, strictly ascending, sub / mul / add rounded separately -- this file is built with
// -ffp-contract=off), ties by lower id.  No screen, no mirror, no MFMA: what it returns is the definition of a10.
//
//   k_exact_scan   grid (B row ranges, Y workers over groups of listed queries) x 512 threads.  Up to 32 listed queries share ONE pass
//                  over the rows: a lane owns a row of the 512-row tile and keeps one running sum per query in registers; the rows
//                  stream through LDS a cache line per row at a time (fetched coalesced, two chunks ahead), the same chunk of every
//                  query beside them, and the reference's three operations run as packed fp32 pairs (two queries per instruction,
//                  each component rounded on its own).  (dist, id) words within the query's bound (start: the k-th result the screen
//                  route returned -- real rows, so an upper bound of the true k-th) go to an LDS buffer, pruned by an exact radix
//                  select when it fills (nothing is ever dropped: a lane whose word did not fit tries again after the prune); each
//                  block leaves its <= k best per query, and the block that finishes LAST merges the B partial lists of every listed
//                  query into the result slots (no second launch).
// The list's length is read on the device: with an empty list (the common case) the launch returns at once.
// Cost when it runs (10M x 768 fp32 rows, profiles/r5/exact_scan_ab.txt): one pass over the rows per 32 listed queries; a list of <= 8
// queries 7.9 ms (3.9 TB/s.  NOT the memory: the same walk without arithmetic reads 7.0 TB/s whatever the piece size; the LDS reads of the queries' values cost
// 12 - 33 % of a pass, and a scalar-load replacement was built and lost 2.4x to SMEM latency -- profiles/r6/exact_scan_ab.txt), 16 queries 14.3 ms, 32 queries
// 26.4 ms = 0.82 ms per query (the arithmetic: 3 N dim operations per query; r4: 10 ms per 8 queries, one pass per 8).
#include "common.h"
#include "wave_select.h"
#include "exact_dist.h"

#pragma clang fp contract(off)

namespace zvdb {

namespace {

// ---- geometry ---------------------------------------------------------------------------------------------------------------
constexpr int XS_T = 512;            // threads per block = rows per tile (a lane owns one row of the tile)
constexpr int XS_CHUNK = 128;        // bytes of a row per chunk: one cache line (32 fp32 / 64 bf16 elements; rows are padded to whole lines)
constexpr int XS_ROWB = XS_CHUNK + 16;  // LDS bytes per row of the tile: 36 dwords -- the 16 lanes of a ds_read_b128 group hit 16 distinct bank quads
constexpr int XS_PIECES = XS_CHUNK / 16;  // 16-byte pieces per row and chunk; also pieces per thread and chunk of the coalesced fill (512 x 8 = 4,096)
constexpr int XS_PRUNERS = 4;        // waves that prune (each with its own histogram + staging)

typedef float xs_f2 __attribute__((ext_vector_type(2)));
typedef float xs_f4 __attribute__((ext_vector_type(4)));

// N values that live in registers: an array member indexed only by unrolled loops
template <typename T, int N> struct XsRegs { T v[N]; };

template <typename RT> struct xs_epc { static constexpr int v = XS_CHUNK / (int)sizeof(RT); };  // elements per chunk

// buffered (dist, id) words per listed query and block: twice the result count (a prune keeps k), at least 64
__host__ __device__ inline uint32_t xs_cap(uint32_t k) { uint32_t c = 64; while (c < 2 * k) c <<= 1; return c; }
// queries that share one pass over the rows: as many as the LDS holds buffers for (32 for k <= 128, 16 for k <= 256) and the
// register file running sums for (64 sums spill)
__host__ __device__ inline uint32_t xs_nq_max(uint32_t k) { const uint32_t c = xs_cap(k); return c <= 256 ? 32u : 16u; }

// compact the n <= 512 words of buf to their k smallest (k <= 256) and return the k-th; one wave
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

struct XsArgs {
    const void *rows; uint32_t ld, dim; int metric; uint32_t N;
    const float *queries; const uint32_t *flist; uint32_t k; uint64_t id_offset;
    const uint64_t *cur_ids; const float *cur_dist; const uint32_t *cur_count; uint64_t *partial;
    uint64_t *out_ids; float *out_dist; uint32_t *out_count; uint32_t *host_repaired;   // (the result slots: the same arrays the bound was read from)
    uint32_t rows_per_block, cap;
};

// One pass of a block over its row range for the listed queries [g0, g0 + nqg), nqg <= NQ: NQ running sums per lane, in registers.
// The rows stream through LDS a cache line per row at a time: 512 rows x 128 B fetched coalesced (8 lanes per line) two chunks
// ahead into registers, written to a padded LDS image, and each lane then walks ITS row's line out of LDS in the reference's order;
// the same chunk of all NQ queries sits beside it in LDS, element-major, so that one broadcast ds_read_b128 feeds four queries and
// the three operations of hnsw.zig:188-189 run as packed fp32 pairs (v_pk_add / v_pk_mul / v_pk_add: two queries per instruction,
// each component rounded on its own -- the arithmetic and its order per (query, row) are the scalar loop's).
template <typename RT, int NQ>
__device__ __forceinline__ void xs_pass(const XsArgs &a, uint32_t g0, uint32_t nqg, unsigned char *smem) {
    constexpr int EPC = xs_epc<RT>::v;            // elements per chunk
    constexpr int QG = NQ * EPC / 4;                   // 4-element granules of the query chunk: granule g = (elements 4 (g / NQ) .. + 3, query g % NQ)
    constexpr int QPT = (QG + XS_T - 1) / XS_T;        // granules per thread and chunk
    const RT *rows = static_cast<const RT *>(a.rows);
    unsigned char *tile = smem;                                              // [XS_T][XS_ROWB]
    float *qs = reinterpret_cast<float *>(tile + XS_T * XS_ROWB);            // [EPC][NQ]   (element-major)
    uint64_t *buf = reinterpret_cast<uint64_t *>(qs + EPC * NQ);             // [NQ][cap]
    uint64_t *tmp = buf + (size_t)NQ * a.cap;                                // [XS_PRUNERS][256]
    uint64_t *thr = tmp + XS_PRUNERS * 256;                                  // [NQ]
    uint32_t *cnt = reinterpret_cast<uint32_t *>(thr + NQ);                  // [NQ]
    uint32_t *hist = cnt + NQ;                                               // [XS_PRUNERS][256]
    uint32_t *qrow = hist + XS_PRUNERS * 256;                                // [NQ] row of the query in `queries`, ~0 = none
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t B = gridDim.x, b = blockIdx.x, k = a.k, cap = a.cap;
    const uint32_t r_begin = b * a.rows_per_block, r_end = min(a.N, r_begin + a.rows_per_block);
    const uint32_t ntiles = r_end > r_begin ? (r_end - r_begin + XS_T - 1) / XS_T : 0;
    const uint32_t nch = a.ld / EPC;               // chunks per row (ld is a whole number of lines)
    const uint32_t total = ntiles * nch;           // the block's chunk stream: tile-major
    __syncthreads();                               // (the previous pass has left the LDS)
    if (tid < NQ) {
        uint64_t t = ~0ull - 1;  // no bound
        uint32_t qr = ~0u;
        if ((uint32_t)tid < nqg) {
            const uint32_t qi = a.flist[g0 + tid];
            qr = qi;
            // start bound: the k-th result the screen route returned -- real rows, so an upper bound of the true k-th.  (The re-rank
            // writes a zero dot / cosine distance as -0.0 = -(+0.0); the words of this kernel are built from d + 0.0f, one key per
            // value, so the bound is normalised the same way -- else every row at distance 0 would fall outside it.)
            if (a.cur_count[qi] >= k) t = ((uint64_t)f32_key(a.cur_dist[(size_t)qi * k + k - 1] + 0.0f) << 32) | (uint32_t)(a.cur_ids[(size_t)qi * k + k - 1] - a.id_offset);
        }
        thr[tid] = t;
        cnt[tid] = 0;
        qrow[tid] = qr;
    }
    __syncthreads();
    // ---- the two streams: row pieces (8 per thread and chunk) and query granules (QPT per thread and chunk), held in registers between
    // their fetch and their LDS store.  (ext-vector element types: HIP's uint4 / float4 structs inside a by-value aggregate end up in
    // scratch memory -- load, wait, scratch store -- where ext vectors stay in registers.)
    using Regs = XsRegs<u32x4, XS_PIECES>;
    using QRegs = XsRegs<xs_f4, QPT>;
    const uint32_t fr = (uint32_t)tid >> 3, fp = (uint32_t)tid & 7;   // fill: rows fr + 64 i of the tile, piece fp
    const unsigned char *rows_b = reinterpret_cast<const unsigned char *>(rows);
    const size_t row_bytes = (size_t)a.ld * sizeof(RT);
    const uint32_t n_last = a.N - 1;
    auto fetch_rows = [&](uint32_t c) __attribute__((always_inline)) -> Regs {
        c = min(c, total - 1);   // (past the end of the stream: the last chunk again -- loads without a branch, never stored)
        const uint32_t t = c / nch, cc = c - t * nch;
        Regs r;
#pragma unroll
        for (int i = 0; i < XS_PIECES; ++i)   // (rows past the end: the last row; their lanes append nothing)
            r.v[i] = *reinterpret_cast<const u32x4 *>(rows_b + (size_t)min(r_begin + t * XS_T + fr + 64u * i, n_last) * row_bytes + (size_t)cc * XS_CHUNK + fp * 16);
        return r;
    };
    auto fetch_q = [&](uint32_t c) __attribute__((always_inline)) -> QRegs {
        QRegs qa;
        const uint32_t cc = min(c, total - 1) % nch;
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const uint32_t g = (uint32_t)tid + XS_T * i;          // consecutive threads: consecutive queries (the LDS image is element-major)
            const uint32_t j = g % NQ, col = cc * EPC + 4 * (g / NQ);
            const uint32_t qr = g < (uint32_t)QG ? qrow[j] : ~0u;
            const float *src = a.queries + (size_t)(qr != ~0u ? qr : 0) * a.dim;
            // (padding: (0 - 0)^2 = +0.0 and 0 * x = +-0.0 leave a running sum as it is -- it starts at +0.0 and never becomes -0.0)
            qa.v[i].x = (qr != ~0u && col + 0 < a.dim) ? src[col + 0] : 0.0f;
            qa.v[i].y = (qr != ~0u && col + 1 < a.dim) ? src[col + 1] : 0.0f;
            qa.v[i].z = (qr != ~0u && col + 2 < a.dim) ? src[col + 2] : 0.0f;
            qa.v[i].w = (qr != ~0u && col + 3 < a.dim) ? src[col + 3] : 0.0f;
        }
        return qa;
    };
    auto store = [&](const Regs &r, const QRegs &qa) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < XS_PIECES; ++i) *reinterpret_cast<u32x4 *>(tile + (size_t)(fr + 64u * i) * XS_ROWB + fp * 16) = r.v[i];
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const uint32_t g = (uint32_t)tid + XS_T * i;
            if (g < (uint32_t)QG) {
                float *dst = qs + 4 * (g / NQ) * NQ + g % NQ;
                dst[0] = qa.v[i].x; dst[NQ] = qa.v[i].y; dst[2 * NQ] = qa.v[i].z; dst[3 * NQ] = qa.v[i].w;
            }
        }
    };
    xs_f2 sum[NQ / 2];
#pragma unroll
    for (int j = 0; j < NQ / 2; ++j) sum[j] = xs_f2{0.0f, 0.0f};
    const unsigned char *myrow = tile + (size_t)tid * XS_ROWB;
    const bool l2 = a.metric == ZVDB_METRIC_L2SQ;
    auto elem = [&](auto l2c, float x, uint32_t el) __attribute__((always_inline)) {   // one element of the lane's row against the NQ queries
        const xs_f2 xx = xs_f2{x, x};
#pragma unroll
        for (int jq = 0; jq < NQ / 4; ++jq) {
            const xs_f4 q4 = *reinterpret_cast<const xs_f4 *>(qs + el * NQ + 4 * jq);
            const xs_f2 q01 = xs_f2{q4.x, q4.y}, q23 = xs_f2{q4.z, q4.w};
            if constexpr (decltype(l2c)::value) {
                xs_f2 t0 = q01 - xx; t0 = t0 * t0; sum[2 * jq] = sum[2 * jq] + t0;
                xs_f2 t1 = q23 - xx; t1 = t1 * t1; sum[2 * jq + 1] = sum[2 * jq + 1] + t1;
            } else {
                const xs_f2 t0 = q01 * xx; sum[2 * jq] = sum[2 * jq] + t0;
                const xs_f2 t1 = q23 * xx; sum[2 * jq + 1] = sum[2 * jq + 1] + t1;
            }
        }
    };
    auto compute_m = [&](auto l2c) __attribute__((always_inline)) {
#pragma unroll 2
        for (int p = 0; p < XS_PIECES; ++p) {
            const u32x4 w = *reinterpret_cast<const u32x4 *>(myrow + p * 16);
            if constexpr (sizeof(RT) == 4) {
                elem(l2c, __uint_as_float(w.x), 4 * p); elem(l2c, __uint_as_float(w.y), 4 * p + 1);
                elem(l2c, __uint_as_float(w.z), 4 * p + 2); elem(l2c, __uint_as_float(w.w), 4 * p + 3);
            } else {
                elem(l2c, bf16_lo(w.x), 8 * p); elem(l2c, bf16_hi(w.x), 8 * p + 1); elem(l2c, bf16_lo(w.y), 8 * p + 2); elem(l2c, bf16_hi(w.y), 8 * p + 3);
                elem(l2c, bf16_lo(w.z), 8 * p + 4); elem(l2c, bf16_hi(w.z), 8 * p + 5); elem(l2c, bf16_lo(w.w), 8 * p + 6); elem(l2c, bf16_hi(w.w), 8 * p + 7);
            }
        }
    };
    auto compute = [&]() __attribute__((always_inline)) {
        if (l2) compute_m(std::true_type{}); else compute_m(std::false_type{});
    };
    // a tile's rows are done: every (dist, id) word within its query's bound goes to that query's buffer; a buffer that overflows is
    // pruned to its k best (which tightens the bound) and the lanes still holding a word try again -- nothing is ever dropped
    auto flush = [&](uint32_t t) __attribute__((always_inline)) {
        const uint32_t row = r_begin + t * XS_T + tid;
        unsigned long long pend = 0;
        if (row < r_end) pend = (1ull << nqg) - 1;   // (nqg <= 32)
        for (;;) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                if (pend >> j & 1) {
                    const float sj = sum[j >> 1][j & 1];
                    const float d = (l2 ? sj : -sj) + 0.0f;  // (-0 -> +0: one key per value)
                    const uint64_t w = ((uint64_t)f32_key(d) << 32) | row;
                    bool done = true;
                    if (w <= thr[j]) {
                        const uint32_t slot = atomicAdd(&cnt[j], 1u);
                        if (slot < cap) buf[(size_t)j * cap + slot] = w; else done = false;
                    }
                    if (done) pend &= ~(1ull << j);
                }
            }
            __syncthreads();
            if (wave < XS_PRUNERS) {
                for (uint32_t j = (uint32_t)wave; j < nqg; j += XS_PRUNERS) {
                    if (cnt[j] > cap) {   // some append did not fit: keep the k best of the cap buffered, new bound = the k-th
                        const uint64_t T = xs_prune(buf + (size_t)j * cap, cap, k, hist + wave * 256, tmp + wave * 256, lane);
                        if (lane == 0) { cnt[j] = k; thr[j] = T; }
                    }
                }
            }
            if (!__syncthreads_or(pend != 0)) break;
        }
#pragma unroll
        for (int j = 0; j < NQ / 2; ++j) sum[j] = xs_f2{0.0f, 0.0f};
    };
    // ---- the pipeline: chunk c in LDS; c + 1 (and, with <= 16 running sums, c + 2) in flight in registers.  A pass with 32 sums is
    // bound by its arithmetic (and by the register file: 32 sums + two chunks in flight + the walk's temporaries would spill) ---------
    if (total > 0) {
        // (vmcnt counts in issue order: a store waits for everything issued BEFORE the youngest load it needs -- so the small query
        // fetch of chunk c + 1 is issued ahead of the row fetch of chunk c + 2, and a store leaves that row fetch in flight)
        Regs ra = fetch_rows(0);
        QRegs qa = fetch_q(0);
        store(ra, qa);
        ra = fetch_rows(1);
        qa = fetch_q(1);
        uint32_t cc = 0, t = 0;
        if constexpr (NQ <= 16) {
            Regs rb = fetch_rows(2);
            __syncthreads();
            for (uint32_t c = 0; c < total; c += 2) {
                // even step: LDS holds c; in flight, oldest first: ra = rows c + 1, qa = queries c + 1, rb = rows c + 2
                compute();
                if (++cc == nch) { cc = 0; flush(t++); } else __syncthreads();
                if (c + 1 >= total) break;
                store(ra, qa);
                qa = fetch_q(c + 2);
                ra = fetch_rows(c + 3);
                __syncthreads();
                // odd step: LDS holds c + 1; in flight: rb = rows c + 2, qa = queries c + 2, ra = rows c + 3
                compute();
                if (++cc == nch) { cc = 0; flush(t++); } else __syncthreads();
                if (c + 2 >= total) break;
                store(rb, qa);
                qa = fetch_q(c + 3);
                rb = fetch_rows(c + 4);
                __syncthreads();
            }
        } else {
            __syncthreads();
            for (uint32_t c = 0; c < total; ++c) {
                compute();
                if (++cc == nch) { cc = 0; flush(t++); } else __syncthreads();
                if (c + 1 >= total) break;
                store(ra, qa);
                ra = fetch_rows(c + 2);
                qa = fetch_q(c + 2);
                __syncthreads();
            }
        }
    }
    // ---- the block's <= k best per query -> its partial list -----------------------------------------------------------------------
    __syncthreads();
    if (wave < XS_PRUNERS) {
        for (uint32_t j = (uint32_t)wave; j < nqg; j += XS_PRUNERS) {
            uint32_t n = min(cnt[j], cap);
            if (n > k) { xs_prune(buf + (size_t)j * cap, n, k, hist + wave * 256, tmp + wave * 256, lane); n = k; }
            uint64_t *out = a.partial + ((size_t)(g0 + j) * B + b) * k;
            for (uint32_t i = lane; i < k; i += kWave) out[i] = i < n ? buf[(size_t)j * cap + i] : ~0ull;
        }
    }
}

// (Measured and removed in round 5: a second pass for lists of <= 8 queries in which a lane is a (row, query) PAIR and the tile holds
// 1 KiB of each of 64 rows, fetched as whole contiguous KiB -- meant to turn the 128-byte pieces of the pass above into long reads.
// 10.4 - 10.8 ms per pass at 10M x 768 against 7.9 ms here, fp32 and bf16 rows alike, A/B of two builds on one box
// (profiles/r5/exact_scan_ab.txt): the walk's 256 dependent adds per chunk and its two LDS reads per four elements cost more than the
// longer pieces gain.)

// LDS bytes of a pass with NQ queries (host + device agree)
template <typename RT>
__host__ __device__ inline size_t xs_lds(uint32_t nq_pass, uint32_t cap) {
    return (size_t)XS_T * XS_ROWB + (size_t)xs_epc<RT>::v * nq_pass * 4 + (size_t)nq_pass * cap * 8 + XS_PRUNERS * 256 * 8 + (size_t)nq_pass * 8 +
           (size_t)nq_pass * 4 + XS_PRUNERS * 256 * 4 + (size_t)nq_pass * 4 + 16;
}

// The partial lists of listed query f (B lists of k words, ~0 = none) -> exact top-k -> the query's result slots; one wave, its own
// LDS (XS_MERGE_LDS bytes).  The lists go through LDS in rounds of XS_MPOOL words beside the best k so far.  (One wave: its LDS
// operations complete in order, a load sees every store issued before it.)
constexpr uint32_t XS_MPOOL = 2048;
constexpr uint32_t XS_MERGE_LDS = SEL_HIST_BYTES + 256 * 8 + (256 + XS_MPOOL) * 8;
constexpr int XS_MERGERS = 4;   // waves of the last block that merge
__device__ __forceinline__ void xs_merge(const XsArgs &a, uint32_t f, uint32_t B, unsigned char *lds, int lane) {
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds);
    uint64_t *sel = reinterpret_cast<uint64_t *>(lds + SEL_HIST_BYTES);  // [256]
    uint64_t *pool = sel + 256;                                          // [256 + XS_MPOOL]
    const uint32_t k = a.k, n = B * k;
    const uint64_t *p = a.partial + (size_t)f * n;
    uint32_t have = 0;   // valid words kept at the front of the pool (<= k)
    for (uint32_t base = 0; base < n; base += XS_MPOOL) {
        const uint32_t m = min(XS_MPOOL, n - base);
        uint32_t nv_l = 0;
        for (uint32_t i = lane; i < m; i += kWave) {
            const uint64_t w = __builtin_nontemporal_load(p + base + i);
            pool[have + i] = w;
            nv_l += w != ~0ull ? 1u : 0u;
        }
        uint32_t nv = nv_l;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nv += __shfl_xor(nv, off);
        const uint32_t tot = have + m, keep = min(have + nv, k);
        if (keep > 0) {
            const uint32_t E = (tot + kWave - 1) / kWave;
            const uint64_t T = wave_kth_key<0>([&](uint32_t e) { const uint32_t i = e * kWave + lane; return i < tot ? pool[i] : ~0ull; }, E, keep, hist, lane);
            uint32_t cb = 0;
            for (uint32_t e = 0; e < E; ++e) {
                const uint32_t i = e * kWave + lane;
                const uint64_t w = i < tot ? pool[i] : ~0ull;
                const bool sl = w <= T;
                const unsigned long long mk = __ballot(sl);
                const uint32_t pos = cb + (uint32_t)__popcll(mk & ((1ull << lane) - 1));
                if (sl && pos < 256u) sel[pos] = w;
                cb += (uint32_t)__popcll(mk);
            }
            for (uint32_t i = lane; i < keep; i += kWave) pool[i] = sel[i];
        }
        have = keep;
    }
    const uint32_t nres = have;
    const uint32_t qi = a.flist[f];
    for (uint32_t i = lane; i < nres; i += kWave) {  // (dist, id) words are distinct: the rank is the position
        const uint64_t w = pool[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nres; ++j) rank += pool[j] < w ? 1u : 0u;
        a.out_ids[(size_t)qi * k + rank] = (uint64_t)(uint32_t)w + a.id_offset;
        // a zero dot / cosine distance is -(+0.0) = -0.0 in the reference's arithmetic (the oracle and the re-rank write that);
        // the words carry +0.0 so that a value has one key
        const float dv = f32_unkey((uint32_t)(w >> 32));
        a.out_dist[(size_t)qi * k + rank] = (a.metric != ZVDB_METRIC_L2SQ && dv == 0.0f) ? -0.0f : dv;
    }
    for (uint32_t i = nres + lane; i < k; i += kWave) {
        a.out_ids[(size_t)qi * k + i] = ~0ull;
        a.out_dist[(size_t)qi * k + i] = __builtin_inff();
    }
    if (lane == 0) {
        a.out_count[qi] = nres;
        if (a.host_repaired) __hip_atomic_fetch_add(a.host_repaired, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// grid (B row ranges, Y workers over the groups of listed queries) x 512 threads.  The list's length is read on the device: empty (the
// common case) -> return at once; else groups of up to nq_max queries share one pass over the rows, and a short list takes the
// instantiation with as few running sums as it needs (8 / 16 / 32).  The block that finishes LAST (a counter next to the list's
// length, reset by the merge kernels of the screen and by its last user) merges the partial lists of every listed query into the
// result slots: no second launch -- the release / acquire fences around the counter are paid only when there is a list.
template <typename RT>
__global__ __launch_bounds__(XS_T) void k_exact_scan(XsArgs a, uint32_t *__restrict__ nflag, uint32_t nq_max) {
    const uint32_t nf = nflag[0];
    if (nf == 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_last;
    const uint32_t per = min(nq_max, nf <= 8 ? 8u : nf <= 16 ? 16u : 32u);   // (block-uniform)
    for (uint32_t g0 = blockIdx.y * per; g0 < nf; g0 += gridDim.y * per) {
        const uint32_t nqg = min(per, nf - g0);
        if (per == 8) xs_pass<RT, 8>(a, g0, nqg, smem);
        else if (per == 16) xs_pass<RT, 16>(a, g0, nqg, smem);
        else xs_pass<RT, 32>(a, g0, nqg, smem);
    }
    __threadfence();   // every wave that wrote partial lists: its own stores are out (agent scope) before the barrier lets thread 0 count
    __syncthreads();   // (every wave's partial lists are stored)
    if (threadIdx.x == 0) {
        __threadfence();   // release: this block's lists before its count
        s_last = atomicAdd(nflag + 1, 1u) == gridDim.x * gridDim.y - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();       // acquire: every block's lists
    if (threadIdx.x == 0) nflag[1] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < XS_MERGERS)
        for (uint32_t f = (uint32_t)wave; f < nf; f += XS_MERGERS) xs_merge(a, f, gridDim.x, smem + (size_t)wave * XS_MERGE_LDS, lane);
}

// row ranges (= partial lists per listed query): two per CU when the scratch allows (nq B k words kept under 128 MiB for large batches,
// never fewer than 16 ranges)
uint32_t xs_blocks(uint32_t nq, uint32_t k) {
    k = std::max(k, 1u);
    const uint64_t by_mem = (128ull << 20) / (8ull * k * std::max(nq, 1u));
    return (uint32_t)std::max<uint64_t>(16, std::min<uint64_t>(512, by_mem));
}

}  // namespace

uint64_t exact_scan_scratch_bytes(uint32_t nq, uint32_t k) { return (uint64_t)nq * xs_blocks(nq, k) * k * 8; }

int launch_exact_scan(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t k, uint64_t id_offset, const uint32_t *nflag,
                      const uint32_t *flist, void *scratch, uint64_t scratch_bytes, uint64_t *d_ids, float *d_dist, uint32_t *d_count,
                      uint32_t *host_repaired, hipStream_t s) {
    // (nflag: [0] the list's length, [1] the finished-block counter of the scan -- both reset by the screen's merge kernels)
    if (nq == 0 || k == 0 || h->count == 0) return ZVDB_OK;
    if (k > 256) { set_error("exact scan: k <= 256"); return ZVDB_ERR_UNSUPPORTED; }
    const uint32_t N = (uint32_t)h->count;
    uint32_t B = xs_blocks(nq, k);
    B = std::min<uint32_t>(B, (N + XS_T - 1) / XS_T);
    if (scratch_bytes < (uint64_t)nq * B * k * 8) { set_error("exact scan: scratch too small"); return ZVDB_ERR_INVALID; }
    const uint32_t rpb = round_up((N + B - 1) / B, XS_T);
    B = (N + rpb - 1) / rpb;
    const int metric = h->metric == ZVDB_METRIC_L2SQ ? ZVDB_METRIC_L2SQ : ZVDB_METRIC_DOT;
    const uint32_t cap = xs_cap(k);
    // the queries of one pass: what the LDS holds buffers for, and no more running sums than the batch can list
    uint32_t nq_pass = xs_nq_max(k);
    while (nq_pass > 8 && nq_pass / 2 >= nq) nq_pass /= 2;
    const uint32_t Y = std::max(1u, std::min((nq + nq_pass - 1) / nq_pass, std::max(1u, 1024u / B)));
    ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = row_elem_t<decltype(rows)>;
        const size_t lds = std::max<size_t>(xs_lds<RT>(nq_pass, cap), (size_t)XS_MERGERS * XS_MERGE_LDS);
        // (gfx950 only: the 512-row tile alone is 72 KB and a 32-query pass plans ~160 KB -- the CDNA4 LDS; a device with less fails here, loudly)
        if (lds > 160 * 1024) { set_error("exact scan: LDS plan of %zu bytes exceeds the 160 KB of a gfx950 CU", lds); return ZVDB_ERR_UNSUPPORTED; }
        XsArgs a;
        a.rows = rows; a.ld = ld; a.dim = h->dim; a.metric = metric; a.N = N;
        a.queries = d_queries; a.flist = flist; a.k = k; a.id_offset = id_offset;
        a.cur_ids = d_ids; a.cur_dist = d_dist; a.cur_count = d_count; a.partial = reinterpret_cast<uint64_t *>(scratch);
        a.out_ids = d_ids; a.out_dist = d_dist; a.out_count = d_count; a.host_repaired = host_repaired;
        a.rows_per_block = rpb; a.cap = cap;
        auto kern = k_exact_scan<RT>;
        if (lds > 64 * 1024) ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(B, Y), dim3(XS_T), lds, s, a, const_cast<uint32_t *>(nflag), nq_pass);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    }));
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

}  // namespace zvdb
