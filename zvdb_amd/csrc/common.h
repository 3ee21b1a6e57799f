// common.h -- internal declarations shared by the translation units of libzvdb_hip.so.
// gfx950 only; wave64 is assumed everywhere.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/zvdb_hip.h"

namespace zvdb {

constexpr int kWave = 64;

void set_error(const char *fmt, ...);
// value of a tuning switch (zvdb_hip_set_tuning; DESIGN.md "Runtime switches"), or nullptr; the pointer stays valid for a few further calls on this thread
const char *tuning(const char *name);

#define ZV_HIP(expr)                                                                  \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            zvdb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                            __FILE__, __LINE__);                                      \
            return _e == hipErrorOutOfMemory ? ZVDB_ERR_OOM : ZVDB_ERR_HIP;           \
        }                                                                             \
    } while (0)

#define ZV_CHECK(expr)                 \
    do {                               \
        int _rc = (expr);              \
        if (_rc != ZVDB_OK) return _rc; \
    } while (0)

inline uint32_t round_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }
inline uint64_t round_up64(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

// splitmix-style counter hash shared with oracle/zvdb_oracle.c:orc_mix64
__host__ __device__ inline uint64_t mix64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// monotone map of an fp32 (any sign) onto u32: a < b  <=>  key(a) < key(b)
__host__ __device__ inline uint32_t f32_key(float f) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    __builtin_memcpy(&u, &f, 4);
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float f32_unkey(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}

// OCP fp4 E2M1: magnitudes {0, .5, 1, 1.5, 2, 3, 4, 6}, bit 3 = sign.  Nearest value, ties to the smaller
// magnitude; ONE function for rows and queries so equal inputs get equal codes.
__host__ __device__ inline uint32_t fp4_e2m1_code(float y) {
    const float a = y < 0.0f ? -y : y;
    const uint32_t m = (uint32_t)(a > 0.25f) + (uint32_t)(a > 0.75f) + (uint32_t)(a > 1.25f) + (uint32_t)(a > 1.75f) +
                       (uint32_t)(a > 2.5f) + (uint32_t)(a > 3.5f) + (uint32_t)(a > 5.0f);
    return m | (y < 0.0f ? 8u : 0u);
}
__host__ __device__ inline float fp4_e2m1_sq(uint32_t code) {  // value^2 of a code
    const uint32_t m = code & 7u;
    const float v = m < 4u ? 0.5f * (float)m : m == 4u ? 2.0f : m == 5u ? 3.0f : m == 6u ? 4.0f : 6.0f;
    return v * v;
}
// the shard's elements are mapped so that mean +- sqrt(3) sigma (= the whole range of uniformly spread data) lands on
// [-kFp4Amax, kFp4Amax]: |y| > 5 rounds to 6, (3.5, 5] to 4; with 4.5 the quantisation error of uniformly spread data is
// smallest (DESIGN.md "fp4 screen": recall sweep), and bell-shaped data uses all 15 values (data_kernels.hip:fp4_range)
constexpr float kFp4Amax = 4.5f;

struct PinnedBuf {
    void *ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace zvdb

// Device-resident index.  Layout (DESIGN.md "Data layout in HBM"):
//   rows   : fp32 [cap][ld]   row-major, ld = round_up(dim, 32), zero padded (absent with ZVDB_FLAG_BF16_ROWS)
//   rows_h : bf16 [cap][ldh]  row-major, ldh = round_up(dim, 64): optional mirror (ZVDB_FLAG_BF16_MIRROR), or THE row storage
//            (ZVDB_FLAG_BF16_ROWS: every exact kernel reads it, widening to fp32; row_types.h)
//   norm   : fp32 [cap]       sum v^2 over the fp32 row   (MFMA screen only)
//   norm_h : fp32 [cap]       sum v^2 over the bf16 row
//   adj    : u32  [cap][gstride]  layer-0 lists, deg : u32 [cap]
struct zvdb_hip_index {
    std::mutex mu;
    int device = 0;
    uint32_t dim = 0, ld = 0, ldh = 0;
    uint32_t cu_count = 0;  // compute units of the index's device (the graph search plans its queries per CU with it)
    uint32_t metric = 0, m = 16, ef_construction = 200, gstride = 16, flags = 0;
    uint64_t count = 0, cap = 0;
    float *rows = nullptr;
    uint16_t *rows_h = nullptr;
    float *norm = nullptr;
    float *norm_h = nullptr;
    uint32_t *adj = nullptr;
    uint32_t *deg = nullptr;
    // EVERY layer on the device (zvdb_hip_insert_batch): each node's level and first upper-layer list slot, and the
    // upper-layer lists themselves -- node's layer L >= 1 list is row slot[node] + L - 1 of adj_up [up_cap][gstride]
    uint32_t *lvl = nullptr, *slot = nullptr;       // [cap]
    uint32_t *adj_up = nullptr, *deg_up = nullptr;  // [up_cap][gstride], [up_cap]
    uint64_t up_cap = 0, up_count = 0;
    uint32_t max_level = 0;  // over the nodes inserted through zvdb_hip_insert_batch / of the bulk graph
    bool bulk_graph = false;  // zvdb_hip_build_graph ran: every layer is the bulk graph's, bulk_top its entry node
    uint32_t bulk_top = 0;
    uint64_t bulk_count = 0;  // rows the bulk graph was built over: ZVDB_ENTRY_DESCEND is refused once count differs (api.hip: bulk_graph_valid)
    uint32_t *tie_buf = nullptr;  // [tie_cap] per-query "equal minima" flags of k_search_compat_fast
    uint64_t tie_cap = 0;
    hipEvent_t tie_event = nullptr;  // end of the last launch pair that used tie_buf: the next one (any stream) waits for it
    bool tie_event_set = false;
    // int8 scalar-quantised mirror for ZVDB_BF_I8 (lazy; rebuilt when rows were added)
    int8_t *rows_q = nullptr;   // [cap_q][ldq], ldq = round_up(dim, 128)
    int32_t *norm_q = nullptr;  // [cap_q] sum q^2
    uint32_t *q_range = nullptr;  // device: {key(min), key(max)} of all elements
    uint32_t ldq = 0;
    uint64_t cap_q = 0, count_q = 0;
    float q_lo = 0.0f, q_inv = 1.0f;  // q = clamp(rint((x - q_lo) * q_inv) - 128) (L2) / rint(x * q_inv) (dot)
    // the same int8 codes in MFMA fragment order for the barrier-free stream kernel (bf_i8_stream.hip):
    // [group of 32 rows][k-step of 32 elements][lane 0..63][16 B]; rows padded to a multiple of 256
    unsigned char *rows_qf = nullptr;
    int32_t *norm_qf = nullptr;  // [cap_qf], INT_MAX past count
    uint64_t cap_qf = 0, count_qf = 0;
    // bf16 mirror in MFMA fragment order (Layout16, bf_bf16_stream.hip) for the register-resident-query stream kernel of
    // ZVDB_BF_BF16: [block of 16 rows][k-step of 32 elements][lane 0..63][16 B]; rows padded to a multiple of 1536
    unsigned char *rows_hf = nullptr;
    float *norm_hf = nullptr;  // [cap_hf + 64] sum of the squared bf16 values (0: dot / cosine), +inf past count
    uint64_t cap_hf = 0, count_hf = 0;
    // fp32 rows in MFMA fragment order for the fp32 stream screen (bf_i8_stream.hip, MODE 0; dim <= 128, lazy): [group of 32 rows][KS][64][16 B]
    unsigned char *rows_ff = nullptr;
    float *norm_ff = nullptr;  // [cap_ff] sum v^2 (0: dot / cosine), +inf past count
    uint64_t cap_ff = 0, count_ff = 0;
    // moments of the rows behind the register-query fp32 scan's guessed thresholds (bf_f32_regq.hip: k_rq_guess): E[n], Var(n), mu, Cov(n, v), Cov(v)
    float *kstat = nullptr;
    double *kstat_acc = nullptr;
    uint64_t kstat_count = ~0ull;
    // fp4 (E2M1) scalar-quantised mirror for ZVDB_BF_FP4 (lazy, like the int8 one)
    unsigned char *rows_p = nullptr;  // [cap_p][ldp] bytes, 2 codes per byte (low nibble = even element), ldp = round_up(dim, 256) / 2
    float *norm_p = nullptr;          // [cap_p] sum of the squared fp4 VALUES (exact in fp32)
    uint32_t ldp = 0;
    uint64_t cap_p = 0, count_p = 0;
    float p_mid = 0.0f, p_scale = 1.0f;  // code = fp4_e2m1_code((x - p_mid) * p_scale)
    // the same codes in MFMA fragment order for the barrier-free stream kernel (bf_fp4_stream.hip):
    // [group of 32 rows][k-step of 64 elements][lane 0..63][16 B]; rows padded to a multiple of 256
    unsigned char *rows_pf = nullptr;
    float *norm_pf = nullptr;         // [cap_pf], +inf past count
    uint64_t cap_pf = 0, count_pf = 0;
    // guessed-threshold pipeline: a repair (guess too tight for some query) raises a flag in host-mapped memory; the next
    // calls loosen the guess by guess_adapt standard deviations (no synchronisation: the flag is read whenever it has arrived)
    uint32_t *guess_flag_host = nullptr;  // hipHostMalloc'ed, mapped
    uint32_t *guess_flag_dev = nullptr;   // its device alias
    mutable uint32_t *ef_hint_dev = nullptr;   // [16] largest visited-set fill seen per ef bucket (ef_search.hip: kEfTight)
    mutable uint32_t *ef_hint_host = nullptr;  // pinned copy, refreshed asynchronously after the first ef searches and then every 8th (a 64-byte
                                               // copy is ~6 us of the launch stream; the hint only ever grows and is read by LATER calls)
    mutable uint32_t ef_hint_calls = 0;
    unsigned long long *spec_wpos0 = nullptr, *spec_wposU = nullptr;  // speculative insert: per-list write stamps (layer 0 / upper slots)
    uint64_t spec_cap0 = 0, spec_capU = 0;
    uint32_t spec_epoch = 0;
    mutable uint32_t *ef_visit_buf = nullptr;  // visited sets in global memory for searches whose safe table does not fit the LDS
    mutable size_t ef_visit_bytes = 0;
    mutable uint32_t *ef_net_buf = nullptr;  // the net under the LDS passes: [work list: ef_net_list_words][kEfNetBlocks tables]
    mutable size_t ef_net_list_words = 0, ef_net_table_words = 0;
    mutable hipEvent_t ef_event = nullptr;   // end of the last ef search that used the buffers above: the next one (any stream) waits for it
    mutable bool ef_event_set = false;
    mutable hipStream_t ef_event_stream = nullptr;   // the stream that event was recorded on
    mutable bool ef_net_dirty = false;               // a call is between its first LDS pass and its net launch (or failed there): the net's header words may be non-zero
    float guess_adapt = 0.0f;
    // bf16 stream scan: the rows are split over the 8 XCDs in proportion to the speed each one showed in earlier launches (their
    // clocks differ by a few per cent under the power limit).  Every workgroup leaves its duration in host-mapped memory; the
    // next launch reads whatever has arrived (no synchronisation) -- bf_bf16_stream.hip: bh_balance
    mutable uint32_t *bal_host = nullptr, *bal_dev = nullptr;  // [4096] durations (100 MHz ticks) of the last scan's workgroups
    mutable float bal_w[8] = {1, 1, 1, 1, 1, 1, 1, 1};        // relative share of the units per XCD
    mutable uint32_t bal_hist[256][9] = {};  // the splits (units per XCD, prefix form) of the last 255 scans, by launch tag (1 .. 255)
    mutable uint32_t bal_nb = 0, bal_seq = 0;  // workgroups of those scans; tag of the last one
    // zvdb_hip_insert_layer0: host-mapped [row | result] block -- the kernels read the point and write the result through
    // it, so an insert submits no copy-engine transfers (each costs ~10 us for a few hundred bytes)
    void *ins_host = nullptr, *ins_dev = nullptr;
    size_t ins_bytes = 0;
    uint32_t ins_seq = 0;
    bool own_stream_busy = false;  // a fused insert left its finalize kernel running on h->stream (see settle_own_stream)
    uint32_t screen_survivors = 0;  // zvdb_hip_index_set_screen_survivors: minimum k' of the fp4 screen (0 = from k)
    uint32_t ef_width = 1;     // ZVDB_OPT_EF_SEARCH_WIDTH: nodes the ef-mode beam search expands per step (1 .. 4)
    uint32_t shard_count = 1;  // zvdb_hip_index_set_shard_count: the approximate screen keeps its share of the survivors
    // exactness certificate of the brute-force screens (bf_cert.h): per screen mode {largest residual |x' - x|, largest |x'|} of the
    // mirror (device words; data_kernels.hip: ensure_cert_stats), valid while cert_count[mode] == count
    uint32_t *cert_dev = nullptr;  // [4][2] float bit patterns
    uint64_t cert_count[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    hipEvent_t cert_event = nullptr;   // behind the last k_cert_stats: certified searches on other streams wait for it before they read the words
    bool cert_event_set = false;
    bool cert_on = true;           // ZVDB_OPT_CERTIFY
    uint32_t cert_level = 0;       // survivors kept per query: 0 = as few as the route allows, 1: 64, 2: 128, 3: 256 -- raised when a batch needed the exact scan
    uint32_t cert_level_min = 0;   // ZVDB_OPT_CERT_MIN_LEVEL
    bool cert_force = false;       // ZVDB_OPT_CERT_FORCE_REPAIR
    uint32_t cert_clean = 0;       // batches since the last repair at this level (lowers the level again after a long clean run)
    uint32_t *cert_host = nullptr, *cert_hdev = nullptr;  // host-mapped {flag raised by a failing query, repaired queries (cumulative)}
    uint64_t cert_queries = 0, cert_batches = 0;          // queries / batches that went through the certificate
    float *stage_f32 = nullptr;  // ZVDB_FLAG_BF16_ROWS: bounded fp32 staging chunk of the add paths (rows are stored as bf16 only)
    uint64_t stage_rows = 0;
    float *qrow_f32 = nullptr;   // ZVDB_FLAG_BF16_ROWS: one row widened to fp32 (a row of the index used as a query)
    hipStream_t stream = nullptr;
    // pinned host staging + device scratch for the host-pointer entry points
    zvdb::PinnedBuf pin;
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // timing hook (zvdb_hip_timing_*)
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
};

namespace zvdb {

// The graph builders' self-kNN passes (k = degree + 1 up to 97 of <= 256 survivors: too few for the certificate's bound on
// high-dimensional rows, so nearly every row would take the exact scan) run the screens UNCERTIFIED, as in rounds 1-3: what such a
// pass can get wrong is a candidate at the tail of a node's list, i.e. an edge of the graph -- never a search result, and the
// certificate exists for those.
struct CertPause {
    zvdb_hip_index *h;
    bool was;
    explicit CertPause(zvdb_hip_index *hh) : h(hh), was(hh->cert_on) { h->cert_on = false; }
    ~CertPause() { h->cert_on = was; }
};

int ensure_pin(zvdb_hip_index *h, size_t bytes);
int ensure_scratch(zvdb_hip_index *h, size_t bytes);
int ensure_capacity(zvdb_hip_index *h, uint64_t want);
void timing_begin(zvdb_hip_index *h, hipStream_t s);
void timing_end(zvdb_hip_index *h, hipStream_t s);

// ---- exact_kernels.hip, ef_search.hip, rerank.hip (compiled with -ffp-contract=off) ---
int launch_exact_distances(const zvdb_hip_index *h, const float *d_query, const uint32_t *d_ids,
                           uint32_t n, float *d_out, hipStream_t s);
// result block of k_insert_layer0 (device scratch -> pinned host memory)
struct InsertL0Out {
    uint32_t ep;        // layer-0 arg-min the new node was connected to
    float ep_dist;
    uint32_t owner_n;   // length of the new node's list after the connect (0 or 1 unless it already had entries)
    uint32_t ep_n;      // length of ep's list after connect + shrink
    unsigned long long evals;  // distance evaluations of the greedy scan (the counter zvdb_hip_greedy_layer0 reports)
    uint32_t error;     // 1: a list does not fit the device stride
    uint32_t seq;       // written LAST (system scope): the host polls it instead of paying a stream-synchronise wake-up
    // followed by uint32_t ep_list[gstride], owner_list[gstride]
};
size_t insert_l0_out_bytes(const zvdb_hip_index *h);
// k_insert_full: per-insert parameters (host-computed) and per-(insert, layer) result records of 4 + gstride words
struct InsertParam { uint32_t level, slot, rec_off, maxlev_pre; };
int launch_insert_full(const zvdb_hip_index *h, uint32_t first_owner, uint32_t n, const float *points, const InsertParam *params,
                       uint32_t *recs, uint32_t *seq_dst, uint32_t seq, hipStream_t s);
int launch_copy_row(const float *src, float *dst, uint32_t dim, hipStream_t s);
// one speculative round of B inserts (exact_kernels.hip: k_insert_spec / _check / _commit); *cmin (device, preset to B) <- points committed
int launch_insert_spec_round(const zvdb_hip_index *h, uint32_t first_owner, uint32_t B, const float *points, const InsertParam *params,
                             uint32_t *recs, unsigned long long *wpos0, unsigned long long *wposU, uint32_t epoch, uint32_t *rset,
                             uint32_t *rcnt, uint32_t *cmin, hipStream_t s);
int launch_insert_spec_loop(const zvdb_hip_index *h, uint32_t first_owner, uint32_t n_chunk, uint32_t B, const float *points,
                            const InsertParam *params, uint32_t *recs, unsigned long long *wpos0, unsigned long long *wposU, uint32_t *rset,
                            uint32_t *rcnt, uint32_t *ctl, uint32_t epoch, hipStream_t s);
uint32_t insert_spec_rmax();
// dst[0..n) <- src[0..n) (64-bit words), then *seq_dst = seq at system scope: publishes a result block to a polling host
int launch_publish(unsigned long long *dst, const unsigned long long *src, uint32_t n, uint32_t *seq_dst, uint32_t seq, hipStream_t s);
int ensure_ins_block(zvdb_hip_index *h, size_t bytes);
int launch_insert_layer0(const zvdb_hip_index *h, uint32_t owner, uint32_t ep, void *d_out, uint32_t seq, const float *point_src,
                         hipStream_t s, uint32_t n_batch = 1, uint32_t out_stride = 0);
int launch_greedy_layer0(const zvdb_hip_index *h, uint32_t owner, uint32_t *d_ep_io,
                         float *d_dist_io, unsigned long long *d_evals, hipStream_t s);
int launch_search_compat(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t k,
                         uint32_t entry, uint64_t id_offset, uint64_t *d_ids, float *d_dist,
                         uint32_t *d_count, unsigned long long *d_stats, hipStream_t s, uint32_t *d_tie = nullptr /* [nq] scratch: enables the unordered-frontier kernel */);
// one query, one launch: counters cleared, search (unordered frontier, heap emulation on ties), counters + sequence word published to host-mapped memory
int launch_search_compat_single(const zvdb_hip_index *h, const float *d_query, uint32_t k, uint32_t entry, uint64_t *d_ids, float *d_dist,
                                uint32_t *d_count, unsigned long long *d_stats, uint32_t *d_tie, unsigned long long *host_stats, uint32_t *seq_dst,
                                uint32_t seq, hipStream_t s);
void ef_hint_reset(const zvdb_hip_index *h, hipStream_t s);
int launch_search_ef(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t k,
                     uint32_t ef, uint32_t entry, uint64_t id_offset, uint64_t *d_ids,
                     float *d_dist, uint32_t *d_count, unsigned long long *d_stats, hipStream_t s,
                     uint32_t *d_entries = nullptr /* per-query entry nodes (buffer of nq words; with `descend` the call fills it) */,
                     bool reserve_only = false /* allocate what a launch of this shape needs, launch nothing */,
                     bool descend = false /* `entry` is the bulk graph's top node: descend through the upper layers first (in the first pass of
                                             the register-beam kernel, or as launch_descend ahead of the search) */);
int launch_descend(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t top, uint32_t *d_entries,
                   unsigned long long *d_stats, hipStream_t s);
// exact re-rank of candidate lists: cand[nq][kc] u32 (0xFFFFFFFF = empty) -> top-k by (dist,id)
struct CertArgs;  // bf_cert.h
int launch_rerank(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, const uint32_t *d_cand,
                  uint32_t kc, uint32_t k, uint64_t id_offset, uint64_t *d_ids, float *d_dist,
                  uint32_t *d_count, hipStream_t s,
                  uint32_t kp_valid = 0 /* > 0: the lists hold at most this many candidates (compacted to the front) */,
                  const CertArgs *cert = nullptr /* lists sorted by screen key + certificate (bf_cert.h) */);
// exact_scan.hip: for every query on the list (*nflag entries) the exact top-k over ALL rows in the reference's arithmetic,
// written over the query's results; scratch: >= exact_scan_scratch_bytes(nq, k); a no-op launch when the list is empty
uint64_t exact_scan_scratch_bytes(uint32_t nq, uint32_t k);
int launch_exact_scan(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t k, uint64_t id_offset, const uint32_t *nflag,
                      const uint32_t *flist, void *scratch, uint64_t scratch_bytes, uint64_t *d_ids, float *d_dist, uint32_t *d_count,
                      uint32_t *host_repaired, hipStream_t s);

// ---- data_kernels.hip -------------------------------------------------------
int launch_fill_synthetic(float *d_out, uint64_t nrows, uint32_t dim, uint32_t ld, uint64_t seed,
                          uint64_t first_row, hipStream_t s);
int launch_pack_rows(const float *d_src, uint32_t dim, float *d_dst, uint32_t ld, uint64_t n,
                     hipStream_t s);
int launch_normalize_rows(float *d_rows, uint32_t dim, uint32_t ld, uint64_t n, hipStream_t s);
int launch_finalize_rows(const zvdb_hip_index *h, uint64_t first, uint64_t n, hipStream_t s);
int launch_sanitize_loaded_rows(const zvdb_hip_index *h, uint64_t n, hipStream_t s);  // zvdb_hip_load: zero padding, bf16 rows' norms from the rows
int launch_store_rows_bf16(const zvdb_hip_index *h, const float *d_stage, uint64_t first, uint64_t n, hipStream_t s);  // ZVDB_FLAG_BF16_ROWS
int ensure_i8_mirror(zvdb_hip_index *h, hipStream_t s);
int ensure_fp4_mirror(zvdb_hip_index *h, hipStream_t s);
int fp4_range(zvdb_hip_index *h, hipStream_t s);  // sets p_mid / p_scale from the rows' value range
int ensure_cert_stats(zvdb_hip_index *h, uint32_t mode, hipStream_t s);
int launch_results_to_adj(const zvdb_hip_index *h, const uint64_t *d_ids, const uint32_t *d_cnt, uint32_t k,
                          uint64_t r0, uint32_t n, uint32_t degree, hipStream_t s);
int launch_validate_graph(const zvdb_hip_index *h, uint64_t n, uint64_t up_count, uint32_t max_level, uint32_t *d_bad, hipStream_t s);
int launch_merge_topk(const uint64_t *d_ids, const float *d_dist, const uint32_t *d_count,
                      uint32_t G, uint32_t nq, uint32_t k, uint64_t *o_ids, float *o_dist,
                      uint32_t *o_count, hipStream_t s);

int launch_merge_topk_strided(const uint64_t *d_ids, uint64_t rs_ids, const float *d_dist, uint64_t rs_dist,
                              const uint32_t *d_count, uint64_t rs_cnt, uint32_t G, uint32_t nq, uint32_t k,
                              uint64_t *o_ids, float *o_dist, uint32_t *o_count, hipStream_t s);

// ---- bf_i8_stream.hip --------------------------------------------------------
bool i8s_eligible(const zvdb_hip_index *h);
uint32_t i8s_ksteps(uint32_t dim);
uint64_t i8s_query_bytes(const zvdb_hip_index *h, uint32_t nq_pad);
int i8_range(zvdb_hip_index *h, hipStream_t s);  // sets q_lo / q_inv from the rows' value range (data_kernels.hip)
int ensure_i8_frag_mirror(zvdb_hip_index *h, hipStream_t s);
uint32_t f32s_ksteps(uint32_t dim);
bool f32s_eligible(const zvdb_hip_index *h);
uint64_t f32s_query_bytes(const zvdb_hip_index *h, uint32_t nq_pad);
int ensure_f32_frag_mirror(zvdb_hip_index *h, hipStream_t s);
int launch_f32s_prep_queries(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t nq_pad, unsigned char *q_op, hipStream_t s);
int launch_f32s_scan(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                     uint64_t *cand, uint64_t *partial, uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot,
                     uint32_t *pcnt, const uint32_t *gate, uint32_t kpv, hipStream_t s);
// the register-resident-query fp32 scan (bf_f32_regq.hip): 256-query tiles, the same fragment-order mirror and packed queries
bool f32rq_eligible(const zvdb_hip_index *h);
int ensure_f32rq_stats(zvdb_hip_index *h, hipStream_t s);
int launch_f32rq_guess(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t nq_pad, unsigned char *q_op, float z, uint32_t *tau_g,
                       uint32_t *verify_flag, hipStream_t s);
int launch_f32rq_scan(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint64_t *cand, uint64_t *partial,
                      uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot, uint32_t *pcnt, const uint32_t *gate, uint32_t kpv,
                      hipStream_t s);
int launch_f32s_guess(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                      float *mom, float z, uint32_t *tau_g, uint32_t *verify_flag, hipStream_t s);
int launch_i8s_prep_queries(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t nq_pad, unsigned char *q_op,
                            hipStream_t s);
int launch_i8s_scan(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                    uint64_t *cand, uint64_t *partial, uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot,
                    uint32_t *pcnt, const uint32_t *gate, uint32_t kpv, hipStream_t s);
int launch_i8s_guess(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt, float *mom,
                     float z, uint32_t *tau_g, uint32_t *verify_flag, hipStream_t s);

uint32_t fp4s128_ksteps(uint32_t dim);
bool fp4s128_eligible(const zvdb_hip_index *h);
int launch_fp4s128_scan(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                        uint64_t *cand, uint64_t *partial, uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot,
                        uint32_t *pcnt, const uint32_t *gate, uint32_t kpv, hipStream_t s);
int launch_fp4s128_guess(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                         float *mom, float z, uint32_t *tau_g, uint32_t *verify_flag, hipStream_t s);

// ---- graph_build.hip ---------------------------------------------------------
uint32_t bulk_level(uint64_t seed, uint64_t id);
int launch_rows_to_f32(const zvdb_hip_index *h, uint64_t first, uint64_t n, float *d_out, hipStream_t s);
int build_bulk_graph(zvdb_hip_index *h, uint32_t M, uint32_t ef_construction, uint32_t mode, uint64_t level_seed,
                     int (*ensure_upper_fn)(zvdb_hip_index *, uint64_t));

// ---- bf_bf16_stream.hip ------------------------------------------------------
bool bf16s_eligible(const zvdb_hip_index *h);
uint32_t bf16s_ksteps(uint32_t dim);
uint64_t bf16s_query_bytes(const zvdb_hip_index *h, uint32_t nq_pad);
uint64_t bf16s_moment_bytes(uint32_t QT, uint32_t S);
int ensure_bf16_frag_mirror(zvdb_hip_index *h, hipStream_t s);
int launch_bf16s_prep_queries(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t nq_pad, unsigned char *q_op, hipStream_t s);
int launch_bf16s_scan(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint64_t *cand,
                      uint64_t *partial, uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot, uint32_t *pcnt, const uint32_t *gate,
                      uint32_t kpv, hipStream_t s);
int launch_bf16s_guess(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, float *mom, float z,
                       uint32_t *tau_g, uint32_t *verify_flag, hipStream_t s);

// ---- bf_fp4_stream.hip -------------------------------------------------------
bool fp4s_eligible(const zvdb_hip_index *h);
uint32_t fp4s_ksteps(uint32_t dim);
uint64_t fp4s_query_bytes(const zvdb_hip_index *h, uint32_t nq_pad);
int ensure_fp4_frag_mirror(zvdb_hip_index *h, hipStream_t s);
int launch_fp4s_prep_queries(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t nq_pad,
                             unsigned char *q_op, hipStream_t s);
int launch_fp4s_scan(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                     uint64_t *cand, uint64_t *partial, uint32_t *tau_g, uint32_t kp, uint32_t t_first, uint32_t list_base,
                     uint32_t Stot, uint32_t *pcnt, const uint32_t *gate, uint32_t kpv /* list width: 64 or 128 */, hipStream_t s);
uint64_t fp4s_moment_bytes(uint32_t QT, uint32_t S);
int launch_fp4s_guess(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                      float *mom, float z, uint32_t *tau_g, uint32_t *verify_flag, hipStream_t s);

// ---- bruteforce.hip ---------------------------------------------------------
uint64_t bruteforce_workspace_bytes(const zvdb_hip_index *h, uint32_t nq, uint32_t k, uint32_t mode);
int launch_bruteforce(zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t k,
                      uint32_t mode, uint64_t id_offset, void *d_ws, uint64_t ws_bytes,
                      uint64_t *d_ids, float *d_dist, uint32_t *d_count, hipStream_t s);

}  // namespace zvdb
