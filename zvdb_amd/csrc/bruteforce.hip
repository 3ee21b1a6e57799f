// bruteforce.hip -- K4: batched brute-force Q x V^T on MFMA with a fused, never-materialised per-query top-k'
// selection, followed by a merge of the per-stream lists and (in rerank.hip) the exact fp32 re-rank.  No
// reference symbol exists for this path (a10): north_star names it; it is the recall-guaranteed route.
//
//   screen key        |v'|^2 / 2 - q'.v'  (the |q|^2 term is constant per query; dot / cosine: -q'.v')
//   q'.v'             fp32 MFMA v_mfma_f32_32x32x2_f32 on the fp32 rows, bf16 / int8 / fp4 MFMA on mirrors
//   selection         per (query tile, corpus stream) workgroup: exact k' best (key, id) pairs by radix select
//                     (bf_select.h), shared thresholds across workgroups, "ties by lower id" exact
//
// This file holds the STAGED scan (k_bf_scan5: 256 x 256 output tile, 8 waves of 128 x 64, operands through a 2-stage
// LDS-DMA ring of 128-byte row slices), used for the fp32 screen, for small shards of every screen and for rows too
// wide for the stream kernels (bf_bf16_stream.hip, bf_i8_stream.hip, bf_fp4_stream.hip), plus the host-side pipeline
// of every route (launch_bruteforce).  Earlier generations (v1: 128 x 128 register-staged; v4: 64-byte slices, 4-stage
// ring; a one-wave-per-SIMD 128 x 128 wave tile) are in the git history and in profiles/r1/ (A/B numbers).
#include "common.h"
#include "wave_select.h"
#include "bf_select.h"
#include "row_types.h"
#include "bf_cert.h"

namespace zvdb {

namespace {

constexpr int KB = 128;             // bytes of every row consumed per K slice
constexpr int HW = 192;             // prune when a buffer holds more than this

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
// =====================================================================================
constexpr bool kFp4Pipe = false;
static_assert(T2 == 256, "bf_select.h");  //                    // tile edge (queries and corpus rows)
constexpr int CAP2 = HW + T2;                // per-row candidate buffer of the v2 scan

using i32x16 = __attribute__((ext_vector_type(16))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x8 = __attribute__((ext_vector_type(8))) int;
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int MODE> struct AccT { using type = f32x16; };
template <> struct AccT<2> { using type = i32x16; };

// ring geometry shared by the 64-byte-slice variants
constexpr int KB3 = 64;                       // bytes of every row per slice
constexpr int STAGE3 = 2 * T2 * KB3;          // A + B = 32 KiB
constexpr int NSTAGE3 = 4;

constexpr int PTHRESH = 32;  // start a prune round once this many sub-buffers crossed the soft mark
constexpr int LDS4_VN = NSTAGE3 * STAGE3;        // vn[4][256]
constexpr int LDS4_TG = LDS4_VN + 4 * T2 * 4;    // tg[4][256]
constexpr int LDS4_TAU = LDS4_TG + 4 * T2 * 4;   // tau[256]
constexpr int LDS4_NK = LDS4_TAU + T2 * 4;       // nkept[256]
constexpr int LDS4_SC = LDS4_NK + T2 * 4;        // subcnt[256][4]
constexpr int LDS4_FLAG = LDS4_SC + T2 * NSUB * 4;
constexpr int LDS4_SEL = LDS4_FLAG + 16;          // 8 waves x (histogram + compaction stage)
constexpr int LDS4 = LDS4_SEL + 8 * SEL_SCRATCH;

// k_bf_scan5: the v4 kernel generalised over the wave tile: NI = 2 -> 8 waves of 128 x 64 (two per
// SIMD); NI = 4 -> 4 waves of 128 x 128, ONE per SIMD with the whole 512-register file (256 accumulators):
// a third fewer LDS fragment bytes per MFMA and half the barrier participants.
// KPV: width of the kept lists (64; 128 = two entries per lane, carries k <= 112).
template <int MODE, bool DBG, int NI, int KBV, int KPV = KP>
__global__ __launch_bounds__(NI == 4 ? 256 : 512, NI == 4 ? 1 : 2) void k_bf_scan5(
    const unsigned char *__restrict__ Q /* query operand */, const unsigned char *__restrict__ V /* corpus operand */,
    uint32_t row_bytes, uint32_t KT /* KBV-byte slices per row */, const void *__restrict__ vnorm, uint32_t N,
    uint32_t nq, int alpha_i, uint32_t QT, uint32_t S, uint32_t NT, uint64_t *__restrict__ cand,
    uint64_t *__restrict__ partial, uint32_t *__restrict__ tau_g_all, uint32_t kp, uint32_t t_first,
    uint32_t list_base, uint32_t Stot, uint32_t *__restrict__ pcnt, uint32_t rot_tab, uint32_t dbg_in,
    unsigned long long *__restrict__ dbgbuf, const uint32_t *__restrict__ gate /* non-null: a no-op unless *gate (the repair scan) */) {
    if (gate && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) return;
    const uint32_t dbg = DBG ? dbg_in : 0u;  // all diagnostics compile out of the production instantiation
    using acc_t = typename AccT<MODE>::type;
    constexpr bool INT = MODE == 2;
    constexpr int NW = 16 / NI;          // waves per workgroup: 8 (wave tile 128x64) or 4 (128x128, one wave per SIMD)
    constexpr int NWN = NW / 2;          // waves along the query axis
    constexpr int NCC = KBV / 32;        // 32-byte k-steps (one MFMA deep) per slice: 2 or 4
    constexpr int LPR = KBV / 16;        // lanes (16-B chunks) per row of a slice
    constexpr int RPP = 1024 / KBV;      // rows per 1-KiB DMA piece: 16 (64-B rows) or 8 (full 128-B lines)
    constexpr int NPC = 256 / RPP / NW;  // DMA pieces per operand per wave per slice
    constexpr int STAGE = 2 * T2 * KBV;  // 32 or 64 KiB
    constexpr int NST = 131072 / STAGE;  // ring stages: 4 or 2
    constexpr bool PIPE = MODE == 2 || (MODE == 3 && kFp4Pipe) || NI == 4;  // defer a slice's last k-step behind the next barrier (bf16/fp32 spill with it)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *vn_lds = reinterpret_cast<uint32_t *>(smem + LDS4_VN);
    uint32_t *tg_lds = reinterpret_cast<uint32_t *>(smem + LDS4_TG);
    float *tau = reinterpret_cast<float *>(smem + LDS4_TAU);
    uint32_t *nkept = reinterpret_cast<uint32_t *>(smem + LDS4_NK);
    uint32_t *subcnt = reinterpret_cast<uint32_t *>(smem + LDS4_SC);
    uint32_t *pflag = reinterpret_cast<uint32_t *>(smem + LDS4_FLAG);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;  // wm: corpus half of the tile, wn: query slice

    const uint32_t b = blockIdx.x;
    const uint32_t x = b & 7, jb = b >> 3;
    const uint32_t qt = jb % QT;
    const uint32_t s = (jb / QT) * 8 + x;
    const uint32_t per = (NT + S - 1) / S;
    // this launch covers tiles [t_first, t_first + NT) of the corpus (NT = tiles of this pass)
    const uint32_t t_begin = t_first + min(s * per, NT), t_end = min(t_begin + per, t_first + NT);
    const uint32_t ntiles = t_end - t_begin;
    uint32_t *tau_g = tau_g_all ? tau_g_all + (size_t)qt * T2 : nullptr;

    uint64_t *kept = cand + (size_t)b * (KPV == KP ? BLK4_ENTRIES : BLKV_ENTRIES);
    uint64_t *subs = kept + (size_t)T2 * KPV;
    for (int tq = tid; tq < T2; tq += NW * 64) {
        const int tid_q = tq;
        const bool live = (qt * T2 + tid_q) < nq;
        // screen key of (query, row) = -(q.v - |v|^2/2) = d'/2 (L2) or -q.v (dot): tau lives in that domain.
        // int: live rows satisfy acc >= -2^29 (dim <= 16384), rows past N start at -2^30: tau0 = 0x30000000
        // lets every live row pass and no padded one.
        // start from the shared threshold of the previous pass (if any): pass <=> key < next_up(tau_G)
        const uint32_t g = (tau_g && live) ? __hip_atomic_load(tau_g + tid_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                           : 0xFFFFFFFFu;
        if (INT) {
            int32_t t0 = live ? 0x30000000 : -0x7FFFFFFF;
            if (g != 0xFFFFFFFFu) { const int32_t gv = (int32_t)(g ^ 0x80000000u); t0 = min(t0, gv == 0x7FFFFFFF ? gv : gv + 1); }
            reinterpret_cast<int32_t *>(tau)[tid_q] = t0;
        } else {
            float t0 = live ? __builtin_inff() : -__builtin_inff();
            if (g != 0xFFFFFFFFu) t0 = fminf(t0, f32_unkey(g + 1));
            tau[tid_q] = t0;
        }
        nkept[tid_q] = 0;
    }
    for (int i = tid; i < T2 * NSUB; i += NW * 64) subcnt[i] = 0;
    if (tid == 0) { pflag[0] = 0; pflag[1] = 0; }

    // ---- DMA issue side: A region <- corpus tile, B region <- my query tile ------------------
    const unsigned char *q_base = Q + (size_t)qt * T2 * row_bytes;
    auto swz = [](uint32_t r) -> uint32_t { return KBV == 64 ? ((r >> 2) & 3) : ((r >> 1) & 7); };
    const uint32_t prow = RPP * NPC * wave + lane / LPR;  // + RPP*i : LDS row this lane fills
    uint32_t q_off[NPC], c_off[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const uint32_t r = prow + RPP * i;
        q_off[i] = r * row_bytes + ((((uint32_t)lane % LPR) ^ swz(r)) << 4);
    }
    // the QT workgroups that share a corpus stream (same XCD, same tiles at the same time) walk the k-slices
    // of a tile in ROTATED orders: a slice is then pulled from HBM by one of them and hit in L2 by the others a
    // slice-time later, instead of all of them waiting on the same in-flight L2 miss (tools/dma_l2_bench.hip:
    // 24 -> 36 B/clk per CU).  The dot product does not care about the order of k.
    const uint32_t krot = (rot_tab >> (4 * (qt & 7))) & 15u;
    uint32_t it = t_begin, ikt = 0;
    unsigned char *i_dst = nullptr;
    const unsigned char *i_cp = nullptr, *i_qp = nullptr;
    auto issue_prep = [&](uint32_t stage_idx) {
        const uint32_t tcl = min(it, t_end - 1);
        const uint32_t col0 = tcl * T2;
        if (ikt == 0) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const uint32_t r = prow + RPP * i;
                const uint32_t rc = min(r, N - 1 - col0);
                c_off[i] = rc * row_bytes + ((((uint32_t)lane % LPR) ^ swz(r)) << 4);
            }
            if (tau_g && (NW == 4 || wave >= 4) && it < t_end) {
                __builtin_amdgcn_global_load_lds(
                    (gptr_t)(tau_g + 64 * (wave & 3) + lane),
                    (lptr_t)(reinterpret_cast<unsigned char *>(tg_lds) + ((it - t_begin) & 3) * T2 * 4 + (wave & 3) * 256),
                    4, 0, 16);
            }
            if (vnorm && wave < 4 && it < t_end) {
                const uint32_t c = min(col0 + 64 * wave + lane, N - 1);
                __builtin_amdgcn_global_load_lds(
                    (gptr_t)(reinterpret_cast<const unsigned char *>(vnorm) + (size_t)c * 4),
                    (lptr_t)(reinterpret_cast<unsigned char *>(vn_lds) + ((it - t_begin) & 3) * T2 * 4 + wave * 256), 4, 0, 0);
            }
        }
        i_dst = smem + stage_idx * STAGE;
        uint32_t ks = ikt + krot;
        if (ks >= KT) ks -= KT;
        i_cp = V + (size_t)col0 * row_bytes + (size_t)ks * KBV;
        i_qp = q_base + (size_t)ks * KBV;
        if (++ikt == KT) {
            ikt = 0;
            if (it < t_end) ++it;
        }
    };
    // piece i of the slice prepared by issue_prep: 1 KiB of the corpus tile + 1 KiB of the query tile
    auto issue_piece = [&](int i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(i_cp + c_off[i]), (lptr_t)(i_dst + (NPC * wave + i) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(i_qp + q_off[i]), (lptr_t)(i_dst + T2 * KBV + (NPC * wave + i) * 1024), 16, 0, 0);
    };
    auto issue_slice = [&](uint32_t stage_idx) {
        issue_prep(stage_idx);
#pragma unroll
        for (int i = 0; i < NPC; ++i) issue_piece(i);
    };

    const uint32_t fr = lane & 31, fh = lane >> 5;
    uint32_t f_off[NCC];
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) f_off[cc] = fr * KBV + ((((uint32_t)(2 * cc) + fh) ^ swz(fr)) << 4);

    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
    uint32_t fa_base[NCC], fb_base[NCC];
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) {
        fa_base[cc] = lds0 + (wm * 128) * KBV + f_off[cc];
        fb_base[cc] = lds0 + T2 * KBV + (wn * (32 * NI)) * KBV + f_off[cc];
    }

    // the lane's two queries and its private sub-buffers
    const uint32_t sub_id = wm * 2 + fh;
    uint32_t ql[NI];
    uint64_t *sbuf[NI];
    uint32_t sc_addr[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        ql[ni] = wn * (32 * NI) + ni * 32 + fr;
        sbuf[ni] = subs + ((size_t)ql[ni] * NSUB + sub_id) * SUBCAP;
        sc_addr[ni] = (uint32_t)(size_t)(lptr_t)(subcnt + ql[ni] * NSUB + sub_id);
    }
    const uint32_t rbase = wm * 128 + 4 * fh;  // corpus row of accumulator register (mi, r): rbase + mi*32 + (r&3) + 8*(r>>2)
    const uint32_t pflag_lds_addr = (uint32_t)(size_t)(lptr_t)pflag;

    if (ntiles > 0) {
#pragma unroll
        for (int i = 0; i < NST - 1; ++i) issue_slice(i);
    }
    __syncthreads();  // state arrays initialised (plain LDS stores) before anyone reads them
    uint32_t islice = 0;
    unsigned long long tk0 = 0, t_round = 0, t_phb = 0, t_pha = 0, n_round = 0, t_sync = 0, t_front = 0, t_body = 0, ts0 = 0;
    if (dbg & 32) tk0 = __builtin_amdgcn_s_memtime();

    for (uint32_t t = t_begin; t < t_end; ++t) {
        const uint32_t col0 = t * T2;
        const uint32_t *vn_t = vn_lds + ((t - t_begin) & 3) * T2;
        acc_t acc[4][NI];
        u32x4 fx_a[4], fx_b[NI], fy_a[4], fy_b[NI];
        // MFMAs [lo, hi) of the 4 x NI fragment products of one k-step (constant after unrolling)
        auto mma_some = [&](const u32x4 (&a4)[4], const u32x4 (&b2)[NI], int lo, int hi) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (mi * NI + ni < lo || mi * NI + ni >= hi) continue;
                    if constexpr (MODE == 1) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a4[mi]),
                                                                              __builtin_bit_cast(bf16x8, b2[ni]), acc[mi][ni], 0, 0, 0);
                    } else if constexpr (MODE == 2) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a4[mi]),
                                                                            __builtin_bit_cast(i32x4, b2[ni]), acc[mi][ni], 0, 0, 0);
                    } else if constexpr (MODE == 3) {
                        // a lane's 16-byte chunk = its 32 fp4 k-elements of this 64-deep step (cbsz = blgp = 4: fp4
                        // reads operand registers 0-3 only); E8M0 scale 127 = 2^0 for every 32-element block
                        const i32x4 xa = __builtin_bit_cast(i32x4, a4[mi]), xb = __builtin_bit_cast(i32x4, b2[ni]);
                        const i32x8 wa = {xa.x, xa.y, xa.z, xa.w, 0, 0, 0, 0}, wb = {xb.x, xb.y, xb.z, xb.w, 0, 0, 0, 0};
                        acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, wb, acc[mi][ni], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                    } else {
                        const f32x4 xa = __builtin_bit_cast(f32x4, a4[mi]);
                        const f32x4 xb = __builtin_bit_cast(f32x4, b2[ni]);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.x, xb.x, acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.y, xb.y, acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.z, xb.z, acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.w, xb.w, acc[mi][ni], 0, 0, 0);
                    }
                }
        };
        auto mma8 = [&](const u32x4 (&a4)[4], const u32x4 (&b2)[NI]) { mma_some(a4, b2, 0, 4 * NI); };
        uint32_t thr[NI];   // fp: bits of the accumulator bound; int: tau
        uint32_t cntr[NI], cnt0[NI];

        for (uint32_t kt = 0; kt < KT; ++kt, ++islice) {
            if (dbg & 32) ts0 = __builtin_amdgcn_s_memtime();
            // slice `islice` has landed once at most this wave's (NST-2) youngest slices' DMA ops are pending
            if constexpr (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (NW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (dbg & 32) { const unsigned long long x2 = __builtin_amdgcn_s_memtime(); t_sync += x2 - ts0; ts0 = x2; }
            if (kt == 0) {
                // prune ROUND: batched so that all 8 waves have queries to merge (a round stalls the
                    // whole pipeline); started when enough sub-buffers crossed the soft mark or one is
                    // close to its capacity (a tile adds at most 64 entries to a sub-buffer)
                if (pflag[0] != 0 || pflag[1] >= (uint32_t)PTHRESH) {
                    unsigned long long ts = 0;
                    if (dbg & 32) ts = __builtin_amdgcn_s_memtime();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // everyone's appends are complete
                    __builtin_amdgcn_s_barrier();                     // ... and everyone has read the flags
                    if (tid == 0) { pflag[0] = 0; pflag[1] = 0; }
                    for (int q = wave; q < T2; q += NW) {
                        const u32x4 c4 = *reinterpret_cast<const u32x4 *>(subcnt + q * NSUB);
                        if (max(max(c4.x, c4.y), max(c4.z, c4.w)) > (uint32_t)SUBHW) {
                            if ((dbg & 16) && lane == 0) atomicAdd(&dbgbuf[2], 1ull);
                            prune_query<INT, false, KPV>(kept, subs, tau, nkept, subcnt + q * NSUB, q, lane, kp, tau_g, smem + LDS4_SEL + wave * SEL_SCRATCH);
                        }
                    }
                    __syncthreads();
                    if (dbg & 32) { t_round += __builtin_amdgcn_s_memtime() - ts; n_round += 1; }
                }
                for (int tq = tid; tau_g && tq < T2; tq += NW * 64) {  // fold the shared threshold: pass <=> key < min(tau_local, next_up(tau_G)), non-strict on tau_G
                    const int tid_f = tq;
                    const uint32_t g = tg_lds[((t - t_begin) & 3) * T2 + tid_f];
                    if (g != 0xFFFFFFFFu) {
                        uint32_t nb;
                        bool tighter;
                        if (INT) {
                            const int32_t gv = (int32_t)(g ^ 0x80000000u);
                            const int32_t bv = gv == 0x7FFFFFFF ? gv : gv + 1;
                            tighter = bv < reinterpret_cast<int32_t *>(tau)[tid_f];
                            nb = (uint32_t)bv;
                        } else {
                            const float bv = f32_unkey(g + 1);
                            tighter = bv < tau[tid_f];
                            nb = __float_as_uint(bv);
                        }
                        if (tighter) {
                            const uint32_t addr = (uint32_t)(size_t)(lptr_t)(tau + tid_f);
                            asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(nb) : "memory");
                        }
                    }
                }
                // per-lane selection state for this tile
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    cntr[ni] = subcnt[ql[ni] * NSUB + sub_id];
                    cnt0[ni] = cntr[ni];
                    // pass <=> key < tau <=> acc > -tau
                    if constexpr (INT) thr[ni] = (uint32_t)(-reinterpret_cast<const int32_t *>(tau)[ql[ni]]);
                    else thr[ni] = __float_as_uint(-tau[ql[ni]]);
                }
                // accumulator init = -|v|^2/2 of the register's corpus row (fp) / -(|v|^2 >> 1) (int; the
                // dropped parity bit perturbs the int screen by < 1 unit and equally for equal rows);
                // rows past N start at -inf / -2^30 and can never pass.  The finished accumulator IS the
                // negated screen key, so the selection test is one compare per element.
                const bool full_tile = col0 + T2 <= N;  // wave-uniform: only the corpus' last tile is ragged
                if (full_tile) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        acc_t init;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            u32x4 v4 = {0, 0, 0, 0};
                            if (vnorm) v4 = *reinterpret_cast<const u32x4 *>(vn_t + rbase + mi * 32 + 8 * g);
#pragma unroll
                            for (int j2 = 0; j2 < 4; ++j2) {
                                if constexpr (INT) init[4 * g + j2] = -((int)v4[j2] >> 1);
                                else init[4 * g + j2] = -0.5f * __uint_as_float(v4[j2]);
                            }
                        }
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = init;
                    }
                } else {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        acc_t init;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const uint32_t rl = rbase + mi * 32 + 8 * g;
                            u32x4 v4 = {0, 0, 0, 0};
                            if (vnorm) v4 = *reinterpret_cast<const u32x4 *>(vn_t + rl);
#pragma unroll
                            for (int j2 = 0; j2 < 4; ++j2) {
                                const bool ok = (col0 + rl + j2) < N;
                                if constexpr (INT) init[4 * g + j2] = ok ? -((int)v4[j2] >> 1) : -0x40000000;
                                else init[4 * g + j2] = ok ? -0.5f * __uint_as_float(v4[j2]) : -__builtin_inff();
                            }
                        }
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = init;
                    }
                }
            }
            // generic k-step pipeline over the slice's NCC 32-byte steps: fragments of step j+1 are read
            // (alternating register sets X / Y) while step j is multiplied; the last step's MFMAs are
            // deferred behind the next slice's barrier, reads and DMA issue (PIPE) or run at once.
            // The fragment reads and their waits are hand-written: hipcc waits with lgkmcnt(0) in front of a
            // k-step's MFMAs, i.e. also for the NEXT step's reads it has just issued, which serialises LDS
            // latency and MFMA on every other step.  LDS returns in order, so lgkmcnt(<reads of the newer set>)
            // is the exact wait; the registers are tied to the wait so the MFMAs cannot move above it.
            const uint32_t st_off = (islice % NST) * STAGE;
            auto read_set = [&](u32x4 (&fa)[4], u32x4 (&fb)[NI], int cc) {
                const uint32_t aa = st_off + fa_base[cc], ab = st_off + fb_base[cc];
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%5\n\t"
                             "ds_read_b128 %2, %4 offset:%6\n\tds_read_b128 %3, %4 offset:%7"
                             : "=&v"(fa[0]), "=&v"(fa[1]), "=&v"(fa[2]), "=&v"(fa[3])
                             : "v"(aa), "n"(32 * KBV), "n"(64 * KBV), "n"(96 * KBV));
                if constexpr (NI == 2) {
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3"
                                 : "=&v"(fb[0]), "=&v"(fb[1]) : "v"(ab), "n"(32 * KBV));
                } else {
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%5\n\t"
                                 "ds_read_b128 %2, %4 offset:%6\n\tds_read_b128 %3, %4 offset:%7"
                                 : "=&v"(fb[0]), "=&v"(fb[1]), "=&v"(fb[2]), "=&v"(fb[3])
                                 : "v"(ab), "n"(32 * KBV), "n"(64 * KBV), "n"(96 * KBV));
                }
            };
            // wait until `fa/fb` have landed given that `newer` sets of reads were issued after them
            auto wait_set = [&](u32x4 (&fa)[4], u32x4 (&fb)[NI], auto newer) {
                constexpr int cnt = decltype(newer)::value * (4 + NI);
                if constexpr (NI == 2)
                    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]) : "n"(cnt));
                else
                    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]),
                                 "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]) : "n"(cnt));
            };
            using one_t = std::integral_constant<int, 1>;
            using zero_t = std::integral_constant<int, 0>;
            read_set(fx_a, fx_b, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the next slice's DMA goes out first (its latency bounds the slice) with the previous slice's
            // deferred MFMAs between the pieces: a wave blocks on MFMA issue while the pipe is busy, the
            // VMEM issue of the following piece runs in that shadow
            if constexpr (PIPE) {
                if (!(dbg & 256)) issue_prep((islice + NST - 1) % NST);
                if (kt > 0 && !(dbg & 128)) wait_set(fy_a, fy_b, one_t{});
                constexpr int MPP = 4 * NI / NPC;  // MFMAs per DMA piece pair
#pragma unroll
                for (int i = 0; i < NPC; ++i) {
                    if (!(dbg & 256)) issue_piece(i);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt > 0 && !(dbg & 128)) mma_some(fy_a, fy_b, i * MPP, (i + 1) * MPP);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                if (!(dbg & 256)) issue_slice((islice + NST - 1) % NST);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (dbg & 32) { const unsigned long long x2 = __builtin_amdgcn_s_memtime(); t_front += x2 - ts0; ts0 = x2; }
#pragma unroll
            for (int cc = 1; cc < NCC; ++cc) {
                if (cc & 1) read_set(fy_a, fy_b, cc); else read_set(fx_a, fx_b, cc);
                __builtin_amdgcn_sched_barrier(0);
                if (cc & 1) wait_set(fx_a, fx_b, one_t{}); else wait_set(fy_a, fy_b, one_t{});
                if (!(dbg & 128)) { if (cc & 1) mma8(fx_a, fx_b); else mma8(fy_a, fy_b); }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the slice's last set: landed before the next barrier (the stage is then overwritten by DMA)
            wait_set(fy_a, fy_b, zero_t{});
            if (!PIPE && !(dbg & 128)) mma8(fy_a, fy_b);
            if (dbg & 32) t_body += __builtin_amdgcn_s_memtime() - ts0;
            if (dbg & 128) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) asm volatile("" :: "v"(fx_a[mi]), "v"(fy_a[mi]));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) asm volatile("" :: "v"(fx_b[ni]), "v"(fy_b[ni]));
            }
        }
        if constexpr (PIPE) {
            if (!(dbg & 128)) mma8(fy_a, fy_b);  // flush the tile's last k-step
        }

        // ---- fused selection epilogue: registers only on the common path -----------------------
        if (!(dbg & 1)) {
            unsigned long long te0 = 0;
            if (dbg & 32) te0 = __builtin_amdgcn_s_memtime();
            // Phase A: per 32x32 fragment, the lane's maximum over its 16 accumulator registers
            // (v_max3) against the lane's bound: 8 compares + 8 mask tests per tile on the common path.
            unsigned long long fany[4][NI];
            unsigned long long any = 0;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if constexpr (INT) {
                        int m = acc[mi][ni][0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) m = max(m, acc[mi][ni][r]);
                        fany[mi][ni] = __ballot(m > (int)thr[ni]);
                    } else {
                        float m = acc[mi][ni][0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[mi][ni][r]);
                        fany[mi][ni] = __ballot(m > __uint_as_float(thr[ni]));
                    }
                    any |= fany[mi][ni];
                }
            if (dbg & 32) { const unsigned long long x2 = __builtin_amdgcn_s_memtime(); t_pha += x2 - te0; te0 = x2; }
            if (dbg & 16) { if (lane == 0) { atomicAdd(&dbgbuf[0], 1ull); if (any) atomicAdd(&dbgbuf[1], 1ull); } }
            if (any && !(dbg & 4)) {
                // Phase B: only fragments in which some lane passed are walked element by element
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        if (fany[mi][ni] == 0) continue;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            bool pass;
                            if constexpr (INT) pass = acc[mi][ni][r] > (int)thr[ni];
                            else pass = acc[mi][ni][r] > __uint_as_float(thr[ni]);
                            if (pass) {
                                uint32_t keyhi;
                                if constexpr (INT) keyhi = (uint32_t)(-acc[mi][ni][r]) ^ 0x80000000u;
                                else keyhi = f32_key(-acc[mi][ni][r]);
                                const uint32_t id = col0 + rbase + mi * 32 + (r & 3) + 8 * (r >> 2);
                                if (cntr[ni] < (uint32_t)SUBCAP) sbuf[ni][cntr[ni]] = ((uint64_t)keyhi << 32) | id;
                                cntr[ni] += 1;
                                if (dbg & 16) atomicAdd(&dbgbuf[3], 1ull);
                            }
                        }
                    }
                // publish the fills; request a prune when a sub-buffer passed its high-water mark
uint32_t ncross = 0;
                bool urgent = false;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    asm volatile("ds_write_b32 %0, %1" :: "v"(sc_addr[ni]), "v"(cntr[ni]) : "memory");
                    ncross += (uint32_t)(cnt0[ni] <= (uint32_t)SUBHW && cntr[ni] > (uint32_t)SUBHW);
                    urgent |= cntr[ni] > (uint32_t)SUBURG;
                }
                if (ncross) {
                    const uint32_t a1 = pflag_lds_addr + 4;
                    asm volatile("ds_add_u32 %0, %1" :: "v"(a1), "v"(ncross) : "memory");
                }
                if (urgent) {
                    const uint32_t one = 1;
                    asm volatile("ds_write_b32 %0, %1" :: "v"(pflag_lds_addr), "v"(one) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (dbg & 32) t_phb += __builtin_amdgcn_s_memtime() - te0;
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (dbg & 32) {
        if (tid == 0) {
            atomicAdd(&dbgbuf[4], __builtin_amdgcn_s_memtime() - tk0);
            atomicAdd(&dbgbuf[5], t_round);
            atomicAdd(&dbgbuf[6], t_pha);
            atomicAdd(&dbgbuf[7], t_phb);
            atomicAdd(&dbgbuf[8], n_round);
            atomicAdd(&dbgbuf[9], t_sync);
            atomicAdd(&dbgbuf[10], t_front);
            atomicAdd(&dbgbuf[11], t_body);
        }
    }
    for (int q = wave; q < T2; q += NW) {
        const uint32_t qg = qt * T2 + q;
        if (qg >= nq) continue;
        const uint64_t best = prune_query<INT, false, KPV>(kept, subs, tau, nkept, subcnt + q * NSUB, q, lane, kp, nullptr, smem + LDS4_SEL + wave * SEL_SCRATCH);
        partial[((size_t)qg * Stot + list_base + s) * KPV + lane] = best;
#pragma unroll
        for (int c = 1; c < KPV / 64; ++c)  // (the rest of the list as prune_query left it: compacted, ~0 padded)
            partial[((size_t)qg * Stot + list_base + s) * KPV + lane + 64 * c] = kept[(size_t)q * KPV + lane + 64 * c];
        if (lane == 0) pcnt[(size_t)qg * Stot + list_base + s] = nkept[q];  // valid entries (compacted to the front)
    }
}

// The lists are UNSORTED (valid entries compacted to the front, ~0 padded).  Select-based merge: the
// valid keys of all lists of a query are gathered into LDS (one lane per list, 64 lists at a time),
// then one radix select yields the kp best.  Output: candidate ids (unordered) and/or the kp-th key.
// vmode 0: plain.  vmode 1 (verification of guessed thresholds): a query whose lists hold fewer than kp keys cannot
// prove that it has the kp best -> defq[q] = 1, its repair threshold tau_out[q] = "no bound" and *flag = 1 (the
// gated repair scan re-reads the corpus for those queries only); other queries get tau_out[q] = 0 ("nothing
// passes").  vmode 2 (repair merge, lists [l0, l0 + S)): a no-op unless *flag, and then only for defq queries.
// vmode 3 (verification under the certificate): a deficient query is handed to the certificate's own repair -- its k'-th key is
// written as 0, which no safe key undercuts, so the re-rank lists it and the exact scan recomputes it (exact_scan.hip) -- as long
// as the batch has at most defer_max of them; beyond that *flag is raised and the gated pair behind this launch (repair scan +
// vmode-2 merge, both no-ops otherwise) repairs them all with one more read of the corpus.
__global__ __launch_bounds__(64) void k_bf_merge_sel(const uint64_t *__restrict__ partial,
                                                     const uint32_t *__restrict__ pcnt, uint32_t S, uint32_t Stot,
                                                     uint32_t *__restrict__ cand_ids, uint32_t *__restrict__ tau_out,
                                                     uint32_t kp, uint32_t cap, uint32_t l0 = 0, int vmode = 0,
                                                     uint32_t *__restrict__ flag = nullptr, uint32_t *__restrict__ defq = nullptr,
                                                     uint32_t *__restrict__ host_flag = nullptr /* mapped host memory */,
                                                     uint32_t lkp = KP /* width (stride) of the lists and of cand_ids: 64, 128 or 256 */,
                                                     uint32_t *__restrict__ cand_keys = nullptr /* certificate (bf_cert.h): the candidates leave SORTED by (key, id), their keys here */,
                                                     uint32_t *__restrict__ kth_out = nullptr /* ... and the k'-th key per query (0xFFFFFFFF: fewer than k' rows exist) */,
                                                     uint32_t *__restrict__ cert_nflag = nullptr /* the repair list's counter: reset here, ahead of the re-rank that fills it */,
                                                     uint32_t lw = kWave /* lists taken per round (one lane each): cap >= kp + lw * lkp -- 16 for the wide lists keeps the pool at
                                                                            17 - 35 KB instead of 67 - 133 KB, i.e. 4+ workgroups per CU instead of 1 - 2 */,
                                                     uint32_t *__restrict__ miss_ctr = nullptr /* vmode 3, mapped host memory: guess misses (cumulative; cert_info.guess_misses) */,
                                                     uint32_t defer_max = 0xFFFFFFFFu /* vmode 3: more deficient queries than this in the batch -> *flag = 1 (gated pair) */) {
    if (cert_nflag && vmode != 2 && blockIdx.x == 0 && threadIdx.x == 0) { cert_nflag[0] = 0; cert_nflag[1] = 0; }  // ([1]: the exact scan's finished-block counter)
    if (vmode == 2 && (*flag == 0 || defq[blockIdx.x] == 0)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
    uint64_t *stage = reinterpret_cast<uint64_t *>(smem + SEL_HIST_BYTES);
    uint64_t *pool = reinterpret_cast<uint64_t *>(smem + SEL_HIST_BYTES + max(lkp * 8u, (uint32_t)SEL_STAGE_BYTES));  // [cap]
    const int lane = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint64_t *p = partial + ((size_t)q * Stot + l0) * lkp;
    uint32_t n = 0;       // keys in the pool (wave-uniform)
    uint64_t T = ~0ull - 1;
    auto reduce_pool = [&]() {  // pool -> its kp best, compacted to the front
        if (n <= kp) return;
        const uint32_t E = (n + kWave - 1) / kWave;
        T = wave_kth_key<0>([&](uint32_t e) { const uint32_t i = e * kWave + lane; return i < n ? pool[i] : ~0ull; }, E, kp, hist, lane);
        uint32_t base = 0;
        for (uint32_t e = 0; e < E; ++e) {
            const uint32_t i = e * kWave + lane;
            const uint64_t k = i < n ? pool[i] : ~0ull;
            const bool selv = k <= T;
            const unsigned long long m = __ballot(selv);
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
            if (selv && pos < lkp) stage[pos] = k;
            base += (uint32_t)__popcll(m);
        }
        __syncthreads();
        for (uint32_t i = lane; i < kp; i += kWave) pool[i] = stage[i];
        n = kp;
        __syncthreads();
    };
    for (uint32_t s0 = 0; s0 < S; s0 += lw) {
        // one lane per list; its valid-entry count was written by the scan
        const uint32_t li = s0 + lane;
        const uint64_t *lp = p + (size_t)li * lkp;
        const uint32_t cnt = ((uint32_t)lane < lw && li < S) ? min(pcnt[(size_t)q * Stot + l0 + li], lkp) : 0u;
        const uint32_t inc = wave_incl_scan(cnt, lane);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if (n + tot > cap) reduce_pool();  // make room (cap >= kp + 64 * lkp)
        const uint32_t off = n + inc - cnt;
        for (uint32_t j2 = 0; j2 < cnt; ++j2) pool[off + j2] = lp[j2];
        n += tot;
        __syncthreads();
    }
    reduce_pool();
    if (cand_ids && cand_keys) {  // sorted by (key, id): rank by counting (n <= 256 keys in LDS)
        for (uint32_t i = lane; i < lkp; i += kWave) {
            if (i < n) {
                const uint64_t k = pool[i];
                uint32_t rank = 0;
                const uint32_t n2 = n & ~1u;  // (two keys per 16-byte LDS read, four reads in flight: the one-key loop is a chain of LDS round trips)
#pragma unroll 4
                for (uint32_t j = 0; j < n2; j += 2) {
                    const ulonglong2 o = *reinterpret_cast<const ulonglong2 *>(pool + j);
                    rank += (o.x < k || (o.x == k && j < i)) ? 1u : 0u;
                    rank += (o.y < k || (o.y == k && j + 1 < i)) ? 1u : 0u;
                }
                if (n2 < n) rank += (pool[n2] < k || (pool[n2] == k && n2 < i)) ? 1u : 0u;
                cand_ids[(size_t)q * lkp + rank] = k == ~0ull ? 0xFFFFFFFFu : (uint32_t)k;
                cand_keys[(size_t)q * lkp + rank] = (uint32_t)(k >> 32);
            } else {
                cand_ids[(size_t)q * lkp + i] = 0xFFFFFFFFu;
                cand_keys[(size_t)q * lkp + i] = 0xFFFFFFFFu;
            }
        }
        if (kth_out) {
            uint64_t mxl = 0;
            for (uint32_t i = lane; i < n; i += kWave) mxl = umax64(mxl, pool[i]);
            const uint64_t mx = wave_max64(mxl);
            if (lane == 0) kth_out[q] = n >= kp ? (uint32_t)(mx >> 32) : vmode == 3 ? 0u : 0xFFFFFFFFu;
        }
    } else if (cand_ids) {
        for (uint32_t i = lane; i < lkp; i += kWave) {
            const uint64_t k = i < n ? pool[i] : ~0ull;
            cand_ids[(size_t)q * lkp + i] = k == ~0ull ? 0xFFFFFFFFu : (uint32_t)k;
        }
    }
    if (vmode == 3) {
        // a few stragglers go to the exact scan (their k'-th key 0 makes the re-rank list them); a batch with MORE than defer_max
        // of them (a guess too tight for the whole batch: distribution shift) raises *flag instead, and the gated repair scan +
        // repair merge behind this launch re-read the corpus once for all of them -- an exact scan per query would cost
        // ~0.8 ms x the batch at 10M x 768.  flag[1] counts the deficient queries of this batch (zeroed by the guess kernel).
        if (lane == 0) {
            const bool deficient = n < kp;
            if (defq) defq[q] = deficient ? 1u : 0u;
            if (tau_out) tau_out[q] = deficient ? 0xFFFFFFFFu : 0u;
            if (deficient) {
                if (host_flag) __hip_atomic_store(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (the next guesses are looser)
                if (miss_ctr) __hip_atomic_fetch_add(miss_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (flag && atomicAdd(flag + 1, 1u) == defer_max) atomicOr(flag, 1u);
            }
        }
    } else if (vmode == 1) {
        if (lane == 0) {
            const bool deficient = n < kp;
            defq[q] = deficient ? 1u : 0u;
            tau_out[q] = deficient ? 0xFFFFFFFFu : 0u;
            if (deficient) {
                atomicOr(flag, 1u);
                if (host_flag) __hip_atomic_store(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    } else if (tau_out) {
        uint64_t mxl = 0;
        for (uint32_t i = lane; i < n; i += kWave) mxl = umax64(mxl, pool[i]);
        uint64_t mx = wave_max64(mxl);
        if (lane == 0) tau_out[q] = n >= kp ? (uint32_t)(mx >> 32) : 0xFFFFFFFFu;
    }
}

// standard-normal quantile (Acklam's rational approximation, |rel err| < 1.2e-9): the z of a tail probability
double norm_quantile(double p) {
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    if (p < 0.02425) {
        const double q = std::sqrt(-2.0 * std::log(p));
        return (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0);
    }
    const double q = p - 0.5, r = q * q;
    return (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
           (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0);
}

// queries -> operand layout of the scan: [nq_pad][row_bytes], zero padded.
// mode 0: fp32 copy; 1: bf16 RNE; 2: int8 with the index's quantisation (q_lo, q_inv, affine);
// 3: fp4 codes, two per byte (q_lo = the index's p_mid, q_inv = p_scale; row_elems counts BYTES here).
__device__ __forceinline__ void bf_prep_queries_body(const float *__restrict__ q, uint32_t nq, uint32_t dim, unsigned char *__restrict__ out,
                                                     uint32_t nq_pad, uint32_t row_elems, int mode, float q_lo, float q_inv, int affine,
                                                     uint64_t first, uint64_t stride) {
    const uint64_t total = (uint64_t)nq_pad * row_elems;
    for (uint64_t t = first; t < total; t += stride) {
        const uint32_t r = (uint32_t)(t / row_elems), c = (uint32_t)(t - (uint64_t)r * row_elems);
        if (mode == 3) {
            uint32_t b = 0;
            if (r < nq) {
                if (2 * c < dim) b = fp4_e2m1_code((q[(size_t)r * dim + 2 * c] - q_lo) * q_inv);
                if (2 * c + 1 < dim) b |= fp4_e2m1_code((q[(size_t)r * dim + 2 * c + 1] - q_lo) * q_inv) << 4;
            }
            out[t] = (unsigned char)b;
            continue;
        }
        const bool live = r < nq && c < dim;
        const float v = live ? q[(size_t)r * dim + c] : 0.0f;
        if (mode == 1) {
            uint32_t u = __float_as_uint(v);
            u += 0x7FFFu + ((u >> 16) & 1u);
            reinterpret_cast<uint16_t *>(out)[t] = (uint16_t)(u >> 16);
        } else if (mode == 2) {
            int qv = 0;
            if (live) {
                const float f = affine ? __builtin_rintf((v - q_lo) * q_inv) - 128.0f : __builtin_rintf(v * q_inv);
                qv = (int)fminf(fmaxf(f, -128.0f), 127.0f);
            }
            reinterpret_cast<int8_t *>(out)[t] = (int8_t)qv;
        } else {
            reinterpret_cast<float *>(out)[t] = v;
        }
    }
}

__global__ __launch_bounds__(256) void k_bf_prep_queries(const float *__restrict__ q, uint32_t nq,
                                                         uint32_t dim, unsigned char *__restrict__ out,
                                                         uint32_t nq_pad, uint32_t row_elems, int mode,
                                                         float q_lo, float q_inv, int affine) {
    bf_prep_queries_body(q, nq, dim, out, nq_pad, row_elems, mode, q_lo, q_inv, affine, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x,
                         (uint64_t)gridDim.x * blockDim.x);
}

// dot / cosine with a quantised screen (int8, fp4): the ranking does not depend on the query's length, the quantiser does --
// its step was placed by the ROWS' spread (fp4: RMS, int8: largest magnitude), and a query of another scale (cosine rows
// are normalised at insert, queries arrive as they are) would clip or vanish.  One wave per query: out = q * target / stat(q).
__global__ __launch_bounds__(64) void k_rescale_queries(const float *__restrict__ q, uint32_t nq, uint32_t dim, int use_absmax,
                                                        float target, float *__restrict__ out, float add /* the quantiser's offset */) {
    const uint32_t r = blockIdx.x;
    const int lane = threadIdx.x;
    if (r >= nq) return;
    float acc = 0.0f;
    for (uint32_t c = lane; c < dim; c += kWave) {
        const float v = q[(size_t)r * dim + c];
        acc = use_absmax ? fmaxf(acc, fabsf(v)) : acc + v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(acc, off);
        acc = use_absmax ? fmaxf(acc, o) : acc + o;
    }
    const float stat = use_absmax ? acc : sqrtf(acc / (float)dim);
    const float f = (stat > 0.0f && stat < __builtin_inff()) ? target / stat : 1.0f;
    for (uint32_t c = lane; c < dim; c += kWave) out[(size_t)r * dim + c] = q[(size_t)r * dim + c] * f + add;
}

struct BfPlan {
    uint32_t tile, cap, nq_pad, QT, S, NT, KT, row_bytes, row_elems;
    uint64_t off_q, off_cand, off_partial, off_ids, off_taug, off_verify, off_pcnt, off_qs, off_ckeys, off_kth, off_cert, cand_bytes, total;
    uint32_t QT8, S8;  // int8 stream kernel: tiles of 128 queries, streams
};

uint32_t debug_flags() {
    const char *e = tuning("ZVDB_BF_DEBUG");
    return e ? (uint32_t)atoi(e) : 0u;
}

// Up to 4 dimensions a reduced-precision screen cannot tell rows apart (an 8-bit code has 256 values per dimension; 200k
// points on a line share them by the hundred): such indexes are screened in fp32 whatever screen was asked for -- at 4
// dimensions the MFMA work is nothing.
// An index that STORES bf16 rows (ZVDB_FLAG_BF16_ROWS) has no fp32 operand for the fp32-MFMA screen: its bf16 screen is lossless
// (the operands ARE the rows), so ZVDB_BF_F32 -- and the tiny-dimension rule above -- are served by it.
static uint32_t effective_mode(const zvdb_hip_index *h, uint32_t mode) {
    if (h->flags & ZVDB_FLAG_BF16_ROWS) return (mode == ZVDB_BF_F32 || h->dim <= 4) ? (uint32_t)ZVDB_BF_BF16 : mode;
    return h->dim <= 4 ? (uint32_t)ZVDB_BF_F32 : mode;
}

BfPlan make_plan(const zvdb_hip_index *h, uint32_t nq, uint32_t mode, uint32_t k) {
    mode = effective_mode(h, mode);
    BfPlan p{};
    p.tile = T2;
    p.cap = CAP2;
    p.nq_pad = round_up(std::max(nq, 1u), p.tile);
    p.QT = p.nq_pad / p.tile;
    p.NT = (uint32_t)((h->count + p.tile - 1) / p.tile);
    if (mode == ZVDB_BF_BF16) { p.row_elems = h->ldh; p.row_bytes = h->ldh * 2; }
    else if (mode == ZVDB_BF_I8) { p.row_elems = round_up(h->dim, 128); p.row_bytes = p.row_elems; }
    else if (mode == ZVDB_BF_FP4) { p.row_bytes = round_up(h->dim, 256) / 2; p.row_elems = p.row_bytes; /* prep kernel: per byte */ }
    else { p.row_elems = h->ld; p.row_bytes = h->ld * 4; }
    p.KT = p.row_bytes / KB;
    // resident workgroups: 1 per CU; streams in multiples of 8 (XCD groups)
    const uint32_t resident = 256;
    uint32_t S = std::max(8u, (resident / p.QT) / 8 * 8);
    while (S > 8 && S > p.NT) S -= 8;  // do not create streams with no tiles
    p.S = S;
    p.QT8 = p.nq_pad / 128;
    p.S8 = std::max(8u, (resident / p.QT8) / 8 * 8);
    while (p.S8 > 8 && p.S8 > p.NT) p.S8 -= 8;
    p.S8 = std::min(p.S8, p.S);  // (lists and counters are sized by S)
    uint64_t off = 0;
    p.off_q = off;
    off += round_up64(std::max<uint64_t>((uint64_t)p.nq_pad * p.row_bytes,
                                         mode == ZVDB_BF_FP4 && (fp4s_eligible(h) || fp4s128_eligible(h)) ? fp4s_query_bytes(h, p.nq_pad)
                                         : mode == ZVDB_BF_I8 && i8s_eligible(h) ? i8s_query_bytes(h, p.nq_pad)
                                         : mode == ZVDB_BF_F32 && (f32s_eligible(h) || f32rq_eligible(h)) ? f32s_query_bytes(h, p.nq_pad)
                                         : mode == ZVDB_BF_BF16 && bf16s_eligible(h) ? bf16s_query_bytes(h, p.nq_pad) : 0), 256);
    p.off_cand = off;
    // (after the scan the region is free: the exact scan behind the certificate keeps its partial lists here)
    p.cand_bytes = std::max<uint64_t>(std::max<uint64_t>((uint64_t)p.QT * p.S * p.tile * p.cap,
                                                         (uint64_t)std::max(p.QT * p.S, p.QT8 * p.S8) * BLKV_ENTRIES) * 8,
                                      round_up64(exact_scan_scratch_bytes(std::max(nq, 1u), std::max(k, 1u)), 256));
    off += p.cand_bytes;
    p.off_partial = off;
    off += (uint64_t)p.nq_pad * 3 * p.S * KPMAX * 8;  // lists of up to three passes (widest list)
    p.off_ids = off;
    off += (uint64_t)p.nq_pad * KPMAX * 4;
    p.off_taug = off;
    off += round_up64((uint64_t)p.nq_pad * 4, 256) + 256;  // + debug counters
    p.off_verify = off;  // guessed-threshold pipeline: [flag 256 B][defq nq_pad][tau_rep nq_pad]
    off += 256 + 2 * round_up64((uint64_t)p.nq_pad * 4, 256);
    p.off_pcnt = off;
    off += round_up64((uint64_t)p.nq_pad * 3 * p.S * 4, 256);  // valid-entry count of every list
    p.off_qs = off;  // rescaled fp32 queries (dot / cosine with a quantised screen)
    if (h->metric != ZVDB_METRIC_L2SQ && (mode == ZVDB_BF_I8 || mode == ZVDB_BF_FP4)) off += round_up64((uint64_t)p.nq_pad * h->dim * 4, 256);
    // certificate (bf_cert.h): the candidates' screen keys, the k'-th key per query, the repair list [count 256 B][nq_pad ids]
    p.off_ckeys = off;
    off += (uint64_t)p.nq_pad * KPMAX * 4;
    p.off_kth = off;
    off += round_up64((uint64_t)p.nq_pad * 4, 256);
    p.off_cert = off;
    off += 256 + round_up64((uint64_t)p.nq_pad * 4, 256);
    p.total = off;
    return p;
}

// k-slice rotation per query tile (nibble qt&7), each < KT.  ZVDB_BF_ROT=<hex> overrides (A/B).
uint32_t rot_table(uint32_t KT) {
    const uint32_t env = [] { const char *e = tuning("ZVDB_BF_ROT"); return e ? (uint32_t)strtoul(e, nullptr, 16) : 0x53145420u; }();
    uint32_t t = 0;
    for (int i = 0; i < 8; ++i) t |= (((env >> (4 * i)) & 15u) % KT) << (4 * i);
    return t;
}

// Guessed thresholds for the STAGED kernel's float screens (fp32, bf16 on rows wider than the stream kernel takes): mean and
// variance of the screen key |v|^2 / 2 - q.v (dot / cosine: -q.v) per query over a sample of the shard's rows, evenly strided
// -> tau = mean + z sigma (k_bf16_guess's rule).  Plain fp32 arithmetic on the stored rows: a guess needs no more (the bf16
// screen's keys differ from these by its rounding, far below sigma), and what a guess gets wrong is verified and repaired.
// One workgroup = 4 queries (held in LDS) x nsamp rows, 8 lanes per row.
constexpr int SG_Q = 4, SG_SPLIT = 16;
// grid (ceil(nq / SG_Q), SG_SPLIT): workgroup (g, p) takes steps p, p + SG_SPLIT, ... of the sample (32 rows per step) -- a walk is
// one dependent memory round trip per step, so the sample is cut into SG_SPLIT short chains; k_sample_tau adds the partial sums.
template <typename RT>
__global__ __launch_bounds__(256) void k_sample_guess(const RT *__restrict__ rows, uint32_t ld, uint32_t dim, const float *__restrict__ norm /* null: dot */,
                                                      uint32_t N, const float *__restrict__ queries, uint32_t nq, uint32_t nsamp, float z,
                                                      uint32_t *__restrict__ verify_flag, double *__restrict__ part /* [SG_SPLIT][nq pad SG_Q][2] */,
                                                      unsigned char *__restrict__ prep_out /* null: the scan's query operand is written elsewhere */,
                                                      uint32_t nq_pad, uint32_t row_elems, int prep_mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *q = reinterpret_cast<float *>(smem);                      // [SG_Q][ld], zero padded like the rows (ld: a multiple of 32)
    double *red = reinterpret_cast<double *>(q + (size_t)SG_Q * ld);  // [4 waves][SG_Q][2]
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && verify_flag) { verify_flag[0] = 0; verify_flag[1] = 0; }
    // the scan's query operand (fp32 copy / bf16, zero padded), this workgroup's share: one launch fewer per batch (k_bf_prep_queries)
    if (prep_out)
        bf_prep_queries_body(queries, nq, dim, prep_out, nq_pad, row_elems, prep_mode, 0.0f, 1.0f, 0,
                             ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x, (uint64_t)gridDim.x * gridDim.y * 256);
    const uint32_t q0 = blockIdx.x * SG_Q;
    for (uint32_t i = threadIdx.x; i < SG_Q * ld; i += 256) {
        const uint32_t qi = q0 + i / ld, c = i % ld;
        q[i] = (qi < nq && c < dim) ? queries[(size_t)qi * dim + c] : 0.0f;
    }
    __syncthreads();
    // the sample: runs of 64 consecutive rows, evenly spread (single rows strided over the shard cost a page walk each: 4,096
    // rows of a 1M x 128 shard took 100 us, the runs take 25 us); sample row i = (i / 64) * bstep + i % 64
    const uint32_t ns = min(nsamp, N) / 64 * 64 ? min(nsamp, N) / 64 * 64 : min(nsamp, N), bstep = ns >= 64 ? N / (ns / 64) : 1u;
    float f1[SG_Q], f2[SG_Q];  // per thread: <= nsamp / (32 SG_SPLIT) keys (the sums go to double below)
#pragma unroll
    for (int j = 0; j < SG_Q; ++j) f1[j] = f2[j] = 0.0f;
    // 8 lanes per row: lane g of a group walks elements 4 g + 32 i + [0, 4) (a group reads 128 contiguous bytes of fp32 rows per
    // step), 32 rows per workgroup step; the group's partial dot products meet in its lane 0 (xor 4, 2, 1)
    const uint32_t g8 = threadIdx.x & 7, grp = threadIdx.x >> 3;
    for (uint32_t i0 = 32 * blockIdx.y; i0 < ns; i0 += 32 * SG_SPLIT) {
        const uint32_t i = i0 + grp;
        const bool live = i < ns;
        const uint32_t r = live ? (ns >= 64 ? (i >> 6) * bstep + (i & 63) : i) : 0u;
        const RT *v = rows + (size_t)r * ld;
        float d[SG_Q];
#pragma unroll
        for (int j = 0; j < SG_Q; ++j) d[j] = 0.0f;
        for (uint32_t c0 = 4 * g8; c0 < ld; c0 += 32) {
            const float4 e = row_load4(v + c0);
#pragma unroll
            for (int j = 0; j < SG_Q; ++j) {
                const float4 qq = *reinterpret_cast<const float4 *>(q + j * ld + c0);
                d[j] += e.x * qq.x + e.y * qq.y + e.z * qq.z + e.w * qq.w;
            }
        }
#pragma unroll
        for (int j = 0; j < SG_Q; ++j) {
            d[j] += __shfl_xor(d[j], 4);
            d[j] += __shfl_xor(d[j], 2);
            d[j] += __shfl_xor(d[j], 1);
        }
        if (live && g8 == 0) {
            const float hn = norm ? 0.5f * norm[r] : 0.0f;
#pragma unroll
            for (int j = 0; j < SG_Q; ++j) {
                const float key = hn - d[j];
                f1[j] += key;
                f2[j] += key * key;
            }
        }
    }
    double s1[SG_Q], s2[SG_Q];
#pragma unroll
    for (int j = 0; j < SG_Q; ++j) { s1[j] = (double)f1[j]; s2[j] = (double)f2[j]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < SG_Q; ++j) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s1[j] += __shfl_xor(s1[j], off);
            s2[j] += __shfl_xor(s2[j], off);
        }
        if (lane == 0) { red[(wave * SG_Q + j) * 2] = s1[j]; red[(wave * SG_Q + j) * 2 + 1] = s2[j]; }
    }
    __syncthreads();
    if (threadIdx.x < SG_Q * 2) {  // (plain stores: an agent-scope fence per workgroup -- the "last one reduces" pattern -- cost 65 us here)
        double a = 0.0;
        for (int w = 0; w < 4; ++w) a += red[w * SG_Q * 2 + threadIdx.x];
        part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SG_Q * 2 + threadIdx.x] = a;
    }
}

// partial sums -> tau = mean + z sigma per query (k_bf16_guess's rule)
__global__ __launch_bounds__(256) void k_sample_tau(const double *__restrict__ part, uint32_t groups, uint32_t nq, uint32_t ns, float z,
                                                    uint32_t *__restrict__ tau_g) {
    const uint32_t qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    double a = 0.0, b = 0.0;
    for (uint32_t p = 0; p < (uint32_t)SG_SPLIT; ++p) {
        const double *pp = part + ((size_t)p * groups * SG_Q + qi) * 2;
        a += pp[0];
        b += pp[1];
    }
    uint32_t out = 0xFFFFFFFFu;  // no bound
    if (ns >= 64) {
        const double mu = a / ns, var = b / ns - mu * mu;
        if (var > 0.0) {
            const float tk = (float)(mu + (double)z * sqrt(var));
            if (tk == tk && fabsf(tk) < 3.0e38f) out = f32_key(tk);
        }
    }
    tau_g[qi] = out;
}

template <int MODE, int NI, int KBV, int KPV = KP>
int launch_scan5(const BfPlan &p, const unsigned char *q_op, const unsigned char *rows, const void *vn,
                 uint32_t N, uint32_t nq, int alpha_i, uint64_t *cand, uint64_t *partial, uint32_t *tau_g,
                 uint32_t kp, uint32_t t_first, uint32_t nt, uint32_t list_base, uint32_t Stot, uint32_t *pcnt,
                 hipStream_t s, const uint32_t *gate = nullptr, unsigned long long *dbg_area = nullptr) {
    unsigned long long *dbgbuf = dbg_area ? dbg_area : reinterpret_cast<unsigned long long *>(tau_g + p.nq_pad);
    if constexpr (KPV != KP) {  // wide lists: production build only
        ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bf_scan5<MODE, false, NI, KBV, KPV>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
        hipLaunchKernelGGL((k_bf_scan5<MODE, false, NI, KBV, KPV>), dim3(p.QT * p.S), dim3(NI == 4 ? 256 : 512), LDS4, s, q_op, rows, p.row_bytes,
                           p.row_bytes / KBV, vn, N, nq, alpha_i, p.QT, p.S, nt, cand, partial, tau_g, kp, t_first,
                           list_base, Stot, pcnt, rot_table(p.row_bytes / KBV), 0u, dbgbuf, gate);
        return ZVDB_OK;
    }
    if (debug_flags()) {
        ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bf_scan5<MODE, true, NI, KBV>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
        hipLaunchKernelGGL((k_bf_scan5<MODE, true, NI, KBV>), dim3(p.QT * p.S), dim3(NI == 4 ? 256 : 512), LDS4, s, q_op, rows, p.row_bytes,
                           p.row_bytes / KBV, vn, N, nq, alpha_i, p.QT, p.S, nt, cand, partial,
                           (debug_flags() & 2) ? nullptr : tau_g, kp, t_first, list_base, Stot, pcnt, rot_table(p.row_bytes / KBV),
                           debug_flags(), dbgbuf, gate);
    } else {
        ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bf_scan5<MODE, false, NI, KBV>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
        hipLaunchKernelGGL((k_bf_scan5<MODE, false, NI, KBV>), dim3(p.QT * p.S), dim3(NI == 4 ? 256 : 512), LDS4, s, q_op, rows, p.row_bytes,
                           p.row_bytes / KBV, vn, N, nq, alpha_i, p.QT, p.S, nt, cand, partial, tau_g, kp, t_first,
                           list_base, Stot, pcnt, rot_table(p.row_bytes / KBV), 0u, dbgbuf, gate);
    }
    return ZVDB_OK;
}

}  // namespace

uint64_t bruteforce_workspace_bytes(const zvdb_hip_index *h, uint32_t nq, uint32_t k, uint32_t mode) {
    if (h->dim == 0) return 256;
    return make_plan(h, nq, mode, std::min(k, 256u)).total;
}

int launch_bruteforce(zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t k,
                      uint32_t mode, uint64_t id_offset, void *d_ws, uint64_t ws_bytes,
                      uint64_t *d_ids, float *d_dist, uint32_t *d_count, hipStream_t s) {
    if (nq == 0) return ZVDB_OK;
    if (mode == ZVDB_BF_BF16 && !(h->flags & (ZVDB_FLAG_BF16_MIRROR | ZVDB_FLAG_BF16_ROWS))) {
        set_error("ZVDB_BF_BF16 needs an index created with ZVDB_FLAG_BF16_MIRROR (or bf16 row storage, ZVDB_FLAG_BF16_ROWS)");
        return ZVDB_ERR_INVALID;
    }
    mode = effective_mode(h, mode);
    // the screen keeps k' = 64 candidates per query; the exact re-rank needs a margin over k to
    // absorb the screen's rounding (reduced-precision inputs: wider margin).
    const uint32_t kmax = mode == ZVDB_BF_F32 ? KP - 8 : KP - 16;  // (checked below, once the route is known)
    if (h->count > 0xFFFFFFFFull - T2) {
        set_error("shard too large");
        return ZVDB_ERR_UNSUPPORTED;
    }
    const BfPlan p = make_plan(h, nq, mode, std::min(k, 256u));
    const int guess_off = [] { const char *e = tuning("ZVDB_BF_GUESS"); return e && e[0] == '0'; }();
    // fp4: barrier-free stream kernel over the fragment-order mirror when the query tile fits LDS, else the staged scan
    const bool f7 = mode == ZVDB_BF_FP4 && fp4s_eligible(h);
    // int8: the stream kernel serves large shards (one scan against guessed + verified thresholds); small ones stay on the
    // staged kernel and its threshold cascade
    // needs the 128-wide lists.  The approximate fp4 screen switches earlier (k > 32):
    // k of its k' survivors are asked for, and recall@k falls with the slack (k = 48: 0.83 with k' = 64, 0.97 with 128)
    const uint32_t want = mode == ZVDB_BF_FP4 ? h->screen_survivors : 0u;  // zvdb_hip_index_set_screen_survivors
    // A handful of dimensions and many rows put more rows inside the screen's rounding band of the k-th neighbour than 64
    // survivors hold (the screens evaluate |v|^2 - 2 q.v; 70k points on a line are ~1e-5 apart): keep 128 there.
    const bool dense_low_dim = h->dim <= 8 && h->count > 4096;
    // Certificate (bf_cert.h): the exact screens' results are PROVEN to be the top-k, query by query, or recomputed by the exact
    // scan.  How many survivors the proof needs depends on the data (the k'-th screen key must clear the k-th exact distance by
    // the screen's error bound): the index starts with as few as the route allows and keeps more (64, 128, 256) once a batch had
    // to be repaired -- the flag is raised by the re-rank in host-mapped memory and read here without synchronisation.
    const bool cert = h->cert_on && mode != ZVDB_BF_FP4;
    if (cert) {
        if (!h->cert_host) {
            ZV_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->cert_host), 64, hipHostMallocMapped | hipHostMallocCoherent));
            if (!h->cert_host) { set_error("hipHostMalloc returned no block"); return ZVDB_ERR_OOM; }
            h->cert_host[0] = 0;
            h->cert_host[1] = 0;
            for (int i = 2; i < 16; ++i) h->cert_host[i] = 0;
            ZV_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->cert_hdev), h->cert_host, 0));
        }
        if (h->cert_force) {
            __atomic_store_n(h->cert_host, 0u, __ATOMIC_RELAXED);  // (forced repairs say nothing about the survivors needed)
        } else if (__atomic_load_n(h->cert_host, __ATOMIC_RELAXED) != 0) {
            __atomic_store_n(h->cert_host, 0u, __ATOMIC_RELAXED);
            // one level above what the failing batch ran with (large shards start at 1: see cert_lvl below)
            h->cert_level = std::min(3u, std::max(std::max(h->cert_level, h->cert_level_min), h->count >= (1ull << 20) ? 1u : 0u) + 1u);
            h->cert_clean = 0;
        } else if (++h->cert_clean >= 8192 && h->cert_level > 0) {  // a long clean run: try fewer survivors again
            --h->cert_level;
            h->cert_clean = 0;
        }
        h->cert_queries += nq;
        h->cert_batches += 1;
    }
    // (a shard of a million rows or more starts at 64 survivors: with 32 the bound fails for a large share of the queries of the first
    // batch on high-dimensional rows -- 40 % at 10M x 768 -- and every one of them is an exact scan)
    const uint32_t cert_lvl = cert ? std::max(std::max(h->cert_level, h->cert_level_min), h->count >= (1ull << 20) ? 1u : 0u) : 0u;
    const bool wide_k = k > kmax || (mode == ZVDB_BF_FP4 && k > 32) || want > (uint32_t)KP || dense_low_dim || cert_lvl >= 2;
    const bool big8 = !guess_off && p.NT >= 16 * p.S8 && !(debug_flags() & 64);
    const bool s8 = mode == ZVDB_BF_I8 && i8s_eligible(h) && (big8 || wide_k);
    // fp4, 768 < dim <= 1536: the 128-query stream kernel (same conditions as the int8 one), else the staged scan
    const bool s4 = mode == ZVDB_BF_FP4 && !f7 && fp4s128_eligible(h) && (big8 || wide_k);
    // fp32, dim <= 128: the 128-query stream kernel with fp32 MFMAs on a fragment-order fp32 mirror (bf_i8_stream.hip, MODE 0) -- the
    // staged kernel streams both operands through LDS and reaches 0.68 of the fp32-MFMA peak at BASELINE configs[1]
    // -- measured: +13 % at 64 dimensions, +0 .. 5 % at 128 (both designs sit at ~0.68 there: DESIGN 3, K4), for a mirror as large as
    // the rows; so: by default up to 64 dimensions, and wherever k > 112 needs its wider lists; ZVDB_BF_F32_KERNEL=stream: up to 128
    const bool sf_wide = [] { const char *e = tuning("ZVDB_BF_F32_KERNEL"); return e && e[0] == 's'; }();
    // fp32, dim <= 256, batches of more than 128 queries on large shards: the register-resident-query scan (bf_f32_regq.hip) -- 256
    // queries per workgroup in registers, the corpus HBM -> registers once per batch tile, no LDS staging, no barrier; work split by
    // 32-row groups.  (Smaller batches: half of its waves would carry no query -- the 128-query kernel / the staged kernel serve them.)
    // (smaller shards with k beyond the staged kernel's 112: one flooding scan of the same kernel, like the bf16 / int8 stream routes)
    const bool sr_big = !guess_off && p.NT >= 8 * p.S;
    const bool sr = mode == ZVDB_BF_F32 && f32rq_eligible(h) && nq > 128 && !(debug_flags() & 64) && (sr_big || k > (uint32_t)KPW - 16);
    const bool sf = !sr && mode == ZVDB_BF_F32 && f32s_eligible(h) && ((big8 && (h->dim <= 64 || sf_wide)) || k > (uint32_t)KPW - 16);
    // bf16: the register-resident-query stream kernel (bf_bf16_stream.hip) serves large shards of <= 768 dimensions inside the
    // guessed-threshold pipeline; small shards and wider rows stay on the staged kernel over the row-major bf16 mirror
    // (from 8 tiles = 64 row groups per stream: BASELINE configs[1], 1M x 128 at batch 256 = 256 streams of 15 tiles, rides it)
    // (smaller shards with k beyond the staged kernel's 112: one flooding scan of the stream kernel, like the int8 route)
    const bool sb_big = !guess_off && p.NT >= 8 * p.S;
    const bool sb = mode == ZVDB_BF_BF16 && bf16s_eligible(h) && !(debug_flags() & 64) && (sb_big || k > (uint32_t)KPW - 16);
    if (sb) ZV_CHECK(ensure_bf16_frag_mirror(h, s));
    if (sf || sr) ZV_CHECK(ensure_f32_frag_mirror(h, s));
    // its thresholds: from the rows' second moments, one small launch per batch (ZVDB_BF_F32_GUESS=sample: the 128-query kernel's sampling mode + k_i8_guess)
    const bool sr_moments = sr && sr_big && (h->dim > 128 /* (the sampling mode is the 128-query kernel's: dim <= 128) */ ||
                                   [] { const char *e = tuning("ZVDB_BF_F32_GUESS"); return !e || e[0] != 's'; }());
    if (sr_moments) ZV_CHECK(ensure_f32rq_stats(h, s));
    if (mode == ZVDB_BF_I8) ZV_CHECK(s8 ? ensure_i8_frag_mirror(h, s) : ensure_i8_mirror(h, s));
    if (mode == ZVDB_BF_FP4) ZV_CHECK((f7 || s4) ? ensure_fp4_frag_mirror(h, s) : ensure_fp4_mirror(h, s));
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

    const bool l2 = h->metric == ZVDB_METRIC_L2SQ;
    const float *d_queries_exact = d_queries;  // the exact re-rank always reads the caller's queries
    if (!l2 && (mode == ZVDB_BF_I8 || mode == ZVDB_BF_FP4)) {
        float *qs = reinterpret_cast<float *>(ws + p.off_qs);
        const bool i8m = mode == ZVDB_BF_I8;
        const float target = i8m ? 127.0f / h->q_inv : (kFp4Amax / 1.7320508f) / h->p_scale;  // rows' |max| / RMS
        hipLaunchKernelGGL(k_rescale_queries, dim3(nq), dim3(64), 0, s, d_queries, nq, h->dim, i8m ? 1 : 0, target, qs,
                           i8m ? 0.0f : h->p_mid);  // (fp4 rows are centred on their mean: the quantiser subtracts it again)
        d_queries = qs;
    }
    bool prep_folded = false;  // the staged guessed-threshold route writes the operand inside its sampling kernel
    auto do_prep = [&]() -> int {
    if (prep_folded) return ZVDB_OK;
    if (f7 || s4) {
        ZV_CHECK(launch_fp4s_prep_queries(h, d_queries, nq, p.nq_pad, q_op, s));
    } else if (sb) {
        ZV_CHECK(launch_bf16s_prep_queries(h, d_queries, nq, p.nq_pad, q_op, s));
    } else if (s8) {
        ZV_CHECK(launch_i8s_prep_queries(h, d_queries, nq, p.nq_pad, q_op, s));
    } else if (sf || sr) {
        ZV_CHECK(launch_f32s_prep_queries(h, d_queries, nq, p.nq_pad, q_op, s));
    } else {
        const uint64_t total = (uint64_t)p.nq_pad * p.row_elems;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 2048);
        hipLaunchKernelGGL(k_bf_prep_queries, dim3(grid), dim3(256), 0, s, d_queries, nq, h->dim, q_op,
                           p.nq_pad, p.row_elems, (int)mode, mode == ZVDB_BF_FP4 ? h->p_mid : h->q_lo,
                           mode == ZVDB_BF_FP4 ? h->p_scale : h->q_inv, l2 ? 1 : 0);
    }
    return ZVDB_OK;
    };
    // screen key: L2 -> |v|^2 - 2 q.v ; dot / cosine -> -q.v (no norm term)
    const int alpha_i = l2 ? -2 : -1;
    const uint32_t N = (uint32_t)h->count;
    uint32_t nlists = p.S, stot_lists = p.S;
    // shared thresholds start at "no bound"; the guessed-threshold pipeline overwrites every live entry itself (k_*_guess),
    // so it skips this launch (and clears its verify flag inside the guess kernel): two fewer launches per batch
    const bool guess_pipe = (f7 && !guess_off && p.NT >= 16 * p.S && !(debug_flags() & 64)) || ((s8 || s4 || sf) && big8) || (sb && sb_big) || (sr && sr_big);
    // k' = 64 survivors per query (one per lane of the selects) carry k <= 48 (56 with the exact fp32 screen); the stream
    // kernels also run with k' = 128 (two entries per lane), which carries k <= 112 -- on large shards inside the guessed-
    // threshold pipeline, on small ones as ONE flooding scan (no bound to start with; the prune rounds learn it)
    const bool stream_route = f7 || s8 || s4 || sb || sf || sr;
    // (the staged kernel has the same two list widths; the v1 kernel kept for A/B has only k' = 64)
    // The stream kernels go on to k' = 256 (four entries per lane): k <= 240, and the fp4 screen uses it from k > 64
    // (recall@100: 0.84 with 128 survivors).
    const bool widest = stream_route && (k > (uint32_t)KPW - 16 || (mode == ZVDB_BF_FP4 && k > 64) || want > (uint32_t)KPW || cert_lvl >= 3);
    const uint32_t kpv = !wide_k ? (uint32_t)KP : widest ? (uint32_t)KPMAX : (uint32_t)KPW;
    if (k > (kpv != (uint32_t)KP ? kpv - 16 : kmax)) {
        set_error("brute force supports k <= %u on this route (k' = %u candidates are re-ranked exactly; k <= %d on the stream routes: "
                  "ZVDB_BF_I8 / ZVDB_BF_BF16 with dim <= 768, ZVDB_BF_F32 with dim <= 128 (<= 256 for batches of more than 128 queries), ZVDB_BF_FP4 with dim <= 1536)", kpv != (uint32_t)KP ? kpv - 16 : kmax, kpv, KPMAX - 16);
        return ZVDB_ERR_UNSUPPORTED;
    }
    // The float screens of the STAGED kernel (fp32; bf16 on rows the stream kernel does not take) ride the same guessed +
    // verified thresholds on shards of >= 8 tiles per stream: the cascade below spends a flooded tile per stream, a merge
    // and a second partial scan on learning what a 4,096-row sample predicts (BASELINE configs[1], 1M x 128 fp32 at batch
    // 256: pass A 187 us + merges 90 us of a 1.03 ms step, and pass C runs against the loose bound of 4,096 rows).
    const bool staged_guess = !guess_pipe && !stream_route && !guess_off && (mode == ZVDB_BF_F32 || mode == ZVDB_BF_BF16) &&
                              kpv == (uint32_t)KP && p.NT >= 8 * p.S && !(debug_flags() & ~(16u | 32u)) &&
                              (size_t)SG_Q * std::max(h->ld, h->ldh) * 4 <= 128 * 1024;  // (k_sample_guess keeps 4 padded queries in LDS)
    prep_folded = (staged_guess && d_queries == d_queries_exact && [] { const char *e = tuning("ZVDB_BF_FOLD_PREP"); return !e || e[0] != '0'; }()) ||
                  sr_moments;  // (k_rq_guess packs the queries it reads)
    ZV_CHECK(do_prep());
    if (!guess_pipe && !staged_guess) ZV_HIP(hipMemsetAsync(ws + p.off_taug, 0xFF, round_up64((uint64_t)p.nq_pad * 4, 256), s));
    if (debug_flags() & (16 | 32)) ZV_HIP(hipMemsetAsync(ws + p.off_taug + (size_t)p.nq_pad * 4, 0, 128, s));
    uint32_t kp_used = (uint32_t)KP;
    // certificate plumbing: the last merge of every route leaves the candidates sorted by screen key, with their keys and the
    // k'-th key; the re-rank proves or lists the query; the exact scan recomputes the listed ones (a no-op launch otherwise)
    uint32_t *ckeys = cert ? reinterpret_cast<uint32_t *>(ws + p.off_ckeys) : nullptr;
    uint32_t *kth = cert ? reinterpret_cast<uint32_t *>(ws + p.off_kth) : nullptr;
    uint32_t *nflag = cert ? reinterpret_cast<uint32_t *>(ws + p.off_cert) : nullptr;
    uint32_t *flist = cert ? reinterpret_cast<uint32_t *>(ws + p.off_cert + 256) : nullptr;
    if (cert) ZV_CHECK(ensure_cert_stats(h, mode, s));
    // under the certificate a query whose guessed threshold was too tight goes to the exact scan instead of a gated second scan + merge
    // (ZVDB_BF_CERT_DEFER=0: the gated pair, as without the certificate)
    const bool cert_defer = cert && [] { const char *e = tuning("ZVDB_BF_CERT_DEFER"); return !e || e[0] != '0'; }();
    // ... for at most this many queries of a batch (an exact scan per query: ~0.8 ms each at 10M x 768); a batch with more of them
    // is repaired by the gated scan + merge pair, which then follows the verifying merge as two no-op launches
    // The fp32 screen defers EVERY miss: its scan runs at the fp32-MFMA rate, so an exact scan per query of a whole batch costs ~7 scans
    // (bounded; measured 3.5 ms for 256 queries at 1M x 128 against a 0.5-ms step), while the two no-op launches of the pair cost
    // 9 us of EVERY 0.52-ms configs[1] step.  On the bf16 / int8 screens the same ratio is 80 - 160 scans: they keep the pair.
    const uint32_t defer_max = [&] { const char *e = tuning("ZVDB_BF_DEFER_MAX"); return e ? (uint32_t)atoi(e) : mode == ZVDB_BF_F32 ? 0xFFFFFFFFu : 32u; }();
    auto finish = [&](uint32_t kp_valid) -> int {
        if (!cert) return launch_rerank(h, d_queries_exact, nq, cand_ids, kpv, k, id_offset, d_ids, d_dist, d_count, s, kp_valid);
        CertArgs ca;
        ca.kth = kth;
        ca.keys = ckeys;
        ca.stats = reinterpret_cast<const float *>(h->cert_dev + 2 * mode);
        ca.nflag = nflag;
        ca.flist = flist;
        ca.host_flag = h->cert_hdev;
        ca.mode = (int)mode;
        ca.q_lo = h->q_lo;
        ca.q_inv = h->q_inv;
        ca.affine = l2 ? 1 : 0;
        ca.rescale_target = mode == ZVDB_BF_I8 ? 127.0f / h->q_inv : 0.0f;
        ca.force = h->cert_force ? 1 : 0;
        if (const char *dbg = tuning("ZVDB_CERT_DEBUG")) { ca.dbg_rows = h->cert_hdev + 2; ca.dbg_mode = dbg[0] == '2' ? 2 : dbg[0] == '3' ? 3 : 1; }
        ZV_CHECK(launch_rerank(h, d_queries_exact, nq, cand_ids, kpv, k, id_offset, d_ids, d_dist, d_count, s, kp_valid, &ca));
        return launch_exact_scan(h, d_queries_exact, nq, k, id_offset, nflag, flist, cand, p.cand_bytes, d_ids, d_dist, d_count, h->cert_hdev + 1, s);
    };
    timing_begin(h, s);
    {
        // The passes share one exact threshold: each pass but the last is followed by a merge of all lists;
        // a merge of its per-stream lists yields, per query, the exact k'-th best screen key so far
        // (tau_G); pass B scans the rest against it (folded in at every tile start, together with
        // whatever tighter bound other workgroups publish meanwhile).  Without it every workgroup
        // re-learns a threshold from its own 1/64 of the data and appends ~100x more candidates.
        uint32_t *tau_g = reinterpret_cast<uint32_t *>(ws + p.off_taug);
        uint32_t *pcnt = reinterpret_cast<uint32_t *>(ws + p.off_pcnt);
        // survivors per query: 32 suffice for k <= 16 with the (near-)exact screens; the 4-bit screen keeps 64
        const uint32_t fp4_kp = [] { const char *e = tuning("ZVDB_BF_FP4_KP"); return e ? (uint32_t)atoi(e) : (uint32_t)KP; }();  // tuning
        // one of G id-range shards: its share of the global 64 best screen keys is Binomial(64, 1/G) -> mean + 6 sigma
        // (zvdb_hip_index_set_shard_count; G = 8: 24 survivors instead of 64)
        uint32_t fp4_share = fp4_kp;
        if (h->shard_count > 1) {
            const double pg = 1.0 / (double)h->shard_count, mean = (double)fp4_kp * pg;
            fp4_share = std::max<uint32_t>((uint32_t)std::ceil(mean + 6.0 * std::sqrt(mean * (1.0 - pg))), k + 8);
        }
        const uint32_t kp = kpv != (uint32_t)KP ? kpv
                            : mode == ZVDB_BF_FP4 ? std::min<uint32_t>(h->shard_count > 1 ? fp4_share : std::max<uint32_t>(fp4_kp, k + 16), KP)
                                                : (k <= 16 && !(debug_flags() & 8) && cert_lvl == 0) ? 32u : (uint32_t)KP;
        kp_used = kp;
        const unsigned char *rows = mode == ZVDB_BF_BF16 ? reinterpret_cast<const unsigned char *>(h->rows_h)
                                    : mode == ZVDB_BF_I8 ? reinterpret_cast<const unsigned char *>(h->rows_q)
                                    : mode == ZVDB_BF_FP4 ? reinterpret_cast<const unsigned char *>(h->rows_p)
                                                         : reinterpret_cast<const unsigned char *>(h->rows);
        const void *vn = !l2 ? nullptr : mode == ZVDB_BF_BF16 ? (const void *)h->norm_h
                                       : mode == ZVDB_BF_I8 ? (const void *)h->norm_q
                                       : mode == ZVDB_BF_FP4 ? (const void *)h->norm_p : (const void *)h->norm;
        // pass schedule: {first tile, tiles, streams}; after each pass but the last, a merge of ALL
        // lists so far gives, per query, the exact k'-th best key over the rows scanned so far = the
        // shared threshold of the next pass (learning a threshold costs one flooded tile + one sort per
        // WORKGROUP however long its stream is, so it is paid once, on one tile per stream):
        //   A: 1 tile per stream   -> threshold over S*256 rows (every stream floods + sorts once)
        //   B: ~1/32 of the corpus -> threshold over ~N/32 rows (no flood any more)
        //   C: the rest            -> ~32x fewer candidates than B's threshold would admit
        struct Pass { uint32_t t0, nt, S; };
        Pass passes[3];
        int np = 0;
        const uint32_t Stot = 3 * p.S;
        // ---- fp4 stream kernel, large corpus: ONE scan against GUESSED thresholds, then verification ------------
        // The cascade below learns exact thresholds by flooding a tile per stream (pass A) and scanning N/32 rows
        // against the loose result (pass B): ~0.8 ms of appends / merges at 10M x 768.  Instead: (1) moments of the
        // screen keys over one tile per stream -> tau = mean + z sigma with z = the normal quantile of target / N
        // (k_fp4_guess); (2) one scan of everything against tau; (3) the merge VERIFIES: a query left with fewer
        // than kp candidates has no proof that it holds the kp best -> (4) gated repair scan + merge re-read the corpus
        // for those queries with no bound (exact; no-ops when the guess held, which is the common case).  A guess that
        // is too loose is self-correcting (the append buffers fill and the prune rounds tighten the bound).
        float z = 0.0f;
        if (guess_pipe || staged_guess) {
            const double target = [] { const char *e = tuning("ZVDB_BF_GUESS_TARGET"); return e ? atof(e) : 384.0; }();
            const double zshift = [] { const char *e = tuning("ZVDB_BF_GUESS_ZSHIFT"); return e ? atof(e) : 0.0; }();  // tests: < 0 forces repairs
            // candidates admitted per query: 6 per survivor wanted (384 for the usual 64; fewer survivors, fewer appends)
            const double tail = std::min(0.25, std::max(std::min(target, target * kp / 64.0), 4.0 * kp) / (double)N);
            // adaptive part: a batch that needed a repair (keys not Gaussian enough, e.g. low dimensions) loosens the guess of
            // the following batches by half a standard deviation, up to +2 (too loose is self-correcting, only slower)
            if (!h->guess_flag_host) {
                ZV_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->guess_flag_host), 64, hipHostMallocMapped | hipHostMallocCoherent));  // (fine-grained: read while kernels run)
                if (!h->guess_flag_host) { set_error("hipHostMalloc returned no block"); return ZVDB_ERR_OOM; }
                h->guess_flag_host[0] = 0;
                ZV_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->guess_flag_dev), h->guess_flag_host, 0));
            }
            if (__atomic_load_n(h->guess_flag_host, __ATOMIC_RELAXED) != 0) {
                __atomic_store_n(h->guess_flag_host, 0u, __ATOMIC_RELAXED);
                h->guess_adapt = std::min(2.0f, h->guess_adapt + 0.5f);
            }
            z = (float)(norm_quantile(tail) + zshift + (zshift == 0.0 ? (double)h->guess_adapt : 0.0));
        }
        if (guess_pipe) {
            uint32_t *vflag = reinterpret_cast<uint32_t *>(ws + p.off_verify);
            uint32_t *defq = reinterpret_cast<uint32_t *>(ws + p.off_verify + 256);
            uint32_t *tau_rep = reinterpret_cast<uint32_t *>(ws + p.off_verify + 256 + round_up64((uint64_t)p.nq_pad * 4, 256));
            const bool t128 = s8 || s4 || sf;         // 128-query tiles (QT8 x S8 workgroups)
            const uint32_t SS = t128 ? p.S8 : p.S;  // streams = lists per query and scan
            float *mom = reinterpret_cast<float *>(cand);  // (cand is free until the scan)
            if (sb) {
                ZV_CHECK(launch_bf16s_guess(h, q_op, nq, p.QT, SS, mom, z, tau_g, vflag, s));
                ZV_CHECK(launch_bf16s_scan(h, q_op, nq, p.QT, SS, cand, partial, tau_g, kp, 0, Stot, pcnt, nullptr, kpv, s));
            } else if (sr) {  // (moments by the 128-query kernel's sampling mode: same mirror, same packed queries, thresholds per global query)
                if (sr_moments) ZV_CHECK(launch_f32rq_guess(h, d_queries, nq, p.nq_pad, q_op, z, tau_g, vflag, s));
                else ZV_CHECK(launch_f32s_guess(h, q_op, nq, p.QT8, p.S8, p.NT, mom, z, tau_g, vflag, s));
                ZV_CHECK(launch_f32rq_scan(h, q_op, nq, p.QT, SS, cand, partial, tau_g, kp, 0, Stot, pcnt, nullptr, kpv, s));
            } else if (s8) {
                ZV_CHECK(launch_i8s_guess(h, q_op, nq, p.QT8, SS, p.NT, mom, z, tau_g, vflag, s));
                ZV_CHECK(launch_i8s_scan(h, q_op, nq, p.QT8, SS, p.NT, cand, partial, tau_g, kp, 0, Stot, pcnt, nullptr, kpv, s));
            } else if (sf) {
                ZV_CHECK(launch_f32s_guess(h, q_op, nq, p.QT8, SS, p.NT, mom, z, tau_g, vflag, s));
                ZV_CHECK(launch_f32s_scan(h, q_op, nq, p.QT8, SS, p.NT, cand, partial, tau_g, kp, 0, Stot, pcnt, nullptr, kpv, s));
            } else if (s4) {
                ZV_CHECK(launch_fp4s128_guess(h, q_op, nq, p.QT8, SS, p.NT, mom, z, tau_g, vflag, s));
                ZV_CHECK(launch_fp4s128_scan(h, q_op, nq, p.QT8, SS, p.NT, cand, partial, tau_g, kp, 0, Stot, pcnt, nullptr, kpv, s));
            } else {
                ZV_CHECK(launch_fp4s_guess(h, q_op, nq, p.QT, SS, p.NT, mom, z, tau_g, vflag, s));
                ZV_CHECK(launch_fp4s_scan(h, q_op, nq, p.QT, SS, p.NT, cand, partial, tau_g, kp, 0, 0, Stot, pcnt, nullptr, kpv, s));
            }
            const uint32_t mlw = kpv == (uint32_t)KP ? 64u : 16u, cap = kp + mlw * kpv;
            const size_t mlds = SEL_HIST_BYTES + std::max<size_t>((size_t)kpv * 8, SEL_STAGE_BYTES) + (size_t)cap * 8;
            if (mlds > 64 * 1024)
                ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bf_merge_sel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
            if (cert_defer) {  // a guess that was too tight for a FEW queries is repaired by the certificate's exact scan ...
                hipLaunchKernelGGL(k_bf_merge_sel, dim3(nq), dim3(64), mlds, s, partial, pcnt, SS, Stot, cand_ids, tau_rep,
                                   kp, cap, 0u, 3, vflag, defq, h->guess_flag_dev, kpv, ckeys, kth, nflag, mlw, h->cert_hdev + 10, defer_max);
                if (nq <= defer_max) {  // (... and a batch this small cannot hold more than a few: two launches fewer)
                    timing_end(h, s);
                    ZV_HIP(hipGetLastError());
                    return finish(kp);
                }
            } else {
                hipLaunchKernelGGL(k_bf_merge_sel, dim3(nq), dim3(64), mlds, s, partial, pcnt, SS, Stot, cand_ids, tau_rep,
                                   kp, cap, 0u, 1, vflag, defq, h->guess_flag_dev, kpv, ckeys, kth, nflag, mlw);
            }
            if (sb) ZV_CHECK(launch_bf16s_scan(h, q_op, nq, p.QT, SS, cand, partial, tau_rep, kp, SS, Stot, pcnt, vflag, kpv, s));
            else if (sr) ZV_CHECK(launch_f32rq_scan(h, q_op, nq, p.QT, SS, cand, partial, tau_rep, kp, SS, Stot, pcnt, vflag, kpv, s));
            else if (s8) ZV_CHECK(launch_i8s_scan(h, q_op, nq, p.QT8, SS, p.NT, cand, partial, tau_rep, kp, SS, Stot, pcnt, vflag, kpv, s));
            else if (sf) ZV_CHECK(launch_f32s_scan(h, q_op, nq, p.QT8, SS, p.NT, cand, partial, tau_rep, kp, SS, Stot, pcnt, vflag, kpv, s));
            else if (s4) ZV_CHECK(launch_fp4s128_scan(h, q_op, nq, p.QT8, SS, p.NT, cand, partial, tau_rep, kp, SS, Stot, pcnt, vflag, kpv, s));
            else ZV_CHECK(launch_fp4s_scan(h, q_op, nq, p.QT, SS, p.NT, cand, partial, tau_rep, kp, 0, SS, Stot, pcnt, vflag, kpv, s));
            hipLaunchKernelGGL(k_bf_merge_sel, dim3(nq), dim3(64), mlds, s, partial, pcnt, SS, Stot, cand_ids,
                               (uint32_t *)nullptr, kp, cap, SS, 2, vflag, defq, (uint32_t *)nullptr, kpv, ckeys, kth, nflag, mlw);
            timing_end(h, s);
            ZV_HIP(hipGetLastError());
            return finish(kp);
        }
        if (staged_guess) {
            uint32_t *vflag = reinterpret_cast<uint32_t *>(ws + p.off_verify);
            uint32_t *defq = reinterpret_cast<uint32_t *>(ws + p.off_verify + 256);
            uint32_t *tau_rep = reinterpret_cast<uint32_t *>(ws + p.off_verify + 256 + round_up64((uint64_t)p.nq_pad * 4, 256));
            unsigned long long *dbg_area = reinterpret_cast<unsigned long long *>(tau_g + p.nq_pad);  // (never written in production builds)
            const uint32_t nsamp_env = [] { const char *e = tuning("ZVDB_BF_SAMPLE"); return e ? (uint32_t)atoi(e) : 0u; }();  // (tuning)
            const uint32_t nsamp = nsamp_env ? nsamp_env : 4096;
            ZV_CHECK(rows_dispatch(h, [&](auto rws, uint32_t ld) -> int {
                using RT = row_elem_t<decltype(rws)>;
                const size_t lds = (size_t)SG_Q * ld * 4 + 4 * SG_Q * 2 * 8 + 16;
                if (lds > 64 * 1024) ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sample_guess<RT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                const uint32_t groups = (nq + SG_Q - 1) / SG_Q;
                double *part = reinterpret_cast<double *>(cand);  // (cand is free until the scan)
                const uint32_t ns_eff = std::min(nsamp, N) >= 64 ? std::min(nsamp, N) / 64 * 64 : std::min(nsamp, N);
                hipLaunchKernelGGL(k_sample_guess<RT>, dim3(groups, SG_SPLIT), dim3(256), lds, s, rws, ld, h->dim,
                                   l2 ? ((h->flags & ZVDB_FLAG_BF16_ROWS) ? h->norm_h : h->norm) : (const float *)nullptr, N, d_queries_exact, nq, nsamp, z, vflag, part,
                                   prep_folded ? q_op : (unsigned char *)nullptr, p.nq_pad, p.row_elems, (int)mode);
                hipLaunchKernelGGL(k_sample_tau, dim3((nq + 255) / 256), dim3(256), 0, s, part, groups, nq, ns_eff, z, tau_g);
                ZV_HIP(hipGetLastError());
                return ZVDB_OK;
            }));
            auto scan1 = [&](uint32_t *tau, uint32_t list_base, const uint32_t *gate) -> int {
                if (mode == ZVDB_BF_BF16) return launch_scan5<1, 2, 128>(p, q_op, rows, vn, N, nq, alpha_i, cand, partial, tau, kp, 0, p.NT, list_base, Stot, pcnt, s, gate, dbg_area);
                // one query tile: no other workgroup shares a stream's L2 lines, every slice is an HBM round trip -- the 64-byte slices'
                // 4-stage ring (three in flight) covers it better than two 128-byte stages: +3 % at 1M x 128, +2 % at 256-d
                // (ZVDB_BF_F32_KBV=128 / 64 forces either)
                const int kbv_env = [] { const char *e = tuning("ZVDB_BF_F32_KBV"); return e ? atoi(e) : 0; }();
                const bool kbv64 = kbv_env ? kbv_env == 64 : p.QT == 1;
                if (kbv64) return launch_scan5<0, 2, 64>(p, q_op, rows, vn, N, nq, alpha_i, cand, partial, tau, kp, 0, p.NT, list_base, Stot, pcnt, s, gate, dbg_area);
                return launch_scan5<0, 2, 128>(p, q_op, rows, vn, N, nq, alpha_i, cand, partial, tau, kp, 0, p.NT, list_base, Stot, pcnt, s, gate, dbg_area);
            };
            ZV_CHECK(scan1(tau_g, 0, nullptr));
            if (debug_flags() & (16 | 32)) {  // diagnostic build of the scan: counters / cycle sums of the one pass
                unsigned long long c[12];
                ZV_HIP(hipMemcpyAsync(c, dbg_area, sizeof(c), hipMemcpyDeviceToHost, s));
                ZV_HIP(hipStreamSynchronize(s));
                const double nb = (double)p.QT * p.S;
                fprintf(stderr, "[zvdb dbg] guessed-threshold pass (tiles %u, streams %u): wave-tiles %llu, with-pass %llu, prunes %llu, appends %llu | cycles/blk (wave 0): "
                        "kernel %.0f, phaseA %.0f, phaseB %.0f; slice loop: wait+barrier %.0f, reads+DMA issue %.0f, k-steps %.0f\n",
                        p.NT, p.S, c[0], c[1], c[2], c[3], c[4] / nb, c[6] / nb, c[7] / nb, c[9] / nb, c[10] / nb, c[11] / nb);
                ZV_HIP(hipMemsetAsync(dbg_area, 0, 128, s));
            }
            const uint32_t mlw = kpv == (uint32_t)KP ? 64u : 16u, cap = kp + mlw * kpv;
            const size_t mlds = SEL_HIST_BYTES + std::max<size_t>((size_t)kpv * 8, SEL_STAGE_BYTES) + (size_t)cap * 8;
            if (mlds > 64 * 1024)
                ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bf_merge_sel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
            if (cert_defer) {
                hipLaunchKernelGGL(k_bf_merge_sel, dim3(nq), dim3(64), mlds, s, partial, pcnt, p.S, Stot, cand_ids, tau_rep, kp, cap, 0u, 3, vflag, defq,
                                   h->guess_flag_dev, kpv, ckeys, kth, nflag, mlw, h->cert_hdev + 10, defer_max);
                if (nq <= defer_max) {
                    timing_end(h, s);
                    ZV_HIP(hipGetLastError());
                    return finish(kp);
                }
            } else {
                hipLaunchKernelGGL(k_bf_merge_sel, dim3(nq), dim3(64), mlds, s, partial, pcnt, p.S, Stot, cand_ids, tau_rep, kp, cap, 0u, 1, vflag, defq,
                                   h->guess_flag_dev, kpv, ckeys, kth, nflag, mlw);
            }
            ZV_CHECK(scan1(tau_rep, p.S, vflag));
            hipLaunchKernelGGL(k_bf_merge_sel, dim3(nq), dim3(64), mlds, s, partial, pcnt, p.S, Stot, cand_ids, (uint32_t *)nullptr, kp, cap, p.S, 2, vflag,
                               defq, (uint32_t *)nullptr, kpv, ckeys, kth, nflag, mlw);
            timing_end(h, s);
            ZV_HIP(hipGetLastError());
            return finish(kp);
        }
        if (kpv != (uint32_t)KP && stream_route) {  // (small shard, k > 48) one flooding scan of the stream kernel with 128-wide lists
            const bool t128 = s8 || s4 || sf;
            const uint32_t SS = t128 ? p.S8 : p.S;
            if (s8) ZV_CHECK(launch_i8s_scan(h, q_op, nq, p.QT8, SS, p.NT, cand, partial, tau_g, kp, 0, Stot, pcnt, nullptr, kpv, s));
            else if (sf) ZV_CHECK(launch_f32s_scan(h, q_op, nq, p.QT8, SS, p.NT, cand, partial, tau_g, kp, 0, Stot, pcnt, nullptr, kpv, s));
            else if (s4) ZV_CHECK(launch_fp4s128_scan(h, q_op, nq, p.QT8, SS, p.NT, cand, partial, tau_g, kp, 0, Stot, pcnt, nullptr, kpv, s));
            else if (sb) ZV_CHECK(launch_bf16s_scan(h, q_op, nq, p.QT, SS, cand, partial, tau_g, kp, 0, Stot, pcnt, nullptr, kpv, s));
            else if (sr) ZV_CHECK(launch_f32rq_scan(h, q_op, nq, p.QT, SS, cand, partial, tau_g, kp, 0, Stot, pcnt, nullptr, kpv, s));
            else ZV_CHECK(launch_fp4s_scan(h, q_op, nq, p.QT, SS, p.NT, cand, partial, tau_g, kp, 0, 0, Stot, pcnt, nullptr, kpv, s));
            const uint32_t mlw = kpv == (uint32_t)KP ? 64u : 16u, cap = kp + mlw * kpv;
            const size_t mlds = SEL_HIST_BYTES + std::max<size_t>((size_t)kpv * 8, SEL_STAGE_BYTES) + (size_t)cap * 8;
            ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bf_merge_sel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
            hipLaunchKernelGGL(k_bf_merge_sel, dim3(nq), dim3(64), mlds, s, partial, pcnt, SS, Stot, cand_ids, (uint32_t *)nullptr, kp, cap,
                               0u, 0, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, kpv, ckeys, kth, nflag, mlw);
            timing_end(h, s);
            ZV_HIP(hipGetLastError());
            return finish(kp);
        }
        if (p.NT >= 4 * p.S && !(debug_flags() & 64)) {
            // pass A streams: every workgroup of pass A floods one tile (256 candidates per query, 8-byte
            // scattered stores) -- the fewer the cheaper, as long as pass B's candidates (rows_B * k' /
            // rows_A per query) stay far below the sub-buffer capacity.  ZVDB_BF_SA overrides (tuning).
            const uint32_t sa_env = [] { const char *e = tuning("ZVDB_BF_SA"); return e ? (uint32_t)atoi(e) : 0u; }();
            const uint32_t SA = sa_env ? std::min(p.S, std::max(8u, sa_env / 8 * 8)) : std::min(p.S, 16u);
            uint32_t t0 = SA;
            passes[np++] = {0, SA, SA};
            const uint32_t ntB = p.NT / 32;
            if (ntB >= 2 * p.S) {
                passes[np++] = {t0, ntB, p.S};
                t0 += ntB;
            }
            passes[np++] = {t0, p.NT - t0, p.S};
        } else {
            passes[np++] = {0, p.NT, p.S};
        }
        auto scan = [&](const Pass &ps, uint32_t list_base) -> int {
            BfPlan q = p;
            q.S = ps.S;
            if (f7) return launch_fp4s_scan(h, q_op, nq, q.QT, ps.S, ps.nt, cand, partial, tau_g, kp, ps.t0, list_base, Stot, pcnt, nullptr, (uint32_t)KP, s);
            if (kpv == (uint32_t)KPW) {  // k > 48 (56) on the staged routes: the default tiling with 128-wide lists
                if (mode == ZVDB_BF_FP4) return launch_scan5<3, 2, 128, KPW>(q, q_op, rows, vn, N, nq, alpha_i, cand, partial, tau_g, kp, ps.t0, ps.nt, list_base, Stot, pcnt, s);
                if (mode == ZVDB_BF_BF16) return launch_scan5<1, 2, 128, KPW>(q, q_op, rows, vn, N, nq, alpha_i, cand, partial, tau_g, kp, ps.t0, ps.nt, list_base, Stot, pcnt, s);
                if (mode == ZVDB_BF_I8) return launch_scan5<2, 2, 128, KPW>(q, q_op, rows, vn, N, nq, alpha_i, cand, partial, tau_g, kp, ps.t0, ps.nt, list_base, Stot, pcnt, s);
                return launch_scan5<0, 2, 128, KPW>(q, q_op, rows, vn, N, nq, alpha_i, cand, partial, tau_g, kp, ps.t0, ps.nt, list_base, Stot, pcnt, s);
            }
            if (mode == ZVDB_BF_FP4) return launch_scan5<3, 2, 128>(q, q_op, rows, vn, N, nq, alpha_i, cand, partial, tau_g, kp, ps.t0, ps.nt, list_base, Stot, pcnt, s);
            if (mode == ZVDB_BF_BF16) return launch_scan5<1, 2, 128>(q, q_op, rows, vn, N, nq, alpha_i, cand, partial, tau_g, kp, ps.t0, ps.nt, list_base, Stot, pcnt, s);
            if (mode == ZVDB_BF_I8) return launch_scan5<2, 2, 128>(q, q_op, rows, vn, N, nq, alpha_i, cand, partial, tau_g, kp, ps.t0, ps.nt, list_base, Stot, pcnt, s);
            return launch_scan5<0, 2, 128>(q, q_op, rows, vn, N, nq, alpha_i, cand, partial, tau_g, kp, ps.t0, ps.nt, list_base, Stot, pcnt, s);
        };
        uint32_t lists = 0;
        for (int i = 0; i < np; ++i) {
            ZV_CHECK(scan(passes[i], lists));
            lists += passes[i].S;
            if (debug_flags() & (16 | 32)) {
                unsigned long long c[12];
                ZV_HIP(hipMemcpyAsync(c, ws + p.off_taug + (size_t)p.nq_pad * 4, sizeof(c), hipMemcpyDeviceToHost, s));
                ZV_HIP(hipStreamSynchronize(s));
                const double nb = (double)p.QT * passes[i].S;
                fprintf(stderr, "[zvdb dbg] pass %d slice loop cycles/blk (wave 0): wait+barrier %.0f, reads+DMA issue (+tile init) %.0f, k-steps %.0f\n",
                        i, c[9] / nb, c[10] / nb, c[11] / nb);
                fprintf(stderr, "[zvdb dbg] pass %d (tiles %u, streams %u): wave-tiles %llu, with-pass %llu, prunes %llu, appends %llu, rounds/blk %.1f | "
                        "cycles/blk: kernel %.0f, rounds %.0f, phaseA %.0f, phaseB %.0f\n",
                        i, passes[i].nt, passes[i].S, c[0], c[1], c[2], c[3], (double)c[8] / nb, c[4] / nb, c[5] / nb, c[6] / nb, c[7] / nb);
                ZV_HIP(hipMemsetAsync(ws + p.off_taug + (size_t)p.nq_pad * 4, 0, 128, s));
            }
            if (i + 1 < np)
                {
                    const uint32_t mlw = kpv == (uint32_t)KP ? 64u : 16u, cap = kp + mlw * kpv;
                    const size_t mlds = SEL_HIST_BYTES + std::max<size_t>((size_t)kpv * 8, SEL_STAGE_BYTES) + (size_t)cap * 8;
                    if (mlds > 64 * 1024)
                        ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bf_merge_sel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
                    hipLaunchKernelGGL(k_bf_merge_sel, dim3(nq), dim3(64), mlds, s, partial, pcnt, lists, Stot,
                                       (uint32_t *)nullptr, tau_g, kp, cap, 0u, 0, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, kpv, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, mlw);
                }
        }
        nlists = lists;
        stot_lists = Stot;
    }
    timing_end(h, s);
    ZV_HIP(hipGetLastError());
    if (debug_flags() & (16 | 32)) {  // diagnostic run (results are garbage)
        unsigned long long c[9];
        ZV_HIP(hipMemcpyAsync(c, ws + p.off_taug + (size_t)p.nq_pad * 4, sizeof(c), hipMemcpyDeviceToHost, s));
        ZV_HIP(hipStreamSynchronize(s));
        fprintf(stderr, "[zvdb dbg] wave-tiles %llu, with-pass %llu, prunes %llu, appends %llu | per block (wave 0, shader cycles): "
                "kernel %.0f, prune rounds %.0f (%.1f rounds), phase A %.0f, phase B %.0f\n",
                c[0], c[1], c[2], c[3], (double)c[4] / (p.QT * p.S), (double)c[5] / (p.QT * p.S),
                (double)c[8] / (p.QT * p.S), (double)c[6] / (p.QT * p.S), (double)c[7] / (p.QT * p.S));
    }
    {
        const uint32_t kpf = kpv != (uint32_t)KP ? kpv
                             : mode == ZVDB_BF_FP4 ? (uint32_t)KP : (k <= 16 && !(debug_flags() & 8) && cert_lvl == 0) ? 32u : (uint32_t)KP;
        const uint32_t mlw = kpv == (uint32_t)KP ? 64u : 16u, cap = kpf + mlw * kpv;
        const size_t mlds = SEL_HIST_BYTES + std::max<size_t>((size_t)kpv * 8, SEL_STAGE_BYTES) + (size_t)cap * 8;
        if (mlds > 64 * 1024)
            ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_bf_merge_sel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
        hipLaunchKernelGGL(k_bf_merge_sel, dim3(nq), dim3(64), mlds, s, partial,
                           reinterpret_cast<const uint32_t *>(ws + p.off_pcnt), nlists, stot_lists, cand_ids,
                           (uint32_t *)nullptr, kpf, cap, 0u, 0, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, kpv, ckeys, kth, nflag, mlw);
    }
    ZV_HIP(hipGetLastError());
    (void)kp_used;
    return finish(0);
}

}  // namespace zvdb
