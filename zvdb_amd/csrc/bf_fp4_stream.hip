// bf_fp4_stream.hip -- K4 "v7": barrier-free fp4 (OCP E2M1) brute-force screen.
//
// The LDS-staged 256 x 256 scan (bruteforce.hip) moves BOTH operands through L2 -> LDS for every tile and
// synchronises 8 waves once per k-slice; at fp4 rates (v_mfma_scale_f32_32x32x64_f8f6f4: 64 k per 32 cycles)
// that staging, not the MFMA pipe, sets the time (ablation in DESIGN.md).  Here
//
//   * the QUERY tile (256 queries) is loaded ONCE per workgroup into LDS, already in MFMA fragment order
//     ([group of 32 queries][k-step][lane][16 B]: a B fragment is one conflict-free linear ds_read_b128);
//   * the CORPUS mirror is stored in HBM in the same fragment order ([group of 32 rows][k-step][lane][16 B]),
//     so an A fragment is ONE fully coalesced 1-KiB global_load_dwordx4 straight into registers: no LDS
//     staging, no LDS-DMA, no ds_write, and -- because no wave ever reads LDS data another wave produced --
//     NO BARRIER in the steady state.  Every wave streams its own row groups (32 rows x 256 queries per
//     step, 8 accumulator fragments) with a 6-deep register prefetch ring;
//   * selection: the accumulator starts at -|v|^2/2 (first MFMA takes it as srcC), so the finished
//     accumulator is the negated screen key; a lane owns one query per fragment (8 per step), the max of its
//     16 registers is tested against the query's bound (one v_max3 chain + one compare per fragment); rare
//     passes append (key, id) to the query's buffer through an LDS slot counter.  A prune round (exact radix
//     select, shared with the staged kernel) is a workgroup rendezvous requested through an LDS flag.
//
// fp4 keys are exact: values are multiples of 1/2, products of 1/4, |v|^2/2 of 1/8, all far below 2^24, so
// fp32 accumulation is exact and equal rows get equal keys (ties -> lower id by the (key,id) select).
// The thresholds are applied NON-strictly (key <= k'-th best) because the waves of a workgroup do not walk
// the rows in id order.  No reference symbol (a10); parity: tests/test_gpu_bruteforce.py.
#include "common.h"
#include "wave_select.h"
#include "bf_select.h"
#include "stream_common.h"
#include "row_types.h"

namespace zvdb {

namespace {

constexpr int F7_NF = 8;                  // query fragments per wave step (8 x 32 = 256 queries per workgroup)
constexpr int F7_KC = 6;                  // k-steps per unrolled chunk = depth of the A prefetch ring
constexpr int F7_CAP = NSUB * SUBCAP;     // append-buffer entries per (workgroup, query)
constexpr int F7_HW = 384;                // an append at this slot requests a prune round (CAP - HW > 8 waves x 2 steps x 32 rows)
constexpr int F7_MARK = 64;               // a round merges the queries holding more than this many new entries
constexpr int F7_NW = 8;                  // waves per workgroup
constexpr bool kPrioMfma = true;          // the wave inside its MFMA section wins issue arbitration over its partner's epilogue

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x8 = __attribute__((ext_vector_type(8))) int;

// 8 floats at LDS byte address a + {0, 64, ..., 448}: one address register, immediates, one wait
__device__ __forceinline__ void lds_read8_f32(float *d, uint32_t a) {
    asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:64\n\tds_read_b32 %2, %8 offset:128\n\tds_read_b32 %3, %8 offset:192\n\t"
                 "ds_read_b32 %4, %8 offset:256\n\tds_read_b32 %5, %8 offset:320\n\tds_read_b32 %6, %8 offset:384\n\tds_read_b32 %7, %8 offset:448\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
                 : "v"(a));
}

// LDS map (dynamic): [Q tile: 8 * KS KiB][tau 256 f32][cnt 256][nkept 256][subcnt 1024][ctl 4][select scratch 8 waves]
__host__ __device__ inline uint32_t f7_lds_bytes(uint32_t KS, uint32_t sel_bytes = SEL_SCRATCH) {
    return F7_NF * KS * 1024 + T2 * 4 * 3 + T2 * NSUB * 4 + 16 + F7_NW * sel_bytes;
}

// KS = 64-element k-steps per row (compile time: every LDS offset of the resident tile is an instruction immediate)
// DBG: timing ablations (ZVDB_BF_DEBUG: 1 = no selection epilogue, 2 = no corpus loads, 4 = no query-fragment reads;
// results are garbage); compiled out of the production instantiation.
// DBG == 8: MOMENTS mode (not a debug build): instead of selecting, every wave sums key and key^2 of its rows per
// query -> mom[(block * 8 + wave)][256][2] (+ the row count behind it); k_fp4_guess turns them into thresholds.
// SH = MFMA shape: 32 -> v_mfma_scale_f32_32x32x64 (8 accumulators of 16 registers per 32-row step), 16 ->
// v_mfma_scale_f32_16x16x128 (2 x 16 accumulators of 4 registers): same bytes, same cadence (4 ds_read_b128 per 128
// MFMA cycles), but the chip holds a higher clock on the 16x16 shape (tools/mfma_peak_bench.hip, random operands:
// 7.6 vs 6.7 PFLOP/s).  The operand layouts differ (fragment = 16 rows x 128 k), so the mirror is packed per shape.
// KPV = width of the kept lists / output lists: 64, or 128 / 256 (two / four entries per lane in the selects) for k > 48 / 112.
// GATED = 1: the same code under its own symbol for the gated repair launch (a no-op unless the verification failed), so
// that per-kernel profiler statistics do not average 5-us no-ops into the scan.
template <int KS, int DBG, int SH, int KPV = KP, int GATED = 0>
__global__ __launch_bounds__(512, 2) void k_fp4_stream(
    const unsigned char *__restrict__ Qf /* [query group][KS][64][16 B] */,
    const unsigned char *__restrict__ Vf /* [row group][KS][64][16 B] */, const float *__restrict__ vnorm /* padded to groups, +inf past N */,
    uint32_t N, uint32_t nq, uint32_t QT, uint32_t S, uint32_t NT, uint64_t *__restrict__ cand,
    uint64_t *__restrict__ partial, uint32_t *__restrict__ tau_g_all, uint32_t kp, uint32_t t_first,
    uint32_t list_base, uint32_t Stot, uint32_t *__restrict__ pcnt, uint32_t max_tiles /* per stream; 0 = all */,
    const uint32_t *__restrict__ gate /* optional: the launch is a no-op unless *gate != 0 */, float *__restrict__ mom) {
    if (gate && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) return;
    constexpr bool MOM = DBG == 8;
    constexpr uint32_t dbg = MOM ? 0 : DBG;  // compile-time: a runtime flag inside the unrolled k-steps breaks the schedule
    static_assert(KS % F7_KC == 0 && (4 * KS - 1) * 1024 <= 65535, "ring wrap / ds_read immediate offsets");
    static_assert(SH == 32 || SH == 16, "MFMA shape");
    constexpr int K128 = KS / 2;  // SH == 16: 128-element k-steps per row
    constexpr int RD16 = (K128 % 2 == 0) ? 2 : 3;  // SH == 16: ring depth in 128-element k-steps (x 2 row fragments)
    constexpr int NRING = SH == 32 ? F7_KC : 2 * RD16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t qbytes = F7_NF * KS * 1024;
    float *tau = reinterpret_cast<float *>(smem + qbytes);           // exclusive bound: pass <=> key < tau
    uint32_t *cnt = reinterpret_cast<uint32_t *>(tau + T2);          // append slots handed out since the last merge
    uint32_t *nkept = cnt + T2;
    uint32_t *subcnt = nkept + T2;                                   // prune_query's view of the append buffer
    uint32_t *ctl = subcnt + T2 * NSUB;                              // [0] round request, [1] waves that finished their steps
    unsigned char *sel = reinterpret_cast<unsigned char *>(ctl + 4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t fr = lane & 31, fh = lane >> 5;

    // block -> (stream s, query tile qt): the QT workgroups of a stream share (blockIdx % 8), i.e. one XCD under
    // round-robin placement, and read the same row groups at about the same time (one HBM fetch, QT - 1 L2 hits)
    const uint32_t b = blockIdx.x;
    const uint32_t x = b & 7, jb = b >> 3;
    const uint32_t qt = jb % QT;
    const uint32_t s = (jb / QT) * 8 + x;
    const uint32_t per = (NT + S - 1) / S;
    const uint32_t t_begin = t_first + min(s * per, NT);
    const uint32_t t_end = min(t_begin + (max_tiles ? min(per, max_tiles) : per), t_first + NT);
    const uint32_t g_begin = 8 * t_begin, g_end = min(8 * t_end, (N + 31) / 32);  // row groups of 32
    const uint32_t nsteps = g_end > g_begin ? (g_end - g_begin + F7_NW - 1) / F7_NW : 0;  // the same for every wave
    uint32_t *tau_g = tau_g_all ? tau_g_all + (size_t)qt * T2 : nullptr;
    uint64_t *kept = cand + (size_t)b * BLKV_ENTRIES;  // (regions are sized for the widest list)
    uint64_t *subs = kept + (size_t)T2 * KPMAX;

    for (int tq = tid; tq < T2; tq += 512) {
        const bool live = (qt * T2 + tq) < nq;
        const uint32_t g = (tau_g && live) ? __hip_atomic_load(tau_g + tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
        float t0 = live ? __builtin_inff() : -__builtin_inff();
        if (g != 0xFFFFFFFFu) t0 = fminf(t0, f32_unkey(g + 1));
        tau[tq] = t0;
        cnt[tq] = 0;
        nkept[tq] = 0;
    }
    for (int i = tid; i < T2 * NSUB; i += 512) subcnt[i] = 0;
    if (tid < 4) ctl[tid] = 0;
    // the query tile: a linear 8 * KS KiB copy (the host wrote it in fragment order), 1 KiB per LDS-DMA
    for (uint32_t p = wave; p < F7_NF * KS; p += F7_NW)
        __builtin_amdgcn_global_load_lds((gptr_t)(Qf + (size_t)qt * qbytes + (size_t)p * 1024 + lane * 16),
                                         (lptr_t)(smem + p * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // workgroup rendezvous: merge the append buffers that filled up, tighten tau, publish the k'-th best
    auto round = [&]() {
        __syncthreads();  // every wave is here, all appends have landed (vmcnt(0) is part of the fence)
        if (tid == 0) ctl[0] = 0;
        for (int q = wave; q < T2; q += F7_NW) {
            const uint32_t c = min(lds_load_u32(cnt + q), (uint32_t)F7_CAP);
            if (c > (uint32_t)F7_MARK) {
                if (lane < NSUB) subcnt[q * NSUB + lane] = min(c - min(c, (uint32_t)(lane * SUBCAP)), (uint32_t)SUBCAP);
                if (lane == 0) cnt[q] = 0;
                prune_query<false, true, KPV>(kept, subs, tau, nkept, subcnt + q * NSUB, q, lane, kp, tau_g, sel + wave * sel_scratch_bytes<KPV>());
            }
        }
        __syncthreads();
    };

    // ---- per-wave stream state: the 6-deep A ring runs ahead of the MFMAs across row groups -------------
    // (KS is a multiple of the ring depth, so the ring wraps into the wave's next group at a chunk boundary:
    // no control flow inside the unrolled k-steps)
    constexpr size_t gbytes = (size_t)KS * 1024;
    const uint32_t g_last = g_end - 1;  // (only used when nsteps > 0)
    const uint32_t lane16 = (uint32_t)lane * 16;  // uniform 64-bit base + this 32-bit offset: saddr-form loads, no 64-bit VGPR pointers
    u32x4 A[NRING];  // SH 32: 6 k-steps of 64; SH 16: 2 (3) k-steps of 128 x 2 row fragments
    if (nsteps > 0) {
        const gbytes_t p0 = uniform_ptr(Vf + (size_t)min(g_begin + (uint32_t)wave, g_last) * gbytes);
#pragma unroll
        for (int i = 0; i < NRING; ++i) A[i] = *(const __attribute__((address_space(1))) u32x4 *)(p0 + lane16 + i * 1024);
    }
    const uint32_t qlds = (uint32_t)(size_t)(lptr_t)smem + (uint32_t)lane * 16;
    const uint32_t scale1 = 0x7F7F7F7Fu;  // E8M0 127 = 2^0 for every 32-element block
    u32x4 BX[4], BY[4];
    // fragments f0 .. f0+3 of k-step J: ((f * KS + J) << 10) + lane * 16; two long-lived base registers (one per half
    // of the tile), everything else is an instruction immediate
    const uint32_t qb_lo = qlds, qb_hi = qlds + 4 * KS * 1024;
    // SH 16: fragments 4 qd .. 4 qd + 3 (16 queries each) of 128-element k-step J: ((f * K128 + J) << 10)
    auto rd4s = [&](u32x4 (&d)[4], auto jc, auto qdc) {
        constexpr int J = decltype(jc)::value, QD = decltype(qdc)::value;
        const uint32_t base = QD >= 2 ? qb_hi : qb_lo;  // qb_hi = fragment 8 = byte 8 * K128 KiB = 4 * KS KiB
        constexpr int F0 = 4 * (QD & 1);
        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"
                     "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                     : "v"(base), "n"(((F0 + 0) * K128 + J) * 1024), "n"(((F0 + 1) * K128 + J) * 1024),
                       "n"(((F0 + 2) * K128 + J) * 1024), "n"(((F0 + 3) * K128 + J) * 1024));
    };
    auto rd4 = [&](u32x4 (&d)[4], auto jc, auto hc) {
        constexpr int J = decltype(jc)::value;
        const uint32_t base = decltype(hc)::value ? qb_hi : qb_lo;
        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"
                     "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                     : "v"(base), "n"((0 * KS + J) * 1024), "n"((1 * KS + J) * 1024), "n"((2 * KS + J) * 1024), "n"((3 * KS + J) * 1024));
    };
    // HAZARD of the hand-written reads: hipcc believes the registers written by rd4 are valid when the asm statement ends, but
    // they land only behind wait4's lgkmcnt.  The build must therefore not spill or copy them in between: this holds while the
    // kernel has no VGPR spill inside the step loop (checked in the ISA: tools/cc_f7.sh; an experimental 12-wave build of the
    // int8 kernel that spilled 60 VGPRs returned wrong neighbours), and the bit-exact stream == staged tests would catch it.
    auto wait4 = [&](u32x4 (&d)[4]) {
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
    };
    using c0_t = std::integral_constant<int, 0>;
    using c1_t = std::integral_constant<int, 1>;
    if constexpr (SH == 32) {
        rd4(BX, c0_t{}, c0_t{});  // k-step 0 of the first step (kept in flight across steps: the tile never changes)
        rd4(BY, c0_t{}, c1_t{});
    } else {
        rd4s(BX, c0_t{}, c0_t{});
        rd4s(BY, c0_t{}, c1_t{});
    }

    constexpr int NQF = SH == 32 ? F7_NF : 16;  // query fragments per step
    float m1[NQF], m2[NQF];  // moments mode only
    uint32_t mrows = 0;
#pragma unroll
    for (int f = 0; f < NQF; ++f) { m1[f] = 0.0f; m2[f] = 0.0f; }

    float vn_next = 0.0f;
    if (nsteps > 0 && vnorm) vn_next = vnorm[(size_t)32 * min(g_begin + (uint32_t)wave, g_last) + fr];

    for (uint32_t it = 0; it < nsteps; ++it) {
        if (!MOM && lds_load_u32(ctl) != 0) round();
        const uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)(g_begin + wave + F7_NW * it));
        if (g >= g_end) continue;  // (wave-uniform; only the last iteration of some waves)
        // |v|^2 of the group's 32 rows: lane l < 32 holds row l's (loaded one step ahead, ONE register; a load issued
        // here would sit behind the A ring in the in-order vmcnt queue and waiting for it would drain the prefetch);
        // the accumulator start values are gathered from it by lane shuffles
        const float vn_mine = vn_next;
        if (vnorm) vn_next = vnorm[(size_t)32 * min(g + F7_NW, g_last) + fr];

        if constexpr (SH == 32) {
        // this step's bounds (one query per fragment per lane) and the shared threshold of my 32 queries
        // accumulator start = -|v|^2/2 of the register's corpus row: row = 32 g + (r&3) + 8 (r>>2) + 4 (lane>>5).
        // The 32 norms of the group come through SCALAR loads (uniform address): a vector load here would sit behind
        // the A ring in the in-order vmcnt queue and waiting for it would drain the whole prefetch.
        f32x16 init;
        if (vnorm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) init[r] = -0.5f * __shfl(vn_mine, (r & 3) + 8 * (r >> 2) + 4 * (int)fh);
        } else {  // dot / cosine: no norm term; rows past N must still never pass
            asm volatile("");  // (keeps this path a real branch: if-converted, its 48 VALU ops ran on every L2 step)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                init[r] = (32 * g + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4 * fh) < N ? 0.0f : -__builtin_inff();
        }

        // B fragments: two half sets (fragments 0-3 / 4-7) ping-pong one half k-step ahead of the MFMAs.  The reads
        // and their waits are hand-written (LDS returns in order: lgkmcnt(4) = "all but the 4 younger reads"), and
        // sched_barriers pin the order read -> MFMA -> read, otherwise hipcc hoists all 48 reads of a chunk.
        f32x16 acc[F7_NF];
        const gbytes_t cur = uniform_ptr(Vf + (size_t)g * gbytes);                      // wave-uniform bases
        const gbytes_t nxt = uniform_ptr(Vf + (size_t)min(g + F7_NW, g_last) * gbytes);  // past the end: a re-read, never used
        if constexpr (kPrioMfma) __builtin_amdgcn_s_setprio(1);
        static_for<KS>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr int JN = (J + 1) % KS;    // the query tile is resident: the read-ahead wraps into the next step
            constexpr int SLOT = J % F7_KC;
            const i32x4 xa = __builtin_bit_cast(i32x4, A[SLOT]);
            const i32x8 wa = {xa.x, xa.y, xa.z, xa.w, 0, 0, 0, 0};
            auto mma4 = [&](u32x4 (&bh)[4], int f0) {
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const i32x4 xb = __builtin_bit_cast(i32x4, bh[f]);
                    const i32x8 wb = {xb.x, xb.y, xb.z, xb.w, 0, 0, 0, 0};
                    acc[f0 + f] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, wb, J == 0 ? init : acc[f0 + f], 4, 4, 0,
                                                                                  scale1, 0, scale1);
                }
            };
            wait4(BX);
            __builtin_amdgcn_sched_barrier(0);
            mma4(BX, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(dbg & 4)) rd4(BX, std::integral_constant<int, JN>{}, c0_t{});
            wait4(BY);
            __builtin_amdgcn_sched_barrier(0);
            mma4(BY, 4);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(dbg & 4)) rd4(BY, std::integral_constant<int, JN>{}, c1_t{});
            // the ring slot is free again: k-step J + 6 of my stream (KS is a multiple of the ring depth)
            if constexpr (!(dbg & 2)) {
                if constexpr (J + F7_KC < KS) A[SLOT] = *(const __attribute__((address_space(1))) u32x4 *)(cur + lane16 + (J + F7_KC) * 1024);
                else A[SLOT] = *(const __attribute__((address_space(1))) u32x4 *)(nxt + lane16 + (J + F7_KC - KS) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
        });

        if constexpr (kPrioMfma) __builtin_amdgcn_s_setprio(0);
        // this step's bounds, one query per fragment per lane (read here, not before the MFMAs: 8 fewer live registers)
        if constexpr (dbg & 1) {
#pragma unroll
            for (int f = 0; f < F7_NF; ++f) asm volatile("" :: "v"(acc[f]));
            continue;
        }
        if constexpr (MOM) {  // acc = -key; rows past N sit at -inf and are left out
#pragma unroll
            for (int f = 0; f < F7_NF; ++f) {
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float a = acc[f][r] > -3.0e38f ? acc[f][r] : 0.0f;
                    s1 += a;
                    s2 += a * a;
                }
                m1[f] += s1;
                m2[f] += s2;
            }
            mrows += min(32u, N - 32 * g);
            continue;
        }
        float thr[F7_NF];
#pragma unroll
        for (int f = 0; f < F7_NF; ++f) thr[f] = -tau[f * 32 + fr];
        unsigned long long fany[F7_NF];
        unsigned long long any = 0;
#pragma unroll
        for (int f = 0; f < F7_NF; ++f) {
            float m = acc[f][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[f][r]);
            fany[f] = __ballot(m > thr[f]);
            any |= fany[f];
        }
        if (any) {
#pragma unroll
            for (int f = 0; f < F7_NF; ++f) {
                if (fany[f] == 0) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (acc[f][r] > thr[f]) {
                        uint32_t q = (uint32_t)f * 32 + fr;
                        asm volatile("" : "+v"(q));  // keep the rare path's address arithmetic inside the branch
                        const uint32_t slot = atomicAdd(cnt + q, 1u);
                        const uint32_t id = 32 * g + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4 * fh;
                        if (slot < (uint32_t)F7_CAP) subs[(size_t)q * F7_CAP + slot] = ((uint64_t)f32_key(-acc[f][r]) << 32) | id;
                        if (slot == (uint32_t)F7_HW) ctl[0] = 1;
                    }
                }
            }
        }
        } else {
        // ================= 16x16x128 shape =================
        // accumulator (a, f, r): corpus row 32 g + 16 a + 4 (lane >> 4) + r, query 16 f + (lane & 15)
        const uint32_t l15 = lane & 15, lq = lane >> 4;
        f32x4 init16[2];
        if (vnorm) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) init16[a][r] = -0.5f * __shfl(vn_mine, 16 * a + 4 * (int)lq + r);
        } else {
            asm volatile("");  // (keeps this path a real branch)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) init16[a][r] = (32 * g + 16 * a + 4 * lq + r) < N ? 0.0f : -__builtin_inff();
        }
        f32x4 acc16[2][16];
        const gbytes_t cur = uniform_ptr(Vf + (size_t)g * gbytes);
        const gbytes_t nxt = uniform_ptr(Vf + (size_t)min(g + F7_NW, g_last) * gbytes);
        // sub-step (J, QD): 4 query fragments x 2 row fragments = 8 MFMAs of 16 cycles; the B quarter sets ping-pong
        static_for<K128 * 4>([&](auto sc) {
            constexpr int SS = decltype(sc)::value, J = SS / 4, QD = SS % 4;
            constexpr int SN = (SS + 2) % (K128 * 4), JN = SN / 4, QN = SN % 4;  // the set freed now is needed two sub-steps on
            constexpr int SLOT = (J % RD16) * 2;                                // ring: RD16 k-steps x 2 row fragments
            auto &BS = (SS & 1) ? BY : BX;
            i32x8 wa[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const i32x4 xa = __builtin_bit_cast(i32x4, A[SLOT + a]);
                wa[a] = i32x8{xa.x, xa.y, xa.z, xa.w, 0, 0, 0, 0};
            }
            wait4(BS);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const i32x4 xb = __builtin_bit_cast(i32x4, BS[f]);
                const i32x8 wb = {xb.x, xb.y, xb.z, xb.w, 0, 0, 0, 0};
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc16[a][4 * QD + f] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        wa[a], wb, J == 0 ? init16[a] : acc16[a][4 * QD + f], 4, 4, 0, scale1, 0, scale1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(dbg & 4)) rd4s(BS, std::integral_constant<int, JN>{}, std::integral_constant<int, QN>{});
            if constexpr (QD == 3 && !(dbg & 2)) {  // both row fragments of k-step J are consumed: refill with k-step J + RD16
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    if constexpr (J + RD16 < K128) A[SLOT + a] = *(const __attribute__((address_space(1))) u32x4 *)(cur + lane16 + ((J + RD16) * 2 + a) * 1024);
                    else A[SLOT + a] = *(const __attribute__((address_space(1))) u32x4 *)(nxt + lane16 + ((J + RD16 - K128) * 2 + a) * 1024);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (dbg & 1) {
#pragma unroll
            for (int f = 0; f < 16; ++f) asm volatile("" :: "v"(acc16[0][f]), "v"(acc16[1][f]));
            continue;
        }
        if constexpr (MOM) {
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc16[a][f][r] > -3.0e38f ? acc16[a][f][r] : 0.0f;
                        s1 += v;
                        s2 += v * v;
                    }
                m1[f] += s1;
                m2[f] += s2;
            }
            mrows += min(32u, N - 32 * g);
            continue;
        }
        // bounds of my 16 queries (one per fragment): tau[16 f + (lane & 15)], read in two batches of 8 through ONE address
        // register + immediates (hipcc otherwise keeps 16 invariant addresses live across the whole step loop)
        const uint32_t tau_l = (uint32_t)(size_t)(lptr_t)tau + l15 * 4;
        unsigned long long any16 = 0;
        uint32_t fmask = 0;  // query fragments in which some lane passed
#pragma unroll
        for (int H = 0; H < 2; ++H) {
            float th8[8];  // (the rare path re-reads its bound: only 8 bounds are live at a time)
            lds_read8_f32(th8, tau_l + H * 512);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = 8 * H + i;
                const float m = fmaxf(fmaxf(fmaxf(acc16[0][f][0], acc16[0][f][1]), fmaxf(acc16[0][f][2], acc16[0][f][3])),
                                      fmaxf(fmaxf(acc16[1][f][0], acc16[1][f][1]), fmaxf(acc16[1][f][2], acc16[1][f][3])));
                const unsigned long long bal = __ballot(m > -th8[i]);
                any16 |= bal;
                fmask |= (bal != 0 ? 1u : 0u) << f;
            }
        }
        if (any16) {
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                if (!((fmask >> f) & 1u)) continue;
                const float thr_f = -tau[f * 16 + l15];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (acc16[a][f][r] > thr_f) {
                            uint32_t q = (uint32_t)f * 16 + l15;
                            asm volatile("" : "+v"(q));
                            const uint32_t slot = atomicAdd(cnt + q, 1u);
                            const uint32_t id = 32 * g + 16 * a + 4 * lq + r;
                            if (slot < (uint32_t)F7_CAP) subs[(size_t)q * F7_CAP + slot] = ((uint64_t)f32_key(-acc16[a][f][r]) << 32) | id;
                            if (slot == (uint32_t)F7_HW) ctl[0] = 1;
                        }
                    }
            }
        }
        }
    }

    if constexpr (MOM) {  // fixed slots, no atomics: the reduction order (and so the thresholds) is deterministic
        float *slot = mom + (size_t)(b * F7_NW + wave) * (T2 * 2);
        if constexpr (SH == 32) {
#pragma unroll
            for (int f = 0; f < F7_NF; ++f) {
                const float a1 = m1[f] + __shfl_xor(m1[f], 32), a2 = m2[f] + __shfl_xor(m2[f], 32);
                if (lane < 32) {
                    slot[(f * 32 + fr) * 2 + 0] = a1;
                    slot[(f * 32 + fr) * 2 + 1] = a2;
                }
            }
        } else {
#pragma unroll
            for (int f = 0; f < 16; ++f) {  // the 4 lanes lane & 15 + {0, 16, 32, 48} hold one query's row quarters
                float a1 = m1[f] + __shfl_xor(m1[f], 16), a2 = m2[f] + __shfl_xor(m2[f], 16);
                a1 += __shfl_xor(a1, 32);
                a2 += __shfl_xor(a2, 32);
                if (lane < 16) {
                    slot[(f * 16 + lane) * 2 + 0] = a1;
                    slot[(f * 16 + lane) * 2 + 1] = a2;
                }
            }
        }
        if (lane == 0) mom[(size_t)gridDim.x * F7_NW * (T2 * 2) + b * F7_NW + wave] = (float)mrows;
        return;
    }

    // ---- drain: finished waves keep serving round requests until every wave is done --------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) atomicAdd(ctl + 1, 1u);
    for (;;) {
        if (lds_load_u32(ctl) != 0) { round(); continue; }
        if (lds_load_u32(ctl + 1) >= (uint32_t)F7_NW) {
            // no wave appends any more; a request raised before the last wave finished is visible by now
            if (lds_load_u32(ctl) != 0) { round(); continue; }
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();

    // ---- stream end: the k' best of every query -> partial[q][list][KP] -----------------------------------------
    // Common case (guessed thresholds): a query holds a handful of new entries and kept + new <= KP, so there is
    // nothing to select: ONE lane per query copies them (32 queries of a wave in flight together; the wave-per-query
    // select below costs ~3 us of dependent round trips per query, ~100 us per workgroup).  The rest (floods, repair
    // scans) go through the exact radix select.
    {
        const uint32_t q = (uint32_t)wave + F7_NW * fr;  // lanes 0-31 <-> my 32 queries (lanes 32-63 idle)
        const uint32_t qg = qt * T2 + q;
        const bool mine = lane < 32 && qg < nq;
        const uint32_t c = mine ? min(lds_load_u32(cnt + q), (uint32_t)F7_CAP) : 0u;
        const uint32_t nk = mine ? nkept[q] : 0u;
        const bool simple = mine && nk + c <= (uint32_t)KPV;
        if (simple) {
            uint64_t *dst = partial + ((size_t)qg * Stot + list_base + s) * KPV;
            for (uint32_t e = 0; e < nk; ++e) dst[e] = kept[(size_t)q * KPV + e];
            for (uint32_t e = 0; e < c; ++e) dst[nk + e] = subs[(size_t)q * F7_CAP + e];
            // (entries past the count are not written: every reader of a partial list stops at pcnt; the fill was 62 scattered 8-byte stores per lane, ~20 us per launch)
            pcnt[(size_t)qg * Stot + list_base + s] = nk + c;
        }
        unsigned long long todo = __ballot(mine && !simple);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t q2 = (uint32_t)wave + F7_NW * (uint32_t)l;
            const uint32_t qg2 = qt * T2 + q2;
            const uint32_t c2 = min(lds_load_u32(cnt + q2), (uint32_t)F7_CAP);
            if (lane < NSUB) subcnt[q2 * NSUB + lane] = min(c2 - min(c2, (uint32_t)(lane * SUBCAP)), (uint32_t)SUBCAP);
            prune_query<false, true, KPV>(kept, subs, tau, nkept, subcnt + q2 * NSUB, q2, lane, kp, nullptr, sel + wave * sel_scratch_bytes<KPV>());
#pragma unroll
            for (int c2i = 0; c2i < KPV / 64; ++c2i)  // (the list as prune_query left it: compacted, ~0 padded)
                partial[((size_t)qg2 * Stot + list_base + s) * KPV + lane + 64 * c2i] = kept[(size_t)q2 * KPV + lane + 64 * c2i];
            if (lane == 0) pcnt[(size_t)qg2 * Stot + list_base + s] = nkept[q2];
        }
    }
}

// moments -> per-query screen threshold: tau = mean(key) + z * stddev(key), z = the standard-normal quantile of
// (target candidates) / N.  One wave per query; fixed summation order in fp64.  A GUESS: the scan verifies it (a
// query that ends with fewer than k' candidates is re-scanned exactly), so a bad guess costs time, never results.
__global__ __launch_bounds__(64) void k_fp4_guess(const float *__restrict__ mom, uint32_t nblocks, uint32_t QT, uint32_t S,
                                                  uint32_t nq, float z, uint32_t *__restrict__ tau_g,
        uint32_t *__restrict__ verify_flag /* cleared here: saves a memset launch */) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && verify_flag) { verify_flag[0] = 0; verify_flag[1] = 0; }  // ([1]: deficient queries of the batch)
    const int lane = threadIdx.x;
    const uint32_t qg = blockIdx.x, qt = qg / T2, q = qg % T2;
    double s1 = 0.0, s2 = 0.0, n = 0.0;
    for (uint32_t i = lane; i < S * F7_NW; i += 64) {
        const uint32_t st = i / F7_NW, w = i % F7_NW;
        const uint32_t b = ((st >> 3) * QT + qt) * 8 + (st & 7);  // inverse of the scan's block -> (stream, tile) map
        const float *slot = mom + (size_t)(b * F7_NW + w) * (T2 * 2);
        s1 += slot[q * 2 + 0];
        s2 += slot[q * 2 + 1];
        n += mom[(size_t)nblocks * F7_NW * (T2 * 2) + b * F7_NW + w];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
        n += __shfl_xor(n, off);
    }
    if (lane == 0 && qg < nq) {
        uint32_t out = 0xFFFFFFFFu;  // no bound
        if (n >= 64.0) {
            const double mu = s1 / n, var = s2 / n - mu * mu;
            if (var > 0.0) {
                const float tk = (float)(-mu + (double)z * sqrt(var));  // key = -acc
                if (tk == tk && fabsf(tk) < 3.0e38f) out = f32_key(tk);
            }
        }
        tau_g[qg] = out;
    }
}

// rows (fp32, row-major) -> fp4 fragment order: unit (group g, k-step j, lane l) = 16 bytes = the 32 codes of
// row 32 g + (l & 31), elements 64 j + 32 (l >> 5) + [0, 32), low nibble first.  Rows >= n and elements >= dim are 0.
// layout 1 / 2 (16x16x128 shape): a 1-KiB block = 16 rows x 128 elements, lane l = row (l & 15), elements 32 (l >> 4) +
// [0, 32); a 32-row group holds its KS blocks as [k-step][row half] (1: corpus, the order the stream consumes them) or
// [row half][k-step] (2: queries, so that a 16-query fragment's k-steps are contiguous in the resident tile).
template <typename RT>
__global__ __launch_bounds__(256) void k_fp4_pack_frag(const RT *__restrict__ src, uint32_t dim, uint32_t ld,
                                                       uint64_t n, uint64_t units, uint32_t KS, float mid, float scale,
                                                       unsigned char *__restrict__ out, int layout) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const uint32_t l = (uint32_t)(u & 63);
    const uint64_t gj = u >> 6;
    const uint32_t j = (uint32_t)(gj % KS);
    uint64_t row = (gj / KS) * 32 + (l & 31);
    uint32_t k0 = 64 * j + 32 * (l >> 5);
    if (layout != 0) {
        const uint32_t K128 = KS / 2;
        const uint32_t a = layout == 1 ? (j & 1) : j / K128, j128 = layout == 1 ? (j >> 1) : j % K128;
        row = (gj / KS) * 32 + 16 * a + (l & 15);
        k0 = 128 * j128 + 32 * (l >> 4);
    }
    u32x4 w = {0, 0, 0, 0};
    if (row < n) {
        const RT *v = src + row * ld;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const uint32_t k = k0 + e;
            const uint32_t code = k < dim ? fp4_e2m1_code((row_elem(v, k) - mid) * scale) : 0u;
            w[e >> 3] |= code << (4 * (e & 7));
        }
    }
    *reinterpret_cast<u32x4 *>(out + u * 16) = w;
}

// |v'|^2 of the quantised row (exact in fp32); rows in [n, n_pad) get +inf so that padded rows never pass
template <typename RT>
__global__ __launch_bounds__(256) void k_fp4_norms(const RT *__restrict__ rows, uint32_t dim, uint32_t ld, uint64_t n,
                                                   uint64_t n_pad, float mid, float scale, float *__restrict__ norm) {
    const int lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_pad) return;
    if (r >= n) {
        if (lane == 0) norm[r] = __builtin_inff();
        return;
    }
    const RT *v = rows + r * ld;
    float s = 0.0f;
    for (uint32_t c = lane; c < dim; c += 64) s += fp4_e2m1_sq(fp4_e2m1_code((row_elem(v, c) - mid) * scale));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) norm[r] = s;
}

}  // namespace

// k-steps of 64 elements per row of the fragment-order mirror: 6 / 12 (256-query kernel here), 18 / 24 (768 < dim <= 1536:
// the 128-query kernel in bf_i8_stream.hip)
uint32_t fp4s_ksteps(uint32_t dim) {
    const uint32_t k = (dim + 63) / 64;
    return k <= 12 ? round_up(k, F7_KC) : fp4s128_ksteps(dim);
}

// MFMA shape of the stream kernel (fixed per process: the mirror is packed for it).  Default 32x32x64.
// ZVDB_BF_FP4_SHAPE=16 selects the 16x16x128 form: the bare instruction sustains 7.6 vs 6.7 PFLOP/s on random operands
// (tools/mfma_peak_bench.hip), but in this kernel it measured 9 % SLOWER (3.76 vs 3.42 ms per 10M x 768 batch): a
// 16-cycle MFMA holds the SIMD's vector issue for half of its cycles, which starves the partner wave's selection
// epilogue and LDS reads.  Kept as a tested A/B variant.
int fp4s_shape() {
    const int v = [] { const char *e = tuning("ZVDB_BF_FP4_SHAPE"); return (e && e[0] == '1') ? 16 : 32; }();
    return v;
}

bool fp4s_eligible(const zvdb_hip_index *h) {
    const bool off = [] { const char *e = tuning("ZVDB_BF_FP4_KERNEL"); return e && e[0] == 'v' && e[1] == '6'; }();
    return !off && h->dim > 0 && fp4s_ksteps(h->dim) <= 12;  // instantiated: KS = 6 (dim <= 384), 12 (dim <= 768)
}

uint64_t fp4s_query_bytes(const zvdb_hip_index *h, uint32_t nq_pad) { return (uint64_t)nq_pad * fp4s_ksteps(h->dim) * 32; }

int ensure_fp4_frag_mirror(zvdb_hip_index *h, hipStream_t s) {
    if (h->rows_pf && h->count_pf == h->count) return ZVDB_OK;
    if (h->count == 0) return ZVDB_OK;
    const uint32_t KS = fp4s_ksteps(h->dim);
    const uint64_t rows_pad = round_up64(h->count, 256);
    if (h->cap_pf < rows_pad) {
        if (h->rows_pf) { ZV_HIP(hipDeviceSynchronize()); /* searches of this index may be in flight on OTHER streams: the mutex covers the host side only */ ZV_HIP(hipFree(h->rows_pf)); ZV_HIP(hipFree(h->norm_pf)); }
        h->rows_pf = nullptr;
        h->norm_pf = nullptr;
        const uint64_t ncap = round_up64(std::max<uint64_t>(h->count, h->cap), 256);
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->rows_pf), ncap * KS * 32));
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->norm_pf), ncap * sizeof(float)));
        h->cap_pf = ncap;
    }
    ZV_CHECK(fp4_range(h, s));
    const uint64_t units = rows_pad / 32 * KS * 64;
    ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = row_elem_t<decltype(rows)>;
        hipLaunchKernelGGL(k_fp4_pack_frag<RT>, dim3((uint32_t)((units + 255) / 256)), dim3(256), 0, s, rows, h->dim, ld, h->count, units, KS,
                           h->p_mid, h->p_scale, h->rows_pf, (fp4s_shape() == 16 && KS <= 12) ? 1 : 0);
        hipLaunchKernelGGL(k_fp4_norms<RT>, dim3((uint32_t)((rows_pad + 3) / 4)), dim3(256), 0, s, rows, h->dim, ld, h->count, rows_pad, h->p_mid,
                           h->p_scale, h->norm_pf);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    }));
    h->count_pf = h->count;
    return ZVDB_OK;
}

int launch_fp4s_prep_queries(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t nq_pad,
                             unsigned char *q_op, hipStream_t s) {
    const uint32_t KS = fp4s_ksteps(h->dim);
    const uint64_t units = (uint64_t)nq_pad / 32 * KS * 64;
    hipLaunchKernelGGL(k_fp4_pack_frag<float>, dim3((uint32_t)((units + 255) / 256)), dim3(256), 0, s, d_queries, h->dim, h->dim,
                       (uint64_t)nq, units, KS, h->p_mid, h->p_scale, q_op, (fp4s_shape() == 16 && KS <= 12) ? 2 : 0);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

namespace {
template <int DBG, int KPV = KP>
int launch_stream(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                  uint64_t *cand, uint64_t *partial, uint32_t *tau_g, uint32_t kp, uint32_t t_first, uint32_t list_base,
                  uint32_t Stot, uint32_t *pcnt, uint32_t max_tiles, const uint32_t *gate, float *mom, hipStream_t s) {
    const uint32_t KS = fp4s_ksteps(h->dim);
    const uint32_t lds = f7_lds_bytes(KS, (uint32_t)sel_scratch_bytes<KPV>());
    const float *vn = h->metric == ZVDB_METRIC_L2SQ ? h->norm_pf : (const float *)nullptr;
    auto go = [&](auto kern) -> int {
        ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(QT * S), dim3(512), lds, s, q_op, h->rows_pf, vn, (uint32_t)h->count, nq, QT, S, nt, cand,
                           partial, tau_g, kp, t_first, list_base, Stot, pcnt, max_tiles, gate, mom);
        return ZVDB_OK;
    };
    if constexpr (KPV != KP) {  // wide lists: production instantiations of the 32x32 shape only
        if (KS == 6) ZV_CHECK(go(k_fp4_stream<6, 0, 32, KPV>));
        else ZV_CHECK(go(k_fp4_stream<12, 0, 32, KPV>));
    } else if (fp4s_shape() == 16) {
        if (KS == 6) ZV_CHECK(go(k_fp4_stream<6, DBG == 8 ? 8 : 0, 16>));
        else ZV_CHECK(go(k_fp4_stream<12, DBG, 16>));
    } else if (DBG == 0 && gate) {
        if (KS == 6) ZV_CHECK(go(k_fp4_stream<6, 0, 32, KP, 1>));
        else ZV_CHECK(go(k_fp4_stream<12, 0, 32, KP, 1>));
    } else {
        if (KS == 6) ZV_CHECK(go(k_fp4_stream<6, DBG == 8 ? 8 : 0, 32>));
        else ZV_CHECK(go(k_fp4_stream<12, DBG, 32>));
    }
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}
}  // namespace

int launch_fp4s_scan(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                     uint64_t *cand, uint64_t *partial, uint32_t *tau_g, uint32_t kp, uint32_t t_first, uint32_t list_base,
                     uint32_t Stot, uint32_t *pcnt, const uint32_t *gate, uint32_t kpv, hipStream_t s) {
    if (kpv == (uint32_t)KPMAX)
        return launch_stream<0, KPMAX>(h, q_op, nq, QT, S, nt, cand, partial, tau_g, kp, t_first, list_base, Stot, pcnt, 0u, gate, nullptr, s);
    if (kpv == (uint32_t)KPW)
        return launch_stream<0, KPW>(h, q_op, nq, QT, S, nt, cand, partial, tau_g, kp, t_first, list_base, Stot, pcnt, 0u, gate, nullptr, s);
    const uint32_t dbg = [] { const char *e = tuning("ZVDB_BF_DEBUG"); return e ? (uint32_t)atoi(e) : 0u; }();
#define ZV_F7_ARGS h, q_op, nq, QT, S, nt, cand, partial, tau_g, kp, t_first, list_base, Stot, pcnt, 0u, gate, (float *)nullptr, s
    if (dbg == 1) return launch_stream<1>(ZV_F7_ARGS);
    if (dbg == 3) return launch_stream<3>(ZV_F7_ARGS);
    if (dbg == 5) return launch_stream<5>(ZV_F7_ARGS);
    if (dbg == 7) return launch_stream<7>(ZV_F7_ARGS);
    return launch_stream<0>(ZV_F7_ARGS);
#undef ZV_F7_ARGS
}

// sample moments of the screen keys (one 256-row tile per stream, spread over the corpus) -> tau_g[q] = guessed
// threshold that about `target` rows of the corpus pass.  mom: scratch of fp4s_moment_bytes().
uint64_t fp4s_moment_bytes(uint32_t QT, uint32_t S) { return ((uint64_t)QT * S * F7_NW * (T2 * 2) + (uint64_t)QT * S * F7_NW) * sizeof(float); }

int launch_fp4s_guess(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                      float *mom, float z, uint32_t *tau_g, uint32_t *verify_flag, hipStream_t s) {
    ZV_CHECK(launch_stream<8>(h, q_op, nq, QT, S, nt, nullptr, nullptr, nullptr, 0u, 0u, 0u, 0u, nullptr, 1u, nullptr, mom, s));
    hipLaunchKernelGGL(k_fp4_guess, dim3(QT * T2), dim3(64), 0, s, mom, QT * S, QT, S, nq, z, tau_g, verify_flag);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

}  // namespace zvdb
