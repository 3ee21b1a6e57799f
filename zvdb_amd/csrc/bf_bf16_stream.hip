// bf_bf16_stream.hip -- the bf16 MFMA screen of the brute-force route (north_star: "MFMA bf16 ... for the
// batched brute-force QxV^T contraction") for large shards: REGISTER-RESIDENT QUERIES, corpus streamed once per workgroup
// through an LDS ring.  No reference symbol (hnsw.zig has no brute force); results are pinned by the oracle's exact top-k
// through the exact fp32 re-rank that follows (rerank.hip:k_rerank_staged, hnsw.zig:182-192 arithmetic).
//
// Why not the resident-LDS-query design of the int8 / fp4 stream kernels: a bf16 row is 2 bytes per element, a 256-query tile
// of 768-d rows is 384 KB -- it does not fit LDS (160 KB).  It does fit the REGISTER file (512 KB per CU): a wave keeps the
// MFMA B operand of ITS 32 queries (all of K: 2 x 24 fragments = 192 VGPRs at 768-d), 8 waves = 256 queries per workgroup.
// The corpus mirror is stored in MFMA fragment order (below), so a 32-row group is 48 contiguous KiB that the workgroup
// copies HBM -> LDS once (linear LDS-DMA, 1 KiB per wave-instruction) into a 3-slot ring; every wave then reads every A
// fragment (conflict-free ds_read_b128 at base + immediate) and multiplies it with its own B registers.  Operand bytes per
// flop are half those of the staged 256 x 256 tile (only the corpus moves; L2 -> LDS at full MFMA rate: 16 B/clk/CU instead
// of 32, which is what capped the staged kernel at 0.42 of peak), the steady state has ONE barrier per 32-row group
// (3,072 MFMA cycles) and its LDS-DMA issues are spread over the first 60 % of the k-loop (one piece per 5 fragments).
// In the 512- / 768-d geometries (one group per unit) the operand ring runs AHEAD of that barrier and the selection of a
// 16-row block sits a few fragments behind the block, in the shadow of the following MFMAs ("FLOW loop", in the kernel).
// MFMA shape: v_mfma_f32_16x16x32_bf16 -- the chip is power-limited on random bf16 data and holds a ~10 % higher clock on
// this shape than on 32x32x16 (tools/bf16_regq_bench.hip: 1.54 vs 1.44 PFLOP/s for the same loop; profiles/r2/).
//
// Layout16 (corpus mirror rows_hf and packed queries alike): unit (block of 16 rows, k-step j of 32 elements, lane l) =
// 16 bytes = the 8 bf16 of row 16 blk + (l & 15), elements 32 j + 8 (l >> 4) + [0, 8); a block holds its K2 = KS / 2
// k-steps contiguously (K2 KiB); a 32-row group = 2 blocks = KS KiB; a ring slot ("unit") = NG groups (NG > 1 for short
// rows, so that a barrier interval stays ~48 fragments) + their 32 NG norms.
// Keys: acc starts at -|v'|^2 / 2 (0 for dot / cosine; -inf for rows past N) and accumulates q'.v' (bf16 products, fp32
// sums): acc = -key, pass <=> acc > -tau.  Selection state (per-(workgroup, query) kept list + append buffer in global
// memory, exact (key, id) radix select) is shared with the other scan kernels (bf_select.h); a wave owns its 32 queries,
// so appends and prune rounds need no workgroup rendezvous.
#include <vector>
#include <type_traits>
#include <cmath>
#include <cstring>
#include "common.h"
#include "wave_select.h"
#include "bf_select.h"
#include "stream_common.h"
#include "row_types.h"

namespace zvdb {

namespace {

constexpr int BH_NW = 8;                  // waves per workgroup (2 per SIMD)
constexpr int BH_R = 6;                   // A-fragment register ring of the plain loop (reads in flight per wave; 5: -1.6 %, 7: -3.4 %, 8 spills: measured)
constexpr int BH_RF = 4;                  // ... of the flow loop (the ring is never empty there; 6 spills a query fragment: -17 %)
constexpr int BH_CAP = NSUB * SUBCAP;     // append-buffer entries per (workgroup, query)
constexpr int BH_HW = 384, BH_MARK = 64;  // prune request mark; queries above MARK are pruned in a round
constexpr uint32_t BH_SAMPLE_GROUPS = 4;  // moments mode: 32-row groups sampled per stream (8: the launch takes 28 instead of 20 us, the guess is no better)

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int KS, int NG> struct BhGeo {
    static constexpr int F = KS * NG;                                   // A fragments (1 KiB each) per unit
    static constexpr int P = F / BH_NW;                                 // LDS-DMA pieces per wave and unit
    static constexpr uint32_t NORMB = (NG * 128 + 255) / 256 * 256;     // norms of a unit, in 256-byte DMA pieces
    static constexpr int NP = NORMB / 256;
    static constexpr uint32_t SLOT = F * 1024 + NORMB;
    static_assert(F % BH_NW == 0 && F % 8 == 0 && KS % 2 == 0, "piece distribution");
    static_assert((F - 1) * 1024 <= 65535, "ds_read immediate offsets");
};

template <int KS, int NG, int KPV>
__host__ __device__ constexpr uint32_t bh_lds_bytes() {
    return 3 * BhGeo<KS, NG>::SLOT + T2 * 4 * 3 + BH_NW * NSUB * 4 + 16 + sel_scratch_bytes<KPV>();
}

template <int OFF> __device__ __forceinline__ void bh_lds_read128(u32x4 &d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF));
}
// HAZARD (as in bf_i8_stream.hip): hipcc believes a register written by bh_lds_read128 is valid when the asm statement ends;
// it lands only behind bh_lgkm_wait, which is tied to the same register.  The kernel must not spill inside the unit loop.
template <int O0, int O1> __device__ __forceinline__ void bh_lds_read_norms(f32x4 &a, f32x4 &b, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(a), "=&v"(b) : "v"(addr), "n"(O0), "n"(O1));
}
template <int N> __device__ __forceinline__ void bh_lgkm_wait2(f32x4 &a, f32x4 &b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
// every outstanding LDS read has landed: ties ALL ring registers, so that code behind it (the cold append / prune path, which may
// spill or copy registers around its call) only ever sees landed values
template <int R> __device__ __forceinline__ void bh_lgkm_drain(u32x4 (&a)[R]) {
    static_assert(R == 6 || R == 4, "operand list below");
    if constexpr (R == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[R - 1]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
}
template <int N> __device__ __forceinline__ void bh_lgkm_wait2(u32x4 &d, f32x4 &a) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(d), "+v"(a) : "n"(N)); }
template <int N> __device__ __forceinline__ void bh_lgkm_wait3(u32x4 &d, f32x4 &a, f32x4 &b) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(d), "+v"(a), "+v"(b) : "n"(N)); }
template <int N> __device__ __forceinline__ void bh_lgkm_wait(u32x4 &d) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N)); }

// LDS-DMA in the saddr form: uniform base (SGPR pair) + 32-bit lane offset, LDS destination through M0.  Inline asm because
// hipcc builds a 64-bit vector address per piece from the builtin (a v_lshl_add_u64 and two more VGPRs in a loop that has none
// to spare): +3 % QPS at 10M x 768 (profiles/r3/bf16_stream_ab.txt).  These operations are invisible to hipcc's waitcnt pass;
// the ring's vmcnt bookkeeping is the explicit s_waitcnt pair at the top of the unit loop.
__device__ __forceinline__ void bh_dma16(uint32_t voff, gbytes_t sbase, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void bh_dma4(uint32_t voff, gbytes_t sbase, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

// Exact prune of the calling wave's 32 queries whose append buffer passed MARK.  Out of line and cold on purpose: the scan
// keeps 192 VGPRs of query fragments live, and an inlined radix select pushes the allocator into spilling them in the hot
// loop; a call confines the cost to this path (taken a handful of times per stream when the thresholds are guessed).
// One wave at a time: the select scratch is shared (the ring owns the rest of LDS).
template <int KPV>
__device__ __attribute__((noinline, cold)) void bh_prune_wave(uint64_t *kept, uint32_t *tau_g, uint32_t kp, uint32_t lds_state /* LDS address of tau_f */) {
    float *tau_f = (float *)(__attribute__((address_space(3))) float *)(size_t)lds_state;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(tau_f + T2);
    uint32_t *nkept = cnt + T2;
    uint32_t *subw = nkept + T2;
    uint32_t *ctl = subw + BH_NW * NSUB;
    unsigned char *sel = reinterpret_cast<unsigned char *>(ctl + 4);
    uint64_t *subs = kept + (size_t)T2 * KPMAX;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my appends have left the wave
    for (;;) {
        uint32_t got = 0;
        if (lane == 0) got = atomicCAS(ctl, 0u, 1u) == 0u ? 1u : 0u;
        if (__builtin_amdgcn_readfirstlane((int)got)) break;
        __builtin_amdgcn_s_sleep(4);
    }
    for (int qi = 0; qi < 32; ++qi) {
        const int q = wave * 32 + qi;
        const uint32_t c = min(lds_load_u32(cnt + q), (uint32_t)BH_CAP);
        if (c > (uint32_t)BH_MARK) {
            if (lane < NSUB) subw[wave * NSUB + lane] = min(c - min(c, (uint32_t)(lane * SUBCAP)), (uint32_t)SUBCAP);
            if (lane == 0) cnt[q] = 0;
            prune_query<false, true, KPV>(kept, subs, tau_f, nkept, subw + wave * NSUB, q, lane, kp, tau_g, sel);
        }
    }
    if (lane == 0) __hip_atomic_store(ctl, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// KS = 16-element k-steps per row (KS / 2 MFMA k-steps of 32); NG = 32-row groups per ring slot; MOM: moments mode (sum of
// key, key^2 per query over a sample instead of selecting); GATED = 1: the same code under its own symbol for the gated
// repair launch (a no-op unless the verification failed), so that profiler statistics do not average no-ops into the scan
// units per XCD: XCD x (= blockIdx % 8 under round-robin placement) scans units [off[x], off[x + 1]), split evenly over its S / 8
// streams; off[8] == 0: an even split over all S streams
struct BhSplit { uint32_t off[9]; uint32_t tag; };  // tag: this launch's number (1 .. 255, cyclic), returned with the durations

template <int KS, int NG, bool MOM, int KPV = KP, int GATED = 0>
__global__ __launch_bounds__(512, 2) void k_bf16_stream(
    const unsigned char *__restrict__ Qf /* Layout16, [query block of 16][K2][64][16 B] */, const unsigned char *__restrict__ Vf /* same, rows */,
    const float *__restrict__ vnorm /* [rows padded]: -1/2 sum v'^2 (-0: dot / cosine), -inf past N */, uint32_t N, uint32_t nq, uint32_t QT,
    uint32_t S, uint32_t U /* units of NG groups */, uint64_t *__restrict__ cand, uint64_t *__restrict__ partial,
    uint32_t *__restrict__ tau_g_all, uint32_t kp, uint32_t list_base, uint32_t Stot, uint32_t *__restrict__ pcnt,
    uint32_t max_units, const uint32_t *__restrict__ gate, float *__restrict__ mom, BhSplit split, uint32_t *__restrict__ dur_out) {
    if (gate && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) return;
    unsigned long long t_begin = 0;
    if (dur_out) t_begin = __builtin_amdgcn_s_memrealtime();
    using Geo = BhGeo<KS, NG>;
    constexpr int F = Geo::F, P = Geo::P, NP = Geo::NP, K2 = KS / 2;
    constexpr bool FLOW = NG == 1 && F % BH_RF == 0 && !MOM;  // (the 768-d geometry; see the unit loops below)
    constexpr int R = FLOW ? BH_RF : BH_R;
    constexpr uint32_t SLOT = Geo::SLOT;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    float *tau_f = reinterpret_cast<float *>(smem + 3 * SLOT);  // exclusive bound: pass <=> key < tau
    uint32_t *cnt = reinterpret_cast<uint32_t *>(tau_f + T2);    // append slots handed out since the query's last prune
    uint32_t *nkept = cnt + T2;
    uint32_t *subw = nkept + T2;                                 // per wave: prune_query's view of one append buffer
    uint32_t *ctl = subw + BH_NW * NSUB;                         // [0] lock of the shared select scratch

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t fc = lane & 15, fq = lane >> 4;  // accumulator column (query of a B set), row quarter

    // block -> (stream s, query tile qt): the QT workgroups of a stream share (blockIdx % 8), i.e. one XCD under round-robin
    // placement, and fetch the same units at about the same time (one HBM fetch, QT - 1 L2 hits: PMC-checked)
    const uint32_t b = blockIdx.x;
    const uint32_t x = b & 7, jb = b >> 3;
    const uint32_t qt = jb % QT;
    const uint32_t s = (jb / QT) * 8 + x;
    uint32_t per = (U + S - 1) / S, ulo = s * per, uhi = U;
    if (split.off[8]) {  // my XCD's range, my stream's share of it (ids still ascend along a stream)
        const uint32_t nsx = S >> 3;
        ulo = split.off[x];
        uhi = split.off[x + 1];
        per = (uhi - ulo + nsx - 1) / nsx;
        ulo += (jb / QT) * per;
    }
    const uint32_t u0 = min(ulo, uhi);
    const uint32_t u1 = min(u0 + (max_units ? min(per, max_units) : per), uhi);
    const uint32_t n = u1 - u0;
    uint32_t *tau_g = tau_g_all ? tau_g_all + (size_t)qt * T2 : nullptr;
    uint64_t *kept = cand + (size_t)b * BLKV_ENTRIES;
    uint64_t *subs = kept + (size_t)T2 * KPMAX;

    if constexpr (!MOM) {
        for (int tq = tid; tq < T2; tq += 512) {
            const bool live = (qt * T2 + tq) < nq;
            const uint32_t g = (tau_g && live) ? __hip_atomic_load(tau_g + tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
            float t0 = live ? __builtin_inff() : -__builtin_inff();
            if (g != 0xFFFFFFFFu) t0 = fminf(t0, f32_unkey(g + 1));
            tau_f[tq] = t0;
            cnt[tq] = 0;
            nkept[tq] = 0;
        }
        if (tid < 4) ctl[tid] = 0;
    }

    // B operand: my 32 queries = 2 blocks of 16, every k-step
    u32x4 B[KS];  // (loaded at every entry of the unit loop, below)
    const unsigned char *qsrc = Qf + ((size_t)(qt * (T2 / 16) + 2 * wave) * K2) * 1024 + lane * 16;
    const uint32_t lane16 = (uint32_t)lane * 16, lane4 = (uint32_t)lane * 4;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
    auto dma_unit = [&](uint32_t u, uint32_t slot) {  // prologue form: all pieces back to back
        const gbytes_t src = uniform_ptr(Vf + (size_t)u * F * 1024);
        const uint32_t dst = lds0 + slot * SLOT;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t piece = (uint32_t)wave + (uint32_t)BH_NW * p;
            bh_dma16(lane16, src + piece * 1024, dst + piece * 1024);
        }
        if (wave == 0) {
            const gbytes_t nsrc = uniform_ptr(reinterpret_cast<const unsigned char *>(vnorm + (size_t)u * NG * 32));
#pragma unroll
            for (int p = 0; p < NP; ++p) bh_dma4(lane4, nsrc + p * 256, dst + F * 1024 + p * 256);
        }
    };
    if (n > 0) {
        dma_unit(u0, 0);
        dma_unit(min(u0 + 1, u1 - 1), 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // selection state initialised, the prologue's units landed

    float thr0 = 0.0f, thr1 = 0.0f;  // -tau of my two queries (B set 0: query 32 wave + fc; set 1: + 16)
    float ms1[2] = {0.0f, 0.0f}, ms2[2] = {0.0f, 0.0f};  // moments mode
    uint32_t mrows = 0;

    const uint32_t lds_state = (uint32_t)(size_t)(lptr_t)tau_f;

    // The unit loop is entered once per launch and once more after every prune round of this wave.  A prune is a CALL (header of
    // bh_prune_wave) and nothing of the loop's register state lives across it: inside the loop hipcc spilled the query fragments
    // around the call and re-loaded them on the way back.  So the loop is LEFT for a prune, and every entry loads the fragments
    // and the bounds afresh and drains the counters once, ahead of the loop.
    // Static priority for the younger half of the workgroup, the arbitration loser of every SIMD pair otherwise
    // (MI355X_MICROARCH.md, "Two waves per SIMD", item 4): +0.4 .. 0.9 % (profiles/r3/bf16_stream_ab.txt)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    uint32_t slot = 0, it = 0;
    if constexpr (FLOW) {
    // FLOW loop (one 32-row group per unit, F a multiple of the ring depth: the 768-d geometry).  Two differences to the plain loop
    // below, +3 .. 5 % QPS together (profiles/r3/bf16_stream_ab.txt, r3_ab20 .. 22):
    // (1) the A ring and the norms run AHEAD of the barrier -- the first R fragments and the start values of unit it + 1 are
    //     requested in the tail of unit it, so the MFMAs behind the barrier start on operands that are already in registers.
    //     Legal because every wave drains ALL its pieces (units <= it + 1) ahead of the barrier at the top of unit it: behind
    //     that barrier unit it + 1 is complete in LDS.  (Ring depth 4: with 6 the loop spills a query fragment.)
    // (2) selection per ROW BLOCK, a few fragments after the block's last MFMA, in the shadow of the MFMAs that follow
    //     (MI355X_MICROARCH.md, "Two waves per SIMD", item 9: partners that reach their vector work together leave the matrix
    //     pipe idle): row block 0 behind fragment K2 + 3; row block 1 behind fragment 3 of the NEXT unit -- its accumulators are
    //     free until fragment K2, their start values are read straight into them behind the selection.
    u32x4 A[R];
    f32x4 nv0;         // start values of row block 0 (requested a unit ahead); row block 1's are read straight into its accumulators
    f32x4 acc[2][2];   // [row block][B set]
    while (it < n) {
#pragma unroll
        for (int j = 0; j < KS; ++j) B[j] = *reinterpret_cast<const u32x4 *>(qsrc + (size_t)j * 1024);
        thr0 = -tau_f[wave * 32 + fc];
        thr1 = -tau_f[wave * 32 + 16 + fc];
#ifdef ZVDB_ABL_NOPASS  // (timing-only ablation, WRONG results: nothing passes the selection test, so both sides of a barrier A/B select alike)
        thr0 = thr1 = __builtin_inff();
        asm volatile("" : "+v"(thr0), "+v"(thr1));  // (opaque: the MFMAs stay)
#endif
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0), as an instruction the waitcnt pass sees
        {   // prime the ring with unit `it` (visible: its pieces were drained ahead of a barrier this wave has passed)
            const uint32_t base = lds0 + slot * SLOT + lane16;
            asm volatile("ds_read_b128 %0, %1" : "=&v"(nv0) : "v"(lds0 + slot * SLOT + F * 1024 + 16 * fq));
            static_for<R>([&](auto jc) { constexpr int J = decltype(jc)::value; bh_lds_read128<J * 1024>(A[J], base); });
        }
        bool want_prune = false;
        acc[1][0] = f32x4{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};  // (no row block pending)
        acc[1][1] = acc[1][0];
        // selection of one finished row block (16 rows x my 32 queries): the rare pass path appends to the queries' buffers
        auto select_rb = [&](auto rbc, uint32_t row0) {
            constexpr int rb = decltype(rbc)::value;
            const float m0 = fmaxf(fmaxf(acc[rb][0][0], acc[rb][0][1]), fmaxf(acc[rb][0][2], acc[rb][0][3]));
            const float m1 = fmaxf(fmaxf(acc[rb][1][0], acc[rb][1][1]), fmaxf(acc[rb][1][2], acc[rb][1][3]));
            if (__ballot(m0 > thr0 || m1 > thr1)) {  // (__builtin_expect moves this path out of line: -2 %)
                bh_lgkm_drain(A);  // (the ring holds reads in flight: the append path may move registers)
                asm volatile("" : "+v"(nv0));
                // lane-derived values of this rare path are re-derived HERE from an opaque copy of lane16: hoisted out of the unit
                // loop they cost VGPRs the loop does not have -- hipcc spilled them and re-loaded them in this path, each re-load
                // behind an s_waitcnt vmcnt(0) that also waits for the LDS-DMA pieces in flight (30 scratch operations -> none)
                uint32_t l16 = lane16, one = 1u;
                asm volatile("" : "+v"(l16), "+v"(one));
                const uint32_t fc = (l16 >> 4) & 15u, fq = l16 >> 8;
                bool need = false;
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (acc[rb][qb][r] > (qb ? thr1 : thr0)) {
                            uint32_t q = (uint32_t)wave * 32 + (uint32_t)qb * 16 + fc;
                            const uint32_t sl = atomicAdd(cnt + q, one);
                            const uint32_t id = row0 + 16u * rb + 4u * fq + (uint32_t)r;
                            if (sl < (uint32_t)BH_CAP) subs[(size_t)q * BH_CAP + sl] = ((uint64_t)f32_key(-acc[rb][qb][r]) << 32) | id;
                            need |= sl == (uint32_t)BH_HW;
                        }
                    }
                if (__ballot(need)) want_prune = true;
            }
        };
        do {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of units it and it + 1
#ifndef ZVDB_ABL_NOBARRIER  // (timing-only ablation of round 6, WRONG results: the ceiling of any scheme that splits or relaxes this barrier -- profiles/r6/bf16_stream_ab.txt)
        __builtin_amdgcn_s_barrier();                      // ... everyone's; and everyone is done reading unit it - 1
#endif
        const uint32_t un = min(u0 + it + 2, u1 - 1);
        const gbytes_t nsrc = uniform_ptr(Vf + (size_t)un * F * 1024 + (size_t)wave * 1024);
        const uint32_t ndst = lds0 + (slot >= 1 ? slot - 1 : 2) * SLOT + (uint32_t)wave * 1024;
        const gbytes_t nnorm = uniform_ptr(reinterpret_cast<const unsigned char *>(vnorm + (size_t)un * NG * 32));
        const uint32_t base = lds0 + slot * SLOT + lane16;
        const uint32_t nslot = slot == 2 ? 0 : slot + 1;
        const uint32_t xbase = lds0 + nslot * SLOT + lane16;               // unit it + 1 (past the stream end: stale bytes, never used)
        const uint32_t nbase = lds0 + slot * SLOT + F * 1024 + 16 * fq;
        const uint32_t xnbase = lds0 + nslot * SLOT + F * 1024 + 16 * fq;
        __builtin_amdgcn_sched_barrier(0);
        static_for<F>([&](auto fcn) {
            constexpr int FI = decltype(fcn)::value;
            constexpr int WI = FI % KS, RB = WI / K2, JJ = WI % K2;
            constexpr int SL = FI % R;
            if constexpr (FI == 0) bh_lgkm_wait2<R - 1>(A[SL], nv0);              // (requested ahead of the ring)
            else if constexpr (FI == K2) bh_lgkm_wait3<R - 1>(A[SL], acc[1][0], acc[1][1]);  // (requested at fragment 6)
            else bh_lgkm_wait<R - 1>(A[SL]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FI == 0) {
                acc[0][0] = nv0;
                acc[0][1] = nv0;
            }
            acc[RB][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[SL]), __builtin_bit_cast(bf16x8, B[JJ]), acc[RB][0], 0, 0, 0);
            acc[RB][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[SL]), __builtin_bit_cast(bf16x8, B[K2 + JJ]), acc[RB][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FI == 6) bh_lds_read_norms<64, 64>(acc[1][0], acc[1][1], nbase);
            if constexpr (FI == F - R - 1) asm volatile("ds_read_b128 %0, %1" : "=&v"(nv0) : "v"(xnbase));  // (dead since fragment 0)
            if constexpr (FI + R < F) bh_lds_read128<(FI + R) * 1024>(A[SL], base);
            else bh_lds_read128<(FI + R - F) * 1024>(A[SL], xbase);
            // (placement of the pieces re-swept in this loop: steps of 4 from 0 / 4 from 2 / 6 from 1: -2.8 / -20 / -2.2 %; steps of 5 from
            // fragment 0 / 1 / 3 / 4 instead of 2: +0.8 / -4.3 / -0.5 / -1.3 % -- but from fragment 0 the four query-tile workgroups of a
            // stream share their fetches less: 2 x FETCH_SIZE 17.9 - 18.9 GB per scan instead of 15.5 - 16.0.  Kept at 2.)
            constexpr int DSTEP = 5, NORM_AT = 30;
            if constexpr (FI % DSTEP == 2 && FI / DSTEP < P) {
                constexpr uint32_t POFF = (uint32_t)BH_NW * (FI / DSTEP) * 1024;  // piece = wave + 8 (FI / DSTEP)
                bh_dma16(lane16, nsrc + POFF, ndst + POFF);
            }
            if constexpr (FI == NORM_AT) {
                if (wave == 0) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) bh_dma4(lane4, nnorm + p * 256, ndst + F * 1024 + p * 256);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FI == K2 + 3) select_rb(std::integral_constant<int, 0>{}, 32u * (u0 + it));       // (at + 8: -5 %)
            if constexpr (FI == 3) select_rb(std::integral_constant<int, 1>{}, 32u * (u0 + it) - 32u);  // the previous unit's
        });
        slot = nslot;
        ++it;
        } while (it < n && !want_prune);
        select_rb(std::integral_constant<int, 1>{}, 32u * (u0 + it) - 32u);  // the last unit's row block 1
        bh_lgkm_drain(A);
        asm volatile("" : "+v"(nv0));
        if (want_prune) bh_prune_wave<KPV>(kept, tau_g, kp, lds_state);
    }
    } else {
    while (it < n) {
#pragma unroll
        for (int j = 0; j < KS; ++j) B[j] = *reinterpret_cast<const u32x4 *>(qsrc + (size_t)j * 1024);
        if constexpr (!MOM) {
            thr0 = -tau_f[wave * 32 + fc];
            thr1 = -tau_f[wave * 32 + 16 + fc];
        }
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0), as an instruction the waitcnt pass sees
        bool want_prune = false;
        do {
        // unit `it` has landed: my pieces of it (all but the P (+ NP) youngest operations: those are unit it + 1's), then everyone's
        // (with the drain below this wait is satisfied on arrival; taking it out measured -1.2 %: it moves hipcc's schedule)
        if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P + NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
        __builtin_amdgcn_s_barrier();  // (the first one of a launch is redundant: the __syncthreads above)
        // ... and my pieces of unit it + 1 as well.  Not needed for correctness; it keeps the QT workgroups of a stream within one
        // unit of each other, so that the L2 serves three of them from the fetch of the first: +2 % QPS (A/B on one box,
        // profiles/r3/bf16_stream_ab.txt; round 2's build had this wait by accident -- hipcc placed it for a spill re-load)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the unit fetched during this one: it + 2 (past the stream end: a harmless re-read of its last unit)
        const uint32_t un = min(u0 + it + 2, u1 - 1);
        const gbytes_t nsrc = uniform_ptr(Vf + (size_t)un * F * 1024 + (size_t)wave * 1024);
        const uint32_t ndst = lds0 + (slot >= 1 ? slot - 1 : 2) * SLOT + (uint32_t)wave * 1024;
        const gbytes_t nnorm = uniform_ptr(reinterpret_cast<const unsigned char *>(vnorm + (size_t)un * NG * 32));
        const uint32_t base = lds0 + slot * SLOT + lane16;
        const uint32_t nbase = lds0 + slot * SLOT + F * 1024 + 16 * fq;
        // norms of the unit's rows: group gi, row block rb -> rows 16 rb + 4 fq + [0, 4) of the group
        f32x4 nv[NG][2];
        static_for<NG>([&](auto gc) {
            constexpr int GI = decltype(gc)::value;
            bh_lds_read_norms<GI * 128, GI * 128 + 64>(nv[GI][0], nv[GI][1], nbase);
        });
        u32x4 A[R];
        static_for<R>([&](auto jc) { constexpr int J = decltype(jc)::value; bh_lds_read128<J * 1024>(A[J], base); });
        static_for<NG>([&](auto gc) {
            constexpr int GI = decltype(gc)::value;
            bh_lgkm_wait2<R>(nv[GI][0], nv[GI][1]);
        });
        f32x4 acc[2][2];  // [row block][B set]
        __builtin_amdgcn_sched_barrier(0);
        static_for<F>([&](auto fcn) {  // fragment f = (group f / KS, row block (f % KS) / K2, k-step f % K2)
            constexpr int FI = decltype(fcn)::value;
            constexpr int GI = FI / KS, WI = FI % KS, RB = WI / K2, JJ = WI % K2;
            constexpr int SL = FI % R;
            constexpr int W = (F - 1 - FI) < (R - 1) ? (F - 1 - FI) : (R - 1);  // younger reads allowed in flight
            bh_lgkm_wait<W>(A[SL]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (JJ == 0) {  // (the mirror holds the start value ready-made: -|v'|^2 / 2)
                acc[RB][0] = nv[GI][RB];
                acc[RB][1] = nv[GI][RB];
            }
            acc[RB][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[SL]), __builtin_bit_cast(bf16x8, B[JJ]), acc[RB][0], 0, 0, 0);
            acc[RB][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[SL]), __builtin_bit_cast(bf16x8, B[K2 + JJ]), acc[RB][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FI + R < F) bh_lds_read128<(FI + R) * 1024>(A[SL], base);
            // this wave's pieces of the unit being fetched: one every DSTEP fragments from fragment 2 on, the norms behind the last.
            // 48-fragment units issue theirs in the first 60 % of the loop, so that the wait behind the next barrier finds them
            // landed: steps of 8 / 7 / 6 / 5 / 4 in the final loop structure: 95.0k / 94.4k / 97.9k / 98.9k / 98.9k QPS on one box, with
            // 2 x FETCH_SIZE = 15.4 / 48.2 / 19.9 / 16.8 / 18.1 GB per scan for 15.4 GB of corpus (profiles/r3/bf16_stream_ab.txt): how well
            // the 4 query-tile workgroups of a stream share their fetches depends on the placement too.  A placement also moves
            // hipcc's register allocation (the loop has no VGPR to spare) -- re-measure after any change here
            constexpr int DSTEP = F == 48 ? 5 : 8, NORM_AT = F == 48 ? 30 : F - 3;
            if constexpr (FI % DSTEP == 2 && FI / DSTEP < P) {
                constexpr uint32_t POFF = (uint32_t)BH_NW * (FI / DSTEP) * 1024;  // piece = wave + 8 (FI / DSTEP)
                bh_dma16(lane16, nsrc + POFF, ndst + POFF);
            }
            if constexpr (FI == NORM_AT) {
                if (wave == 0) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) bh_dma4(lane4, nnorm + p * 256, ndst + F * 1024 + p * 256);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (WI == KS - 1) {  // group GI complete
                const uint32_t row0 = 32u * ((u0 + it) * NG + GI);
                if constexpr (MOM) {  // acc = -key; rows past N sit at -inf and are left out
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float a = acc[rb][qb][r] > -3.0e38f ? acc[rb][qb][r] : 0.0f;
                                ms1[qb] += a;
                                ms2[qb] += a * a;
                            }
                    mrows += row0 < N ? min(32u, N - row0) : 0u;
                } else {
                    const float m0 = fmaxf(fmaxf(fmaxf(acc[0][0][0], acc[0][0][1]), fmaxf(acc[0][0][2], acc[0][0][3])),
                                           fmaxf(fmaxf(acc[1][0][0], acc[1][0][1]), fmaxf(acc[1][0][2], acc[1][0][3])));
                    const float m1 = fmaxf(fmaxf(fmaxf(acc[0][1][0], acc[0][1][1]), fmaxf(acc[0][1][2], acc[0][1][3])),
                                           fmaxf(fmaxf(acc[1][1][0], acc[1][1][1]), fmaxf(acc[1][1][2], acc[1][1][3])));
                    if (__ballot(m0 > thr0 || m1 > thr1)) {
                        if constexpr (NG > 1) bh_lgkm_drain(A);  // (mid-unit epilogue: the ring may hold reads still in flight)
                        bool need = false;
#pragma unroll
                        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    if (acc[rb][qb][r] > (qb ? thr1 : thr0)) {
                                        uint32_t q = (uint32_t)wave * 32 + (uint32_t)qb * 16 + fc;
                                        const uint32_t sl = atomicAdd(cnt + q, 1u);
                                        const uint32_t id = row0 + 16u * rb + 4u * fq + (uint32_t)r;
                                        if (sl < (uint32_t)BH_CAP) subs[(size_t)q * BH_CAP + sl] = ((uint64_t)f32_key(-acc[rb][qb][r]) << 32) | id;
                                        need |= sl == (uint32_t)BH_HW;
                                    }
                                }
                        if (__ballot(need)) want_prune = true;  // (<= 32 NG more appends per query before the loop is left: BH_CAP holds them)
                    }
                }
            }
        });
        slot = slot == 2 ? 0 : slot + 1;
        ++it;
        } while (it < n && !want_prune);
        if constexpr (!MOM) {
            if (want_prune) bh_prune_wave<KPV>(kept, tau_g, kp, lds_state);
        }
    }

    }

    if constexpr (MOM) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float a1 = ms1[qb], a2 = ms2[qb];
            a1 += __shfl_xor(a1, 16); a1 += __shfl_xor(a1, 32);
            a2 += __shfl_xor(a2, 16); a2 += __shfl_xor(a2, 32);
            if (lane < 16) {
                float *o = mom + ((size_t)b * T2 + wave * 32 + qb * 16 + lane) * 2;
                o[0] = a1;
                o[1] = a2;
            }
        }
        if (tid == 0) mom[(size_t)gridDim.x * T2 * 2 + b] = (float)mrows;
        return;
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every wave is done with the ring: it becomes the per-wave select scratch of the stream end
    if (dur_out && tid == 0)  // (host-mapped: read by a later launch; 100 MHz ticks under the launch's tag)
        dur_out[b] = (split.tag << 24) | (uint32_t)min((unsigned long long)0xFFFFFFull, __builtin_amdgcn_s_memrealtime() - t_begin);
    {   // stream end: lane-per-query copy when nothing needs selecting, else the radix select
        unsigned char *selw = smem + wave * sel_scratch_bytes<KPV>();
        const uint32_t q = (uint32_t)wave * 32 + (uint32_t)(lane & 31);
        const uint32_t qg = qt * T2 + q;
        const bool mine = lane < 32 && qg < nq;
        const uint32_t c = mine ? min(lds_load_u32(cnt + q), (uint32_t)BH_CAP) : 0u;
        const uint32_t nk = mine ? nkept[q] : 0u;
        const bool simple = mine && nk + c <= (uint32_t)KPV;
        if (simple) {
            uint64_t *dst = partial + ((size_t)qg * Stot + list_base + s) * KPV;
            for (uint32_t e = 0; e < nk; ++e) dst[e] = kept[(size_t)q * KPV + e];
            for (uint32_t e = 0; e < c; ++e) dst[nk + e] = subs[(size_t)q * BH_CAP + e];
            // (entries past the count are not written: every reader of a partial list stops at pcnt; the fill was 62 scattered 8-byte stores per lane, ~20 us per launch)
            pcnt[(size_t)qg * Stot + list_base + s] = nk + c;
        }
        unsigned long long todo = __ballot(mine && !simple);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t q2 = (uint32_t)wave * 32 + (uint32_t)l;
            const uint32_t qg2 = qt * T2 + q2;
            const uint32_t c2 = min(lds_load_u32(cnt + q2), (uint32_t)BH_CAP);
            if (lane < NSUB) subw[wave * NSUB + lane] = min(c2 - min(c2, (uint32_t)(lane * SUBCAP)), (uint32_t)SUBCAP);
            prune_query<false, true, KPV>(kept, subs, tau_f, nkept, subw + wave * NSUB, q2, lane, kp, nullptr, selw);
#pragma unroll
            for (int ci = 0; ci < KPV / 64; ++ci)
                partial[((size_t)qg2 * Stot + list_base + s) * KPV + lane + 64 * ci] = kept[(size_t)q2 * KPV + lane + 64 * ci];
            if (lane == 0) pcnt[(size_t)qg2 * Stot + list_base + s] = nkept[q2];
        }
    }
}

// moments -> guessed threshold tau = -(mean acc) + z sigma per query (see k_fp4_guess); mom = [block][256][2] + [block] rows
__global__ __launch_bounds__(64) void k_bf16_guess(const float *__restrict__ mom, uint32_t nblocks, uint32_t QT, uint32_t S, uint32_t nq,
                                                   float z, uint32_t *__restrict__ tau_g, uint32_t *__restrict__ verify_flag) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && verify_flag) { verify_flag[0] = 0; verify_flag[1] = 0; }  // ([1]: deficient queries of the batch)
    const int lane = threadIdx.x;
    const uint32_t qg = blockIdx.x, qt = qg / T2, q = qg % T2;
    double s1 = 0.0, s2 = 0.0, n = 0.0;
    for (uint32_t st = lane; st < S; st += 64) {
        const uint32_t b = ((st >> 3) * QT + qt) * 8 + (st & 7);  // inverse of the scan's block -> (stream, tile) map
        s1 += mom[((size_t)b * T2 + q) * 2 + 0];
        s2 += mom[((size_t)b * T2 + q) * 2 + 1];
        n += mom[(size_t)nblocks * T2 * 2 + b];
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

__device__ __forceinline__ uint32_t bh_bf16_bits(float v) {  // RNE, finite inputs (hnsw.zig:243 defines inputs finite)
    uint32_t u = __float_as_uint(v);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

// rows (fp32, row-major) -> Layout16 (header).  Rows >= n and elements >= dim are 0.
template <typename RT>
__global__ __launch_bounds__(256) void k_bf16_pack_frag(const RT *__restrict__ src, uint32_t dim, uint32_t ld, uint64_t n, uint64_t units,
                                                        uint32_t K2, unsigned char *__restrict__ out) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const uint32_t l = (uint32_t)(u & 63);
    const uint64_t bj = u >> 6;
    const uint32_t j = (uint32_t)(bj % K2);
    const uint64_t row = (bj / K2) * 16 + (l & 15);
    const uint32_t k0 = 32 * j + 8 * (l >> 4);
    u32x4 w = {0, 0, 0, 0};
    if (row < n) {
        const RT *v = src + row * ld;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t k = k0 + e;
            if (k < dim) w[e >> 1] |= bh_bf16_bits(row_elem(v, k)) << (16 * (e & 1));
        }
    }
    *reinterpret_cast<u32x4 *>(out + u * 16) = w;
}

// norm[r] = -1/2 sum over the bf16-rounded row of v'^2 (L2; -0 for dot / cosine), -inf for r >= n: one wave per row.  This IS the
// accumulator's start value (header: keys), stored ready-made (the halving is exact): the scan copies it, no multiply (+1 % QPS)
template <typename RT>
__global__ __launch_bounds__(256) void k_bf16_frag_norms(const RT *__restrict__ rows, uint32_t dim, uint32_t ld, uint64_t n, uint64_t n_pad,
                                                         int l2, float *__restrict__ norm) {
    const int lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_pad) return;
    if (r >= n) {
        if (lane == 0) norm[r] = -__builtin_inff();
        return;
    }
    float sacc = 0.0f;
    if (l2) {
        const RT *v = rows + r * ld;
        for (uint32_t c = lane; c < dim; c += 64) {
            const float f = __uint_as_float(bh_bf16_bits(row_elem(v, c)) << 16);
            sacc += f * f;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off);
    }
    if (lane == 0) norm[r] = -0.5f * sacc;
}

template <int KS, int NG, bool MOM, int KPV>
int bh_launch_geo(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint64_t *cand,
                  uint64_t *partial, uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot, uint32_t *pcnt, uint32_t max_units,
                  const uint32_t *gate, float *mom, hipStream_t s, const BhSplit &split, uint32_t *dur_out) {
    constexpr uint32_t lds = bh_lds_bytes<KS, NG, KPV>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    const uint32_t G = (uint32_t)((h->count + 31) / 32), U = (G + NG - 1) / NG;
    auto go = [&](auto kern) -> int {
        ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(QT * S), dim3(512), lds, s, q_op, h->rows_hf, h->norm_hf, (uint32_t)h->count, nq, QT, S, U, cand, partial,
                           tau_g, kp, list_base, Stot, pcnt, max_units, gate, mom, split, dur_out);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    };
    if constexpr (!MOM && KPV == KP) {
        if (gate) return go(k_bf16_stream<KS, NG, false, KP, 1>);
    }
    return go(k_bf16_stream<KS, NG, MOM, KPV, 0>);
}

template <bool MOM, int KPV>
int bh_launch(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint64_t *cand, uint64_t *partial,
              uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot, uint32_t *pcnt, uint32_t sample_groups, const uint32_t *gate,
              float *mom, hipStream_t s, const BhSplit &split = BhSplit{}, uint32_t *dur_out = nullptr) {
    const uint32_t KS = bf16s_ksteps(h->dim);
    auto mu = [&](uint32_t ng) { return sample_groups ? (sample_groups + ng - 1) / ng : 0u; };
    if (KS == 8) return bh_launch_geo<8, 6, MOM, KPV>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, mu(6), gate, mom, s, split, dur_out);
    if (KS == 16) return bh_launch_geo<16, 3, MOM, KPV>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, mu(3), gate, mom, s, split, dur_out);
    if (KS == 32) return bh_launch_geo<32, 1, MOM, KPV>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, mu(1), gate, mom, s, split, dur_out);
    return bh_launch_geo<48, 1, MOM, KPV>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, mu(1), gate, mom, s, split, dur_out);
}

}  // namespace

// 16-element k-steps per row; instantiated: 8, 16, 32, 48 (dim <= 128 / 256 / 512 / 768)
uint32_t bf16s_ksteps(uint32_t dim) { return dim <= 128 ? 8u : dim <= 256 ? 16u : dim <= 512 ? 32u : 48u; }
static uint32_t bf16s_groups_per_unit(uint32_t KS) { return KS == 8 ? 6u : KS == 16 ? 3u : 1u; }

bool bf16s_eligible(const zvdb_hip_index *h) {
    const bool off = [] { const char *e = tuning("ZVDB_BF_BF16_KERNEL"); return e && e[0] == 'v' && e[1] == '6'; }();
    return !off && h->dim > 4 && h->dim <= 768;
}

uint64_t bf16s_query_bytes(const zvdb_hip_index *h, uint32_t nq_pad) { return (uint64_t)nq_pad * bf16s_ksteps(h->dim) * 32; }
uint64_t bf16s_moment_bytes(uint32_t QT, uint32_t S) { return ((uint64_t)QT * S * T2 * 2 + (uint64_t)QT * S) * sizeof(float); }

int ensure_bf16_frag_mirror(zvdb_hip_index *h, hipStream_t s) {
    if (h->rows_hf && h->count_hf == h->count) return ZVDB_OK;
    if (h->count == 0) return ZVDB_OK;
    const uint32_t KS = bf16s_ksteps(h->dim);
    // whole units (NG groups) and whole 256-row tiles: 1536 = lcm(192, 256)
    const uint64_t rows_pad = round_up64(h->count, 1536);
    if (h->cap_hf < rows_pad) {
        if (h->rows_hf) { ZV_HIP(hipDeviceSynchronize()); /* searches of this index may be in flight on OTHER streams: the mutex covers the host side only */ ZV_HIP(hipFree(h->rows_hf)); ZV_HIP(hipFree(h->norm_hf)); }
        h->rows_hf = nullptr;
        h->norm_hf = nullptr;
        const uint64_t ncap = round_up64(std::max<uint64_t>(h->count, h->cap), 1536);
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->rows_hf), ncap * KS * 32));
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->norm_hf), (ncap + 64) * sizeof(float)));  // (+64: the last norm piece is a whole 256 B)
        h->cap_hf = ncap;
    }
    const uint64_t units = rows_pad / 16 * (KS / 2) * 64;
    ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {  // (bf16 rows: the mirror is a lossless re-ordering)
        using RT = row_elem_t<decltype(rows)>;
        hipLaunchKernelGGL(k_bf16_pack_frag<RT>, dim3((uint32_t)((units + 255) / 256)), dim3(256), 0, s, rows, h->dim, ld, h->count, units, KS / 2,
                           h->rows_hf);
        hipLaunchKernelGGL(k_bf16_frag_norms<RT>, dim3((uint32_t)((rows_pad + 64 + 3) / 4)), dim3(256), 0, s, rows, h->dim, ld, h->count,
                           rows_pad + 64, h->metric == ZVDB_METRIC_L2SQ ? 1 : 0, h->norm_hf);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    }));
    h->count_hf = h->count;
    (void)bf16s_groups_per_unit;
    return ZVDB_OK;
}

int launch_bf16s_prep_queries(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t nq_pad, unsigned char *q_op, hipStream_t s) {
    const uint32_t KS = bf16s_ksteps(h->dim);
    const uint64_t units = (uint64_t)nq_pad / 16 * (KS / 2) * 64;
    hipLaunchKernelGGL(k_bf16_pack_frag<float>, dim3((uint32_t)((units + 255) / 256)), dim3(256), 0, s, d_queries, h->dim, h->dim, (uint64_t)nq, units,
                       KS / 2, q_op);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// Adaptive split of the scan's units over the 8 XCDs.  Under the power limit the XCDs of one device run this kernel at clocks a
// few per cent apart (profiles/r3/bf16_stream_workgroup_end_times.txt: the same XCDs finish 3 - 6 % early launch after launch),
// and a launch ends with its slowest workgroup.  Every workgroup leaves its duration in host-mapped memory; here the durations of
// an EARLIER launch (whatever has arrived: nothing is waited for) turn into per-XCD speeds = units / mean duration, the shares
// follow them with damping 1/2, and the next launch runs with the new split.  Results do not depend on the split (every row is
// scanned once; the merged lists hold the exact k' best keys either way).  ZVDB_BF16S_BALANCE=0: even split.
static int bh_balance(const zvdb_hip_index *h, uint32_t QT, uint32_t S, BhSplit *split, uint32_t **dur_out) {
    const bool off = [] { const char *e = tuning("ZVDB_BF16S_BALANCE"); return e && e[0] == '0'; }();
    const bool dbg = [] { const char *e = tuning("ZVDB_BF16S_BALANCE"); return e && e[0] == 'v'; }();
    const uint32_t nb = QT * S, G = (uint32_t)((h->count + 31) / 32);
    const uint32_t NG = bf16s_groups_per_unit(bf16s_ksteps(h->dim)), U = (G + NG - 1) / NG;
    if (off || S % 8 != 0 || nb > 4096 || U < 64 * S) return ZVDB_OK;  // (short streams: nothing to balance)
    if (!h->bal_host) {
        ZV_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->bal_host), 4096 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
        if (!h->bal_host) { set_error("hipHostMalloc returned no block"); return ZVDB_ERR_OOM; }
        std::memset(h->bal_host, 0, 4096 * sizeof(uint32_t));
        ZV_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&h->bal_dev), h->bal_host, 0));
    }
    if (h->bal_nb == nb) {
        // durations of ONE earlier launch (all under the same tag: launches are queued ahead of the device, the newest complete set
        // may be several launches old) against the split THAT launch ran with
        const uint32_t tag = __atomic_load_n(h->bal_host, __ATOMIC_RELAXED) >> 24;
        const uint32_t *ho = h->bal_hist[tag & 255];
        double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool all = ho[8] == U;
        for (uint32_t b = 0; b < nb && all; ++b) {
            const uint32_t d = __atomic_load_n(h->bal_host + b, __ATOMIC_RELAXED);
            if ((d >> 24) != tag || (d & 0xFFFFFFu) == 0) all = false;
            t[b & 7] += (double)(d & 0xFFFFFFu);
        }
        if (all) {
            double speed[8], sum = 0.0;
            for (int x = 0; x < 8; ++x) { speed[x] = (double)(ho[x + 1] - ho[x]) / t[x]; sum += speed[x]; }
            double wsum = 0.0;
            for (int x = 0; x < 8; ++x) { h->bal_w[x] = 0.5f * h->bal_w[x] + 0.5f * (float)(speed[x] / sum * 8.0); wsum += h->bal_w[x]; }  // (gain 0.5 / 0.75 / 1: no difference)
            for (int x = 0; x < 8; ++x) h->bal_w[x] = std::min(1.15f, std::max(0.85f, (float)(h->bal_w[x] / wsum * 8.0)));
            if (dbg) {
                double dmax = 0.0, dsum = 0.0, xmax[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xmin[8] = {1e30, 1e30, 1e30, 1e30, 1e30, 1e30, 1e30, 1e30};
                for (uint32_t b = 0; b < nb; ++b) {
                    const double d = (double)(__atomic_load_n(h->bal_host + b, __ATOMIC_RELAXED) & 0xFFFFFFu);
                    dmax = std::max(dmax, d); dsum += d; xmax[b & 7] = std::max(xmax[b & 7], d); xmin[b & 7] = std::min(xmin[b & 7], d);
                }
                fprintf(stderr, "[zvdb bf16s balance] tag %u slowest workgroup / mean = %.4f; per XCD max/mean-of-all:", tag, dmax / (dsum / nb));
                for (int x = 0; x < 8; ++x) fprintf(stderr, " %.3f(min %.3f)", xmax[x] / (dsum / nb), xmin[x] / (dsum / nb));
                fprintf(stderr, "\n");
            }
            if (dbg) fprintf(stderr, "[zvdb bf16s balance] tag %u shares %.3f %.3f %.3f %.3f %.3f %.3f %.3f %.3f\n", tag, h->bal_w[0], h->bal_w[1], h->bal_w[2],
                             h->bal_w[3], h->bal_w[4], h->bal_w[5], h->bal_w[6], h->bal_w[7]);
        }
    } else {
        for (int x = 0; x < 8; ++x) h->bal_w[x] = 1.0f;
        std::memset(h->bal_hist, 0, sizeof(h->bal_hist));
    }
    h->bal_seq = (h->bal_seq + 1) & 255;  // (a caller more than 254 scans ahead of the device would pair durations with a later
    if (h->bal_seq == 0) h->bal_seq = 1;  //  launch's split: the shares are clamped to +-15 % and damped, the results unaffected; tag 0 = "nothing written yet")
    uint32_t *off8 = h->bal_hist[h->bal_seq];
    double wsum = 0.0, acc = 0.0;
    for (int x = 0; x < 8; ++x) wsum += h->bal_w[x];
    off8[0] = 0;
    for (int x = 0; x < 8; ++x) {
        acc += h->bal_w[x];
        off8[x + 1] = x == 7 ? U : std::max(off8[x], (uint32_t)std::llround((double)U * acc / wsum));
    }
    h->bal_nb = nb;
    for (int x = 0; x < 9; ++x) split->off[x] = off8[x];
    split->tag = h->bal_seq;
    *dur_out = h->bal_dev;
    return ZVDB_OK;
}

// QT = tiles of 256 queries; S = streams (QT * S workgroups, one per CU)
int launch_bf16s_scan(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint64_t *cand,
                      uint64_t *partial, uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot, uint32_t *pcnt, const uint32_t *gate,
                      uint32_t kpv, hipStream_t s) {
    // the main scan of the guessed-threshold pipeline runs with the adaptive per-XCD split (bh_balance); the gated repair scan with
    // an even one
    BhSplit split{};
    uint32_t *dur = nullptr;
    if (!gate) ZV_CHECK(bh_balance(h, QT, S, &split, &dur));
    if (kpv == (uint32_t)KPMAX) return bh_launch<false, KPMAX>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, 0u, gate, nullptr, s, split, dur);
    if (kpv == (uint32_t)KPW) return bh_launch<false, KPW>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, 0u, gate, nullptr, s, split, dur);
    return bh_launch<false, KP>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, 0u, gate, nullptr, s, split, dur);
}

int launch_bf16s_guess(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, float *mom, float z,
                       uint32_t *tau_g, uint32_t *verify_flag, hipStream_t s) {
    ZV_CHECK((bh_launch<true, KP>(h, q_op, nq, QT, S, nullptr, nullptr, nullptr, 0u, 0u, 0u, nullptr, BH_SAMPLE_GROUPS, nullptr, mom, s)));
    hipLaunchKernelGGL(k_bf16_guess, dim3(QT * T2), dim3(64), 0, s, mom, QT * S, QT, S, nq, z, tau_g, verify_flag);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

}  // namespace zvdb
