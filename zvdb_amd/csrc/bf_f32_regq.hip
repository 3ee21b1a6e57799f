// bf_f32_regq.hip -- the EXACT-ARITHMETIC fp32 MFMA screen of the brute-force route (BASELINE configs[1]: 1M x 128
// fp32, batch 256) for short rows: REGISTER-RESIDENT QUERIES, the corpus streamed HBM -> registers, no LDS staging and no barrier.
// No reference symbol (hnsw.zig has no brute force); results are pinned by the oracle's exact top-k through the certificate
// (bf_cert.h: its rounding bound holds for ANY accumulation order) and the exact fp32 re-rank (hnsw.zig:182-192 arithmetic).
//
// Why a third fp32 kernel.  At 128 dimensions the staged 256 x 256 tile (k_bf_scan5<0>: both operands through LDS, a barrier per
// slice) and the 128-query stream kernel (k_i8_stream<0>: queries resident in LDS, two workgroups per stream) both sit at 0.65 -
// 0.68 of the fp32-MFMA peak on configs[1] (profiles/r5/c2_1Mx128_f32_passes.txt).  v_mfma_f32_32x32x2_f32 is a 64-cycle
// instruction that takes ONE VGPR per operand: a wave's 32 queries x 128 dimensions are 64 VGPRs -- they stay in registers for the
// whole launch; 8 waves = 256 queries per workgroup, one workgroup per CU, so a batch of 256 reads the corpus ONCE (the 128-query
// kernel: twice, through the L2).  Every wave reads every row: a 32-row group is KS contiguous KiB of the fragment-order mirror
// (rows_ff, shared with k_i8_stream<0>), one coalesced global_load_dwordx4 per 4 MFMAs (256 matrix-pipe cycles), prefetched a whole
// group ahead (8 - 14 pieces beyond 160 dimensions) in a register ring that lives, with the accumulator, ABOVE the compiler's allocation limit; the 8 waves of a workgroup ask for the same lines (the first miss fills the L1 / L2, the rest
// hit).  No wave reads what another wave wrote: NO barrier, NO LDS in the steady state -- SIMD partners drift apart and fill each
// other's selection epilogues.  Work is split by 32-row GROUPS (31,250 at 1M rows: 122.07 per CU, a 0.8 % tail) instead of 256-row
// tiles (15.26 per CU: 5 %).
//
// Layout (k_f32_pack_frag, bf_i8_stream.hip): piece (group g, k-step j, lane l) = 16 bytes = elements 8 j + 4 (l >> 5) + [0, 4) of
// row / query 32 g + (l & 31); a group holds its KS pieces contiguously (KS KiB).  Component c of a piece feeds one 32x32x2 MFMA.
// Keys: acc starts at -|v|^2 / 2 (0 for dot / cosine; -inf for rows past N) and accumulates q.v: acc = -key, pass <=> acc > -tau.
// Accumulator layout (32x32 f32): lane l holds query (column) l & 31, rows 8 (r >> 2) + 4 (l >> 5) + (r & 3) for r = 0 .. 15.
// Selection state (per-(workgroup, query) kept list + append buffer in global memory, exact (key, id) radix select) is shared with
// the other scan kernels (bf_select.h); a wave owns its 32 queries: appends and prune rounds involve that wave only.
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "wave_select.h"
#include "bf_select.h"
#include "stream_common.h"
#include "row_types.h"

namespace zvdb {

namespace {

constexpr int RQ_NW = 8;                   // waves per workgroup (2 per SIMD)
constexpr int RQ_CAP = NSUB * SUBCAP;      // append-buffer entries per (workgroup, query)
constexpr int RQ_HW = 384, RQ_MARK = 64;   // prune request mark; queries above MARK are pruned in a round

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int KPV>
__host__ __device__ constexpr uint32_t rq_lds_bytes() {
    return T2 * 4 * 3 + RQ_NW * NSUB * 4 + 16 + RQ_NW * sel_scratch_bytes<KPV>();
}

// An append's store by inline asm: hipcc's waitcnt pass does not see it, so the counted waits of the operand ring stay exact behind the rare
// pass path (as a visible store it made the pieces behind every selection wait for more of the ring than they need: -3 % at configs[1]).
// In hardware the store is one more (younger) entry of the in-order vmcnt queue: a ring wait can only wait LONGER for it, never shorter;
// its readers (rq_prune_wave, the stream end) sit behind an s_waitcnt vmcnt(0) with a memory clobber.
__device__ __forceinline__ void rq_append(uint64_t *p, uint64_t v) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// ---- the operand ring: registers hipcc cannot allocate, named literally ------------------------------------------------------------
// A kernel of this file is built with amdgpu_num_vgpr: hipcc's allocator ends at v[RA - 1] (RA = rq_acc_base = 236 - 4 R).  Above live
// the group's 32 x 32 accumulator v[RA, RA + 16), the ring (RB = RA + 16: slot SL = v[RB + 4 SL, RB + 4 SL + 4)) and the next group's norms
// (v[252]).  Nothing the compiler does to registers -- copies, rotations, spills -- can touch a load in flight or an MFMA result still
// in the pipe there: the loads, their counted waits, the MFMAs and the first reads of their results are all inside asm statements
// that name the registers.  (The accumulator is up there for the second reason: hipcc pads no wait states behind an MFMA inside a
// string, and it did move the accumulator -- sixteen v_mov between two pieces' statements -- when it owned it.)  (One clobber of v255 makes the kernel descriptor allocate the whole file: 2 waves per SIMD
// either way.)  tools/check_ring_regs.py audits every instantiation's ISA as a CPU test: no register >= RA outside ;;#ASMSTART / ;;#ASMEND,
// no compiler wait on vmcnt inside the group loop, no spill.
// Counting rule: a wait names only the ring's OWN younger loads (vmcnt(R - 1) ahead of the MFMAs of a slot: the R - 1 ring loads issued
// since that slot's); any other memory operation in flight -- the norm load, an append's store -- is one more entry of the in-order
// queue and can only make the wait longer.
__host__ __device__ constexpr int rq_unit_groups(int KS) { return KS == 8 ? 2 : 1; }   // groups per loop iteration (64 dimensions: two, the ring holds both)
__host__ __device__ constexpr int rq_ring_len(int KS) { return KS == 8 ? 16 : KS <= 20 ? KS : KS == 24 ? 12 : KS == 28 ? 14 : 8; }
__host__ __device__ constexpr int rq_ring_base(int KS) { return 252 - 4 * rq_ring_len(KS); }
__host__ __device__ constexpr int rq_acc_base(int KS) { return rq_ring_base(KS) - 16; }
constexpr int RQ_NORM_REG = 252;

// prologue: slot at v[RG, RG + 4) <- 16 bytes per lane at base + off + IMM (the base may be fresh from SALU arithmetic: 5 states ahead of the load)
template <int RG, int IMM>
__device__ __forceinline__ void rq_ring_fill(gbytes_t base, uint32_t off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 v[%1:%1+3], %0, %2 offset:%3" ::"v"(off), "n"(RG), "s"(base), "n"(IMM));
}

// one piece: wait for the slot at v[RG, RG + 4) (WAIT = younger ring loads), its 4 MFMAs into v[RA, RA + 16), refill the slot from
// base + off + IMM.  The group's first piece takes the start values as its C operand (no copy into the accumulator); they may be fresh
// from VALU moves: 2 states ahead of the MFMA that reads them.
#ifdef ZVDB_ABL_RQ_NOLOADS   // (timing-only ablation: the pieces without their refills -- what the MFMA stream alone achieves: profiles/r6/c2_ring_ab.txt)
#define RQ_REFILL6 "s_nop 0"
#define RQ_REFILL5 "s_nop 0"
#else
#define RQ_REFILL6 "global_load_dwordx4 v[%6:%6+3], %5, %8 offset:%9"
#define RQ_REFILL5 "global_load_dwordx4 v[%5:%5+3], %4, %7 offset:%8"
#endif
template <int RA, int RG, int WAIT, int IMM>
__device__ __forceinline__ void rq_piece_first(const f32x16 &init, const u32x4 &b, gbytes_t base, uint32_t off) {
    asm volatile("s_nop 1\n\t"
                 "s_waitcnt vmcnt(%7)\n\t"
                 "v_mfma_f32_32x32x2_f32 v[%10:%10+15], v[%6+0], %1, %0\n\t"
                 "v_mfma_f32_32x32x2_f32 v[%10:%10+15], v[%6+1], %2, v[%10:%10+15]\n\t"
                 "v_mfma_f32_32x32x2_f32 v[%10:%10+15], v[%6+2], %3, v[%10:%10+15]\n\t"
                 "v_mfma_f32_32x32x2_f32 v[%10:%10+15], v[%6+3], %4, v[%10:%10+15]\n\t"
                 RQ_REFILL6 ::"v"(init),
                 "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(off), "n"(RG), "n"(WAIT), "s"(base), "n"(IMM), "n"(RA));
}
template <int RA, int RG, int WAIT, int IMM>
__device__ __forceinline__ void rq_piece(const u32x4 &b, gbytes_t base, uint32_t off) {
    asm volatile("s_waitcnt vmcnt(%6)\n\t"
                 "v_mfma_f32_32x32x2_f32 v[%9:%9+15], v[%5+0], %0, v[%9:%9+15]\n\t"
                 "v_mfma_f32_32x32x2_f32 v[%9:%9+15], v[%5+1], %1, v[%9:%9+15]\n\t"
                 "v_mfma_f32_32x32x2_f32 v[%9:%9+15], v[%5+2], %2, v[%9:%9+15]\n\t"
                 "v_mfma_f32_32x32x2_f32 v[%9:%9+15], v[%5+3], %3, v[%9:%9+15]\n\t"
                 RQ_REFILL5 ::"v"(b.x),
                 "v"(b.y), "v"(b.z), "v"(b.w), "v"(off), "n"(RG), "n"(WAIT), "s"(base), "n"(IMM), "n"(RA));
}

// the next group's norms: issued behind piece 0's refill, read behind piece KS - 1's (KS - 1 younger ring loads by then)
__device__ __forceinline__ void rq_norm_load(gbytes_t base, uint32_t off) {
    asm volatile("s_nop 4\n\tglobal_load_dword v[%1], %0, %2" ::"v"(off), "n"(RQ_NORM_REG), "s"(base));
}
template <int WAIT>
__device__ __forceinline__ float rq_norm_take() {
    float v;
    asm volatile("s_waitcnt vmcnt(%2)\n\tv_mov_b32 %0, v[%1]" : "=v"(v) : "n"(RQ_NORM_REG), "n"(WAIT));
    return v;
}

// The finished group's largest accumulator value.  The last MFMA of the group (16 passes) -> the first VALU read of its result: 19
// states, and hipcc pads nothing for an MFMA inside a string: the pad opens this string.
template <int RA>
__device__ __forceinline__ float rq_group_max() {
    float m;
    asm volatile("s_nop 15\n\ts_nop 3\n\t"
                 "v_max3_f32 %0, v[%1+0], v[%1+1], v[%1+2]\n\t"
                 "v_max3_f32 %0, %0, v[%1+3], v[%1+4]\n\t"
                 "v_max3_f32 %0, %0, v[%1+5], v[%1+6]\n\t"
                 "v_max3_f32 %0, %0, v[%1+7], v[%1+8]\n\t"
                 "v_max3_f32 %0, %0, v[%1+9], v[%1+10]\n\t"
                 "v_max3_f32 %0, %0, v[%1+11], v[%1+12]\n\t"
                 "v_max3_f32 %0, %0, v[%1+13], v[%1+14]\n\t"
                 "v_max_f32 %0, %0, v[%1+15]"
                 : "=v"(m)
                 : "n"(RA));
    return m;
}
// the rare pass path: the accumulator into registers the compiler owns (behind rq_group_max: the results have settled)
template <int RA>
__device__ __forceinline__ void rq_acc_read(float (&a)[16]) {
    asm volatile("v_mov_b32 %0, v[%16+0]\n\tv_mov_b32 %1, v[%16+1]\n\tv_mov_b32 %2, v[%16+2]\n\tv_mov_b32 %3, v[%16+3]\n\t"
                 "v_mov_b32 %4, v[%16+4]\n\tv_mov_b32 %5, v[%16+5]\n\tv_mov_b32 %6, v[%16+6]\n\tv_mov_b32 %7, v[%16+7]\n\t"
                 "v_mov_b32 %8, v[%16+8]\n\tv_mov_b32 %9, v[%16+9]\n\tv_mov_b32 %10, v[%16+10]\n\tv_mov_b32 %11, v[%16+11]\n\t"
                 "v_mov_b32 %12, v[%16+12]\n\tv_mov_b32 %13, v[%16+13]\n\tv_mov_b32 %14, v[%16+14]\n\tv_mov_b32 %15, v[%16+15]"
                 : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]), "=v"(a[4]), "=v"(a[5]), "=v"(a[6]), "=v"(a[7]), "=v"(a[8]), "=v"(a[9]), "=v"(a[10]),
                   "=v"(a[11]), "=v"(a[12]), "=v"(a[13]), "=v"(a[14]), "=v"(a[15])
                 : "n"(RA));
}

// every query fragment has landed before the group loop is entered: hipcc would otherwise place ITS waits for them inside the loop, and a
// compiler wait does not know the ring (its vmcnt(0) for the last fragment would drain the ring once per group)
template <int KS>
__device__ __forceinline__ void rq_frags_landed(const u32x4 (&B)[KS]) {  // (inputs only: a read-write operand made hipcc copy all of them)
#pragma unroll
    for (int j = 0; j < KS; j += 4) asm volatile("" ::"v"(B[j]), "v"(B[j + 1]), "v"(B[j + 2]), "v"(B[j + 3]));
}

// Exact prune of the calling wave's 32 queries whose append buffer passed MARK.  Out of line and cold: the scan keeps the query
// fragments and the operand ring live, an inlined radix select would push the allocator into spilling them in the hot loop.
// (Per-wave select scratch: the LDS is otherwise unused, so no lock as in bf_bf16_stream.hip.)
template <int KPV>
__device__ __attribute__((noinline, cold)) void rq_prune_wave(uint64_t *kept, uint32_t *tau_g, uint32_t kp, uint32_t lds_state /* LDS address of tau_f */) {
    float *tau_f = (float *)(__attribute__((address_space(3))) float *)(size_t)lds_state;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(tau_f + T2);
    uint32_t *nkept = cnt + T2;
    uint32_t *subw = nkept + T2;
    unsigned char *sel = reinterpret_cast<unsigned char *>(subw + RQ_NW * NSUB + 4);
    uint64_t *subs = kept + (size_t)T2 * KPMAX;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my appends have left the wave
    for (int qi = 0; qi < 32; ++qi) {
        const int q = wave * 32 + qi;
        const uint32_t c = min(lds_load_u32(cnt + q), (uint32_t)RQ_CAP);
        if (c > (uint32_t)RQ_MARK) {
            if (lane < NSUB) subw[wave * NSUB + lane] = min(c - min(c, (uint32_t)(lane * SUBCAP)), (uint32_t)SUBCAP);
            if (lane == 0) cnt[q] = 0;
            prune_query<false, true, KPV>(kept, subs, tau_f, nkept, subw + wave * NSUB, q, lane, kp, tau_g, sel + wave * sel_scratch_bytes<KPV>());
        }
    }
}

// The scan of one workgroup.  KS = 8-element k-steps (1-KiB pieces) per row: 8 / 12 / 16 (dim <= 64 / 96 / 128), 20 / 24 / 28 / 32 (dim <= 160 / 192 / 224 / 256).
template <int KS, int KPV>
__device__ __forceinline__ void rq_scan(
    const unsigned char *__restrict__ Qf /* [query group of 32][KS][64][16 B] */, const unsigned char *__restrict__ Vf /* same, rows */,
    const float *__restrict__ vnorm /* [rows padded]: sum v^2 (0: dot / cosine), +inf past N */, uint32_t N, uint32_t nq, uint32_t QT,
    uint32_t S, uint32_t G /* 32-row groups */, uint64_t *__restrict__ cand, uint64_t *__restrict__ partial, uint32_t *__restrict__ tau_g_all,
    uint32_t kp, uint32_t list_base, uint32_t Stot, uint32_t *__restrict__ pcnt, const uint32_t *__restrict__ gate) {
    if (gate && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) return;
    // The operand ring holds R pieces in ACCUMULATION registers (a whole group up to 160 dimensions: 16 KiB, ~4,000 cycles ahead; 8 - 14
    // pieces beyond, where the query fragments need the room).  History of the ring, all measured at configs[1] on one box
    // (profiles/r6/c2_ring_ab.txt): (1) inline-asm loads and waits tied to compiler-allocated ring registers: 495k QPS and WRONG once in
    // ~10^5 queries -- hipcc carried a ring element through a v_mov AHEAD of its asm wait, reading a load in flight (a soak against the
    // int8 route found it); (2) plain compiler-visible loads, ring half a group: correct, 474k; (3) this form: the ring in registers the
    // compiler cannot allocate, named literally (see rq_piece above).
    constexpr int R = rq_ring_len(KS), RB = rq_ring_base(KS), RA = rq_acc_base(KS);
    constexpr int GU = rq_unit_groups(KS), P = GU * KS;   // a loop iteration = one UNIT of GU groups = P pieces; slot of piece p: p % R
    static_assert(P % R == 0 && KS % 4 == 0, "ring wrap");
    asm volatile("" ::: "v255");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *tau_f = reinterpret_cast<float *>(smem);              // exclusive bound: pass <=> key < tau
    uint32_t *cnt = reinterpret_cast<uint32_t *>(tau_f + T2);    // append slots handed out since the query's last prune
    uint32_t *nkept = cnt + T2;
    uint32_t *subw = nkept + T2;                                 // per wave: prune_query's view of one append buffer
    unsigned char *sel = reinterpret_cast<unsigned char *>(subw + RQ_NW * NSUB + 4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t fr = lane & 31, fh = lane >> 5;

    // block -> (stream s, query tile qt): the QT workgroups of a stream share (blockIdx % 8), i.e. one XCD under round-robin
    // placement, and fetch the same groups at about the same time (one HBM fetch, QT - 1 L2 hits)
    const uint32_t b = blockIdx.x;
    const uint32_t x = b & 7, jb = b >> 3;
    const uint32_t qt = jb % QT;
    const uint32_t s = (jb / QT) * 8 + x;
    // my units (ids ascend along a stream); the mirror is padded to whole 256-row tiles = 8 groups, rows past N carry +inf norms (a unit's
    // groups past G are there and pass nothing)
    const uint32_t U = (G + GU - 1) / GU;
    const uint32_t u0 = (uint32_t)(((uint64_t)s * U) / S), u1 = (uint32_t)(((uint64_t)(s + 1) * U) / S);
    uint32_t *tau_g = tau_g_all ? tau_g_all + (size_t)qt * T2 : nullptr;
    uint64_t *kept = cand + (size_t)b * BLKV_ENTRIES;
    uint64_t *subs = kept + (size_t)T2 * KPMAX;

    for (int tq = tid; tq < T2; tq += 512) {
        const bool live = (qt * T2 + tq) < nq;
        const uint32_t g = (tau_g && live) ? __hip_atomic_load(tau_g + tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
        float t0 = live ? __builtin_inff() : -__builtin_inff();
        if (g != 0xFFFFFFFFu) t0 = fminf(t0, f32_unkey(g + 1));
        tau_f[tq] = t0;
        cnt[tq] = 0;
        nkept[tq] = 0;
    }
    __syncthreads();  // (the only barrier ahead of the stream end)

    const unsigned char *qsrc = Qf + ((size_t)(qt * (T2 / 32) + wave) * KS) * 1024 + lane * 16;
    const uint32_t lane16 = (uint32_t)lane * 16;
    const uint32_t lds_state = (uint32_t)(size_t)(lptr_t)tau_f;
    const uint32_t fr4 = fr * 4;
    constexpr size_t ubytes = (size_t)P * 1024;
    // a wave whose 32 queries are all past nq has nothing to select: it still runs (its registers are there), but never passes

    uint32_t u = u0;
#ifdef ZVDB_ABL_RQ_NOLOOP   // (timing-only ablation: what a launch costs without its scan -- prologue + stream end: profiles/r6/c2_ring_ab.txt)
    u = u1;
#endif
    // The group loop is entered once per launch and once more after every prune round of this wave: a prune is a CALL and nothing of
    // the loop's register state lives across it (bf_bf16_stream.hip found hipcc spilling the query fragments around an inlined one).
    while (u < u1) {
        u32x4 B[KS];
#pragma unroll
        for (int j = 0; j < KS; ++j) B[j] = *reinterpret_cast<const u32x4 *>(qsrc + (size_t)j * 1024);
        const float thr = -tau_f[wave * 32 + fr];
        {   // ring prologue: pieces 0 .. R - 1 of unit u
            const gbytes_t cur = uniform_ptr(Vf + (size_t)u * ubytes);
            static_for<R>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                rq_ring_fill<RB + 4 * J, (J % 4) * 1024>(cur + (J / 4) * 4096, lane16);
            });
        }
        f32x16 init;  // start values of the group about to be multiplied: -|v|^2 / 2 of the register's row (rows past N: -inf)
        {
            const float vh = -0.5f * vnorm[(size_t)32 * GU * u + fr];
#pragma unroll
            for (int r = 0; r < 16; ++r) init[r] = __shfl(vh, (r & 3) + 8 * (r >> 2) + 4 * (int)fh);
        }
        rq_frags_landed<KS>(B);
        bool want_prune = false;
        do {
            const uint32_t un = min(u + 1, u1 - 1);  // (past the stream end: a harmless re-read of its last unit)
            const gbytes_t cur = uniform_ptr(Vf + (size_t)u * ubytes);
            const gbytes_t nxt = uniform_ptr(Vf + (size_t)un * ubytes);
            static_for<P>([&](auto pc) {
                constexpr int PI = decltype(pc)::value;
                constexpr int T = PI / KS, J = PI % KS;   // group of the unit, piece of the group
                constexpr int NP = PI + R;  // the piece this piece's slot is refilled with: of this unit, or of the next
                const gbytes_t base = NP < P ? cur + (NP / 4) * 4096 : nxt + ((NP - P) / 4) * 4096;
                const uint32_t g = u * GU + T;
                if constexpr (J == 0) {
                    rq_piece_first<RA, RB + 4 * (PI % R), R - 1, (NP % 4) * 1024>(init, B[J], base, lane16);
                    // the norms of the group after this one
                    const uint32_t gn = T + 1 < GU ? g + 1 : un * GU;
                    rq_norm_load(uniform_ptr(reinterpret_cast<const unsigned char *>(vnorm + (size_t)32 * gn)), fr4);
                } else {
                    rq_piece<RA, RB + 4 * (PI % R), R - 1, (NP % 4) * 1024>(B[J], base, lane16);
                }
                if constexpr (J == KS - 3) {
                    // start values of the next group, three pieces ahead of its first MFMA (their LDS round trip -- ds_bpermute -- hides
                    // behind 12 MFMAs; at the group's end it was exposed: profiles/r6/c2_ring_ab.txt).  JS ring loads are younger than the norms'.
                    const float vh = -0.5f * rq_norm_take<KS - 3>();
#pragma unroll
                    for (int r = 0; r < 16; ++r) init[r] = __shfl(vh, (r & 3) + 8 * (r >> 2) + 4 * (int)fh);
                }
                if constexpr (J == KS - 1) {
                    // selection of the finished group (32 rows x my 32 queries): the rare pass path appends to the queries' buffers
#ifndef ZVDB_ABL_RQ_NOLOADS
                    const float m = rq_group_max<RA>();
#else
                    float m = rq_group_max<RA>();
                    m = fminf(m, -__builtin_inff());   // (the ablation multiplies stale rows: nothing may pass)
#endif
                    if (__ballot(m > thr)) {
                        bool need = false;
                        const uint32_t q = (uint32_t)wave * 32 + fr;
                        float acc[16];
                        rq_acc_read<RA>(acc);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (acc[r] > thr) {
                                const uint32_t sl = atomicAdd(cnt + q, 1u);
                                const uint32_t id = 32 * g + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4 * fh;
                                if (sl < (uint32_t)RQ_CAP) rq_append(subs + (size_t)q * RQ_CAP + sl, ((uint64_t)f32_key(-acc[r]) << 32) | id);
                                need |= sl == (uint32_t)RQ_HW;
                            }
                        }
                        if (__ballot(need)) want_prune = true;  // (<= 32 GU more appends per query before the loop is left: RQ_CAP holds them)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            ++u;
        } while (u < u1 && !want_prune);
        if (want_prune) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's loads in flight have landed before the callee may use accumulation registers
            rq_prune_wave<KPV>(kept, tau_g, kp, lds_state);
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {   // stream end: lane-per-query copy when nothing needs selecting, else the radix select (a wave's own 32 queries: no rendezvous)
        unsigned char *selw = sel + wave * sel_scratch_bytes<KPV>();
        const uint32_t q = (uint32_t)wave * 32 + fr;
        const uint32_t qg = qt * T2 + q;
        const bool mine = lane < 32 && qg < nq;
        const uint32_t c = mine ? min(lds_load_u32(cnt + q), (uint32_t)RQ_CAP) : 0u;
        const uint32_t nk = mine ? nkept[q] : 0u;
        const bool simple = mine && nk + c <= (uint32_t)KPV;
        if (simple) {
            uint64_t *dst = partial + ((size_t)qg * Stot + list_base + s) * KPV;
            for (uint32_t e = 0; e < nk; ++e) dst[e] = kept[(size_t)q * KPV + e];
            for (uint32_t e = 0; e < c; ++e) dst[nk + e] = subs[(size_t)q * RQ_CAP + e];
            // (entries past the count are not written: every reader of a partial list stops at pcnt -- k_bf_merge_sel; filling them cost 62 scattered
            //  8-byte stores per lane, most of the 25 us a launch took without its scan: profiles/r6/c2_ring_ab.txt)
            pcnt[(size_t)qg * Stot + list_base + s] = nk + c;
        }
        unsigned long long todo = __ballot(mine && !simple);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t q2 = (uint32_t)wave * 32 + (uint32_t)l;
            const uint32_t qg2 = qt * T2 + q2;
            const uint32_t c2 = min(lds_load_u32(cnt + q2), (uint32_t)RQ_CAP);
            if (lane < NSUB) subw[wave * NSUB + lane] = min(c2 - min(c2, (uint32_t)(lane * SUBCAP)), (uint32_t)SUBCAP);
            prune_query<false, true, KPV>(kept, subs, tau_f, nkept, subw + wave * NSUB, q2, lane, kp, nullptr, selw);
#pragma unroll
            for (int ci = 0; ci < KPV / 64; ++ci)
                partial[((size_t)qg2 * Stot + list_base + s) * KPV + lane + 64 * ci] = kept[(size_t)q2 * KPV + lane + 64 * ci];
            if (lane == 0) pcnt[(size_t)qg2 * Stot + list_base + s] = nkept[q2];
        }
    }
}

// The kernels: one symbol per (KS, kept-list width, gated) so that amdgpu_num_vgpr -- a literal, not a template argument -- can end the
// compiler's allocation at that KS's ring base.  GATED = 1: the same code under its own symbol for the gated repair launch (a no-op
// unless the verification failed), so that profiler statistics do not average no-ops into the scan.
template <int KS, int KPV, int GATED>
struct RqKernel;
#define RQ_KERNEL(KS, KPV, GATED, NV)                                                                                                        \
    __global__ __attribute__((amdgpu_flat_work_group_size(512, 512), amdgpu_num_vgpr(NV / 2))) void k_f32_regq_##KS##_##KPV##_##GATED(            \
        const unsigned char *__restrict__ Qf, const unsigned char *__restrict__ Vf, const float *__restrict__ vnorm, uint32_t N, uint32_t nq,   \
        uint32_t QT, uint32_t S, uint32_t G, uint64_t *__restrict__ cand, uint64_t *__restrict__ partial, uint32_t *__restrict__ tau_g_all,    \
        uint32_t kp, uint32_t list_base, uint32_t Stot, uint32_t *__restrict__ pcnt, const uint32_t *__restrict__ gate) {                      \
        static_assert(NV == rq_acc_base(KS), "the compiler's last register + 1 = the first of the registers the asm statements own");                                          \
        rq_scan<KS, KPV>(Qf, Vf, vnorm, N, nq, QT, S, G, cand, partial, tau_g_all, kp, list_base, Stot, pcnt, gate);                           \
    }                                                                                                                                        \
    template <>                                                                                                                              \
    struct RqKernel<KS, KPV, GATED> {                                                                                                        \
        static constexpr auto fn = &k_f32_regq_##KS##_##KPV##_##GATED;                                                                        \
    };
#define RQ_KERNELS(KS, NV) RQ_KERNEL(KS, 64, 0, NV) RQ_KERNEL(KS, 64, 1, NV) RQ_KERNEL(KS, 128, 0, NV) RQ_KERNEL(KS, 256, 0, NV)
RQ_KERNELS(8, 172)
RQ_KERNELS(12, 188)
RQ_KERNELS(16, 172)
RQ_KERNELS(20, 156)
RQ_KERNELS(24, 188)
RQ_KERNELS(28, 180)
RQ_KERNELS(32, 204)
static_assert(KP == 64 && KPW == 128 && KPMAX == 256, "kernel symbols are spelled with the kept-list widths");

// ---- thresholds from the corpus' second moments ---------------------------------------------------------------------------
// The guessed-threshold pipeline needs, per query, mean and variance of the screen key over the rows (tau = mean + z sigma,
// k_bf16_guess's rule).  The stream kernels measure them per batch by scanning a sample of the rows with the batch's queries
// (a moments launch + a guess launch: 27 + 9 us of a 540-us configs[1] step).  For short rows they follow from moments of the
// ROWS alone, computed once per mirror build: key = n - q.v with n = |v|^2 / 2 (dot / cosine: n = 0), so
//   mean = E[n] - q.mu,   var = Var(n) - 2 q.c + q' Sigma q,   mu = E[v], c = Cov(n, v), Sigma = Cov(v)
// -- one wave per query reads the D x D matrix (64 KB at 128 dimensions, L2-resident) inside ONE small launch per batch.
// What a guess gets wrong is verified and repaired as before; results do not depend on it (tested: = the sampled guess).
constexpr uint32_t RQ_STAT_RUNS = 512;     // sampled runs of 32 consecutive rows, evenly spread over the shard (16,384 rows)
constexpr uint32_t RQ_STAT_BLOCKS = 64;

// acc (double): [0] rows, [1] sum n, [2] sum n^2, [3, 3 + D) sum x_i, [3 + D, 3 + 2 D) sum n x_i, [3 + 2 D, 3 + 2 D + D^2) sum x_i x_j
template <typename RT>
__global__ __launch_bounds__(256) void k_rq_stats_acc(const RT *__restrict__ rows, uint32_t ld, uint32_t dim, uint32_t D, uint64_t N, int l2,
                                                      uint32_t runs, double *__restrict__ acc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *x = reinterpret_cast<float *>(smem);   // [32][D + 1]
    float *nr = x + 32 * (D + 1);                 // [32]
    __shared__ uint32_t nvalid;
    const int tid = threadIdx.x;
    const uint64_t G = (N + 31) / 32;
    const uint32_t npairs = D * D, xs = D + 1;
    double s1 = 0.0, sn1 = 0.0, cn = 0.0, sn = 0.0, snn = 0.0;
    for (uint32_t r = blockIdx.x; r < runs; r += gridDim.x) {
        const uint64_t g = (uint64_t)r * G / runs;
        __syncthreads();
        for (uint32_t t = tid; t < 32 * D; t += 256) {
            const uint32_t rr = t / D, i = t % D;
            const uint64_t row = 32 * g + rr;
            x[rr * xs + i] = (row < N && i < dim) ? row_elem(rows + row * ld, i) : 0.0f;
        }
        __syncthreads();
        if (tid < 32) {
            float a = 0.0f;
            if (l2)
                for (uint32_t i = 0; i < D; ++i) a += x[tid * xs + i] * x[tid * xs + i];
            nr[tid] = 0.5f * a;
        }
        if (tid == 0) nvalid = (uint32_t)min((uint64_t)32, N - 32 * g);
        __syncthreads();
        const uint32_t nv = nvalid;
        // the run's 32-row partial of every pair sum goes straight to the fp64 accumulator (one-off work per mirror build: 8M atomics at D = 128)
        for (uint32_t pe = tid; pe < npairs; pe += 256) {
            const uint32_t i = pe / D, j = pe % D;
            float a = 0.0f;
            for (uint32_t rr = 0; rr < nv; ++rr) a += x[rr * xs + i] * x[rr * xs + j];
            atomicAdd(acc + 3 + 2 * D + pe, (double)a);
        }
        if ((uint32_t)tid < D) {
            float a = 0.0f, b2 = 0.0f;
            for (uint32_t rr = 0; rr < nv; ++rr) { a += x[rr * xs + tid]; b2 += nr[rr] * x[rr * xs + tid]; }
            s1 += (double)a;
            sn1 += (double)b2;
        }
        if (tid == 0) {
            for (uint32_t rr = 0; rr < nv; ++rr) { sn += (double)nr[rr]; snn += (double)nr[rr] * (double)nr[rr]; }
            cn += (double)nv;
        }
    }
    if ((uint32_t)tid < D) { atomicAdd(acc + 3 + tid, s1); atomicAdd(acc + 3 + D + tid, sn1); }
    if (tid == 0) { atomicAdd(acc + 0, cn); atomicAdd(acc + 1, sn); atomicAdd(acc + 2, snn); }
}

// kstat (float): [0] E[n], [1] Var(n), [2, 2 + D) mu, [2 + D, 2 + 2 D) c = Cov(n, v), [2 + 2 D, ...) Sigma (D x D, symmetric)
__global__ __launch_bounds__(256) void k_rq_stats_fin(const double *__restrict__ acc, uint32_t D, float *__restrict__ kstat) {
    const double n = acc[0] > 0.0 ? acc[0] : 1.0;
    const double en = acc[1] / n;
    for (uint32_t t = threadIdx.x; t < D * D; t += 256) {
        const uint32_t i = t / D, j = t % D;
        kstat[2 + 2 * D + t] = (float)(acc[3 + 2 * D + t] / n - (acc[3 + i] / n) * (acc[3 + j] / n));
    }
    for (uint32_t i = threadIdx.x; i < D; i += 256) {
        kstat[2 + i] = (float)(acc[3 + i] / n);
        kstat[2 + D + i] = (float)(acc[3 + D + i] / n - en * (acc[3 + i] / n));
    }
    if (threadIdx.x == 0) {
        kstat[0] = (float)en;
        kstat[1] = (float)fmax(acc[2] / n - en * en, 0.0);
    }
}

// one workgroup of 4 waves per query: tau_g[q] = key(mean + z sigma) of the screen key over the rows, from the rows' moments (header of
// this section).  Wave w walks rows [w D / 4, (w + 1) D / 4) of Sigma for the lane's columns (coalesced: lanes = consecutive columns; a
// single wave walking all D rows is one dependent chain of D L2 round trips: 19 us per batch instead of 6).
// The same launch writes the scan's operand (k_f32_pack_frag's layout: piece (group, k-step j, lane l) = elements 8 j + 4 (l >> 5) + [0, 4)
// of query 32 group + (l & 31)): blocks nq .. nq_pad - 1 write the zero rows of the padding and no threshold.
__global__ __launch_bounds__(256) void k_rq_guess(const float *__restrict__ queries, uint32_t nq, uint32_t dim, uint32_t D, const float *__restrict__ kstat,
                                                  float z, int l2, uint32_t *__restrict__ tau_g, uint32_t *__restrict__ verify_flag,
                                                  unsigned char *__restrict__ q_op) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && verify_flag) { verify_flag[0] = 0; verify_flag[1] = 0; }  // ([1]: deficient queries of the batch)
    __shared__ __attribute__((aligned(16))) float q[256];
    __shared__ double part[4][3];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t qi = blockIdx.x;
    if ((uint32_t)tid < D) q[tid] = ((uint32_t)tid < dim && qi < nq) ? queries[(size_t)qi * dim + tid] : 0.0f;
    __syncthreads();
    if (q_op && (uint32_t)tid < D / 4) {  // 16-byte chunk t = elements [4 t, 4 t + 4): k-step t / 2, half t & 1
        const uint32_t KS = D / 8, j = (uint32_t)tid >> 1, hh = (uint32_t)tid & 1;
        *reinterpret_cast<u32x4 *>(q_op + ((size_t)(qi >> 5) * KS + j) * 1024 + (hh * 32 + (qi & 31)) * 16) = *reinterpret_cast<const u32x4 *>(q + 4 * tid);
    }
    if (qi >= nq) return;
    const float *mu = kstat + 2, *c = kstat + 2 + D, *Sg = kstat + 2 + 2 * D;
    const uint32_t j0 = (uint32_t)w * (D / 4), j1 = j0 + D / 4;
    double qmu = 0.0, qc = 0.0, qsq = 0.0;
    for (uint32_t i = lane; i < D; i += 64) {
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
        for (uint32_t j = j0; j < j1; j += 8) {  // (D / 4 = 16 .. 64 rows per wave, a multiple of 8; 8 independent loads in flight per column, then the sums)
            float sv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sv[e] = Sg[(size_t)(j + e) * D + i];
#pragma unroll
            for (int e = 0; e < 8; e += 4) {
                t0 += sv[e] * q[j + e];
                t1 += sv[e + 1] * q[j + e + 1];
                t2 += sv[e + 2] * q[j + e + 2];
                t3 += sv[e + 3] * q[j + e + 3];
            }
        }
        qsq += (double)q[i] * (double)((t0 + t1) + (t2 + t3));
        if (w == 0) {
            qmu += (double)q[i] * (double)mu[i];
            qc += (double)q[i] * (double)c[i];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        qmu += __shfl_xor(qmu, off);
        qc += __shfl_xor(qc, off);
        qsq += __shfl_xor(qsq, off);
    }
    if (lane == 0) { part[w][0] = qmu; part[w][1] = qc; part[w][2] = qsq; }
    __syncthreads();
    if (tid == 0) {
        qmu = part[0][0];
        qc = part[0][1];
        qsq = (part[0][2] + part[1][2]) + (part[2][2] + part[3][2]);
        uint32_t out = 0xFFFFFFFFu;  // no bound
        const double mean = (l2 ? (double)kstat[0] : 0.0) - qmu;
        const double var = l2 ? (double)kstat[1] - 2.0 * qc + qsq : qsq;
        if (var > 0.0) {
            const float tk = (float)(mean + (double)z * sqrt(var));
            if (tk == tk && fabsf(tk) < 3.0e38f) out = f32_key(tk);
        }
        tau_g[qi] = out;
    }
}

template <int KS, int KPV>
int rq_launch_geo(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint64_t *cand, uint64_t *partial,
                  uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot, uint32_t *pcnt, const uint32_t *gate, hipStream_t s) {
    constexpr uint32_t lds = rq_lds_bytes<KPV>();
    static_assert(lds <= 64 * 1024, "LDS budget");
    const uint32_t G = (uint32_t)((h->count + 31) / 32);
    const float *vn = h->norm_ff;  // (k_f32_frag_norms: zeros for dot / cosine, +inf past the last row)
    auto go = [&](auto kern) -> int {
        hipLaunchKernelGGL(kern, dim3(QT * S), dim3(512), lds, s, q_op, h->rows_ff, vn, (uint32_t)h->count, nq, QT, S, G, cand, partial, tau_g, kp,
                           list_base, Stot, pcnt, gate);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    };
    if constexpr (KPV == KP) {
        if (gate) return go(RqKernel<KS, KP, 1>::fn);
    }
    return go(RqKernel<KS, KPV, 0>::fn);
}

template <int KPV>
int rq_launch(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint64_t *cand, uint64_t *partial,
              uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot, uint32_t *pcnt, const uint32_t *gate, hipStream_t s) {
    const uint32_t KS = f32s_ksteps(h->dim);
    if (KS == 8) return rq_launch_geo<8, KPV>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, gate, s);
    if (KS == 12) return rq_launch_geo<12, KPV>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, gate, s);
    if (KS == 16) return rq_launch_geo<16, KPV>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, gate, s);
    if (KS == 20) return rq_launch_geo<20, KPV>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, gate, s);
    if (KS == 24) return rq_launch_geo<24, KPV>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, gate, s);
    if (KS == 28) return rq_launch_geo<28, KPV>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, gate, s);
    if (KS == 32) return rq_launch_geo<32, KPV>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, gate, s);
    set_error("fp32 register-query scan: dim %u has no instantiation", h->dim);
    return ZVDB_ERR_UNSUPPORTED;
}

}  // namespace

// fp32 screens of 5 .. 256 dimensions; ZVDB_BF_F32_KERNEL=v6 (staged) / stream (the 128-query kernel) select the older kernels
bool f32rq_eligible(const zvdb_hip_index *h) {
    const char *e = tuning("ZVDB_BF_F32_KERNEL");
    if (e && ((e[0] == 'v' && e[1] == '6') || e[0] == 's')) return false;
    return h->dim > 4 && h->dim <= 256;
}

// the rows' moments behind k_rq_guess: rebuilt with the fragment mirror (lazily, when rows were added)
int ensure_f32rq_stats(zvdb_hip_index *h, hipStream_t s) {
    if (h->kstat && h->kstat_count == h->count) return ZVDB_OK;
    if (h->count == 0) return ZVDB_OK;
    const uint32_t D = f32s_ksteps(h->dim) * 8;
    const size_t nacc = 3 + 2 * (size_t)D + (size_t)D * D, nst = 2 + 2 * (size_t)D + (size_t)D * D;
    if (!h->kstat) {  // (sized by the index's dimension, which never changes)
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->kstat_acc), nacc * sizeof(double)));
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->kstat), nst * sizeof(float)));
    }
    ZV_HIP(hipMemsetAsync(h->kstat_acc, 0, nacc * sizeof(double), s));
    const uint64_t G = (h->count + 31) / 32;
    const uint32_t runs = (uint32_t)std::min<uint64_t>(RQ_STAT_RUNS, G);
    const bool l2 = h->metric == ZVDB_METRIC_L2SQ;
    ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = row_elem_t<decltype(rows)>;
        hipLaunchKernelGGL(k_rq_stats_acc<RT>, dim3(std::min(runs, RQ_STAT_BLOCKS)), dim3(256), (32 * (D + 1) + 32) * sizeof(float), s, rows, ld, h->dim, D, h->count,
                           l2 ? 1 : 0, runs, h->kstat_acc);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    }));
    hipLaunchKernelGGL(k_rq_stats_fin, dim3(1), dim3(256), 0, s, h->kstat_acc, D, h->kstat);
    ZV_HIP(hipGetLastError());
    // (a search of this index on ANOTHER stream may follow at once: the moments are complete before this call returns -- once per mirror build)
    ZV_HIP(hipStreamSynchronize(s));
    h->kstat_count = h->count;
    return ZVDB_OK;
}

// thresholds of a batch from the rows' moments + the scan's packed queries: ONE launch (it also clears the verify flag, like the other guess kernels)
int launch_f32rq_guess(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t nq_pad, unsigned char *q_op, float z, uint32_t *tau_g,
                       uint32_t *verify_flag, hipStream_t s) {
    const uint32_t D = f32s_ksteps(h->dim) * 8;
    hipLaunchKernelGGL(k_rq_guess, dim3(q_op ? nq_pad : nq), dim3(256), 0, s, d_queries, nq, h->dim, D, h->kstat, z, h->metric == ZVDB_METRIC_L2SQ ? 1 : 0, tau_g,
                       verify_flag, q_op);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// QT = tiles of 256 queries; S = streams (QT * S workgroups); the mirror is the 128-query kernel's (ensure_f32_frag_mirror)
int launch_f32rq_scan(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint64_t *cand, uint64_t *partial,
                      uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot, uint32_t *pcnt, const uint32_t *gate, uint32_t kpv,
                      hipStream_t s) {
    if (kpv == (uint32_t)KPMAX) return rq_launch<KPMAX>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, gate, s);
    if (kpv == (uint32_t)KPW) return rq_launch<KPW>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, gate, s);
    return rq_launch<KP>(h, q_op, nq, QT, S, cand, partial, tau_g, kp, list_base, Stot, pcnt, gate, s);
}

}  // namespace zvdb
