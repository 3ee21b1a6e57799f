// bf_i8_stream.hip -- the barrier-free stream scan (bf_fp4_stream.hip) for the int8 screen: recall-exact route.
//
// Same structure as the fp4 kernel -- query tile resident in LDS in MFMA fragment order, corpus mirror stored in
// fragment order and streamed HBM -> registers per wave, no barrier in the steady state, one scan against guessed
// and verified thresholds -- with the int8 instruction's geometry: v_mfma_i32_32x32x32_i8 sums 32 elements per 32
// cycles, so a 32-byte k-step covers 32 elements (fp4: 64), a 768-d row has KS = 24 k-steps, and a 96-KiB resident
// tile holds 128 queries (4 fragments) instead of 256: a wave step is 32 rows x 128 queries = 4 accumulators of 16
// i32 registers, 96 MFMAs.  Keys are exact integers (the staged kernel's int path): key = (|v'|^2 >> 1) - q'.v';
// the (key, id) radix select is shared (bf_select.h).  The result is the exact k' best by int8 key -- identical to
// the staged kernel's (tested) -- so with k' >= k + margin the final top-k is the exact one (recall 1.0 in bench.py).
//
// MODE 0 (r3) instantiates it for the EXACT-ARITHMETIC fp32 screen on short rows (dim <= 128: BASELINE configs[1], 1M x 128 at batch
// 256): v_mfma_f32_32x32x2_f32 takes one fp32 per lane, so a 16-byte fragment piece feeds 4 MFMAs and a k-step covers 8 elements
// (lane l: elements 8 j + 4 (l >> 5) + [0, 4) of row / query l & 31); KS = 8 / 16 (dim <= 64 / 128): the 128-query tile is 32 / 64
// KiB of LDS.  The staged 256 x 256 kernel streams BOTH operands through LDS and, with one query tile, shares the corpus stream
// with nobody: 0.68 of the fp32-MFMA peak at that shape; here only the corpus moves, HBM -> registers.  It reads a fragment-order
// fp32 mirror (rows_ff: the rows' bytes once more -- which is why this route is for short rows).
//
// MODE 3 instantiates the same 128-query geometry for the fp4 screen (v_mfma_scale_f32_32x32x64_f8f6f4, fp32 keys): a
// 32-byte k-step then covers 64 elements, so KS = 24 is a 1536-d row -- the dimensions the 256-query fp4 kernel
// (bf_fp4_stream.hip, KS <= 12) cannot keep resident.  It reads the fp4 fragment-order mirror built there.
#include "common.h"
#include "wave_select.h"
#include "bf_select.h"
#include "stream_common.h"
#include "row_types.h"

namespace zvdb {

namespace {

constexpr int I8_NF = 4;                  // query fragments per step (128 queries per workgroup)
constexpr int I8_TQ = 32 * I8_NF;
constexpr int I8_RING_MAX = 12;           // A prefetch ring (k-steps): min(KS, this); 6 -> 12 measured +3.5 %
constexpr int I8_CAP = NSUB * SUBCAP;     // append-buffer entries per (workgroup, query)
constexpr int I8_HW = 384, I8_MARK = 64, I8_NW = 8;

using i32x16 = __attribute__((ext_vector_type(16))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x8 = __attribute__((ext_vector_type(8))) int;
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int MODE> struct Acc128 { using type = i32x16; };
template <> struct Acc128<3> { using type = f32x16; };
template <> struct Acc128<0> { using type = f32x16; };


__host__ __device__ inline uint32_t i8s_lds_bytes(uint32_t KS, uint32_t sel_bytes = SEL_SCRATCH) {
    return I8_NF * KS * 1024 + T2 * 4 * 3 + T2 * NSUB * 4 + 16 + I8_NW * sel_bytes;
}

// KS = 32-byte (32-element) k-steps per row; MOM: moments mode (sum of key, key^2 per query instead of selecting)
// MODE: 2 = int8 (int keys), 3 = fp4 (fp32 keys), 0 = fp32 (fp32 keys; a k-step = 8 elements)
// GATED = 1: the same code under its own symbol for the gated repair launch (see k_fp4_stream)
template <int MODE, int KS, bool MOM, int KPV = KP, int GATED = 0>
__global__ __launch_bounds__(512, 2) void k_i8_stream(
    const unsigned char *__restrict__ Qf /* [query group of 32][KS][64][16 B] */, const unsigned char *__restrict__ Vf /* same, rows */,
    const void *__restrict__ vnorm_v /* int8: sum q^2 (i32, INT_MAX past N); fp4: sum v'^2 (f32, +inf past N); nullptr: dot / cosine */, uint32_t N,
    uint32_t nq, uint32_t QT /* tiles of 128 queries */, uint32_t S, uint32_t NT, uint64_t *__restrict__ cand,
    uint64_t *__restrict__ partial, uint32_t *__restrict__ tau_g_all, uint32_t kp, uint32_t list_base, uint32_t Stot,
    uint32_t *__restrict__ pcnt, uint32_t max_tiles, const uint32_t *__restrict__ gate, float *__restrict__ mom) {
    if (gate && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) return;
    constexpr int I8_RING = KS % I8_RING_MAX == 0 ? I8_RING_MAX : KS % 8 == 0 ? 8 : 6;
    static_assert(KS % I8_RING == 0 && KS % 2 == 0, "ring wrap / set parity");
    constexpr bool INT = MODE == 2;
    using acc_t = typename Acc128<MODE>::type;
    const int32_t *vnorm = reinterpret_cast<const int32_t *>(vnorm_v);  // (read as raw 32-bit words; fp4 reinterprets)
    static_assert((2 * KS - 1) * 1024 <= 65535, "ds_read immediate offsets");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t qbytes = I8_NF * KS * 1024;
    float *tau_f = reinterpret_cast<float *>(smem + qbytes);  // int32 bounds (prune_query's INTKEY view): pass <=> key < tau
    int32_t *tau = reinterpret_cast<int32_t *>(tau_f);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(tau_f + T2);
    uint32_t *nkept = cnt + T2;
    uint32_t *subcnt = nkept + T2;
    uint32_t *ctl = subcnt + T2 * NSUB;
    unsigned char *sel = reinterpret_cast<unsigned char *>(ctl + 4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t fr = lane & 31, fh = lane >> 5;

    const uint32_t b = blockIdx.x;
    const uint32_t x = b & 7, jb = b >> 3;
    const uint32_t qt = jb % QT;
    const uint32_t s = (jb / QT) * 8 + x;
    const uint32_t per = (NT + S - 1) / S;
    const uint32_t t_begin = min(s * per, NT);
    const uint32_t t_end = min(t_begin + (max_tiles ? min(per, max_tiles) : per), NT);
    const uint32_t g_begin = 8 * t_begin, g_end = min(8 * t_end, (N + 31) / 32);
    const uint32_t nsteps = g_end > g_begin ? (g_end - g_begin + I8_NW - 1) / I8_NW : 0;
    uint32_t *tau_g = tau_g_all ? tau_g_all + (size_t)qt * I8_TQ : nullptr;
    uint64_t *kept = cand + (size_t)b * BLKV_ENTRIES;  // (regions are sized for the widest list)
    uint64_t *subs = kept + (size_t)T2 * KPMAX;

    for (int tq = tid; tq < I8_TQ; tq += 512) {
        const bool live = (qt * I8_TQ + tq) < nq;
        const uint32_t g = (tau_g && live) ? __hip_atomic_load(tau_g + tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
        // live rows satisfy acc >= -2^29 (dim <= 16384), rows past N start near -2^30: 0x30000000 admits every live row
        if constexpr (INT) {
            int32_t t0 = live ? 0x30000000 : -0x7FFFFFFF;
            if (g != 0xFFFFFFFFu) { const int32_t gv = (int32_t)(g ^ 0x80000000u); t0 = min(t0, gv == 0x7FFFFFFF ? gv : gv + 1); }
            tau[tq] = t0;
        } else {
            float t0 = live ? __builtin_inff() : -__builtin_inff();
            if (g != 0xFFFFFFFFu) t0 = fminf(t0, f32_unkey(g + 1));
            tau_f[tq] = t0;
        }
        cnt[tq] = 0;
        nkept[tq] = 0;
    }
    for (int i = tid; i < I8_TQ * NSUB; i += 512) subcnt[i] = 0;
    if (tid < 4) ctl[tid] = 0;
    for (uint32_t p = wave; p < I8_NF * KS; p += I8_NW)
        __builtin_amdgcn_global_load_lds((gptr_t)(Qf + (size_t)qt * qbytes + (size_t)p * 1024 + lane * 16),
                                         (lptr_t)(smem + p * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto round = [&]() {
        __syncthreads();
        if (tid == 0) ctl[0] = 0;
        for (int q = wave; q < I8_TQ; q += I8_NW) {
            const uint32_t c = min(lds_load_u32(cnt + q), (uint32_t)I8_CAP);
            if (c > (uint32_t)I8_MARK) {
                if (lane < NSUB) subcnt[q * NSUB + lane] = min(c - min(c, (uint32_t)(lane * SUBCAP)), (uint32_t)SUBCAP);
                if (lane == 0) cnt[q] = 0;
                prune_query<INT, true, KPV>(kept, subs, tau_f, nkept, subcnt + q * NSUB, q, lane, kp, tau_g, sel + wave * sel_scratch_bytes<KPV>());
            }
        }
        __syncthreads();
    };

    constexpr size_t gbytes = (size_t)KS * 1024;
    const uint32_t g_last = g_end - 1;
    const uint32_t lane16 = (uint32_t)lane * 16;
    u32x4 A[I8_RING];
    if (nsteps > 0) {
        const gbytes_t p0 = uniform_ptr(Vf + (size_t)min(g_begin + (uint32_t)wave, g_last) * gbytes);
#pragma unroll
        for (int i = 0; i < I8_RING; ++i) A[i] = *(const __attribute__((address_space(1))) u32x4 *)(p0 + lane16 + i * 1024);
    }
    const uint32_t qlds = (uint32_t)(size_t)(lptr_t)smem + lane16;
    const uint32_t qb0 = qlds, qb1 = qlds + 2 * KS * 1024;
    u32x4 BX[4], BY[4];
    // the 4 query fragments of k-step J: ((f * KS + J) << 10) + lane * 16; fragments 0, 1 from qb0, fragments 2, 3 from qb1
    auto rd4 = [&](u32x4 (&d)[4], auto jc) {
        constexpr int J = decltype(jc)::value;
        const uint32_t b0 = qb0, b1 = qb1;  // (locals: clang rejects captured variables as asm operands of a nested generic lambda)
        asm volatile("ds_read_b128 %0, %4 offset:%6\n\tds_read_b128 %1, %4 offset:%7\n\t"
                     "ds_read_b128 %2, %5 offset:%6\n\tds_read_b128 %3, %5 offset:%7"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                     : "v"(b0), "v"(b1), "n"((0 * KS + J) * 1024), "n"((1 * KS + J) * 1024));
    };
    // HAZARD of the hand-written reads: hipcc believes the registers written by rd4 are valid when the asm statement ends, but
    // they land only behind wait4's lgkmcnt.  The build must therefore not spill or copy them in between: this holds while the
    // kernel has no VGPR spill inside the step loop (checked in the ISA: tools/cc_f7.sh; an experimental 12-wave build of the
    // int8 kernel that spilled 60 VGPRs returned wrong neighbours), and the bit-exact stream == staged tests would catch it.
    auto wait4 = [&](u32x4 (&d)[4]) {
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
    };
    rd4(BX, std::integral_constant<int, 0>{});
    rd4(BY, std::integral_constant<int, 1>{});

    float m1[I8_NF], m2[I8_NF];  // moments mode only
    uint32_t mrows = 0;
#pragma unroll
    for (int f = 0; f < I8_NF; ++f) { m1[f] = 0.0f; m2[f] = 0.0f; }
    int32_t vn_next = 0;
    if (nsteps > 0 && vnorm) vn_next = vnorm[(size_t)32 * min(g_begin + (uint32_t)wave, g_last) + fr];

    for (uint32_t it = 0; it < nsteps; ++it) {
        if (!MOM && lds_load_u32(ctl) != 0) round();
        const uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)(g_begin + wave + I8_NW * it));
        if (g >= g_end) continue;
        // (hipcc waits for this load with s_waitcnt vmcnt(0) at the top of EVERY step -- its first use sits behind the loop's back
        // edge -- which also drains the A fragments in flight.  Requesting it by inline asm and waiting with vmcnt(KS) instead (r3,
        // profiles/r3/bf16_stream_ab.txt: r3_ab17) made the int8 scan 2.7 % SLOWER and left the fp4 scan unchanged: as in the bf16
        // kernel, the drain keeps the query-tile workgroups of a stream close enough to share their fetches in the L2.)
        const int32_t vn_mine = vn_next;  // loaded one step ahead (see bf_fp4_stream.hip)
        if (vnorm) vn_next = vnorm[(size_t)32 * min(g + I8_NW, g_last) + fr];
        // accumulator start = -(|v'|^2 >> 1) of the register's corpus row; rows past N: INT_MAX >> 1 ~ 2^30, never pass
        acc_t init;
        if (vnorm) {  // (one uniform branch around 16 shuffles: per-register selects on `vnorm` cost 11 % of the scan)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int w = __shfl(vn_mine, (r & 3) + 8 * (r >> 2) + 4 * (int)fh);
                if constexpr (INT) init[r] = -(w >> 1);
                else init[r] = -0.5f * __int_as_float(w);
            }
        } else {
            asm volatile("");  // (keeps this path a real branch: if-converted, its VALU ops would run on every L2 step)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool in = (32 * g + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4 * fh) < N;
                if constexpr (INT) init[r] = in ? 0 : -0x40000000;
                else init[r] = in ? 0.0f : -__builtin_inff();
            }
        }

        acc_t acc[I8_NF];
        const gbytes_t cur = uniform_ptr(Vf + (size_t)g * gbytes);
        const gbytes_t nxt = uniform_ptr(Vf + (size_t)min(g + I8_NW, g_last) * gbytes);
        __builtin_amdgcn_s_setprio(1);
        // one sub-step per k-step: 4 fragments / 4 MFMAs; the two B sets alternate by k-step parity
        static_for<KS>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr int JN = (J + 2) % KS;  // the set freed now is needed two k-steps on (wraps into the next step)
            constexpr int SLOT = J % I8_RING;
            auto &BS = (J & 1) ? BY : BX;
            const i32x4 xa = __builtin_bit_cast(i32x4, A[SLOT]);
            wait4(BS);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < I8_NF; ++f) {
                const i32x4 xb = __builtin_bit_cast(i32x4, BS[f]);
                if constexpr (INT) {
                    acc[f] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa, xb, J == 0 ? init : acc[f], 0, 0, 0);
                } else if constexpr (MODE == 0) {  // four 2-deep fp32 MFMAs per piece: component c = elements 8 J + c (lanes < 32), + 4 (lanes >= 32)
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(xa.x), __int_as_float(xb.x), J == 0 ? init : acc[f], 0, 0, 0);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(xa.y), __int_as_float(xb.y), acc[f], 0, 0, 0);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(xa.z), __int_as_float(xb.z), acc[f], 0, 0, 0);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(xa.w), __int_as_float(xb.w), acc[f], 0, 0, 0);
                } else {
                    const i32x8 wa = {xa.x, xa.y, xa.z, xa.w, 0, 0, 0, 0}, wb = {xb.x, xb.y, xb.z, xb.w, 0, 0, 0, 0};
                    acc[f] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, wb, J == 0 ? init : acc[f], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            rd4(BS, std::integral_constant<int, JN>{});
            if constexpr (J + I8_RING < KS) A[SLOT] = *(const __attribute__((address_space(1))) u32x4 *)(cur + lane16 + (J + I8_RING) * 1024);
            else A[SLOT] = *(const __attribute__((address_space(1))) u32x4 *)(nxt + lane16 + (J + I8_RING - KS) * 1024);
            __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_s_setprio(0);

        if constexpr (MOM) {  // acc = -key; padded rows sit below -2^29 and are left out
#pragma unroll
            for (int f = 0; f < I8_NF; ++f) {
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float a;
                    if constexpr (INT) a = acc[f][r] > -0x30000000 ? (float)acc[f][r] : 0.0f;
                    else a = acc[f][r] > -3.0e38f ? acc[f][r] : 0.0f;
                    s1 += a;
                    s2 += a * a;
                }
                m1[f] += s1;
                m2[f] += s2;
            }
            mrows += min(32u, N - 32 * g);
            continue;
        }
        using thr_t = std::conditional_t<INT, int, float>;
        thr_t thr[I8_NF];
#pragma unroll
        for (int f = 0; f < I8_NF; ++f) {
            if constexpr (INT) thr[f] = -tau[f * 32 + fr];
            else thr[f] = -tau_f[f * 32 + fr];
        }
        unsigned long long fany[I8_NF];
        unsigned long long any = 0;
#pragma unroll
        for (int f = 0; f < I8_NF; ++f) {
            auto m = acc[f][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) {
                if constexpr (INT) m = max(m, acc[f][r]);
                else m = fmaxf(m, acc[f][r]);
            }
            fany[f] = __ballot(m > thr[f]);
            any |= fany[f];
        }
        if (any) {
#pragma unroll
            for (int f = 0; f < I8_NF; ++f) {
                if (fany[f] == 0) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (acc[f][r] > thr[f]) {
                        uint32_t q = (uint32_t)f * 32 + fr;
                        asm volatile("" : "+v"(q));
                        const uint32_t slot = atomicAdd(cnt + q, 1u);
                        const uint32_t id = 32 * g + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4 * fh;
                        uint32_t keyhi;
                        if constexpr (INT) keyhi = (uint32_t)(-acc[f][r]) ^ 0x80000000u;
                        else keyhi = f32_key(-acc[f][r]);
                        if (slot < (uint32_t)I8_CAP) subs[(size_t)q * I8_CAP + slot] = ((uint64_t)keyhi << 32) | id;
                        if (slot == (uint32_t)I8_HW) ctl[0] = 1;
                    }
                }
            }
        }
    }

    if constexpr (MOM) {
        float *slot = mom + (size_t)(b * I8_NW + wave) * (I8_TQ * 2);
#pragma unroll
        for (int f = 0; f < I8_NF; ++f) {
            const float a1 = m1[f] + __shfl_xor(m1[f], 32), a2 = m2[f] + __shfl_xor(m2[f], 32);
            if (lane < 32) {
                slot[(f * 32 + fr) * 2 + 0] = a1;
                slot[(f * 32 + fr) * 2 + 1] = a2;
            }
        }
        if (lane == 0) mom[(size_t)gridDim.x * I8_NW * (I8_TQ * 2) + b * I8_NW + wave] = (float)mrows;
        return;
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) atomicAdd(ctl + 1, 1u);
    for (;;) {
        if (lds_load_u32(ctl) != 0) { round(); continue; }
        if (lds_load_u32(ctl + 1) >= (uint32_t)I8_NW) {
            if (lds_load_u32(ctl) != 0) { round(); continue; }
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();

    {   // stream end (see bf_fp4_stream.hip): lane-per-query copy when nothing needs selecting, else the radix select
        const uint32_t q = (uint32_t)wave + I8_NW * (uint32_t)(lane & 15);  // lanes 0-15 <-> my 16 queries
        const uint32_t qg = qt * I8_TQ + q;
        const bool mine = lane < 16 && qg < nq;
        const uint32_t c = mine ? min(lds_load_u32(cnt + q), (uint32_t)I8_CAP) : 0u;
        const uint32_t nk = mine ? nkept[q] : 0u;
        const bool simple = mine && nk + c <= (uint32_t)KPV;
        if (simple) {
            uint64_t *dst = partial + ((size_t)qg * Stot + list_base + s) * KPV;
            for (uint32_t e = 0; e < nk; ++e) dst[e] = kept[(size_t)q * KPV + e];
            for (uint32_t e = 0; e < c; ++e) dst[nk + e] = subs[(size_t)q * I8_CAP + e];
            // (entries past the count are not written: every reader of a partial list stops at pcnt; the fill was 62 scattered 8-byte stores per lane, ~20 us per launch)
            pcnt[(size_t)qg * Stot + list_base + s] = nk + c;
        }
        unsigned long long todo = __ballot(mine && !simple);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t q2 = (uint32_t)wave + I8_NW * (uint32_t)l;
            const uint32_t qg2 = qt * I8_TQ + q2;
            const uint32_t c2 = min(lds_load_u32(cnt + q2), (uint32_t)I8_CAP);
            if (lane < NSUB) subcnt[q2 * NSUB + lane] = min(c2 - min(c2, (uint32_t)(lane * SUBCAP)), (uint32_t)SUBCAP);
            prune_query<INT, true, KPV>(kept, subs, tau_f, nkept, subcnt + q2 * NSUB, q2, lane, kp, nullptr, sel + wave * sel_scratch_bytes<KPV>());
#pragma unroll
            for (int c2i = 0; c2i < KPV / 64; ++c2i)
                partial[((size_t)qg2 * Stot + list_base + s) * KPV + lane + 64 * c2i] = kept[(size_t)q2 * KPV + lane + 64 * c2i];
            if (lane == 0) pcnt[(size_t)qg2 * Stot + list_base + s] = nkept[q2];
        }
    }
}

// moments -> guessed int threshold (see k_fp4_guess): key domain = int32 stored as key ^ 0x80000000
__global__ __launch_bounds__(64) void k_i8_guess(const float *__restrict__ mom, uint32_t nblocks, uint32_t QT, uint32_t S, uint32_t nq,
                                                 float z, int int_keys, uint32_t *__restrict__ tau_g,
        uint32_t *__restrict__ verify_flag /* cleared here: saves a memset launch */) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && verify_flag) { verify_flag[0] = 0; verify_flag[1] = 0; }  // ([1]: deficient queries of the batch)
    const int lane = threadIdx.x;
    const uint32_t qg = blockIdx.x, qt = qg / I8_TQ, q = qg % I8_TQ;
    double s1 = 0.0, s2 = 0.0, n = 0.0;
    for (uint32_t i = lane; i < S * I8_NW; i += 64) {
        const uint32_t st = i / I8_NW, w = i % I8_NW;
        const uint32_t b = ((st >> 3) * QT + qt) * 8 + (st & 7);
        const float *slot = mom + (size_t)(b * I8_NW + w) * (I8_TQ * 2);
        s1 += slot[q * 2 + 0];
        s2 += slot[q * 2 + 1];
        n += mom[(size_t)nblocks * I8_NW * (I8_TQ * 2) + b * I8_NW + w];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
        n += __shfl_xor(n, off);
    }
    if (lane == 0 && qg < nq) {
        uint32_t out = 0xFFFFFFFFu;
        if (n >= 64.0) {
            const double mu = s1 / n, var = s2 / n - mu * mu;
            if (var > 0.0) {
                const double tk = -mu + (double)z * sqrt(var);
                if (tk == tk && fabs(tk) < 1.0e9) out = int_keys ? ((uint32_t)(int32_t)floor(tk) ^ 0x80000000u) : f32_key((float)tk);
            }
        }
        tau_g[qg] = out;
    }
}

// rows (fp32, row-major) -> int8 fragment order: unit (group g, k-step j, lane l) = the 16 codes of row 32 g + (l & 31),
// elements 32 j + 16 (l >> 5) + [0, 16); the quantiser is the int8 mirror's (data_kernels.hip:k_quantize_rows)
template <typename RT>
__global__ __launch_bounds__(256) void k_i8_pack_frag(const RT *__restrict__ src, uint32_t dim, uint32_t ld, uint64_t n,
                                                      uint64_t units, uint32_t KS, float q_lo, float q_inv, int affine,
                                                      unsigned char *__restrict__ out) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const uint32_t l = (uint32_t)(u & 63);
    const uint64_t gj = u >> 6;
    const uint32_t j = (uint32_t)(gj % KS);
    const uint64_t row = (gj / KS) * 32 + (l & 31);
    const uint32_t k0 = 32 * j + 16 * (l >> 5);
    u32x4 w = {0, 0, 0, 0};
    if (row < n) {
        const RT *v = src + row * ld;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint32_t k = k0 + e;
            int qv = 0;
            if (k < dim) {
                const float x = row_elem(v, k);
                const float f = affine ? __builtin_rintf((x - q_lo) * q_inv) - 128.0f : __builtin_rintf(x * q_inv);
                qv = (int)fminf(fmaxf(f, -128.0f), 127.0f);
            }
            w[e >> 2] |= (uint32_t)(qv & 255) << (8 * (e & 3));
        }
    }
    *reinterpret_cast<u32x4 *>(out + u * 16) = w;
}

template <typename RT>
__global__ __launch_bounds__(256) void k_i8_norms(const RT *__restrict__ rows, uint32_t dim, uint32_t ld, uint64_t n,
                                                  uint64_t n_pad, float q_lo, float q_inv, int affine, int32_t *__restrict__ norm) {
    const int lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_pad) return;
    if (r >= n) {
        if (lane == 0) norm[r] = 0x7FFFFFFF;
        return;
    }
    const RT *v = rows + r * ld;
    int s = 0;
    for (uint32_t c = lane; c < dim; c += 64) {
        const float x = row_elem(v, c);
        const float f = affine ? __builtin_rintf((x - q_lo) * q_inv) - 128.0f : __builtin_rintf(x * q_inv);
        const int qv = (int)fminf(fmaxf(f, -128.0f), 127.0f);
        s += qv * qv;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) norm[r] = s;
}

// fp32 rows (or queries) -> fragment order of MODE 0: unit (group of 32 rows, k-step j, lane l) = the 4 fp32 of row 32 g + (l & 31),
// elements 8 j + 4 (l >> 5) + [0, 4); rows >= n and elements >= dim are 0
template <typename RT>
__global__ __launch_bounds__(256) void k_f32_pack_frag(const RT *__restrict__ src, uint32_t dim, uint32_t ld, uint64_t n, uint64_t units,
                                                       uint32_t KS, unsigned char *__restrict__ out) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const uint32_t l = (uint32_t)(u & 63);
    const uint64_t gj = u >> 6;
    const uint32_t j = (uint32_t)(gj % KS);
    const uint64_t row = (gj / KS) * 32 + (l & 31);
    const uint32_t k0 = 8 * j + 4 * (l >> 5);
    u32x4 w = {0, 0, 0, 0};
    if (row < n) {
        const RT *v = src + row * ld;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k0 + e < dim) w[e] = __float_as_uint(row_elem(v, k0 + e));
    }
    *reinterpret_cast<u32x4 *>(out + u * 16) = w;
}
// the screen's |v|^2 per row (the index's fp32 norms; 0: dot / cosine), +inf past n
__global__ __launch_bounds__(256) void k_f32_frag_norms(const float *__restrict__ norm_in, uint64_t n, uint64_t n_pad, int l2, float *__restrict__ out) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_pad) return;
    out[r] = r >= n ? __builtin_inff() : (l2 ? norm_in[r] : 0.0f);
}

}  // namespace
uint32_t f32s_ksteps(uint32_t dim);
namespace {

template <int MODE, bool MOM, int KPV = KP>
int launch_i8_stream(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                     uint64_t *cand, uint64_t *partial, uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot,
                     uint32_t *pcnt, uint32_t max_tiles, const uint32_t *gate, float *mom, hipStream_t s) {
    const uint32_t KS = MODE == 2 ? i8s_ksteps(h->dim) : MODE == 0 ? f32s_ksteps(h->dim) : fp4s128_ksteps(h->dim);
    const uint32_t lds = i8s_lds_bytes(KS, (uint32_t)sel_scratch_bytes<KPV>());
    const bool l2 = h->metric == ZVDB_METRIC_L2SQ;
    const void *vn = !l2 ? nullptr : MODE == 2 ? (const void *)h->norm_qf : MODE == 0 ? (const void *)h->norm_ff : (const void *)h->norm_pf;
    const unsigned char *rows = MODE == 2 ? h->rows_qf : MODE == 0 ? h->rows_ff : h->rows_pf;
    auto go = [&](auto kern) -> int {
        ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(QT * S), dim3(512), lds, s, q_op, rows, vn, (uint32_t)h->count, nq, QT, S, nt, cand,
                           partial, tau_g, kp, list_base, Stot, pcnt, max_tiles, gate, mom);
        return ZVDB_OK;
    };
    if constexpr (MODE == 2 && !MOM && KPV == KP) {
        if (gate && KS == 24) {  // (the headline shape only: 768 dimensions)
            ZV_CHECK(go(k_i8_stream<2, 24, false, KP, 1>));
            ZV_HIP(hipGetLastError());
            return ZVDB_OK;
        }
    }
    if constexpr (MODE == 2) {
        if (KS == 6) ZV_CHECK(go(k_i8_stream<2, 6, MOM, KPV>));
        else if (KS == 12) ZV_CHECK(go(k_i8_stream<2, 12, MOM, KPV>));
        else ZV_CHECK(go(k_i8_stream<2, 24, MOM, KPV>));
    } else if constexpr (MODE == 0) {
        if (KS == 8) ZV_CHECK(go(k_i8_stream<0, 8, MOM, KPV>));
        else if (KS == 12) ZV_CHECK(go(k_i8_stream<0, 12, MOM, KPV>));
        else ZV_CHECK(go(k_i8_stream<0, 16, MOM, KPV>));
    } else {
        if (KS == 18) ZV_CHECK(go(k_i8_stream<3, 18, MOM, KPV>));
        else ZV_CHECK(go(k_i8_stream<3, 24, MOM, KPV>));
    }
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

}  // namespace

// k-steps of 32 elements; instantiated: 6, 12, 24 (dim <= 192 / 384 / 768)
uint32_t i8s_ksteps(uint32_t dim) {
    const uint32_t k = (dim + 31) / 32;
    return k <= 6 ? 6 : k <= 12 ? 12 : 24;
}

bool i8s_eligible(const zvdb_hip_index *h) {
    const bool off = [] { const char *e = tuning("ZVDB_BF_I8_KERNEL"); return e && e[0] == 'v' && e[1] == '6'; }();
    return !off && h->dim > 0 && h->dim <= 768;
}

uint64_t i8s_query_bytes(const zvdb_hip_index *h, uint32_t nq_pad) { return (uint64_t)nq_pad * i8s_ksteps(h->dim) * 32; }
uint64_t i8s_moment_bytes(uint32_t QT, uint32_t S) { return ((uint64_t)QT * S * I8_NW * (I8_TQ * 2) + (uint64_t)QT * S * I8_NW) * sizeof(float); }

int ensure_i8_frag_mirror(zvdb_hip_index *h, hipStream_t s) {
    if (h->rows_qf && h->count_qf == h->count) return ZVDB_OK;
    if (h->count == 0) return ZVDB_OK;
    const uint32_t KS = i8s_ksteps(h->dim);
    const uint64_t rows_pad = round_up64(h->count, 256);
    if (h->cap_qf < rows_pad) {
        if (h->rows_qf) { ZV_HIP(hipDeviceSynchronize()); /* searches of this index may be in flight on OTHER streams: the mutex covers the host side only */ ZV_HIP(hipFree(h->rows_qf)); ZV_HIP(hipFree(h->norm_qf)); }
        h->rows_qf = nullptr;
        h->norm_qf = nullptr;
        const uint64_t ncap = round_up64(std::max<uint64_t>(h->count, h->cap), 256);
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->rows_qf), ncap * KS * 32));
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->norm_qf), ncap * sizeof(int32_t)));
        h->cap_qf = ncap;
    }
    ZV_CHECK(i8_range(h, s));
    const int affine = h->metric == ZVDB_METRIC_L2SQ ? 1 : 0;
    const uint64_t units = rows_pad / 32 * KS * 64;
    ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = row_elem_t<decltype(rows)>;
        hipLaunchKernelGGL(k_i8_pack_frag<RT>, dim3((uint32_t)((units + 255) / 256)), dim3(256), 0, s, rows, h->dim, ld, h->count, units, KS,
                           h->q_lo, h->q_inv, affine, h->rows_qf);
        hipLaunchKernelGGL(k_i8_norms<RT>, dim3((uint32_t)((rows_pad + 3) / 4)), dim3(256), 0, s, rows, h->dim, ld, h->count, rows_pad, h->q_lo,
                           h->q_inv, affine, h->norm_qf);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    }));
    h->count_qf = h->count;
    return ZVDB_OK;
}

int launch_i8s_prep_queries(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t nq_pad, unsigned char *q_op,
                            hipStream_t s) {
    const uint32_t KS = i8s_ksteps(h->dim);
    const uint64_t units = (uint64_t)nq_pad / 32 * KS * 64;
    hipLaunchKernelGGL(k_i8_pack_frag<float>, dim3((uint32_t)((units + 255) / 256)), dim3(256), 0, s, d_queries, h->dim, h->dim, (uint64_t)nq,
                       units, KS, h->q_lo, h->q_inv, h->metric == ZVDB_METRIC_L2SQ ? 1 : 0, q_op);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// QT = tiles of 128 queries; S = streams (QT * S workgroups)
int launch_i8s_scan(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                    uint64_t *cand, uint64_t *partial, uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot,
                    uint32_t *pcnt, const uint32_t *gate, uint32_t kpv, hipStream_t s) {
    if (kpv == (uint32_t)KPMAX)
        return launch_i8_stream<2, false, KPMAX>(h, q_op, nq, QT, S, nt, cand, partial, tau_g, kp, list_base, Stot, pcnt, 0u, gate, nullptr, s);
    if (kpv == (uint32_t)KPW)
        return launch_i8_stream<2, false, KPW>(h, q_op, nq, QT, S, nt, cand, partial, tau_g, kp, list_base, Stot, pcnt, 0u, gate, nullptr, s);
    return launch_i8_stream<2, false>(h, q_op, nq, QT, S, nt, cand, partial, tau_g, kp, list_base, Stot, pcnt, 0u, gate, nullptr, s);
}

int launch_i8s_guess(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt, float *mom,
                     float z, uint32_t *tau_g, uint32_t *verify_flag, hipStream_t s) {
    ZV_CHECK((launch_i8_stream<2, true>(h, q_op, nq, QT, S, nt, nullptr, nullptr, nullptr, 0u, 0u, 0u, nullptr, 1u, nullptr, mom, s)));
    hipLaunchKernelGGL(k_i8_guess, dim3(QT * I8_TQ), dim3(64), 0, s, mom, QT * S, QT, S, nq, z, 1, tau_g, verify_flag);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// ---- the same 128-query kernel on the fp4 fragment-order mirror: 768 < dim <= 1536 (KS = 18, 24) -------------------------
uint32_t fp4s128_ksteps(uint32_t dim) { return (dim + 63) / 64 <= 18 ? 18 : 24; }
bool fp4s128_eligible(const zvdb_hip_index *h) {
    const bool off = [] { const char *e = tuning("ZVDB_BF_FP4_KERNEL"); return e && e[0] == 'v' && e[1] == '6'; }();
    return !off && h->dim > 768 && h->dim <= 1536;
}
int launch_fp4s128_scan(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                        uint64_t *cand, uint64_t *partial, uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot,
                        uint32_t *pcnt, const uint32_t *gate, uint32_t kpv, hipStream_t s) {
    if (kpv == (uint32_t)KPMAX)
        return launch_i8_stream<3, false, KPMAX>(h, q_op, nq, QT, S, nt, cand, partial, tau_g, kp, list_base, Stot, pcnt, 0u, gate, nullptr, s);
    if (kpv == (uint32_t)KPW)
        return launch_i8_stream<3, false, KPW>(h, q_op, nq, QT, S, nt, cand, partial, tau_g, kp, list_base, Stot, pcnt, 0u, gate, nullptr, s);
    return launch_i8_stream<3, false>(h, q_op, nq, QT, S, nt, cand, partial, tau_g, kp, list_base, Stot, pcnt, 0u, gate, nullptr, s);
}
// ---- the same 128-query kernel with fp32 MFMAs on a fragment-order fp32 mirror: dim <= 128 (KS = 8, 12, 16: <= 64 / 96 / 128 dimensions) --------
// (beyond 128 dimensions -- the register-query scan only, bf_f32_regq.hip -- whole 32-element steps: 20 / 24 / 28 / 32 pieces for <= 160 / 192 / 224 / 256)
uint32_t f32s_ksteps(uint32_t dim) { return dim <= 64 ? 8u : dim <= 96 ? 12u : dim <= 128 ? 16u : (dim + 31) / 32 * 4; }
bool f32s_eligible(const zvdb_hip_index *h) {
    const bool off = [] { const char *e = tuning("ZVDB_BF_F32_KERNEL"); return e && e[0] == 'v' && e[1] == '6'; }();
    return !off && h->dim > 4 && h->dim <= 128;
}
uint64_t f32s_query_bytes(const zvdb_hip_index *h, uint32_t nq_pad) { return (uint64_t)nq_pad * f32s_ksteps(h->dim) * 32; }
int ensure_f32_frag_mirror(zvdb_hip_index *h, hipStream_t s) {
    if (h->rows_ff && h->count_ff == h->count) return ZVDB_OK;
    if (h->count == 0) return ZVDB_OK;
    const uint32_t KS = f32s_ksteps(h->dim);
    const uint64_t rows_pad = round_up64(h->count, 256);
    if (h->cap_ff < rows_pad) {
        if (h->rows_ff) { ZV_HIP(hipDeviceSynchronize()); /* searches of this index may be in flight on OTHER streams: the mutex covers the host side only */ ZV_HIP(hipFree(h->rows_ff)); ZV_HIP(hipFree(h->norm_ff)); }
        h->rows_ff = nullptr;
        h->norm_ff = nullptr;
        const uint64_t ncap = round_up64(std::max<uint64_t>(h->count, h->cap), 256);
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->rows_ff), ncap * KS * 32));
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->norm_ff), ncap * sizeof(float)));
        h->cap_ff = ncap;
    }
    const uint64_t units = rows_pad / 32 * KS * 64;
    const bool l2 = h->metric == ZVDB_METRIC_L2SQ;
    ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        using RT = row_elem_t<decltype(rows)>;
        hipLaunchKernelGGL(k_f32_pack_frag<RT>, dim3((uint32_t)((units + 255) / 256)), dim3(256), 0, s, rows, h->dim, ld, h->count, units, KS, h->rows_ff);
        hipLaunchKernelGGL(k_f32_frag_norms, dim3((uint32_t)((rows_pad + 255) / 256)), dim3(256), 0, s, (h->flags & ZVDB_FLAG_BF16_ROWS) ? h->norm_h : h->norm,
                           h->count, rows_pad, l2 ? 1 : 0, h->norm_ff);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    }));
    h->count_ff = h->count;
    return ZVDB_OK;
}
int launch_f32s_prep_queries(const zvdb_hip_index *h, const float *d_queries, uint32_t nq, uint32_t nq_pad, unsigned char *q_op, hipStream_t s) {
    const uint32_t KS = f32s_ksteps(h->dim);
    const uint64_t units = (uint64_t)nq_pad / 32 * KS * 64;
    hipLaunchKernelGGL(k_f32_pack_frag<float>, dim3((uint32_t)((units + 255) / 256)), dim3(256), 0, s, d_queries, h->dim, h->dim, (uint64_t)nq, units, KS, q_op);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}
int launch_f32s_scan(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                     uint64_t *cand, uint64_t *partial, uint32_t *tau_g, uint32_t kp, uint32_t list_base, uint32_t Stot,
                     uint32_t *pcnt, const uint32_t *gate, uint32_t kpv, hipStream_t s) {
    if (kpv == (uint32_t)KPMAX)
        return launch_i8_stream<0, false, KPMAX>(h, q_op, nq, QT, S, nt, cand, partial, tau_g, kp, list_base, Stot, pcnt, 0u, gate, nullptr, s);
    if (kpv == (uint32_t)KPW)
        return launch_i8_stream<0, false, KPW>(h, q_op, nq, QT, S, nt, cand, partial, tau_g, kp, list_base, Stot, pcnt, 0u, gate, nullptr, s);
    return launch_i8_stream<0, false>(h, q_op, nq, QT, S, nt, cand, partial, tau_g, kp, list_base, Stot, pcnt, 0u, gate, nullptr, s);
}
int launch_f32s_guess(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                      float *mom, float z, uint32_t *tau_g, uint32_t *verify_flag, hipStream_t s) {
    ZV_CHECK((launch_i8_stream<0, true>(h, q_op, nq, QT, S, nt, nullptr, nullptr, nullptr, 0u, 0u, 0u, nullptr, 1u, nullptr, mom, s)));
    hipLaunchKernelGGL(k_i8_guess, dim3(QT * I8_TQ), dim3(64), 0, s, mom, QT * S, QT, S, nq, z, 0, tau_g, verify_flag);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

int launch_fp4s128_guess(const zvdb_hip_index *h, const unsigned char *q_op, uint32_t nq, uint32_t QT, uint32_t S, uint32_t nt,
                         float *mom, float z, uint32_t *tau_g, uint32_t *verify_flag, hipStream_t s) {
    ZV_CHECK((launch_i8_stream<3, true>(h, q_op, nq, QT, S, nt, nullptr, nullptr, nullptr, 0u, 0u, 0u, nullptr, 1u, nullptr, mom, s)));
    hipLaunchKernelGGL(k_i8_guess, dim3(QT * I8_TQ), dim3(64), 0, s, mom, QT * S, QT, S, nq, z, 0, tau_g, verify_flag);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

}  // namespace zvdb
