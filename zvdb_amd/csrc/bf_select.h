// bf_select.h -- selection state shared by the brute-force scan kernels (bruteforce.hip: LDS-staged 256 x 256
// tiles; bf_fp4_stream.hip: barrier-free fp4 stream): per-(workgroup, query) kept list + append buffers in global
// memory, merged by an exact wave radix select.
#pragma once
#include "common.h"
#include "wave_select.h"

namespace zvdb {

constexpr int KP = 64;   // k' : candidates kept per (query, stream) and re-ranked
constexpr int T2 = 256;  // queries per workgroup / corpus rows per tile
constexpr int SUBCAP = 256, SUBHW = 48, SUBURG = 192, NSUB = 4;  // capacity, soft mark, must-prune mark
constexpr size_t BLK4_KEPT = (size_t)T2 * KP;                 // u64 entries
constexpr size_t BLK4_ENTRIES = BLK4_KEPT + (size_t)T2 * NSUB * SUBCAP;

constexpr int KPW = 128;    // wide kept list (2 entries per lane): k <= 112, every route
constexpr int KPMAX = 256;  // widest kept list (4 entries per lane): k <= 240 on the stream pipelines
constexpr size_t BLKV_ENTRIES = (size_t)T2 * KPMAX + (size_t)T2 * NSUB * SUBCAP;  // per-workgroup region sized for KPMAX
// per-wave scratch of prune_query: radix histogram + compaction stage of KPV keys
template <int KPV> constexpr int sel_scratch_bytes() { return SEL_HIST_BYTES + (KPV * 8 > SEL_STAGE_BYTES ? KPV * 8 : SEL_STAGE_BYTES); }

// merge query q's kept list with its sub-buffers -> exact best kp (UNSORTED, compacted to the front of
// kept[q]); resets the sub-buffers.  Rounds of <= KPV carried + 256 new keys, radix-selected.
// INCL: tau becomes next_up(k'-th key), i.e. later keys EQUAL to the k'-th best still pass (needed when rows do not
// arrive in id order).  KPV = capacity (and row stride) of the kept lists: 64 (one entry per lane), 128 (two) or 256
// (four; scratch must then hold sel_scratch_bytes<KPV>()).
// Returns the lane's first kept entry (KPV = 64: the whole list, one per lane).
template <bool INTKEY, bool INCL = false, int KPV = KP>
__device__ __forceinline__ uint64_t prune_query(uint64_t *__restrict__ kept, const uint64_t *__restrict__ sub,
                                                float *tau, uint32_t *nkept, uint32_t *subcnt /* the query's NSUB fill counters */, int q, int lane,
                                                uint32_t kp, uint32_t *tau_g, unsigned char *scratch) {
    constexpr int EPL = KPV / kWave;  // kept entries per lane
    static_assert(KPV == 64 || KPV == 128 || KPV == 256, "kept list width");
    uint32_t *hist = reinterpret_cast<uint32_t *>(scratch);
    uint64_t *stage = reinterpret_cast<uint64_t *>(scratch + SEL_HIST_BYTES);  // [KPV]
    const uint32_t n0 = min(subcnt[0], (uint32_t)SUBCAP), n1 = min(subcnt[1], (uint32_t)SUBCAP);
    const uint32_t n2 = min(subcnt[2], (uint32_t)SUBCAP), n3 = min(subcnt[3], (uint32_t)SUBCAP);
    const uint32_t p1 = n0, p2 = n0 + n1, p3 = p2 + n2, total = p3 + n3;
    uint32_t ncarry = nkept[q];
    uint64_t carry[EPL];
#pragma unroll
    for (int c = 0; c < EPL; ++c) carry[c] = (uint32_t)(lane + kWave * c) < ncarry ? kept[(size_t)q * KPV + lane + kWave * c] : ~0ull;
    uint32_t done = 0;
    while (done < total) {
        uint64_t k[EPL + 4];
#pragma unroll
        for (int c = 0; c < EPL; ++c) k[c] = carry[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t i = done + e * kWave + lane;
            uint64_t v = ~0ull;
            if (i < total) {
                const uint32_t si = (i >= p1) + (i >= p2) + (i >= p3);
                const uint32_t off = i - (si == 0 ? 0u : si == 1 ? p1 : si == 2 ? p2 : p3);
                v = sub[((size_t)q * NSUB + si) * SUBCAP + off];
            }
            k[EPL + e] = v;
        }
        const uint32_t nnew = min(256u, total - done);
        done += nnew;
        const uint32_t nvalid = ncarry + nnew;
        uint64_t T = ~0ull - 1;  // everything valid passes
        if (nvalid > kp) T = wave_kth_key<EPL + 4>([&](uint32_t e) { return k[e]; }, EPL + 4, kp, hist, lane);
        uint32_t base = 0;
#pragma unroll
        for (int e = 0; e < EPL + 4; ++e) {
            const bool selv = k[e] <= T;
            const unsigned long long m = __ballot(selv);
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
            if (selv && pos < (uint32_t)KPV) stage[pos] = k[e];
            base += (uint32_t)__popcll(m);
        }
        ncarry = min(nvalid, kp);
#pragma unroll
        for (int c = 0; c < EPL; ++c) carry[c] = (uint32_t)(lane + kWave * c) < ncarry ? stage[lane + kWave * c] : ~0ull;
    }
    uint64_t mx = 0;
#pragma unroll
    for (int c = 0; c < EPL; ++c) {
        kept[(size_t)q * KPV + lane + kWave * c] = carry[c];
        if ((uint32_t)(lane + kWave * c) < ncarry) mx = umax64(mx, carry[c]);
    }
    const uint64_t last = ncarry == kp ? wave_max64(mx) : ~0ull;
    if (lane == 0) {
        nkept[q] = ncarry;
        subcnt[0] = 0; subcnt[1] = 0; subcnt[2] = 0; subcnt[3] = 0;
        if (ncarry == kp) {
            if (INTKEY) {
                int32_t *ti = reinterpret_cast<int32_t *>(tau);
                int32_t nt = (int32_t)((uint32_t)(last >> 32) ^ 0x80000000u);
                if (INCL && nt != 0x7FFFFFFF) nt += 1;
                if (nt < ti[q]) ti[q] = nt;
            } else {
                const float nt = f32_unkey((uint32_t)(last >> 32) + (INCL ? 1u : 0u));
                if (nt < tau[q]) tau[q] = nt;
            }
            if (tau_g) atomicMin(tau_g + q, (uint32_t)(last >> 32));
        }
    }
    return carry[0];
}


}  // namespace zvdb
