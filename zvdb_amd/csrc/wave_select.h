// wave_select.h -- wave64 selection primitives shared by the brute-force scan (prune / merge) and the
// ef-mode graph search (beam maintenance): exact k-th smallest of 64-bit (key,id) words by MSD radix
// select with a wave-private LDS histogram, DPP prefix scan, 64-bit shuffles / reductions.
#pragma once
#include "common.h"

namespace zvdb {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
    return (uint64_t)__shfl((unsigned long long)v, src);
}
__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int m) {
    return (uint64_t)__shfl_xor((unsigned long long)v, m);
}
__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b) { return a < b ? b : a; }

// =====================================================================================
// Exact selection without sorting: MSD radix select (8 bits per pass) on 64-bit (key,id) words
// held E per lane, using a wave-private 256-bin LDS histogram.  Distances of a candidate set
// share their leading bits, so the walk typically ends after 3-4 passes (early exit as soon as
// the chosen bin holds one candidate).  Replaces the bitonic sort/merge networks (21 + 7
// dependent ds_bpermute stages per 64 keys) in the prune rounds and the list merges.
// =====================================================================================
constexpr int SEL_HIST_BYTES = 1024, SEL_STAGE_BYTES = 1024;  // stage: up to 128 keys (the widest kept list)
constexpr int SEL_SCRATCH = SEL_HIST_BYTES + SEL_STAGE_BYTES;  // per wave

// inclusive prefix sum across the wave on DPP (row_shr 1/2/4/8 + row_bcast15/31): 6 VALU ops, no LDS
// crossbar round trips (the __shfl_up form costs six dependent ds_bpermute)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int /*lane*/) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

// kth (1-based) smallest of the valid (!= ~0) keys; precondition: kth <= number of valid keys.
// KeyAt(e) returns the lane's e-th key, e < E (E may be a runtime count for LDS-resident sets).
// EC > 0: compile-time element count (register-resident keys, fully unrolled); EC == 0: runtime E.
template <int EC, typename KeyAt>
__device__ __forceinline__ uint64_t wave_kth_key(KeyAt key_at, uint32_t E_rt, uint32_t kth, uint32_t *hist, int lane) {
    const uint32_t E = EC > 0 ? (uint32_t)EC : E_rt;
    uint64_t prefix = 0;
    uint32_t need = kth;
    for (int shift = 56; shift >= 0; shift -= 8) {
        *reinterpret_cast<u32x4 *>(hist + 4 * lane) = u32x4{0, 0, 0, 0};
#pragma unroll(EC > 0 ? EC : 1)
        for (uint32_t e = 0; e < E; ++e) {
            const uint64_t k = key_at(e);
            if (k != ~0ull && (shift == 56 || (k >> (shift + 8)) == prefix))
                atomicAdd(&hist[(uint32_t)(k >> shift) & 255u], 1u);
        }
        const u32x4 c = *reinterpret_cast<const u32x4 *>(hist + 4 * lane);
        const uint32_t sl = c.x + c.y + c.z + c.w;
        const uint32_t inc = wave_incl_scan(sl, lane);
        const unsigned long long b = __ballot(inc >= need);
        const int L = __builtin_amdgcn_readfirstlane(__builtin_ctzll(b));  // wave-uniform -> v_readlane below
        uint32_t r = need - (inc - sl);  // rank inside this lane's four bins (meaningful in lane L)
        uint32_t j = 0, cj = c.x;
        if (r > c.x) { r -= c.x; j = 1; cj = c.y; if (r > c.y) { r -= c.y; j = 2; cj = c.z; if (r > c.z) { r -= c.z; j = 3; cj = c.w; } } }
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)(4 * lane + j), L);
        need = (uint32_t)__builtin_amdgcn_readlane((int)r, L);
        const uint32_t cnt_d = (uint32_t)__builtin_amdgcn_readlane((int)cj, L);
        prefix = (prefix << 8) | d;
        if (cnt_d == 1 && shift > 0) {  // a single candidate left on the path: it is the answer
            uint64_t found = ~0ull;
#pragma unroll(EC > 0 ? EC : 1)
            for (uint32_t e = 0; e < E; ++e) {
                const uint64_t k = key_at(e);
                if (k != ~0ull && (k >> shift) == prefix) found = k;
            }
            const unsigned long long fb = __ballot(found != ~0ull);
            const int FL = __builtin_amdgcn_readfirstlane(__builtin_ctzll(fb));
            return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(found >> 32), FL) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)found, FL);
        }
    }
    return prefix;
}

// 64-bit wave reductions on DPP: four row_shr steps leave each 16-lane row's result in its last lane, four readlanes
// combine the rows on the scalar unit.  (The __shfl_xor form costs twelve dependent ds_bpermute round trips: ~1,200 cycles
// per reduction, measured as a tenth of a beam-search pop.)  MINR: minimum (identity ~0) / maximum (identity 0).
template <bool MINR>
__device__ __forceinline__ uint64_t wave_reduce64(uint64_t v) {
    const uint32_t idn = MINR ? 0xFFFFFFFFu : 0u;
#define ZV_RED_STEP(ctrl)                                                                                  \
    {                                                                                                      \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)idn, (int)(uint32_t)v, ctrl, 0xf, 0xf, false);         \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)idn, (int)(uint32_t)(v >> 32), ctrl, 0xf, 0xf, false);  \
        const uint64_t o = ((uint64_t)hi << 32) | lo;                                                      \
        v = MINR ? umin64(v, o) : umax64(v, o);                                                            \
    }
    ZV_RED_STEP(0x111) ZV_RED_STEP(0x112) ZV_RED_STEP(0x114) ZV_RED_STEP(0x118)  // row_shr:1, 2, 4, 8
#undef ZV_RED_STEP
    uint64_t r = MINR ? ~0ull : 0ull;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const uint64_t x = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 16 * row + 15) << 32) |
                           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 16 * row + 15);
        r = MINR ? umin64(r, x) : umax64(r, x);
    }
    return r;
}
__device__ __forceinline__ uint64_t wave_max64(uint64_t v) { return wave_reduce64<false>(v); }
__device__ __forceinline__ uint64_t wave_min64(uint64_t v) { return wave_reduce64<true>(v); }

// Order key of a distance for a wave arg-min over (distance, position) packed as key << 32 | position: ascending like the float
// compare of the reference (hnsw.zig:96 `d < best`), -0 = +0 (so equal distances fall to the lower position), a NaN after everything.
__device__ __forceinline__ uint32_t dist_order_key(float d) { return d != d ? 0xFFFFFFFFu : f32_key(d + 0.0f); }
// Wave arg-min over (d, j) of the active lanes (j distinct per lane), ties to the lower j -- the reference's strict `<` walked in
// list order.  mj = 0xFFFFFFFF when no lane is active; md / mid = the winning lane's own d and id.  One DPP reduction, a ballot and
// two readlanes: the __shfl_xor butterfly over three values it replaces (r4) was 18 dependent ds_bpermute round trips.
__device__ __forceinline__ void wave_argmin_dj(bool act, float d, uint32_t j, uint32_t id, float &md, uint32_t &mj, uint32_t &mid) {
    const uint64_t best = wave_min64(act ? (((uint64_t)dist_order_key(d) << 32) | j) : ~0ull);
    mj = (uint32_t)best;
    md = __builtin_inff();
    mid = id;
    if (best == ~0ull) { mj = 0xFFFFFFFFu; return; }  // (wave-uniform)
    const int wl = __builtin_ctzll(__ballot(act && j == mj));
    md = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d), wl));
    mid = (uint32_t)__builtin_amdgcn_readlane((int)id, wl);
}

}  // namespace zvdb
