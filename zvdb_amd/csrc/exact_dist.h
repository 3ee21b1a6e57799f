// exact_dist.h -- the reference's distance arithmetic (hnsw.zig:182-192) as device functions, shared by the translation units
// that must reproduce it bit for bit (exact_kernels.hip, ef_search.hip, rerank.hip, graph_build.hip).  Both are compiled with -ffp-contract=off; the
// pragma below additionally pins it, so hipcc's default -ffp-contract=fast-honor-pragmas cannot fuse mul + add into v_fma.
#pragma once
#include <type_traits>
#include "common.h"
#include "row_types.h"

#pragma clang fp contract(off)

namespace zvdb {

// ---- a1 ---------------------------------------------------------------------
// One lane walks one whole row sequentially (the ONLY order that is bit-exact);
// parallelism comes from different lanes taking different rows.  q may live in
// LDS (broadcast reads) or global memory (uniform address).
#define ZV_A1_STEP(av, vv)                                                            \
    if (METRIC == ZVDB_METRIC_L2SQ) { float t_ = (av) - (vv); t_ = t_ * t_; sum = sum + t_; } \
    else { const float t_ = (av) * (vv); sum = sum + t_; }

template <int METRIC>
__device__ __forceinline__ float exact_dist(const float *__restrict__ q,
                                            const float *__restrict__ row, uint32_t dim) {
    float sum = 0.0f;
    const uint32_t d4 = dim & ~3u;
    uint32_t i = 0;
    // 16 x 16-byte loads in flight per lane: the walk is bound by memory latency (rows are random 3-KB gathers),
    // not by the dependent add chain
#pragma unroll 16
    for (; i < d4; i += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(row + i);
        const float4 a = *reinterpret_cast<const float4 *>(q + i);
        ZV_A1_STEP(a.x, v.x) ZV_A1_STEP(a.y, v.y) ZV_A1_STEP(a.z, v.z) ZV_A1_STEP(a.w, v.w)
    }
    for (; i < dim; ++i) { ZV_A1_STEP(q[i], row[i]) }
    return METRIC == ZVDB_METRIC_L2SQ ? sum : -sum;
}

// bf16 rows: 16-byte loads carry 8 elements
template <int METRIC>
__device__ __forceinline__ float exact_dist(const float *__restrict__ q,
                                            const bf16_t *__restrict__ row, uint32_t dim) {
    float sum = 0.0f;
    const uint32_t d8 = dim & ~7u;
    uint32_t i = 0;
#pragma unroll 16
    for (; i < d8; i += 8) {
        const uint4 w = *reinterpret_cast<const uint4 *>(row + i);
        const float4 a = *reinterpret_cast<const float4 *>(q + i);
        const float4 b = *reinterpret_cast<const float4 *>(q + i + 4);
        ZV_A1_STEP(a.x, bf16_lo(w.x)) ZV_A1_STEP(a.y, bf16_hi(w.x)) ZV_A1_STEP(a.z, bf16_lo(w.y)) ZV_A1_STEP(a.w, bf16_hi(w.y))
        ZV_A1_STEP(b.x, bf16_lo(w.z)) ZV_A1_STEP(b.y, bf16_hi(w.z)) ZV_A1_STEP(b.z, bf16_lo(w.w)) ZV_A1_STEP(b.w, bf16_hi(w.w))
    }
    for (; i < dim; ++i) { ZV_A1_STEP(q[i], row_elem(row, i)) }
    return METRIC == ZVDB_METRIC_L2SQ ? sum : -sum;
}

template <typename RT>
__device__ __forceinline__ float exact_dist_rt(int metric, const float *q, const RT *row, uint32_t dim) {
    return metric == ZVDB_METRIC_L2SQ ? exact_dist<ZVDB_METRIC_L2SQ>(q, row, dim)
                                      : exact_dist<ZVDB_METRIC_DOT>(q, row, dim);
}

__device__ __forceinline__ bool lex_less(float da, uint32_t ia, float db, uint32_t ib) {
    return da < db || (da == db && ia < ib);
}

// Sum over a group of 8 / 16 consecutive lanes with the association of the xor butterfly (8,) 4, 2, 1 AS LANE 0 OF THE GROUP sees
// it: ((s0 + s4) + (s2 + s6)) + ((s1 + s5) + (s3 + s7)) -- only that lane's value is the sum, the others hold partial terms.  DPP
// row shifts inside the 16-lane row (lane i reads lane i + n) instead of __shfl_xor, which hipcc lowers to ds_bpermute: three or
// four dependent LDS-crossbar round trips per row of a distance pass, and the pass is a link of every query's chain (r4).
template <int GROUP>
__device__ __forceinline__ float group_sum_lane0(float v) {
    auto shl = [](float x, auto ctrl) {
        return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    if constexpr (GROUP == 16) v = v + shl(v, std::integral_constant<int, 0x108>{});  // row_shl:8
    v = v + shl(v, std::integral_constant<int, 0x104>{});                            // row_shl:4
    v = v + shl(v, std::integral_constant<int, 0x102>{});                            // row_shl:2
    v = v + shl(v, std::integral_constant<int, 0x101>{});                            // row_shl:1
    return v;
}

// 16-way BLOCKED traversal distance of the ef-mode beam search and the entry descent (specification:
// orc_distance_blocked16_f32 / orc_neg_dot_blocked16_f32): lane t of a 16-lane group takes the 4-element chunks t, t + 16, ...
// of the row, accumulates them sequentially, un-contracted; the 16 partials are combined by the xor butterfly 8, 4, 2, 1.
template <int METRIC, typename RT>
__device__ __forceinline__ float blocked16_dist(const float *__restrict__ q /*LDS, zero padded to x4*/,
                                                const RT *__restrict__ row, uint32_t nchunks, int t) {
    float sum = 0.0f;
    for (uint32_t c = t; c < nchunks; c += 16) {
        const float4 v = row_load4(row + 4 * c);  // rows are zero padded to ld
        const float4 a = *reinterpret_cast<const float4 *>(q + 4 * c);
        ZV_A1_STEP(a.x, v.x) ZV_A1_STEP(a.y, v.y) ZV_A1_STEP(a.z, v.z) ZV_A1_STEP(a.w, v.w)
    }
    sum = group_sum_lane0<16>(sum);
    sum = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sum)));  // (every group walked the same row: the wave's lane 0 speaks for all)
    return METRIC == ZVDB_METRIC_L2SQ ? sum : -sum;
}

}  // namespace zvdb
