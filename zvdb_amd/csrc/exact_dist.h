// exact_dist.h -- the reference's distance arithmetic (hnsw.zig:182-192) as device functions, shared by the translation units
// that must reproduce it bit for bit (exact_kernels.hip, graph_build.hip).  Both are compiled with -ffp-contract=off; the
// pragma below additionally pins it, so hipcc's default -ffp-contract=fast-honor-pragmas cannot fuse mul + add into v_fma.
#pragma once
#include "common.h"

#pragma clang fp contract(off)

namespace zvdb {

// ---- a1 ---------------------------------------------------------------------
// One lane walks one whole row sequentially (the ONLY order that is bit-exact);
// parallelism comes from different lanes taking different rows.  q may live in
// LDS (broadcast reads) or global memory (uniform address).
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
        if (METRIC == ZVDB_METRIC_L2SQ) {
            float t;
            t = a.x - v.x; t = t * t; sum = sum + t;
            t = a.y - v.y; t = t * t; sum = sum + t;
            t = a.z - v.z; t = t * t; sum = sum + t;
            t = a.w - v.w; t = t * t; sum = sum + t;
        } else {
            float t;
            t = a.x * v.x; sum = sum + t;
            t = a.y * v.y; sum = sum + t;
            t = a.z * v.z; sum = sum + t;
            t = a.w * v.w; sum = sum + t;
        }
    }
    for (; i < dim; ++i) {
        if (METRIC == ZVDB_METRIC_L2SQ) {
            float t = q[i] - row[i];
            t = t * t;
            sum = sum + t;
        } else {
            const float t = q[i] * row[i];
            sum = sum + t;
        }
    }
    return METRIC == ZVDB_METRIC_L2SQ ? sum : -sum;
}

__device__ __forceinline__ float exact_dist_rt(int metric, const float *q, const float *row,
                                               uint32_t dim) {
    return metric == ZVDB_METRIC_L2SQ ? exact_dist<ZVDB_METRIC_L2SQ>(q, row, dim)
                                      : exact_dist<ZVDB_METRIC_DOT>(q, row, dim);
}

__device__ __forceinline__ bool lex_less(float da, uint32_t ia, float db, uint32_t ib) {
    return da < db || (da == db && ia < ib);
}

// 16-way BLOCKED traversal distance of the ef-mode beam search and the entry descent (specification:
// orc_distance_blocked16_f32 / orc_neg_dot_blocked16_f32): lane t of a 16-lane group takes the 4-float chunks t, t + 16, ...
// of the row, accumulates them sequentially, un-contracted; the 16 partials are combined by the xor butterfly 8, 4, 2, 1.
template <int METRIC>
__device__ __forceinline__ float blocked16_dist(const float *__restrict__ q /*LDS, zero padded to x4*/,
                                                const float *__restrict__ row, uint32_t nchunks, int t) {
    float s = 0.0f;
    for (uint32_t c = t; c < nchunks; c += 16) {
        const float4 v = *reinterpret_cast<const float4 *>(row + 4 * c);  // rows are zero padded to ld
        const float4 a = *reinterpret_cast<const float4 *>(q + 4 * c);
        float u;
        if (METRIC == ZVDB_METRIC_L2SQ) {
            u = a.x - v.x; u = u * u; s = s + u;
            u = a.y - v.y; u = u * u; s = s + u;
            u = a.z - v.z; u = u * u; s = s + u;
            u = a.w - v.w; u = u * u; s = s + u;
        } else {
            u = a.x * v.x; s = s + u;
            u = a.y * v.y; s = s + u;
            u = a.z * v.z; s = s + u;
            u = a.w * v.w; s = s + u;
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s = s + __shfl_xor(s, off);
    return METRIC == ZVDB_METRIC_L2SQ ? s : -s;
}

}  // namespace zvdb
