// data_kernels.hip -- row ingestion (pack / synthetic / normalise / norms + bf16 mirror)
// and the k-way merge that consumes the RCCL all-gather of per-shard top-k lists.
#include "common.h"
#include "row_types.h"

#include <cmath>

namespace zvdb {

// Synthetic U[0,1) rows (distribution of benchmarks/shared_benchmarks.zig:53-59; the
// stream definition is shared with oracle/zvdb_oracle.c:orc_fill_uniform).
__global__ __launch_bounds__(256) void k_fill_synthetic(float *__restrict__ out, uint64_t nrows,
                                                        uint32_t dim, uint32_t ld, uint64_t seed,
                                                        uint64_t first_row) {
    const uint32_t ld4 = ld >> 2;
    const uint64_t total = nrows * ld4;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = t / ld4;
        const uint32_t c = (uint32_t)(t - r * ld4) * 4;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t cc = c + j;
            if (cc < dim) {
                const uint64_t z = mix64(seed, (first_row + r) * dim + cc);
                v[j] = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
            } else {
                v[j] = 0.0f;
            }
        }
        *reinterpret_cast<float4 *>(out + r * ld + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

int launch_fill_synthetic(float *d_out, uint64_t nrows, uint32_t dim, uint32_t ld, uint64_t seed,
                          uint64_t first_row, hipStream_t s) {
    if (nrows == 0) return ZVDB_OK;
    const uint64_t total = nrows * (ld / 4);
    const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_fill_synthetic, dim3(grid), dim3(256), 0, s, d_out, nrows, dim, ld, seed,
                       first_row);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// dense [n][dim] -> padded [n][ld]
__global__ __launch_bounds__(256) void k_pack_rows(const float *__restrict__ src, uint32_t dim,
                                                   float *__restrict__ dst, uint32_t ld, uint64_t n) {
    const uint64_t total = n * ld;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = t / ld;
        const uint32_t c = (uint32_t)(t - r * ld);
        dst[t] = c < dim ? src[r * dim + c] : 0.0f;
    }
}

int launch_pack_rows(const float *d_src, uint32_t dim, float *d_dst, uint32_t ld, uint64_t n,
                     hipStream_t s) {
    if (n == 0) return ZVDB_OK;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n * ld + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_pack_rows, dim3(grid), dim3(256), 0, s, d_src, dim, d_dst, ld, n);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// cosine metric: v <- v / sqrt(sum v^2) with the sum taken strictly left to right in fp32
// (one lane per row, so the arithmetic is reproducible on the CPU).
__global__ __launch_bounds__(256) void k_normalize_rows(float *__restrict__ rows, uint32_t dim,
                                                        uint32_t ld, uint64_t n) {
#pragma clang fp contract(off)
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    float *v = rows + r * ld;
    float s = 0.0f;
    for (uint32_t i = 0; i < dim; ++i) {
        const float t = v[i] * v[i];
        s = s + t;
    }
    const float nrm = __fsqrt_rn(s);
    if (nrm > 0.0f)
        for (uint32_t i = 0; i < dim; ++i) v[i] = __fdiv_rn(v[i], nrm);
}

int launch_normalize_rows(float *d_rows, uint32_t dim, uint32_t ld, uint64_t n, hipStream_t s) {
    if (n == 0) return ZVDB_OK;
    hipLaunchKernelGGL(k_normalize_rows, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, d_rows,
                       dim, ld, n);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)((u >> 16) | ((u & 0x007FFFFFu) ? 0x40u : 0u));
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// per row: norm = sum v^2 (fp32), bf16 mirror row + its own norm.  One wave per row.
__global__ __launch_bounds__(256) void k_finalize_rows(const float *__restrict__ rows, uint32_t dim,
                                                       uint32_t ld, uint16_t *__restrict__ rows_h,
                                                       uint32_t ldh, float *__restrict__ norm,
                                                       float *__restrict__ norm_h, uint64_t first,
                                                       uint64_t n) {
    const int lane = threadIdx.x & 63;
    const uint64_t r = first + (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= first + n) return;
    const float *v = rows + r * ld;
    float s = 0.0f, sh = 0.0f;
    for (uint32_t c = lane * 4; c < ld; c += 256) {
        const float4 x = *reinterpret_cast<const float4 *>(v + c);
        s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        if (rows_h) {
            const uint16_t b0 = f32_to_bf16_rne(x.x), b1 = f32_to_bf16_rne(x.y);
            const uint16_t b2 = f32_to_bf16_rne(x.z), b3 = f32_to_bf16_rne(x.w);
            const float f0 = __uint_as_float((uint32_t)b0 << 16), f1 = __uint_as_float((uint32_t)b1 << 16);
            const float f2 = __uint_as_float((uint32_t)b2 << 16), f3 = __uint_as_float((uint32_t)b3 << 16);
            sh += f0 * f0 + f1 * f1 + f2 * f2 + f3 * f3;
            uint2 pk;
            pk.x = (uint32_t)b0 | ((uint32_t)b1 << 16);
            pk.y = (uint32_t)b2 | ((uint32_t)b3 << 16);
            *reinterpret_cast<uint2 *>(rows_h + r * ldh + c) = pk;
        }
    }
    if (rows_h)
        for (uint32_t c = ld + lane * 4; c < ldh; c += 256)
            *reinterpret_cast<uint2 *>(rows_h + r * ldh + c) = make_uint2(0, 0);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        sh += __shfl_xor(sh, off);
    }
    if (lane == 0) {
        norm[r] = s;
        if (norm_h) norm_h[r] = sh;
    }
}

// ZVDB_FLAG_BF16_ROWS: rows staged as fp32 [n][ld] (already normalised under cosine) -> the index's bf16 row storage
// rows_h[first + r] (RNE, zero padded to ldh) and the rows' squared norms over the stored values.  One wave per row.
__global__ __launch_bounds__(256) void k_store_rows_bf16(const float *__restrict__ stage, uint32_t dim, uint32_t ld, uint16_t *__restrict__ rows_h,
                                                         uint32_t ldh, float *__restrict__ norm, float *__restrict__ norm_h, uint64_t first,
                                                         uint64_t n) {
    const int lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float *v = stage + r * ld;
    uint16_t *o = rows_h + (first + r) * ldh;
    float sh = 0.0f;
    for (uint32_t c = lane * 4; c < ldh; c += 256) {
        uint16_t b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b[j] = (c + j < dim) ? f32_to_bf16_rne(v[c + j]) : (uint16_t)0;
            const float f = __uint_as_float((uint32_t)b[j] << 16);
            sh += f * f;
        }
        uint2 pk;
        pk.x = (uint32_t)b[0] | ((uint32_t)b[1] << 16);
        pk.y = (uint32_t)b[2] | ((uint32_t)b[3] << 16);
        *reinterpret_cast<uint2 *>(o + c) = pk;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sh += __shfl_xor(sh, off);
    if (lane == 0) {
        norm[first + r] = sh;
        norm_h[first + r] = sh;
    }
}

int launch_store_rows_bf16(const zvdb_hip_index *h, const float *d_stage, uint64_t first, uint64_t n, hipStream_t s) {
    if (n == 0) return ZVDB_OK;
    hipLaunchKernelGGL(k_store_rows_bf16, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, s, d_stage, h->dim, h->ld, h->rows_h, h->ldh, h->norm,
                       h->norm_h, first, n);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// Rows that did not come through the add paths (zvdb_hip_load copies them out of a file): elements [dim, ld) are zeroed -- the
// exact kernels walk whole chunks up to ld and rely on the padding adding +0.0 (rerank.hip: k_rerank_staged) -- and, for
// bf16 row storage, the norms are recomputed from the rows the way k_store_rows_bf16 sums them instead of being taken on trust.
__global__ __launch_bounds__(256) void k_sanitize_rows_f32(float *__restrict__ rows, uint32_t dim, uint32_t ld, uint64_t n) {
    const uint32_t pad = ld - dim;
    const uint64_t total = n * pad;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x)
        rows[(t / pad) * ld + dim + (uint32_t)(t % pad)] = 0.0f;
}
__global__ __launch_bounds__(256) void k_sanitize_rows_bf16(uint16_t *__restrict__ rows_h, uint32_t dim, uint32_t ldh, float *__restrict__ norm,
                                                            float *__restrict__ norm_h, uint64_t n) {
    const int lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    uint16_t *o = rows_h + r * ldh;
    float sh = 0.0f;
    for (uint32_t c = lane * 4; c < ldh; c += 256) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (c + j >= dim) o[c + j] = 0;
            const float f = c + j < dim ? __uint_as_float((uint32_t)o[c + j] << 16) : 0.0f;
            sh += f * f;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sh += __shfl_xor(sh, off);
    if (lane == 0) {
        norm[r] = sh;
        norm_h[r] = sh;
    }
}
int launch_sanitize_loaded_rows(const zvdb_hip_index *h, uint64_t n, hipStream_t s) {
    if (n == 0) return ZVDB_OK;
    if (h->flags & ZVDB_FLAG_BF16_ROWS) {
        hipLaunchKernelGGL(k_sanitize_rows_bf16, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, s, h->rows_h, h->dim, h->ldh, h->norm, h->norm_h, n);
    } else if (h->ld > h->dim) {
        const uint64_t total = n * (h->ld - h->dim);
        hipLaunchKernelGGL(k_sanitize_rows_f32, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, 256 * 16)), dim3(256), 0, s, h->rows, h->dim, h->ld, n);
    }
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

int launch_finalize_rows(const zvdb_hip_index *h, uint64_t first, uint64_t n, hipStream_t s) {
    if (n == 0) return ZVDB_OK;
    hipLaunchKernelGGL(k_finalize_rows, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, s, h->rows, h->dim,
                       h->ld, h->rows_h, h->ldh, h->norm, h->norm_h, first, n);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}


// ---- int8 scalar-quantised mirror (ZVDB_BF_I8) -------------------------------------------------
// One global range for the whole shard: L2 uses the affine map q = rint((x - lo) * 255/(hi-lo)) - 128
// (offsets cancel in q_a - q_b); dot/cosine use the symmetric map q = rint(x * 127/max|x|).
template <typename RT>
__global__ __launch_bounds__(256) void k_minmax(const RT *__restrict__ rows, uint32_t dim, uint32_t ld,
                                                uint64_t n, uint32_t *__restrict__ range) {
    float lo = __builtin_inff(), hi = -__builtin_inff();
    const uint64_t total = n * ld;
    for (uint64_t t = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; t < total;
         t += (uint64_t)gridDim.x * blockDim.x * 4) {
        const uint32_t c = (uint32_t)(t % ld);
        const float4 v = row_load4(rows + t);
        if (c + 0 < dim) { lo = fminf(lo, v.x); hi = fmaxf(hi, v.x); }
        if (c + 1 < dim) { lo = fminf(lo, v.y); hi = fmaxf(hi, v.y); }
        if (c + 2 < dim) { lo = fminf(lo, v.z); hi = fmaxf(hi, v.z); }
        if (c + 3 < dim) { lo = fminf(lo, v.w); hi = fmaxf(hi, v.w); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&range[0], f32_key(lo));
        atomicMax(&range[1], f32_key(hi));
    }
}

template <typename RT>
__global__ __launch_bounds__(256) void k_quantize_rows(const RT *__restrict__ rows, uint32_t dim,
                                                       uint32_t ld, int8_t *__restrict__ rows_q,
                                                       uint32_t ldq, int32_t *__restrict__ norm_q,
                                                       uint64_t n, float q_lo, float q_inv, int affine) {
    const int lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const RT *v = rows + r * ld;
    int s = 0;
    for (uint32_t c = lane * 4; c < ldq; c += 256) {
        int q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int qv = 0;
            if (c + j < dim) {
                const float x = row_elem(v, c + j);
                const float f = affine ? __builtin_rintf((x - q_lo) * q_inv) - 128.0f : __builtin_rintf(x * q_inv);
                qv = (int)fminf(fmaxf(f, -128.0f), 127.0f);
            }
            q[j] = qv;
            s += qv * qv;
        }
        *reinterpret_cast<uint32_t *>(rows_q + r * ldq + c) =
            (uint32_t)(q[0] & 255) | ((uint32_t)(q[1] & 255) << 8) | ((uint32_t)(q[2] & 255) << 16) |
            ((uint32_t)(q[3] & 255) << 24);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) norm_q[r] = s;
}

int i8_range(zvdb_hip_index *h, hipStream_t s) {
    if (!h->q_range) ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->q_range), 8));
    const uint32_t init[2] = {0xFFFFFFFFu, 0u};
    ZV_HIP(hipMemcpyAsync(h->q_range, init, 8, hipMemcpyHostToDevice, s));
    ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        const uint64_t total4 = h->count * ld / 4;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((total4 + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(k_minmax<row_elem_t<decltype(rows)>>, dim3(grid), dim3(256), 0, s, rows, h->dim, ld, h->count, h->q_range);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    }));
    uint32_t got[2];
    ZV_HIP(hipMemcpyAsync(got, h->q_range, 8, hipMemcpyDeviceToHost, s));
    ZV_HIP(hipStreamSynchronize(s));  // (lazy one-off build; not on the steady-state search path)
    const float lo = f32_unkey(got[0]), hi = f32_unkey(got[1]);
    const bool affine = h->metric == ZVDB_METRIC_L2SQ;
    if (affine) {
        h->q_lo = lo;
        h->q_inv = hi > lo ? 255.0f / (hi - lo) : 1.0f;
    } else {
        const float m = std::max(std::fabs(lo), std::fabs(hi));
        h->q_lo = 0.0f;
        h->q_inv = m > 0.0f ? 127.0f / m : 1.0f;
    }
    return ZVDB_OK;
}

int ensure_i8_mirror(zvdb_hip_index *h, hipStream_t s) {
    if (h->rows_q && h->count_q == h->count) return ZVDB_OK;
    if (h->count == 0) return ZVDB_OK;
    h->ldq = round_up(h->dim, 128);
    if (h->cap_q < h->count) {
        if (h->rows_q) { ZV_HIP(hipDeviceSynchronize()); /* searches of this index may be in flight on OTHER streams: the mutex covers the host side only */ ZV_HIP(hipFree(h->rows_q)); ZV_HIP(hipFree(h->norm_q)); }
        h->rows_q = nullptr;
        h->norm_q = nullptr;
        const uint64_t ncap = std::max<uint64_t>(h->count, h->cap);
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->rows_q), ncap * h->ldq));
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->norm_q), ncap * sizeof(int32_t)));
        h->cap_q = ncap;
    }
    ZV_CHECK(i8_range(h, s));
    const bool affine = h->metric == ZVDB_METRIC_L2SQ;
    ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        hipLaunchKernelGGL(k_quantize_rows<row_elem_t<decltype(rows)>>, dim3((uint32_t)((h->count + 3) / 4)), dim3(256), 0, s, rows, h->dim, ld,
                           h->rows_q, h->ldq, h->norm_q, h->count, h->q_lo, h->q_inv, affine ? 1 : 0);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    }));
    h->count_q = h->count;
    return ZVDB_OK;
}

// ---- fp4 (E2M1) scalar-quantised mirror (ZVDB_BF_FP4) -------------------------------------------
// Same global range as the int8 mirror, mapped onto [-kFp4Amax, kFp4Amax] and rounded to the nearest fp4
// value: L2 centres the range (offsets cancel in q_a - q_b), dot/cosine scale symmetrically around 0.
// One wave per row; a lane packs 8 consecutive elements into one dword (low nibble = even element).
template <typename RT>
__global__ __launch_bounds__(256) void k_quantize_rows_fp4(const RT *__restrict__ rows, uint32_t dim, uint32_t ld,
                                                           unsigned char *__restrict__ rows_p, uint32_t ldp,
                                                           float *__restrict__ norm_p, uint64_t n, float mid,
                                                           float scale) {
    const int lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const RT *v = rows + r * ld;
    float s = 0.0f;  // multiples of 1/4 below 2^24: exact in any summation order
    for (uint32_t c = lane * 8; c < ldp * 2; c += 512) {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t code = 0;
            if (c + j < dim) {
                code = fp4_e2m1_code((row_elem(v, c + j) - mid) * scale);
                s += fp4_e2m1_sq(code);
            }
            w |= code << (4 * j);
        }
        *reinterpret_cast<uint32_t *>(rows_p + r * ldp + c / 2) = w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) norm_p[r] = s;
}

int ensure_fp4_mirror(zvdb_hip_index *h, hipStream_t s) {
    if (h->rows_p && h->count_p == h->count) return ZVDB_OK;
    if (h->count == 0) return ZVDB_OK;
    h->ldp = round_up(h->dim, 256) / 2;
    if (h->cap_p < h->count) {
        if (h->rows_p) { ZV_HIP(hipDeviceSynchronize()); /* searches of this index may be in flight on OTHER streams: the mutex covers the host side only */ ZV_HIP(hipFree(h->rows_p)); ZV_HIP(hipFree(h->norm_p)); }
        h->rows_p = nullptr;
        h->norm_p = nullptr;
        const uint64_t ncap = std::max<uint64_t>(h->count, h->cap);
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->rows_p), ncap * h->ldp));
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->norm_p), ncap * sizeof(float)));
        h->cap_p = ncap;
    }
    ZV_CHECK(fp4_range(h, s));
    ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        hipLaunchKernelGGL(k_quantize_rows_fp4<row_elem_t<decltype(rows)>>, dim3((uint32_t)((h->count + 3) / 4)), dim3(256), 0, s, rows, h->dim, ld,
                           h->rows_p, h->ldp, h->norm_p, h->count, h->p_mid, h->p_scale);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    }));
    h->count_p = h->count;
    return ZVDB_OK;
}

// sum and sum of squares of all (unpadded) elements: one fp64 pair of atomics per wave
template <typename RT>
__global__ __launch_bounds__(256) void k_moments_all(const RT *__restrict__ rows, uint32_t dim, uint32_t ld, uint64_t n,
                                                     double *__restrict__ out /* [2] */) {
    double s1 = 0.0, s2 = 0.0;
    const uint64_t total = n * ld;
    for (uint64_t t = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; t < total; t += (uint64_t)gridDim.x * blockDim.x * 4) {
        const uint32_t c = (uint32_t)(t % ld);
        const float4 v = row_load4(rows + t);
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c + j < dim) { s1 += (double)e[j]; s2 += (double)e[j] * (double)e[j]; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], s1);
        atomicAdd(&out[1], s2);
    }
}

// The fp4 map is placed by the MEAN and STANDARD DEVIATION of the shard's elements, not by their extremes: sigma maps to
// kFp4Amax / sqrt(3) = 2.6 units.  For uniformly spread data this is exactly "range -> [-4.5, 4.5]" (the measured optimum
// for the benchmark's generator); for bell-shaped data it keeps the 15 values busy (the extremes clip to +-6 at 2.3 sigma)
// where a min/max map would leave most elements in the 7 innermost values (recall@10 on N(0,1) data, 1M x 128:
// 0.93 with the min/max map).  Every metric centres the rows on the mean (L2: offsets cancel in q - v; dot / cosine: see below).
int fp4_range(zvdb_hip_index *h, hipStream_t s) {
    double *d_mom = nullptr;
    ZV_HIP(hipMalloc(reinterpret_cast<void **>(&d_mom), 16));
    ZV_HIP(hipMemsetAsync(d_mom, 0, 16, s));
    ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
        const uint64_t total4 = h->count * ld / 4;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((total4 + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(k_moments_all<row_elem_t<decltype(rows)>>, dim3(grid), dim3(256), 0, s, rows, h->dim, ld, h->count, d_mom);
        ZV_HIP(hipGetLastError());
        return ZVDB_OK;
    }));
    double got[2];
    ZV_HIP(hipMemcpyAsync(got, d_mom, 16, hipMemcpyDeviceToHost, s));
    ZV_HIP(hipStreamSynchronize(s));  // (lazy one-off build; not on the steady-state search path)
    ZV_HIP(hipFree(d_mom));
    const double cnt = (double)h->count * (double)h->dim;
    const double mean = got[0] / cnt, ms = got[1] / cnt;
    const double unit = (double)kFp4Amax / 1.7320508075688772;  // fp4 units per standard deviation
    // dot / cosine: the ROWS are centred too -- q.(v - mu 1) = q.v - mu sum(q) ranks like q.v for every query -- so that rows
    // of one sign (e.g. embeddings after a ReLU, U[0,1) data) use both halves of the code range; the QUERY is not centred
    // (that term would depend on the row): k_rescale_queries hands the quantiser q * f + mu, which it maps to q * f * scale.
    const double var = ms - mean * mean;
    h->p_mid = (float)mean;
    h->p_scale = var > 0.0 ? (float)(unit / std::sqrt(var)) : 1.0f;
    return ZVDB_OK;
}

// kNN-graph construction: search results of rows [r0, r0+n) (ids u64 [n][k], counts) -> adjacency rows
// (self dropped, nearest first, at most `degree` entries)
__global__ __launch_bounds__(256) void k_results_to_adj(const uint64_t *__restrict__ ids,
                                                        const uint32_t *__restrict__ cnt, uint32_t k,
                                                        uint64_t r0, uint32_t n, uint32_t degree,
                                                        uint32_t *__restrict__ adj, uint32_t *__restrict__ deg,
                                                        uint32_t gstride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t self = r0 + i;
    uint32_t w = 0;
    const uint32_t c = min(cnt[i], k);
    for (uint32_t j = 0; j < c && w < degree; ++j) {
        const uint64_t id = ids[(size_t)i * k + j];
        if (id != self) adj[self * gstride + w++] = (uint32_t)id;
    }
    deg[self] = w;
}

int launch_results_to_adj(const zvdb_hip_index *h, const uint64_t *d_ids, const uint32_t *d_cnt, uint32_t k,
                          uint64_t r0, uint32_t n, uint32_t degree, hipStream_t s) {
    if (n == 0) return ZVDB_OK;
    hipLaunchKernelGGL(k_results_to_adj, dim3((n + 255) / 256), dim3(256), 0, s, d_ids, d_cnt, k, r0, n, degree,
                       h->adj, h->deg, h->gstride);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// Structural check of a graph that came from OUTSIDE (zvdb_hip_load): every index the search / descent / insert kernels use
// without a bounds test.  Thread i < n: node i (degree, layer-0 ids, level, its upper-layer slots); thread n + r: upper row r
// (degree, ids).  bad[0] = OR of: 1 degree > stride, 2 neighbour id >= n, 4 level > max_level, 8 upper slots out of range.
__global__ __launch_bounds__(256) void k_validate_graph(uint64_t n, uint32_t gstride, uint64_t up_count, uint32_t max_level,
                                                        const uint32_t *__restrict__ adj, const uint32_t *__restrict__ deg,
                                                        const uint32_t *__restrict__ lvl, const uint32_t *__restrict__ slot,
                                                        const uint32_t *__restrict__ adj_up, const uint32_t *__restrict__ deg_up,
                                                        uint32_t *__restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t b = 0;
    if (i < n) {
        const uint32_t dg = deg[i];
        if (dg > gstride) b |= 1u;
        for (uint32_t j = 0; j < min(dg, gstride); ++j)
            if (adj[i * gstride + j] >= n) b |= 2u;
        const uint32_t lv = lvl[i];
        if (lv > max_level) b |= 4u;
        if (lv > 0 && (uint64_t)slot[i] + lv > up_count) b |= 8u;
    } else if (i < n + up_count) {
        const uint64_t r = i - n;
        const uint32_t dg = deg_up[r];
        if (dg > gstride) b |= 1u;
        for (uint32_t j = 0; j < min(dg, gstride); ++j)
            if (adj_up[r * gstride + j] >= n) b |= 2u;
    }
    if (b) atomicOr(bad, b);
}

int launch_validate_graph(const zvdb_hip_index *h, uint64_t n, uint64_t up_count, uint32_t max_level, uint32_t *d_bad, hipStream_t s) {
    const uint64_t tot = n + up_count;
    if (tot == 0) return ZVDB_OK;
    hipLaunchKernelGGL(k_validate_graph, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, s, n, h->gstride, up_count, max_level, h->adj, h->deg,
                       h->lvl, h->slot, h->adj_up, h->deg_up, d_bad);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// C1 consumer: global top-k of G all-gathered shard lists, order (dist, id).  One wave per
// query; rank by counting over the <= G*k live candidates held in LDS.
// rank g's arrays start at ids0 + g*rs_i, dist0 + g*rs_d, count0 + g*rs_c (byte strides)
__global__ __launch_bounds__(64) void k_merge_topk(const uint64_t *__restrict__ ids0,
                                                   const float *__restrict__ dist0,
                                                   const uint32_t *__restrict__ count0, uint64_t rs_i, uint64_t rs_d, uint64_t rs_c, uint32_t G,
                                                   uint32_t nq, uint32_t k,
                                                   uint64_t *__restrict__ o_ids,
                                                   float *__restrict__ o_dist,
                                                   uint32_t *__restrict__ o_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *ci = reinterpret_cast<uint64_t *>(smem);
    float *cd = reinterpret_cast<float *>(ci + (size_t)G * k);
    const int lane = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    const uint32_t tot = G * k;
    for (uint32_t t = lane; t < tot; t += kWave) {
        const uint32_t g = t / k, j = t - g * k;
        const uint64_t *ids = reinterpret_cast<const uint64_t *>(reinterpret_cast<const unsigned char *>(ids0) + g * rs_i);
        const float *dist = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(dist0) + g * rs_d);
        const uint32_t *count = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(count0) + g * rs_c);
        const bool live = j < count[qi];
        const size_t src = (size_t)qi * k + j;
        ci[t] = live ? ids[src] : ~0ull;
        cd[t] = live ? dist[src] : __builtin_inff();
    }
    __syncthreads();
    uint32_t total = 0;
    for (uint32_t base = 0; base < tot; base += kWave) {
        const uint32_t t = base + lane;
        const bool valid = t < tot && ci[t] != ~0ull;
        if (valid) {
            const float dt = cd[t];
            const uint64_t it = ci[t];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < tot; ++j) {
                const uint64_t ij = ci[j];
                const float dj = cd[j];
                if (ij != ~0ull && (dj < dt || (dj == dt && (ij < it || (ij == it && j < t))))) ++rank;
            }
            if (rank < k) {
                o_ids[(size_t)qi * k + rank] = it;
                o_dist[(size_t)qi * k + rank] = dt;
            }
        }
        total += (uint32_t)__popcll(__ballot(valid));
    }
    const uint32_t nres = total < k ? total : k;
    for (uint32_t i = nres + lane; i < k; i += kWave) {
        o_ids[(size_t)qi * k + i] = ~0ull;
        o_dist[(size_t)qi * k + i] = __builtin_inff();
    }
    if (lane == 0) o_count[qi] = nres;
}

int launch_merge_topk(const uint64_t *d_ids, const float *d_dist, const uint32_t *d_count,
                      uint32_t G, uint32_t nq, uint32_t k, uint64_t *o_ids, float *o_dist,
                      uint32_t *o_count, hipStream_t s) {
    // separate [G][nq][k] arrays: rank stride differs per array -> express via per-array strides = 0 and
    // pre-offset pointers is impossible; launch once per layout: ids stride nq*k*8, dist nq*k*4, count nq*4
    return launch_merge_topk_strided(d_ids, (uint64_t)nq * k * 8, d_dist, (uint64_t)nq * k * 4, d_count,
                                     (uint64_t)nq * 4, G, nq, k, o_ids, o_dist, o_count, s);
}

int launch_merge_topk_strided(const uint64_t *d_ids, uint64_t rs_ids, const float *d_dist, uint64_t rs_dist,
                              const uint32_t *d_count, uint64_t rs_cnt, uint32_t G, uint32_t nq, uint32_t k,
                              uint64_t *o_ids, float *o_dist, uint32_t *o_count, hipStream_t s) {
    if (nq == 0 || k == 0) return ZVDB_OK;
    const size_t lds = (size_t)G * k * 12;
    if (lds > 160 * 1024) {
        set_error("merge_topk: G*k = %u candidates exceed LDS", G * k);
        return ZVDB_ERR_UNSUPPORTED;
    }
    if (lds > 48 * 1024)
        ZV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_merge_topk),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_merge_topk, dim3(nq), dim3(64), lds, s, d_ids, d_dist, d_count, rs_ids, rs_dist, rs_cnt,
                       G, nq, k, o_ids, o_dist, o_count);
    ZV_HIP(hipGetLastError());
    return ZVDB_OK;
}

// ---- certificate statistics of a screen's mirror (bf_cert.h) ------------------------------------
// Per row the residual |x' - x| between the stored row x and what the screen multiplies (x' = the row rounded to bf16, or
// reconstructed from its int8 code) and the norm |x'|; out[0] / out[1] <- the largest of each over the shard (atomicMax on the
// bit patterns of non-negative floats, rounded up).  fp64 sums, one wave per row.  mode: enum zvdb_bf_mode (F32: x' = x).
template <typename RT>
__global__ __launch_bounds__(256) void k_cert_stats(const RT *__restrict__ rows, uint32_t dim, uint32_t ld, uint64_t n, int mode, float q_lo,
                                                    float q_inv, int affine, uint32_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    double mr = 0.0, mu = 0.0;
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (uint64_t)gridDim.x * 4) {
        const RT *v = rows + r * ld;
        double sr = 0.0, su = 0.0;
        for (uint32_t c = lane; c < dim; c += 64) {
            const float x = row_elem(v, c);
            double rec = (double)x;
            if (mode == ZVDB_BF_BF16) rec = (double)__uint_as_float((uint32_t)f32_to_bf16_rne(x) << 16);
            else if (mode == ZVDB_BF_I8) {
                const float f = affine ? __builtin_rintf((x - q_lo) * q_inv) - 128.0f : __builtin_rintf(x * q_inv);
                const double code = (double)fminf(fmaxf(f, -128.0f), 127.0f);
                rec = affine ? (double)q_lo + (code + 128.0) / (double)q_inv : code / (double)q_inv;
            }
            sr += (rec - (double)x) * (rec - (double)x);
            su += rec * rec;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sr += __shfl_xor(sr, off);
            su += __shfl_xor(su, off);
        }
        mr = fmax(mr, sr);
        mu = fmax(mu, su);
    }
    if (lane == 0) {
        auto up = [](double sq) {  // sqrt, rounded up to a float
            const double d = sqrt(sq) * (1.0 + 1e-12);
            float f = (float)d;
            if ((double)f < d) f = __uint_as_float(__float_as_uint(f) + 1u);
            return __float_as_uint(f);
        };
        atomicMax(out + 0, up(mr));
        atomicMax(out + 1, up(mu));
    }
}

int ensure_cert_stats(zvdb_hip_index *h, uint32_t mode, hipStream_t s) {
    if (mode > ZVDB_BF_I8) return ZVDB_OK;
    if (!h->cert_dev) {
        ZV_HIP(hipMalloc(reinterpret_cast<void **>(&h->cert_dev), 8 * sizeof(uint32_t)));
        ZV_HIP(hipMemsetAsync(h->cert_dev, 0, 8 * sizeof(uint32_t), s));
    }
    if (h->cert_count[mode] == h->count) {
        // computed already -- but perhaps by a call on ANOTHER stream that has only been enqueued: a certified search reads the
        // words on its own stream, so it waits for the event recorded behind the kernel that filled them (a no-op once complete)
        if (h->cert_event_set) ZV_HIP(hipStreamWaitEvent(s, h->cert_event, 0));
        return ZVDB_OK;
    }
    if (h->cert_event_set) ZV_HIP(hipStreamWaitEvent(s, h->cert_event, 0));  // (another mode's statistics may still be in flight on another stream: same words' neighbours, same memset unit)
    ZV_HIP(hipMemsetAsync(h->cert_dev + 2 * mode, 0, 2 * sizeof(uint32_t), s));
    if (h->count > 0)
        ZV_CHECK(rows_dispatch(h, [&](auto rows, uint32_t ld) -> int {
            const uint32_t grid = (uint32_t)std::min<uint64_t>((h->count + 3) / 4, 256 * 16);
            hipLaunchKernelGGL(k_cert_stats<row_elem_t<decltype(rows)>>, dim3(grid), dim3(256), 0, s, rows, h->dim, ld, h->count, (int)mode, h->q_lo,
                               h->q_inv, h->metric == ZVDB_METRIC_L2SQ ? 1 : 0, h->cert_dev + 2 * mode);
            ZV_HIP(hipGetLastError());
            return ZVDB_OK;
        }));
    if (!h->cert_event) ZV_HIP(hipEventCreateWithFlags(&h->cert_event, hipEventDisableTiming));
    ZV_HIP(hipEventRecord(h->cert_event, s));
    h->cert_event_set = true;
    h->cert_count[mode] = h->count;
    return ZVDB_OK;
}

}  // namespace zvdb
