// mfma_peak_bench.hip -- measured dense MFMA issue rate on this part for the three screens' instructions:
// 8 independent 32x32 accumulators per wave, 8 waves per workgroup (2 per SIMD), 1 workgroup per CU, no memory.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// RND = 1: operand registers filled with hashed bits (the chip holds a lower clock on random data than on
// constants: MI355X_MICROARCH.md "DVFS give-back"); MODE 3 = fp4 E2M1 via v_mfma_scale_f32_32x32x64_f8f6f4.
template <int MODE, int WAVES, int RND = 0>
__global__ __launch_bounds__(WAVES * 64) void k(int iters, int *sink) {
    i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    if constexpr (RND) {
        const unsigned t = threadIdx.x + 1024u * blockIdx.x;
        a = i32x4{(int)hash32(t * 8 + 0), (int)hash32(t * 8 + 1), (int)hash32(t * 8 + 2), (int)hash32(t * 8 + 3)};
        b = i32x4{(int)hash32(t * 8 + 4), (int)hash32(t * 8 + 5), (int)hash32(t * 8 + 6), (int)hash32(t * 8 + 7)};
        if constexpr (MODE == 1) {  // keep bf16 exponents finite and similar: sign + 7 mantissa bits, exponent 127
            for (int i = 0; i < 4; ++i) { a[i] = (a[i] & 0x807F807F) | 0x3F803F80; b[i] = (b[i] & 0x807F807F) | 0x3F803F80; }
        }
    }
    if constexpr (MODE == 4) {  // fp4, 16x16x128 shape: 16 independent accumulators of 4 registers
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 acc[16] = {};
        const i32x8 wa = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, wb = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j)
                acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, wb, acc[j], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        float s = 0;
        for (int j = 0; j < 16; ++j) s += acc[j][0];
        if (s == 12345.f) sink[0] = 1;
    } else if constexpr (MODE == 3) {
        f32x16 acc[8] = {};
        const i32x8 wa = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, wb = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, wb, acc[j], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        float s = 0;
        for (int j = 0; j < 8; ++j) s += acc[j][0];
        if (s == 12345.f) sink[0] = 1;
    } else if constexpr (MODE == 2) {
        i32x16 acc[8] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
        int s = 0;
        for (int j = 0; j < 8; ++j) s += acc[j][0];
        if (s == 12345) sink[0] = s;
    } else if constexpr (MODE == 1) {
        f32x16 acc[8] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[j], 0, 0, 0);
        float s = 0;
        for (int j = 0; j < 8; ++j) s += acc[j][0];
        if (s == 12345.f) sink[0] = 1;
    } else {
        f32x16 acc[8] = {};
        const float fa = threadIdx.x, fb = 2.f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[j], 0, 0, 0);
        float s = 0;
        for (int j = 0; j < 8; ++j) s += acc[j][0];
        if (s == 12345.f) sink[0] = 1;
    }
}

template <int MODE, int WAVES, int RND = 0>
static void run(const char *name, double ops_per_mfma, int *sink) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, WAVES, RND>), dim3(256), dim3(WAVES * 64), 0, 0, iters, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ops = 256.0 * WAVES * iters * 8 * ops_per_mfma;
    printf("%-28s %d waves/CU: %.3f ms  %.1f TOP/s  (%.1f clk per MFMA per SIMD at 2.4 GHz)\n", name, WAVES, best,
           ops / best / 1e9, best * 1e-3 * 2.4e9 / (iters * 8.0 * WAVES / 4));
}

int main() {
    int *sink; (void)hipMalloc(&sink, 4);
    run<2, 8>("mfma_i32_32x32x32_i8", 2.0 * 32 * 32 * 32, sink);
    run<2, 4>("mfma_i32_32x32x32_i8", 2.0 * 32 * 32 * 32, sink);
    run<1, 8>("mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, sink);
    run<1, 4>("mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, sink);
    run<0, 8>("mfma_f32_32x32x2f32", 2.0 * 32 * 32 * 2, sink);
    run<3, 8>("mfma_scale_32x32x64 fp4", 2.0 * 32 * 32 * 64, sink);
    run<3, 8, 1>("mfma_scale_32x32x64 fp4 RANDOM", 2.0 * 32 * 32 * 64, sink);
    run<3, 4, 1>("mfma_scale_32x32x64 fp4 RANDOM", 2.0 * 32 * 32 * 64, sink);
    run<4, 8>("mfma_scale_16x16x128 fp4 (x2 per iter)", 2.0 * 2 * 16 * 16 * 128, sink);
    run<4, 8, 1>("mfma_scale_16x16x128 fp4 RANDOM (x2)", 2.0 * 2 * 16 * 16 * 128, sink);
    run<3, 8, 1>("mfma_scale_32x32x64 fp4 RANDOM again", 2.0 * 32 * 32 * 64, sink);
    run<2, 8, 1>("mfma_i32_32x32x32_i8 RANDOM", 2.0 * 32 * 32 * 32, sink);
    run<1, 8, 1>("mfma_f32_32x32x16_bf16 RANDOM", 2.0 * 32 * 32 * 16, sink);
    return 0;
}
