// What a bare stream of v_mfma_f32_32x32x2_f32 reaches on an MI355X, by waves per SIMD and accumulation chains per wave -- no memory, no LDS.
// (bf_f32_regq.hip runs two waves per SIMD, one chain each, and measures MFMA-busy 0.86 - 0.88 at 2.35 GHz whatever feeds it: is that the instruction's
// ceiling, or the price of sharing a SIMD?)     hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_f32_issue tools/probes/mfma_f32_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int CHAINS>
__global__ void k(float *out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
    }
    float s = 0.0f;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = s;
}

template <int CHAINS>
int run(int waves_per_simd, float *out) {
    const int threads = 256 * waves_per_simd, iters = 20000 / CHAINS;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f, 2.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double flops = 256.0 * (threads / 64) * iters * 16.0 * CHAINS * (32.0 * 32 * 2 * 2);
    printf("%d wave(s) per SIMD, %d chain(s) per wave: %8.3f ms  %6.1f TFLOP/s = %.3f of 157.3\n", waves_per_simd, CHAINS, best, flops / (best * 1e-3) / 1e12,
           flops / (best * 1e-3) / 157.3e12);
    return 0;
}

int main() {
    float *out;
    CK(hipMalloc(&out, 64));
    if (run<1>(1, out) || run<2>(1, out) || run<1>(2, out) || run<2>(2, out) || run<1>(4, out)) return 1;
    return 0;
}
