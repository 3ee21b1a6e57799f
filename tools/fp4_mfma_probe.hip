#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
using i32x8 = __attribute__((ext_vector_type(8))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using f32x16 = __attribute__((ext_vector_type(16))) float;
// A: [32 rows][64 k] fp4 codes (one per byte on host), B: [32 cols][64 k]; lane l loads 16 B = 32 nibbles of row l&31, k-half l>>5
__global__ void probe(const uint8_t *A, const uint8_t *B, float *C, int scale) {
    const int l = threadIdx.x;
    i32x4 a = *reinterpret_cast<const i32x4 *>(A + (l & 31) * 32 + (l >> 5) * 16);
    i32x4 b = *reinterpret_cast<const i32x4 *>(B + (l & 31) * 32 + (l >> 5) * 16);
    i32x8 a8 = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, b8 = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, scale, 0, scale);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
using f32x4 = __attribute__((ext_vector_type(4))) float;
// 16x16x128: A [16 rows][128 k], B [16 cols][128 k]; assumed lane map: row/col = l & 15, k-block = 32 * (l >> 4)
__global__ void probe16(const uint8_t *A, const uint8_t *B, float *C, int scale) {
    const int l = threadIdx.x;
    i32x4 a = *reinterpret_cast<const i32x4 *>(A + (l & 15) * 64 + (l >> 4) * 16);
    i32x4 b = *reinterpret_cast<const i32x4 *>(B + (l & 15) * 64 + (l >> 4) * 16);
    i32x8 a8 = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, b8 = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
    f32x4 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c, 4, 4, 0, scale, 0, scale);
    for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
int main16() {
    const float lv[16] = {0, .5f, 1, 1.5f, 2, 3, 4, 6, -0.0f, -.5f, -1, -1.5f, -2, -3, -4, -6};
    std::vector<uint8_t> ca(16 * 128), cb(16 * 128), pa(16 * 64), pb(16 * 64);
    uint32_t s = 777;
    for (auto *v : {&ca, &cb}) for (auto &x : *v) { s = s * 1664525u + 1013904223u; x = (s >> 24) & 15; }
    for (int r = 0; r < 16; ++r) for (int k = 0; k < 128; k += 2) {
        pa[r * 64 + k / 2] = ca[r * 128 + k] | (ca[r * 128 + k + 1] << 4);
        pb[r * 64 + k / 2] = cb[r * 128 + k] | (cb[r * 128 + k + 1] << 4);
    }
    uint8_t *dA, *dB; float *dC;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024);
    hipMemcpy(dA, pa.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, pb.data(), 1024, hipMemcpyHostToDevice);
    probe16<<<1, 64>>>(dA, dB, dC, 0x7F7F7F7F);
    std::vector<float> C(256);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float ref = 0;
        for (int k = 0; k < 128; ++k) ref += lv[ca[i * 128 + k]] * lv[cb[j * 128 + k]];
        if (C[i * 16 + j] != ref) { if (bad < 4) printf("  [%d][%d] got %g want %g\n", i, j, C[i * 16 + j], ref); ++bad; }
    }
    printf("16x16x128: mismatches %d / 256\n", bad);
    return 0;
}
int main() {
    main16();
    const float lv[16] = {0, .5f, 1, 1.5f, 2, 3, 4, 6, -0.0f, -.5f, -1, -1.5f, -2, -3, -4, -6};
    std::vector<uint8_t> ca(32 * 64), cb(32 * 64), pa(32 * 32), pb(32 * 32);
    uint32_t s = 12345;
    for (auto *v : {&ca, &cb}) for (auto &x : *v) { s = s * 1664525u + 1013904223u; x = (s >> 24) & 15; }
    for (int r = 0; r < 32; ++r) for (int k = 0; k < 64; k += 2) {
        pa[r * 32 + k / 2] = ca[r * 64 + k] | (ca[r * 64 + k + 1] << 4);
        pb[r * 32 + k / 2] = cb[r * 64 + k] | (cb[r * 64 + k + 1] << 4);
    }
    uint8_t *dA, *dB; float *dC;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
    hipMemcpy(dA, pa.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, pb.data(), 1024, hipMemcpyHostToDevice);
    for (int scale : {0x7F7F7F7F, 0}) {
        probe<<<1, 64>>>(dA, dB, dC, scale);
        std::vector<float> C(1024);
        hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
        int bad = 0; double maxabs = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            float ref = 0;
            for (int k = 0; k < 64; ++k) ref += lv[ca[i * 64 + k]] * lv[cb[j * 64 + k]];
            if (C[i * 32 + j] != ref) { if (bad < 4) printf("  [%d][%d] got %g want %g\n", i, j, C[i * 32 + j], ref); ++bad; }
            if (fabs(ref) > maxabs) maxabs = fabs(ref);
        }
        printf("scale %08x: mismatches %d / 1024 (max |ref| %g)\n", scale, bad, maxabs);
    }
    return 0;
}
