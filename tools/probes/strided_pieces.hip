// HBM throughput of the exact scan's access pattern (exact_scan.hip): a block walks tiles of 512 consecutive rows of 3,072 bytes, one
// PIECE of every row per step (128 bytes in the product: 8 lanes per row), all pieces of a tile before the next tile.  How much does
// the piece size change what the memory gives?  PAIR = 1: pieces of 128 bytes, but steps c and c + 1 issued back to back.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/strided_pieces tools/probes/strided_pieces.hip && /tmp/strided_pieces
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int PIECE, int PAIR>
__global__ __launch_bounds__(512) void k_walk(const unsigned char *rows, uint64_t N, uint32_t row_bytes, uint32_t tiles_per_block, uint32_t *out) {
    constexpr int LPR = PIECE / 16;            // lanes per row piece
    constexpr int RPI = 512 / LPR;             // rows per load instruction of the block
    constexpr int IT = 512 / RPI;              // load instructions per step
    const uint32_t tid = threadIdx.x, fr = tid / LPR, fp = tid % LPR;
    const uint32_t nch = row_bytes / PIECE;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t t = 0; t < tiles_per_block; ++t) {
        const uint64_t r0 = ((uint64_t)blockIdx.x * tiles_per_block + t) * 512;
        if (r0 + 512 > N) break;
        for (uint32_t c = 0; c < nch; c += 1 + PAIR) {
            u32x4 v[IT * (1 + PAIR)];
#pragma unroll
            for (int p = 0; p <= PAIR; ++p)
#pragma unroll
                for (int i = 0; i < IT; ++i)
                    v[p * IT + i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rows + (r0 + fr + (uint64_t)RPI * i) * row_bytes + (size_t)(c + p) * PIECE + fp * 16));
#pragma unroll
            for (int i = 0; i < IT * (1 + PAIR); ++i) acc ^= v[i];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int PIECE, int PAIR>
int run(const unsigned char *rows, uint64_t N, uint32_t *out, const char *name) {
    const uint32_t blocks = 256 * 4, tpb = (uint32_t)(N / 512 / blocks);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_walk<PIECE, PAIR>), dim3(blocks), dim3(512), 0, 0, rows, N, 3072u, tpb, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)blocks * tpb * 512 * 3072;
    printf("%-44s %7.3f ms  %6.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    const uint64_t N = 10000000;
    unsigned char *rows; uint32_t *out;
    CK(hipMalloc(&rows, N * 3072)); CK(hipMalloc(&out, 64));
    CK(hipMemset(rows, 1, N * 3072));
    if (run<128, 0>(rows, N, out, "128-byte pieces (the product's pattern)")) return 1;
    if (run<128, 1>(rows, N, out, "128-byte pieces, two steps back to back")) return 1;
    if (run<256, 0>(rows, N, out, "256-byte pieces")) return 1;
    if (run<512, 0>(rows, N, out, "512-byte pieces")) return 1;
    if (run<1024, 0>(rows, N, out, "1-KiB pieces")) return 1;
    return 0;
}
