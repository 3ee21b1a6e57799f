// dma_geometry_bench.hip -- micro-benchmark behind a DESIGN.md decision: LDS-DMA throughput per CU of the
// scan kernel's operand stream in two piece geometries at equal bytes:
//   G64 : 16 rows x 64 B per wave-instruction (the v3/v4 ring: 64-byte K slices)
//   G128:  8 rows x 128 B per wave-instruction (full cache lines; 128-byte K slices)
// Same traffic shape as k_bf_scan4: per step each workgroup (512 threads, 1 per CU) pulls 16 KB of a
// 256-row "query" panel (re-read every tile: L2 resident) + 16 KB of a streaming "corpus" panel.
// Build: hipcc --offload-arch=gfx950 -O3 tools/dma_geometry_bench.hip -o /tmp/dma_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int ROWB>  // bytes of a row per slice: 64 or 128
__global__ __launch_bounds__(512) void k(const unsigned char *Q, const unsigned char *V, unsigned row_bytes,
                                          unsigned tiles_per_block, unsigned long long *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int RPP = 1024 / ROWB;              // rows per 1-KiB piece
    constexpr int PIECES = 256 / RPP;             // pieces per operand per slice
    constexpr int NPW = PIECES / 8;               // pieces per wave
    constexpr int LPR = ROWB / 16;                // lanes per row
    constexpr int STAGE = 2 * 256 * ROWB;
    constexpr int NST = 131072 / STAGE;           // ring stages in 128 KiB
    const unsigned KT = row_bytes / ROWB;
    unsigned off[NPW];
    for (int i = 0; i < NPW; ++i) off[i] = ((wave * NPW + i) * RPP + lane / LPR) * row_bytes + (lane % LPR) * 16;
    unsigned long long it = 0;
    const unsigned char *vb = V + (size_t)blockIdx.x * tiles_per_block * 256 * row_bytes;
    for (unsigned t = 0; t < tiles_per_block; ++t)
        for (unsigned kt = 0; kt < KT; ++kt, ++it) {
            unsigned char *dst = smem + (it % NST) * STAGE;
            for (int i = 0; i < NPW; ++i) {
                __builtin_amdgcn_global_load_lds((gptr_t)(vb + (size_t)t * 256 * row_bytes + kt * ROWB + off[i]),
                                                 (lptr_t)(dst + (wave * NPW + i) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(Q + kt * ROWB + off[i]),
                                                 (lptr_t)(dst + 256 * ROWB + (wave * NPW + i) * 1024), 16, 0, 0);
            }
            if (ROWB == 64) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // 2 slices (of 4 ops) younger
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");              // 1 slice (of 8 ops) younger
            __builtin_amdgcn_s_barrier();
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = smem[0];
}

int main() {
    const unsigned row_bytes = 768, tiles = 148, blocks = 256;
    const size_t vbytes = (size_t)blocks * tiles * 256 * row_bytes;  // 7.4 GB like pass C (int8 rows)
    unsigned char *Q, *V; unsigned long long *sink;
    hipMalloc(&Q, 256 * row_bytes); hipMalloc(&V, vbytes); hipMalloc(&sink, blocks * 8);
    hipMemset(Q, 1, 256 * row_bytes); hipMemset(V, 2, vbytes);
    hipFuncSetAttribute((const void *)k<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep)
        for (int g = 0; g < 2; ++g) {
            hipEventRecord(a);
            if (g == 0) hipLaunchKernelGGL(k<64>, dim3(blocks), dim3(512), 131072, 0, Q, V, row_bytes, tiles, sink);
            else hipLaunchKernelGGL(k<128>, dim3(blocks), dim3(512), 131072, 0, Q, V, row_bytes, tiles, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double bytes = 2.0 * vbytes;  // corpus + equal query bytes
            printf("geometry %3d B rows: %.3f ms  %.2f TB/s aggregate  %.1f GB/s per CU (%s)\n", g == 0 ? 64 : 128, ms,
                   bytes / ms / 1e9, bytes / ms / 1e6 / 256, hipGetErrorString(hipGetLastError()));
        }
    return 0;
}
