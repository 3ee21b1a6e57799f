// dma_l2_bench.hip -- how fast can one CU pull operand panels through LDS-DMA when they HIT L2?
// Same piece geometry as k_bf_scan5 (8 rows x 128 B per wave instruction, 512 threads, 1 workgroup per CU).
// mode 0: all workgroups re-read the same `vt` corpus tiles (L2 resident) + the query panel (L2 resident)
// mode 1: groups of 4 consecutive workgroups share a corpus stream (the scan kernel's pattern: 1 HBM read + 3 L2 hits)
// mode 2: every workgroup streams its own corpus (HBM)
// DEPTH = slices allowed in flight behind the one being waited for (ring has DEPTH+1 stages of 64 KB... LDS permitting)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int NST, int HALF>  // ring stages; HALF=1: only the corpus panel is loaded (32 KB per slice)
__global__ __launch_bounds__(512) void k(const unsigned char *Q, const unsigned char *V, unsigned row_bytes,
                                          unsigned tiles_per_block, unsigned vmod, unsigned share, unsigned dephase,
                                          unsigned long long *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int ROWB = 128, RPP = 8, NPW = 4, LPR = 8;
    constexpr int STAGE = (HALF ? 1 : 2) * 256 * ROWB;
    const unsigned KT = row_bytes / ROWB;
    const bool slice_major = dephase & 4;   // corpus stored [tile][k-slice][256 rows][128 B]: a slice is one contiguous 32 KB block
    dephase &= 3;
    unsigned off[NPW], offv[NPW];
    for (int i = 0; i < NPW; ++i) {
        off[i] = ((wave * NPW + i) * RPP + lane / LPR) * row_bytes + (lane % LPR) * 16;
        offv[i] = slice_major ? ((wave * NPW + i) * RPP + lane / LPR) * ROWB + (lane % LPR) * 16 : off[i];
    }
    const unsigned kstride = slice_major ? 256 * ROWB : ROWB;
    unsigned long long it = 0;
    // XCD-aware like the scan kernel: workgroup b runs on XCD b%8; `share` consecutive workgroups OF ONE XCD share a stream
    const unsigned stream = share == 1 ? blockIdx.x : (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / share);
    const unsigned j = share == 1 ? 0 : (blockIdx.x >> 3) % share;   // my position among the sharers
    const unsigned rot = dephase == 1 ? (j == 0 ? 0 : j == 1 ? 2 : j == 2 ? 4 : 1) : 0;   // rotated k-slice order
    const unsigned lag = dephase == 2 ? j : 0;                                            // rotated tile order
    for (unsigned t0 = 0; t0 < tiles_per_block; ++t0) {
        const unsigned t = (t0 + tiles_per_block - lag) % tiles_per_block;
        const unsigned tile = (stream * tiles_per_block + t) % vmod;
        const unsigned char *vb = V + (size_t)tile * 256 * row_bytes;
        for (unsigned kt0 = 0; kt0 < KT; ++kt0, ++it) {
            const unsigned kt = (kt0 + rot) % KT;
            unsigned char *dst = smem + (it % NST) * STAGE;
            for (int i = 0; i < NPW; ++i) {
                __builtin_amdgcn_global_load_lds((gptr_t)(vb + kt * kstride + offv[i]),
                                                 (lptr_t)(dst + (wave * NPW + i) * 1024), 16, 0, 0);
                if (!HALF)
                    __builtin_amdgcn_global_load_lds((gptr_t)(Q + kt * ROWB + off[i]),
                                                     (lptr_t)(dst + 256 * ROWB + (wave * NPW + i) * 1024), 16, 0, 0);
            }
            constexpr int OPS = HALF ? 4 : 8;
            if (NST == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
            else if (NST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * OPS) : "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = smem[0];
}

template <int NST, int HALF>
static void run(const char *name, const unsigned char *Q, const unsigned char *V, unsigned tiles, unsigned vmod,
                unsigned share, unsigned dephase, unsigned long long *sink) {
    constexpr int STAGE = (HALF ? 1 : 2) * 256 * 128;
    const unsigned row_bytes = 768, blocks = 256;
    hipFuncSetAttribute((const void *)k<NST, HALF>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<NST, HALF>), dim3(blocks), dim3(512), NST * STAGE, 0, Q, V, row_bytes, tiles, vmod, share, dephase, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double bytes = (HALF ? 1.0 : 2.0) * blocks * (double)tiles * 256 * row_bytes;
    const double per_cu = bytes / best / 1e6 / 256;
    printf("%-44s dephase %u stages %d %s: %.3f ms  %.2f TB/s aggregate  %.1f GB/s = %.1f B/clk per CU (%s)\n", name, dephase, NST,
           HALF ? "corpus only" : "corpus+query", best, bytes / best / 1e9, per_cu, per_cu / 2.4,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    const unsigned row_bytes = 768, tiles = 148, blocks = 256;
    const size_t vbytes = (size_t)blocks * tiles * 256 * row_bytes;
    unsigned char *Q, *V; unsigned long long *sink;
    hipMalloc(&Q, 256 * row_bytes); hipMalloc(&V, vbytes); hipMalloc(&sink, blocks * 8);
    hipMemset(Q, 1, 256 * row_bytes); hipMemset(V, 2, vbytes);
    const unsigned all = blocks * tiles;
    run<2, 0>("L2-resident (8 shared tiles)", Q, V, tiles, 8, 1, 0, sink);
    run<2, 1>("L2-resident (8 shared tiles)", Q, V, tiles, 8, 1, 0, sink);
    run<4, 1>("L2-resident (8 shared tiles)", Q, V, tiles, 8, 1, 0, sink);
    run<2, 0>("4 workgroups share a stream (1 HBM + 3 L2)", Q, V, tiles, all, 4, 0, sink);
    run<2, 1>("4 workgroups share a stream (1 HBM + 3 L2)", Q, V, tiles, all, 4, 0, sink);
    run<4, 1>("4 workgroups share a stream (1 HBM + 3 L2)", Q, V, tiles, all, 4, 0, sink);
    run<2, 0>("4 workgroups share a stream (1 HBM + 3 L2)", Q, V, tiles, all, 4, 1, sink);
    run<2, 1>("4 workgroups share a stream (1 HBM + 3 L2)", Q, V, tiles, all, 4, 1, sink);
    run<2, 0>("4 workgroups share a stream (1 HBM + 3 L2)", Q, V, tiles, all, 4, 2, sink);
    run<2, 1>("4 workgroups share a stream (1 HBM + 3 L2)", Q, V, tiles, all, 4, 2, sink);
    run<4, 1>("4 workgroups share a stream (1 HBM + 3 L2)", Q, V, tiles, all, 4, 2, sink);
    run<2, 0>("shared, slice-major corpus", Q, V, tiles, all, 4, 4, sink);
    run<2, 0>("shared, slice-major corpus", Q, V, tiles, all, 4, 5, sink);
    run<2, 1>("shared, slice-major corpus", Q, V, tiles, all, 4, 4, sink);
    run<2, 1>("shared, slice-major corpus", Q, V, tiles, all, 4, 5, sink);
    run<2, 1>("private, slice-major corpus", Q, V, tiles, all, 1, 4, sink);
    run<4, 1>("private, slice-major corpus", Q, V, tiles, all, 1, 4, sink);
    run<2, 0>("private streams (HBM)", Q, V, tiles, all, 1, 0, sink);
    run<4, 1>("private streams (HBM)", Q, V, tiles, all, 1, 0, sink);
    return 0;
}
