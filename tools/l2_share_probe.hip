// l2_share_probe.hip -- do the workgroups of one XCD that stream the SAME bytes at about the same time share one fabric fetch?
// Layout of the fp4 stream kernel: blockIdx & 7 = XCD (round-robin placement), the R workgroups with the same (blockIdx & 7)
// read the same region, each with 8 waves x a 6-deep ring of 1-KB loads.  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./l2_share_probe
// and compare FETCH_SIZE x 2 (gfx950 16-B/lane correction) with the printed "unique bytes" of each launch.
// build: hipcc --offload-arch=gfx950 -O3 tools/l2_share_probe.hip -o tools/l2_share_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int RING>
__global__ __launch_bounds__(512) void k_read(const unsigned char *__restrict__ buf, size_t region_bytes, int readers, int skew_kb,
                                              unsigned *__restrict__ sink) {
    // blockIdx = ((stream * readers) + r) * 8 + xcd: `readers` workgroups read the region of (xcd, stream)
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, r = jb % readers, stream = jb / readers;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char *base = buf + ((size_t)stream * 8 + xcd) * region_bytes;
    const size_t steps = region_bytes / (8 * 1024);  // one step = 8 waves x 1 KB
    const size_t skew = ((size_t)skew_kb * 1024 / (8 * 1024)) * (size_t)r;  // reader r starts `skew` steps into the region (wraps)
    u32x4 ring[RING];
    unsigned acc = 0;
    for (int i = 0; i < RING; ++i) {
        const size_t st = (skew + i) % steps;
        ring[i] = *reinterpret_cast<const u32x4 *>(base + st * 8192 + wave * 1024 + lane * 16);
    }
    for (size_t s = 0; s < steps; ++s) {
        const u32x4 v = ring[s % RING];
        acc += v.x ^ v.y ^ v.z ^ v.w;
        const size_t st = (skew + s + RING) % steps;
        ring[s % RING] = *reinterpret_cast<const u32x4 *>(base + st * 8192 + wave * 1024 + lane * 16);
        // a little ALU work per step so that readers run at the pace of a compute-bound consumer
        for (int k = 0; k < 64; ++k) acc = acc * 1664525u + 1013904223u;
    }
    if (acc == 0x12345678u) sink[0] = acc;
    (void)readers;
}

int main() {
    const size_t region = 96ull << 20;  // per (XCD, stream): 96 MB (past the 4-MB L2; the whole buffer past the 256-MB Infinity Cache)
    unsigned char *buf = nullptr;
    unsigned *sink = nullptr;
    hipMalloc(&buf, region * 8 * 8);
    hipMalloc(&sink, 64);
    hipMemset(buf, 1, region * 8 * 8);
    auto run = [&](int streams, int readers, int skew_kb) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_read<6>, dim3(8 * streams * readers), dim3(512), 0, 0, buf, region, readers, skew_kb, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("streams/XCD %d x readers %d, start skew %4d KB between readers: unique bytes %.1f MB, read bytes %.1f MB, %.3f ms\n", streams,
               readers, skew_kb, region * 8.0 * streams / 1e6, region * 8.0 * streams * readers / 1e6, ms);
    };
    for (int readers : {1, 2, 4, 8})
        for (int skew_kb : {0, 256, 2048}) run(1, readers, skew_kb);
    // the scan's occupancy of an XCD: 8 streams x 4 readers = 32 workgroups, one per CU
    for (int skew_kb : {0, 96, 256, 512}) run(8, 4, skew_kb);
    return 0;
}
