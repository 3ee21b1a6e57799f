// bf16_regq_bench.hip -- prototype / ablation harness of the bf16 "register-resident queries" scan (bf_bf16_stream.hip):
// every wave keeps the MFMA B operand of its 32 queries (all of K) in VGPRs, the corpus streams once per workgroup through
// a 3-slot LDS ring in MFMA fragment order (LDS-DMA, one barrier per 32-row group), every wave reads every A fragment.
// Checks the per-query maximum of (q.v - |v|^2/2) against a naive kernel on a small corpus, then times a large one.
//   hipcc -O3 --offload-arch=gfx950 tools/bf16_regq_bench.hip -o tools/bf16_regq_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#include <utility>
#include <cstring>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ inline uint64_t mix64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ inline uint16_t f2bf(float v) {
    uint32_t u = __float_as_uint(v);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ inline float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ float g_shift = 0.0f;
__device__ inline float elem(uint64_t seed, uint64_t row, uint32_t k, uint32_t dim) {
    return (float)(mix64(seed, row * dim + k) >> 40) * (1.0f / 16777216.0f) - g_shift;
}

// fragment order: unit (group g, k-step j, lane l) = 8 bf16 of row 32 g + (l & 31), elements 16 j + 8 (l >> 5) + [0, 8)
__global__ void k_gen_frag(uint64_t seed, uint64_t nrows, uint32_t dim, uint32_t KS, uint64_t units, unsigned char *out, float *norm) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const uint32_t l = (uint32_t)(u & 63);
    const uint64_t gj = u >> 6;
    const uint32_t j = (uint32_t)(gj % KS);
    const uint64_t row = (gj / KS) * 32 + (l & 31);
    const uint32_t k0 = 16 * j + 8 * (l >> 5);
    u32x4 w = {0, 0, 0, 0};
    if (row < nrows)
        for (int e = 0; e < 8; ++e) {
            const uint32_t k = k0 + e;
            const uint16_t h = k < dim ? f2bf(elem(seed, row, k, dim)) : 0;
            w[e >> 1] |= (uint32_t)h << (16 * (e & 1));
        }
    *reinterpret_cast<u32x4 *>(out + u * 16) = w;
    if (norm && j == 0 && l < 32) {
        float s = __builtin_inff();
        if (row < nrows) {
            s = 0.0f;
            for (uint32_t k = 0; k < dim; ++k) { const float v = bf2f(f2bf(elem(seed, row, k, dim))); s += v * v; }
        }
        norm[row] = s;
    }
}

__device__ __forceinline__ const __attribute__((address_space(1))) unsigned char *uniform_ptr(const unsigned char *p) {
    const uint64_t v = (uint64_t)(size_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (const __attribute__((address_space(1))) unsigned char *)(size_t)(((uint64_t)hi << 32) | lo);
}

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

template <int OFF> __device__ __forceinline__ void lds_read128(u32x4 &d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void lgkm_wait(u32x4 &d) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N)); }

__device__ inline uint32_t fkey(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

// ABL bits: 1 = no DMA in the loop, 2 = no MFMA, 4 = no LDS fragment reads (one read per group), 8 = no epilogue
template <int KS, int R, int ABL, int STAG>
__global__ __launch_bounds__(512, 2) void k_regq(const unsigned char *__restrict__ Qf, const unsigned char *__restrict__ Vf,
                                                  const float *__restrict__ vnorm, uint32_t G, uint32_t S, uint32_t QT,
                                                  uint32_t *__restrict__ out, unsigned long long *__restrict__ stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    constexpr int P = KS / 8;                     // DMA pieces per wave and group
    constexpr uint32_t SLOT = KS * 1024 + 256;    // fragments + 32 norms (+ pad)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t fh = lane >> 5;
    const uint32_t b = blockIdx.x, x = b & 7, jb = b >> 3;
    const uint32_t qt = jb % QT, s = (jb / QT) * 8 + x;
    const uint32_t per = (G + S - 1) / S;
    const uint32_t g0 = min(s * per, G), g1 = min(g0 + per, G);
    const uint32_t n = g1 - g0;
    if (n == 0) return;

    u32x4 B[KS];
    {
        const unsigned char *qsrc = Qf + ((size_t)(qt * 8 + wave) * KS) * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < KS; ++j) B[j] = *reinterpret_cast<const u32x4 *>(qsrc + (size_t)j * 1024);
    }
    const uint32_t lane16 = (uint32_t)lane * 16;
    auto dma_group = [&](uint32_t g, uint32_t slot) {
        const auto src = uniform_ptr(Vf + (size_t)g * KS * 1024);
        unsigned char *dst = smem + slot * SLOT;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t piece = (uint32_t)wave + 8u * p;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + piece * 1024 + lane16), (lptr_t)(dst + piece * 1024), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((gptr_t)(vnorm + (size_t)g * 32 + lane), (lptr_t)(dst + KS * 1024), 4, 0, 0);
    };
    dma_group(g0, 0);
    dma_group(min(g0 + 1, g1 - 1), 1);
    const __attribute__((address_space(1))) unsigned char *nsrc = nullptr;  // spread mode: source / slot of the group being fetched
    unsigned char *ndst = nullptr;
    const float *nnorm = nullptr;
    auto dma_piece = [&](int p) {
        const uint32_t piece = (uint32_t)wave + 8u * p;
        __builtin_amdgcn_global_load_lds((gptr_t)(nsrc + piece * 1024 + lane16), (lptr_t)(ndst + piece * 1024), 16, 0, 0);
    };

    float best = -__builtin_inff();
    uint32_t slot = 0;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
    for (uint32_t it = 0; it < n; ++it) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P + 1) : "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (STAG == 0) {
            if (!(ABL & 1)) dma_group(min(g0 + it + 2, g1 - 1), slot >= 1 ? slot - 1 : 2);
            else { asm volatile("s_nop 0"); }
        } else {
            const uint32_t gn = min(g0 + it + 2, g1 - 1);
            nsrc = uniform_ptr(Vf + (size_t)gn * KS * 1024);
            ndst = smem + (slot >= 1 ? slot - 1 : 2) * SLOT;
            nnorm = vnorm + (size_t)gn * 32 + lane;
        }
        const uint32_t base = lds0 + slot * SLOT + lane16;
        const uint32_t nbase = lds0 + slot * SLOT + KS * 1024 + 16 * fh;
        f32x4 nv[4];
        asm volatile("ds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:32\n\t"
                     "ds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:96"
                     : "=&v"(nv[0]), "=&v"(nv[1]), "=&v"(nv[2]), "=&v"(nv[3]) : "v"(nbase));
        u32x4 A[R];
        static_for<R>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            lds_read128<J * 1024>(A[J], base);
        });
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(nv[0]), "+v"(nv[1]), "+v"(nv[2]), "+v"(nv[3]) : "n"(R));
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = -0.5f * nv[r >> 2][r & 3];
        __builtin_amdgcn_sched_barrier(0);
        static_for<KS>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr int SL = J % R;
            constexpr int W = (KS - 1 - J) < (R - 1) ? (KS - 1 - J) : (R - 1);  // reads younger than fragment J still allowed in flight
            if constexpr (!(ABL & 4) || J < R) lgkm_wait<W>(A[SL]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ABL & 2))
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[SL]), __builtin_bit_cast(bf16x8, B[J]), acc, 0, 0, 0);
            else { acc[0] += __uint_as_float(A[SL].x ^ B[J].x); }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (J + R < KS && !(ABL & 4)) {
                lds_read128<(J + R) * 1024>(A[SL], base);
            }
            if constexpr (STAG == 1) {        // one piece every 8 k-steps, both halves at the same k-step
                if constexpr (J % 8 == 2) dma_piece(J / 8);
                if constexpr (J == KS - 3) __builtin_amdgcn_global_load_lds((gptr_t)nnorm, (lptr_t)(ndst + KS * 1024), 4, 0, 0);
            } else if constexpr (STAG == 2) { // waves 0-3 at k-steps 2 mod 8, their SIMD partners (waves 4-7) at 6 mod 8
                if constexpr (J % 8 == 2) { if (wave < 4) dma_piece(J / 8); }
                if constexpr (J % 8 == 6) { if (wave >= 4) dma_piece(J / 8); }
                if constexpr (J == KS - 1) __builtin_amdgcn_global_load_lds((gptr_t)nnorm, (lptr_t)(ndst + KS * 1024), 4, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (!(ABL & 8)) {
            float m = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
            best = fmaxf(best, m);
        } else {
            best = fmaxf(best, acc[0]);
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    best = fmaxf(best, __shfl_xor(best, 32));
    if (lane < 32) atomicMax(out + qt * 256 + wave * 32 + lane, fkey(best));
    if (stamps && tid == 0) {
        stamps[2 * b] = __builtin_amdgcn_s_memtime() - t0;
        stamps[2 * b + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}


// ---- variant: 4 waves (one per SIMD) x 64 queries each (two B sets in 384 VGPRs): half the LDS fragment reads per MFMA ----
template <int KS, int R>
__global__ __launch_bounds__(256, 1) void k_regq64(const unsigned char *__restrict__ Qf, const unsigned char *__restrict__ Vf,
                                                    const float *__restrict__ vnorm, uint32_t G, uint32_t S, uint32_t QT,
                                                    uint32_t *__restrict__ out, unsigned long long *__restrict__ stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    constexpr int P = KS / 4;
    constexpr uint32_t SLOT = KS * 1024 + 256;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t fh = lane >> 5;
    const uint32_t b = blockIdx.x, x = b & 7, jb = b >> 3;
    const uint32_t qt = jb % QT, s = (jb / QT) * 8 + x;
    const uint32_t per = (G + S - 1) / S;
    const uint32_t g0 = min(s * per, G), g1 = min(g0 + per, G);
    const uint32_t n = g1 - g0;
    if (n == 0) return;
    u32x4 B0[KS], B1[KS];
    {
        const unsigned char *qsrc = Qf + ((size_t)(qt * 8 + 2 * wave) * KS) * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < KS; ++j) { B0[j] = *reinterpret_cast<const u32x4 *>(qsrc + (size_t)j * 1024); B1[j] = *reinterpret_cast<const u32x4 *>(qsrc + (size_t)(KS + j) * 1024); }
    }
    const uint32_t lane16 = (uint32_t)lane * 16;
    auto dma_group = [&](uint32_t g, uint32_t slot) {
        const auto src = uniform_ptr(Vf + (size_t)g * KS * 1024);
        unsigned char *dst = smem + slot * SLOT;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t piece = (uint32_t)wave + 4u * p;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + piece * 1024 + lane16), (lptr_t)(dst + piece * 1024), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((gptr_t)(vnorm + (size_t)g * 32 + lane), (lptr_t)(dst + KS * 1024), 4, 0, 0);
    };
    dma_group(g0, 0);
    dma_group(min(g0 + 1, g1 - 1), 1);
    float best0 = -__builtin_inff(), best1 = best0;
    uint32_t slot = 0;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
    for (uint32_t it = 0; it < n; ++it) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P + 1) : "memory");
        __builtin_amdgcn_s_barrier();
        dma_group(min(g0 + it + 2, g1 - 1), slot >= 1 ? slot - 1 : 2);
        const uint32_t base = lds0 + slot * SLOT + lane16;
        const uint32_t nbase = lds0 + slot * SLOT + KS * 1024 + 16 * fh;
        f32x4 nv[4];
        asm volatile("ds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:32\n\t"
                     "ds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:96"
                     : "=&v"(nv[0]), "=&v"(nv[1]), "=&v"(nv[2]), "=&v"(nv[3]) : "v"(nbase));
        u32x4 A[R];
        static_for<R>([&](auto jc) { constexpr int J = decltype(jc)::value; lds_read128<J * 1024>(A[J], base); });
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(nv[0]), "+v"(nv[1]), "+v"(nv[2]), "+v"(nv[3]) : "n"(R));
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = -0.5f * nv[r >> 2][r & 3]; acc1[r] = acc0[r]; }
        __builtin_amdgcn_sched_barrier(0);
        static_for<KS>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr int SL = J % R;
            constexpr int W = (KS - 1 - J) < (R - 1) ? (KS - 1 - J) : (R - 1);
            lgkm_wait<W>(A[SL]);
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[SL]), __builtin_bit_cast(bf16x8, B0[J]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[SL]), __builtin_bit_cast(bf16x8, B1[J]), acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (J + R < KS) lds_read128<(J + R) * 1024>(A[SL], base);
            __builtin_amdgcn_sched_barrier(0);
        });
        float m0 = acc0[0], m1 = acc1[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) { m0 = fmaxf(m0, acc0[r]); m1 = fmaxf(m1, acc1[r]); }
        best0 = fmaxf(best0, m0);
        best1 = fmaxf(best1, m1);
        slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    best0 = fmaxf(best0, __shfl_xor(best0, 32));
    best1 = fmaxf(best1, __shfl_xor(best1, 32));
    if (lane < 32) { atomicMax(out + qt * 256 + wave * 64 + lane, fkey(best0)); atomicMax(out + qt * 256 + wave * 64 + 32 + lane, fkey(best1)); }
    if (stamps && tid == 0) { stamps[2 * b] = __builtin_amdgcn_s_memtime() - t0; stamps[2 * b + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

// ---- variant: v_mfma_f32_16x16x32_bf16.  Layout16: unit (block of 16 rows, k-step of 32 elements, lane l) = 8 bf16 of row
// 16 blk + (l & 15), elements 32 j + 8 (l >> 4) + [0, 8); a 32-row group = [row block 0..1][KS/2 k-steps][1 KiB].
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int KS /* 16-element k-steps, as above */, int R, int SPR = 0>
__global__ __launch_bounds__(512, 2) void k_regq16(const unsigned char *__restrict__ Qf, const unsigned char *__restrict__ Vf,
                                                    const float *__restrict__ vnorm, uint32_t G, uint32_t S, uint32_t QT,
                                                    uint32_t *__restrict__ out, unsigned long long *__restrict__ stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    constexpr int P = KS / 8;
    constexpr int K2 = KS / 2;  // 32-element k-steps
    constexpr uint32_t SLOT = KS * 1024 + 256;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t fq = lane >> 4;
    const uint32_t b = blockIdx.x, x = b & 7, jb = b >> 3;
    const uint32_t qt = jb % QT, s = (jb / QT) * 8 + x;
    const uint32_t per = (G + S - 1) / S;
    const uint32_t g0 = min(s * per, G), g1 = min(g0 + per, G);
    const uint32_t n = g1 - g0;
    if (n == 0) return;
    u32x4 B[KS];  // [query block 0..1][K2]
    {
        const unsigned char *qsrc = Qf + ((size_t)(qt * 8 + wave) * KS) * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < KS; ++j) B[j] = *reinterpret_cast<const u32x4 *>(qsrc + (size_t)j * 1024);
    }
    const uint32_t lane16 = (uint32_t)lane * 16;
    auto dma_group = [&](uint32_t g, uint32_t slot) {
        const auto src = uniform_ptr(Vf + (size_t)g * KS * 1024);
        unsigned char *dst = smem + slot * SLOT;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t piece = (uint32_t)wave + 8u * p;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + piece * 1024 + lane16), (lptr_t)(dst + piece * 1024), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((gptr_t)(vnorm + (size_t)g * 32 + lane), (lptr_t)(dst + KS * 1024), 4, 0, 0);
    };
    dma_group(g0, 0);
    dma_group(min(g0 + 1, g1 - 1), 1);
    float best0 = -__builtin_inff(), best1 = best0;
    uint32_t slot = 0;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
    for (uint32_t it = 0; it < n; ++it) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P + 1) : "memory");
        __builtin_amdgcn_s_barrier();
        const uint32_t gn = min(g0 + it + 2, g1 - 1);
        const uint32_t nslot = slot >= 1 ? slot - 1 : 2;
        if constexpr (!SPR) dma_group(gn, nslot);
        const auto nsrc = uniform_ptr(Vf + (size_t)gn * KS * 1024);
        unsigned char *ndst = smem + nslot * SLOT;
        const uint32_t base = lds0 + slot * SLOT + lane16;
        const uint32_t nbase = lds0 + slot * SLOT + KS * 1024 + 16 * fq;
        f32x4v nv[2];
        asm volatile("ds_read_b128 %0, %2 offset:0\n\tds_read_b128 %1, %2 offset:64" : "=&v"(nv[0]), "=&v"(nv[1]) : "v"(nbase));
        u32x4 A[R];
        static_for<R>([&](auto jc) { constexpr int J = decltype(jc)::value; lds_read128<J * 1024>(A[J], base); });
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(nv[0]), "+v"(nv[1]) : "n"(R));
        f32x4v acc[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[rb][0][r] = -0.5f * nv[rb][r]; acc[rb][1][r] = acc[rb][0][r]; }
        __builtin_amdgcn_sched_barrier(0);
        static_for<KS>([&](auto jc) {  // fragment J = (row block J / K2, k-step J % K2)
            constexpr int J = decltype(jc)::value;
            constexpr int SL = J % R;
            constexpr int RB = J / K2, JJ = J % K2;
            constexpr int W = (KS - 1 - J) < (R - 1) ? (KS - 1 - J) : (R - 1);
            lgkm_wait<W>(A[SL]);
            __builtin_amdgcn_sched_barrier(0);
            acc[RB][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[SL]), __builtin_bit_cast(bf16x8, B[JJ]), acc[RB][0], 0, 0, 0);
            acc[RB][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[SL]), __builtin_bit_cast(bf16x8, B[K2 + JJ]), acc[RB][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (J + R < KS) lds_read128<(J + R) * 1024>(A[SL], base);
            if constexpr (SPR) {
                if constexpr (J % 8 == 2) {
                    const uint32_t piece = (uint32_t)wave + 8u * (J / 8);
                    __builtin_amdgcn_global_load_lds((gptr_t)(nsrc + piece * 1024 + lane16), (lptr_t)(ndst + piece * 1024), 16, 0, 0);
                }
                if constexpr (J == KS - 3) __builtin_amdgcn_global_load_lds((gptr_t)(vnorm + (size_t)gn * 32 + lane), (lptr_t)(ndst + KS * 1024), 4, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        float m0 = acc[0][0][0], m1 = acc[0][1][0];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) { m0 = fmaxf(m0, acc[rb][0][r]); m1 = fmaxf(m1, acc[rb][1][r]); }
        best0 = fmaxf(best0, m0);
        best1 = fmaxf(best1, m1);
        slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    best0 = fmaxf(best0, __shfl_xor(best0, 16)); best0 = fmaxf(best0, __shfl_xor(best0, 32));
    best1 = fmaxf(best1, __shfl_xor(best1, 16)); best1 = fmaxf(best1, __shfl_xor(best1, 32));
    if (lane < 16) { atomicMax(out + qt * 256 + wave * 32 + lane, fkey(best0)); atomicMax(out + qt * 256 + wave * 32 + 16 + lane, fkey(best1)); }
    if (stamps && tid == 0) { stamps[2 * b] = __builtin_amdgcn_s_memtime() - t0; stamps[2 * b + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

// ---- round 6: 16x16x32, FOUR waves (one per SIMD) x 64 queries each (four B sets of 16 queries: 384 registers of the unified 512-entry
// file), FLOW-style pipeline of the product kernel: the A ring and the start values run ahead of the barrier, every wave drains its
// DMA ahead of it, the DMA pieces are spread over the fragments, the row block's reduction sits 3 fragments behind its last MFMA.
// Half the LDS fragment reads per MFMA of the 8 x 32 form (4 MFMAs per ds_read_b128 instead of 2) -- the question is whether ONE
// wave per SIMD keeps the matrix pipe fed once nothing of its bookkeeping is exposed (round 2's k_regq64 did not: 50.4 cycles per MFMA).
template <int KS, int R, int DSTEP>
__global__ __launch_bounds__(256, 1) void k_regq16w4(const unsigned char *__restrict__ Qf, const unsigned char *__restrict__ Vf,
                                                      const float *__restrict__ vnorm, uint32_t G, uint32_t S, uint32_t QT,
                                                      uint32_t *__restrict__ out, unsigned long long *__restrict__ stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    constexpr int NW = 4, P = KS / NW, K2 = KS / 2, F = KS;
    constexpr uint32_t SLOT = KS * 1024 + 256;
    static_assert(F % R == 0 && 2 + DSTEP * (P - 1) < F, "ring wrap / DMA placement");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t fq = lane >> 4;
    const uint32_t b = blockIdx.x, x = b & 7, jb = b >> 3;
    const uint32_t qt = jb % QT, s = (jb / QT) * 8 + x;
    const uint32_t per = (G + S - 1) / S;
    const uint32_t g0 = min(s * per, G), g1 = min(g0 + per, G);
    const uint32_t n = g1 - g0;
    if (n == 0) return;
    u32x4 B[4][K2];
    {
        const unsigned char *qsrc = Qf + ((size_t)(qt * 16 + 4 * wave) * K2) * 1024 + lane * 16;
#pragma unroll
        for (int sct = 0; sct < 4; ++sct)
#pragma unroll
            for (int j = 0; j < K2; ++j) B[sct][j] = *reinterpret_cast<const u32x4 *>(qsrc + (size_t)(sct * K2 + j) * 1024);
    }
    const uint32_t lane16 = (uint32_t)lane * 16, lane4 = (uint32_t)lane * 4;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
    auto dma16 = [&](uint32_t voff, const __attribute__((address_space(1))) unsigned char *sbase, uint32_t lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
    };
    auto dma4 = [&](uint32_t voff, const __attribute__((address_space(1))) unsigned char *sbase, uint32_t lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
    };
    auto dma_unit = [&](uint32_t g, uint32_t slot) {
        const auto src = uniform_ptr(Vf + (size_t)g * KS * 1024);
        const uint32_t dst = lds0 + slot * SLOT;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t piece = (uint32_t)wave + (uint32_t)NW * p;
            dma16(lane16, src + piece * 1024, dst + piece * 1024);
        }
        if (wave == 0) dma4(lane4, uniform_ptr(reinterpret_cast<const unsigned char *>(vnorm + (size_t)g * 32)), dst + KS * 1024);
    };
    dma_unit(g0, 0);
    dma_unit(min(g0 + 1, g1 - 1), 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float best[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    uint32_t slot = 0;
    u32x4 A[R];
    f32x4v nv0;
    f32x4v acc[2][4];
    {
        const uint32_t base = lds0 + lane16;
        asm volatile("ds_read_b128 %0, %1" : "=&v"(nv0) : "v"(lds0 + KS * 1024 + 16 * fq));
        static_for<R>([&](auto jc) { constexpr int J = decltype(jc)::value; lds_read128<J * 1024>(A[J], base); });
    }
#pragma unroll
    for (int sct = 0; sct < 4; ++sct) acc[1][sct] = f32x4v{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    auto reduce_rb = [&](auto rbc) {
        constexpr int rb = decltype(rbc)::value;
#pragma unroll
        for (int sct = 0; sct < 4; ++sct)
            best[sct] = fmaxf(best[sct], fmaxf(fmaxf(acc[rb][sct][0], acc[rb][sct][1]), fmaxf(acc[rb][sct][2], acc[rb][sct][3])));
    };
    for (uint32_t it = 0; it < n; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const uint32_t un = min(g0 + it + 2, g1 - 1);
        const auto nsrc = uniform_ptr(Vf + (size_t)un * KS * 1024 + (size_t)wave * 1024);
        const uint32_t ndst = lds0 + (slot >= 1 ? slot - 1 : 2) * SLOT + (uint32_t)wave * 1024;
        const auto nnorm = uniform_ptr(reinterpret_cast<const unsigned char *>(vnorm + (size_t)un * 32));
        const uint32_t base = lds0 + slot * SLOT + lane16;
        const uint32_t nslot = slot == 2 ? 0 : slot + 1;
        const uint32_t xbase = lds0 + nslot * SLOT + lane16;
        const uint32_t nbase = lds0 + slot * SLOT + F * 1024 + 16 * fq;
        const uint32_t xnbase = lds0 + nslot * SLOT + F * 1024 + 16 * fq;
        __builtin_amdgcn_sched_barrier(0);
        static_for<F>([&](auto fcn) {
            constexpr int FI = decltype(fcn)::value;
            constexpr int RB = FI / K2, JJ = FI % K2;
            constexpr int SL = FI % R;
            if constexpr (FI == 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(A[SL]), "+v"(nv0) : "n"(R - 1));
            else if constexpr (FI == K2) asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(A[SL]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]) : "n"(R - 1));
            else lgkm_wait<R - 1>(A[SL]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FI == 0) {
#pragma unroll
                for (int sct = 0; sct < 4; ++sct) acc[0][sct] = -0.5f * nv0;  // (the product's mirror stores the start values ready-made)
            }
            if constexpr (FI == K2) {
#pragma unroll
                for (int sct = 0; sct < 4; ++sct) acc[1][sct] = -0.5f * acc[1][sct];
            }
#pragma unroll
            for (int sct = 0; sct < 4; ++sct)
                acc[RB][sct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[SL]), __builtin_bit_cast(bf16x8, B[sct][JJ]), acc[RB][sct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FI == 6)
                asm volatile("ds_read_b128 %0, %4 offset:64\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:64"
                             : "=&v"(acc[1][0]), "=&v"(acc[1][1]), "=&v"(acc[1][2]), "=&v"(acc[1][3]) : "v"(nbase));
            if constexpr (FI == F - R - 1) asm volatile("ds_read_b128 %0, %1" : "=&v"(nv0) : "v"(xnbase));
            if constexpr (FI + R < F) lds_read128<(FI + R) * 1024>(A[SL], base);
            else lds_read128<(FI + R - F) * 1024>(A[SL], xbase);
            if constexpr (FI >= 2 && (FI - 2) % DSTEP == 0 && (FI - 2) / DSTEP < P) {
                constexpr uint32_t POFF = (uint32_t)NW * ((FI - 2) / DSTEP) * 1024;
                dma16(lane16, nsrc + POFF, ndst + POFF);
            }
            if constexpr (FI == F - 6) {
                if (wave == 0) dma4(lane4, nnorm, ndst - (uint32_t)wave * 1024 + F * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FI == K2 + 3) reduce_rb(std::integral_constant<int, 0>{});
            if constexpr (FI == 3) reduce_rb(std::integral_constant<int, 1>{});
        });
        slot = nslot;
    }
    reduce_rb(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(nv0));
#pragma unroll
    for (int sct = 0; sct < 4; ++sct) {
        float v = best[sct];
        v = fmaxf(v, __shfl_xor(v, 16));
        v = fmaxf(v, __shfl_xor(v, 32));
        if (lane < 16) atomicMax(out + qt * 256 + wave * 64 + sct * 16 + lane, fkey(v));
    }
    if (stamps && tid == 0) { stamps[2 * b] = __builtin_amdgcn_s_memtime() - t0; stamps[2 * b + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

__device__ __forceinline__ void pin_agpr(u32x4 &x) { asm volatile("" : "+a"(x)); }
__device__ __forceinline__ void pin_vgpr(u32x4 &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void tie4(f32x4v &a, f32x4v &b, f32x4v &c, f32x4v &d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
template <bool B_IN_AGPR, bool NOP_FIRST>
__device__ __forceinline__ void mfma_asm(f32x4v &acc, const u32x4 &a, const u32x4 &b) {
    if constexpr (B_IN_AGPR) {
        if constexpr (NOP_FIRST) asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
    } else {
        if constexpr (NOP_FIRST) asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
}
// ---- the same with the MFMAs by inline asm and the B fragments PINNED: NA of the 96 in AGPRs ("a" constraints), the rest in VGPRs --
// hipcc's allocation of the intrinsic form shuffles B fragments between AGPRs inside the loop (4 v_accvgpr_mov per fragment) and spills.
// ---- (base text:) 16x16x32, FOUR waves (one per SIMD) x 64 queries each (four B sets of 16 queries: 384 registers of the unified 512-entry
// file), FLOW-style pipeline of the product kernel: the A ring and the start values run ahead of the barrier, every wave drains its
// DMA ahead of it, the DMA pieces are spread over the fragments, the row block's reduction sits 3 fragments behind its last MFMA.
// Half the LDS fragment reads per MFMA of the 8 x 32 form (4 MFMAs per ds_read_b128 instead of 2) -- the question is whether ONE
// wave per SIMD keeps the matrix pipe fed once nothing of its bookkeeping is exposed (round 2's k_regq64 did not: 50.4 cycles per MFMA).
template <int KS, int R, int DSTEP, int NA, int HZ = 0>
__global__ __launch_bounds__(256, 1) void k_regq16w4a(const unsigned char *__restrict__ Qf, const unsigned char *__restrict__ Vf,
                                                      const float *__restrict__ vnorm, uint32_t G, uint32_t S, uint32_t QT,
                                                      uint32_t *__restrict__ out, unsigned long long *__restrict__ stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    constexpr int NW = 4, P = KS / NW, K2 = KS / 2, F = KS;
    constexpr uint32_t SLOT = KS * 1024 + 256;
    static_assert(F % R == 0 && 2 + DSTEP * (P - 1) < F, "ring wrap / DMA placement");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t fq = lane >> 4;
    const uint32_t b = blockIdx.x, x = b & 7, jb = b >> 3;
    const uint32_t qt = jb % QT, s = (jb / QT) * 8 + x;
    const uint32_t per = (G + S - 1) / S;
    const uint32_t g0 = min(s * per, G), g1 = min(g0 + per, G);
    const uint32_t n = g1 - g0;
    if (n == 0) return;
    u32x4 B[4][K2];
    {
        const unsigned char *qsrc = Qf + ((size_t)(qt * 16 + 4 * wave) * K2) * 1024 + lane * 16;
#pragma unroll
        for (int sct = 0; sct < 4; ++sct)
#pragma unroll
            for (int j = 0; j < K2; ++j) B[sct][j] = *reinterpret_cast<const u32x4 *>(qsrc + (size_t)(sct * K2 + j) * 1024);
        static_for<4 * K2>([&](auto fc) {  // pin every fragment to its register file
            constexpr int FF = decltype(fc)::value;
            if constexpr (FF < NA) { if constexpr (!(HZ & 64)) pin_agpr(B[FF / K2][FF % K2]); }
            else if constexpr (!(HZ & 32)) pin_vgpr(B[FF / K2][FF % K2]);
        });
    }
    const uint32_t lane16 = (uint32_t)lane * 16, lane4 = (uint32_t)lane * 4;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
    auto dma16 = [&](uint32_t voff, const __attribute__((address_space(1))) unsigned char *sbase, uint32_t lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
    };
    auto dma4 = [&](uint32_t voff, const __attribute__((address_space(1))) unsigned char *sbase, uint32_t lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
    };
    auto dma_unit = [&](uint32_t g, uint32_t slot) {
        const auto src = uniform_ptr(Vf + (size_t)g * KS * 1024);
        const uint32_t dst = lds0 + slot * SLOT;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t piece = (uint32_t)wave + (uint32_t)NW * p;
            dma16(lane16, src + piece * 1024, dst + piece * 1024);
        }
        if (wave == 0) dma4(lane4, uniform_ptr(reinterpret_cast<const unsigned char *>(vnorm + (size_t)g * 32)), dst + KS * 1024);
    };
    dma_unit(g0, 0);
    dma_unit(min(g0 + 1, g1 - 1), 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float best[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    uint32_t slot = 0;
    u32x4 A[R];
    f32x4v nv0;
    f32x4v acc[2][4];
    {
        const uint32_t base = lds0 + lane16;
        asm volatile("ds_read_b128 %0, %1" : "=&v"(nv0) : "v"(lds0 + KS * 1024 + 16 * fq));
        static_for<R>([&](auto jc) { constexpr int J = decltype(jc)::value; lds_read128<J * 1024>(A[J], base); });
    }
#pragma unroll
    for (int sct = 0; sct < 4; ++sct) acc[1][sct] = f32x4v{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    auto reduce_rb = [&](auto rbc) {
        constexpr int rb = decltype(rbc)::value;
        // The accumulators are written by inline-asm MFMAs: hipcc knows no MFMA there and inserts no wait states in front of a VALU read
        // or copy of their results (it sank this whole reduction to the loop latch, right behind the unit's last MFMA: wrong maxima).
        // This empty asm redefines them HERE, 3 fragments = 12 MFMA issues behind the row block's last MFMA: nothing can read them earlier.
        if constexpr (!(HZ & 16)) tie4(acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]);
#pragma unroll
        for (int sct = 0; sct < 4; ++sct)
            best[sct] = fmaxf(best[sct], fmaxf(fmaxf(acc[rb][sct][0], acc[rb][sct][1]), fmaxf(acc[rb][sct][2], acc[rb][sct][3])));
    };
    for (uint32_t it = 0; it < n; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const uint32_t un = min(g0 + it + 2, g1 - 1);
        const auto nsrc = uniform_ptr(Vf + (size_t)un * KS * 1024 + (size_t)wave * 1024);
        const uint32_t ndst = lds0 + (slot >= 1 ? slot - 1 : 2) * SLOT + (uint32_t)wave * 1024;
        const auto nnorm = uniform_ptr(reinterpret_cast<const unsigned char *>(vnorm + (size_t)un * 32));
        const uint32_t base = lds0 + slot * SLOT + lane16;
        const uint32_t nslot = slot == 2 ? 0 : slot + 1;
        const uint32_t xbase = lds0 + nslot * SLOT + lane16;
        const uint32_t nbase = lds0 + slot * SLOT + F * 1024 + 16 * fq;
        const uint32_t xnbase = lds0 + nslot * SLOT + F * 1024 + 16 * fq;
        __builtin_amdgcn_sched_barrier(0);
        static_for<F>([&](auto fcn) {
            constexpr int FI = decltype(fcn)::value;
            constexpr int RB = FI / K2, JJ = FI % K2;
            constexpr int SL = FI % R;
            if constexpr (FI == 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(A[SL]), "+v"(nv0) : "n"(R - 1));
            else if constexpr (FI == K2) asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(A[SL]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]) : "n"(R - 1));
            else lgkm_wait<R - 1>(A[SL]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FI == 0) {
#pragma unroll
                for (int sct = 0; sct < 4; ++sct) acc[0][sct] = -0.5f * nv0;  // (the product's mirror stores the start values ready-made)
            }
            if constexpr (FI == K2) {
#pragma unroll
                for (int sct = 0; sct < 4; ++sct) acc[1][sct] = -0.5f * acc[1][sct];
            }
            static_for<4>([&](auto sc) {
                constexpr int SCT = decltype(sc)::value;
                // (VALU wrote the start values just above at FI == 0 / K2: hipcc does not see an MFMA in the asm, so the wait states are ours)
                if constexpr (HZ & 8) acc[RB][SCT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[SL]), __builtin_bit_cast(bf16x8, B[SCT][JJ]), acc[RB][SCT], 0, 0, 0);
                else mfma_asm<(SCT * K2 + JJ < NA), (JJ == 0 || (HZ & 2))>(acc[RB][SCT], A[SL], B[SCT][JJ]);
            });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FI == 6)
                asm volatile("ds_read_b128 %0, %4 offset:64\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:64"
                             : "=&v"(acc[1][0]), "=&v"(acc[1][1]), "=&v"(acc[1][2]), "=&v"(acc[1][3]) : "v"(nbase));
            if constexpr (FI == F - R - 1) asm volatile("ds_read_b128 %0, %1" : "=&v"(nv0) : "v"(xnbase));
            if constexpr (HZ & 4) asm volatile("s_nop 7\n\ts_nop 7");
            if constexpr (FI + R < F) lds_read128<(FI + R) * 1024>(A[SL], base);
            else lds_read128<(FI + R - F) * 1024>(A[SL], xbase);
            if constexpr (FI >= 2 && (FI - 2) % DSTEP == 0 && (FI - 2) / DSTEP < P) {
                constexpr uint32_t POFF = (uint32_t)NW * ((FI - 2) / DSTEP) * 1024;
                dma16(lane16, nsrc + POFF, ndst + POFF);
            }
            if constexpr (FI == F - 6) {
                if (wave == 0) dma4(lane4, nnorm, ndst - (uint32_t)wave * 1024 + F * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FI == K2 + 3) { if constexpr (HZ & 1) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7"); reduce_rb(std::integral_constant<int, 0>{}); }
            if constexpr (FI == 3) { if constexpr (HZ & 1) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7"); reduce_rb(std::integral_constant<int, 1>{}); }
        });
        slot = nslot;
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");  // (the last unit's MFMAs have left the pipe)
    reduce_rb(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(nv0));
#pragma unroll
    for (int sct = 0; sct < 4; ++sct) {
        float v = best[sct];
        v = fmaxf(v, __shfl_xor(v, 16));
        v = fmaxf(v, __shfl_xor(v, 32));
        if (lane < 16) atomicMax(out + qt * 256 + wave * 64 + sct * 16 + lane, fkey(v));
    }
    if (stamps && tid == 0) { stamps[2 * b] = __builtin_amdgcn_s_memtime() - t0; stamps[2 * b + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

// ---- variant: 16x16x32, STAGGERED halves.  Ring of 6 half-units (24 fragments each), a barrier every 24 fragments placed
// mid-stream (before fragments 12 and 36 of a unit); waves 4-7 (the SIMD partners of waves 0-3) run 24 fragments behind, so a
// wave's epilogue / unit turn-over falls into the steady MFMA stream of its partner.  The A-read ring never drains.
// Barrier i (per wave, counted from 0) validates half-unit i + 1; after it the wave issues its pieces of half-unit i + 4.
template <int KS, int R, int STAGGER>
__global__ __launch_bounds__(512, 2) void k_regq16s(const unsigned char *__restrict__ Qf, const unsigned char *__restrict__ Vf,
                                                     const float *__restrict__ vnorm, uint32_t G, uint32_t S, uint32_t QT,
                                                     uint32_t *__restrict__ out, unsigned long long *__restrict__ stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    static_assert(KS == 48 && 48 % R == 0, "geometry");
    constexpr int F = 48, H = 24, K2 = 24;
    constexpr uint32_t HSLOT = H * 1024 + 256;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = STAGGER && wave >= 4;
    const uint32_t fq = lane >> 4;
    const uint32_t b = blockIdx.x, x = b & 7, jb = b >> 3;
    const uint32_t qt = jb % QT, s = (jb / QT) * 8 + x;
    const uint32_t per = (G + S - 1) / S;
    const uint32_t g0 = min(s * per, G), g1 = min(g0 + per, G);
    const uint32_t n = g1 - g0;
    if (n == 0) return;
    const uint32_t hmax = 2 * n - 1;
    u32x4 B[KS];
    {
        const unsigned char *qsrc = Qf + ((size_t)(qt * 8 + wave) * KS) * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < KS; ++j) B[j] = *reinterpret_cast<const u32x4 *>(qsrc + (size_t)j * 1024);
    }
    const uint32_t lane16 = (uint32_t)lane * 16;
    const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
    // DMA of half-unit h (clamped source) into half-slot h % 6: 3 pieces per wave (+ the unit's norms by wave 0 for even h)
    const __attribute__((address_space(1))) unsigned char *nsrc = nullptr;
    unsigned char *ndst = nullptr;
    auto dma_setup = [&](uint32_t h) {
        const uint32_t hc = min(h, hmax);
        nsrc = uniform_ptr(Vf + ((size_t)(g0 + (hc >> 1)) * F + (size_t)(hc & 1) * H) * 1024);
        ndst = smem + (h % 6) * HSLOT;
        if (wave == 0 && !(h & 1))
            __builtin_amdgcn_global_load_lds((gptr_t)(vnorm + (size_t)(g0 + (hc >> 1)) * 32 + lane), (lptr_t)(ndst + H * 1024), 4, 0, 0);
    };
    auto dma_piece = [&](int p) {
        const uint32_t piece = (uint32_t)wave + 8u * p;
        __builtin_amdgcn_global_load_lds((gptr_t)(nsrc + piece * 1024 + lane16), (lptr_t)(ndst + piece * 1024), 16, 0, 0);
    };
    for (uint32_t h = 0; h < 4; ++h) { dma_setup(h); dma_piece(0); dma_piece(1); dma_piece(2); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t bi = 0;  // barriers executed
    auto barrier_hook = [&]() {
        if (wave == 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        dma_setup(bi + 4);
        bi += 1;
    };
    if (late) { barrier_hook(); dma_piece(0); dma_piece(1); }

    float best0 = -__builtin_inff(), best1 = best0;
    // first unit: norms + the first R fragments
    uint32_t base = lds0 + lane16, nb = lds0 + H * 1024 + 16 * fq;
    f32x4v nv[2];
    asm volatile("ds_read_b128 %0, %2 offset:0\n\tds_read_b128 %1, %2 offset:64" : "=&v"(nv[0]), "=&v"(nv[1]) : "v"(nb));
    u32x4 A[R];
    static_for<R>([&](auto jc) { constexpr int J = decltype(jc)::value; lds_read128<J * 1024>(A[J], base); });
    for (uint32_t t = 0; t < n; ++t) {
        const uint32_t sn = (2 * (t + 1)) % 6;
        const uint32_t base_next = lds0 + sn * HSLOT + lane16, nb_next = lds0 + sn * HSLOT + H * 1024 + 16 * fq;
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(nv[0]), "+v"(nv[1]) : "n"(R));
        f32x4v acc[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[rb][0][r] = -0.5f * nv[rb][r]; acc[rb][1][r] = acc[rb][0][r]; }
        __builtin_amdgcn_sched_barrier(0);
        static_for<F>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr int SL = J % R;
            constexpr int RB = J / K2, JJ = J % K2;
            if constexpr (J == 12 || J == 36) {
                if (!(t == 0 && J == 12 && !STAGGER)) barrier_hook();  // (unstaggered: everything up to half-unit 3 is resident at the start)
            }
            lgkm_wait<R - 1>(A[SL]);
            __builtin_amdgcn_sched_barrier(0);
            acc[RB][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[SL]), __builtin_bit_cast(bf16x8, B[JJ]), acc[RB][0], 0, 0, 0);
            acc[RB][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[SL]), __builtin_bit_cast(bf16x8, B[K2 + JJ]), acc[RB][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int N2 = J + R;
            if constexpr (N2 < H) lds_read128<N2 * 1024>(A[SL], base);
            else if constexpr (N2 < F) lds_read128<(int)HSLOT + (N2 - H) * 1024>(A[SL], base);
            else lds_read128<(N2 - F) * 1024>(A[SL], base_next);
            if constexpr (J == 14 || J == 38) { if (bi > 0) dma_piece(0); }
            if constexpr (J == 22 || J == 46) { if (bi > 0) dma_piece(1); }
            if constexpr (J == 30 || J == 6) { if (bi > 0) dma_piece(2); }
            if constexpr (J == 40) {
                asm volatile("ds_read_b128 %0, %2 offset:0\n\tds_read_b128 %1, %2 offset:64" : "=&v"(nv[0]), "=&v"(nv[1]) : "v"(nb_next));
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        float m0 = acc[0][0][0], m1 = acc[0][1][0];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) { m0 = fmaxf(m0, acc[rb][0][r]); m1 = fmaxf(m1, acc[rb][1][r]); }
        best0 = fmaxf(best0, m0);
        best1 = fmaxf(best1, m1);
        base = base_next;
    }
    if (STAGGER && !late) { if (wave == 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    best0 = fmaxf(best0, __shfl_xor(best0, 16)); best0 = fmaxf(best0, __shfl_xor(best0, 32));
    best1 = fmaxf(best1, __shfl_xor(best1, 16)); best1 = fmaxf(best1, __shfl_xor(best1, 32));
    if (lane < 16) { atomicMax(out + qt * 256 + wave * 32 + lane, fkey(best0)); atomicMax(out + qt * 256 + wave * 32 + 16 + lane, fkey(best1)); }
    if (stamps && tid == 0) { stamps[2 * b] = __builtin_amdgcn_s_memtime() - t0; stamps[2 * b + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

// Layout16 generator (rows and queries alike)
__global__ void k_gen_frag16(uint64_t seed, uint64_t nrows, uint32_t dim, uint32_t K2, uint64_t units, unsigned char *out) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const uint32_t l = (uint32_t)(u & 63);
    const uint64_t bj = u >> 6;
    const uint32_t j = (uint32_t)(bj % K2);
    const uint64_t row = (bj / K2) * 16 + (l & 15);
    const uint32_t k0 = 32 * j + 8 * (l >> 4);
    u32x4 w = {0, 0, 0, 0};
    if (row < nrows)
        for (int e = 0; e < 8; ++e) {
            const uint32_t k = k0 + e;
            const uint16_t h = k < dim ? f2bf(elem(seed, row, k, dim)) : 0;
            w[e >> 1] |= (uint32_t)h << (16 * (e & 1));
        }
    *reinterpret_cast<u32x4 *>(out + u * 16) = w;
}

__global__ void k_ref(const unsigned char *Qf, const unsigned char *Vf, const float *vnorm, uint64_t nrows, uint32_t KS, uint32_t *out) {
    const uint32_t q = blockIdx.x;
    float best = -__builtin_inff();
    for (uint64_t r = threadIdx.x; r < nrows; r += blockDim.x) {
        float acc = 0.0f;
        for (uint32_t j = 0; j < KS; ++j)
            for (uint32_t hh = 0; hh < 2; ++hh) {
                const uint16_t *a = reinterpret_cast<const uint16_t *>(Vf + (((r >> 5) * KS + j) * 64 + (r & 31) + 32 * hh) * 16);
                const uint16_t *bq = reinterpret_cast<const uint16_t *>(Qf + ((((uint64_t)q >> 5) * KS + j) * 64 + (q & 31) + 32 * hh) * 16);
                for (int e = 0; e < 8; ++e) acc += bf2f(a[e]) * bf2f(bq[e]);
            }
        best = fmaxf(best, acc - 0.5f * vnorm[r]);
    }
    atomicMax(out + q, fkey(best));
}

template <int KS, int R, int ABL, int STAG>
static double run(const char *name, const unsigned char *Qf, const unsigned char *Vf, const float *vn, uint32_t G, uint32_t nq, uint32_t *out, int reps) {
    static unsigned long long *stamps = nullptr;
    if (!stamps) CK(hipMalloc(&stamps, 256 * 16));
    const uint32_t QT = nq / 256, S = 256 / QT;
    const size_t lds = 3 * (KS * 1024 + 256);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_regq<KS, R, ABL, STAG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipMemsetAsync(out, 0, nq * 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_regq<KS, R, ABL, STAG>), dim3(QT * S), dim3(512), lds, 0, Qf, Vf, vn, G, S, QT, out, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        if (rep) sum += ms;
    }
    std::vector<unsigned long long> st(512);
    CK(hipMemcpy(st.data(), stamps, 256 * 16, hipMemcpyDeviceToHost));
    double cyc = 0, clk = 0;
    for (int i = 0; i < 256; ++i) { cyc += (double)st[2 * i] / 256; clk += (double)st[2 * i] / (double)st[2 * i + 1] * 100.0 / 256; }
    const double flops = 2.0 * nq * (double)G * 32 * KS * 16;
    printf("   [in-kernel: %.0f kcycles per workgroup, clock %.0f MHz, %.1f cycles per MFMA per SIMD] ", cyc / 1e3, clk, cyc / ((double)((G + S - 1) / S) * KS * 2));
    printf("%-44s G=%u: best %.3f ms (%.1f TFLOP/s), mean %.3f ms (%.1f TFLOP/s)\n", name, G, best, flops / best / 1e9,
           sum / (reps - 1), flops / (sum / (reps - 1)) / 1e9);
    fflush(stdout);
    return best;
}

template <class K>
static double run_k(const char *name, K kern, int threads, const unsigned char *Qf, const unsigned char *Vf, const float *vn, uint32_t G, uint32_t nq, uint32_t *out, int reps) {
    constexpr int KS = 48;
    static unsigned long long *stamps = nullptr;
    if (!stamps) CK(hipMalloc(&stamps, 256 * 16));
    const uint32_t QT = nq / 256, S = 256 / QT;
    const size_t lds = 6 * (24 * 1024 + 256);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipMemsetAsync(out, 0, nq * 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(QT * S), dim3(threads), lds, 0, Qf, Vf, vn, G, S, QT, out, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        if (rep) sum += ms;
    }
    std::vector<unsigned long long> st(512);
    CK(hipMemcpy(st.data(), stamps, 256 * 16, hipMemcpyDeviceToHost));
    double cyc = 0, clk = 0;
    for (int i = 0; i < 256; ++i) { cyc += (double)st[2 * i] / 256; clk += (double)st[2 * i] / (double)st[2 * i + 1] * 100.0 / 256; }
    const double flops = 2.0 * nq * (double)G * 32 * KS * 16;
    printf("   [in-kernel: %.0f kcycles per workgroup, clock %.0f MHz, %.1f cycles per 32x32x16-equivalent MFMA per SIMD] ", cyc / 1e3, clk, cyc / ((double)((G + S - 1) / S) * KS * 2));
    printf("%-44s G=%u: best %.3f ms (%.1f TFLOP/s), mean %.3f ms (%.1f TFLOP/s)\n", name, G, best, flops / best / 1e9,
           sum / (reps - 1), flops / (sum / (reps - 1)) / 1e9);
    fflush(stdout);
    return best;
}

static double check(const char *what, const uint32_t *out, const uint32_t *ref, uint32_t nq) {
    std::vector<uint32_t> a(nq), c(nq);
    CK(hipMemcpy(a.data(), out, nq * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c.data(), ref, nq * 4, hipMemcpyDeviceToHost));
    auto unkey = [](uint32_t k) { uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; float f; memcpy(&f, &u, 4); return f; };
    double worst = 0;
    uint32_t bad = 0;
    for (uint32_t i = 0; i < nq; ++i) {
        const double df = std::fabs(unkey(a[i]) - unkey(c[i]));
        worst = std::max(worst, df);
        if (df > 0.01) { if (bad < 12) printf("   query %u: kernel %.4f ref %.4f\n", i, unkey(a[i]), unkey(c[i])); ++bad; }
    }
    printf("%s: max |kernel - ref| over %u queries = %.3g (values ~ %.3f), %u queries off\n", what, nq, worst, unkey(c[0]), bad);
    return worst;
}

int main(int argc, char **argv) {
    constexpr int KS = 48;
    const uint32_t dim = 768, nq = 1024;
    const uint64_t nbig = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000ull;
    const uint64_t nsmall = 40000;
    const uint64_t gbig = (nbig + 31) / 32, gsmall = (nsmall + 31) / 32;
    unsigned char *Qf, *Vf;
    float *vn, *qn;
    uint32_t *out, *ref;
    CK(hipMalloc(&Qf, (size_t)nq / 32 * KS * 1024));
    CK(hipMalloc(&Vf, gbig * KS * 1024));
    CK(hipMalloc(&vn, (gbig * 32 + 64) * 4));
    CK(hipMalloc(&qn, nq * 4));
    CK(hipMalloc(&out, nq * 4));
    CK(hipMalloc(&ref, nq * 4));
    {
        const uint64_t uq = (uint64_t)nq / 32 * KS * 64;
        hipLaunchKernelGGL(k_gen_frag, dim3((uq + 255) / 256), dim3(256), 0, 0, 0xC0FFEEull, (uint64_t)nq, dim, (uint32_t)KS, uq, Qf, qn);
        const uint64_t uv = gbig * KS * 64;
        hipLaunchKernelGGL(k_gen_frag, dim3((uint32_t)((uv + 255) / 256)), dim3(256), 0, 0, 0x5EEDull, nbig, dim, (uint32_t)KS, uv, Vf, vn);
        CK(hipDeviceSynchronize());
    }
    // correctness on the first nsmall rows (norm of rows past nsmall is whatever the big corpus has: both kernels read the same)
    {
        CK(hipMemset(ref, 0, nq * 4));
        hipLaunchKernelGGL(k_ref, dim3(nq), dim3(256), 0, 0, Qf, Vf, vn, gsmall * 32, (uint32_t)KS, ref);
        run<KS, 4, 0, 0>("check", Qf, Vf, vn, (uint32_t)gsmall, nq, out, 2);
        std::vector<uint32_t> a(nq), c(nq);
        CK(hipMemcpy(a.data(), out, nq * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c.data(), ref, nq * 4, hipMemcpyDeviceToHost));
        auto unkey = [](uint32_t k) { uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; float f; memcpy(&f, &u, 4); return f; };
        double worst = 0;
        for (uint32_t i = 0; i < nq; ++i) worst = std::max(worst, (double)std::fabs(unkey(a[i]) - unkey(c[i])));
        printf("check: max |regq - ref| over %u queries = %.3g (values ~ %.3f)\n", nq, worst, unkey(c[0]));
    }
    const uint32_t G = (uint32_t)gbig;
    {   // 64 queries per wave, one wave per SIMD
        run_k("check64", k_regq64<KS, 4>, 256, Qf, Vf, vn, (uint32_t)gsmall, nq, out, 2);
        check("check64", out, ref, nq);
        run_k("4 waves x 64 queries, R=4", k_regq64<KS, 4>, 256, Qf, Vf, vn, G, nq, out, 6);

    }
    if (argc > 2) {   // 16x16x32 shape on its own layout (second corpus buffer)
        unsigned char *Qf16, *Vf16;
        CK(hipMalloc(&Qf16, (size_t)nq / 32 * KS * 1024));
        CK(hipMalloc(&Vf16, gbig * KS * 1024));
        const uint64_t uq = (uint64_t)nq / 32 * KS * 64, uv = gbig * KS * 64;
        hipLaunchKernelGGL(k_gen_frag16, dim3((uq + 255) / 256), dim3(256), 0, 0, 0xC0FFEEull, (uint64_t)nq, dim, (uint32_t)KS / 2, uq, Qf16);
        hipLaunchKernelGGL(k_gen_frag16, dim3((uint32_t)((uv + 255) / 256)), dim3(256), 0, 0, 0x5EEDull, nbig, dim, (uint32_t)KS / 2, uv, Vf16);
        CK(hipDeviceSynchronize());
        run_k("check16", k_regq16<KS, 4>, 512, Qf16, Vf16, vn, (uint32_t)gsmall, nq, out, 2);
        check("check16", out, ref, nq);
        run_k("16x16x32, 8 waves x 32 queries, R=4", k_regq16<KS, 4>, 512, Qf16, Vf16, vn, G, nq, out, 6);
        run_k("16x16x32, 8 waves x 32 queries, R=6", k_regq16<KS, 6>, 512, Qf16, Vf16, vn, G, nq, out, 6);
        run_k("16x16x32, R=6, spread DMA", k_regq16<KS, 6, 1>, 512, Qf16, Vf16, vn, G, nq, out, 6);
        run_k("check16w4", k_regq16w4<KS, 4, 3>, 256, Qf16, Vf16, vn, (uint32_t)gsmall, nq, out, 2);
        check("check16w4 (4 waves x 64 queries, flow pipeline)", out, ref, nq);
        run_k("16x16x32, 4 waves x 64 queries, flow, R=4, DMA every 2", k_regq16w4<KS, 4, 2>, 256, Qf16, Vf16, vn, G, nq, out, 6);
        run_k("16x16x32, 4 waves x 64 queries, flow, R=4, DMA every 3", k_regq16w4<KS, 4, 3>, 256, Qf16, Vf16, vn, G, nq, out, 6);
        run_k("16x16x32, 4 waves x 64 queries, flow, R=6, DMA every 3", k_regq16w4<KS, 6, 3>, 256, Qf16, Vf16, vn, G, nq, out, 6);
        run_k("check16w4a", k_regq16w4a<KS, 4, 2, 60>, 256, Qf16, Vf16, vn, (uint32_t)gsmall, nq, out, 2);
        check("check16w4a (asm MFMAs, 60 B fragments in AGPRs)", out, ref, nq);
        run_k("16x16x32, 4 x 64, flow, asm MFMA, NA=60, R=4, DMA every 2", k_regq16w4a<KS, 4, 2, 60>, 256, Qf16, Vf16, vn, G, nq, out, 6);
        run_k("16x16x32, 4 x 64, flow, asm MFMA, NA=64, R=4, DMA every 2", k_regq16w4a<KS, 4, 2, 64>, 256, Qf16, Vf16, vn, G, nq, out, 6);
        run_k("16x16x32, 4 x 64, flow, asm MFMA, NA=60, R=6, DMA every 2", k_regq16w4a<KS, 6, 2, 60>, 256, Qf16, Vf16, vn, G, nq, out, 6);
        run_k("check16w4a3", k_regq16w4a<KS, 4, 3, 60>, 256, Qf16, Vf16, vn, (uint32_t)gsmall, nq, out, 2);
        check("check16w4a3 (asm MFMAs, DMA every 3)", out, ref, nq);
        run_k("16x16x32, 4 x 64, flow, asm MFMA, NA=60, R=4, DMA every 3", k_regq16w4a<KS, 4, 3, 60>, 256, Qf16, Vf16, vn, G, nq, out, 6);
        run_k("16x16x32, 4 x 64, flow, asm MFMA, NA=60, R=6, DMA every 3", k_regq16w4a<KS, 6, 3, 60>, 256, Qf16, Vf16, vn, G, nq, out, 6);
        if (argc > 4) return 0;
        run_k("check16s", k_regq16s<KS, 6, 1>, 512, Qf16, Vf16, vn, (uint32_t)gsmall, nq, out, 2);
        check("check16s (staggered)", out, ref, nq);
        run_k("check16s0", k_regq16s<KS, 6, 0>, 512, Qf16, Vf16, vn, (uint32_t)gsmall, nq, out, 2);
        check("check16s0 (half-unit ring, no stagger)", out, ref, nq);
        run_k("16x16x32 half-unit ring, STAGGERED, R=6", k_regq16s<KS, 6, 1>, 512, Qf16, Vf16, vn, G, nq, out, 6);
        run_k("16x16x32 half-unit ring, no stagger, R=6", k_regq16s<KS, 6, 0>, 512, Qf16, Vf16, vn, G, nq, out, 6);
        run_k("16x16x32 half-unit ring, STAGGERED, R=4", k_regq16s<KS, 4, 1>, 512, Qf16, Vf16, vn, G, nq, out, 6);
        run_k("16x16x32, R=4, spread DMA", k_regq16<KS, 4, 1>, 512, Qf16, Vf16, vn, G, nq, out, 6);
        if (argc > 3) {  // the same on rows and queries centred on 0 (L2 distances do not change)
            const float half = 0.5f;
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_shift), &half, 4));
            hipLaunchKernelGGL(k_gen_frag16, dim3((uq + 255) / 256), dim3(256), 0, 0, 0xC0FFEEull, (uint64_t)nq, dim, (uint32_t)KS / 2, uq, Qf16);
            hipLaunchKernelGGL(k_gen_frag16, dim3((uint32_t)((uv + 255) / 256)), dim3(256), 0, 0, 0x5EEDull, nbig, dim, (uint32_t)KS / 2, uv, Vf16);
            CK(hipDeviceSynchronize());
            run_k("CENTRED 16x16x32, R=6", k_regq16<KS, 6>, 512, Qf16, Vf16, vn, G, nq, out, 6);
            run_k("CENTRED 16x16x32, R=6, spread DMA", k_regq16<KS, 6, 1>, 512, Qf16, Vf16, vn, G, nq, out, 6);
            const float zero = 0.0f;
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_shift), &zero, 4));
        }
        CK(hipFree(Vf16));
    }
    run<KS, 4, 0, 0>("full R=4", Qf, Vf, vn, G, nq, out, 6);
    run<KS, 4, 0, 1>("spread DMA (every 8 k-steps)", Qf, Vf, vn, G, nq, out, 6);

    run<KS, 4, 1, 0>("no DMA in loop", Qf, Vf, vn, G, nq, out, 4);

    run<KS, 4, 5, 0>("no DMA, no fragment reads (MFMA + barrier)", Qf, Vf, vn, G, nq, out, 4);
    run<KS, 4, 2, 0>("no MFMA (DMA + reads + barrier)", Qf, Vf, vn, G, nq, out, 4);
    return 0;
}
