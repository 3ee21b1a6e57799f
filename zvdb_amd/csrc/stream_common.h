// stream_common.h -- helpers shared by the barrier-free stream scan kernels (bf_fp4_stream.hip, bf_i8_stream.hip)
#pragma once
#include "common.h"

namespace zvdb {

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {  // f(integral_constant<int, i>) for i = 0 .. N-1: i usable as an asm immediate
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// tell the compiler a pointer is wave-uniform (SGPR pair) and GLOBAL: loads become saddr + 32-bit lane offset.  (A generic
// pointer would turn the stream loads into flat_load, which is counted in lgkmcnt as well as vmcnt: measured 2 % slower.)
typedef const __attribute__((address_space(1))) unsigned char *gbytes_t;
__device__ __forceinline__ gbytes_t uniform_ptr(const unsigned char *p) {
    const uint64_t v = (uint64_t)(size_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (gbytes_t)(size_t)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ uint32_t lds_load_u32(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace zvdb
