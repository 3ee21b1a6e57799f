// row_types.h -- the stored row element types of an index (fp32, or bf16 with ZVDB_FLAG_BF16_ROWS) and the launch-side
// dispatch on them.  No floating-point pragmas here: included by the exact translation units (through exact_dist.h) and by
// the mirror builders / screens, which keep the compiler's default contraction.
#pragma once
#include <type_traits>

#include "common.h"

namespace zvdb {

// ---- row element types ---------------------------------------------------------
// An index stores its rows as fp32 (the reference's T = f32) or, with ZVDB_FLAG_BF16_ROWS, as bf16 (SURVEY 8a row a11:
// "bf16 = storage type with fp32 accumulate"; BASELINE configs 3-4).  Every kernel below is templated on the stored
// element type RT; a bf16 element is widened to fp32 exactly (its 16 bits are the upper half of the fp32 pattern) and the
// arithmetic on the widened values is the reference's, in the reference's order -- half the bytes per row, same results
// as fp32 rows holding bf16-representable values.
struct bf16_t { uint16_t bits; };

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }          // even element of a pair
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }  // odd element
__device__ __forceinline__ float row_elem(const float *row, uint32_t i) { return row[i]; }
__device__ __forceinline__ float row_elem(const bf16_t *row, uint32_t i) { return __uint_as_float((uint32_t)row[i].bits << 16); }
// four consecutive elements (16 B of fp32 / 8 B of bf16; the address is a multiple of 4 elements)
__device__ __forceinline__ float4 row_load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 row_load4(const bf16_t *p) {
    const uint2 w = *reinterpret_cast<const uint2 *>(p);
    return float4{bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y)};
}

// launch-side dispatch on the index's stored row type: f(rows pointer of the right type, elements per row)
template <class F>
inline int rows_dispatch(const zvdb_hip_index *h, F f) {
    if (h->flags & ZVDB_FLAG_BF16_ROWS) return f(reinterpret_cast<const bf16_t *>(h->rows_h), h->ldh);
    return f(static_cast<const float *>(h->rows), h->ld);
}

// element type of a rows pointer handed out by rows_dispatch
template <class P> using row_elem_t = std::remove_cv_t<std::remove_pointer_t<P>>;

}  // namespace zvdb
