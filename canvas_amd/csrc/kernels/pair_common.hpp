// pair_common.hpp -- pieces shared by the two-columns-per-lane sweeps (blur_pair_ops.hip, blur_halve_pair_ops.hip).
#pragma once
#include <type_traits>
#include <utility>
#include "kernels.h"
#include "chain_math.hpp"

namespace pairsweep {

using cvs::f32x2;
using cvs::u32x4;
using cvs::u32x2;

struct Px { f32x2 rg, ba; };

// f(0), f(1), ... with the index as a compile-time constant, until one returns false
template <class F, int... Js>
__device__ __forceinline__ void each_slot(F &f, std::integer_sequence<int, Js...>) {
    (void)(f(std::integral_constant<int, Js>{}) && ...);
}

// One row of a buffer as a raw buffer resource (stride 0: offsets are bytes, range-checked against `bytes`).  Everything that
// goes into it is wave-uniform (kernel arguments and block indices), so the descriptor lives in scalar registers; a lane's
// access outside the row -- a column left or right of the window, any column of a row given zero bytes, a lane whose offset
// was set to 0x80000000 -- loads zeros or stores nothing, without a predicate.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t row_rsrc(const void *base, size_t row_offset, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(base)) + row_offset, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 load_pair(rsrc_t r, uint32_t voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
}

// one rgba_f16 pixel (dwords g:r, a:b), widened
__device__ __forceinline__ float4 widen_px(uint32_t lo, uint32_t hi) {
    return make_float4(cvs::h2f(lo & 0xFFFFu), cvs::h2f(lo >> 16), cvs::h2f(hi & 0xFFFFu), cvs::h2f(hi >> 16));
}

inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

template <class K>
int resident_per_cu(K kernel, int block) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, 0) != hipSuccess || n < 1) n = 1;
    return n;
}

}  // namespace pairsweep
