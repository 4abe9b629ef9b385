// grade.hpp -- one layer pixel through the color.c structure (LUT -> widen -> 3x3 -> truncate -> LUT).
#pragma once
#include "lut_common.hpp"

namespace cvs {

// one layer pixel through the color.c structure; returns the f32 value the stack then sees
template <bool PRE, bool POST>
__device__ __forceinline__ px32 grade(uint2 p, const MatR &mat, const uint16_t *lds_lut, const uint16_t *glb_post) {
    if (PRE) { p.x = gather2<true>(lds_lut, p.x); p.y = gather2<true>(lds_lut, p.y); }
    uint2 h = narrow(mat3(widen(p), mat));
    if (POST) {
        // the LDS slot belongs to the pre table when there is one
        if (PRE) { h.x = gather2<false>(glb_post, h.x); h.y = gather2<false>(glb_post, h.y); }
        else     { h.x = gather2<true>(lds_lut, h.x);   h.y = gather2<true>(lds_lut, h.y); }
    }
    return widen(h);
}

template <bool PRE, bool POST>
__device__ __forceinline__ uint2 grade_h(uint2 p, const MatR &mat, const uint16_t *lds_lut, const uint16_t *glb_post) {
    if (PRE) { p.x = gather2<true>(lds_lut, p.x); p.y = gather2<true>(lds_lut, p.y); }
    uint2 h = narrow(mat3(widen(p), mat));
    if (POST) {
        if (PRE) { h.x = gather2<false>(glb_post, h.x); h.y = gather2<false>(glb_post, h.y); }
        else     { h.x = gather2<true>(lds_lut, h.x);   h.y = gather2<true>(lds_lut, h.y); }
    }
    return h;
}

}  // namespace cvs
