// pixel_math.hpp -- per-channel arithmetic shared by every kernel of the path (gfx950 only).
//
// The reference converts through lookup tables (src/cprocess/half.c:31-51, genhalf.py); here the
// same mappings come from two CDNA4 instructions plus one fix-up:
//   h2f : v_cvt_f32_f16          -- exact for all 65536 codes (signalling NaNs come out quiet)
//   f2h : v_cvt_pkrtz_f16_f32    -- round toward zero, subnormals included, which is what
//                                   `base + (mantissa >> shift)` computes; the one difference is
//                                   finite overflow: RTZ saturates at 65504 where the table gives
//                                   +-Inf for |x| >= 65536 (genhalf.py:36-37), fixed below.
// Arithmetic flavours.  The reference has two builds (SConstruct:46-48,75-83): gcc -std=c99 rounds every multiply and every
// add on its own; clang -- preferred when installed -- contracts a * b + c INSIDE ONE EXPRESSION into a fused multiply-add
// (-ffp-contract=on).  Every translation unit here is built with -ffp-contract=off, so nothing fuses by accident; the
// places where the reference's C has a product and a sum in one expression go through madd() / nmadd() below, and the units
// that hold arithmetic are compiled twice: plain (kContract == false: bit-equal to the reference's gcc build) and
// with -DCVS_CONTRACT (kContract == true: bit-equal to its clang build; launchers renamed *_fma in
// kernels.h).  Which product of an expression clang fuses was read off its IR (llvm.fmuladd operands) for the same C expressions:
//   a*b + c*d        -> fma(a, b, c*d)                 (left product fused, right one rounded)
//   a*b + c*d + e*f  -> fma(e, f, fma(a, b, c*d))
//   t += s*c         -> fma(s, c, t)
//   1 - a*b          -> fma(-a, b, 1)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cvs {

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifdef CVS_CONTRACT
constexpr bool kContract = true;
#else
constexpr bool kContract = false;
#endif

// a * b + c written in one expression by the reference: rounded twice (gcc build) or once (clang build)
__device__ __forceinline__ float madd(float a, float b, float c) {
    if constexpr (kContract) return __builtin_fmaf(a, b, c); else return a * b + c;
}
__device__ __forceinline__ f32x2 madd(f32x2 a, f32x2 b, f32x2 c) {
    if constexpr (kContract) return __builtin_elementwise_fma(a, b, c); else return a * b + c;
}
__device__ __forceinline__ f32x2 madd(f32x2 a, float b, f32x2 c) { return madd(a, f32x2{ b, b }, c); }
// c - a * b in one expression
__device__ __forceinline__ float nmadd(float a, float b, float c) {
    if constexpr (kContract) return __builtin_fmaf(-a, b, c); else return c - a * b;
}
__device__ __forceinline__ f32x2 nmadd(f32x2 a, f32x2 b, f32x2 c) {
    if constexpr (kContract) return __builtin_elementwise_fma(-a, b, c); else return c - a * b;
}

__device__ __forceinline__ float h2f(uint32_t code) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)code);
}

// |x| >= 65536 must become +-Inf.  (x * 2^112) overflows to Inf exactly for those x and is an
// exact scaling otherwise; multiplying back by 2^-112 restores every other value bit for bit
// (f32 denormals are kept on gfx950), and NaN stays NaN.
__device__ __forceinline__ float saturate_to_inf(float x) {
    return (x * 0x1p112f) * 0x1p-112f;
}

// two channels -> one dword of two truncated halfs (lo = a, hi = b)
__device__ __forceinline__ uint32_t f2h_rz2(float a, float b) {
    auto v = __builtin_amdgcn_cvt_pkrtz(saturate_to_inf(a), saturate_to_inf(b));
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ uint32_t f2h_rz(float a) { return f2h_rz2(a, 0.0f) & 0xFFFFu; }

// bit-twiddled variants the reference also exports (half.c:39-45, 53-59); only exact for normals
__device__ __forceinline__ float h2f_fast(uint32_t v) {
    return __uint_as_float(((v & 0x8000u) << 16) | (((v & 0x7c00u) + 0x1C000u) << 13) | ((v & 0x03FFu) << 13));
}
__device__ __forceinline__ uint32_t f2h_fast(float f) {
    uint32_t u = __float_as_uint(f);
    return ((u >> 16) & 0x8000u) | ((((u & 0x7f800000u) - 0x38000000u) >> 13) & 0x7c00u) | ((u >> 13) & 0x03ffu);
}

struct px32 { float r, g, b, a; };

// one rgba_f16 pixel = two dwords: lo = g:r, hi = a:b
__device__ __forceinline__ px32 widen(uint2 p) {
    return { h2f(p.x & 0xFFFFu), h2f(p.x >> 16), h2f(p.y & 0xFFFFu), h2f(p.y >> 16) };
}
__device__ __forceinline__ uint2 narrow(px32 v) {
    return make_uint2(f2h_rz2(v.r, v.g), f2h_rz2(v.b, v.a));
}

// color.c:34-42 -- left-to-right, alpha copied.  m is column-major (see canvas_hip.h).
// `mat`: the nine coefficients as named members m0..m8 (cvs::MatR)
template <class M>
__device__ __forceinline__ px32 mat3(px32 v, const M &mat) {
    px32 o;
    o.r = madd(v.b, mat.m6, madd(v.r, mat.m0, v.g * mat.m3));
    o.g = madd(v.b, mat.m7, madd(v.r, mat.m1, v.g * mat.m4));
    o.b = madd(v.b, mat.m8, madd(v.r, mat.m2, v.g * mat.m5));
    o.a = v.a;
    return o;
}

// video_mix.c:323-337
__device__ __forceinline__ px32 blend_over(px32 lo, px32 b, float mix_b) {
    float alpha_b = b.a * mix_b;
    float alpha_a = lo.a * nmadd(b.a, mix_b, 1.0f);
    float a = alpha_a + alpha_b;
    px32 o = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (a != 0.0f) {
        o.r = madd(lo.r, alpha_a, b.r * alpha_b) / a;
        o.g = madd(lo.g, alpha_a, b.g * alpha_b) / a;
        o.b = madd(lo.b, alpha_a, b.b * alpha_b) / a;
        o.a = a;
    }
    return o;
}

// video_mix.c:193-205
__device__ __forceinline__ px32 blend_cross(px32 a, px32 b, float mix_a, float mix_b) {
    float alpha_a = a.a * mix_a;
    float alpha_b = b.a * mix_b;
    float oa = alpha_a + alpha_b;
    px32 o = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (oa != 0.0f) {
        o.r = madd(a.r, alpha_a, b.r * alpha_b) / oa;
        o.g = madd(a.g, alpha_a, b.g * alpha_b) / oa;
        o.b = madd(a.b, alpha_a, b.b * alpha_b) / oa;
        o.a = oa;
    }
    return o;
}

}  // namespace cvs

// ---------------------------------------------------------------- two pixels at a time
// A lane of the chain kernel owns a PAIR of pixels (one 16-byte word).  Keeping the pair as
// 2-wide vectors lets every mul/add of the matrix and of the blend issue as v_pk_mul_f32 /
// v_pk_add_f32 (2 f32 results per instruction, full rate on CDNA3/4): same operations, same
// rounding, half the VALU issue slots.  Conversions, gathers and the divides stay per channel.
namespace cvs {

struct px32x2 { f32x2 r, g, b, a; };   // .x = first pixel of the pair, .y = second

// 16-byte word: x = g0:r0, y = a0:b0, z = g1:r1, w = a1:b1
__device__ __forceinline__ px32x2 widen2(uint4 p) {
    px32x2 v;
    v.r = f32x2{ h2f(p.x & 0xFFFFu), h2f(p.z & 0xFFFFu) };
    v.g = f32x2{ h2f(p.x >> 16), h2f(p.z >> 16) };
    v.b = f32x2{ h2f(p.y & 0xFFFFu), h2f(p.w & 0xFFFFu) };
    v.a = f32x2{ h2f(p.y >> 16), h2f(p.w >> 16) };
    return v;
}

__device__ __forceinline__ f32x2 saturate_to_inf2(f32x2 x) { return (x * 0x1p112f) * 0x1p-112f; }

__device__ __forceinline__ uint32_t pkrtz(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
}

__device__ __forceinline__ uint4 narrow2(px32x2 v) {
    f32x2 r = saturate_to_inf2(v.r), g = saturate_to_inf2(v.g), b = saturate_to_inf2(v.b), a = saturate_to_inf2(v.a);
    return make_uint4(pkrtz(r.x, g.x), pkrtz(b.x, a.x), pkrtz(r.y, g.y), pkrtz(b.y, a.y));
}

template <class M>
__device__ __forceinline__ px32x2 mat3x2(px32x2 v, const M &mat) {
    px32x2 o;
    o.r = madd(v.b, mat.m6, madd(v.r, mat.m0, v.g * mat.m3));
    o.g = madd(v.b, mat.m7, madd(v.r, mat.m1, v.g * mat.m4));
    o.b = madd(v.b, mat.m8, madd(v.r, mat.m2, v.g * mat.m5));
    o.a = v.a;
    return o;
}

// video_mix.c:323-337 with mix_b == 1.0f (workspace.c:543): b.a * 1.0f is b.a exactly
__device__ __forceinline__ px32x2 blend_over2_mix1(px32x2 lo, px32x2 b) {
    f32x2 alpha_b = b.a;
    f32x2 alpha_a = lo.a * (1.0f - b.a);
    f32x2 a = alpha_a + alpha_b;
    f32x2 nr = madd(lo.r, alpha_a, b.r * alpha_b);
    f32x2 ng = madd(lo.g, alpha_a, b.g * alpha_b);
    f32x2 nb = madd(lo.b, alpha_a, b.b * alpha_b);
    px32x2 o;
    // IEEE divides, per pixel; a == 0 selects the zero pixel afterwards (x/0 is never used)
    o.r = f32x2{ nr.x / a.x, nr.y / a.y };
    o.g = f32x2{ ng.x / a.x, ng.y / a.y };
    o.b = f32x2{ nb.x / a.x, nb.y / a.y };
    o.a = a;
    if (a.x == 0.0f) { o.r.x = 0.0f; o.g.x = 0.0f; o.b.x = 0.0f; o.a.x = 0.0f; }
    if (a.y == 0.0f) { o.r.y = 0.0f; o.g.y = 0.0f; o.b.y = 0.0f; o.a.y = 0.0f; }
    return o;
}

}  // namespace cvs
