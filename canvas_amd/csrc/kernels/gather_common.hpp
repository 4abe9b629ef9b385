// gather_common.hpp -- what the two per-target-line gather kernels share (sweep_vh_ops.hip: vertical pass first;
// sweep_hv_ops.hip: horizontal pass first): pixels as two channel pairs, rows in storage format, the sum of a line's taps,
// the truncating store.
#pragma once
#include "chain_math.hpp"

namespace {

using cvs::f32x2;

struct Px { f32x2 lo, hi; };                                             // r,g | b,a
template <bool INH> struct Raw;
template <> struct Raw<true> { uint2 v; };                               // four halfs
template <> struct Raw<false> { float4 v; };
__device__ __forceinline__ Px widen(const Raw<true> &r) {
    return { f32x2{ cvs::h2f(r.v.x & 0xFFFFu), cvs::h2f(r.v.x >> 16) }, f32x2{ cvs::h2f(r.v.y & 0xFFFFu), cvs::h2f(r.v.y >> 16) } };
}
__device__ __forceinline__ Px widen(const Raw<false> &r) { return { f32x2{ r.v.x, r.v.y }, f32x2{ r.v.z, r.v.w } }; }

// the vertical sum of one line with exactly N taps (video_scale.c:82-85: t += s * coeff, from 0, ascending).  The first
// addition, 0 + p0, is p0 itself except for the sign of a zero product (the reference is built -fno-signed-zeros; every
// comparison in tests/ and every fixture folds it): left out, two packed adds per pixel and pass.
template <int N, int W>
__device__ __forceinline__ Px vsum(const Px (&win)[W], const float (&w)[W]) {
    const f32x2 w0 = { w[0], w[0] };
    Px t = { win[0].lo * w0, win[0].hi * w0 };
#pragma unroll
    for (int k = 1; k < N; k++) {
        const f32x2 wk = { w[k], w[k] };
        t.lo = cvs::madd(win[k].lo, wk, t.lo);          // t += s * coeff: two roundings, or one in the contracted build
        t.hi = cvs::madd(win[k].hi, wk, t.hi);
    }
    return t;
}

// four channels -> two dwords of truncated halfs; |x| >= 65536 must become Inf where v_cvt_pkrtz saturates (pixel_math.hpp):
// the exact scaling that does it costs two multiplies a channel, so it sits behind a wave-uniform test (chain_math.hpp does
// the same for the chain's stores)
__device__ __forceinline__ uint2 narrow4(f32x2 lo, f32x2 hi) {
    const float big = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(lo.x), __builtin_fabsf(lo.y)), __builtin_fmaxf(__builtin_fabsf(hi.x), __builtin_fabsf(hi.y)));
    if (cvs::wave_any(!(big < 65536.0f))) {             // (also taken for NaN: harmless, the scaling keeps NaN)
        cvs::rare_path();
        return make_uint2(cvs::f2h_rz2(lo.x, lo.y), cvs::f2h_rz2(hi.x, hi.y));
    }
    return make_uint2(cvs::pkrtz(lo.x, lo.y), cvs::pkrtz(hi.x, hi.y));
}


// the same sum with the count known only at run time (wave-uniform): taps the line does not have are replaced by zeros
// BEFORE the multiply (a select per channel), so a row beyond the line's last tap may hold Inf or NaN without harm.
// For the rare counts that have no chain of their own.
template <int W>
__device__ __forceinline__ Px vsum_any(const Px (&win)[W], const float (&w)[W], int n) {
    Px t = { f32x2{ 0.0f, 0.0f }, f32x2{ 0.0f, 0.0f } };
#pragma unroll
    for (int k = 0; k < W; k++) {
        const bool has = k < n;                          // uniform
        const f32x2 wk = { w[k], w[k] };
        const f32x2 xlo = { has ? win[k].lo.x : 0.0f, has ? win[k].lo.y : 0.0f }, xhi = { has ? win[k].hi.x : 0.0f, has ? win[k].hi.y : 0.0f };
        t.lo = cvs::madd(xlo, wk, t.lo);
        t.hi = cvs::madd(xhi, wk, t.hi);
    }
    return t;
}

}  // namespace
