// chain_math.hpp -- the arithmetic of the fused chain on PAIRS of pixels, trimmed to what the
// reference's results actually require.  Every shortcut below is exact (bit-identical to the
// straightforward form in pixel_math.hpp / grade.hpp, which the v0 kernel keeps for A/B checks):
//
//   * alpha never goes through truncate+widen: the colour filter copies alpha (color.c:41), the value
//     is already a half, and f2h(h2f(c)) == c for every code c;
//   * the |x| >= 65536 -> Inf fix-up of the truncation (pixel_math.hpp) is only executed by waves in
//     which some lane actually holds such a value (one v_max3 chain + one wave-uniform branch
//     instead of two multiplies per channel);
//   * x / 1.0f == x: when the blended alpha of both pixels of the pair is exactly 1.0f (always the
//     case over an opaque lower layer: a*(1-b)+b with a == 1 is exactly 1 for half-valued b) the
//     three IEEE divides are skipped; lanes that need them take the general path;
//   * gathered table entries are converted straight from the gather result, without re-packing two
//     halfs into a dword first.
// VALU cost matters here: measured on gfx950 (tools/valubench.hip) the kernel's instruction mix
// issues at ~4.5 cycles per wave-instruction per SIMD with 4 waves per SIMD, which made the first
// version of this kernel compute-bound (245 us of arithmetic against 290 us of HBM time per launch).
#pragma once
#include "lut_common.hpp"

namespace cvs {

__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }

// Placed at the top of a rarely taken, wave-uniform branch.  Without it hipcc if-converts a short branch body: the
// |x| >= 65536 fix-up below became 6 packed multiplies + 6 selects executed on EVERY trip (seen in the ISA of the
// first pipelined kernel: 18 issue slots per layer pair for a case that ordinary footage never reaches).  An asm
// statement cannot be speculated, so the body stays behind its s_cbranch.
__device__ __forceinline__ void rare_path() { asm volatile("; rare path"); }

// truncate a pair of f32 to half and widen back (the rounding point between the colour filter and the
// stack, color.c:132 / :159 + main.c:128-136); POST applies the post-table to the half codes
template <bool POST_LDS, bool POST_GLB>
__device__ __forceinline__ f32x2 through_half(f32x2 v, const uint16_t *lds_lut, const uint16_t *glb_post) {
    uint32_t pk = pkrtz(v.x, v.y);
    uint32_t lo = pk & 0xFFFFu, hi = pk >> 16;
    if (POST_LDS) { lo = lds_lut[lo]; hi = lds_lut[hi]; }
    if (POST_GLB) { lo = glb_post[lo]; hi = glb_post[hi]; }
    return f32x2{ h2f(lo), h2f(hi) };
}

template <bool PRE, bool POST>
__device__ __forceinline__ px32x2 grade_pair(u32x4 p, const MatR &mat, const uint16_t *lds_lut, const uint16_t *glb_post) {
    uint32_t c[8] = { p.x & 0xFFFFu, p.x >> 16, p.y & 0xFFFFu, p.y >> 16, p.z & 0xFFFFu, p.z >> 16, p.w & 0xFFFFu, p.w >> 16 };
    if (PRE) {
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = lds_lut[c[i]];
    }
    px32x2 v;
    v.r = f32x2{ h2f(c[0]), h2f(c[4]) };
    v.g = f32x2{ h2f(c[1]), h2f(c[5]) };
    v.b = f32x2{ h2f(c[2]), h2f(c[6]) };
    px32x2 o = mat3x2(v, mat);
    // rare: a channel at or beyond the half range must become Inf, not 65504
    const float big = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.r.x), __builtin_fabsf(o.g.x)), __builtin_fabsf(o.b.x)),
                                      __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.r.y), __builtin_fabsf(o.g.y)), __builtin_fabsf(o.b.y)));
    if (wave_any(big >= 65536.0f)) { rare_path(); o.r = saturate_to_inf2(o.r); o.g = saturate_to_inf2(o.g); o.b = saturate_to_inf2(o.b); }
    constexpr bool post_lds = POST && !PRE, post_glb = POST && PRE;     // the LDS slot belongs to the pre table when both exist
    o.r = through_half<post_lds, post_glb>(o.r, lds_lut, glb_post);
    o.g = through_half<post_lds, post_glb>(o.g, lds_lut, glb_post);
    o.b = through_half<post_lds, post_glb>(o.b, lds_lut, glb_post);
    // alpha: copied by the matrix, already a half -> only the optional post-table touches it
    uint32_t a0 = c[3], a1 = c[7];
    if (post_lds) { a0 = lds_lut[a0]; a1 = lds_lut[a1]; }
    if (post_glb) { a0 = glb_post[a0]; a1 = glb_post[a1]; }
    o.a = f32x2{ h2f(a0), h2f(a1) };
    return o;
}

// ---- three IEEE quotients by one denominator, for both pixels of a pair --------------------------------------
// hipcc expands an f32 '/' into v_div_scale x2, v_rcp, 2 FMAs refining the reciprocal, a multiply, 3 FMAs refining
// the quotient with exact residuals, v_div_fmas, v_div_fixup: 11 instructions per quotient, 66 per blended pair.
// When neither operand needs the scaling step (the scale instructions return their input, fmas is a plain FMA,
// fixup passes its input through) the same arithmetic can share the reciprocal among the three channels and run
// on both pixels at once as v_pk_fma_f32: 2 rcp + 2 + 3 * 5 packed instructions per pair.  The condition is
// checked per wave (every operand normal with magnitude in [2^-60, 2^60): quotients, products and residuals all
// stay normal); any zero, denormal, Inf or extreme value sends the wave through the plain '/' instead.
// Same instructions on the same values => the same bits as '/', which is what the reference's IEEE divide gives.
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ bool div_band(f32x2 n0, f32x2 n1, f32x2 n2, f32x2 d) {
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(n0.x), __builtin_fabsf(n0.y)), __builtin_fmaxf(__builtin_fabsf(n1.x), __builtin_fabsf(n1.y))),
                                     __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(n2.x), __builtin_fabsf(n2.y)), __builtin_fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y))));
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fabsf(n0.x), __builtin_fabsf(n0.y)), __builtin_fminf(__builtin_fabsf(n1.x), __builtin_fabsf(n1.y))),
                                     __builtin_fminf(__builtin_fminf(__builtin_fabsf(n2.x), __builtin_fabsf(n2.y)), __builtin_fminf(__builtin_fabsf(d.x), __builtin_fabsf(d.y))));
    // fmax/fmin drop a NaN operand, so a NaN can slip through: harmless, both forms then give NaN (rcp(NaN), NaN*r
    // and fma(.., NaN) are NaN; payload and sign of a NaN are not pinned)
    return lo >= 0x1p-60f && hi < 0x1p60f;
}

__device__ __forceinline__ void div3_pair(f32x2 &n0, f32x2 &n1, f32x2 &n2, f32x2 d) {
#ifdef CVS_DIV_ALWAYS_FAST
    if (true) {
#else
    if (!wave_any(!div_band(n0, n1, n2, d))) {          // every active lane is inside the band
#endif
        f32x2 r = { __builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y) };
        const f32x2 nd = -d;
        r = fma2(fma2(nd, r, f32x2{ 1.0f, 1.0f }), r, r);
        // the three channels step by step side by side: a packed FMA right behind the one it depends on costs a hazard slot
        f32x2 q0 = n0 * r, q1 = n1 * r, q2 = n2 * r;
        f32x2 e0 = fma2(nd, q0, n0), e1 = fma2(nd, q1, n1), e2 = fma2(nd, q2, n2);
        q0 = fma2(e0, r, q0); q1 = fma2(e1, r, q1); q2 = fma2(e2, r, q2);
        e0 = fma2(nd, q0, n0); e1 = fma2(nd, q1, n1); e2 = fma2(nd, q2, n2);
        n0 = fma2(e0, r, q0); n1 = fma2(e1, r, q1); n2 = fma2(e2, r, q2);
    } else {
        n0 = f32x2{ n0.x / d.x, n0.y / d.y };
        n1 = f32x2{ n1.x / d.x, n1.y / d.y };
        n2 = f32x2{ n2.x / d.x, n2.y / d.y };
    }
}

// video_mix.c:323-337 with mix_b == 1.0f (workspace.c:543)
__device__ __forceinline__ px32x2 over_pair(px32x2 lo, px32x2 b) {
    const f32x2 alpha_b = b.a;                       // b.a * 1.0f
    const f32x2 alpha_a = lo.a * (1.0f - b.a);       // (contracted build: fma(-b.a, 1.0f, 1.0f), the same value)
    const f32x2 a = alpha_a + alpha_b;
    px32x2 o;
    o.r = madd(lo.r, alpha_a, b.r * alpha_b);
    o.g = madd(lo.g, alpha_a, b.g * alpha_b);
    o.b = madd(lo.b, alpha_a, b.b * alpha_b);
    o.a = a;
    if (!(a.x == 1.0f && a.y == 1.0f)) {             // x / 1.0f == x: nothing to do for unit alpha
        div3_pair(o.r, o.g, o.b, a);
        if (a.x == 0.0f) { o.r.x = 0.0f; o.g.x = 0.0f; o.b.x = 0.0f; o.a.x = 0.0f; }
        if (a.y == 0.0f) { o.r.y = 0.0f; o.g.y = 0.0f; o.b.y = 0.0f; o.a.y = 0.0f; }
    }
    return o;
}

// video_mix.c:193-205: crossfade of two pixels, weights wa = 1 - mix_b and wb = mix_b made by the host
__device__ __forceinline__ px32x2 cross_pair(px32x2 p, px32x2 q, float wa, float wb) {
    const f32x2 alpha_a = p.a * wa;
    const f32x2 alpha_b = q.a * wb;
    const f32x2 a = alpha_a + alpha_b;
    px32x2 o;
    o.r = madd(p.r, alpha_a, q.r * alpha_b);
    o.g = madd(p.g, alpha_a, q.g * alpha_b);
    o.b = madd(p.b, alpha_a, q.b * alpha_b);
    o.a = a;
    if (!(a.x == 1.0f && a.y == 1.0f)) {
        div3_pair(o.r, o.g, o.b, a);
        if (a.x == 0.0f) { o.r.x = 0.0f; o.g.x = 0.0f; o.b.x = 0.0f; o.a.x = 0.0f; }
        if (a.y == 0.0f) { o.r.y = 0.0f; o.g.y = 0.0f; o.b.y = 0.0f; o.a.y = 0.0f; }
    }
    return o;
}

// The same blend for ONE pixel per lane (the FIR epilogue): (r, g) ride as a packed pair, b and alpha as scalars;
// the three quotients share one refined reciprocal under the same band rule as div3_pair.
struct px1 { f32x2 rg; float b, a; };

__device__ __forceinline__ px1 over_px(px1 lo, px1 up) {
    const float alpha_b = up.a;                      // up.a * 1.0f
    const float alpha_a = lo.a * (1.0f - up.a);
    const float a = alpha_a + alpha_b;
    f32x2 nrg = madd(lo.rg, alpha_a, up.rg * alpha_b);
    float nb = madd(lo.b, alpha_a, up.b * alpha_b);
    if (a != 1.0f) {
        const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(nrg.x), __builtin_fabsf(nrg.y)), __builtin_fmaxf(__builtin_fabsf(nb), __builtin_fabsf(a)));
        const float lw = __builtin_fminf(__builtin_fminf(__builtin_fabsf(nrg.x), __builtin_fabsf(nrg.y)), __builtin_fminf(__builtin_fabsf(nb), __builtin_fabsf(a)));
        if (!wave_any(!(lw >= 0x1p-60f && hi < 0x1p60f))) {
            float r = __builtin_amdgcn_rcpf(a);
            r = __builtin_fmaf(__builtin_fmaf(-a, r, 1.0f), r, r);
            const f32x2 nd = { -a, -a }, rr = { r, r };
            f32x2 q = nrg * rr;
            q = fma2(fma2(nd, q, nrg), rr, q);
            nrg = fma2(fma2(nd, q, nrg), rr, q);
            float qb = nb * r;
            qb = __builtin_fmaf(__builtin_fmaf(-a, qb, nb), r, qb);
            nb = __builtin_fmaf(__builtin_fmaf(-a, qb, nb), r, qb);
        } else {
            nrg = f32x2{ nrg.x / a, nrg.y / a };
            nb = nb / a;
        }
    }
    px1 o = { nrg, nb, a };
    if (a == 0.0f) { o.rg = f32x2{ 0.0f, 0.0f }; o.b = 0.0f; o.a = 0.0f; }
    return o;
}

// main.c:43-71: the stack's f32 result, truncated
__device__ __forceinline__ u32x4 narrow_pair(px32x2 v) {
    const float big = __builtin_fmaxf(
        __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v.r.x), __builtin_fabsf(v.g.x)), __builtin_fabsf(v.b.x)), __builtin_fabsf(v.a.x)),
        __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v.r.y), __builtin_fabsf(v.g.y)), __builtin_fabsf(v.b.y)), __builtin_fabsf(v.a.y)));
    if (wave_any(big >= 65536.0f)) { rare_path(); v.r = saturate_to_inf2(v.r); v.g = saturate_to_inf2(v.g); v.b = saturate_to_inf2(v.b); v.a = saturate_to_inf2(v.a); }
    return u32x4{ pkrtz(v.r.x, v.g.x), pkrtz(v.b.x, v.a.x), pkrtz(v.r.y, v.g.y), pkrtz(v.b.y, v.a.y) };
}

// ---- the same blend and truncation with the checks in their cheapest form (blur_pair_ops.hip, whose sweep is bound by vector
// issue, not by memory like the chain kernel's) -------------------------------------------------------------------
// |x| extrema with v_max3_f32 / v_min3_f32 and |abs| operand modifiers: hipcc's fmaxf(fabsf(a), fabsf(b)) first canonicalises each
// operand with a v_max_f32 of its own (16 instructions for the band check of one blended pair, 8 this way).  A quiet NaN
// operand is dropped and a signalling one comes out as NaN (which fails both comparisons below: the plain path, as before).
// largest and smallest magnitude of eight values: eight instructions in ONE statement (after every asm statement hipcc leaves
// a wait state, not knowing what it ended with; plain VALU results feeding plain VALU operands need none on gfx950)
__device__ __forceinline__ void abs_extrema8(float a, float b, float c, float d, float e, float f, float g, float h, float &hi, float &lo) {
    float t0, t1, t2, u0, u1, u2;
    asm("v_max3_f32 %2, |%8|, |%9|, |%10|\n\t"
        "v_max3_f32 %3, |%11|, |%12|, |%13|\n\t"
        "v_max3_f32 %4, |%14|, |%15|, |%15|\n\t"
        "v_min3_f32 %5, |%8|, |%9|, |%10|\n\t"
        "v_min3_f32 %6, |%11|, |%12|, |%13|\n\t"
        "v_min3_f32 %7, |%14|, |%15|, |%15|\n\t"
        "v_max3_f32 %0, %2, %3, %4\n\t"
        "v_min3_f32 %1, %5, %6, %7"
        : "=&v"(hi), "=&v"(lo), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(u0), "=&v"(u1), "=&v"(u2)
        : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h));
}

__device__ __forceinline__ bool div_band_lean(f32x2 n0, f32x2 n1, f32x2 n2, f32x2 d) {
    float hi, lo;
    abs_extrema8(n0.x, n0.y, n1.x, n1.y, n2.x, n2.y, d.x, d.y, hi, lo);
    return lo >= 0x1p-60f && hi < 0x1p60f;
}

// over_pair with wave-uniform control flow only: when ANY lane of the wave has a blended alpha other than 1.0f every lane
// divides (x / 1.0f == x exactly, by either path), so no exec-mask region and no merge copies surround the quotients; and
// the alpha == 0 fix-up lives on the plain path alone (inside the band |alpha| >= 2^-60 in every lane).
// The colour channels of the upper layer times their weight, straight from the halfs: v_fma_mix_f32 widens its f16 operand and
// multiplies in one instruction (fma(widen(h), w, +0): the product, rounded once -- what v_cvt_f32_f16 + v_mul_f32 give in
// two, except that an exactly zero product is always +0: the sign of a zero, which nothing here pins; half denormals are
// kept, as by the conversion).  `code`: the dword holding the half, HI: its upper 16 bits.
// All six products of a layer pair in ONE statement (hipcc leaves a wait state behind every asm statement).
__device__ __forceinline__ void halves_times(u32x4 up, f32x2 w, f32x2 &qr, f32x2 &qg, f32x2 &qb) {
    float r0, r1, g0, g1, b0, b1;
    asm("v_fma_mix_f32 %0, %6, %10, 0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %8, %11, 0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %6, %10, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %3, %8, %11, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %4, %7, %10, 0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %5, %9, %11, 0 op_sel_hi:[1,0,0]"
        : "=&v"(r0), "=&v"(r1), "=&v"(g0), "=&v"(g1), "=&v"(b0), "=&v"(b1)
        : "v"(up.x), "v"(up.y), "v"(up.z), "v"(up.w), "v"(w.x), "v"(w.y));
    qr = f32x2{ r0, r1 }; qg = f32x2{ g0, g1 }; qb = f32x2{ b0, b1 };
}

// `up`: the upper layer's two pixels as loaded (dwords g:r, a:b of the first pixel, then of the second)
__device__ __forceinline__ px32x2 over_pair_uniform(px32x2 lo, u32x4 up) {
    const f32x2 alpha_b = { h2f(up.y >> 16), h2f(up.w >> 16) };       // b.a * 1.0f
    const f32x2 alpha_a = lo.a * (1.0f - alpha_b);
    const f32x2 a = alpha_a + alpha_b;
    // the refined reciprocal of the blended alpha is started here, ahead of the products it does not depend on, so that the
    // two waits of its chain (reciprocal -> packed FMA -> packed FMA) are filled with them; unused when every alpha is 1.0f
    f32x2 r = { __builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y) };
    const f32x2 nd = -a;
    px32x2 o;
    f32x2 pr, pg, pb;
    if constexpr (!kContract) { pr = lo.r * alpha_a; pg = lo.g * alpha_a; pb = lo.b * alpha_a; }
    f32x2 qr, qg, qb;
    halves_times(up, alpha_b, qr, qg, qb);
    const f32x2 r1 = fma2(nd, r, f32x2{ 1.0f, 1.0f });
    if constexpr (kContract) {
        // clang's build: fma(lower, alpha_a, upper * alpha_b) -- the upper product rounded, the lower one fused
        o.r = fma2(lo.r, alpha_a, qr); o.g = fma2(lo.g, alpha_a, qg);
        r = fma2(r1, r, r);
        asm volatile("" : "+v"(r));
        o.b = fma2(lo.b, alpha_a, qb);
    } else {
        o.r = pr + qr; o.g = pg + qg;
        r = fma2(r1, r, r);
        asm volatile("" : "+v"(r));                  // (keeps the chain up here: hipcc would sink it into the branch that uses it)
        o.b = pb + qb;
    }
    o.a = a;
    if (wave_any(!(a.x == 1.0f && a.y == 1.0f))) {
        // the quotients are formed without asking first (outside the band they are garbage, never a trap) and replaced on
        // the plain path: written this way round, the copies that join the two paths sit on the rare one
        const bool in_band = div_band_lean(o.r, o.g, o.b, a);
        f32x2 q0 = o.r * r, q1 = o.g * r, q2 = o.b * r;
        f32x2 e0 = fma2(nd, q0, o.r), e1 = fma2(nd, q1, o.g), e2 = fma2(nd, q2, o.b);
        q0 = fma2(e0, r, q0); q1 = fma2(e1, r, q1); q2 = fma2(e2, r, q2);
        e0 = fma2(nd, q0, o.r); e1 = fma2(nd, q1, o.g); e2 = fma2(nd, q2, o.b);
        q0 = fma2(e0, r, q0); q1 = fma2(e1, r, q1); q2 = fma2(e2, r, q2);
        if (wave_any(!in_band)) {
            rare_path();
            q0 = f32x2{ o.r.x / a.x, o.r.y / a.y };
            q1 = f32x2{ o.g.x / a.x, o.g.y / a.y };
            q2 = f32x2{ o.b.x / a.x, o.b.y / a.y };
            if (a.x == 0.0f) { q0.x = 0.0f; q1.x = 0.0f; q2.x = 0.0f; o.a.x = 0.0f; }
            if (a.y == 0.0f) { q0.y = 0.0f; q1.y = 0.0f; q2.y = 0.0f; o.a.y = 0.0f; }
        }
        o.r = q0; o.g = q1; o.b = q2;
    }
    return o;
}

__device__ __forceinline__ u32x4 narrow_pair_lean(px32x2 v) {
    float big, t0, t1;
    asm("v_max3_f32 %1, |%3|, |%4|, |%5|\n\t"
        "v_max3_f32 %2, |%6|, |%7|, |%8|\n\t"
        "v_max3_f32 %0, |%9|, |%10|, %1\n\t"
        "v_max_f32 %0, %0, %2"
        : "=&v"(big), "=&v"(t0), "=&v"(t1)
        : "v"(v.r.x), "v"(v.g.x), "v"(v.b.x), "v"(v.a.x), "v"(v.r.y), "v"(v.g.y), "v"(v.b.y), "v"(v.a.y));
    if (wave_any(!(big < 65536.0f))) { rare_path(); v.r = saturate_to_inf2(v.r); v.g = saturate_to_inf2(v.g); v.b = saturate_to_inf2(v.b); v.a = saturate_to_inf2(v.a); }
    return u32x4{ pkrtz(v.r.x, v.g.x), pkrtz(v.b.x, v.a.x), pkrtz(v.r.y, v.g.y), pkrtz(v.b.y, v.a.y) };
}

// What a launch of the chain kernel does with its layers (one instantiation each, so that the trip loop carries no
// wave-uniform mode branches and no dead code of the other modes):
//   CHAIN_GRADE  every layer through the colour filter, then the stack      (config 2)
//   CHAIN_PLAIN  layers widened as they are, then the stack                 (a VideoWorkspace of half-native items, config 4)
//   CHAIN_CROSS  two layers crossfaded with weights wa, wb                  (VideoMixFilter on half-native inputs)
enum { CHAIN_GRADE = 0, CHAIN_PLAIN = 1, CHAIN_CROSS = 2 };

__device__ __forceinline__ px32x2 widen_pair(u32x4 p) { return widen2(make_uint4(p.x, p.y, p.z, p.w)); }

template <int NL, bool PRE, bool POST, int MODE>
__device__ __forceinline__ u32x4 chain_pair_lean(const u32x4 (&w)[NL], const MatR &mat, const uint16_t *lut, const uint16_t *post) {
    if constexpr (MODE == CHAIN_CROSS) {
        static_assert(NL == 2, "a crossfade has two inputs");
        return narrow_pair(cross_pair(widen_pair(w[0]), widen_pair(w[1]), mat.wa, mat.wb));     // widen, video_mix.c:193-205, truncate
    } else {
        px32x2 acc = MODE == CHAIN_PLAIN ? widen_pair(w[0]) : grade_pair<PRE, POST>(w[0], mat, lut, post);
#pragma unroll
        for (int k = 1; k < NL; k++) acc = over_pair(acc, MODE == CHAIN_PLAIN ? widen_pair(w[k]) : grade_pair<PRE, POST>(w[k], mat, lut, post));
        return narrow_pair(acc);
    }
}

// the colour filter alone, codes in -> codes out (color.c structure on a pair of pixels)
template <bool PRE, bool POST>
__device__ __forceinline__ u32x4 color_pair_codes(u32x4 p, const MatR &mat, const uint16_t *lds_lut, const uint16_t *glb_post) {
    uint32_t c[8] = { p.x & 0xFFFFu, p.x >> 16, p.y & 0xFFFFu, p.y >> 16, p.z & 0xFFFFu, p.z >> 16, p.w & 0xFFFFu, p.w >> 16 };
    if (PRE) {
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = lds_lut[c[i]];
    }
    px32x2 v;
    v.r = f32x2{ h2f(c[0]), h2f(c[4]) };
    v.g = f32x2{ h2f(c[1]), h2f(c[5]) };
    v.b = f32x2{ h2f(c[2]), h2f(c[6]) };
    px32x2 o = mat3x2(v, mat);
    const float big = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.r.x), __builtin_fabsf(o.g.x)), __builtin_fabsf(o.b.x)),
                                      __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.r.y), __builtin_fabsf(o.g.y)), __builtin_fabsf(o.b.y)));
    if (wave_any(big >= 65536.0f)) { rare_path(); o.r = saturate_to_inf2(o.r); o.g = saturate_to_inf2(o.g); o.b = saturate_to_inf2(o.b); }
    uint32_t rr = pkrtz(o.r.x, o.r.y), gg = pkrtz(o.g.x, o.g.y), bb = pkrtz(o.b.x, o.b.y);    // lo: first pixel, hi: second
    uint32_t q[8] = { rr & 0xFFFFu, gg & 0xFFFFu, bb & 0xFFFFu, c[3], rr >> 16, gg >> 16, bb >> 16, c[7] };
    constexpr bool post_lds = POST && !PRE, post_glb = POST && PRE;
    if (post_lds) {
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = lds_lut[q[i]];
    }
    if (post_glb) {
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = glb_post[q[i]];
    }
    return u32x4{ q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (q[5] << 16), q[6] | (q[7] << 16) };
}

}  // namespace cvs
