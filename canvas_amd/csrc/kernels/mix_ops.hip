// mix_ops.hip -- crossfade and un-premultiplied alpha-over on f32 frames in HBM.
//
// Replaces src/cprocess/video_mix.c:107-235 (cross) and :237-370 (over).  The reference walks nine
// rectangular regions with row loops; here one lane owns one pixel of `outer` and decides which
// region it is in.  The host (csrc/host/mix.c) has already made every decision the reference makes
// once per call -- outer/inner, the gap flags, which input is "top/bottom/left/right" (including
// the reference's min.x-vs-min.y comparison for `left`, video_mix.c:137,265) -- so the result is
// the reference's for every window configuration, junk included.
//
// Bound: HBM.  Algorithmic bytes per outer pixel: over = 16 (out r) + 16 (b r) + 16 (out w) = 48;
// cross = 16 + 16 + 16 = 48.  3 IEEE divides per blended pixel (v_div_scale/fmas/fixup).
#include "kernels.h"
#include "pixel_math.hpp"

using namespace cvs;

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float4 *at(const cvk_view &v, int x, int y) {
    return reinterpret_cast<float4 *>(v.data) + (size_t)(y - v.fy0) * (size_t)v.pitch + (size_t)(x - v.fx0);
}
// A lone-region read can fall outside the source's CURRENT window -- even outside its row (the `left` selector
// quirk, video_mix.c:137,265): the reference indexes `row[x]` with whatever x the region has, i.e. it reads
// data + (y - y0) * pitch + (x - x0) wherever that lands.  Inside the allocation that is a neighbouring row's pixel,
// reproduced here; outside it the reference reads foreign memory, here the pixel is zero and nothing faults.
__device__ __forceinline__ float4 fetch(const cvk_view &v, int x, int y) {
    const long long off = (long long)(y - v.fy0) * (long long)v.pitch + (long long)(x - v.fx0);
    const long long n = (long long)v.pitch * (long long)(v.fy1 - v.fy0 + 1);
    return (off >= 0 && off < n) ? reinterpret_cast<const float4 *>(v.data)[off] : make_float4(0.f, 0.f, 0.f, 0.f);
}

enum Action { LEAVE, ZERO, COPY_P, COPY_Q, BLEND };

__global__ __launch_bounds__(kBlock) void k_mix(cvk_mix_params mp) {
    int x = mp.outer.x0 + (int)(blockIdx.x * kBlock + threadIdx.x);
    int y = mp.outer.y0 + (int)blockIdx.y;
    if (x > mp.outer.x1) return;

    int act;
    if (y < mp.inner.y0 || y > mp.inner.y1) {
        // rows where a single frame lives: zero | that frame | zero  (video_mix.c:142-159, 217-232)
        bool is_p = (y < mp.inner.y0) ? mp.top_is_p : mp.bottom_is_p;
        const cvk_rect &w = is_p ? mp.pw : mp.qw;
        if (x < w.x0 || x > w.x1) act = ZERO;
        else act = is_p ? (mp.p_in_place ? LEAVE : COPY_P) : COPY_Q;
    } else if (mp.gap_y) {
        // rows between two vertically disjoint frames: only the inner column span is cleared (:163-170)
        act = (x >= mp.inner.x0 && x <= mp.inner.x1) ? ZERO : LEAVE;
    } else if (x < mp.inner.x0) {
        act = mp.left_is_p ? (mp.p_in_place ? LEAVE : COPY_P) : COPY_Q;
    } else if (x > mp.inner.x1) {
        act = mp.right_is_p ? (mp.p_in_place ? LEAVE : COPY_P) : COPY_Q;
    } else {
        act = mp.gap_x ? ZERO : BLEND;
    }

    float4 o;
    switch (act) {
    case LEAVE: return;
    case ZERO: o = make_float4(0.f, 0.f, 0.f, 0.f); break;
    case COPY_P: o = fetch(mp.p, x, y); o.w = o.w * mp.wp; break;
    case COPY_Q: o = fetch(mp.q, x, y); o.w = o.w * mp.wq; break;
    default: {
        float4 a = *at(mp.p, x, y), b = *at(mp.q, x, y);
        px32 r = (mp.mode == CVK_MIX_OVER) ? blend_over({ a.x, a.y, a.z, a.w }, { b.x, b.y, b.z, b.w }, mp.wq)
                                           : blend_cross({ a.x, a.y, a.z, a.w }, { b.x, b.y, b.z, b.w }, mp.wp, mp.wq);
        o = make_float4(r.r, r.g, r.b, r.a);
    }
    }
    *at(mp.out, x, y) = o;
}

}  // namespace

extern "C" int cvk_mix(const cvk_mix_params *mp, void *stream) {
    if (mp->outer.x1 < mp->outer.x0 || mp->outer.y1 < mp->outer.y0) return 0;
    dim3 grid((unsigned)((mp->outer.x1 - mp->outer.x0 + 1 + kBlock - 1) / kBlock), (unsigned)(mp->outer.y1 - mp->outer.y0 + 1));
    hipLaunchKernelGGL(k_mix, grid, dim3(kBlock), 0, (hipStream_t)stream, *mp);
    return (int)hipGetLastError();
}
