// gain_ops.hip -- VideoGainOffsetFilter on an f16 frame (A15): rgb * gain + offset, alpha untouched.
//
// The reference has this only as GLSL (src/cprocess/video_filter.c:34-39; window rule src/cprocess/gl.c:584), so the rounding
// is defined here: widen, c * gain + offset in f32 -- two roundings in the plain build, ONE in the contracted build (the
// expression is a single a * b + c, which the reference's clang build would fuse; pixel_math.hpp) -- truncate.
// Bound: HBM, 8 B read + 8 B written per pixel; one pixel per lane, grid.x over the rect's columns, grid.y its rows.
#include "kernels.h"
#include "pixel_math.hpp"

namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void k_gain_offset(cvk_view out, cvk_view in, cvk_rect r, float gain, float offset) {
    const int x = r.x0 + (int)(blockIdx.x * kBlock + threadIdx.x), y = r.y0 + (int)blockIdx.y;
    if (x > r.x1) return;
    const uint2 *src = reinterpret_cast<const uint2 *>(in.data) + (size_t)(y - in.fy0) * (size_t)in.pitch + (size_t)(x - in.fx0);
    uint2 *dst = reinterpret_cast<uint2 *>(out.data) + (size_t)(y - out.fy0) * (size_t)out.pitch + (size_t)(x - out.fx0);
    cvs::px32 v = cvs::widen(*src);
    v.r = cvs::madd(v.r, gain, offset);
    v.g = cvs::madd(v.g, gain, offset);
    v.b = cvs::madd(v.b, gain, offset);
    *dst = cvs::narrow(v);
}

}  // namespace

extern "C" int cvk_gain_offset_f16(cvk_view out, cvk_view in, cvk_rect r, float gain, float offset, void *stream) {
    if (r.x1 < r.x0 || r.y1 < r.y0) return 0;
    const dim3 grid((unsigned)((r.x1 - r.x0 + 1 + kBlock - 1) / kBlock), (unsigned)(r.y1 - r.y0 + 1), 1);
    hipLaunchKernelGGL(k_gain_offset, grid, dim3(kBlock), 0, (hipStream_t)stream, out, in, r, gain, offset);
    return (int)hipGetLastError();
}
