// dv_ops.hip -- the DV 4:1:1 edge: planar 8-bit Y'CbCr (720x480, chroma 180 wide, co-sited left) <-> half RGBA.
//
//   k_dv_reconstruct   src/cprocess/video_reconstruct.c:50-137: per pixel, chroma = the triangle-weighted sum of
//                      the one or two chroma samples whose support covers it, accumulated from 0.0f in ascending
//                      sample order (the reference scatters each sample over a zeroed row, :95-109); Y'CbCr ->
//                      R'G'B' with the Rec.709 matrix evaluated left to right, truncate to half, Rec.709 ->
//                      linear table over all four halfs.
//   k_dv_luma / k_dv_chroma   src/cprocess/video_subsample.c:99-187: linear -> Rec.709 table over all four
//                      halfs, widen, Y' = (uint8)(y * 219 + 16) per pixel; Cb/Cr = gather of seven neighbours'
//                      Pb/Pr through the normalised triangle, from 0.0f ascending, (uint8)(c * 224 + 128).
//                      (uint8)float is what the reference's x86 build does: truncate to int32, keep the low byte.
//   k_dv_encode        the in-place transfer encode of the input rows that video_subsample.c:144 leaves behind.
// A DV frame is 345 600 pixels: these kernels are launch-bound; one lane per output, tables gathered from L2.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "pixel_math.hpp"

namespace {

constexpr int kW = 720, kSub = 4, kOffY = -1, kBlock = 256;

__device__ __forceinline__ uint2 *px16(const cvk_view &v, int x, int y) {
    return reinterpret_cast<uint2 *>(v.data) + (size_t)(y - v.fy0) * (size_t)v.pitch + (size_t)(x - v.fx0);
}

__global__ __launch_bounds__(kBlock) void k_dv_reconstruct(cvk_view frame, cvk_rect cur, cvk_dv_planes pl, cvk_dv_taps tri, const uint16_t *__restrict__ lut) {
    const int x = cur.x0 + (int)(blockIdx.x * kBlock + threadIdx.x), y = cur.y0 + (int)blockIdx.y;
    if (x > cur.x1) return;
    const int row = y - kOffY;
    const uint8_t *cbrow = pl.cb + (size_t)row * pl.scb, *crrow = pl.cr + (size_t)row * pl.scr;
    // chroma samples xs with 4*xs - center <= x <= 4*xs + (width - center - 1), ascending
    const int hi_off = tri.width - tri.center - 1;
    int lo = x - hi_off;
    lo = lo <= 0 ? 0 : (lo + kSub - 1) / kSub;
    int hi = (x + tri.center) / kSub;
    hi = hi > (kW - 1) / kSub ? (kW - 1) / kSub : hi;
    float cb = 0.0f, cr = 0.0f;
    for (int xs = lo; xs <= hi; xs++) {
        const float c = tri.coeff[x - xs * kSub + tri.center];
        cb = cvs::madd(((float)cbrow[xs] - 128.0f) / 224.0f, c, cb);
        cr = cvs::madd(((float)crrow[xs] - 128.0f) / 224.0f, c, cr);
    }
    const float yy = ((float)pl.y[(size_t)row * pl.sy + x] - 16.0f) / 219.0f;
    const float r = cvs::madd(cr, 1.5748f, cvs::madd(yy, 1.0f, cb * 0.0f));
    const float g = cvs::madd(cr, -0.468124f, cvs::madd(yy, 1.0f, cb * -0.187324f));
    const float b = cvs::madd(cr, 0.0f, cvs::madd(yy, 1.0f, cb * 1.8556f));
    const uint32_t rg = cvs::f2h_rz2(r, g), ba = cvs::f2h_rz2(b, 1.0f);
    *px16(frame, x, y) = make_uint2((uint32_t)lut[rg & 0xFFFFu] | ((uint32_t)lut[rg >> 16] << 16),
                                    (uint32_t)lut[ba & 0xFFFFu] | ((uint32_t)lut[ba >> 16] << 16));
}

struct Ypbpr { float y, pb, pr; };

__device__ __forceinline__ Ypbpr encode(uint2 p, const uint16_t *lut) {
    const float r = cvs::h2f(lut[p.x & 0xFFFFu]), g = cvs::h2f(lut[p.x >> 16]), b = cvs::h2f(lut[p.y & 0xFFFFu]);
    Ypbpr o;
    o.y = cvs::madd(b, 0.0722f, cvs::madd(r, 0.2126f, g * 0.7152f));
    o.pb = cvs::madd(b, 0.5f, cvs::madd(r, -0.114572f, g * -0.385428f));
    o.pr = cvs::madd(b, -0.045847f, cvs::madd(r, 0.5f, g * -0.454153f));
    return o;
}

__device__ __forceinline__ uint8_t low_byte(float v) { return (uint8_t)(int32_t)v; }

__global__ __launch_bounds__(kBlock) void k_dv_luma(cvk_dv_planes pl, cvk_view frame, cvk_rect w, const uint16_t *__restrict__ lut) {
    const int x = w.x0 + (int)(blockIdx.x * kBlock + threadIdx.x), y = w.y0 + (int)blockIdx.y;
    if (x > w.x1) return;
    const Ypbpr e = encode(*px16(frame, x, y), lut);
    pl.y[(size_t)(y - kOffY) * pl.sy + x] = low_byte(cvs::madd(e.y, 219.0f, 16.0f));
}

__global__ __launch_bounds__(kBlock) void k_dv_chroma(cvk_dv_planes pl, cvk_view frame, cvk_rect w, cvk_dv_taps tri, const uint16_t *__restrict__ lut) {
    const int tx = w.x0 / kSub + (int)(blockIdx.x * kBlock + threadIdx.x), y = w.y0 + (int)blockIdx.y;
    if (tx > w.x1 / kSub) return;
    int lo = tx * kSub - tri.center, hi = tx * kSub + (tri.width - tri.center - 1);
    lo = lo < w.x0 ? w.x0 : lo;
    hi = hi > w.x1 ? w.x1 : hi;
    float cb = 0.0f, cr = 0.0f;
    for (int sx = lo; sx <= hi; sx++) {
        const Ypbpr e = encode(*px16(frame, sx, y), lut);
        const float c = tri.coeff[sx - tx * kSub + tri.center];
        cb = cvs::madd(e.pb, c, cb);
        cr = cvs::madd(e.pr, c, cr);
    }
    const int row = y - kOffY;
    pl.cb[(size_t)row * pl.scb + tx] = low_byte(cvs::madd(cb, 224.0f, 128.0f));
    pl.cr[(size_t)row * pl.scr + tx] = low_byte(cvs::madd(cr, 224.0f, 128.0f));
}

__global__ __launch_bounds__(kBlock) void k_dv_encode(cvk_view frame, cvk_rect w, const uint16_t *__restrict__ lut) {
    const int x = w.x0 + (int)(blockIdx.x * kBlock + threadIdx.x), y = w.y0 + (int)blockIdx.y;
    if (x > w.x1) return;
    uint2 *p = px16(frame, x, y);
    const uint2 v = *p;
    *p = make_uint2((uint32_t)lut[v.x & 0xFFFFu] | ((uint32_t)lut[v.x >> 16] << 16), (uint32_t)lut[v.y & 0xFFFFu] | ((uint32_t)lut[v.y >> 16] << 16));
}

}  // namespace

extern "C" int cvk_dv_reconstruct(cvk_view frame, cvk_rect cur, const cvk_dv_planes *pl, const cvk_dv_taps *tri, const uint16_t *lut, void *stream) {
    if (cur.x1 < cur.x0 || cur.y1 < cur.y0) return 0;
    dim3 grid((unsigned)((cur.x1 - cur.x0 + kBlock) / kBlock), (unsigned)(cur.y1 - cur.y0 + 1));
    hipLaunchKernelGGL(k_dv_reconstruct, grid, dim3(kBlock), 0, (hipStream_t)stream, frame, cur, *pl, *tri, lut);
    return (int)hipGetLastError();
}

extern "C" int cvk_dv_subsample(const cvk_dv_planes *pl, cvk_view frame, cvk_rect w, const cvk_dv_taps *tri, const uint16_t *lut, int encode_in_place, void *stream) {
    if (w.x1 < w.x0 || w.y1 < w.y0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const unsigned rows = (unsigned)(w.y1 - w.y0 + 1);
    hipLaunchKernelGGL(k_dv_luma, dim3((unsigned)((w.x1 - w.x0 + kBlock) / kBlock), rows), dim3(kBlock), 0, s, *pl, frame, w, lut);
    hipLaunchKernelGGL(k_dv_chroma, dim3((unsigned)((w.x1 / kSub - w.x0 / kSub + kBlock) / kBlock), rows), dim3(kBlock), 0, s, *pl, frame, w, *tri, lut);
    if (encode_in_place) hipLaunchKernelGGL(k_dv_encode, dim3((unsigned)((w.x1 - w.x0 + kBlock) / kBlock), rows), dim3(kBlock), 0, s, frame, w, lut);
    return (int)hipGetLastError();
}
