// frame_ops.hip -- flat half<->float conversion and the windowed one-input frame kernels.
//
// Bound: HBM.  Algorithmic bytes per element / pixel are listed per kernel.  Every kernel is a
// pure stream: consecutive lanes touch consecutive 8- or 16-byte words, no LDS, no reuse.
//   flat widen   : 2 B read + 4 B written per half          (A1, half.c:62-65)
//   flat narrow  : 4 B read + 2 B written per half          (A2, half.c:67-70)
//   copy f16     : 8 + 8 B per pixel                        (A5, video_mix.c:27-44)
//   copy a f32   : 16 + 16 B per pixel                      (A5, video_mix.c:73-105)
//   widen frame  : 8 + 16 B per pixel                       (A4, main.c:115-139)
//   narrow frame : 16 + 8 B per pixel                       (A4, main.c:43-71)
//   fill         : 8 / 16 B written per pixel               (A16, SolidColorVideoSource.c:52-101)
//   (gain/offset, the one kernel of this family with arithmetic, lives in gain_ops.hip: built in both flavours)
#include "kernels.h"
#include "pixel_math.hpp"

using namespace cvs;

namespace {

constexpr int kBlock = 256;

// ---------------------------------------------------------------- flat arrays

// 8 halfs (16 B) in, 8 floats (2 x 16 B) out per lane and step
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_widen_flat(float *__restrict__ out, const uint16_t *__restrict__ in, size_t count) {
    size_t nvec = count / 8;
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
        uint4 v = reinterpret_cast<const uint4 *>(in)[i];
        uint32_t w[4] = { v.x, v.y, v.z, v.w };
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            f[2 * k] = FAST ? h2f_fast(w[k] & 0xFFFFu) : h2f(w[k] & 0xFFFFu);
            f[2 * k + 1] = FAST ? h2f_fast(w[k] >> 16) : h2f(w[k] >> 16);
        }
        float4 *o = reinterpret_cast<float4 *>(out) + 2 * i;
        o[0] = make_float4(f[0], f[1], f[2], f[3]);
        o[1] = make_float4(f[4], f[5], f[6], f[7]);
    }
    // ragged tail (count % 8), one lane each
    size_t tail = nvec * 8 + (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (blockIdx.x == 0 && tail < count) out[tail] = FAST ? h2f_fast(in[tail]) : h2f(in[tail]);
}

template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_narrow_flat(uint16_t *__restrict__ out, const float *__restrict__ in, size_t count) {
    size_t nvec = count / 8;
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
        const float4 *p = reinterpret_cast<const float4 *>(in) + 2 * i;
        float4 a = p[0], b = p[1];
        uint4 o;
        if (FAST) {
            o.x = f2h_fast(a.x) | (f2h_fast(a.y) << 16);
            o.y = f2h_fast(a.z) | (f2h_fast(a.w) << 16);
            o.z = f2h_fast(b.x) | (f2h_fast(b.y) << 16);
            o.w = f2h_fast(b.z) | (f2h_fast(b.w) << 16);
        } else {
            o.x = f2h_rz2(a.x, a.y);
            o.y = f2h_rz2(a.z, a.w);
            o.z = f2h_rz2(b.x, b.y);
            o.w = f2h_rz2(b.z, b.w);
        }
        reinterpret_cast<uint4 *>(out)[i] = o;
    }
    size_t tail = nvec * 8 + (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (blockIdx.x == 0 && tail < count) out[tail] = (uint16_t)(FAST ? f2h_fast(in[tail]) : f2h_rz(in[tail]));
}

// element-per-lane forms for buffers that are not 16-byte aligned
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_widen_scalar(float *__restrict__ out, const uint16_t *__restrict__ in, size_t count) {
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride)
        out[i] = FAST ? h2f_fast(in[i]) : h2f(in[i]);
}
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_narrow_scalar(uint16_t *__restrict__ out, const float *__restrict__ in, size_t count) {
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride)
        out[i] = (uint16_t)(FAST ? f2h_fast(in[i]) : f2h_rz(in[i]));
}

// ---------------------------------------------------------------- windowed frame kernels
// grid.x covers the rect's columns, grid.y its rows; one pixel per lane.

template <typename T>
__device__ __forceinline__ T *at(const cvk_view &v, int x, int y) {
    return reinterpret_cast<T *>(v.data) + (size_t)(y - v.fy0) * (size_t)v.pitch + (size_t)(x - v.fx0);
}

#define CVK_PIXEL_XY(r)                                         \
    int x = (r).x0 + (int)(blockIdx.x * kBlock + threadIdx.x);  \
    int y = (r).y0 + (int)blockIdx.y;                           \
    if (x > (r).x1) return;

__global__ __launch_bounds__(kBlock) void k_copy16(cvk_view out, cvk_view in, cvk_rect r) {
    CVK_PIXEL_XY(r)
    *at<uint2>(out, x, y) = *at<const uint2>(in, x, y);
}

// 2:3 pulldown removal, the mixed-field case (src/process/Pulldown23RemovalFilter.c:88-104): the even rows of the
// frame's current window are replaced by rows of `other`, a packed frame allocated for exactly that window.  The
// reference addresses `other` from x = 0, not from the window's min.x (:101): the row it copies starts cur.x0 pixels
// before the row it means.  Reproduced as linear addressing inside the allocation; outside it (the reference reads
// foreign memory there) and outside other's current window (uninitialised there) the pixel is zero.
__global__ __launch_bounds__(kBlock) void k_weave16(cvk_view frame, cvk_rect cur, const uint2 *__restrict__ other, cvk_rect ocur, int first) {
    const int width = cur.x1 - cur.x0 + 1;
    const int c = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (c >= width) return;
    const int i = first + 2 * (int)blockIdx.y;
    const long long n = (long long)width * (long long)(cur.y1 - cur.y0 + 1);
    const long long l = (long long)(i - cur.y0) * (long long)width - (long long)cur.x0 + (long long)c;
    uint2 v = make_uint2(0u, 0u);
    if (l >= 0 && l < n) {
        const int ty = cur.y0 + (int)(l / width), tx = cur.x0 + (int)(l % width);
        if (tx >= ocur.x0 && tx <= ocur.x1 && ty >= ocur.y0 && ty <= ocur.y1) v = other[l];
    }
    *at<uint2>(frame, cur.x0 + c, i) = v;
}

__global__ __launch_bounds__(kBlock) void k_copy_alpha32(cvk_view out, cvk_view in, cvk_rect r, float alpha, int scale) {
    CVK_PIXEL_XY(r)
    float4 v = *at<const float4>(in, x, y);
    if (scale) v.w = v.w * alpha;
    *at<float4>(out, x, y) = v;
}

__global__ __launch_bounds__(kBlock) void k_widen(cvk_view out, cvk_view in, cvk_rect r) {
    CVK_PIXEL_XY(r)
    px32 v = widen(*at<const uint2>(in, x, y));
    *at<float4>(out, x, y) = make_float4(v.r, v.g, v.b, v.a);
}

__global__ __launch_bounds__(kBlock) void k_narrow(cvk_view out, cvk_view in, cvk_rect r) {
    CVK_PIXEL_XY(r)
    float4 v = *at<const float4>(in, x, y);
    *at<uint2>(out, x, y) = narrow({ v.x, v.y, v.z, v.w });
}

// the colour arrives as four floats and is truncated here (SolidColorVideoSource.c:68-69 converts once per frame
// with rgba_f32_to_f16: same truncation, no round trip to the host for the bits)
__global__ __launch_bounds__(kBlock) void k_fill16(cvk_view out, cvk_rect r, float4 c) {
    CVK_PIXEL_XY(r)
    *at<uint2>(out, x, y) = make_uint2(cvs::f2h_rz2(c.x, c.y), cvs::f2h_rz2(c.z, c.w));
}

__global__ __launch_bounds__(kBlock) void k_fill32(cvk_view out, cvk_rect r, float4 c) {
    CVK_PIXEL_XY(r)
    *at<float4>(out, x, y) = c;
}

__global__ __launch_bounds__(kBlock) void k_zero32(float4 *p, size_t n) {
    size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

inline dim3 rect_grid(const cvk_rect &r) {
    return dim3((unsigned)((r.x1 - r.x0 + 1 + kBlock - 1) / kBlock), (unsigned)(r.y1 - r.y0 + 1), 1);
}
inline bool rect_empty(const cvk_rect &r) { return r.x1 < r.x0 || r.y1 < r.y0; }
inline unsigned flat_grid(size_t nvec) {
    size_t b = (nvec + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > 256 * 8) b = 256 * 8;       // 8 blocks per CU, grid-stride the rest
    return (unsigned)b;
}

}  // namespace

#define LAUNCH(kern, grid, ...)                                                        \
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), 0, (hipStream_t)stream, __VA_ARGS__); \
    return (int)hipGetLastError();

static inline bool aligned16(const void *a, const void *b) { return (((uintptr_t)a | (uintptr_t)b) & 15u) == 0; }

extern "C" int cvk_half_to_float(float *out, const uint16_t *in, size_t count, int fast, void *stream) {
    if (!count) return 0;
    if (!aligned16(out, in)) {
        if (fast) { LAUNCH(k_widen_scalar<true>, dim3(flat_grid(count)), out, in, count) }
        LAUNCH(k_widen_scalar<false>, dim3(flat_grid(count)), out, in, count)
    }
    if (fast) { LAUNCH(k_widen_flat<true>, dim3(flat_grid(count / 8)), out, in, count) }
    LAUNCH(k_widen_flat<false>, dim3(flat_grid(count / 8)), out, in, count)
}

extern "C" int cvk_float_to_half(uint16_t *out, const float *in, size_t count, int fast, void *stream) {
    if (!count) return 0;
    if (!aligned16(out, in)) {
        if (fast) { LAUNCH(k_narrow_scalar<true>, dim3(flat_grid(count)), out, in, count) }
        LAUNCH(k_narrow_scalar<false>, dim3(flat_grid(count)), out, in, count)
    }
    if (fast) { LAUNCH(k_narrow_flat<true>, dim3(flat_grid(count / 8)), out, in, count) }
    LAUNCH(k_narrow_flat<false>, dim3(flat_grid(count / 8)), out, in, count)
}

extern "C" int cvk_copy_f16(cvk_view out, cvk_view in, cvk_rect r, void *stream) {
    if (rect_empty(r)) return 0;
    LAUNCH(k_copy16, rect_grid(r), out, in, r)
}

extern "C" int cvk_weave_f16(cvk_view frame, cvk_rect cur, const void *other, cvk_rect other_cur, void *stream) {
    if (rect_empty(cur)) return 0;
    const int first = (cur.y0 + 1) & ~1;                  // Pulldown23RemovalFilter.c:100
    if (first > cur.y1) return 0;
    dim3 grid((unsigned)((cur.x1 - cur.x0 + 1 + kBlock - 1) / kBlock), (unsigned)((cur.y1 - first) / 2 + 1));
    LAUNCH(k_weave16, grid, frame, cur, reinterpret_cast<const uint2 *>(other), other_cur, first)
}

extern "C" int cvk_copy_alpha_f32(cvk_view out, cvk_view in, cvk_rect r, float alpha, void *stream) {
    if (rect_empty(r)) return 0;
    LAUNCH(k_copy_alpha32, rect_grid(r), out, in, r, alpha, alpha != 1.0f ? 1 : 0)
}

extern "C" int cvk_widen(cvk_view out32, cvk_view in16, cvk_rect r, void *stream) {
    if (rect_empty(r)) return 0;
    LAUNCH(k_widen, rect_grid(r), out32, in16, r)
}

extern "C" int cvk_narrow(cvk_view out16, cvk_view in32, cvk_rect r, void *stream) {
    if (rect_empty(r)) return 0;
    LAUNCH(k_narrow, rect_grid(r), out16, in32, r)
}

extern "C" int cvk_fill_f16(cvk_view out, cvk_rect r, const float c[4], void *stream) {
    if (rect_empty(r)) return 0;
    LAUNCH(k_fill16, rect_grid(r), out, r, make_float4(c[0], c[1], c[2], c[3]))
}

extern "C" int cvk_fill_f32(cvk_view out, cvk_rect r, const float c[4], void *stream) {
    if (rect_empty(r)) return 0;
    LAUNCH(k_fill32, rect_grid(r), out, r, make_float4(c[0], c[1], c[2], c[3]))
}

extern "C" int cvk_zero_f32(cvk_view v, void *stream) {
    size_t n = (size_t)v.pitch * (size_t)(v.fy1 - v.fy0 + 1);
    if (!n) return 0;
    LAUNCH(k_zero32, dim3(flat_grid(n)), reinterpret_cast<float4 *>(v.data), n)
}
