// fir_ops.hip -- one separable FIR pass (resample or blur) as a gather over per-line tap lists.
//
// Replaces the inner loops of src/cprocess/video_scale.c:63-122 (vertical) and :161-226
// (horizontal).  The reference regenerates the triangle filter for every source (upscale, scatter)
// or target (downscale, gather) line and accumulates `t += s * coeff` into a zero-filled target.
// Here the host turns either form into, per TARGET line, the list of (source line, coefficient)
// pairs in ascending source order -- the order in which the reference's loops reach that target
// pixel -- and one lane accumulates one output pixel from 0.0f with separate mul and add.
// Lines that receive nothing keep the zeros written by cvk_zero_f32 (video_scale.c:25-32).
//
// Bound: HBM.  Algorithmic bytes per pass: 16 B per source pixel read once + 16 B per target pixel
// written; the tap re-reads (2-11 per output) are served by L1/L2 because neighbouring lanes and
// neighbouring taps touch neighbouring lines.
#include "kernels.h"
#include "pixel_math.hpp"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float4 *at(const cvk_view &v, int x, int y) {
    return reinterpret_cast<float4 *>(v.data) + (size_t)(y - v.fy0) * (size_t)v.pitch + (size_t)(x - v.fx0);
}

// axis 0: lines are rows -> grid.y walks target rows, lanes walk x (coalesced both sides)
// axis 1: lines are columns -> lanes walk target columns (each with its own tap list), grid.y walks rows
__global__ __launch_bounds__(kBlock) void k_fir(cvk_fir_params fp) {
    int line, other;
    if (fp.axis == 0) {
        other = fp.lo + (int)(blockIdx.x * kBlock + threadIdx.x);
        line = fp.t0 + (int)blockIdx.y;
        if (other > fp.hi) return;
    } else {
        line = fp.t0 + (int)(blockIdx.x * kBlock + threadIdx.x);
        other = fp.lo + (int)blockIdx.y;
        if (line > fp.t1) return;
    }
    const int row = line - fp.t0;
    const int n = fp.ntaps[row];
    const int *src = fp.tap_src + (size_t)row * fp.stride;
    const float *c = fp.taps + (size_t)row * fp.stride;
    float r = 0.0f, g = 0.0f, b = 0.0f, a = 0.0f;
    for (int k = 0; k < n; k++) {
        const int s = src[k];
        const float w = c[k];
        const float4 v = fp.axis == 0 ? *at(fp.source, other, s) : *at(fp.source, s, other);
        r = r + v.x * w;
        g = g + v.y * w;
        b = b + v.z * w;
        a = a + v.w * w;
    }
    *(fp.axis == 0 ? at(fp.target, other, line) : at(fp.target, line, other)) = make_float4(r, g, b, a);
}

}  // namespace

extern "C" int cvk_fir_gather(const cvk_fir_params *fp, void *stream) {
    if (fp->t1 < fp->t0 || fp->hi < fp->lo) return 0;
    const int lines = fp->t1 - fp->t0 + 1, span = fp->hi - fp->lo + 1;
    dim3 grid = fp->axis == 0 ? dim3((unsigned)((span + kBlock - 1) / kBlock), (unsigned)lines)
                              : dim3((unsigned)((lines + kBlock - 1) / kBlock), (unsigned)span);
    hipLaunchKernelGGL(k_fir, grid, dim3(kBlock), 0, (hipStream_t)stream, *fp);
    return (int)hipGetLastError();
}
