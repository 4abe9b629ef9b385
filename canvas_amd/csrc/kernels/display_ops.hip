// display_ops.hip -- the display / export edge: f16 RGBA -> 8-bit, on device, so a preview or a thumbnail
// costs a 4 B/px download instead of 8 B/px of halfs plus a CPU loop.
//
// Three consumers in the reference turn a pulled f16 frame into bytes, all through the half->u8 ramp of
// gammatab.c:13-38 (`table` here; for the widget it is composed with the linear->sRGB table first):
//   src/cprocess/widget_gl.c:291-307   sRGB table on all four halfs, then ramp -> rgba_u8 {r,g,b,a}
//   src/libav/writeVideo.c:328-340     ramp -> rgba_u8 {r,g,b,a}
//   src/process/RgbaFrameF16.c:114-149 ramp, then premultiplied ARGB32: a<<24 | (r*a>>8)<<16 | (g*a>>8)<<8 | (b*a>>8)
// Integer work throughout; the 64 KiB byte table sits in LDS (two workgroups fit a CU), four ds_read_u8 per
// pixel.  Output is packed over the rectangle: dst[(y - y0) * width + (x - x0)].
// Bound: HBM.  Algorithmic bytes: 8 read + 4 written per pixel.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace {

constexpr int kLanes = 512, kTable = 65536;
typedef uint32_t v4 __attribute__((ext_vector_type(4)));
typedef uint32_t v2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void stage(uint8_t *lds, const uint8_t *table) {
    const uint4 *t = reinterpret_cast<const uint4 *>(table);
    uint4 *d = reinterpret_cast<uint4 *>(lds);
    for (int i = threadIdx.x; i < kTable / 16; i += blockDim.x) d[i] = t[i];
    __syncthreads();
}

template <int MODE>
__device__ __forceinline__ uint32_t pack(const uint8_t *t, uint2 p) {
    const uint32_t r = t[p.x & 0xFFFFu], g = t[p.x >> 16], b = t[p.y & 0xFFFFu], a = t[p.y >> 16];
    if (MODE == CVK_DISPLAY_RGBA8) return r | (g << 8) | (b << 16) | (a << 24);
    return (a << 24) | ((((r * a) >> 8) & 0xFFu) << 16) | ((((g * a) >> 8) & 0xFFu) << 8) | (((b * a) >> 8) & 0xFFu);
}

// any rectangle of the frame
template <int MODE>
__global__ __launch_bounds__(kLanes) void k_display(uint32_t *__restrict__ dst, cvk_view src, cvk_rect r, const uint8_t *__restrict__ table) {
    __shared__ __attribute__((aligned(16))) uint8_t t[kTable];
    stage(t, table);
    const int w = r.x1 - r.x0 + 1, h = r.y1 - r.y0 + 1;
    const size_t n = (size_t)w * (size_t)h, stride = (size_t)gridDim.x * kLanes;
    for (size_t i = (size_t)blockIdx.x * kLanes + threadIdx.x; i < n; i += stride) {
        const int row = (int)(i / (size_t)w), col = (int)(i - (size_t)row * (size_t)w);
        const uint2 p = reinterpret_cast<const uint2 *>(src.data)[(size_t)(r.y0 + row - src.fy0) * (size_t)src.pitch + (size_t)(r.x0 + col - src.fx0)];
        dst[i] = pack<MODE>(t, p);
    }
}

// whole rows: source and destination are both one contiguous run; two pixels per lane
template <int MODE>
__global__ __launch_bounds__(kLanes) void k_display_flat(uint32_t *__restrict__ dst, const uint2 *__restrict__ src, size_t npixels, const uint8_t *__restrict__ table) {
    __shared__ __attribute__((aligned(16))) uint8_t t[kTable];
    const size_t npairs = npixels / 2, stride = (size_t)gridDim.x * kLanes;
    size_t i = (size_t)blockIdx.x * kLanes + threadIdx.x;
    // the first pair is requested before the table is staged, every later one a trip ahead of its use
    v4 cur = { 0u, 0u, 0u, 0u };
    if (i < npairs) cur = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(src) + i);
    stage(t, table);
    for (; i < npairs; i += stride) {
        v4 nxt = cur;
        if (i + stride < npairs) nxt = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(src) + i + stride);
        const v2 o = { pack<MODE>(t, make_uint2(cur.x, cur.y)), pack<MODE>(t, make_uint2(cur.z, cur.w)) };
        __builtin_nontemporal_store(o, reinterpret_cast<v2 *>(dst) + i);
        cur = nxt;
    }
    if ((npixels & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[npixels - 1] = pack<MODE>(t, src[npixels - 1]);
}

}  // namespace

extern "C" int cvk_display(void *dst, cvk_view src, cvk_rect r, const uint8_t *table, int mode, int cus, void *stream) {
    if (r.x1 < r.x0 || r.y1 < r.y0) return 0;
    if (mode != CVK_DISPLAY_RGBA8 && mode != CVK_DISPLAY_ARGB32_PREMUL) return (int)hipErrorInvalidValue;
    const size_t n = (size_t)(r.x1 - r.x0 + 1) * (size_t)(r.y1 - r.y0 + 1);
    hipStream_t s = (hipStream_t)stream;
    const size_t per_wg = (size_t)kLanes * 8;                 // enough work to be worth a 64 KiB table load
    size_t want = (n + per_wg - 1) / per_wg, most = (size_t)(cus > 0 ? cus : 256) * 2;
    dim3 grid((unsigned)(want < 1 ? 1 : want > most ? most : want)), block(kLanes);
    const bool rows = r.x0 == src.fx0 && r.x1 == src.fx1;
    const uint2 *first = reinterpret_cast<const uint2 *>(src.data) + (size_t)(r.y0 - src.fy0) * (size_t)src.pitch;
    if (rows && ((((uintptr_t)first) & 15u) == 0) && ((((uintptr_t)dst) & 7u) == 0)) {
        if (mode == CVK_DISPLAY_RGBA8) hipLaunchKernelGGL((k_display_flat<CVK_DISPLAY_RGBA8>), grid, block, 0, s, (uint32_t *)dst, first, n, table);
        else hipLaunchKernelGGL((k_display_flat<CVK_DISPLAY_ARGB32_PREMUL>), grid, block, 0, s, (uint32_t *)dst, first, n, table);
    } else {
        if (mode == CVK_DISPLAY_RGBA8) hipLaunchKernelGGL((k_display<CVK_DISPLAY_RGBA8>), grid, block, 0, s, (uint32_t *)dst, src, r, table);
        else hipLaunchKernelGGL((k_display<CVK_DISPLAY_ARGB32_PREMUL>), grid, block, 0, s, (uint32_t *)dst, src, r, table);
    }
    return (int)hipGetLastError();
}
