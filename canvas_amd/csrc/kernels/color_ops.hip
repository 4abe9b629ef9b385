// color_ops.hip -- transfer-LUT gathers, the 3x3 colour matrix, and the fused
// colour-matrix + alpha-over chain (BASELINE config 2, the north-star kernel).
//
// What the reference does per layer (src/cprocess/color.c:104-165) and per stack
// (src/cprocess/workspace.c:530-544, src/cprocess/main.c:43-71,115-139):
//   [LUT over all 4 halfs] -> widen -> 3x3 in f32 -> truncate to half -> [LUT] -> widen ->
//   over(acc, layer, 1.0) for every layer above the first -> truncate to half.
// Each arrow is a full pass over memory there (f32 intermediates, 4K: 132 MB each).  Here the whole
// chain stays in registers: 8 B read per layer pixel + 8 B written per output pixel.
//
// LUT placement: a transfer table is 65536 x u16 = 128 KiB.  CDNA4 has 160 KiB of LDS per CU, so
// one table is staged into LDS by a single 1024-lane workgroup per CU (the grid is persistent:
// <= one workgroup per CU, looping over pixels) and every gather is a ds_read_u16.  Serving 4-8
// random 2-byte gathers per pixel from L1/L2 instead would cap the kernel far below HBM rate.
// When a pre- AND a post-table are both requested the second one does not fit; it is gathered
// from global memory (L2-resident).  That combination is off the north-star path.
//
// Bound: HBM.  Algorithmic bytes: colour = 8 r + 8 w per pixel; lookup = 2 r + 2 w per half;
// chain = 8 * nlayers r + 8 w per output pixel (config 2: 24 B/px).
#include <cstdlib>
#include <atomic>
#include "lut_common.hpp"
#include "grade.hpp"
#include "chain_math.hpp"

using namespace cvs;

namespace {

// ---------------------------------------------------------------- half_lookup on a flat array (half.c:82-85)
// (no arithmetic: exists once, in the plain build of this file)
#ifndef CVS_CONTRACT
__global__ __launch_bounds__(kWG) void k_lookup(const uint16_t *__restrict__ table, uint16_t *__restrict__ out,
                                                const uint16_t *__restrict__ in, size_t count, int vec_ok) {
    __shared__ uint16_t lut[kLutHalfs];
    stage_lut(lut, table);
    size_t stride = (size_t)gridDim.x * kWG;
    size_t first = (size_t)blockIdx.x * kWG + threadIdx.x;
    if (vec_ok) {
        size_t nvec = count / 8;
        for (size_t i = first; i < nvec; i += stride) {
            uint4 v = reinterpret_cast<const uint4 *>(in)[i];
            v.x = gather2<true>(lut, v.x); v.y = gather2<true>(lut, v.y);
            v.z = gather2<true>(lut, v.z); v.w = gather2<true>(lut, v.w);
            reinterpret_cast<uint4 *>(out)[i] = v;
        }
        size_t tail = nvec * 8 + first;
        if (blockIdx.x == 0 && tail < count) out[tail] = lut[in[tail]];
    } else {
        for (size_t i = first; i < count; i += stride) out[i] = lut[in[i]];
    }
}

#endif
// ---------------------------------------------------------------- colour matrix on a window (dst may be src)

template <bool PRE, bool POST>
__global__ __launch_bounds__(kWG) void k_color(cvk_view dst, cvk_view src, cvk_rect r, Mat kmat, const uint16_t *__restrict__ pre,
                                               const uint16_t *__restrict__ post) {
    const MatR mat = CVS_MAT_REGS(kmat);
    __shared__ uint16_t lut[(PRE || POST) ? kLutHalfs : 1];
    if (PRE) stage_lut(lut, pre);
    else if (POST) stage_lut(lut, post);
    const int w = r.x1 - r.x0 + 1, h = r.y1 - r.y0 + 1;
    const size_t n = (size_t)w * (size_t)h, stride = (size_t)gridDim.x * kWG;
    for (size_t i = (size_t)blockIdx.x * kWG + threadIdx.x; i < n; i += stride) {
        int row = (int)(i / (size_t)w), col = (int)(i - (size_t)row * (size_t)w);
        const uint2 *p = reinterpret_cast<const uint2 *>(src.data) + (size_t)(r.y0 + row - src.fy0) * (size_t)src.pitch + (size_t)(r.x0 + col - src.fx0);
        uint2 *q = reinterpret_cast<uint2 *>(dst.data) + (size_t)(r.y0 + row - dst.fy0) * (size_t)dst.pitch + (size_t)(r.x0 + col - dst.fx0);
        *q = grade_h<PRE, POST>(*p, mat, lut, post);
    }
}

// whole rows, both buffers packed the same way: a flat stream of pixel pairs (16 B per lane, non-temporal), one
// 1024-lane workgroup per CU, every load a trip ahead of its use (measured on single 4K frames: 0.0345 ms without the
// prefetch at 512 lanes, 0.030 with it, 0.025 at 1024 lanes; CVS_COLOR_BLOCK overrides the lane count).
// 8 B read + 8 B written per pixel.
template <bool PRE, bool POST>
__global__ __launch_bounds__(kWG) void k_color_flat(uint16_t *__restrict__ dst, const uint16_t *__restrict__ src, size_t npixels, Mat kmat,
                                                    const uint16_t *__restrict__ pre, const uint16_t *__restrict__ post) {
    const MatR mat = CVS_MAT_REGS(kmat);
    __shared__ uint16_t lut[(PRE || POST) ? kLutHalfs : 1];
    const size_t npairs = npixels / 2, stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the first pair is requested before the table is staged, every later one a trip ahead of its use
    u32x4 cur = { 0u, 0u, 0u, 0u };
    if (i < npairs) cur = __builtin_nontemporal_load((g_cu4)src + i);
    if (PRE || POST) {
        const uint4 *t = reinterpret_cast<const uint4 *>(PRE ? pre : post);
        uint4 *d = reinterpret_cast<uint4 *>(lut);
        for (int k = threadIdx.x; k < kLutHalfs * 2 / 16; k += blockDim.x) d[k] = t[k];
        __syncthreads();
    }
    for (; i < npairs; i += stride) {
        u32x4 nxt = cur;
        if (i + stride < npairs) nxt = __builtin_nontemporal_load((g_cu4)src + i + stride);
        __builtin_nontemporal_store(color_pair_codes<PRE, POST>(cur, mat, lut, post), (g_u4)dst + i);
        cur = nxt;
    }
    if ((npixels & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const uint2 p = reinterpret_cast<const uint2 *>(src)[npixels - 1];
        reinterpret_cast<uint2 *>(dst)[npixels - 1] = grade_h<PRE, POST>(p, mat, lut, post);
    }
}

}  // namespace

#ifndef CVS_CONTRACT
extern "C" int cvk_half_lookup(const uint16_t *table, uint16_t *out, const uint16_t *in, size_t count, int cus, void *stream) {
    if (!count) return 0;
    int vec_ok = ((((uintptr_t)out | (uintptr_t)in) & 15u) == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_lookup, dim3(persistent_grid(cus, count / 8 + 1)), dim3(kWG), 0, (hipStream_t)stream, table, out, in, count, vec_ok);
    return (int)hipGetLastError();
}
#endif

extern "C" int cvk_color_matrix(cvk_view dst, cvk_view src, cvk_rect r, const float m[9], const uint16_t *pre, const uint16_t *post,
                                int cus, void *stream) {
    if (r.x1 < r.x0 || r.y1 < r.y0) return 0;
    size_t n = (size_t)(r.x1 - r.x0 + 1) * (size_t)(r.y1 - r.y0 + 1);
    Mat mat = make_mat(m);
    hipStream_t s = (hipStream_t)stream;
    // full rows of both buffers, same pitch: the rectangle is one contiguous run of pixels
    const bool rows = r.x0 == dst.fx0 && r.x1 == dst.fx1 && r.x0 == src.fx0 && r.x1 == src.fx1 && dst.pitch == src.pitch;
    if (rows) {
        uint16_t *d = reinterpret_cast<uint16_t *>(dst.data) + (size_t)(r.y0 - dst.fy0) * (size_t)dst.pitch * 4;
        const uint16_t *q = reinterpret_cast<const uint16_t *>(src.data) + (size_t)(r.y0 - src.fy0) * (size_t)src.pitch * 4;
        if (((((uintptr_t)d) | ((uintptr_t)q)) & 15u) == 0) {
            static std::atomic<int> env_cached{ -1 };       // (several threads may launch at once)
            int env_block = env_cached.load(std::memory_order_relaxed);
            if (env_block < 0) {
                const char *e = CVS_DIAG_ENV("CVS_COLOR_BLOCK");
                env_block = e ? atoi(e) : 0;
                if (env_block < 64 || env_block > kWG || (env_block & 63)) env_block = 0;
                env_cached.store(env_block, std::memory_order_relaxed);
            }
            dim3 grid((unsigned)(cus > 0 ? cus : 256)), block(env_block ? env_block : kWG);
            if (pre && post)  hipLaunchKernelGGL((k_color_flat<true, true>), grid, block, 0, s, d, q, n, mat, pre, post);
            else if (pre)     hipLaunchKernelGGL((k_color_flat<true, false>), grid, block, 0, s, d, q, n, mat, pre, post);
            else if (post)    hipLaunchKernelGGL((k_color_flat<false, true>), grid, block, 0, s, d, q, n, mat, pre, post);
            else              hipLaunchKernelGGL((k_color_flat<false, false>), grid, block, 0, s, d, q, n, mat, pre, post);
            return (int)hipGetLastError();
        }
    }
    dim3 grid(persistent_grid(cus, n)), block(kWG);
    if (pre && post)  hipLaunchKernelGGL((k_color<true, true>), grid, block, 0, s, dst, src, r, mat, pre, post);
    else if (pre)     hipLaunchKernelGGL((k_color<true, false>), grid, block, 0, s, dst, src, r, mat, pre, post);
    else if (post)    hipLaunchKernelGGL((k_color<false, true>), grid, block, 0, s, dst, src, r, mat, pre, post);
    else              hipLaunchKernelGGL((k_color<false, false>), grid, block, 0, s, dst, src, r, mat, pre, post);
    return (int)hipGetLastError();
}
