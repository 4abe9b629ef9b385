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
#include "kernels.h"
#include "pixel_math.hpp"
#include <string.h>

using namespace cvs;

namespace {

constexpr int kWG = 1024;                  // 16 waves: the LDS table allows one workgroup per CU
constexpr int kLutHalfs = 65536;

struct Mat { float m[9]; };

// Pointers that arrive inside a job record are generic to the compiler, which then emits flat_load
// (counts on BOTH vmcnt and lgkmcnt and so serialises against the LDS gathers).  Tell it they are global.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const u32x4 __attribute__((address_space(1))) *g_cu4;
typedef u32x4 __attribute__((address_space(1))) *g_u4;
typedef const u32x2 __attribute__((address_space(1))) *g_cu2;
typedef u32x2 __attribute__((address_space(1))) *g_u2;
__device__ __forceinline__ uint4 ld4(const void *p, size_t i) { u32x4 v = ((g_cu4)p)[i]; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st4(void *p, size_t i, uint4 v) { u32x4 t = { v.x, v.y, v.z, v.w }; ((g_u4)p)[i] = t; }
__device__ __forceinline__ uint2 ld2(const void *p, size_t i) { u32x2 v = ((g_cu2)p)[i]; return make_uint2(v.x, v.y); }
__device__ __forceinline__ void st2(void *p, size_t i, uint2 v) { u32x2 t = { v.x, v.y }; ((g_u2)p)[i] = t; }

// 128 KiB table -> LDS: 8 sweeps of 1024 lanes x 16 B
__device__ __forceinline__ void stage_lut(uint16_t *lds, const uint16_t *__restrict__ table) {
    const uint4 *src = reinterpret_cast<const uint4 *>(table);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
#pragma unroll
    for (int i = 0; i < kLutHalfs * 2 / 16 / kWG; i++) dst[i * kWG + threadIdx.x] = src[i * kWG + threadIdx.x];
    __syncthreads();
}

template <bool IN_LDS>
__device__ __forceinline__ uint32_t gather2(const uint16_t *lut, uint32_t pair) {
    // two halfs packed in one dword -> two gathers -> repack
    uint32_t lo = lut[pair & 0xFFFFu], hi = lut[pair >> 16];
    return lo | (hi << 16);
}

// one layer pixel through the color.c structure; returns the f32 value the stack then sees
template <bool PRE, bool POST>
__device__ __forceinline__ px32 grade(uint2 p, const Mat &mat, const uint16_t *lds_lut, const uint16_t *glb_post) {
    if (PRE) { p.x = gather2<true>(lds_lut, p.x); p.y = gather2<true>(lds_lut, p.y); }
    uint2 h = narrow(mat3(widen(p), mat.m));
    if (POST) {
        // the LDS slot belongs to the pre table when there is one
        if (PRE) { h.x = gather2<false>(glb_post, h.x); h.y = gather2<false>(glb_post, h.y); }
        else     { h.x = gather2<true>(lds_lut, h.x);   h.y = gather2<true>(lds_lut, h.y); }
    }
    return widen(h);
}

template <bool PRE, bool POST>
__device__ __forceinline__ uint2 grade_h(uint2 p, const Mat &mat, const uint16_t *lds_lut, const uint16_t *glb_post) {
    if (PRE) { p.x = gather2<true>(lds_lut, p.x); p.y = gather2<true>(lds_lut, p.y); }
    uint2 h = narrow(mat3(widen(p), mat.m));
    if (POST) {
        if (PRE) { h.x = gather2<false>(glb_post, h.x); h.y = gather2<false>(glb_post, h.y); }
        else     { h.x = gather2<true>(lds_lut, h.x);   h.y = gather2<true>(lds_lut, h.y); }
    }
    return h;
}

// ---------------------------------------------------------------- half_lookup on a flat array (half.c:82-85)

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

// ---------------------------------------------------------------- colour matrix in place on a window

template <bool PRE, bool POST>
__global__ __launch_bounds__(kWG) void k_color(cvk_view f, cvk_rect r, Mat mat, const uint16_t *__restrict__ pre,
                                               const uint16_t *__restrict__ post) {
    __shared__ uint16_t lut[(PRE || POST) ? kLutHalfs : 1];
    if (PRE) stage_lut(lut, pre);
    else if (POST) stage_lut(lut, post);
    const int w = r.x1 - r.x0 + 1, h = r.y1 - r.y0 + 1;
    const size_t n = (size_t)w * (size_t)h, stride = (size_t)gridDim.x * kWG;
    uint2 *base = reinterpret_cast<uint2 *>(f.data);
    for (size_t i = (size_t)blockIdx.x * kWG + threadIdx.x; i < n; i += stride) {
        int row = (int)(i / (size_t)w), col = (int)(i - (size_t)row * (size_t)w);
        uint2 *p = base + (size_t)(r.y0 + row - f.fy0) * (size_t)f.pitch + (size_t)(r.x0 + col - f.fx0);
        *p = grade_h<PRE, POST>(*p, mat, lut, post);
    }
}

// ---------------------------------------------------------------- the fused chain
//
// Persistent grid (<= 1 workgroup per CU).  A lane owns pixel PAIRS: one 16-byte load per layer,
// one 16-byte store, so a wave moves 1 KiB per instruction.  Two pairs are kept in flight per lane
// (2 x nlayers independent dwordx4 loads) to cover HBM latency with only 16 waves per CU.
// NL > 0: every job has exactly NL layers (fully unrolled); NL == 0: per-job layer count.

template <int MAXL, bool PRE, bool POST>
__device__ __forceinline__ uint2 chain_pixel(const uint2 (&px)[MAXL], int nl, const Mat &mat, const uint16_t *lut, const uint16_t *post) {
    px32 acc = grade<PRE, POST>(px[0], mat, lut, post);
#pragma unroll
    for (int k = 1; k < MAXL; k++)      // static indices only: a runtime-indexed array would live in scratch
        if (k < nl) acc = blend_over(acc, grade<PRE, POST>(px[k], mat, lut, post), 1.0f);
    return narrow(acc);
}

// Job records travel in the kernel-argument segment (copied at launch, read with scalar loads):
// no device-side record buffer to recycle, no H2D copy on the launch path.
constexpr int kJobsPerLaunch = 32;
struct Batch { cvk_chain_job jobs[kJobsPerLaunch]; };

template <int NL, bool PRE, bool POST>
__global__ __launch_bounds__(kWG) void k_chain(Batch batch, int njobs, Mat mat,
                                               const uint16_t *__restrict__ pre, const uint16_t *__restrict__ post) {
    __shared__ uint16_t lut[(PRE || POST) ? kLutHalfs : 1];
    if (PRE) stage_lut(lut, pre);
    else if (POST) stage_lut(lut, post);

    constexpr int MAXL = NL > 0 ? NL : CVK_CHAIN_MAX_LAYERS;
    const size_t stride = (size_t)gridDim.x * kWG;

    for (int j = 0; j < njobs; j++) {
        const cvk_chain_job &job = batch.jobs[j];
        const int nl = NL > 0 ? NL : job.nlayers;
        const size_t npairs = job.npixels / 2;

        size_t i = (size_t)blockIdx.x * kWG + threadIdx.x;
        // two pairs per trip
        for (; i + stride < npairs; i += 2 * stride) {
            uint4 a[MAXL], b[MAXL];
#pragma unroll
            for (int k = 0; k < MAXL; k++) if (k < nl) a[k] = ld4(job.layer[k], i);
#pragma unroll
            for (int k = 0; k < MAXL; k++) if (k < nl) b[k] = ld4(job.layer[k], i + stride);
            uint2 p0[MAXL], p1[MAXL], q0[MAXL], q1[MAXL];
#pragma unroll
            for (int k = 0; k < MAXL; k++) {
                p0[k] = make_uint2(a[k].x, a[k].y); p1[k] = make_uint2(a[k].z, a[k].w);
                q0[k] = make_uint2(b[k].x, b[k].y); q1[k] = make_uint2(b[k].z, b[k].w);
            }
            uint2 r0 = chain_pixel<MAXL, PRE, POST>(p0, nl, mat, lut, post), r1 = chain_pixel<MAXL, PRE, POST>(p1, nl, mat, lut, post);
            uint2 s0 = chain_pixel<MAXL, PRE, POST>(q0, nl, mat, lut, post), s1 = chain_pixel<MAXL, PRE, POST>(q1, nl, mat, lut, post);
            st4(job.out, i, make_uint4(r0.x, r0.y, r1.x, r1.y));
            st4(job.out, i + stride, make_uint4(s0.x, s0.y, s1.x, s1.y));
        }
        if (i < npairs) {
            uint4 a[MAXL];
#pragma unroll
            for (int k = 0; k < MAXL; k++) if (k < nl) a[k] = ld4(job.layer[k], i);
            uint2 p0[MAXL], p1[MAXL];
#pragma unroll
            for (int k = 0; k < MAXL; k++) { p0[k] = make_uint2(a[k].x, a[k].y); p1[k] = make_uint2(a[k].z, a[k].w); }
            uint2 r0 = chain_pixel<MAXL, PRE, POST>(p0, nl, mat, lut, post), r1 = chain_pixel<MAXL, PRE, POST>(p1, nl, mat, lut, post);
            st4(job.out, i, make_uint4(r0.x, r0.y, r1.x, r1.y));
        }
        // odd pixel count: the last pixel on its own
        if ((job.npixels & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            uint2 p[MAXL];
#pragma unroll
            for (int k = 0; k < MAXL; k++) if (k < nl) p[k] = ld2(job.layer[k], job.npixels - 1);
            st2(job.out, job.npixels - 1, chain_pixel<MAXL, PRE, POST>(p, nl, mat, lut, post));
        }
    }
}

inline unsigned persistent_grid(int cus, size_t work_items) {
    size_t need = (work_items + kWG - 1) / kWG;
    size_t g = cus > 0 ? (size_t)cus : 256;
    if (need < g) g = need ? need : 1;
    return (unsigned)g;
}

inline Mat make_mat(const float m[9]) {
    Mat r;
    for (int i = 0; i < 9; i++) r.m[i] = m[i];
    return r;
}

template <int NL>
int launch_chain(const Batch &jobs, int njobs, const Mat &mat, const uint16_t *pre, const uint16_t *post,
                 unsigned grid, hipStream_t s) {
    if (pre && post)  hipLaunchKernelGGL((k_chain<NL, true, true>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else if (pre)     hipLaunchKernelGGL((k_chain<NL, true, false>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else if (post)    hipLaunchKernelGGL((k_chain<NL, false, true>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else              hipLaunchKernelGGL((k_chain<NL, false, false>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int cvk_half_lookup(const uint16_t *table, uint16_t *out, const uint16_t *in, size_t count, int cus, void *stream) {
    if (!count) return 0;
    int vec_ok = ((((uintptr_t)out | (uintptr_t)in) & 15u) == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_lookup, dim3(persistent_grid(cus, count / 8 + 1)), dim3(kWG), 0, (hipStream_t)stream, table, out, in, count, vec_ok);
    return (int)hipGetLastError();
}

extern "C" int cvk_color_matrix(cvk_view frame, cvk_rect r, const float m[9], const uint16_t *pre, const uint16_t *post,
                                int cus, void *stream) {
    if (r.x1 < r.x0 || r.y1 < r.y0) return 0;
    size_t n = (size_t)(r.x1 - r.x0 + 1) * (size_t)(r.y1 - r.y0 + 1);
    dim3 grid(persistent_grid(cus, n)), block(kWG);
    Mat mat = make_mat(m);
    hipStream_t s = (hipStream_t)stream;
    if (pre && post)  hipLaunchKernelGGL((k_color<true, true>), grid, block, 0, s, frame, r, mat, pre, post);
    else if (pre)     hipLaunchKernelGGL((k_color<true, false>), grid, block, 0, s, frame, r, mat, pre, post);
    else if (post)    hipLaunchKernelGGL((k_color<false, true>), grid, block, 0, s, frame, r, mat, pre, post);
    else              hipLaunchKernelGGL((k_color<false, false>), grid, block, 0, s, frame, r, mat, pre, post);
    return (int)hipGetLastError();
}

extern "C" int cvk_chain_color_over(const cvk_chain_job *jobs, int njobs, int uniform_layers, const float m[9],
                                    const uint16_t *pre, const uint16_t *post, int cus, void *stream) {
    Mat mat = make_mat(m);
    unsigned grid = (unsigned)(cus > 0 ? cus : 256);
    hipStream_t s = (hipStream_t)stream;
    for (int first = 0; first < njobs; first += kJobsPerLaunch) {
        int n = njobs - first < kJobsPerLaunch ? njobs - first : kJobsPerLaunch;
        Batch b;
        memset(&b, 0, sizeof b);
        memcpy(b.jobs, jobs + first, sizeof(cvk_chain_job) * (size_t)n);
        int rc;
        switch (uniform_layers) {
        case 1: rc = launch_chain<1>(b, n, mat, pre, post, grid, s); break;
        case 2: rc = launch_chain<2>(b, n, mat, pre, post, grid, s); break;
        case 3: rc = launch_chain<3>(b, n, mat, pre, post, grid, s); break;
        case 4: rc = launch_chain<4>(b, n, mat, pre, post, grid, s); break;
        default: rc = launch_chain<0>(b, n, mat, pre, post, grid, s); break;
        }
        if (rc != 0) return rc;
    }
    return 0;
}
