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
#include <atomic>
#include "pixel_math.hpp"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float4 *at(const cvk_view &v, int x, int y) {
    return reinterpret_cast<float4 *>(v.data) + (size_t)(y - v.fy0) * (size_t)v.pitch + (size_t)(x - v.fx0);
}

// a source pixel as f32: f16 frames are widened on the way in (main.c:105-144), exactly
__device__ __forceinline__ float4 px_in(const cvk_view &v, bool half, int x, int y) {
    const size_t i = (size_t)(y - v.fy0) * (size_t)v.pitch + (size_t)(x - v.fx0);
    if (!half) return reinterpret_cast<const float4 *>(v.data)[i];
    const uint2 p = reinterpret_cast<const uint2 *>(v.data)[i];
    return make_float4(cvs::h2f(p.x & 0xFFFFu), cvs::h2f(p.x >> 16), cvs::h2f(p.y & 0xFFFFu), cvs::h2f(p.y >> 16));
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
        const float4 v = fp.axis == 0 ? px_in(fp.source, fp.in_half != 0, other, s) : px_in(fp.source, fp.in_half != 0, s, other);
        r = cvs::madd(v.x, w, r);        // t += s * coeff (video_scale.c:82-85)
        g = cvs::madd(v.y, w, g);
        b = cvs::madd(v.z, w, b);
        a = cvs::madd(v.w, w, a);
    }
    const int ox = fp.axis == 0 ? other : line, oy = fp.axis == 0 ? line : other;
    if (fp.out_half) {       // the consumer pulls f16: truncate here (main.c:43-71) instead of in a pass of its own
        const size_t o = (size_t)(oy - fp.target.fy0) * (size_t)fp.target.pitch + (size_t)(ox - fp.target.fx0);
        reinterpret_cast<uint2 *>(fp.target.data)[o] = make_uint2(cvs::f2h_rz2(r, g), cvs::f2h_rz2(b, a));
    } else {
        *at(fp.target, ox, oy) = make_float4(r, g, b, a);
    }
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

// ---------------------------------------------------------------- both passes in one launch, tiles in LDS
//
// A 256-lane workgroup owns a 32 x 16 tile of TARGET pixels.
//   1. the tile's tap tables (both axes) and the source pixels under its footprint go to LDS
//      (source tile: one float4 per pixel; f16 sources are widened on the way in);
//   2. horizontal pass: LDS -> LDS, one lane per (footprint row, target column);
//   3. vertical pass: LDS -> registers -> one coalesced store per pixel (f16 targets are truncated here).
// Source pixels are read from HBM once per tile (halo overlap between neighbouring tiles is served by
// L2), the horizontal result never leaves the CU.  Arithmetic and order are those of k_fir: every output
// starts at 0.0f and adds s * c per tap, ascending, mul and add separately rounded -- so the result
// equals running the two passes through HBM, bit for bit.
// Bound: HBM.  Algorithmic bytes: source pixel size (read once) + target pixel size (written once).
namespace {

constexpr int kTX = CVK_FIR2D_TILE_X, kTY = CVK_FIR2D_TILE_Y, kRows = 8;   // 32 x 8 lanes

__device__ __forceinline__ float4 load_px(const cvk_view &v, bool half, int x, int y) {
    const size_t i = (size_t)(y - v.fy0) * (size_t)v.pitch + (size_t)(x - v.fx0);
    if (!half) return reinterpret_cast<const float4 *>(v.data)[i];
    const uint2 p = reinterpret_cast<const uint2 *>(v.data)[i];
    return make_float4(cvs::h2f(p.x & 0xFFFFu), cvs::h2f(p.x >> 16), cvs::h2f(p.y & 0xFFFFu), cvs::h2f(p.y >> 16));
}

template <int MAXT>
__global__ __launch_bounds__(kTX * kRows) void k_fir2d(cvk_fir2d_params fp) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int lx = threadIdx.x, ly = threadIdx.y, tid = ly * kTX + lx;
    const int c0 = fp.tx0 + (int)blockIdx.x * kTX, r0 = fp.ty0 + (int)blockIdx.y * kTY;
    const int ncols = min(kTX, fp.tx1 - c0 + 1), nrows = min(kTY, fp.ty1 - r0 + 1);
    const int sx0 = fp.h.foot[2 * blockIdx.x], sx1 = fp.h.foot[2 * blockIdx.x + 1];
    const int sy0 = fp.v.foot[2 * blockIdx.y], sy1 = fp.v.foot[2 * blockIdx.y + 1];
    const int sw = sx1 >= sx0 ? sx1 - sx0 + 1 : 0, sh = sy1 >= sy0 ? sy1 - sy0 + 1 : 0;

    // LDS carve-up (sizes from the launch-wide maxima so that every block uses the same offsets)
    float4 *S = reinterpret_cast<float4 *>(lds_raw);                       // [max_sh][max_sw]
    float4 *M = S + (size_t)fp.max_sh * fp.max_sw;                         // [max_sh][kTX]
    float *ht = reinterpret_cast<float *>(M + (size_t)fp.max_sh * kTX);    // [kTX][h.stride]
    float *vt = ht + kTX * fp.h.stride;                                    // [kTY][v.stride]
    int *hs = reinterpret_cast<int *>(vt + kTY * fp.v.stride);             // [kTX][h.stride], relative to sx0
    int *vs = hs + kTX * fp.h.stride;                                      // [kTY][v.stride], relative to sy0
    int *hn = vs + kTY * fp.v.stride;                                      // [kTX]
    int *vn = hn + kTX;                                                    // [kTY]

    for (int i = tid; i < kTX * fp.h.stride; i += kTX * kRows) {
        const int c = i / fp.h.stride, k = i - c * fp.h.stride;
        const bool live = c < ncols;
        const size_t g = (size_t)(c0 - fp.tx0 + c) * fp.h.stride + k;
        ht[i] = live ? fp.h.taps[g] : 0.0f;
        hs[i] = live ? fp.h.src[g] - sx0 : 0;
    }
    for (int i = tid; i < kTY * fp.v.stride; i += kTX * kRows) {
        const int r = i / fp.v.stride, k = i - r * fp.v.stride;
        const bool live = r < nrows;
        const size_t g = (size_t)(r0 - fp.ty0 + r) * fp.v.stride + k;
        vt[i] = live ? fp.v.taps[g] : 0.0f;
        vs[i] = live ? fp.v.src[g] - sy0 : 0;
    }
    if (tid < kTX) hn[tid] = tid < ncols ? fp.h.ntaps[c0 - fp.tx0 + tid] : 0;
    if (tid < kTY) vn[tid] = tid < nrows ? fp.v.ntaps[r0 - fp.ty0 + tid] : 0;

    for (int y = ly; y < sh; y += kRows)
        for (int x = lx; x < sw; x += kTX)
            S[y * fp.max_sw + x] = load_px(fp.source, fp.in_half != 0, sx0 + x, sy0 + y);
    __syncthreads();

    // horizontal: lane lx = target column, rows strided by 8.  A lane's taps do not change from row to row:
    // they are read from LDS once, into registers (static indices only, so no scratch).
    {
        const int n = hn[lx];
        int sidx[MAXT];
        float wt[MAXT];
#pragma unroll
        for (int k = 0; k < MAXT; k++) {
            const bool live = k < n;
            sidx[k] = live ? hs[lx * fp.h.stride + k] : 0;
            wt[k] = live ? ht[lx * fp.h.stride + k] : 0.0f;
        }
        for (int y = ly; y < sh; y += kRows) {
            // (r,g) and (b,a) as packed pairs: v_pk_mul_f32 + v_pk_add_f32, same roundings as four scalar mul/add
            cvs::f32x2 rg = { 0.0f, 0.0f }, ba = { 0.0f, 0.0f };
            const float4 *row = S + y * fp.max_sw;
            // all reads first (independent of the running sums: MAXT b128 reads in flight), then the adds in
            // tap order; a dead tap reads pixel 0 of the row (a valid address) and is not added
            float4 v[MAXT];
#pragma unroll
            for (int k = 0; k < MAXT; k++) v[k] = row[sidx[k]];
#pragma unroll
            for (int k = 0; k < MAXT; k++) {
                const cvs::f32x2 nrg = cvs::madd(cvs::f32x2{ v[k].x, v[k].y }, wt[k], rg);
                const cvs::f32x2 nba = cvs::madd(cvs::f32x2{ v[k].z, v[k].w }, wt[k], ba);
                const bool live = k < n;
                rg = live ? nrg : rg;
                ba = live ? nba : ba;
            }
            M[y * kTX + lx] = make_float4(rg.x, rg.y, ba.x, ba.y);
        }
    }
    __syncthreads();

    // vertical: lane lx = target column, two target rows per lane
    if (lx < ncols) {
        for (int rr = ly; rr < nrows; rr += kRows) {
            const int n = vn[rr];
            const float *w = vt + rr * fp.v.stride;
            const int *s = vs + rr * fp.v.stride;
            cvs::f32x2 rg = { 0.0f, 0.0f }, ba = { 0.0f, 0.0f };
            float4 v[MAXT];
#pragma unroll
            for (int k = 0; k < MAXT; k++) v[k] = M[(k < n ? s[k] : 0) * kTX + lx];
#pragma unroll
            for (int k = 0; k < MAXT; k++) {
                const float c = k < n ? w[k] : 0.0f;
                const cvs::f32x2 nrg = cvs::madd(cvs::f32x2{ v[k].x, v[k].y }, c, rg);
                const cvs::f32x2 nba = cvs::madd(cvs::f32x2{ v[k].z, v[k].w }, c, ba);
                const bool live = k < n;
                rg = live ? nrg : rg;
                ba = live ? nba : ba;
            }
            const float r = rg.x, g = rg.y, b = ba.x, a = ba.y;
            const int x = c0 + lx, y = r0 + rr;
            const size_t o = (size_t)(y - fp.target.fy0) * (size_t)fp.target.pitch + (size_t)(x - fp.target.fx0);
            if (fp.out_half) reinterpret_cast<uint2 *>(fp.target.data)[o] = make_uint2(cvs::f2h_rz2(r, g), cvs::f2h_rz2(b, a));
            else reinterpret_cast<float4 *>(fp.target.data)[o] = make_float4(r, g, b, a);
        }
    }
}

}  // namespace

extern "C" size_t cvk_fir2d_lds_bytes(const cvk_fir2d_params *fp) {
    size_t px = (size_t)fp->max_sh * (size_t)fp->max_sw + (size_t)fp->max_sh * kTX;
    size_t tab = (size_t)(kTX * fp->h.stride + kTY * fp->v.stride) * (sizeof(float) + sizeof(int)) + (kTX + kTY) * sizeof(int);
    return px * sizeof(float4) + tab;
}

extern "C" int cvk_fir2d(const cvk_fir2d_params *fp, void *stream) {
    if (fp->tx1 < fp->tx0 || fp->ty1 < fp->ty0) return 0;
    const size_t lds = cvk_fir2d_lds_bytes(fp);
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    static std::atomic<bool> raised{ false };       // (several threads may launch at once; setting the attribute twice is harmless)
    if (!raised.load(std::memory_order_acquire)) {      // allow more than the default 64 KiB of dynamic LDS
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir2d<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir2d<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir2d<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir2d<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised.store(true, std::memory_order_release);
    }
    dim3 grid((unsigned)((fp->tx1 - fp->tx0 + kTX) / kTX), (unsigned)((fp->ty1 - fp->ty0 + kTY) / kTY));
    const int most = fp->h.stride > fp->v.stride ? fp->h.stride : fp->v.stride;
    if (most > 64) return (int)hipErrorInvalidValue;          // the caller falls back to the two-pass path
    if (most <= 12)      hipLaunchKernelGGL(k_fir2d<12>, grid, dim3(kTX, kRows), lds, (hipStream_t)stream, *fp);
    else if (most <= 16) hipLaunchKernelGGL(k_fir2d<16>, grid, dim3(kTX, kRows), lds, (hipStream_t)stream, *fp);
    else if (most <= 32) hipLaunchKernelGGL(k_fir2d<32>, grid, dim3(kTX, kRows), lds, (hipStream_t)stream, *fp);
    else                 hipLaunchKernelGGL(k_fir2d<64>, grid, dim3(kTX, kRows), lds, (hipStream_t)stream, *fp);
    return (int)hipGetLastError();
}
