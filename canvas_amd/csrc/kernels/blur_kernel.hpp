// blur_kernel.hpp (the kernel of blur_ops.hip and blur_long_ops.hip) -- separable FIR blur with one fixed tap list, both passes in one sweep down the frame.
//
// The blur node (DESIGN.md "A11") is the video_scale.c FIR structure with the same taps for every line:
//   target(x, y) = sum_k taps[k] * H(x, y - c + k),   H(x, y) = sum_k taps[k] * src(x - c + k, y),   c = ntaps / 2,
// every sum started at 0.0f and taken in ascending k with separately rounded mul and add, taps that fall
// outside the source's current window skipped.  k_fir2d does this for arbitrary per-line tap tables with
// LDS gathers on both axes; here the taps are uniform, so the vertical window can live in REGISTERS:
//
//   a workgroup owns a strip of W-(NT-1) target columns and marches down a segment of rows;
//   per source row: one coalesced load per lane (f16 widened on the way in) -> LDS row buffer (double
//   buffered, one barrier per row) -> each lane forms H for its column from NT neighbouring LDS pixels ->
//   the H row is pushed into an NT-deep register ring -> the vertical sum of the ring is one output row,
//   stored coalesced (f16 targets truncated here).
//
// A skipped tap and a tap on a zero pixel give the same sum when the tap is finite (acc + 0*w == acc; the
// sign of a zero result is not pinned by the reference build, -fno-signed-zeros), so pixels outside the
// source window are fed in as zeros; the host sends non-finite tap lists to k_fir2d instead.
// Source pixels are read once per segment (+ NT-1 halo rows, + NT-1 halo columns per strip, both served
// by L2), LDS traffic is 1 write + NT reads per pixel, no intermediate frame exists.
// Algorithmic bytes: source pixel size + target pixel size per target pixel.  Bound: HBM up to about 5 taps; from 9 taps on
// the issue of 2 passes x 2 channel pairs x NT x (multiply + add) -- separately rounded, as the reference build rounds
// them, so no FMA -- takes longer than the memory traffic (4K, 9 taps: 0.045 ms against 0.024 ms of traffic).
#pragma once
#include <cstdlib>
#include <atomic>
#include <type_traits>
#include <utility>
#include "kernels.h"
#include "chain_math.hpp"

namespace {

using cvs::f32x2;

struct Px { f32x2 rg, ba; };

template <bool INH> struct Raw;
template <> struct Raw<true> { uint2 v; };
template <> struct Raw<false> { float4 v; };

template <bool INH>
__device__ __forceinline__ Raw<INH> fetch(const char *base, size_t row_bytes, int ys, int fy0, bool live) {
    Raw<INH> r;
    if constexpr (INH) r.v = make_uint2(0u, 0u); else r.v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (live) {
        const char *p = base + (size_t)(ys - fy0) * row_bytes;
        if constexpr (INH) r.v = *reinterpret_cast<const uint2 *>(p); else r.v = *reinterpret_cast<const float4 *>(p);
    }
    return r;
}

template <bool INH>
__device__ __forceinline__ float4 widen(const Raw<INH> &r) {
    if constexpr (INH) return make_float4(cvs::h2f(r.v.x & 0xFFFFu), cvs::h2f(r.v.x >> 16), cvs::h2f(r.v.y & 0xFFFFu), cvs::h2f(r.v.y >> 16));
    else return r.v;
}

template <class F, int... Js>
__device__ __forceinline__ void each_slot(F &f, std::integer_sequence<int, Js...>) {
    (void)(f(std::integral_constant<int, Js>{}) && ...);
}

// EPI: the blurred pixel is the bottom layer of a workspace stack -- bp.nover f16 frames (same geometry as the
// target) are blended over it with video_mix.c:323-337 at mix 1.0 before the truncating store, so the f32 blur
// result never leaves the registers (a workspace pulls its items as f32: no rounding between blur and over).
//
// STEP > 1: the same sweep as a decimating resampler -- target line t reads source lines STEP*t - c + k.  That
// is what the Lanczos gather degenerates to when the scale factor is 1/STEP with STEP a power of two: every
// line centre t / factor is an integer, every fractional offset is 0, every line gets the same taps
// (scale.c plan_lanczos).  A lane then loads STEP source columns per row, the LDS row is kept de-interleaved
// (one array per column phase, so tap reads stay contiguous across lanes), H is formed for target columns only,
// every source row is pushed into the ring, and an output row leaves every STEP-th step.
template <int NT, int W, bool INH, bool EPI, int STEP>
__global__ __launch_bounds__(W) void k_blur(cvk_blur_params bp) {
    constexpr int C = NT / 2, OUTW = (STEP * W - NT) / STEP + 1, PITCH = W + (NT <= 17 ? 16 : 32), CH = NT <= 15 ? NT : 8;
    static_assert(!(EPI && STEP != 1), "the over epilogue is for the 1:1 blur");
    __shared__ float4 rowbuf[2][STEP][PITCH];
    const int lane = threadIdx.x;
    const int xo = bp.tx0 + (int)blockIdx.x * OUTW;          // first target column of the strip
    const int sfirst = STEP * xo - C;                        // first source column of the strip
    const int tcol = xo + lane;                              // the target column this lane produces
    const bool out_live = lane < OUTW && tcol <= bp.tx1;
    const int ta = bp.ty0 + (int)blockIdx.y * bp.rows_per_wg;
    const int tb = min(ta + bp.rows_per_wg - 1, bp.ty1);
    const int ys0 = STEP * ta - C;                           // first source row the segment needs
    const int steps = STEP * (tb - ta) + NT;

    float w[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) w[k] = bp.taps[k];

    // a batch of frames: grid.z picks the frame (uniform)
    // (the pointer arrays are read through the kernel-argument segment itself: indexing the by-value struct with
    // blockIdx.z made hipcc copy all of it to scratch memory, 688 bytes per lane)
    const int z = (int)blockIdx.z;
    typedef const cvk_blur_params __attribute__((address_space(4))) *kargs_t;
    const kargs_t ka = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();          // `bp` is the kernel's first and only argument
    const void *src_data = bp.batch.n ? ka->batch.source[z] : bp.source.data;
    void *dst_data = bp.batch.n ? ka->batch.target[z] : bp.target.data;
    const void *over_data[CVK_BLUR_MAX_OVER];
#pragma unroll
    for (int l = 0; l < CVK_BLUR_MAX_OVER; l++) over_data[l] = bp.batch.n ? ka->batch.over[z][l] : bp.over[l];

    constexpr size_t SPX = INH ? 8 : 16;
    const size_t srow = (size_t)bp.source.pitch * SPX;
    // lane loads source columns sfirst + lane + r * W, r < STEP
    const char *sbase[STEP];
    bool col_live[STEP];
#pragma unroll
    for (int r = 0; r < STEP; r++) {
        const int scol = sfirst + lane + r * W;
        col_live[r] = scol >= bp.sx0 && scol <= bp.sx1;
        sbase[r] = reinterpret_cast<const char *>(src_data) + (ptrdiff_t)(scol - bp.source.fx0) * (ptrdiff_t)SPX;
    }
    const size_t tpx = bp.out_half ? 8 : 16;
    char *tbase = reinterpret_cast<char *>(dst_data) + (ptrdiff_t)(tcol - bp.target.fx0) * (ptrdiff_t)tpx;
    const size_t trow = (size_t)bp.target.pitch * tpx;

    if (lane < PITCH - W) {
#pragma unroll
        for (int r = 0; r < STEP; r++) { rowbuf[0][r][W + lane] = make_float4(0.f, 0.f, 0.f, 0.f); rowbuf[1][r][W + lane] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }

    Px ring[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) ring[k].rg = ring[k].ba = f32x2{ 0.0f, 0.0f };

    struct Row { Raw<INH> v[STEP]; };
    auto fetch_row = [&](int ys, bool wanted) {
        Row row;
        const bool live = wanted && ys >= bp.sy0 && ys <= bp.sy1;
#pragma unroll
        for (int r = 0; r < STEP; r++) row.v[r] = fetch<INH>(sbase[r], srow, ys, bp.source.fy0, live && col_live[r]);
        return row;
    };
    Row cur = fetch_row(ys0, true);
    Row nxt = fetch_row(ys0 + 1, steps > 1);
    uint2 ov_next[CVK_BLUR_MAX_OVER];
#pragma unroll
    for (int l = 0; l < CVK_BLUR_MAX_OVER; l++) ov_next[l] = make_uint2(0u, 0u);
    // (NT >= 3: the first emitting step is never step 0, so the first request always has a step to travel)

    for (int i0 = 0; i0 < steps; i0 += NT) {
        // NT steps with the ring slot as a compile-time constant (a runtime index would send the ring to scratch)
        auto step = [&](auto jc) -> bool {
            constexpr int j = decltype(jc)::value;
            const int i = i0 + j;
            if (i >= steps) return false;                     // uniform over the workgroup
            const int ys = ys0 + i;
            const bool emits = i >= NT - 1 && (STEP == 1 || (i - (NT - 1)) % STEP == 0);     // uniform
            const int t = ta + (i - (NT - 1)) / STEP;         // the target row this step completes
            // two rows ahead goes out now; this row's data was requested two steps ago
            const Row far = fetch_row(ys + 2, i + 2 < steps);
            // the upper layers of the NEXT step's output pixel go out now; this step's were requested a step ago
            uint2 ov[CVK_BLUR_MAX_OVER];
            if constexpr (EPI) {
                const bool next_emits = i + 1 >= NT - 1 && i + 1 < steps;
                const size_t o = (size_t)(t + 1 - bp.target.fy0) * (size_t)bp.target.pitch + (size_t)(tcol - bp.target.fx0);
#pragma unroll
                for (int l = 0; l < CVK_BLUR_MAX_OVER; l++) {
                    ov[l] = ov_next[l];
                    ov_next[l] = make_uint2(0u, 0u);
                    if (l < bp.nover && next_emits && out_live) ov_next[l] = reinterpret_cast<const uint2 *>(over_data[l])[o];
                }
            }
            float4 (*buf)[PITCH] = rowbuf[i & 1];
#pragma unroll
            for (int r = 0; r < STEP; r++) {
                const int q = lane + r * W;                   // offset from sfirst; phase q % STEP, slot q / STEP
                buf[q % STEP][q / STEP] = widen<INH>(cur.v[r]);
            }
            cur = nxt;
            nxt = far;
            __syncthreads();
            // all products (of a group of CH taps) first, then the two add chains interleaved: a packed add right behind the
            // packed multiply it depends on costs a hazard slot (s_nop) per tap; the rounding and the order of the
            // additions do not change.  Up to 15 taps are one group; longer lists go in groups of 8 to stay in registers.
            f32x2 rg = { 0.0f, 0.0f }, ba = { 0.0f, 0.0f };
#pragma unroll
            for (int k0 = 0; k0 < NT; k0 += CH) {
                float4 v[CH];
#pragma unroll
                for (int c = 0; c < CH; c++) if (k0 + c < NT) v[c] = buf[(k0 + c) % STEP][lane + (k0 + c) / STEP];
                if constexpr (cvs::kContract) {
                    // the clang build's t += s * c: one fused multiply-add per tap (the first: fma(s, c, 0) = the product)
#pragma unroll
                    for (int c = 0; c < CH; c++) if (k0 + c < NT) {
                        if (k0 + c == 0) { rg = f32x2{ v[c].x, v[c].y } * w[0]; ba = f32x2{ v[c].z, v[c].w } * w[0]; }
                        else { rg = cvs::madd(f32x2{ v[c].x, v[c].y }, w[k0 + c], rg); ba = cvs::madd(f32x2{ v[c].z, v[c].w }, w[k0 + c], ba); }
                    }
                } else {
                f32x2 prg[CH], pba[CH];
#pragma unroll
                for (int c = 0; c < CH; c++) if (k0 + c < NT) {
                    prg[c] = f32x2{ v[c].x, v[c].y } * w[k0 + c];
                    pba[c] = f32x2{ v[c].z, v[c].w } * w[k0 + c];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < CH; c++) if (k0 + c < NT) {
                    // (0 + p0 is p0 except for the sign of a zero, which the reference's -fno-signed-zeros build leaves open)
                    if (k0 + c == 0) { rg = prg[c]; ba = pba[c]; }
                    else { rg = rg + prg[c]; ba = ba + pba[c]; }
                }
                }
            }
            ring[j].rg = rg;
            ring[j].ba = ba;
            if (emits) {
                // ring[(j+1) % NT] is the oldest row = tap 0
                f32x2 org = { 0.0f, 0.0f }, oba = { 0.0f, 0.0f };
#pragma unroll
                for (int k0 = 0; k0 < NT; k0 += CH) {
                    if constexpr (cvs::kContract) {
#pragma unroll
                        for (int c = 0; c < CH; c++) if (k0 + c < NT) {
                            const Px &p = ring[(j + 1 + k0 + c) % NT];
                            if (k0 + c == 0) { org = p.rg * w[0]; oba = p.ba * w[0]; }
                            else { org = cvs::madd(p.rg, w[k0 + c], org); oba = cvs::madd(p.ba, w[k0 + c], oba); }
                        }
                    } else {
                    f32x2 qrg[CH], qba[CH];
#pragma unroll
                    for (int c = 0; c < CH; c++) if (k0 + c < NT) {
                        const Px &p = ring[(j + 1 + k0 + c) % NT];
                        qrg[c] = p.rg * w[k0 + c];
                        qba[c] = p.ba * w[k0 + c];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < CH; c++) if (k0 + c < NT) {
                        if (k0 + c == 0) { org = qrg[c]; oba = qba[c]; }
                        else { org = org + qrg[c]; oba = oba + qba[c]; }
                    }
                    }
                }
                if constexpr (EPI) {
                    cvs::px1 acc = { org, oba.x, oba.y };
#pragma unroll
                    for (int l = 0; l < CVK_BLUR_MAX_OVER; l++)
                        if (l < bp.nover) {
                            const cvs::px1 up = { f32x2{ cvs::h2f(ov[l].x & 0xFFFFu), cvs::h2f(ov[l].x >> 16) }, cvs::h2f(ov[l].y & 0xFFFFu), cvs::h2f(ov[l].y >> 16) };
                            acc = cvs::over_px(acc, up);
                        }
                    org = acc.rg;
                    oba = f32x2{ acc.b, acc.a };
                }
                if (out_live) {
                    char *o = tbase + (size_t)(t - bp.target.fy0) * trow;
                    if (bp.out_half) *reinterpret_cast<uint2 *>(o) = make_uint2(cvs::f2h_rz2(org.x, org.y), cvs::f2h_rz2(oba.x, oba.y));
                    else *reinterpret_cast<float4 *>(o) = make_float4(org.x, org.y, oba.x, oba.y);
                }
            }
            return true;
        };
        each_slot(step, std::make_integer_sequence<int, NT>{});
    }
}

// rows per workgroup: as many workgroups as the chip holds at once (occupancy of this instance x CUs), all in
// one wave of the grid -- a second, partly filled wave costs more than the halo rows a shorter segment adds
template <class K>
int resident_per_cu(K kernel, int block) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, 0) != hipSuccess || n < 1) n = 1;
    static std::atomic<int> cap_cached{ -1 };       // (several threads may launch at once: pull-queue workers)
    int cap = cap_cached.load(std::memory_order_relaxed);
    if (cap < 0) { const char *e = CVS_DIAG_ENV("CVS_BLUR_WGS_PER_CU"); cap = e ? atoi(e) : 0; cap_cached.store(cap, std::memory_order_relaxed); }
    return cap > 0 ? cap : n;
}

template <int NT, int W, int STEP>
int launch(cvk_blur_params bp, int cus, hipStream_t s) {
    constexpr int OUTW = (STEP * W - NT) / STEP + 1;
    const int cols = bp.tx1 - bp.tx0 + 1, rows = bp.ty1 - bp.ty0 + 1;
    const int strips = (cols + OUTW - 1) / OUTW;
    const bool epi = STEP == 1 && bp.nover > 0;
    static std::atomic<int> occ[3];                 // per instance: [epilogue, f16 in, f32 in]
    std::atomic<int> &cached = occ[epi ? 0 : bp.in_half ? 1 : 2];
    int mine = cached.load(std::memory_order_relaxed);
    if (!mine) {
        if constexpr (STEP == 1) { if (epi) mine = resident_per_cu(k_blur<NT, W, true, true, 1>, W); }
        if (!mine) mine = bp.in_half ? resident_per_cu(k_blur<NT, W, true, false, STEP>, W) : resident_per_cu(k_blur<NT, W, false, false, STEP>, W);
        cached.store(mine, std::memory_order_relaxed);
    }
    const int nframes = bp.batch.n > 0 ? bp.batch.n : 1;
    if (bp.rows_per_wg <= 0) {
        int segs = (mine * cus) / (strips * nframes);
        if (segs < 1) segs = 1;
        int r = (rows + segs - 1) / segs;
        const int lo = (NT - 1) / STEP;            // halo rows cost at most as much as the rows produced
        if (r < lo) r = lo;
        if (r > rows) r = rows;
        bp.rows_per_wg = r;
    }
    dim3 grid((unsigned)strips, (unsigned)((rows + bp.rows_per_wg - 1) / bp.rows_per_wg), (unsigned)nframes);
    if constexpr (STEP == 1) {
        if (epi) { hipLaunchKernelGGL((k_blur<NT, W, true, true, 1>), grid, dim3(W), 0, s, bp); return (int)hipGetLastError(); }     // f16 in, f16 out
    }
    if (bp.in_half) hipLaunchKernelGGL((k_blur<NT, W, true, false, STEP>), grid, dim3(W), 0, s, bp);
    else            hipLaunchKernelGGL((k_blur<NT, W, false, false, STEP>), grid, dim3(W), 0, s, bp);
    return (int)hipGetLastError();
}

}  // namespace
