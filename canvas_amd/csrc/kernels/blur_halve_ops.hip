// blur_halve_ops.hip -- BASELINE config 3 in ONE sweep: separable blur (one odd tap list, 1:1) followed by the Lanczos
// resampler at factor 1/2 on both axes, f16 (or f32) in, f16 (or f32) out, nothing in between ever in HBM.
//
// What it replaces: k_blur<NT1,...,1> writing a full-size f32 frame and k_blur<NT2,...,2> reading it back -- for a 4K
// frame 133 MB written + 185 MB read (1.39x: the decimating pass re-reads its halo rows) beside 66 MB of source and 17 MB
// of result.  Both sums are the video_scale.c gather structure: every sum starts at 0.0f and takes its
// taps in ascending order with separately rounded multiply and add; the order of the four passes is the reference order
// of the two nodes: blur x, blur y (the blur node, horizontal then vertical), resample x, resample y (x pass then y pass).
//
//   B(x, y)  = sum_j w1[j] * ( sum_i w1[i] * S(x - c1 + i, y - c1 + j) )            S = source, 0 outside its window
//   T(t, u)  = sum_j w2[j] * ( sum_i w2[i] * B(2t - c2 + i, 2u - c2 + j) )          B = 0 outside the SOURCE's window:
// the blur node's output window is its source's window, and the resampler skips taps outside ITS source's window
// (video_scale.c:106-107,211-212), so a blurred pixel outside the window must contribute nothing -- it is NOT the
// blur formula evaluated there.
//
// Sweep (one workgroup = a strip of W source columns x a segment of target rows), per source row:
//   load (one pixel per lane, two rows ahead) -> LDS row S -> barrier -> H1 from NT1 neighbours -> ring1 (registers) ->
//   V1 = blurred row, zeroed outside the window -> LDS row B (de-interleaved by column parity) -> [next step, after its
//   barrier] H2 on the target-column lanes from NT2 neighbours -> ring2 (registers) -> every second blurred row V2 =
//   one target row, stored.  ONE barrier per source row: row B of step i is read in step i + 1 behind that step's
//   barrier, both LDS rows are double-buffered.
// Second stage on ALL lanes: a strip has fewer than W / 2 target columns, so the lower half of the workgroup takes the
// (r, g) pair of each target pixel and the upper half the (b, a) pair -- every wave carries the same work (with whole
// pixels on the first OUTW lanes two of four waves did all of the resampler's arithmetic and the others waited at the barrier).
// Both rings have RL = 12 slots (NT1 <= 12, NT2 == 11) so that one unrolled body of 12 steps sees every slot as a
// compile-time register index and the emit parity (every second blurred row) is a compile-time fact as well.
//
// Cost model (4K -> 1080p, 9 + 11 taps): the arithmetic is the sum of the two kernels' (~105 packed multiply/add per
// source pixel at one pixel per lane) times the halo factor of a strip x segment decomposition, which is larger here
// because every segment re-blurs the resampler's 10 halo rows; the traffic falls from ~400 MB to ~100 MB.
#include <type_traits>
#include <atomic>
#include <utility>
#include "kernels.h"
#include "chain_math.hpp"

namespace {

using cvs::f32x2;
struct Px { f32x2 rg, ba; };

constexpr int RL = 12;                     // ring length of both rings; the step loop is unrolled RL times
constexpr int kTwoRowsDefault = 1;         // the product launches the two-rows-per-barrier form (3 % faster, same pixels: profiles/r03/config3_batches.txt)

template <class F, int... Js>
__device__ __forceinline__ void each_slot(F &f, std::integer_sequence<int, Js...>) {
    (void)(f(std::integral_constant<int, Js>{}) && ...);
}

template <bool INH>
__device__ __forceinline__ float4 fetch_px(const char *p, bool live) {
    if (!live) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if constexpr (INH) {
        const uint2 v = *reinterpret_cast<const uint2 *>(p);
        return make_float4(cvs::h2f(v.x & 0xFFFFu), cvs::h2f(v.x >> 16), cvs::h2f(v.y & 0xFFFFu), cvs::h2f(v.y >> 16));
    } else {
        return *reinterpret_cast<const float4 *>(p);
    }
}

// sum of NT taps over values fetched by `at(k)`, ascending k.  Products are formed in groups of CH before their adds (a
// packed add right behind the packed multiply it depends on costs a hazard slot); the groups keep the products of a long
// list from occupying 4 x NT registers at once.  Same rounding and order as mul/add in sequence.
constexpr int CH = 6;
// (The first addition of a sum, 0 + p0, is p0 itself except for the sign of a zero product -- the reference is built
// -fno-signed-zeros and every comparison and fixture folds that sign -- so a sum starts with its first product: one packed
// add per channel pair and pass less, four passes per pixel.)
template <int NT, class At>
__device__ __forceinline__ Px fir(const float (&w)[NT], At at) {
    Px o = { f32x2{ 0.0f, 0.0f }, f32x2{ 0.0f, 0.0f } };
    if constexpr (cvs::kContract) {
        // the clang build's t += s * c: one fused multiply-add per tap, the first (fma(s, c, 0)) the plain product
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const Px v = at(k);
            if (k == 0) { o.rg = v.rg * w[0]; o.ba = v.ba * w[0]; }
            else { o.rg = cvs::madd(v.rg, w[k], o.rg); o.ba = cvs::madd(v.ba, w[k], o.ba); }
        }
        return o;
    }
#pragma unroll
    for (int k0 = 0; k0 < NT; k0 += CH) {
        f32x2 prg[CH], pba[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) if (k0 + c < NT) {
            const Px v = at(k0 + c);
            prg[c] = v.rg * w[k0 + c];
            pba[c] = v.ba * w[k0 + c];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CH; c++) if (k0 + c < NT) {
            if (k0 + c == 0) { o.rg = prg[c]; o.ba = pba[c]; }
            else { o.rg = o.rg + prg[c]; o.ba = o.ba + pba[c]; }
        }
    }
    return o;
}

// the same for one channel pair
template <int NT, class At>
__device__ __forceinline__ f32x2 fir1(const float (&w)[NT], At at) {
    f32x2 o = { 0.0f, 0.0f };
    if constexpr (cvs::kContract) {
#pragma unroll
        for (int k = 0; k < NT; k++) o = (k == 0) ? at(0) * w[0] : cvs::madd(at(k), w[k], o);
        return o;
    }
#pragma unroll
    for (int k0 = 0; k0 < NT; k0 += 2 * CH) {
        f32x2 p[2 * CH];
#pragma unroll
        for (int c = 0; c < 2 * CH; c++) if (k0 + c < NT) p[c] = at(k0 + c) * w[k0 + c];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 2 * CH; c++) if (k0 + c < NT) o = (k0 + c == 0) ? p[c] : o + p[c];
    }
    return o;
}

#ifdef CVS_DIAG       /* the one-row-per-barrier form: superseded by k_blur_halve2 below, kept for A/B runs of the diagnostic build only */
template <int NT1, int NT2, int W, bool INH>
__global__ __launch_bounds__(W) void k_blur_halve(cvk_blur_halve_params bp) {
    static_assert(NT1 % 2 == 1 && NT2 % 2 == 1 && NT1 <= RL && NT2 < RL && RL % 2 == 0, "ring layout");
    constexpr int C1 = NT1 / 2, C2 = NT2 / 2;
    constexpr int OUTW = (W - NT1 - NT2 + 1) / 2 + 1;          // target columns per strip
    constexpr int PITCH = W + 16, HALF = W / 2 + 16;
    __shared__ float4 rowS[2][PITCH];                          // source row, widened
    __shared__ float4 rowB[2][2][HALF];                        // blurred row, [buffer][column parity][column / 2]
    const int lane = threadIdx.x;
    const int xo = bp.tx0 + (int)blockIdx.x * OUTW;            // first target column of the strip
    const int sfirst = 2 * xo - C2 - C1;                       // source column of lane 0
    const int bcol = sfirst + C1 + lane;                       // the blurred column this lane produces (valid for lane <= W - NT1)
    static_assert(OUTW <= W / 2, "the two halves of the workgroup share the strip's target columns");
    const int tl = lane & (W / 2 - 1);                         // target column index within the strip
    const int pair = lane / (W / 2);                           // 0: this lane carries (r, g) of its target pixel, 1: (b, a); wave-uniform
    const int tcol = xo + tl;                                  // the target column this lane produces (tl < OUTW)
    const bool out_live = tl < OUTW && tcol <= bp.tx1;
    const bool bcol_live = bcol >= bp.sx0 && bcol <= bp.sx1;
    const int ta = bp.ty0 + (int)blockIdx.y * bp.rows_per_wg;
    const int tb = min(ta + bp.rows_per_wg - 1, bp.ty1);
    const int ys0 = 2 * ta - C2 - C1;                          // first source row the segment needs
    const int steps = 2 * (tb - ta) + NT2 + NT1 - 1;

    float w1[NT1], w2[NT2];
#pragma unroll
    for (int k = 0; k < NT1; k++) w1[k] = bp.taps1[k];
#pragma unroll
    for (int k = 0; k < NT2; k++) w2[k] = bp.taps2[k];

    constexpr size_t SPX = INH ? 8 : 16;
    const size_t srow = (size_t)bp.source.pitch * SPX;
    const int scol = sfirst + lane;
    const bool scol_live = scol >= bp.sx0 && scol <= bp.sx1;
    // a batch of frames: grid.z picks the frame (uniform)
    // (read through the kernel-argument segment: indexing the by-value struct with blockIdx.z sends it to scratch memory)
    typedef const cvk_blur_halve_params __attribute__((address_space(4))) *kargs_t;
    const kargs_t ka = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();          // `bp` is the kernel's first and only argument
    const void *src_data = bp.batch.n ? ka->batch.source[blockIdx.z] : bp.source.data;
    void *dst_data = bp.batch.n ? ka->batch.target[blockIdx.z] : bp.target.data;
    const char *sbase = reinterpret_cast<const char *>(src_data) + (ptrdiff_t)(scol - bp.source.fx0) * (ptrdiff_t)SPX;
    const size_t tpx = bp.out_half ? 8 : 16;
    char *tbase = reinterpret_cast<char *>(dst_data) + (ptrdiff_t)(tcol - bp.target.fx0) * (ptrdiff_t)tpx + (size_t)pair * (tpx / 2);
    const size_t trow = (size_t)bp.target.pitch * tpx;

    if (lane < PITCH - W) { rowS[0][W + lane] = make_float4(0.f, 0.f, 0.f, 0.f); rowS[1][W + lane] = make_float4(0.f, 0.f, 0.f, 0.f); }
    if (lane < 2 * (HALF - W / 2)) {
        const int par = lane & 1, slot = W / 2 + (lane >> 1);
        rowB[0][par][slot] = make_float4(0.f, 0.f, 0.f, 0.f); rowB[1][par][slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    Px ring1[RL];
    f32x2 ring2[RL];
#pragma unroll
    for (int k = 0; k < RL; k++) { ring1[k].rg = ring1[k].ba = ring2[k] = f32x2{ 0.0f, 0.0f }; }

    auto fetch_row = [&](int ys, bool wanted) -> float4 {
        const bool live = wanted && scol_live && ys >= bp.sy0 && ys <= bp.sy1;
        return fetch_px<INH>(sbase + (ptrdiff_t)(ys - bp.source.fy0) * (ptrdiff_t)srow, live);
    };
    // three rows in flight, slot = row % 3 (RL is a multiple of 3, so the slot is a compile-time fact of the unrolled body)
    static_assert(RL % 3 == 0, "prefetch slots rotate with the unrolled step index");
    float4 pre[3];
    pre[0] = fetch_row(ys0, true);
    pre[1] = fetch_row(ys0 + 1, steps > 1);
    // column part of "is this blurred pixel inside the blurred frame's window", as an all-ones / all-zeros word
    const uint32_t col_mask = bcol_live ? 0xFFFFFFFFu : 0u;

    // step i: source row ys0 + i enters; blurred row rb = i - (NT1 - 1) leaves the first ring (rb >= 0) and is handed over
    // through LDS; in step i + 1 it enters the second ring; blurred row rb completes target row ta + (rb - (NT2 - 1)) / 2
    // when that is a whole number >= 0.  The loop runs one step past the last source row to drain the hand-over.
    for (int i0 = 0; i0 <= steps; i0 += RL) {
        auto step = [&](auto jc) -> bool {
            constexpr int j = decltype(jc)::value;             // == i % RL
            const int i = i0 + j;
            if (i > steps) return false;                       // uniform over the workgroup
            pre[(j + 2) % 3] = fetch_row(ys0 + i + 2, i + 2 < steps);
            if (i < steps) rowS[i & 1][lane] = pre[j % 3];
            __syncthreads();                                   // row S of this step and row B of the previous one are visible

            // ---- second stage first (it consumes what the previous step produced): blurred row rb2 = i - NT1
            constexpr int s2 = (j + RL - (NT1 % RL)) % RL;     // ring2 slot of rb2 = i - NT1 (== rb2 % RL when i0 % RL == 0)
            const int rb2 = i - NT1;
            if (rb2 >= 0) {
                // this lane's channel pair of the blurred row: 8 of the 16 bytes of each float4
                const float2 *bb0 = reinterpret_cast<const float2 *>(&rowB[(i + 1) & 1][0][0]) + pair;       // written in step i - 1
                const float2 *bb1 = reinterpret_cast<const float2 *>(&rowB[(i + 1) & 1][1][0]) + pair;
                const int jl = tl < OUTW ? tl : 0;             // lanes past the strip's last target column stay inside the row
                ring2[s2] = fir1<NT2>(w2, [&](int k) { const float2 v = ((k & 1) ? bb1 : bb0)[2 * (jl + (k >> 1))]; return f32x2{ v.x, v.y }; });
                // rb2 - (NT2 - 1) even <=> rb2 even (NT2 odd) <=> i - NT1 even <=> j odd (NT1 odd, RL even)
                if constexpr (((j - NT1) & 1) == 0) {
                    if (rb2 >= NT2 - 1) {
                        const int t = ta + (rb2 - (NT2 - 1)) / 2;
                        const f32x2 o = fir1<NT2>(w2, [&](int k) { return ring2[(s2 + RL - (NT2 - 1) + k) % RL]; });
                        if (out_live && t <= tb) {
                            char *dst = tbase + (size_t)(t - bp.target.fy0) * trow;
                            if (bp.out_half) *reinterpret_cast<uint32_t *>(dst) = cvs::f2h_rz2(o.x, o.y);
                            else *reinterpret_cast<float2 *>(dst) = make_float2(o.x, o.y);
                        }
                    }
                }
            }

            // ---- first stage: H1 of this source row, then the blurred row it completes
            if (i < steps) {
                const float4 *sb = rowS[i & 1];
                ring1[j] = fir<NT1>(w1, [&](int k) { const float4 v = sb[lane + k]; return Px{ f32x2{ v.x, v.y }, f32x2{ v.z, v.w } }; });
                if (i >= NT1 - 1) {
                    Px b = fir<NT1>(w1, [&](int k) { return ring1[(j + RL - (NT1 - 1) + k) % RL]; });
                    const int by = ys0 + i - C1;               // the blurred row just completed
                    // outside the blurred frame's window: a skipped tap, i.e. zero (AND with a mask: four selects cost more)
                    const uint32_t m = (by >= bp.sy0 && by <= bp.sy1) ? col_mask : 0u;
                    rowB[i & 1][lane & 1][lane >> 1] = make_float4(__uint_as_float(__float_as_uint(b.rg.x) & m), __uint_as_float(__float_as_uint(b.rg.y) & m),
                                                                   __uint_as_float(__float_as_uint(b.ba.x) & m), __uint_as_float(__float_as_uint(b.ba.y) & m));
                }
            }
            return true;
        };
        each_slot(step, std::make_integer_sequence<int, RL>{});
    }
}
#endif

// The same sweep with TWO source rows per barrier.  A row step is a latency chain (row -> LDS -> barrier -> H1 -> ring ->
// V1 -> LDS -> barrier -> H2 -> ring -> V2) that three waves per SIMD do not hide, and a good part of it is the barrier
// itself: four waves on four SIMDs meet once per row (one-wave workgroups, with no barrier to wait at, are 22 % more
// efficient per lane and step -- and lose it to their horizontal halo, profiles/r03).  Here a step takes rows i and i + 1:
// both go to LDS, ONE barrier, stage 2 consumes the two blurred rows of the previous step (exactly one of them completes
// a target row: a compile-time fact, NT1 and NT2 are odd), stage 1 filters both rows.  Same sums, same order; the second
// stage runs one row later than in the one-row form, so the loop drains one step more.
template <int NT1, int NT2, int W, bool INH>
__global__ __launch_bounds__(W) void k_blur_halve2(cvk_blur_halve_params bp) {
    static_assert(NT1 % 2 == 1 && NT2 % 2 == 1 && NT1 <= RL && NT2 < RL && RL % 4 == 0, "ring layout");
    constexpr int C1 = NT1 / 2, C2 = NT2 / 2;
    constexpr int OUTW = (W - NT1 - NT2 + 1) / 2 + 1;          // target columns per strip
    constexpr int PITCH = W + 16, HALF = W / 2 + 16;
    __shared__ float4 rowS[2][2][PITCH];                       // [step parity][row of the pair]: source rows, widened
    __shared__ float4 rowB[2][2][2][HALF];                     // [step parity][row of the pair][column parity][column / 2]: blurred rows
    const int lane = threadIdx.x;
    const int xo = bp.tx0 + (int)blockIdx.x * OUTW;
    const int sfirst = 2 * xo - C2 - C1;
    const int bcol = sfirst + C1 + lane;
    static_assert(OUTW <= W / 2, "the two halves of the workgroup share the strip's target columns");
    const int tl = lane & (W / 2 - 1);
    const int pair = lane / (W / 2);
    const int tcol = xo + tl;
    const bool out_live = tl < OUTW && tcol <= bp.tx1;
    const bool bcol_live = bcol >= bp.sx0 && bcol <= bp.sx1;
    const int ta = bp.ty0 + (int)blockIdx.y * bp.rows_per_wg;
    const int tb = min(ta + bp.rows_per_wg - 1, bp.ty1);
    const int ys0 = 2 * ta - C2 - C1;
    const int steps = 2 * (tb - ta) + NT2 + NT1 - 1;           // source rows 0 .. steps - 1 of the segment

    float w1[NT1], w2[NT2];
#pragma unroll
    for (int k = 0; k < NT1; k++) w1[k] = bp.taps1[k];
#pragma unroll
    for (int k = 0; k < NT2; k++) w2[k] = bp.taps2[k];

    constexpr size_t SPX = INH ? 8 : 16;
    const size_t srow = (size_t)bp.source.pitch * SPX;
    const int scol = sfirst + lane;
    const bool scol_live = scol >= bp.sx0 && scol <= bp.sx1;
    typedef const cvk_blur_halve_params __attribute__((address_space(4))) *kargs_t;
    const kargs_t ka = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    const void *src_data = bp.batch.n ? ka->batch.source[blockIdx.z] : bp.source.data;
    void *dst_data = bp.batch.n ? ka->batch.target[blockIdx.z] : bp.target.data;
    const char *sbase = reinterpret_cast<const char *>(src_data) + (ptrdiff_t)(scol - bp.source.fx0) * (ptrdiff_t)SPX;
    const size_t tpx = bp.out_half ? 8 : 16;
    char *tbase = reinterpret_cast<char *>(dst_data) + (ptrdiff_t)(tcol - bp.target.fx0) * (ptrdiff_t)tpx + (size_t)pair * (tpx / 2);
    const size_t trow = (size_t)bp.target.pitch * tpx;

    if (lane < PITCH - W) {
#pragma unroll
        for (int a = 0; a < 2; a++) { rowS[a][0][W + lane] = make_float4(0.f, 0.f, 0.f, 0.f); rowS[a][1][W + lane] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    if (lane < 2 * (HALF - W / 2)) {
        const int par = lane & 1, slot = W / 2 + (lane >> 1);
#pragma unroll
        for (int a = 0; a < 2; a++) { rowB[a][0][par][slot] = make_float4(0.f, 0.f, 0.f, 0.f); rowB[a][1][par][slot] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }

    Px ring1[RL];
    f32x2 ring2[RL];
#pragma unroll
    for (int k = 0; k < RL; k++) { ring1[k].rg = ring1[k].ba = ring2[k] = f32x2{ 0.0f, 0.0f }; }

    auto fetch_row = [&](int ys, bool wanted) -> float4 {
        const bool live = wanted && scol_live && ys >= bp.sy0 && ys <= bp.sy1;
        return fetch_px<INH>(sbase + (ptrdiff_t)(ys - bp.source.fy0) * (ptrdiff_t)srow, live);
    };
    float4 pre[4];                                             // slot = row % 4 (RL is a multiple of 4)
    pre[0] = fetch_row(ys0, true);
    pre[1] = fetch_row(ys0 + 1, steps > 1);
    const uint32_t col_mask = bcol_live ? 0xFFFFFFFFu : 0u;

    // blurred row rb (0-based in the segment) is complete when source row rb + NT1 - 1 has been filtered; it completes
    // target row ta + (rb - (NT2 - 1)) / 2 when that is a whole number >= 0.  Step (i, i + 1), i even: stage 2 takes blurred
    // rows i - NT1 - 1 and i - NT1 (completed by the previous step); the first of them is even, the second odd.
    for (int i0 = 0; i0 <= steps + 1; i0 += RL) {
        auto step = [&](auto jc) -> bool {
            constexpr int j = 2 * decltype(jc)::value;         // == i % RL, even
            const int i = i0 + j;
            if (i > steps + 1) return false;                   // uniform over the workgroup
            const int par = (i >> 1) & 1;
            pre[(j + 2) % 4] = fetch_row(ys0 + i + 2, i + 2 < steps);
            pre[(j + 3) % 4] = fetch_row(ys0 + i + 3, i + 3 < steps);
            if (i < steps) rowS[par][0][lane] = pre[j % 4];
            if (i + 1 < steps) rowS[par][1][lane] = pre[(j + 1) % 4];
            __syncthreads();                                   // both rows S of this step and both rows B of the previous one are visible

            // ---- second stage: blurred rows rbx = i - NT1 - 1 (even) and rby = i - NT1 (odd)
            constexpr int sx = (j + 2 * RL - ((NT1 + 1) % RL)) % RL, sy = (sx + 1) % RL;      // their ring2 slots (== rb % RL)
            const int rbx = i - NT1 - 1, rby = i - NT1;
            const int jl = tl < OUTW ? tl : 0;
            auto h2 = [&](int which) {
                const float2 *bb0 = reinterpret_cast<const float2 *>(&rowB[par ^ 1][which][0][0]) + pair;      // written by the previous step
                const float2 *bb1 = reinterpret_cast<const float2 *>(&rowB[par ^ 1][which][1][0]) + pair;
                return fir1<NT2>(w2, [&](int k) { const float2 v = ((k & 1) ? bb1 : bb0)[2 * (jl + (k >> 1))]; return f32x2{ v.x, v.y }; });
            };
            if (rbx >= 0) {
                ring2[sx] = h2(0);
                if (rbx >= NT2 - 1) {
                    const int t = ta + (rbx - (NT2 - 1)) / 2;
                    const f32x2 o = fir1<NT2>(w2, [&](int k) { return ring2[(sx + RL - (NT2 - 1) + k) % RL]; });
                    if (out_live && t <= tb) {
                        char *dst = tbase + (size_t)(t - bp.target.fy0) * trow;
                        if (bp.out_half) *reinterpret_cast<uint32_t *>(dst) = cvs::f2h_rz2(o.x, o.y);
                        else *reinterpret_cast<float2 *>(dst) = make_float2(o.x, o.y);
                    }
                }
            }
            if (rby >= 0) ring2[sy] = h2(1);

            // ---- first stage: H1 of both source rows, then the blurred rows they complete
            auto stage1 = [&](auto wc) {
                constexpr int which = decltype(wc)::value, jj = j + which;
                const int ii = i + which;
                if (ii < steps) {
                    const float4 *sb = rowS[par][which];
                    ring1[jj] = fir<NT1>(w1, [&](int k) { const float4 v = sb[lane + k]; return Px{ f32x2{ v.x, v.y }, f32x2{ v.z, v.w } }; });
                    if (ii >= NT1 - 1) {
                        Px b = fir<NT1>(w1, [&](int k) { return ring1[(jj + RL - (NT1 - 1) + k) % RL]; });
                        const int by = ys0 + ii - C1;
                        const uint32_t m = (by >= bp.sy0 && by <= bp.sy1) ? col_mask : 0u;
                        rowB[par][which][lane & 1][lane >> 1] = make_float4(__uint_as_float(__float_as_uint(b.rg.x) & m), __uint_as_float(__float_as_uint(b.rg.y) & m),
                                                                               __uint_as_float(__float_as_uint(b.ba.x) & m), __uint_as_float(__float_as_uint(b.ba.y) & m));
                    }
                }
            };
            stage1(std::integral_constant<int, 0>{});
            stage1(std::integral_constant<int, 1>{});
            return true;
        };
        each_slot(step, std::make_integer_sequence<int, RL / 2>{});
    }
}

template <class K>
int resident_per_cu(K kernel, int block) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, 0) != hipSuccess || n < 1) n = 1;
    return n;
}

// which form of the sweep (diagnostic build: CVS_BLUR_HALVE_FORM=1 / 2 for A/B runs)
inline bool two_rows_per_barrier() {
    static std::atomic<int> cached{ -1 };
    int v = cached.load(std::memory_order_relaxed);
    if (v < 0) { const char *e = CVS_DIAG_ENV("CVS_BLUR_HALVE_FORM"); v = e ? (atoi(e) == 2) : kTwoRowsDefault; cached.store(v, std::memory_order_relaxed); }
#ifndef CVS_DIAG
    return true;                      // the product has the two-row form only
#endif
    return v != 0;
}

template <int NT1, int NT2, int W>
int launch(cvk_blur_halve_params bp, int cus, hipStream_t s) {
    constexpr int OUTW = (W - NT1 - NT2 + 1) / 2 + 1;
    const int cols = bp.tx1 - bp.tx0 + 1, rows = bp.ty1 - bp.ty0 + 1;
    const int strips = (cols + OUTW - 1) / OUTW;
    const bool two = two_rows_per_barrier();
    static std::atomic<int> occ[2][2];               // (several threads may launch at once: pull-queue workers)
    std::atomic<int> &cached = occ[two ? 1 : 0][bp.in_half ? 1 : 0];
    int mine = cached.load(std::memory_order_relaxed);
    if (!mine) {
        if (two) mine = bp.in_half ? resident_per_cu(k_blur_halve2<NT1, NT2, W, true>, W) : resident_per_cu(k_blur_halve2<NT1, NT2, W, false>, W);
#ifdef CVS_DIAG
        else     mine = bp.in_half ? resident_per_cu(k_blur_halve<NT1, NT2, W, true>, W) : resident_per_cu(k_blur_halve<NT1, NT2, W, false>, W);
#endif
        cached.store(mine, std::memory_order_relaxed);
    }
#ifdef CVS_DIAG
    if (bp.rows_per_wg <= 0) { const char *e = CVS_DIAG_ENV("CVS_BLUR_HALVE_ROWS"); if (e && atoi(e) > 0) bp.rows_per_wg = atoi(e); }
#endif
    const int nframes = bp.batch.n > 0 ? bp.batch.n : 1;
    if (bp.rows_per_wg <= 0) {
        // one wave of resident workgroups over all the frames of the batch; a segment never shorter than its own halo
        // (NT1 + NT2 - 2 source rows = that many / 2 target rows)
        int segs = (mine * cus) / (strips * nframes);
        if (segs < 1) segs = 1;
        int r = (rows + segs - 1) / segs;
        const int lo = (NT1 + NT2) / 2;
        if (r < lo) r = lo;
        if (r > rows) r = rows;
        bp.rows_per_wg = r;
    }
    dim3 grid((unsigned)strips, (unsigned)((rows + bp.rows_per_wg - 1) / bp.rows_per_wg), (unsigned)nframes);
    if (two) {
        if (bp.in_half) hipLaunchKernelGGL((k_blur_halve2<NT1, NT2, W, true>), grid, dim3(W), 0, s, bp);
        else            hipLaunchKernelGGL((k_blur_halve2<NT1, NT2, W, false>), grid, dim3(W), 0, s, bp);
    }
#ifdef CVS_DIAG
    else {
        if (bp.in_half) hipLaunchKernelGGL((k_blur_halve<NT1, NT2, W, true>), grid, dim3(W), 0, s, bp);
        else            hipLaunchKernelGGL((k_blur_halve<NT1, NT2, W, false>), grid, dim3(W), 0, s, bp);
    }
#endif
    return (int)hipGetLastError();
}

template <int W>
int pick(const cvk_blur_halve_params &bp, int cus, hipStream_t s) {
    switch (bp.ntaps1) {
    case 3:  return launch<3, 11, W>(bp, cus, s);
    case 5:  return launch<5, 11, W>(bp, cus, s);
    case 7:  return launch<7, 11, W>(bp, cus, s);
    case 9:  return launch<9, 11, W>(bp, cus, s);
    case 11: return launch<11, 11, W>(bp, cus, s);
    default: return (int)hipErrorInvalidValue;
    }
}

}  // namespace

extern "C" int cvk_blur_halve_supported(int ntaps1, int ntaps2) {
    return ntaps2 == 11 && (ntaps1 == 3 || ntaps1 == 5 || ntaps1 == 7 || ntaps1 == 9 || ntaps1 == 11);
}

// Two source columns per lane, one wave per workgroup (blur_halve_pair_ops.hip) for frames of at least two of its strips
extern "C" int cvk_blur_halve_takes_pairs(const cvk_blur_halve_params *bp) {
    if (bp->flags & CVK_BLUR_ONE_COLUMN) return 0;
    static std::atomic<int> env_cached{ -2 };
    int env = env_cached.load(std::memory_order_relaxed);
    if (env == -2) { const char *e = CVS_DIAG_ENV("CVS_BLUR_HALVE_PAIR"); env = e ? atoi(e) : -1; env_cached.store(env, std::memory_order_relaxed); }
    if (env == 0) return 0;
    if (!(bp->flags & CVK_BLUR_TWO_COLUMNS) && bp->tx1 - bp->tx0 + 1 < 108) return 0;
    return cvk_blur_halve_pair_supported(bp);
}

extern "C" int cvk_blur_halve(const cvk_blur_halve_params *bp, int cus, void *stream) {
    if (bp->tx1 < bp->tx0 || bp->ty1 < bp->ty0) return 0;
    if (!cvk_blur_halve_supported(bp->ntaps1, bp->ntaps2)) return (int)hipErrorInvalidValue;
    if (cvk_blur_halve_takes_pairs(bp)) return cvk_blur_halve_pair(bp, cus, stream);
    static std::atomic<int> env_cached{ -1 };
    int env_w = env_cached.load(std::memory_order_relaxed);
    if (env_w < 0) { const char *e = CVS_DIAG_ENV("CVS_BLUR_HALVE_WIDTH"); env_w = e ? atoi(e) : 0; env_cached.store(env_w, std::memory_order_relaxed); }
    const int cols = bp->tx1 - bp->tx0 + 1;
    // strips of 256 lanes, except narrow frames -- and batches: callers of the batch entry keep calls in flight, and with
    // two of them overlapping two-wave workgroups (twice as many barrier groups per CU) are 3 % faster at 4K, 1 % slower
    // for one call at a time (profiles/r03/config3_batches.txt)
    const int width = env_w == 64 || env_w == 128 || env_w == 256 ? env_w : (cols <= 60 || bp->batch.n >= 2 ? 128 : 256);
#ifdef CVS_DIAG
    if (width == 64) return pick<64>(*bp, cus, (hipStream_t)stream);         // experiment: one wave per workgroup, the barrier costs nothing
#endif
    return width == 128 ? pick<128>(*bp, cus, (hipStream_t)stream) : pick<256>(*bp, cus, (hipStream_t)stream);
}
