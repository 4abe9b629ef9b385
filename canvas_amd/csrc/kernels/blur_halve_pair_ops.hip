// blur_halve_pair_ops.hip -- BASELINE config 3 in one sweep (blur_halve_ops.hip: separable blur, then the Lanczos resampler at
// factor 1/2, nothing in between ever in HBM) with TWO source columns per lane and one or two waves per workgroup.
//
// Same four sums in the same order as k_blur_halve (blur x, blur y, resample x, resample y; every product and every addition
// rounded on its own; blurred pixels outside the SOURCE's window are zero, not the blur formula evaluated there).  What
// changes is the shape of the sweep, for the reason k_blur_pair (blur_pair_ops.hip) changed k_blur's: k_blur_halve's row step
// is a latency chain through two LDS rows and a workgroup barrier that three waves per SIMD leave the vector ALU idle half
// the time on (profiles/r03); a workgroup of ONE wave has no barrier at all -- LDS runs a wave's accesses in order -- and with
// two columns per lane its strip is 128 source columns wide, so the horizontal halo stays at 14 %:
//   * a lane loads source columns (2l, 2l + 1) of the strip as one 16-byte buffer load (range-checked row descriptors: columns
//     outside the window, rows outside it and rows past the segment come back as zeros without a predicate), three rows ahead;
//     a row goes to LDS (de-interleaved by column parity) during the step before the one that filters it;
//   * first stage: H1 for the lane's two blurred columns from NT1 + 1 shared neighbours, ring of NT1 rows x two columns in
//     registers (slot = a compile-time constant: the loop body is written out 2 NT1 times), V1 -> the blurred pair, zeroed
//     outside the window, -> LDS row B (de-interleaved the same way);
//   * second stage one step later, on the lanes that own a target column (54 of 64 at 9 + 11 taps): H2 from NT2 neighbours of
//     row B -> a WINDOW of NT2 + 1 rows in registers that moves down by two every second step (the step parity, i.e. whether
//     the row completes a target row, is a compile-time fact of the unrolled body; a ring with compile-time slots for both
//     stages would need lcm(NT1, NT2 + 1) copies of the body), V2 -> one f16 pixel, an 8-byte buffer store.
// f16 in, f16 out, NT2 = 11 (Lanczos3 at 1/2), NT1 = 3..11 odd -- or 1, the identity: the resampler alone; cvk_blur_halve sends a launch here when
// cvk_blur_halve_pair_supported says so.
#include <cstdlib>
#include <atomic>
#include "pair_common.hpp"

namespace {

using namespace pairsweep;

__device__ __forceinline__ float4 masked(Px b, uint32_t m) {
    return make_float4(__uint_as_float(__float_as_uint(b.rg.x) & m), __uint_as_float(__float_as_uint(b.rg.y) & m),
                       __uint_as_float(__float_as_uint(b.ba.x) & m), __uint_as_float(__float_as_uint(b.ba.y) & m));
}

// geometry of a strip (compile-time): D moves the strip's first source column one to the left when C1 + C2 is odd, so that
// source pairs start on even columns; BL lanes own a blurred pair, OUTW lanes a target column
template <int NT1, int NT2, int W_> struct Strip {
    static constexpr int W = W_, C1 = NT1 / 2, C2 = NT2 / 2, D = (C1 + C2) & 1;
    static constexpr int BL = ((2 * W - 1 - NT1 - D) >> 1) + 1;         // 2 l + 1 + (NT1 - 1) + D <= 2 W - 1
    static constexpr int OUTW = (2 * BL - NT2) / 2 + 1;                  // 2 tl + NT2 - 1 <= 2 BL - 1
    static constexpr int PITCH = W + 8;
    static_assert(C1 + 1 <= 8 && C2 + 1 <= 8 && OUTW >= 8 && OUTW <= W, "strip layout");
};

template <int NT1, int NT2, int WG>
__global__ __launch_bounds__(WG) void k_blur_halve_pair(cvk_blur_halve_params bp) {
    typedef Strip<NT1, NT2, WG> G;
    constexpr int W = G::W, C1 = G::C1, C2 = G::C2, D = G::D, OUTW = G::OUTW, PITCH = G::PITCH;
    static_assert(NT1 % 2 == 1 && NT2 % 2 == 1 && NT1 >= 1, "odd tap counts");
    // Steps per copy of the loop body: the first ring's slot (jj % NT1), the step's parity and the SECOND stage's ring slot
    // are its index.  The second stage keeps the last NT2 + 1 rows it filtered in a ring of UB slots -- as long as the body, so
    // that a row's slot (row % UB) is a compile-time fact of the step that writes or reads it and the ring looks the same at
    // the end of a body as at its start: no row ever moves.  (Until round 4 this was a window of NT2 + 1 rows shifted down by
    // two every second step: 22 register moves per target row, a tenth of the contracted build's vector instructions.)
    constexpr int WL = NT2 + 1;                                // rows of the second stage that are alive at any time
    constexpr int UB = 2 * NT1 * ((WL + 2 * NT1 - 1) / (2 * NT1));
    static_assert(UB % NT1 == 0 && UB % 2 == 0 && UB >= WL, "body length");
    __shared__ float4 rowS[2][2][PITCH];                       // source row, widened: [step parity][column parity][column / 2]
    __shared__ float4 rowB[2][2][PITCH];                       // blurred row, the same way
    const int lane = threadIdx.x;
    const int xo = bp.tx0 + (int)blockIdx.x * OUTW;            // first target column of the strip
    const int bo = 2 * xo - C2;                                // first blurred column of the strip (lane l owns bo + 2 l, bo + 2 l + 1)
    const int so = bo - C1 - D;                                // first source column of the strip (lane l loads so + 2 l, so + 2 l + 1)
    const int tcol = xo + lane;                                // the target column this lane produces (lane < OUTW)
    const int ta = bp.ty0 + (int)blockIdx.y * bp.rows_per_wg;
    const int tb = min(ta + bp.rows_per_wg - 1, bp.ty1);
    const int ys0 = 2 * ta - C2 - C1;                          // first source row the segment needs
    const int steps = 2 * (tb - ta) + NT2 + NT1 - 1;           // source rows 0 .. steps - 1 of the segment

    float w1[NT1], w2[NT2];
#pragma unroll
    for (int k = 0; k < NT1; k++) w1[k] = bp.taps1[k];
#pragma unroll
    for (int k = 0; k < NT2; k++) w2[k] = bp.taps2[k];

    // a batch of frames: grid.z picks the frame (pointers read through the kernel-argument segment, see blur_kernel.hpp)
    typedef const cvk_blur_halve_params __attribute__((address_space(4))) *kargs_t;
    const kargs_t ka = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    const void *src_data = bp.batch.n ? ka->batch.source[blockIdx.z] : bp.source.data;
    void *dst_data = bp.batch.n ? ka->batch.target[blockIdx.z] : bp.target.data;

    // byte offsets inside one row of the source window / of the target rectangle; a column left of either is a huge unsigned
    // offset, and so is every lane without a target column: out of range
    const size_t srow = (size_t)bp.source.pitch * 8, trow = (size_t)bp.target.pitch * 8;
    const uint32_t soff = (uint32_t)((so + 2 * lane - bp.sx0) * 8);
    const uint32_t toff = lane < OUTW ? (uint32_t)((tcol - bp.tx0) * 8) : 0x80000000u;
    const uint32_t swin = (uint32_t)(bp.sx1 - bp.sx0 + 1) * 8u, trect = (uint32_t)(bp.tx1 - bp.tx0 + 1) * 8u;
    const char *swin0 = reinterpret_cast<const char *>(src_data) + (ptrdiff_t)(bp.sx0 - bp.source.fx0) * 8;
    const ptrdiff_t trect0 = (ptrdiff_t)(bp.tx0 - bp.target.fx0) * 8;
    // is this lane's blurred pixel inside the blurred frame's window (= the source's)?  column part, as all-ones / all-zeros
    // (lanes past the last blurred pair hold anything: their slots of row B feed only lanes without a target column)
    const int bcol = bo + 2 * lane;
    const uint32_t cm0 = (bcol >= bp.sx0 && bcol <= bp.sx1) ? 0xFFFFFFFFu : 0u;
    const uint32_t cm1 = (bcol + 1 >= bp.sx0 && bcol + 1 <= bp.sx1) ? 0xFFFFFFFFu : 0u;
    const bool cols_inside = bo >= bp.sx0 && bo + 2 * W - 1 <= bp.sx1;     // uniform: no column of the strip needs the mask

    if (lane < PITCH - W) {
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int ph = 0; ph < 2; ph++) { rowS[b][ph][W + lane] = make_float4(0.f, 0.f, 0.f, 0.f); rowB[b][ph][W + lane] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int ph = 0; ph < 2; ph++) rowB[b][ph][lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    Px ring1[NT1][2];
    Px ring2[UB];
#pragma unroll
    for (int k = 0; k < NT1; k++) ring1[k][0].rg = ring1[k][0].ba = ring1[k][1].rg = ring1[k][1].ba = f32x2{ 0.0f, 0.0f };
#pragma unroll
    for (int k = 0; k < UB; k++) ring2[k].rg = ring2[k].ba = f32x2{ 0.0f, 0.0f };

    auto fetch_row = [&](int ys, bool wanted) -> u32x4 {
        const bool live = wanted && ys >= bp.sy0 && ys <= bp.sy1;           // uniform: a dead row is a descriptor without records
        return load_pair(row_rsrc(swin0, (size_t)((ptrdiff_t)(ys - bp.source.fy0) * (ptrdiff_t)srow), live ? swin : 0u), soff);
    };
    // row i goes to LDS during step i - 1; the rows in flight are those of steps i + 1 .. i + 3
    u32x4 cur = fetch_row(ys0, true);
    rowS[0][0][lane] = widen_px(cur.x, cur.y);
    rowS[0][1][lane] = widen_px(cur.z, cur.w);
    cur = fetch_row(ys0 + 1, steps > 1);
    u32x4 nxt = fetch_row(ys0 + 2, steps > 2);
    __syncthreads();

    // step i: source row ys0 + i is filtered; blurred row rb = i - (NT1 - 1) leaves the first ring (rb >= 0) and is handed over
    // through LDS; in step i + 1 it enters the second stage as rb2 = (i + 1) - NT1; an EVEN rb2 >= NT2 - 1 completes target row
    // ta + (rb2 - (NT2 - 1)) / 2 -- rb2 is even exactly in the odd steps (NT1 odd).  One step past the last row drains the hand-over.
    for (int i0 = 0; i0 <= steps; i0 += UB) {
        auto step = [&](auto jc) -> bool {
            constexpr int jj = decltype(jc)::value;            // == i % UB: first ring's slot jj % NT1, step parity jj & 1
            constexpr int j1 = jj % NT1, par = jj & 1;
            constexpr int s2 = ((jj - NT1) % UB + UB) % UB;    // second ring's slot of blurred row rb2 = i - NT1
            const int i = i0 + jj;
            if (i > steps) return false;                       // uniform over the workgroup
            const u32x4 far = fetch_row(ys0 + i + 3, i + 3 < steps);

            // ---- second stage first: blurred row rb2 = i - NT1, written to row B by the previous step
            const int rb2 = i - NT1;
            if (rb2 >= 0) {
                float4 (*bb)[PITCH] = rowB[par ^ 1];
                float4 u[NT2];
#pragma unroll
                for (int k = 0; k < NT2; k++) u[k] = bb[k & 1][lane + (k >> 1)];
                f32x2 hrg, hba;
#pragma unroll
                for (int k = 0; k < NT2; k++) {
                    // t += s * c: product and sum rounded apart, or (the clang build, CVS_CONTRACT) as one fused multiply-add
                    if constexpr (cvs::kContract) {
                        if (k == 0) { hrg = f32x2{ u[0].x, u[0].y } * w2[0]; hba = f32x2{ u[0].z, u[0].w } * w2[0]; }
                        else { hrg = cvs::madd(f32x2{ u[k].x, u[k].y }, w2[k], hrg); hba = cvs::madd(f32x2{ u[k].z, u[k].w }, w2[k], hba); }
                    } else {
                    const f32x2 p = f32x2{ u[k].x, u[k].y } * w2[k], q = f32x2{ u[k].z, u[k].w } * w2[k];
                    if (k == 0) { hrg = p; hba = q; } else { hrg = hrg + p; hba = hba + q; }
                    }
                }
                ring2[s2].rg = hrg;
                ring2[s2].ba = hba;
            }
            if constexpr (par == 1) {
                if (rb2 >= NT2 - 1) {
                    const int t = ta + (rb2 - (NT2 - 1)) / 2;
                    f32x2 org, oba;
#pragma unroll
                    for (int k = 0; k < NT2; k++) {
                        const Px &row = ring2[(s2 + UB - (NT2 - 1) + k) % UB];       // blurred row rb2 - (NT2 - 1) + k
                        if constexpr (cvs::kContract) {
                            if (k == 0) { org = row.rg * w2[0]; oba = row.ba * w2[0]; }
                            else { org = cvs::madd(row.rg, w2[k], org); oba = cvs::madd(row.ba, w2[k], oba); }
                        } else {
                        const f32x2 p = row.rg * w2[k], q = row.ba * w2[k];
                        if (k == 0) { org = p; oba = q; } else { org = org + p; oba = oba + q; }
                        }
                    }
                    // truncation to half: |x| >= 65536 must become an infinity, which only waves that hold such a value pay for
                    float big;
                    asm("v_max3_f32 %0, |%1|, |%2|, |%3|\n\tv_max_f32 %0, %0, |%4|" : "=&v"(big) : "v"(org.x), "v"(org.y), "v"(oba.x), "v"(oba.y));
                    if (cvs::wave_any(!(big < 65536.0f))) { cvs::rare_path(); org = cvs::saturate_to_inf2(org); oba = cvs::saturate_to_inf2(oba); }
                    const u32x2 codes = { cvs::pkrtz(org.x, org.y), cvs::pkrtz(oba.x, oba.y) };
                    __builtin_amdgcn_raw_buffer_store_b64(codes, row_rsrc(dst_data, (size_t)((ptrdiff_t)(t - bp.target.fy0) * (ptrdiff_t)trow + trect0), t <= tb ? trect : 0u), (int)toff, 0, 0);
                }
            }

            // ---- first stage: H1 of this source row, then the blurred row it completes
            if (i < steps) {
                float4 (*sb)[PITCH] = rowS[par];
                float4 v[NT1 + 1];
#pragma unroll
                for (int c = 0; c <= NT1; c++) v[c] = sb[(c + D) & 1][lane + ((c + D) >> 1)];
                // the next step's row, into the other buffer (last read a step ago)
                float4 (*nb)[PITCH] = rowS[par ^ 1];
                nb[0][lane] = widen_px(cur.x, cur.y);
                nb[1][lane] = widen_px(cur.z, cur.w);
                f32x2 rg0, ba0, rg1, ba1;
                // (NT1 == 1: the host sends only the identity, one tap of weight 1.0f -- x * 1.0f is x, twice: the resampler alone)
                if constexpr (NT1 == 1) { rg0 = f32x2{ v[0].x, v[0].y }; ba0 = f32x2{ v[0].z, v[0].w }; rg1 = f32x2{ v[1].x, v[1].y }; ba1 = f32x2{ v[1].z, v[1].w }; }
#pragma unroll
                for (int k = 0; k < (NT1 == 1 ? 0 : NT1); k++) {
                    if constexpr (cvs::kContract) {
                        if (k == 0) { rg0 = f32x2{ v[0].x, v[0].y } * w1[0]; ba0 = f32x2{ v[0].z, v[0].w } * w1[0]; rg1 = f32x2{ v[1].x, v[1].y } * w1[0]; ba1 = f32x2{ v[1].z, v[1].w } * w1[0]; }
                        else {
                            rg0 = cvs::madd(f32x2{ v[k].x, v[k].y }, w1[k], rg0); ba0 = cvs::madd(f32x2{ v[k].z, v[k].w }, w1[k], ba0);
                            rg1 = cvs::madd(f32x2{ v[k + 1].x, v[k + 1].y }, w1[k], rg1); ba1 = cvs::madd(f32x2{ v[k + 1].z, v[k + 1].w }, w1[k], ba1);
                        }
                    } else {
                    const f32x2 p0 = f32x2{ v[k].x, v[k].y } * w1[k], q0 = f32x2{ v[k].z, v[k].w } * w1[k];
                    const f32x2 p1 = f32x2{ v[k + 1].x, v[k + 1].y } * w1[k], q1 = f32x2{ v[k + 1].z, v[k + 1].w } * w1[k];
                    if (k == 0) { rg0 = p0; ba0 = q0; rg1 = p1; ba1 = q1; }
                    else { rg0 = rg0 + p0; ba0 = ba0 + q0; rg1 = rg1 + p1; ba1 = ba1 + q1; }
                    }
                }
                ring1[j1][0].rg = rg0; ring1[j1][0].ba = ba0;
                ring1[j1][1].rg = rg1; ring1[j1][1].ba = ba1;
                if (i >= NT1 - 1) {
                    Px b0, b1;
                    if constexpr (NT1 == 1) { b0 = ring1[0][0]; b1 = ring1[0][1]; }
#pragma unroll
                    for (int k = 0; k < (NT1 == 1 ? 0 : NT1); k++) {
                        const Px &a = ring1[(j1 + 1 + k) % NT1][0], &b = ring1[(j1 + 1 + k) % NT1][1];
                        if constexpr (cvs::kContract) {
                            if (k == 0) { b0.rg = a.rg * w1[0]; b0.ba = a.ba * w1[0]; b1.rg = b.rg * w1[0]; b1.ba = b.ba * w1[0]; }
                            else { b0.rg = cvs::madd(a.rg, w1[k], b0.rg); b0.ba = cvs::madd(a.ba, w1[k], b0.ba); b1.rg = cvs::madd(b.rg, w1[k], b1.rg); b1.ba = cvs::madd(b.ba, w1[k], b1.ba); }
                        } else {
                        const f32x2 p0 = a.rg * w1[k], q0 = a.ba * w1[k], p1 = b.rg * w1[k], q1 = b.ba * w1[k];
                        if (k == 0) { b0.rg = p0; b0.ba = q0; b1.rg = p1; b1.ba = q1; }
                        else { b0.rg = b0.rg + p0; b0.ba = b0.ba + q0; b1.rg = b1.rg + p1; b1.ba = b1.ba + q1; }
                        }
                    }
                    const int by = ys0 + i - C1;               // the blurred row just completed
                    const bool row_in = by >= bp.sy0 && by <= bp.sy1;       // uniform
                    if (cols_inside && row_in) {
                        rowB[par][0][lane] = make_float4(b0.rg.x, b0.rg.y, b0.ba.x, b0.ba.y);
                        rowB[par][1][lane] = make_float4(b1.rg.x, b1.rg.y, b1.ba.x, b1.ba.y);
                    } else {
                        rowB[par][0][lane] = masked(b0, row_in ? cm0 : 0u);
                        rowB[par][1][lane] = masked(b1, row_in ? cm1 : 0u);
                    }
                }
            }
            cur = nxt;
            nxt = far;
            __syncthreads();                                   // (one wave: a compiler fence; the hardware keeps the wave's LDS accesses in order)
            return true;
        };
        each_slot(step, std::make_integer_sequence<int, UB>{});
    }
}

template <int NT1, int NT2, int WG>
int launch(cvk_blur_halve_params bp, int cus, hipStream_t s) {
    constexpr int OUTW = Strip<NT1, NT2, WG>::OUTW;
    const int cols = bp.tx1 - bp.tx0 + 1, rows = bp.ty1 - bp.ty0 + 1;
    const int strips = (cols + OUTW - 1) / OUTW;
    static std::atomic<int> cached{ 0 };            // (several threads may launch at once: pull-queue workers)
    int mine = cached.load(std::memory_order_relaxed);
    if (!mine) {
        mine = resident_per_cu(k_blur_halve_pair<NT1, NT2, WG>, WG);
        const char *e = CVS_DIAG_ENV("CVS_BLUR_HALVE_PAIR_WGS");     // diagnostic build: workgroups per CU the segments are sized for
        if (e && atoi(e) > 0) mine = atoi(e);
        cached.store(mine, std::memory_order_relaxed);
    }
    const int nframes = bp.batch.n > 0 ? bp.batch.n : 1;
    if (bp.rows_per_wg <= 0) {
        int segs = (mine * cus) / (strips * nframes);
        if (segs < 1) segs = 1;
        int r = (rows + segs - 1) / segs;
        const int lo = (NT1 + NT2) / 2;             // a segment never shorter than its own halo (NT1 + NT2 - 2 source rows)
        if (r < lo) r = lo;
        if (r > rows) r = rows;
        bp.rows_per_wg = r;
    }
    dim3 grid((unsigned)strips, (unsigned)((rows + bp.rows_per_wg - 1) / bp.rows_per_wg), (unsigned)nframes);
    hipLaunchKernelGGL((k_blur_halve_pair<NT1, NT2, WG>), grid, dim3(WG), 0, s, bp);
    return (int)hipGetLastError();
}

}  // namespace

// f16 in and out, Lanczos3 halving behind a blur of 3..11 taps, and every pair of source columns whole and on a 16-byte boundary
extern "C" int cvk_blur_halve_pair_supported(const cvk_blur_halve_params *bp) {
    if (!(bp->in_half && bp->out_half) || bp->ntaps2 != 11) return 0;
    if (!(bp->ntaps1 & 1) || bp->ntaps1 < 1 || bp->ntaps1 > 11) return 0;
    if (bp->ntaps1 == 1 && bp->taps1[0] != 1.0f) return 0;               /* one tap: the identity only (the resampler alone) */
    if (bp->sx1 < bp->sx0 || bp->sy1 < bp->sy0) return 0;
    const int c1 = bp->ntaps1 / 2, c2 = bp->ntaps2 / 2, d = (c1 + c2) & 1;
    // even pitch, the window on a pair boundary of its buffer and of the strips' pair grid, an even width
    if (bp->source.pitch & 1) return 0;
    if (((bp->sx0 - bp->source.fx0) | (bp->sx1 - bp->sx0 + 1) | (2 * bp->tx0 - c2 - c1 - d - bp->sx0)) & 1) return 0;
    const int n = bp->batch.n > 0 ? bp->batch.n : 1;
    if (n > CVK_FRAME_BATCH) return 0;
    for (int z = 0; z < n; z++) {
        const void *src = bp->batch.n ? bp->batch.source[z] : bp->source.data;
        const void *dst = bp->batch.n ? bp->batch.target[z] : bp->target.data;
        if (!aligned16(src) || (((uintptr_t)dst) & 7u)) return 0;
    }
    return 1;
}

extern "C" int cvk_blur_halve_pair(const cvk_blur_halve_params *bp, int cus, void *stream) {
    if (bp->tx1 < bp->tx0 || bp->ty1 < bp->ty0) return 0;
    if (!cvk_blur_halve_pair_supported(bp)) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    // 128 lanes (two waves, one barrier per row, 236 of 256 source columns useful instead of 108 of 128) from two such strips
    // on: 3 % faster at 4K (0.0465 -> 0.0451 ms per frame); narrower targets keep the one-wave workgroups
    static std::atomic<int> env_cached{ -1 };
    int env_w = env_cached.load(std::memory_order_relaxed);
    if (env_w < 0) { const char *e = CVS_DIAG_ENV("CVS_BLUR_HALVE_PAIR_WIDTH"); env_w = e ? atoi(e) : 0; env_cached.store(env_w, std::memory_order_relaxed); }
    const int cols = bp->tx1 - bp->tx0 + 1;
    const bool wide = env_w == 128 || (env_w != 64 && cols >= 2 * Strip<9, 11, 128>::OUTW);       // (118 columns at 9 taps)
    if (wide) {
        switch (bp->ntaps1) {
        case 1:  return launch<1, 11, 128>(*bp, cus, s);
        case 3:  return launch<3, 11, 128>(*bp, cus, s);
        case 5:  return launch<5, 11, 128>(*bp, cus, s);
        case 7:  return launch<7, 11, 128>(*bp, cus, s);
        case 9:  return launch<9, 11, 128>(*bp, cus, s);
        case 11: return launch<11, 11, 128>(*bp, cus, s);
        }
    }
    switch (bp->ntaps1) {
    case 1:  return launch<1, 11, 64>(*bp, cus, s);
    case 3:  return launch<3, 11, 64>(*bp, cus, s);
    case 5:  return launch<5, 11, 64>(*bp, cus, s);
    case 7:  return launch<7, 11, 64>(*bp, cus, s);
    case 9:  return launch<9, 11, 64>(*bp, cus, s);
    case 11: return launch<11, 11, 64>(*bp, cus, s);
    }
    return (int)hipErrorInvalidValue;
}
