// blur_pair_ops.hip -- the register-window blur of blur_kernel.hpp with TWO neighbouring target columns per lane (f16 in,
// f16 out, 1:1, 3..13 taps odd), with or without the workspace stack's over epilogue.
//
// Same sums in the same order as k_blur (target = sum_k taps[k] * H(y - c + k), H = sum_k taps[k] * src(x - c + k), every
// product and every addition rounded on its own, first product instead of 0 + p0); what changes is who computes them:
//   * a lane loads the two source pixels (2l, 2l+1) of its strip as ONE 16-byte load and stores its two results as one;
//   * the LDS row is kept de-interleaved (even columns, odd columns), so the NT + 1 neighbours the two sums share are
//     NT + 1 conflict-free 16-byte reads instead of 2 NT: LDS time per pixel drops to (NT + 1) / 2 NT of k_blur's;
//   * the epilogue blends PAIRS (chain_math.hpp over_pair: every step a packed instruction over both pixels, one refined
//     reciprocal per pixel shared by three quotients) -- k_blur's per-pixel form spends 43 vector instructions per layer,
//     this one 31 per pixel -- and everything per step that is not arithmetic (addresses, loop control, the barrier)
//     is paid once per two pixels;
//   * the vertical ring holds both columns: 8 NT registers;
//   * every global access is a buffer operation through a descriptor of ONE ROW of the window (source) or of the target
//     rectangle (upper layers, store): columns outside it, rows outside the window (a descriptor of zero records) and the
//     surplus lanes of a strip are dropped or zero-filled by the range check of the memory pipeline -- no per-lane
//     predicates, no exec-mask regions, one 32-bit offset register per buffer.  The host sends a launch here only when no
//     pair straddles an edge (window and rectangle start on a pair boundary and have even widths).
// Measured where DESIGN.md section 4.2 says; cvk_blur (blur_ops.hip) sends a launch here when cvk_blur_pair_supported says so.
#include <cstdlib>
#include <atomic>
#include "pair_common.hpp"

namespace {

using namespace pairsweep;

// NOV: the number of upper layers, exact (0: no epilogue) -- a kernel per count, so that no step carries loads, registers
// or branches for layers that are not there
template <int NT, int W, int NOV>
__global__ __launch_bounds__(W) void k_blur_pair(cvk_blur_params bp) {
    // D: the strip's first source column is moved one to the left when the centre tap is odd, so that source pairs and
    // target pairs both start on even columns of their strips
    constexpr int C = NT / 2, D = C & 1, OUTW = 2 * W - 2 * C - 2 * D, PITCH = W + 8;
    static_assert(NT & 1 && NT >= 3 && NT <= 15, "odd tap counts up to 15");
    static_assert(NOV >= 0 && NOV <= CVK_BLUR_MAX_OVER, "layer count");
    __shared__ float4 rowbuf[2][2][PITCH];                   // [row parity][column phase][slot]
    const int lane = threadIdx.x;
    const int xo = bp.tx0 + (int)blockIdx.x * OUTW;          // first target column of the strip
    const int sfirst = xo - C - D;                           // first source column of the strip
    const int tcol = xo + 2 * lane;                          // this lane's target columns: tcol, tcol + 1
    const int ta = bp.ty0 + (int)blockIdx.y * bp.rows_per_wg;
    const int tb = min(ta + bp.rows_per_wg - 1, bp.ty1);
    const int ys0 = ta - C;                                  // first source row the segment needs
    const int steps = (tb - ta) + NT;

    float w[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) w[k] = bp.taps[k];

    // a batch of frames: grid.z picks the frame (pointers read through the kernel-argument segment, see blur_kernel.hpp)
    const int z = (int)blockIdx.z;
    typedef const cvk_blur_params __attribute__((address_space(4))) *kargs_t;
    const kargs_t ka = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    const void *src_data = bp.batch.n ? ka->batch.source[z] : bp.source.data;
    void *dst_data = bp.batch.n ? ka->batch.target[z] : bp.target.data;
    const void *over_data[NOV > 0 ? NOV : 1];
#pragma unroll
    for (int l = 0; l < NOV; l++) over_data[l] = bp.batch.n ? ka->batch.over[z][l] : bp.over[l];

    // byte offsets of this lane's pairs inside one row of the source window / of the target rectangle; a column left of
    // either comes out as a huge unsigned offset, and so does every lane beyond the strip's last target pair: out of range
    const size_t srow = (size_t)bp.source.pitch * 8, trow = (size_t)bp.target.pitch * 8;
    const uint32_t soff = (uint32_t)((sfirst + 2 * lane - bp.sx0) * 8);
    const uint32_t toff = 2 * lane < OUTW ? (uint32_t)((tcol - bp.tx0) * 8) : 0x80000000u;
#ifdef CVS_DIAG
    // timing only (tools/): descriptors without records drop the source loads (1), the layers' loads (2), the stores (4)
    const uint32_t swin = (bp.flags & 0x100) ? 0u : (uint32_t)(bp.sx1 - bp.sx0 + 1) * 8u, trect = (uint32_t)(bp.tx1 - bp.tx0 + 1) * 8u;
    const uint32_t orect = (bp.flags & 0x200) ? 0u : trect, wrect = (bp.flags & 0x400) ? 0u : trect;
#else
    const uint32_t swin = (uint32_t)(bp.sx1 - bp.sx0 + 1) * 8u, trect = (uint32_t)(bp.tx1 - bp.tx0 + 1) * 8u, orect = trect, wrect = trect;
#endif
    const char *swin0 = reinterpret_cast<const char *>(src_data) + (ptrdiff_t)(bp.sx0 - bp.source.fx0) * 8;
    const ptrdiff_t trect0 = (ptrdiff_t)(bp.tx0 - bp.target.fx0) * 8;

    if (lane < PITCH - W) {
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int ph = 0; ph < 2; ph++) rowbuf[b][ph][W + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    Px ring[NT][2];
#pragma unroll
    for (int k = 0; k < NT; k++) ring[k][0].rg = ring[k][0].ba = ring[k][1].rg = ring[k][1].ba = f32x2{ 0.0f, 0.0f };

    auto fetch_row = [&](int ys, bool wanted) {
        const bool live = wanted && ys >= bp.sy0 && ys <= bp.sy1;           // uniform: a dead row is a descriptor without records
        return load_pair(row_rsrc(swin0, (size_t)((ptrdiff_t)(ys - bp.source.fy0) * (ptrdiff_t)srow), live ? swin : 0u), soff);
    };
    // the upper layers of the pixels that step i completes (a step that completes nothing: descriptors without records)
    struct Layers { u32x4 v[NOV > 0 ? NOV : 1]; };
    auto fetch_layers = [&](int i) {
        Layers r;
        const bool emits = i >= NT - 1 && i < steps;
        const size_t o = (size_t)((ptrdiff_t)(ta + i - (NT - 1) - bp.target.fy0) * (ptrdiff_t)trow + trect0);
#pragma unroll
        for (int l = 0; l < NOV; l++) r.v[l] = load_pair(row_rsrc(over_data[l], o, emits ? orect : 0u), toff);
        return r;
    };
    // A row goes to LDS during the step BEFORE the one that filters it (right behind that step's reads, into the other
    // buffer), so no step starts by waiting for its own write (2 % on the launch); the rows in flight are those of steps
    // i+1 .. i+3, the upper layers are requested one step ahead (a second set in flight would cost the third wave per SIMD).
    u32x4 cur = fetch_row(ys0, true);
    rowbuf[0][0][lane] = widen_px(cur.x, cur.y);
    rowbuf[0][1][lane] = widen_px(cur.z, cur.w);
    cur = fetch_row(ys0 + 1, steps > 1);
    u32x4 nxt = fetch_row(ys0 + 2, steps > 2);
    Layers ov_cur = fetch_layers(0);                          // (NT >= 3: step 0 completes nothing)

    for (int i0 = 0; i0 < steps; i0 += NT) {
        // NT steps with the ring slot as a compile-time constant
        auto step = [&](auto jc) -> bool {
            constexpr int j = decltype(jc)::value;
            const int i = i0 + j;
            if (i >= steps) return false;                     // uniform over the workgroup
            const bool emits = i >= NT - 1;                   // uniform
            const int t = ta + (i - (NT - 1));                // the target row this step completes
            const u32x4 far = fetch_row(ys0 + i + 3, i + 3 < steps);
            const Layers ov_nxt = fetch_layers(i + 1);
            float4 (*buf)[PITCH] = rowbuf[i & 1];
            __syncthreads();
            // the NT + 1 neighbours both sums draw on: source column 2 * lane + D + c of the strip
            float4 v[NT + 1];
#pragma unroll
            for (int c = 0; c <= NT; c++) v[c] = buf[(c + D) & 1][lane + ((c + D) >> 1)];
            // the next step's row, into the other buffer (last read a step ago)
            float4 (*nbuf)[PITCH] = rowbuf[(i + 1) & 1];
            nbuf[0][lane] = widen_px(cur.x, cur.y);
            nbuf[1][lane] = widen_px(cur.z, cur.w);
            cur = nxt;
            nxt = far;
            // four chains side by side (two channel pairs x two pixels): a packed add never sits right behind the multiply it needs
            f32x2 rg0, ba0, rg1, ba1;
#pragma unroll
            for (int k = 0; k < NT; k++) {
                if constexpr (cvs::kContract) {
                    // the clang build's t += s * c: one fused multiply-add per tap, the first a plain product
                    if (k == 0) { rg0 = f32x2{ v[0].x, v[0].y } * w[0]; ba0 = f32x2{ v[0].z, v[0].w } * w[0]; rg1 = f32x2{ v[1].x, v[1].y } * w[0]; ba1 = f32x2{ v[1].z, v[1].w } * w[0]; }
                    else {
                        rg0 = cvs::madd(f32x2{ v[k].x, v[k].y }, w[k], rg0); ba0 = cvs::madd(f32x2{ v[k].z, v[k].w }, w[k], ba0);
                        rg1 = cvs::madd(f32x2{ v[k + 1].x, v[k + 1].y }, w[k], rg1); ba1 = cvs::madd(f32x2{ v[k + 1].z, v[k + 1].w }, w[k], ba1);
                    }
                } else {
                const f32x2 p0 = f32x2{ v[k].x, v[k].y } * w[k], q0 = f32x2{ v[k].z, v[k].w } * w[k];
                const f32x2 p1 = f32x2{ v[k + 1].x, v[k + 1].y } * w[k], q1 = f32x2{ v[k + 1].z, v[k + 1].w } * w[k];
                if (k == 0) { rg0 = p0; ba0 = q0; rg1 = p1; ba1 = q1; }
                else { rg0 = rg0 + p0; ba0 = ba0 + q0; rg1 = rg1 + p1; ba1 = ba1 + q1; }
                }
            }
            ring[j][0].rg = rg0; ring[j][0].ba = ba0;
            ring[j][1].rg = rg1; ring[j][1].ba = ba1;
            if (emits) {
                // ring[(j+1) % NT] is the oldest row = tap 0
                f32x2 org0, oba0, org1, oba1;
#pragma unroll
                for (int k = 0; k < NT; k++) {
                    const Px &a = ring[(j + 1 + k) % NT][0], &b = ring[(j + 1 + k) % NT][1];
                    if constexpr (cvs::kContract) {
                        if (k == 0) { org0 = a.rg * w[0]; oba0 = a.ba * w[0]; org1 = b.rg * w[0]; oba1 = b.ba * w[0]; }
                        else { org0 = cvs::madd(a.rg, w[k], org0); oba0 = cvs::madd(a.ba, w[k], oba0); org1 = cvs::madd(b.rg, w[k], org1); oba1 = cvs::madd(b.ba, w[k], oba1); }
                    } else {
                    const f32x2 p0 = a.rg * w[k], q0 = a.ba * w[k], p1 = b.rg * w[k], q1 = b.ba * w[k];
                    if (k == 0) { org0 = p0; oba0 = q0; org1 = p1; oba1 = q1; }
                    else { org0 = org0 + p0; oba0 = oba0 + q0; org1 = org1 + p1; oba1 = oba1 + q1; }
                    }
                }
                u32x4 codes;
                if constexpr (NOV > 0) {
                    cvs::px32x2 acc;
                    acc.r = f32x2{ org0.x, org1.x }; acc.g = f32x2{ org0.y, org1.y };
                    acc.b = f32x2{ oba0.x, oba1.x }; acc.a = f32x2{ oba0.y, oba1.y };
#pragma unroll
                    for (int l = 0; l < NOV; l++) acc = cvs::over_pair_uniform(acc, ov_cur.v[l]);
                    codes = cvs::narrow_pair_lean(acc);
                } else {
                    codes = u32x4{ cvs::f2h_rz2(org0.x, org0.y), cvs::f2h_rz2(oba0.x, oba0.y), cvs::f2h_rz2(org1.x, org1.y), cvs::f2h_rz2(oba1.x, oba1.y) };
                }
                __builtin_amdgcn_raw_buffer_store_b128(codes, row_rsrc(dst_data, (size_t)((ptrdiff_t)(t - bp.target.fy0) * (ptrdiff_t)trow + trect0), wrect), (int)toff, 0, 0);
            }
            ov_cur = ov_nxt;
            return true;
        };
        each_slot(step, std::make_integer_sequence<int, NT>{});
    }
}

template <int NT, int W, int NOV>
int launch(cvk_blur_params bp, int cus, hipStream_t s) {
    constexpr int C = NT / 2, D = C & 1, OUTW = 2 * W - 2 * C - 2 * D;
    const int cols = bp.tx1 - bp.tx0 + 1, rows = bp.ty1 - bp.ty0 + 1;
    const int strips = (cols + OUTW - 1) / OUTW;
    static std::atomic<int> cached{ 0 };            // (several threads may launch at once)
    int mine = cached.load(std::memory_order_relaxed);
    if (!mine) {
        mine = resident_per_cu(k_blur_pair<NT, W, NOV>, W);
        const char *e = CVS_DIAG_ENV("CVS_BLUR_PAIR_WGS");          // diagnostic build: workgroups per CU the segments are sized for
        if (e && atoi(e) > 0) mine = atoi(e);
        cached.store(mine, std::memory_order_relaxed);
    }
    const int nframes = bp.batch.n > 0 ? bp.batch.n : 1;
    if (bp.rows_per_wg <= 0) {
        int segs = (mine * cus) / (strips * nframes);
        if (segs < 1) segs = 1;
        int r = (rows + segs - 1) / segs;
        if (r < NT - 1) r = NT - 1;                 // halo rows cost at most as much as the rows produced
        if (r > rows) r = rows;
        bp.rows_per_wg = r;
    }
    dim3 grid((unsigned)strips, (unsigned)((rows + bp.rows_per_wg - 1) / bp.rows_per_wg), (unsigned)nframes);
    hipLaunchKernelGGL((k_blur_pair<NT, W, NOV>), grid, dim3(W), 0, s, bp);
    return (int)hipGetLastError();
}

template <int NT, int W>
int pick_layers(const cvk_blur_params *bp, int cus, hipStream_t s) {
    switch (bp->nover) {
    case 0: return launch<NT, W, 0>(*bp, cus, s);
    case 1: return launch<NT, W, 1>(*bp, cus, s);
    case 2: return launch<NT, W, 2>(*bp, cus, s);
    case 3: return launch<NT, W, 3>(*bp, cus, s);
    case 4: return launch<NT, W, 4>(*bp, cus, s);
    }
    return (int)hipErrorInvalidValue;
}

template <int W>
int pick(const cvk_blur_params *bp, int cus, hipStream_t s) {
    switch (bp->ntaps) {
    case 3: return pick_layers<3, W>(bp, cus, s);
    case 5: return pick_layers<5, W>(bp, cus, s);
    case 7: return pick_layers<7, W>(bp, cus, s);
    case 9: return pick_layers<9, W>(bp, cus, s);
    case 11: return pick_layers<11, W>(bp, cus, s);
    case 13: return pick_layers<13, W>(bp, cus, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace

// f16 in and out, 1:1, an odd tap count up to 13, and every pair of columns whole and on a 16-byte boundary in every buffer
extern "C" int cvk_blur_pair_supported(const cvk_blur_params *bp) {
    if (!(bp->in_half && bp->out_half) || (bp->step != 0 && bp->step != 1)) return 0;
    if (!(bp->ntaps & 1) || bp->ntaps < 3 || bp->ntaps > 13) return 0;      // (15 taps: 230 VGPRs, no faster than k_blur)
    if (bp->nover < 0 || bp->nover > CVK_BLUR_MAX_OVER) return 0;
    const int c = bp->ntaps / 2, d = c & 1;
    // every pair one aligned 16-byte access, none straddling an edge: even pitches, the rectangle and the window start on
    // pair boundaries of their buffers AND of the strips' pair grid, and both have even widths
    if ((bp->source.pitch | bp->target.pitch) & 1) return 0;
    if (((bp->tx0 - bp->target.fx0) | (bp->tx1 - bp->tx0 + 1)) & 1) return 0;
    if (((bp->sx0 - bp->source.fx0) | (bp->sx1 - bp->sx0 + 1) | (bp->tx0 - c - d - bp->sx0)) & 1) return 0;
    if (bp->sx1 < bp->sx0 || bp->sy1 < bp->sy0) return 0;
    const int n = bp->batch.n > 0 ? bp->batch.n : 1;
    if (n > CVK_FRAME_BATCH) return 0;
    for (int z = 0; z < n; z++) {
        const void *src = bp->batch.n ? bp->batch.source[z] : bp->source.data;
        const void *dst = bp->batch.n ? bp->batch.target[z] : bp->target.data;
        if (!aligned16(src) || !aligned16(dst)) return 0;
        for (int l = 0; l < bp->nover; l++)
            if (!aligned16(bp->batch.n ? bp->batch.over[z][l] : bp->over[l])) return 0;
    }
    return 1;
}

extern "C" int cvk_blur_pair(const cvk_blur_params *bp_in, int cus, void *stream) {
    if (bp_in->tx1 < bp_in->tx0 || bp_in->ty1 < bp_in->ty0) return 0;
    if (!cvk_blur_pair_supported(bp_in)) return (int)hipErrorInvalidValue;
    cvk_blur_params bp = *bp_in;
    bp.step = 1;
    bp.flags = 0;
#ifdef CVS_DIAG
    { const char *e = getenv("CVS_BLUR_PAIR_DROP"); if (e) bp.flags = atoi(e) << 8; }
#endif
    return pick<64>(&bp, cus, (hipStream_t)stream);
}
