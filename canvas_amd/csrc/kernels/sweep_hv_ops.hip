// sweep_hv_ops.hip -- separable FIR with per-line tap tables, HORIZONTAL pass first, both passes in one sweep down the frame:
// the Lanczos resampler at any factor (x pass, then y pass: DESIGN.md 4.2), blurs whose tap lists the register-window
// kernel has no instance for, and video_scale_bilinear_f32 when the horizontal factor is the smaller (video_scale.c:252).
//
// The per-target-line gather of sweep_vh_ops.hip in the other pass order.  There the window holds source rows and a line
// goes vertical sum -> LDS -> horizontal gather; here a source row goes LDS -> horizontal gather as it ENTERS the window,
// the window holds rows that are already filtered horizontally (one pixel per target column of the lane), and a line is
// the vertical sum of the window's first n rows, straight to the store:
//   * one wave per workgroup, 64 target columns (lane = column) or 128 (lane = two columns, PXL) and the source columns
//     under them (two or four tiles of the horizontal table's footprint list, at most NQ * 64 pixels);
//   * a line whose first tap is source row s needs hw[k] = H(row s + k); first taps never decrease down the table
//     (cvk_fir_axis.streamable), so the window only moves forward: shift by one, take the oldest of three rows requested
//     ahead (a ring whose position is a place in the program, as in sweep_vh_ops.hip), widen it into the wave's LDS row,
//     gather each column's horizontal taps from it -- sum in ascending tap order, padded taps read a zero pixel with
//     weight 0 -- and that is the window's newest row;
//   * V: hw[0] w0 + hw[1] w1 ... in ascending source order over exactly the line's n taps (a chain for the three usual
//     counts, a select per tap for the others: a row beyond the line's last tap may hold Inf or NaN and must not be
//     multiplied, not even by zero); one scalar record per line (cvk_fir_axis.lrec).
// It replaces the channel-pair sweep (sweep_ops.hip: one accumulator per target line in flight, a record per source row,
// register-indexed hand-over, hand-written loads) as the first choice for these tables; that kernel stays for tables this
// one has no instance for and as a second, independent implementation the tests pin.
// Algorithmic bytes: source pixel once + target pixel once.
#include <atomic>
#include <climits>
#include "kernels.h"
#include "chain_math.hpp"
#include "sweep_common.hpp"
#include "gather_common.hpp"

namespace {

constexpr int kLanes = 64;       // one wave per workgroup; a lane owns one or two target columns
constexpr int kPF = 3;           // source rows requested ahead of the window
static_assert(kLanes == 2 * CVK_FIR2D_TILE_X, "a strip is two or four tiles of the footprint table");

// horizontal sum of one target column from the wave's LDS row: MAXTH taps in groups of eight (products first, then the adds)
template <int MAXTH>
__device__ __forceinline__ Px hsum(const float4 *row, const int (&aoff)[MAXTH], const float (&wt)[MAXTH]) {
    Px h;
#pragma unroll
    for (int k0 = 0; k0 < MAXTH; k0 += 8) {
        float4 t[8];
#pragma unroll
        for (int c = 0; c < 8; c++) if (k0 + c < MAXTH) t[c] = row[aoff[k0 + c]];
#pragma unroll
        for (int c = 0; c < 8; c++) if (k0 + c < MAXTH) {
            const f32x2 wk = { wt[k0 + c], wt[k0 + c] };
            const f32x2 plo = f32x2{ t[c].x, t[c].y } * wk, phi = f32x2{ t[c].z, t[c].w } * wk;
            if (k0 + c == 0) { h.lo = plo; h.hi = phi; }             // (0 + p0 is p0: gather_common.hpp)
            else { h.lo = h.lo + plo; h.hi = h.hi + phi; }
        }
    }
    return h;
}

template <int WV, int MAXTH, int NQ, bool INH, int PXL>
__global__ __launch_bounds__(kLanes) void k_fir_hv(cvk_fir2d_params fp, int rows_per_wg) {
    static_assert(WV >= 1 && WV <= CVK_FIR_LREC - 2 && MAXTH >= 1 && MAXTH <= 32 && NQ >= 1 && NQ <= 6 && (PXL == 1 || PXL == 2), "instances");
    constexpr int kZero = NQ * kLanes;                                   // the zero pixel behind the source row
    constexpr int kStrip = kLanes * PXL;                                 // target columns per workgroup
    __shared__ float4 srow[kZero + 1];
    const int lane = threadIdx.x;
    const bool out_half = fp.out_half != 0;
    // the lane's columns: halfs out -> the adjacent pair 2 lane, 2 lane + 1 (one 16-byte store); floats out -> lane and lane + 64
    const int cstep = PXL == 2 && !out_half ? kLanes : 1;
    const int tcol = fp.tx0 + (int)blockIdx.x * kStrip + (PXL == 2 && out_half ? 2 * lane : lane);
    const int nlines = fp.ty1 - fp.ty0 + 1;
    const int ia = (int)blockIdx.y * rows_per_wg, ib = min(ia + rows_per_wg - 1, nlines - 1);      // target lines, 0-based
    const konst foot = as_konst(fp.h.foot);
    const int hstride = fp.h.stride;

    // source columns under the strip: its tiles of the footprint table (first > last: the tile touches nothing)
    constexpr int kTiles = kStrip / CVK_FIR2D_TILE_X;
    const int ntiles = (fp.tx1 - fp.tx0) / CVK_FIR2D_TILE_X + 1, t0 = kTiles * (int)blockIdx.x;
    int sx_lo = INT_MAX, sx_hi = INT_MIN;
#pragma unroll
    for (int t = 0; t < kTiles; t++) {
        if (t0 + t < ntiles) {
            const int lo1 = (int)foot[2 * (t0 + t)], hi1 = (int)foot[2 * (t0 + t) + 1];
            if (hi1 >= lo1) { sx_lo = min(sx_lo, lo1); sx_hi = max(sx_hi, hi1); }
        }
    }
    if (sx_hi < sx_lo) sx_lo = sx_hi = fp.source.fx0;                    // no column of the strip has taps: any pixel will do
    const int npx = min(sx_hi - sx_lo + 1, NQ * kLanes);                 // (the host chose NQ to cover them)

    // the horizontal taps of this lane's columns: offsets into the source row, weights; padded taps -> the zero pixel, weight 0
    bool col_live[PXL];
    int aoff[PXL][MAXTH];
    float wt[PXL][MAXTH];
#pragma unroll
    for (int p = 0; p < PXL; p++) {
        col_live[p] = tcol + p * cstep <= fp.tx1;
        const int hline = tcol + p * cstep - fp.tx0;
        const int hn = col_live[p] ? min(fp.h.ntaps[hline], MAXTH) : 0;
#pragma unroll
        for (int k = 0; k < MAXTH; k++) {
            const bool live = k < hn;
            const int a = live ? fp.h.src[(size_t)hline * hstride + k] - sx_lo : kZero;
            aoff[p][k] = min(max(a, 0), kZero);
            wt[p][k] = live ? fp.h.taps[(size_t)hline * hstride + k] : 0.0f;
        }
    }
    if (lane == 0) srow[kZero] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const bool all_live = fp.tx0 + ((int)blockIdx.x + 1) * kStrip - 1 <= fp.tx1;      // (uniform) every lane's columns exist

    const uint32_t tpx = out_half ? 8 : 16;
    char *optr = reinterpret_cast<char *>(fp.target.data) + ((size_t)(tcol - fp.target.fx0)) * tpx
               + (size_t)(fp.ty0 + ia - fp.target.fy0) * (size_t)fp.target.pitch * tpx;
    const size_t trow = (size_t)fp.target.pitch * tpx;

    // one record per line (cvk_fir_axis.lrec): count, first source row, weights -- one scalar load, requested a line ahead
    constexpr int LR = CVK_FIR_LREC;
    konst lrec = as_konst(fp.v.lrec) + (size_t)ia * LR;
    struct Line { int n, first; float w[WV]; };
    auto load_line = [&]() __attribute__((always_inline)) {
        Line l;
        l.n = (int)lrec[0];                                              // (<= WV: the host picked the instance by the longest list)
        l.first = (int)lrec[1];
#pragma unroll
        for (int k = 0; k < WV; k++) l.w[k] = __uint_as_float(lrec[2 + k]);
        lrec += LR;                                                      // (a spare record follows the table's last)
        return l;
    };

    // first and last source row the segment's lines reach (first taps and last taps never decrease down the table)
    int s_lo = INT_MAX, s_hi = INT_MIN;
    {
        konst r = lrec;
        for (int i = ia; i <= ib; i++, r += LR)
            if ((int)r[0] > 0) { s_lo = (int)r[1]; break; }
        r = lrec + (size_t)(ib - ia) * LR;
        for (int i = ib; i >= ia; i--, r -= LR)
            if ((int)r[0] > 0) { s_hi = (int)r[1] + min((int)r[0], WV) - 1; break; }
    }
    const bool any_taps = s_lo <= s_hi;

    constexpr int PXB = INH ? 8 : 16;
    uint32_t uoff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) uoff[q] = (uint32_t)(min(lane + q * kLanes, npx - 1) * PXB);
    const uint32_t rowb = (uint32_t)fp.source.pitch * PXB;
    const char *rp = reinterpret_cast<const char *>(fp.source.data) + (size_t)(sx_lo - fp.source.fx0) * PXB
                   + (size_t)((any_taps ? s_lo : fp.source.fy0) - fp.source.fy0) * (size_t)rowb;
    int s_left = any_taps ? s_hi - s_lo : 0;                             // rows after the one `rp` points at

    Raw<INH> pf[kPF][NQ];
    Px hw[WV][PXL];
    auto request = [&](Raw<INH> (&dst)[NQ]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; q++) dst[q].v = *reinterpret_cast<const decltype(dst[q].v) *>(rp + uoff[q]);
        const bool more = s_left > 0;                                    // uniform; past the segment's last row the pointer stays
        rp += more ? rowb : 0u;
        s_left -= more ? 1 : 0;
    };
#pragma unroll
    for (int d = 0; d < kPF; d++) request(pf[d]);
#pragma unroll
    for (int j = 0; j < WV; j++) {
#pragma unroll
        for (int p = 0; p < PXL; p++) hw[j][p] = Px{ f32x2{ 0.0f, 0.0f }, f32x2{ 0.0f, 0.0f } };
    }
    int win0 = any_taps ? s_lo - WV : 0;                                 // source row of hw[0] (rows before s_lo: never a tap)
    // the next source row enters: through the LDS row and the horizontal gather into the window's last place
    auto advance_from = [&](Raw<INH> (&oldest)[NQ]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const Px v = widen(oldest[q]);
            srow[lane + q * kLanes] = make_float4(v.lo.x, v.lo.y, v.hi.x, v.hi.y);
        }
        request(oldest);
        __builtin_amdgcn_wave_barrier();                                 // (compiler fence; the hardware keeps a wave's LDS accesses in order)
        Px h[PXL];
#pragma unroll
        for (int p = 0; p < PXL; p++) h[p] = hsum<MAXTH>(srow, aoff[p], wt[p]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j + 1 < WV; j++) {
#pragma unroll
            for (int p = 0; p < PXL; p++) hw[j][p] = hw[j + 1][p];
        }
#pragma unroll
        for (int p = 0; p < PXL; p++) hw[WV - 1][p] = h[p];
        win0++;
    };

    Line cur = load_line();
    int left = ib - ia + 1;                                              // lines still to produce
    // lines until one needs the window moved (true) or the segment is done (false)
    auto run_lines = [&]() __attribute__((always_inline)) -> bool {
        for (;;) {
            if (win0 < cur.first) return true;                           // (a line without taps has first = INT_MIN)
            const Line nxt = load_line();
            Px o[PXL];
#define CVK_VSUM(N) { _Pragma("unroll") for (int p = 0; p < PXL; p++) { Px col[WV]; _Pragma("unroll") for (int j = 0; j < WV; j++) col[j] = hw[j][p]; o[p] = vsum<N, WV>(col, cur.w); } }
            if (cur.n == WV) CVK_VSUM(WV)
            else if (WV > 1 && cur.n == WV - 1) CVK_VSUM((WV > 1 ? WV - 1 : 1))
            else if (WV > 2 && cur.n == WV - 2) CVK_VSUM((WV > 2 ? WV - 2 : 1))
            else {                                                       // any other count (0: a line without taps is zeros)
#pragma unroll
                for (int p = 0; p < PXL; p++) { Px col[WV]; _Pragma("unroll") for (int j = 0; j < WV; j++) col[j] = hw[j][p]; o[p] = vsum_any<WV>(col, cur.w, cur.n); }
            }
#undef CVK_VSUM
            if (out_half) {
                uint32_t h16[PXL][2];
#pragma unroll
                for (int p = 0; p < PXL; p++) { const uint2 v = narrow4(o[p].lo, o[p].hi); h16[p][0] = v.x; h16[p][1] = v.y; }
                if constexpr (PXL == 2) {
                    if (all_live || col_live[1]) *reinterpret_cast<uint4 *>(optr) = make_uint4(h16[0][0], h16[0][1], h16[1][0], h16[1][1]);
                    else if (col_live[0]) *reinterpret_cast<uint2 *>(optr) = make_uint2(h16[0][0], h16[0][1]);
                } else {
                    if (all_live || col_live[0]) *reinterpret_cast<uint2 *>(optr) = make_uint2(h16[0][0], h16[0][1]);
                }
            } else {
#pragma unroll
                for (int p = 0; p < PXL; p++)
                    if (all_live || col_live[p]) *reinterpret_cast<float4 *>(optr + 16 * kLanes * p) = make_float4(o[p].lo.x, o[p].lo.y, o[p].hi.x, o[p].hi.y);
            }
            optr += trow;
            cur = nxt;
            if (--left == 0) return false;
        }
    };
    static_assert(kPF == 3, "three positions written out");
    if (left > 0) {
        for (;;) {
            if (!run_lines()) break;
            advance_from(pf[0]);
            if (!run_lines()) break;
            advance_from(pf[1]);
            if (!run_lines()) break;
            advance_from(pf[2]);
        }
    }
}

template <int WV, int MAXTH, int NQ, bool INH, int PXL>
int launch(const cvk_fir2d_params &fp, int cus, hipStream_t s) {
    constexpr int kStrip = kLanes * PXL;
    const int cols = fp.tx1 - fp.tx0 + 1, rows = fp.ty1 - fp.ty0 + 1;
    const int strips = (cols + kStrip - 1) / kStrip;
    static std::atomic<int> cached{ 0 };            // (several threads may launch at once: pull-queue workers)
    int per_cu = cached.load(std::memory_order_relaxed);
    if (!per_cu) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fir_hv<WV, MAXTH, NQ, INH, PXL>, kLanes, 0) != hipSuccess || n < 1) n = 1;
        per_cu = n;
        cached.store(n, std::memory_order_relaxed);
    }
    // one round of resident workgroups over the frame; a segment re-filters the WV - 1 source rows its first line reaches
    // back to: never shorter than twice the target lines those rows are worth
    int segs = (per_cu * (cus > 0 ? cus : 256)) / strips;
    if (segs < 1) segs = 1;
    int r = (rows + segs - 1) / segs;
    const int lo = 2 * (fp.v.max_active > 0 ? fp.v.max_active : 1);
    if (r < lo) r = lo;
    if (r < 8) r = 8;
    if (r > 512) r = 512;
    if (r > rows) r = rows;
    dim3 grid((unsigned)strips, (unsigned)((rows + r - 1) / r));
    hipLaunchKernelGGL((k_fir_hv<WV, MAXTH, NQ, INH, PXL>), grid, dim3(kLanes), 0, s, fp, r);
    return (int)hipGetLastError();
}

// (longest vertical list = window rows, longest horizontal list, source pixels a lane holds per row, target pixels per lane)
typedef int (*launch_fn)(const cvk_fir2d_params &, int, hipStream_t);
struct Instance { int wv, maxth, nq, pxl; launch_fn f16, f32; };
#define CVK_HV_INSTANCE(W, T, Q, P) { W, T, Q, P, launch<W, T, Q, true, P>, launch<W, T, Q, false, P> }
const Instance kInstances[] = {
    // enlarging: Lanczos3 above 1x has 6-7 taps, the triangle 2-3; 128 columns per wave for large targets
    CVK_HV_INSTANCE(3, 4, 2, 2), CVK_HV_INSTANCE(8, 8, 2, 2), CVK_HV_INSTANCE(8, 8, 3, 2),
    CVK_HV_INSTANCE(4, 4, 2, 1), CVK_HV_INSTANCE(8, 8, 2, 1), CVK_HV_INSTANCE(8, 8, 3, 1),
    // reducing: Lanczos3 at 0.75x 8-9 taps, 0.5x 11-12, 0.4x 15-16, 0.33x 18-19, 0.25x 24
    CVK_HV_INSTANCE(10, 10, 2, 1), CVK_HV_INSTANCE(12, 12, 3, 1), CVK_HV_INSTANCE(16, 16, 3, 1), CVK_HV_INSTANCE(16, 16, 4, 1),
    CVK_HV_INSTANCE(20, 20, 4, 1), CVK_HV_INSTANCE(24, 24, 5, 1),
    // long blurs and mixed factors
    CVK_HV_INSTANCE(16, 8, 2, 1), CVK_HV_INSTANCE(8, 16, 4, 1), CVK_HV_INSTANCE(24, 8, 2, 1), CVK_HV_INSTANCE(8, 24, 5, 1),
};

const Instance *pick(const cvk_fir2d_params *fp) {
    const int cols = fp->tx1 - fp->tx0 + 1;
    for (const Instance &in : kInstances) {
        const int tiles = kLanes * in.pxl / CVK_FIR2D_TILE_X;
        const int nq = (tiles * fp->max_sw + kLanes - 1) / kLanes;
        if (in.pxl == 2 && cols < 1024) continue;                  // small targets: more, narrower strips
        if (in.pxl == 2 && fp->out_half && ((((uintptr_t)fp->target.data) & 15u) || (fp->target.pitch & 1) || ((fp->tx0 - fp->target.fx0) & 1))) continue;
        if (fp->v.max_taps <= in.wv && fp->h.max_taps <= in.maxth && nq <= in.nq) return &in;
    }
    return NULL;
}

}  // namespace

extern "C" int cvk_fir_hv_supported(const cvk_fir2d_params *fp) {
    return fp->v.streamable && fp->v.lrec != NULL && fp->v.max_taps >= 1 && fp->h.max_taps >= 1 && fp->max_sw >= 1 && pick(fp) != NULL;
}

extern "C" int cvk_fir_hv(const cvk_fir2d_params *fp, int cus, void *stream) {
    if (fp->tx1 < fp->tx0 || fp->ty1 < fp->ty0) return 0;
    if (!cvk_fir_hv_supported(fp)) return (int)hipErrorInvalidValue;
    const Instance *in = pick(fp);
    return fp->in_half ? in->f16(*fp, cus, (hipStream_t)stream) : in->f32(*fp, cus, (hipStream_t)stream);
}
