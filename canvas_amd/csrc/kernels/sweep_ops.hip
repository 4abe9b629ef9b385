// sweep_ops.hip -- separable FIR with per-line tap tables, both passes in one sweep down the frame, one lane per target
// column and CHANNEL PAIR.
//
// Third form of the general FIR path (after the LDS tiles of fir_ops.hip and the lane-per-pixel sweep of resample_ops.hip).
// What the lane-per-pixel sweep measured (profiles/r02/general_fir_attempts.txt): a 4K -> 1536 x 864 Lanczos has 24 waves
// of target columns; to fill 1 024 SIMDs it must cut the frame into 54 row segments (15 halo rows for 40 useful ones) and
// still runs 1.3 waves per SIMD, every wave paying its own chain of LDS, scalar-table and barrier latencies: 11 cycles per
// instruction.  The sums of a pixel's four channels are independent, so here a lane owns ONE channel of one target column:
//   * four times the waves for the same frame: segments three times as long (halo ~30 %) at ~4 waves per SIMD;
//   * H: the lane gathers its taps from the LDS row with ds_read_b32 -- the four lanes of a column read 16 contiguous
//     bytes, columns 16 / factor bytes apart: near conflict-free at any factor;  (LDS: 16 KiB per workgroup)
//   * V: the vertical table comes TURNED ROUND from the host (cvk_fir_axis.rec): per SOURCE line which accumulator slots
//     take it, with what weight, and which slots end there.  A line's slot is its index modulo the slot count, so the
//     record does not depend on where a segment starts and the kernel does no bookkeeping at all: two scalar loads per
//     source row (weights arrive in SGPRs and feed v_mul_f32 directly), one bit test per slot.
//     Slots of lines that began above the segment accumulate partial sums that are never stored; each slot is cleared
//     when its line ends, before the next line of the same slot begins (the host guarantees lines slot-count apart never
//     overlap: max_active <= nacc).
// Arithmetic: every sum starts at 0.0f, products and additions rounded separately, ascending source order -- bit for bit
// the two gather passes of video_scale.c:93-122,193-226 on the planners' tables.  Padded horizontal taps (lists shorter
// than MAXT) read the ZERO PIXEL kept behind the LDS row with weight 0: acc + 0 * 0 == acc.
// Bound: LDS gather rate / VALU issue (HBM traffic is source once + target once).  Algorithmic bytes: source pixel once +
// target pixel once.
#include <climits>
#include "kernels.h"
#include "chain_math.hpp"

namespace {

using cvs::f32x2;

constexpr int kCols = 64;       // target columns per workgroup
constexpr int kLanes = 128;     // = kCols x 2 channel pairs
constexpr int kPFD = 4;         // source rows in flight
constexpr int kRowPx = 512;     // source pixels under a strip at most; the LDS row is this long whatever the factor, so that
                                // the second row buffer and the zero pixel sit at compile-time offsets
constexpr int kRowFl = (kRowPx + 1) * 4;

// tables the kernel only reads, at wave-uniform addresses: through the constant address space, so that hipcc may use scalar
// loads (a plain global pointer next to the kernel's own stores gets vector loads: nothing tells it the two never alias)
typedef const __attribute__((address_space(4))) uint32_t *konst;
__device__ __forceinline__ konst as_konst(const void *p) { return (konst)(uintptr_t)p; }

// x * w with "0 * anything = 0" (measured on gfx950, tools/legacy_mul_test.hip: bit-equal to v_mul_f32 whenever neither
// operand is zero, +0 when either is).  The accumulator slots that do not take a source row get weight 0 from the host:
// acc + 0 is acc whatever the row holds, Inf and NaN included, so the slots need no branch.  A sum never is -0 (it starts
// at +0 and x + (-x) = +0), so the +0 this gives where v_mul_f32 gives -0 adds up to the same bits.
__device__ __forceinline__ float mul_zero_wins(float x, float w) {
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "s"(w), "v"(x));
    return r;
}

template <bool INH> struct SrcPx;
template <> struct SrcPx<true> { uint2 v; };        // rgba_f16
template <> struct SrcPx<false> { uint4 v; };       // rgba_f32

template <int MAXT, int NACC, int NQ, bool INH>
__global__ __launch_bounds__(kLanes) void k_fir_lanes(cvk_fir2d_params fp, int rows_per_wg) {
    static_assert(NQ * kLanes <= kRowPx, "a lane stages pixels lane + q * kLanes of the strip's footprint");
    __shared__ __align__(16) float lds[2 * kRowFl];     // two source rows, one zero pixel behind each
    __shared__ int seg[3];                          // first / last source row of the segment, "a line has no taps"
    const int lane = threadIdx.x, pr = lane & 1;
    const int c0 = fp.tx0 + (int)blockIdx.x * kCols, c1 = min(c0 + kCols - 1, fp.tx1);
    const int tcol = c0 + (lane >> 1);
    const bool col_live = tcol <= fp.tx1;
    const int nlines = fp.ty1 - fp.ty0 + 1;
    const int ia = (int)blockIdx.y * rows_per_wg, ib = min(ia + rows_per_wg - 1, nlines - 1);     // target lines, 0-based
    const int vstride = fp.v.stride, hstride = fp.h.stride;

    if (lane == 0) { seg[0] = INT_MAX; seg[1] = INT_MIN; seg[2] = 0; }
    __syncthreads();
    for (int i = ia + lane; i <= ib; i += kLanes) {
        const int n = min(fp.v.ntaps[i], vstride);
        if (n > 0) {
            const int a = fp.v.src[(size_t)i * vstride];
            atomicMin(&seg[0], a);
            atomicMax(&seg[1], a + n - 1);
        } else seg[2] = 1;
    }
    // source columns under the strip: union of the footprints of its 32-column tiles (host-built, first > last = empty)
    int sx_lo = INT_MAX, sx_hi = INT_MIN;
    for (int t = (c0 - fp.tx0) / CVK_FIR2D_TILE_X; t <= (c1 - fp.tx0) / CVK_FIR2D_TILE_X; t++) {
        const int lo = fp.h.foot[2 * t], hi = fp.h.foot[2 * t + 1];
        if (hi >= lo) { sx_lo = min(sx_lo, lo); sx_hi = max(sx_hi, hi); }
    }
    if (sx_hi < sx_lo) sx_lo = sx_hi = fp.source.fx0;                         // no column of the strip has taps: any pixel will do
    const int sw = min(sx_hi - sx_lo + 1, NQ * kLanes);                       // (the host chose NQ to cover every strip)
    // this lane's horizontal taps as float offsets into the LDS row.  Lists shorter than MAXT are padded with weight 0 on
    // the zero pixel: acc + 0 * 0 == acc (a padded tap on a real pixel would turn an Inf or NaN there into a NaN of the sum)
    const int hline = tcol - fp.tx0;
    const int hn = col_live ? min(fp.h.ntaps[hline], MAXT) : 0;
    int aoff[MAXT];
    float wt[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; k++) {
        const bool live = k < hn;
        aoff[k] = (live ? fp.h.src[(size_t)hline * hstride + k] - sx_lo : kRowPx) * 4 + 2 * pr;
        wt[k] = live ? fp.h.taps[(size_t)hline * hstride + k] : 0.0f;
    }
    if (lane < 8) lds[(size_t)(lane >> 2) * kRowFl + kRowPx * 4 + (lane & 3)] = 0.0f;
    __syncthreads();
    const int s_lo = __builtin_amdgcn_readfirstlane(seg[0]), s_hi = __builtin_amdgcn_readfirstlane(seg[1]);
    const bool some_empty = __builtin_amdgcn_readfirstlane(seg[2]) != 0;

    const size_t tpx = fp.out_half ? 8 : 16;
    char *tbase = reinterpret_cast<char *>(fp.target.data) + ((size_t)(tcol - fp.target.fx0)) * tpx + (size_t)pr * (tpx / 2);
    const size_t trow = (size_t)fp.target.pitch * tpx;
    const bool out_half = fp.out_half != 0;
    auto store_line = [&](int i, f32x2 v) {
        if (!col_live) return;
        char *o = tbase + (size_t)(fp.ty0 + i - fp.target.fy0) * trow;
        if (out_half) *reinterpret_cast<uint32_t *>(o) = cvs::f2h_rz2(v.x, v.y);
        else *reinterpret_cast<float2 *>(o) = make_float2(v.x, v.y);
    };
    if (some_empty) {                               // lines without taps are zeros (frame edges; rare)
        for (int i = ia; i <= ib; i++)
            if (fp.v.ntaps[i] <= 0) store_line(i, f32x2{ 0.0f, 0.0f });
    }
    if (s_lo > s_hi) return;                        // uniform

    f32x2 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = f32x2{ 0.0f, 0.0f };

    // staging: a lane fetches the pixels lane + q * kLanes of the strip's footprint (clamped to its last pixel: every load
    // is unconditional; what lands beyond the footprint in LDS is never read).  The row pointer is wave-uniform and moves
    // by one row per step; past the segment's last row it stays where it is.
    constexpr int PXB = INH ? 8 : 16;
    int loff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) loff[q] = min(lane + q * kLanes, sw - 1) * PXB;
    const size_t rowb = (size_t)fp.source.pitch * PXB;
    const char *rp = reinterpret_cast<const char *>(fp.source.data) + (size_t)(sx_lo - fp.source.fx0) * PXB + (size_t)(s_lo - fp.source.fy0) * rowb;
    int s_next = s_lo;
    SrcPx<INH> pf[kPFD][NQ];
    auto fetch_row = [&](SrcPx<INH> (&dst)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if constexpr (INH) dst[q].v = *reinterpret_cast<const uint2 *>(rp + loff[q]);
            else dst[q].v = *reinterpret_cast<const uint4 *>(rp + loff[q]);
        }
        if (s_next < s_hi) { rp += rowb; s_next++; }               // uniform
    };
    auto stage_row = [&](float *buf, const SrcPx<INH> (&src)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            float4 v;
            if constexpr (INH) v = make_float4(cvs::h2f(src[q].v.x & 0xFFFFu), cvs::h2f(src[q].v.x >> 16), cvs::h2f(src[q].v.y & 0xFFFFu), cvs::h2f(src[q].v.y >> 16));
            else v = make_float4(__uint_as_float(src[q].v.x), __uint_as_float(src[q].v.y), __uint_as_float(src[q].v.z), __uint_as_float(src[q].v.w));
            *reinterpret_cast<float4 *>(buf + 4 * (lane + q * kLanes)) = v;
        }
    };
#pragma unroll
    for (int d = 0; d < kPFD; d++) fetch_row(pf[d]);

    // a source row's record (cvk_fir_axis.rec): slots that end there, first line that ends there, weight per slot (0 for
    // the slots that do not take the row).  Scalar loads, a row ahead; the host keeps one spare record behind the last.
    struct Rec { uint32_t ends; int first_end; float w[NACC]; };
    konst rec_next = as_konst(fp.v.rec) + (ptrdiff_t)(s_lo - fp.v.rec_s0) * (NACC + 4);
    auto load_rec = [&]() {
        Rec r;
        r.ends = rec_next[1]; r.first_end = (int)rec_next[2];
#pragma unroll
        for (int j = 0; j < NACC; j++) r.w[j] = __uint_as_float(rec_next[4 + j]);
        rec_next += NACC + 4;
        return r;
    };
    auto filter_row = [&](const float *buf, const Rec &rec) {
        f32x2 h = { 0.0f, 0.0f };
#pragma unroll
        for (int k0 = 0; k0 < MAXT; k0 += 8) {                     // reads first, then products, then the sum in tap order
            constexpr int CH = MAXT - 0 < 8 ? MAXT : 8;
            f32x2 x[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) if (k0 + c < MAXT) x[c] = *reinterpret_cast<const f32x2 *>(buf + aoff[k0 + c]);
#pragma unroll
            for (int c = 0; c < CH; c++) if (k0 + c < MAXT) x[c] = x[c] * wt[k0 + c];
#pragma unroll
            for (int c = 0; c < CH; c++) if (k0 + c < MAXT) h = h + x[c];
        }
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = acc[j] + f32x2{ mul_zero_wins(h.x, rec.w[j]), mul_zero_wins(h.y, rec.w[j]) };
        if (rec.ends) {                                            // uniform
#pragma unroll
            for (int j = 0; j < NACC; j++) {
                if (rec.ends & (1u << j)) {
                    const int i = rec.first_end + ((j - rec.first_end) & (NACC - 1));
                    if (i >= ia && i <= ib) store_line(i, acc[j]);
                    acc[j] = f32x2{ 0.0f, 0.0f };
                }
            }
        }
    };
    static_assert(kPFD == 4, "the loop body below is four steps written out: row buffer and record alternate");
    Rec ra = load_rec(), rb;
    for (int sb = s_lo; sb <= s_hi; sb += kPFD) {                  // uniform bounds: every wave runs every iteration
        // step 0
        stage_row(lds, pf[0]); fetch_row(pf[0]);
        __syncthreads();
        rb = load_rec();
        filter_row(lds, ra);
        if (sb + 1 > s_hi) break;
        // step 1
        stage_row(lds + kRowFl, pf[1]); fetch_row(pf[1]);
        __syncthreads();
        ra = load_rec();
        filter_row(lds + kRowFl, rb);
        if (sb + 2 > s_hi) break;
        // step 2
        stage_row(lds, pf[2]); fetch_row(pf[2]);
        __syncthreads();
        rb = load_rec();
        filter_row(lds, ra);
        if (sb + 3 > s_hi) break;
        // step 3
        stage_row(lds + kRowFl, pf[3]); fetch_row(pf[3]);
        __syncthreads();
        ra = load_rec();
        filter_row(lds + kRowFl, rb);
    }
}

// rows per workgroup: one round of resident workgroups over the frame, but segments of at least three times the rows a
// source row feeds (a segment re-filters the source rows its first lines reach back to: about max_active target rows' worth)
template <int MAXT, int NACC, int NQ, bool INH>
int launch(const cvk_fir2d_params &fp, int cus, hipStream_t s) {
    const int cols = fp.tx1 - fp.tx0 + 1, rows = fp.ty1 - fp.ty0 + 1;
    const int strips = (cols + kCols - 1) / kCols;
    static int per_cu = 0;
    if (!per_cu) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fir_lanes<MAXT, NACC, NQ, INH>, kLanes, 0) != hipSuccess || n < 1) n = 1;
        per_cu = n;
    }
    int segs = (per_cu * (cus > 0 ? cus : 256)) / strips;
    if (segs < 1) segs = 1;
    int r = (rows + segs - 1) / segs;
    if (r < 3 * fp.v.max_active) r = 3 * fp.v.max_active;
    if (r > 256) r = 256;
    if (r > rows) r = rows;
    dim3 grid((unsigned)strips, (unsigned)((rows + r - 1) / r));
    hipLaunchKernelGGL((k_fir_lanes<MAXT, NACC, NQ, INH>), grid, dim3(kLanes), 0, s, fp, r);
    return (int)hipGetLastError();
}

// The instances: (longest horizontal list, accumulator slots, pixels a lane stages per row).  What resamplers and blurs need
// in practice, smallest first, and per slot count one that takes everything the host admits; a call gets the first that
// covers its lists and footprint.  The slot count must be the table's own: it is the stride of the records and the modulus
// of the line -> slot mapping.
struct Instance { int maxt, nacc, nq; int (*f16)(const cvk_fir2d_params &, int, hipStream_t); int (*f32)(const cvk_fir2d_params &, int, hipStream_t); };
#define CVK_LANES_INSTANCE(T, A, Q) { T, A, Q, launch<T, A, Q, true>, launch<T, A, Q, false> }
const Instance kInstances[] = {
    CVK_LANES_INSTANCE(8, 8, 1),   CVK_LANES_INSTANCE(8, 16, 1),  CVK_LANES_INSTANCE(8, 32, 1),      // enlarging (7 taps), short blurs
    CVK_LANES_INSTANCE(12, 8, 1),  CVK_LANES_INSTANCE(12, 8, 2),  CVK_LANES_INSTANCE(12, 16, 1),     // 0.5 < factor < 1
    CVK_LANES_INSTANCE(16, 8, 2),  CVK_LANES_INSTANCE(16, 16, 1),                                    // 0.4 <= factor <= 0.5
    CVK_LANES_INSTANCE(24, 8, 4),  CVK_LANES_INSTANCE(24, 32, 1),                                    // down to 0.26; blurs of 17..24
    CVK_LANES_INSTANCE(32, 8, 4),  CVK_LANES_INSTANCE(32, 32, 1),                                    // down to 0.19; blurs of 25..32
    CVK_LANES_INSTANCE(32, 16, 4), CVK_LANES_INSTANCE(32, 32, 4),
};

const Instance *pick(const cvk_fir2d_params *fp) {
    const int nq = (fp->h.foot64 + kLanes - 1) / kLanes;
    for (const Instance &in : kInstances)
        if (fp->h.max_taps <= in.maxt && fp->v.nacc == in.nacc && nq <= in.nq) return &in;
    return NULL;
}

}  // namespace

extern "C" int cvk_fir_lanes_supported(const cvk_fir2d_params *fp) {
    return fp->v.rec != NULL && !fp->v.rec_zero_weight && (fp->v.nacc == 8 || fp->v.nacc == 16 || fp->v.nacc == 32) &&
           fp->v.max_active >= 1 && fp->v.max_active <= fp->v.nacc && fp->h.max_taps >= 1 && fp->h.foot64 >= 1 && pick(fp) != NULL;
}

extern "C" int cvk_fir_lanes(const cvk_fir2d_params *fp, int cus, void *stream) {
    if (fp->tx1 < fp->tx0 || fp->ty1 < fp->ty0) return 0;
    if (!cvk_fir_lanes_supported(fp)) return (int)hipErrorInvalidValue;
    const Instance *in = pick(fp);
    return fp->in_half ? in->f16(*fp, cus, (hipStream_t)stream) : in->f32(*fp, cus, (hipStream_t)stream);
}
