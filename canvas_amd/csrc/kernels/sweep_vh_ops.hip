// sweep_vh_ops.hip -- separable FIR with per-line tap tables, VERTICAL pass first, both passes in one sweep down the frame.
//
// video_scale_bilinear_f32 (video_scale.c:231-286) runs the pass with the smaller factor first and the vertical one when
// the factors are equal (:252) -- the usual case -- each pass adding its products to a zero-filled f32 frame in ascending
// source order (:63-122 vertical, :161-226 horizontal).  sweep_ops.hip has the other order (rows filtered horizontally, then
// accumulated); the order of the two roundings differs, so this is a kernel of its own, on the same tables.
//
// Third form of this kernel.  The second kept one accumulator per target line in flight and walked the SOURCE rows with a
// record per row (which slots take the row, which lines end on it): 1.7 scalar instructions and 0.4 branches per vector
// instruction, VALU busy 6 % of the time (profiles/r02).  This one walks the TARGET lines and gathers, which is what the
// table says literally:
//   * one wave per workgroup owns 64 target columns (lane = column) or, for large targets, 128 (lane = two columns, PXL)
//     and the source columns under them (two or four tiles of the horizontal table's footprint list, at most NQ * 64
//     pixels: lane + 64 q is source pixel q of the lane);
//   * the source rows a line can reach live WIDENED in a register window win[0 .. W-1] (W = longest vertical list): a line
//     whose first tap is source row s needs win[k] = row s + k.  First taps never decrease down the table
//     (cvk_fir_axis.streamable), so the window only ever moves forward: shift by one, take the oldest row of a short ring
//     of rows requested ahead (pf[0 .. 2], storage format), request the next into the same registers.  Plain loads:
//     hipcc counts them (a counted vmcnt per row), tools/check_asm_loads.py checks them like every other load;
//   * V: mid = 0 + win[0] w0 + win[1] w1 ... in ascending source order, exactly the line's n taps (a chain per count: no
//     padded tap multiplies a row the line does not take, so Inf / NaN spread only where the reference spreads them);
//     weights, count and first row are ONE scalar load per line (cvk_fir_axis.lrec, built by the host), requested a line ahead;
//   * the line's mid row goes to ONE LDS row (the wave's own: LDS runs a wave's accesses in order, no barrier); each lane
//     gathers the horizontal taps of its column from it -- sum from 0.0f in ascending tap order, padded taps read a zero
//     pixel with weight 0 -- and stores its pixel.
// No accumulator slots, no per-row records, no register indexing, no hand-written waits.  Per line of 128 pixels the second
// form executed 455 scalar instructions and 105 branches; this one 47 and 14 (profiles/r03).
// Algorithmic bytes: source pixel once + target pixel once.
#include <atomic>
#include <climits>
#include "kernels.h"
#include "chain_math.hpp"
#include "sweep_common.hpp"
#include "gather_common.hpp"

#if defined(CVS_DIAG) && !defined(CVS_CONTRACT)
#define CVS_VH_PROBES 1
// timing probes (diagnostic build only): per workgroup, the shader clock at set-up start, after the tap lists and the first
// record, after the first row has landed in the window, at the first store, and at the end -> tools/vh_clocks.py
__device__ unsigned long long *g_vh_clocks;
extern "C" __attribute__((visibility("default"))) int cvk_fir_vh_clock_buffer(void *dev) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_vh_clocks), &dev, sizeof dev); }
#define CVS_VH_CLOCK(slot) do { if (g_vh_clocks && threadIdx.x == 0) g_vh_clocks[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CVS_VH_CLOCK(slot) do { } while (0)
#endif

namespace {

using cvs::f32x2;
typedef uint32_t nt_u4 __attribute__((ext_vector_type(4)));
typedef float nt_f4 __attribute__((ext_vector_type(4)));

constexpr int kLanes = 64;       // one wave per workgroup; a lane owns one or two target columns
constexpr int kPF = 3;           // source rows requested ahead of the window
static_assert(kLanes == 2 * CVK_FIR2D_TILE_X, "a strip is two or four tiles of the footprint table");

// PXL: target pixels per lane (1: a strip of 64 columns; 2: 128 columns, the lane owns the adjacent pair 2 lane, 2 lane + 1
// and stores 16 bytes of halfs at once -- half the waves, half the scalar work per pixel; for large targets)
// The leading arguments are what the set-up's first table reads need, as plain scalars: they arrive in SGPRs with the wave
// (-amdgpu-kernarg-preload-count, csrc/Makefile), so those reads do not wait for the fetch of the struct behind them.
// hpack: the horizontal table's packed lists when their width is MAXTH, else NULL.
constexpr int kVhFpOffset = 48;  // where cvk_fir2d_params starts in the kernel-argument segment: three pointers, six ints (checked in the kernel)
template <int W, int MAXTH, int NQ, bool INH, int PXL>
__global__ __launch_bounds__(kLanes) void k_fir_vh(const uint32_t *hpack, const int *hfoot, const uint32_t *vlrec, int rows_per_wg, int line0,
                                                   int tx0, int tx1, int ty0, int ty1, cvk_fir2d_params fp) {
    static_assert(W >= 1 && W <= 8 && MAXTH >= 1 && MAXTH <= 8 && NQ >= 1 && NQ <= 5 && (PXL == 1 || PXL == 2), "instances");
    constexpr int kZero = NQ * kLanes;                                   // the zero pixel behind the mid row
    constexpr int kStrip = kLanes * PXL;                                 // target columns per workgroup
    __shared__ float4 mid[kZero + 1];
    const int lane = threadIdx.x;
    CVS_VH_CLOCK(0);
#ifdef CVS_VH_PROBES
    if (g_vh_clocks && threadIdx.x == 0) g_vh_clocks[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = __builtin_amdgcn_s_memrealtime();     // the start again, on the clock all XCDs share
#endif
    // the lane's columns: halfs out -> the adjacent pair 2 lane, 2 lane + 1 (one 16-byte store); floats out -> lane and
    // lane + 64 (two 16-byte stores, each contiguous across the wave)
    constexpr bool out_half = INH;                                       // (both frames of a scaler call have the caller's format)
    constexpr int cstep = PXL == 2 && !out_half ? kLanes : 1;            // from the lane's first column to its second
    // a batch of frames of this geometry: grid.z picks the frame (its pointers read through the kernel-argument segment: a
    // scalar load at a computed offset; the struct is the kernel's first argument)
    typedef const cvk_fir2d_params __attribute__((address_space(4))) *kargs_t;
    const kargs_t ka = (kargs_t)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + kVhFpOffset);
    // (read whether there is a batch or not, so that the reads are requested with the rest of the arguments; the check: the
    // segment's layout is the compiler's -- were the struct not where kVhFpOffset says, no pointer from it may be used)
    const void *const zsrc = ka->frame_source[blockIdx.z];
    void *const ztgt = ka->frame_target[blockIdx.z];
    const int nframes = fp.nframes, chk_n = ka->nframes, chk_x = ka->tx1, chk_y = ka->ty1, chk_p = ka->source.pitch;
    const void *const one_src = fp.source.data;
    void *const one_tgt = fp.target.data;
    const int tcol = tx0 + (int)blockIdx.x * kStrip + (PXL == 2 && out_half ? 2 * lane : lane);
    const int nlines = ty1 - ty0 + 1;
    // target lines, counted from the vertical table's first (fp.ty0); the launch covers lines line0 .. nlines - 1
    const int ia = line0 + (int)blockIdx.y * rows_per_wg, ib = min(ia + rows_per_wg - 1, nlines - 1);
    const konst foot = as_konst(hfoot);
    // the packed tap lists of this lane's columns (when the table has them at this width): requested first, on the preloaded
    // arguments alone; decoded below, once the strip's first source column is known
    constexpr bool kPackable = MAXTH == 2 || MAXTH == 4;
    const bool packed = kPackable && hpack != nullptr;                   // (uniform)
    uint32_t hrec[PXL][kPackable ? 2 * MAXTH : 1];
    if (packed) {
#pragma unroll
        for (int p = 0; p < PXL; p++) {
            const int hline = min(tcol + p * cstep, tx1) - tx0;
            const uint4 *g = reinterpret_cast<const uint4 *>(hpack + (size_t)hline * 2 * MAXTH);
#pragma unroll
            for (int q = 0; q < MAXTH / 2; q++) { const uint4 x = g[q]; hrec[p][4 * q] = x.x; hrec[p][4 * q + 1] = x.y; hrec[p][4 * q + 2] = x.z; hrec[p][4 * q + 3] = x.w; }
        }
    }
    const int hstride = fp.h.stride;

    // source columns under the strip: its tiles of the footprint table (first > last: the tile touches nothing)
    constexpr int kTiles = kStrip / CVK_FIR2D_TILE_X;
    const int ntiles = (tx1 - tx0) / CVK_FIR2D_TILE_X + 1, t0 = kTiles * (int)blockIdx.x;
    // (every entry read whether the strip has that tile or not -- the list is padded to a multiple of four, kernels.h -- so
    // that the reads are one request, not one trip each behind a test)
    int flo[kTiles], fhi[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; t++) { flo[t] = (int)foot[2 * (t0 + t)]; fhi[t] = (int)foot[2 * (t0 + t) + 1]; }
    int sx_lo = INT_MAX, sx_hi = INT_MIN;
#pragma unroll
    for (int t = 0; t < kTiles; t++)
        if (t0 + t < ntiles && fhi[t] >= flo[t]) { sx_lo = min(sx_lo, flo[t]); sx_hi = max(sx_hi, fhi[t]); }
    if (sx_hi < sx_lo) sx_lo = sx_hi = fp.source.fx0;                    // no column of the strip has taps: any pixel will do
    const int npx = min(sx_hi - sx_lo + 1, NQ * kLanes);                 // (the host chose NQ to cover them)

    // the horizontal taps of this lane's columns: offsets into the mid row, weights; padded taps -> the zero pixel, weight 0
    bool col_live[PXL];
    int aoff[PXL][MAXTH];
    float wt[PXL][MAXTH];
    if (packed) {
        // short lists come packed (kernels.h cvk_fir_axis.pack): one aligned read per column, lanes reading consecutive
        // records, instead of 1 + 2 MAXTH scattered ones -- the set-up's first trip to memory is mostly this
#pragma unroll
        for (int p = 0; p < PXL; p++) {
            col_live[p] = tcol + p * cstep <= tx1;
#pragma unroll
            for (int k = 0; k < MAXTH; k++) {
                const bool live = col_live[p] && (int)hrec[p][k] != INT_MIN;
                aoff[p][k] = live ? min(max((int)hrec[p][k] - sx_lo, 0), kZero) : kZero;
                wt[p][k] = live ? __uint_as_float(hrec[p][kPackable ? MAXTH + k : 0]) : 0.0f;
            }
        }
    } else {
#pragma unroll
        for (int p = 0; p < PXL; p++) {
            col_live[p] = tcol + p * cstep <= tx1;
            const int hline = tcol + p * cstep - tx0;
            const int hn = col_live[p] ? min(fp.h.ntaps[hline], MAXTH) : 0;
#pragma unroll
            for (int k = 0; k < MAXTH; k++) {
                const bool live = k < hn;
                const int a = live ? fp.h.src[(size_t)hline * hstride + k] - sx_lo : kZero;
                aoff[p][k] = min(max(a, 0), kZero);
                wt[p][k] = live ? fp.h.taps[(size_t)hline * hstride + k] : 0.0f;
            }
        }
    }
    if (lane == 0) mid[kZero] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#ifdef CVS_VH_PROBES
    { float acc = 0.0f; _Pragma("unroll") for (int p = 0; p < PXL; p++) { _Pragma("unroll") for (int k = 0; k < MAXTH; k++) acc += wt[p][k] + (float)aoff[p][k]; } asm volatile("" :: "v"(acc)); }
#endif
    CVS_VH_CLOCK(1);                                                     // the tap lists have landed
    if (nframes > 1 && !(chk_n == nframes && chk_x == tx1 && chk_y == ty1 && chk_p == fp.source.pitch)) return;
    const void *const src_data = nframes > 1 ? zsrc : one_src;
    void *const dst_data = nframes > 1 ? ztgt : one_tgt;
    const bool all_live = tx0 + ((int)blockIdx.x + 1) * kStrip - 1 <= tx1;      // (uniform) every lane's columns exist

    constexpr uint32_t tpx = out_half ? 8 : 16;
    char *optr = reinterpret_cast<char *>(dst_data) + ((size_t)(tcol - fp.target.fx0)) * tpx
               + (size_t)(ty0 + ia - fp.target.fy0) * (size_t)fp.target.pitch * tpx;
    const size_t trow = (size_t)fp.target.pitch * tpx;

    // one record per line (cvk_fir_axis.lrec): count, first source row, weights -- one scalar load, requested a line ahead
    constexpr int LR = CVK_FIR_LREC;
    konst lrec = as_konst(vlrec) + (size_t)ia * LR;
    struct Line { int n, first; float w[W]; };
    auto load_line = [&]() __attribute__((always_inline)) {              // the record `lrec` points at; then on to the next
        Line l;
        l.n = (int)lrec[0];                                              // (<= W: the host picked the instance by the longest list)
        l.first = (int)lrec[1];
#pragma unroll
        for (int k = 0; k < W; k++) l.w[k] = __uint_as_float(lrec[2 + k]);
        lrec += LR;                                                      // (a spare record follows the table's last)
        return l;
    };

    // first and last source row the segment's lines reach (first taps and last taps never decrease down the table)
    // (the first and the last record requested at once; lines without taps at an end of the segment -- rare -- look further in)
    const konst lend = lrec + (size_t)(ib - ia) * LR;
    const int n_a = (int)lrec[0], f_a = (int)lrec[1], n_b = (int)lend[0], f_b = (int)lend[1];
    int s_lo = f_a, s_hi = f_b + min(n_b, W) - 1;
    if (n_a <= 0 || n_b <= 0) {
        s_lo = INT_MAX; s_hi = INT_MIN;
        konst r = lrec;
        for (int i = ia; i <= ib; i++, r += LR)                          // uniform, scalar loads; segments are short
            if ((int)r[0] > 0) { s_lo = (int)r[1]; break; }
        r = lend;
        for (int i = ib; i >= ia; i--, r -= LR)
            if ((int)r[0] > 0) { s_hi = (int)r[1] + min((int)r[0], W) - 1; break; }
    }
    const bool any_taps = s_lo <= s_hi;                                  // (uniform) else every line of the segment is zeros
    CVS_VH_CLOCK(2);                                                     // the segment's row range is known

    // rows are requested through a uniform row pointer + one 32-bit lane offset per unit (clamped: every load unconditional)
    constexpr int PXB = INH ? 8 : 16;
    uint32_t uoff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) uoff[q] = (uint32_t)(min(lane + q * kLanes, npx - 1) * PXB);
    const uint32_t rowb = (uint32_t)fp.source.pitch * PXB;
    const char *rp = reinterpret_cast<const char *>(src_data) + (size_t)(sx_lo - fp.source.fx0) * PXB
                   + (size_t)((any_taps ? s_lo : fp.source.fy0) - fp.source.fy0) * (size_t)rowb;
    int s_left = any_taps ? s_hi - s_lo : 0;                             // rows after the one `rp` points at

    Raw<INH> pf[kPF][NQ];
    Px win[W][NQ];
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
    for (int j = 0; j < W; j++) {
#pragma unroll
        for (int q = 0; q < NQ; q++) win[j][q] = Px{ f32x2{ 0.0f, 0.0f }, f32x2{ 0.0f, 0.0f } };
    }
    int win0 = any_taps ? s_lo - W : 0;                                  // source row of win[0] (rows before s_lo: never a tap)
    // The queue of requested rows is a ring, and the ring position is a place in the PROGRAM: the line loop below is
    // written out once per position, each copy naming the registers its row was requested into.  (Moving a requested row
    // from one register to another would have to wait for it to land -- the first form: vmcnt(0) in front of every request;
    // a position kept in a variable came back from hipcc as exactly those moves.)
    auto advance_from = [&](Raw<INH> (&oldest)[NQ]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j + 1 < W; j++) {
#pragma unroll
            for (int q = 0; q < NQ; q++) win[j][q] = win[j + 1][q];
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) win[W - 1][q] = widen(oldest[q]);
        request(oldest);
        win0++;
    };

    Line cur = load_line();
#ifdef CVS_VH_PROBES
    int probe_lines = 0;
#endif
    int left = ib - ia + 1;                                              // lines still to produce
    // lines until one needs the window moved (true) or the segment is done (false)
    auto run_lines = [&]() __attribute__((always_inline)) -> bool {
        for (;;) {
            if (win0 < cur.first) return true;                           // (a line without taps has first = INT_MIN)
            const Line nxt = load_line();
            Px m[NQ];
            // exactly the line's taps: the usual count first
#define CVK_VSUM(N) { _Pragma("unroll") for (int q = 0; q < NQ; q++) { Px col[W]; _Pragma("unroll") for (int j = 0; j < W; j++) col[j] = win[j][q]; m[q] = vsum<N, W>(col, cur.w); } }
            if (cur.n == W) CVK_VSUM(W)
            else if (W > 1 && cur.n == W - 1) CVK_VSUM((W > 1 ? W - 1 : 1))
            else if (W > 2 && cur.n == W - 2) CVK_VSUM((W > 2 ? W - 2 : 1))
            else if (W > 3 && cur.n == W - 3) CVK_VSUM((W > 3 ? W - 3 : 1))
            else if (W > 4 && cur.n == W - 4) CVK_VSUM((W > 4 ? W - 4 : 1))
            else if (W > 5 && cur.n == W - 5) CVK_VSUM((W > 5 ? W - 5 : 1))
            else if (W > 6 && cur.n == W - 6) CVK_VSUM((W > 6 ? W - 6 : 1))
            else if (cur.n == 1) CVK_VSUM(1)
            else {
#pragma unroll
                for (int q = 0; q < NQ; q++) m[q] = Px{ f32x2{ 0.0f, 0.0f }, f32x2{ 0.0f, 0.0f } };     // a line without taps is zeros
            }
#undef CVK_VSUM
#pragma unroll
            for (int q = 0; q < NQ; q++) mid[lane + q * kLanes] = make_float4(m[q].lo.x, m[q].lo.y, m[q].hi.x, m[q].hi.y);
            __builtin_amdgcn_wave_barrier();                             // (compiler fence; the hardware keeps a wave's LDS accesses in order)
            float4 t[PXL][MAXTH];
#pragma unroll
            for (int p = 0; p < PXL; p++) {
#pragma unroll
                for (int k = 0; k < MAXTH; k++) t[p][k] = mid[aoff[p][k]];
            }
            f32x2 hlo[PXL], hhi[PXL];
#pragma unroll
            for (int p = 0; p < PXL; p++) {
                const f32x2 w0 = { wt[p][0], wt[p][0] };                 // (0 + p0 is p0: see vsum)
                hlo[p] = f32x2{ t[p][0].x, t[p][0].y } * w0; hhi[p] = f32x2{ t[p][0].z, t[p][0].w } * w0;
#pragma unroll
                for (int k = 1; k < MAXTH; k++) {
                    const f32x2 wk = { wt[p][k], wt[p][k] };
                    hlo[p] = cvs::madd(f32x2{ t[p][k].x, t[p][k].y }, wk, hlo[p]);          // t += s * coeff (video_scale.c:82-85)
                    hhi[p] = cvs::madd(f32x2{ t[p][k].z, t[p][k].w }, wk, hhi[p]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if constexpr (out_half) {
                uint32_t h[PXL][2];
#pragma unroll
                for (int p = 0; p < PXL; p++) { const uint2 v = narrow4(hlo[p], hhi[p]); h[p][0] = v.x; h[p][1] = v.y; }
                if constexpr (PXL == 2) {
                    if (all_live || col_live[1]) __builtin_nontemporal_store(nt_u4{ h[0][0], h[0][1], h[1][0], h[1][1] }, reinterpret_cast<nt_u4 *>(optr));      // (streamed: nothing reads the target back)
                    else if (col_live[0]) *reinterpret_cast<uint2 *>(optr) = make_uint2(h[0][0], h[0][1]);
                } else {
                    if (all_live || col_live[0]) *reinterpret_cast<uint2 *>(optr) = make_uint2(h[0][0], h[0][1]);
                }
            } else {
#pragma unroll
                for (int p = 0; p < PXL; p++)
                    if (all_live || col_live[p]) {
                        // (streamed where the target is the large frame -- the two-columns-per-lane instances: 3..9 % on floats; on the
                        // reducing instances it cost 10 % at 0.75x: profiles/r04/vh_streamed_stores_ab.txt)
                        if constexpr (PXL == 2) __builtin_nontemporal_store(nt_f4{ hlo[p].x, hlo[p].y, hhi[p].x, hhi[p].y }, reinterpret_cast<nt_f4 *>(optr + 16 * kLanes * p));
                        else *reinterpret_cast<float4 *>(optr + 16 * kLanes * p) = make_float4(hlo[p].x, hlo[p].y, hhi[p].x, hhi[p].y);
                    }
            }
            optr += trow;
#ifdef CVS_VH_PROBES
            if (probe_lines == 0) CVS_VH_CLOCK(4);                       // the first line's store is issued
            if (probe_lines == 7) CVS_VH_CLOCK(5);                       // ... the eighth's
            probe_lines++;
#endif
            cur = nxt;
            if (--left == 0) return false;
        }
    };
    static_assert(kPF == 3, "three positions written out");
    CVS_VH_CLOCK(3);                                                     // set-up done: the line loop starts
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
    CVS_VH_CLOCK(6);                                                     // the last line's store is issued
}

template <int W, int MAXTH, int NQ, bool INH, int PXL>
int launch(const cvk_fir2d_params &fp, int line0, int cus, hipStream_t s) {
    constexpr int kStrip = kLanes * PXL;
    const int cols = fp.tx1 - fp.tx0 + 1, rows = fp.ty1 - fp.ty0 + 1 - line0;
    const int strips = (cols + kStrip - 1) / kStrip;
    static std::atomic<int> cached{ 0 };            // (several threads may launch at once: pull-queue workers)
    int per_cu = cached.load(std::memory_order_relaxed);
    if (!per_cu) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fir_vh<W, MAXTH, NQ, INH, PXL>, kLanes, 0) != hipSuccess || n < 1) n = 1;
        per_cu = n;
        cached.store(n, std::memory_order_relaxed);
    }
    // one round of resident workgroups over the frame; a segment re-reads the W - 1 source rows its first line reaches back to
    const int frames = fp.nframes > 1 ? fp.nframes : 1;
    int segs = (per_cu * (cus > 0 ? cus : 256)) / (strips * frames);
    if (segs < 1) segs = 1;
    int r = (rows + segs - 1) / segs;
    // measured (1080p -> 4K and 4K -> 1080p, profiles/r03): the wide strips want 16 lines, the narrow ones as many waves as the chip holds
    if (r < (PXL == 2 ? 16 : 8)) r = PXL == 2 ? 16 : 8;
    if (r > 256) r = 256;
    if (r > rows) r = rows;
    dim3 grid((unsigned)strips, (unsigned)((rows + r - 1) / r), (unsigned)frames);
    const uint32_t *hpack = (MAXTH == 2 || MAXTH == 4) && fp.h.pack != nullptr && fp.h.pack_width == MAXTH ? fp.h.pack : nullptr;
    hipLaunchKernelGGL((k_fir_vh<W, MAXTH, NQ, INH, PXL>), grid, dim3(kLanes), 0, s, hpack, fp.h.foot, fp.v.lrec, r, line0, fp.tx0, fp.tx1, fp.ty0, fp.ty1, fp);
    return (int)hipGetLastError();
}

// (longest vertical list = window rows, longest horizontal list, source pixels a lane holds per row, target pixels per lane);
// a call gets the first that covers it
typedef int (*launch_fn)(const cvk_fir2d_params &, int, int, hipStream_t);
struct Instance { int w, maxth, nq, pxl; launch_fn f16, f32; };
#define CVK_VH_INSTANCE(W, T, Q, P) { W, T, Q, P, launch<W, T, Q, true, P>, launch<W, T, Q, false, P> }
const Instance kInstances[] = {
    // enlarging (and 1 : 1 shifts): 128 columns per wave, at most ~66 (2x) .. 130 (1x) source pixels under them
    CVK_VH_INSTANCE(2, 2, 2, 2), CVK_VH_INSTANCE(3, 4, 2, 2), CVK_VH_INSTANCE(3, 4, 3, 2),
    // reducing: 64 columns per wave (smaller targets: more waves)
    CVK_VH_INSTANCE(3, 4, 2, 1), CVK_VH_INSTANCE(4, 4, 2, 1),      // down to ~0.6x
    CVK_VH_INSTANCE(3, 4, 3, 1), CVK_VH_INSTANCE(4, 4, 3, 1),      // 0.5x: three taps, 130 source columns under 64
    CVK_VH_INSTANCE(6, 8, 3, 1), CVK_VH_INSTANCE(8, 8, 4, 1),      // down to ~0.3x; below, the two launches
};

// source pixels under a strip of `tiles` tiles: at most the sum of their footprints
const Instance *pick(const cvk_fir2d_params *fp) {
    const int cols = fp->tx1 - fp->tx0 + 1;
    for (const Instance &in : kInstances) {
        const int tiles = kLanes * in.pxl / CVK_FIR2D_TILE_X;
        const int nq = (tiles * fp->max_sw + kLanes - 1) / kLanes;
        if (in.pxl == 2 && cols < 1024) continue;                  // small targets: more, narrower strips
        // two halfs-pixels per lane are ONE 16-byte store: the pair must sit on a 16-byte boundary in every row
        if (in.pxl == 2 && fp->out_half && ((((uintptr_t)fp->target.data) & 15u) || (fp->target.pitch & 1) || ((fp->tx0 - fp->target.fx0) & 1))) continue;
        if (fp->v.max_taps <= in.w && fp->h.max_taps <= in.maxth && nq <= in.nq) return &in;
    }
    return NULL;
}

}  // namespace

extern "C" int cvk_fir_vh_supported(const cvk_fir2d_params *fp) {
    return fp->in_half == fp->out_half && fp->v.streamable && fp->v.lrec != NULL && fp->v.max_taps >= 1 && fp->h.max_taps >= 1 && fp->max_sw >= 1 && pick(fp) != NULL;
}

// fp->ty0 is the vertical table's first line; lines fp->ty0 + line0 .. fp->ty1 are produced
extern "C" int cvk_fir_vh(const cvk_fir2d_params *fp, int line0, int cus, void *stream) {
    if (fp->tx1 < fp->tx0 || fp->ty1 < fp->ty0 + line0 || line0 < 0) return 0;
    if (!cvk_fir_vh_supported(fp)) return (int)hipErrorInvalidValue;
    const Instance *in = pick(fp);
    return fp->in_half ? in->f16(*fp, line0, cus, (hipStream_t)stream) : in->f32(*fp, line0, cus, (hipStream_t)stream);
}
