// sweep_vh_ops.hip -- separable FIR with per-line tap tables, VERTICAL pass first, both passes in one sweep down the frame.
//
// video_scale_bilinear_f32 (video_scale.c:231-286) runs the pass with the smaller factor first and the vertical one when
// the factors are equal (:252) -- the usual case -- each pass adding its products to a zero-filled f32 frame in ascending
// source order (:63-122 vertical, :161-226 horizontal).  sweep_ops.hip has the other order (rows filtered horizontally, then
// accumulated); the order of the two roundings differs, so this is a kernel of its own, on the same tables and records:
//   * one wave per workgroup owns 32 target columns (one tile of the horizontal table) and the source columns under them
//     (h.foot: at most 128); a lane owns channel pairs of source pixels: unit u = lane + 64 q is pair (u & 1) of pixel u / 2;
//   * V: per source row one coalesced load per unit (straight from memory into the multiply, no LDS); every accumulator
//     slot of the unit takes the row with its weight from the row's record (0 for the slots that do not: cvk_fir_axis.rec,
//     see sweep_ops.hip), so each vertical sum adds its taps in ascending source order;
//   * a line that ends on this row: its sums leave their slot through the GPR index, go to ONE LDS row (the wave's own,
//     LDS runs a wave's accesses in order: no barrier), and each lane gathers the horizontal taps of its target column and
//     pair from there -- sum from 0.0f in ascending tap order -- and stores.
// LDS is touched once per TARGET line, not per source row.  Padded horizontal taps read the zero pixel behind the row.
// Bound: VALU issue / latency at 2-4 waves per SIMD.  Algorithmic bytes: source pixel once + target pixel once.
#include <atomic>
#include <climits>
#include "kernels.h"
#include "chain_math.hpp"
#include "sweep_common.hpp"

namespace {

using cvs::f32x2;

constexpr int kCols = 32;        // target columns per workgroup
constexpr int kLanes = 64;       // one wave
constexpr int kPFD = 4;          // source rows in flight
constexpr int kRowPx = 128;      // source pixels under a strip at most (NQ <= 4 units per lane)
constexpr int kRowFl = (kRowPx + 1) * 4;

template <bool INH> struct Unit;
template <> struct Unit<true> { uint32_t v; };                                           // two halfs
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
template <> struct Unit<false> { u32x2v v; };                                            // two floats
__device__ __forceinline__ void asm_ld(Unit<true> &dst, const void *row, uint32_t voff) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst.v) : "v"(voff), "s"(row));
}
__device__ __forceinline__ void asm_ld(Unit<false> &dst, const void *row, uint32_t voff) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst.v) : "v"(voff), "s"(row));
}
__device__ __forceinline__ f32x2 widen(const Unit<true> &u) { return f32x2{ cvs::h2f(u.v & 0xFFFFu), cvs::h2f(u.v >> 16) }; }
__device__ __forceinline__ f32x2 widen(const Unit<false> &u) { return f32x2{ __uint_as_float(u.v.x), __uint_as_float(u.v.y) }; }

template <int MAXT, int NACC, int NQ, bool INH>
__global__ __launch_bounds__(kLanes) void k_fir_vh(cvk_fir2d_params fp, int rows_per_wg, int line0) {
    static_assert(NQ >= 1 && NQ <= 4 && NQ * kLanes <= 2 * kRowPx && (NACC == 8 || NACC == 16), "units per lane; one register vector of slots per unit");
    __shared__ __align__(16) float lds[kRowFl];     // the vertical sums of one target line, a zero pixel behind them
    __shared__ int seg[4];                          // first / last source row of the segment, "a line has no taps", first line with taps
    const int lane = threadIdx.x, pr = lane & 1;
    const int c0 = fp.tx0 + (int)blockIdx.x * kCols;
    const int tcol = c0 + (lane >> 1);
    const bool col_live = tcol <= fp.tx1;
    const int nlines = fp.ty1 - fp.ty0 + 1;
    // target lines, counted from the vertical table's first (fp.ty0); the launch covers lines line0 .. nlines - 1
    const int ia = line0 + (int)blockIdx.y * rows_per_wg, ib = min(ia + rows_per_wg - 1, nlines - 1);
    const int vstride = fp.v.stride, hstride = fp.h.stride;

    if (lane == 0) { seg[0] = INT_MAX; seg[1] = INT_MIN; seg[2] = 0; seg[3] = INT_MAX; }
    __syncthreads();
    for (int i = ia + lane; i <= ib; i += kLanes) {
        const int n = min(fp.v.ntaps[i], vstride);
        if (n > 0) {
            const int a = fp.v.src[(size_t)i * vstride];
            atomicMin(&seg[0], a);
            atomicMax(&seg[1], a + n - 1);
            atomicMin(&seg[3], i);
        } else seg[2] = 1;
    }
    static_assert(kCols == CVK_FIR2D_TILE_X, "a strip is one tile of the footprint table");
    const konst foot = as_konst(fp.h.foot);
    int sx_lo = (int)foot[2 * blockIdx.x], sx_hi = (int)foot[2 * blockIdx.x + 1];
    if (sx_hi < sx_lo) sx_lo = sx_hi = fp.source.fx0;                         // no column of the strip has taps: any pixel will do
    sx_lo = __builtin_amdgcn_readfirstlane(sx_lo);
    sx_hi = __builtin_amdgcn_readfirstlane(sx_hi);
    const int nu = 2 * min(sx_hi - sx_lo + 1, NQ * kLanes / 2);               // units under the strip (the host chose NQ to cover them)
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
    if (lane < 4) lds[kRowPx * 4 + lane] = 0.0f;
    __syncthreads();
    const int s_lo = __builtin_amdgcn_readfirstlane(seg[0]), s_hi = __builtin_amdgcn_readfirstlane(seg[1]);
    const bool some_empty = __builtin_amdgcn_readfirstlane(seg[2]) != 0;

    const size_t tpx = fp.out_half ? 8 : 16;
    char *tbase = reinterpret_cast<char *>(fp.target.data) + ((size_t)(tcol - fp.target.fx0)) * tpx + (size_t)pr * (tpx / 2);
    const size_t trow = (size_t)fp.target.pitch * tpx;
    const bool out_half = fp.out_half != 0;
    auto store_at = [&](char *o, f32x2 v) __attribute__((always_inline)) {
        if (!col_live) return;
        if (out_half) *reinterpret_cast<uint32_t *>(o) = cvs::f2h_rz2(v.x, v.y);
        else *reinterpret_cast<float2 *>(o) = make_float2(v.x, v.y);
    };
    if (some_empty) {                               // lines without taps are zeros (frame edges; rare)
        for (int i = ia; i <= ib; i++)
            if (fp.v.ntaps[i] <= 0) store_at(tbase + (size_t)(fp.ty0 + i - fp.target.fy0) * trow, f32x2{ 0.0f, 0.0f });
    }
    if (s_lo > s_hi) return;                        // uniform
    // the lines with taps are one run and end in ascending order: the stores go to consecutive target rows
    char *optr = tbase + (size_t)(fp.ty0 + __builtin_amdgcn_readfirstlane(seg[3]) - fp.target.fy0) * trow;

    typedef float accvec __attribute__((ext_vector_type(2 * NACC)));
    // one register vector per unit, as NAMED variables: an array of three or more is merged by hipcc into one vector wider
    // than the widest register tuple, and lands in scratch
    accvec acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
#define CVK_ACC(q) ((q) == 0 ? acc0 : (q) == 1 ? acc1 : (q) == 2 ? acc2 : acc3)      /* q: a constant after unrolling */

    // source units lane + 64 q of the strip's footprint (clamped to its last unit: every load unconditional), fetched a group
    // of four rows ahead by asm loads and a hand-written vmcnt(0) (see sweep_ops.hip)
    constexpr int UB = INH ? 4 : 8;
    uint32_t uoff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) uoff[q] = (uint32_t)(min(lane + q * kLanes, nu - 1) * UB);
    const uint32_t rowb = __builtin_amdgcn_readfirstlane((uint32_t)fp.source.pitch * (2 * UB));
    const char *rp;
    {
        const uint64_t a = reinterpret_cast<uint64_t>(fp.source.data) + (uint64_t)(sx_lo - fp.source.fx0) * (2 * UB) + (uint64_t)(s_lo - fp.source.fy0) * (uint64_t)rowb;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        rp = reinterpret_cast<const char *>(((uint64_t)hi << 32) | lo);
    }
    int s_next = s_lo;
    typedef Unit<INH> Group[kPFD][NQ];
    auto issue_group = [&](Group &g) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < kPFD; d++) {
#pragma unroll
            for (int q = 0; q < NQ; q++) asm_ld(g[d][q], rp, uoff[q]);
            const bool more = s_next < s_hi;                       // uniform; past the segment's last row the pointer stays
            rp += more ? rowb : 0u;
            s_next += more ? 1 : 0;
        }
    };
    auto wait_group = [&](Group &g) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#pragma unroll
        for (int d = 0; d < kPFD; d++) {
#pragma unroll
            for (int q = 0; q < NQ; q++) asm volatile("" : "+v"(g[d][q].v));
        }
    };

    struct Rec { uint32_t ends; int first_end; f32x2 w[NACC]; };
    konst rec_next = as_konst(fp.v.rec) + (ptrdiff_t)(s_lo - fp.v.rec_s0) * (2 * NACC + 4);
    auto load_rec = [&]() __attribute__((always_inline)) {
        Rec r;
        r.ends = rec_next[1]; r.first_end = (int)rec_next[2];
#pragma unroll
        for (int j = 0; j < NACC; j++) r.w[j] = f32x2{ __uint_as_float(rec_next[4 + 2 * j]), __uint_as_float(rec_next[5 + 2 * j]) };
        rec_next += 2 * NACC + 4;
        return r;
    };
    // one source row: every slot of every unit takes it; then the lines that end on it go through the horizontal pass
    auto step = [&](const Unit<INH> (&px)[NQ], const Rec &rec, Rec &rec_after) __attribute__((always_inline)) {
        f32x2 x[NQ];
        bool odd = false;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            x[q] = widen(px[q]);
            odd = odd || __builtin_amdgcn_class(x[q].x, 0x207) || __builtin_amdgcn_class(x[q].y, 0x207);     // NaN, -Inf, +Inf
        }
        rec_after = load_rec();
        // (weight 0 for the slots that do not take the row; Inf and NaN pixels through a pass of their own: sweep_ops.hip)
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const bool mine = __builtin_amdgcn_class(x[q].x, 0x207) || __builtin_amdgcn_class(x[q].y, 0x207);
            const f32x2 xp = mine ? f32x2{ 0.0f, 0.0f } : x[q];
            // all products, then all sums: a packed add right behind the packed multiply it depends on costs a hazard slot
            // (an s_nop per slot, and this kernel is short of scalar issue, not of registers)
            f32x2 p[NACC];
#pragma unroll
            for (int j = 0; j < NACC; j++) p[j] = xp * rec.w[j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NACC; j++) {
                const f32x2 t = f32x2{ CVK_ACC(q)[2 * j], CVK_ACC(q)[2 * j + 1] } + p[j];
                CVK_ACC(q)[2 * j] = t.x; CVK_ACC(q)[2 * j + 1] = t.y;
            }
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd) != 0, 0)) {
            cvs::rare_path();
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const bool mine = __builtin_amdgcn_class(x[q].x, 0x207) || __builtin_amdgcn_class(x[q].y, 0x207);
                const f32x2 xo = mine ? x[q] : f32x2{ 0.0f, 0.0f };
#pragma unroll
                for (int j = 0; j < NACC; j++) {
                    const f32x2 t = f32x2{ CVK_ACC(q)[2 * j], CVK_ACC(q)[2 * j + 1] } + f32x2{ mul_zero_wins(xo.x, rec.w[j].x), mul_zero_wins(xo.y, rec.w[j].x) };
                    CVK_ACC(q)[2 * j] = t.x; CVK_ACC(q)[2 * j + 1] = t.y;
                }
            }
        }
        if (rec.ends) {                                            // uniform
            int n_end = __builtin_popcount(rec.ends);
            int i = rec.first_end;
            do {
                const int e = 2 * (i & (NACC - 1));
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const f32x2 v = { CVK_ACC(q)[e], CVK_ACC(q)[e + 1] };
                    CVK_ACC(q)[e] = 0.0f; CVK_ACC(q)[e + 1] = 0.0f;
                    *reinterpret_cast<f32x2 *>(lds + 2 * (lane + q * kLanes)) = v;
                }
                __builtin_amdgcn_wave_barrier();                   // (compiler fence; the hardware keeps a wave's LDS accesses in order)
                if (i >= ia && i <= ib) {                          // uniform
                    f32x2 t[MAXT], h = { 0.0f, 0.0f };
#pragma unroll
                    for (int k = 0; k < MAXT; k++) t[k] = *reinterpret_cast<const f32x2 *>(lds + aoff[k]);
#pragma unroll
                    for (int k = 0; k < MAXT; k++) t[k] = t[k] * wt[k];
#pragma unroll
                    for (int k = 0; k < MAXT; k++) h = h + t[k];
                    store_at(optr, h);
                    optr += trow;
                }
                __builtin_amdgcn_wave_barrier();
                i++;
            } while (--n_end);
        }
    };
    static_assert(kPFD == 4, "four steps written out: the records alternate");
    Rec ra = load_rec(), rb;
    int s = s_lo;
    Group ga, gb;
    auto four_rows = [&](Group &cur, Group &nxt) __attribute__((always_inline)) -> bool {
        step(cur[0], ra, rb);
        if (++s > s_hi) return false;
        step(cur[1], rb, ra);
        if (++s > s_hi) return false;
        step(cur[2], ra, rb);
        if (++s > s_hi) return false;
        wait_group(nxt);
        step(cur[3], rb, ra);
        issue_group(cur);
        return ++s <= s_hi;
    };
    issue_group(ga);
    wait_group(ga);
    issue_group(gb);
    while (four_rows(ga, gb) && four_rows(gb, ga)) {}
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");             // nothing of this wave is in flight when it ends
#undef CVK_ACC
}

template <int MAXT, int NACC, int NQ, bool INH>
int launch(const cvk_fir2d_params &fp, int line0, int cus, hipStream_t s) {
    const int cols = fp.tx1 - fp.tx0 + 1, rows = fp.ty1 - fp.ty0 + 1 - line0;
    const int strips = (cols + kCols - 1) / kCols;
    static std::atomic<int> cached{ 0 };            // (several threads may launch at once: pull-queue workers)
    int per_cu = cached.load(std::memory_order_relaxed);
    if (!per_cu) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fir_vh<MAXT, NACC, NQ, INH>, kLanes, 0) != hipSuccess || n < 1) n = 1;
        per_cu = n;
        cached.store(n, std::memory_order_relaxed);
    }
    // one round of resident workgroups over the frame; a segment re-reads (but does not re-filter horizontally) the source
    // rows its first lines reach back to
    int segs = (per_cu * (cus > 0 ? cus : 256)) / strips;
    if (segs < 1) segs = 1;
    int r = (rows + segs - 1) / segs;
    if (r < 3 * fp.v.max_active) r = 3 * fp.v.max_active;
    if (r > 256) r = 256;
    if (r > rows) r = rows;
    dim3 grid((unsigned)strips, (unsigned)((rows + r - 1) / r));
    hipLaunchKernelGGL((k_fir_vh<MAXT, NACC, NQ, INH>), grid, dim3(kLanes), 0, s, fp, r, line0);
    return (int)hipGetLastError();
}

// (longest horizontal list, accumulator slots, units a lane holds per row); a call gets the first that covers it
struct Instance { int maxt, nacc, nq; int (*f16)(const cvk_fir2d_params &, int, int, hipStream_t); int (*f32)(const cvk_fir2d_params &, int, int, hipStream_t); };
#define CVK_VH_INSTANCE(T, A, Q) { T, A, Q, launch<T, A, Q, true>, launch<T, A, Q, false> }
const Instance kInstances[] = {
    CVK_VH_INSTANCE(4, 8, 1),  CVK_VH_INSTANCE(4, 16, 1),          // the triangle scaler enlarging up to 2x (and 1 : 1 shifts)
    CVK_VH_INSTANCE(8, 8, 2),  CVK_VH_INSTANCE(8, 16, 1),          // reducing down to ~0.55x; enlarging up to 4x
    CVK_VH_INSTANCE(4, 8, 3),  CVK_VH_INSTANCE(8, 8, 3),           // down to ~0.4x (0.5x: three taps, 67 source columns)
    CVK_VH_INSTANCE(8, 8, 4),                                      // down to ~0.3x; below, the two launches
};

const Instance *pick(const cvk_fir2d_params *fp) {
    const int nq = (2 * fp->max_sw + kLanes - 1) / kLanes;
    for (const Instance &in : kInstances)
        if (fp->h.max_taps <= in.maxt && fp->v.nacc == in.nacc && nq <= in.nq) return &in;
    return NULL;
}

}  // namespace

extern "C" int cvk_fir_vh_supported(const cvk_fir2d_params *fp) {
    return fp->v.rec != NULL && !fp->v.rec_zero_weight && (fp->v.nacc == 8 || fp->v.nacc == 16) &&
           fp->v.max_active >= 1 && fp->v.max_active <= fp->v.nacc && fp->h.max_taps >= 1 && fp->max_sw >= 1 && fp->max_sw <= kRowPx && pick(fp) != NULL;
}

// fp->ty0 is the vertical table's first line; lines fp->ty0 + line0 .. fp->ty1 are produced
extern "C" int cvk_fir_vh(const cvk_fir2d_params *fp, int line0, int cus, void *stream) {
    if (fp->tx1 < fp->tx0 || fp->ty1 < fp->ty0 + line0 || line0 < 0) return 0;
    if (!cvk_fir_vh_supported(fp)) return (int)hipErrorInvalidValue;
    const Instance *in = pick(fp);
    return fp->in_half ? in->f16(*fp, line0, cus, (hipStream_t)stream) : in->f32(*fp, line0, cus, (hipStream_t)stream);
}
