// sweep_ops.hip -- separable FIR with per-line tap tables (Lanczos at any factor, the triangle scaler when its horizontal
// pass comes first), both passes in one sweep down the frame: one lane per target column and CHANNEL PAIR, one wave per
// workgroup.
//
// Third form of the general FIR path (after the LDS tiles of fir_ops.hip and the lane-per-pixel sweep of resample_ops.hip).
// What the lane-per-pixel sweep measured (profiles/r02/general_fir_attempts.txt): a 4K -> 1536 x 864 Lanczos has 24 waves
// of target columns; to fill 1 024 SIMDs it must cut the frame into 54 row segments (15 halo rows for 40 useful ones) and
// still runs 1.3 waves per SIMD, every wave paying its own chain of LDS, table and barrier latencies: 11 cycles per
// instruction.  What this kernel does instead (each step measured: the same file):
//   * the sums of a pixel's channel pairs are independent: a lane owns one PAIR of one target column (packed f32 costs the
//     issue slot of scalar f32, so pairs halve the arithmetic per pixel; single channels would not) -- twice the waves for
//     the same frame, segments twice as long;
//   * a workgroup is ONE wave (32 columns): its two LDS rows are its own and LDS runs a wave's accesses in order, so a row
//     step has no barrier;
//   * H: the lane gathers its taps from the LDS row (ds_read_b64 at per-tap offsets held in registers, weights in
//     registers: both are constant down the sweep);
//   * V: the vertical table comes TURNED ROUND from the host (cvk_fir_axis.rec, scale.c axis_upload): per SOURCE line the
//     slots that end there, the first line that ends there and one weight per accumulator slot -- as a scalar register
//     pair, 0 for the slots that do not take the line.  A line's slot is its index modulo the slot count, so the record
//     does not depend on where a segment starts and the kernel keeps no books: two scalar loads per source row, a row
//     ahead.  Slots of lines that began above the segment accumulate partial sums that are never stored; a slot is cleared
//     when its line ends, before the next line of the same slot begins (max_active <= nacc, checked by the host);
//   * the row step is straight-line code: a taken branch costs a wave an instruction fetch (~64 cycles) that two to four
//     waves per SIMD do not hide.  The accumulators are a register VECTOR; the slot that ends is read and cleared through
//     the GPR index register; staging loads are unconditional (clamped lane offsets); pointers advance by scalar adds;
//   * source rows are fetched a group of four ahead by asm loads with a hand-written wait (instances with registers to
//     spare), 8-slot instances also gather the next row's taps before the accumulator pass of the current one.
// Arithmetic: every sum starts at 0.0f, products and additions rounded separately, ascending source order -- bit for bit
// the two gather passes of video_scale.c:93-122,193-226 on the planners' tables.  Padded horizontal taps (lists shorter
// than MAXT) read the ZERO PIXEL kept behind the LDS row with weight 0: acc + 0 * 0 == acc; a slot's weight 0 on an Inf or
// NaN row is handled by an additive pass with v_mul_legacy_f32 (see `filter`).
// Bound: LDS gather latency / VALU issue at 2-4 waves per SIMD (HBM traffic is source once + target once).
// Algorithmic bytes: source pixel once + target pixel once.
#include <atomic>
#include <climits>
#include <cstdlib>
#include "kernels.h"
#include "chain_math.hpp"
#include "sweep_common.hpp"

namespace {

using cvs::f32x2;

constexpr int kCols = 32;       // target columns per workgroup (one tile of the host's footprint table)
constexpr int kLanes = 64;      // = kCols x 2 channel pairs: ONE WAVE.  Its LDS rows are its own, LDS executes a wave's accesses
                                // in order, so a row step needs no barrier and no wave ever waits for another
constexpr int kPFD = 4;         // source rows in flight
constexpr int kRowPx = 256;     // source pixels under a strip at most; the LDS row is this long whatever the factor, so that
                                // the second row buffer and the zero pixel sit at compile-time offsets
constexpr int kRowFl = (kRowPx + 1) * 4;

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool INH> struct SrcPx;
template <> struct SrcPx<true> { u32x2 v; };        // rgba_f16
template <> struct SrcPx<false> { u32x4 v; };       // rgba_f32

// Source rows are fetched a GROUP of four rows ahead, from inline asm.  Written as plain loads hipcc keeps the four rows
// in flight in loop-carried registers and answers them with s_waitcnt vmcnt(0) at the loop header -- right behind the
// newest load, one full memory latency every fourth row (cdna_hip_programming.md 5.7; the first form of this kernel spent
// two thirds of its time there).  The compiler cannot see asm loads, so the wait is written by hand too: vmcnt(0) at the
// top of a group, when the loads it waits for are a whole group of rows old.
// address = 64-bit scalar row pointer + 32-bit lane offset (the global saddr form)
__device__ __forceinline__ void asm_ld(SrcPx<true> &dst, const void *row, uint32_t voff) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst.v) : "v"(voff), "s"(row));
}
__device__ __forceinline__ void asm_ld(SrcPx<false> &dst, const void *row, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst.v) : "v"(voff), "s"(row));
}

// HAND: whether this instance fetches its rows by the asm pipeline above.  Registers that asm loads are in flight into must
// stay where they are until the hand-written wait; under register pressure hipcc moves live values about (an instance with
// 32 slots copied a just-requested pixel to an AGPR and took the register for something else: wrong pixels, then a load
// landing on an address -- a fault).  So only instances with registers to spare take it (tests/test_abi_cpu.py reads their
// counts from the code object: no AGPRs, at most 224 VGPRs); the others use plain loads and leave the waiting to the compiler.
// (8 slots, lists up to 16: the instances that gather a row ahead and hold two rows of taps, see `filter`)
constexpr bool taps_row_ahead(int maxt, int nacc) { return nacc == 8 && maxt <= 16; }
constexpr bool hand_pipelined(int maxt, int nacc, int nq, bool inh) {
    return nacc <= 16 && (taps_row_ahead(maxt, nacc) ? 4 : 2) * maxt + 2 * nacc + 2 * kPFD * nq * (inh ? 2 : 4) + 60 <= 200;
}

template <int MAXT, int NACC, int NQ, bool INH, bool HAND>
__global__ __launch_bounds__(kLanes) void k_fir_lanes(cvk_fir2d_params fp, int rows_per_wg) {
    static_assert(NQ * kLanes <= kRowPx, "a lane stages pixels lane + q * kLanes of the strip's footprint");
    __shared__ __align__(16) float lds[2 * kRowFl];     // two source rows, one zero pixel behind each
    __shared__ int seg[4];                          // first / last source row of the segment, "a line has no taps", first line with taps
#ifdef CVS_DIAG
    const int skip = rows_per_wg >> 16;             // tools/: parts of the row step switched off (WRONG pixels: timing only)
    rows_per_wg &= 0xFFFF;
#else
    constexpr int skip = 0;
#endif
    const int lane = threadIdx.x, pr = lane & 1;
    const int c0 = fp.tx0 + (int)blockIdx.x * kCols;
    const int tcol = c0 + (lane >> 1);
    const bool col_live = tcol <= fp.tx1;
    const int nlines = fp.ty1 - fp.ty0 + 1;
    const int ia = (int)blockIdx.y * rows_per_wg, ib = min(ia + rows_per_wg - 1, nlines - 1);     // target lines, 0-based
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
    // source columns under the strip (host-built; first > last = empty)
    static_assert(kCols == CVK_FIR2D_TILE_X, "a strip is one tile of the footprint table");
    const konst foot = as_konst(fp.h.foot);
    int sx_lo = (int)foot[2 * blockIdx.x], sx_hi = (int)foot[2 * blockIdx.x + 1];
    if (sx_hi < sx_lo) sx_lo = sx_hi = fp.source.fx0;                         // no column of the strip has taps: any pixel will do
    sx_lo = __builtin_amdgcn_readfirstlane(sx_lo);                            // (the row pointer below must be in scalar registers)
    sx_hi = __builtin_amdgcn_readfirstlane(sx_hi);
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
    auto store_line = [&](int i, f32x2 v) __attribute__((always_inline)) {
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
    // The lines with taps are one run of consecutive lines (the host checked) and end in ascending order, so the stores of
    // this segment go to consecutive target rows: a pointer that moves down a row per store, no address arithmetic.
    char *optr = tbase + (size_t)(fp.ty0 + __builtin_amdgcn_readfirstlane(seg[3]) - fp.target.fy0) * trow;
    auto store_next = [&](f32x2 v) __attribute__((always_inline)) {
        if (col_live) {
            if (out_half) *reinterpret_cast<uint32_t *>(optr) = cvs::f2h_rz2(v.x, v.y);
            else *reinterpret_cast<float2 *>(optr) = make_float2(v.x, v.y);
        }
        optr += trow;
    };

    // The accumulators: slot j is elements 2j, 2j + 1 of a register vector (the widest register tuple is 32 wide: 16 slots;
    // a 32-slot instance has two).  The row step addresses the slots with constants; the line that ends on a row is in
    // a slot only known at run time -- wave-uniform, so hipcc reads and clears it through the GPR index register
    // (s_set_gpr_idx_on + v_mov), with no branch.  (A taken branch costs this kernel an instruction fetch, ~64 cycles with
    // two or three waves per SIMD: the first form chose the slot with a switch and spent a third of a 1.5x enlargement
    // in it, profiles/r02/general_fir_attempts.txt.)
    static_assert(NACC == 8 || NACC == 16 || NACC == 32, "register vectors of 32 floats at most: one, or two of 16 slots");
    constexpr int VS = NACC < 16 ? NACC : 16;          // slots per register vector
    typedef float accvec __attribute__((ext_vector_type(2 * VS)));
    accvec acc = 0.0f, acc_hi = 0.0f;                   // acc_hi: slots 16..31 of a 32-slot instance (enlargements 1.6x .. 2.2x)

    // staging: a lane fetches the pixels lane + q * kLanes of the strip's footprint (clamped to its last pixel: every load
    // is unconditional; what lands beyond the footprint in LDS is never read).  The row pointer is wave-uniform and moves
    // by one row per fetch; past the segment's last row it stays where it is (the same row again: valid memory, never used).
    constexpr int PXB = INH ? 8 : 16;
    uint32_t loff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) loff[q] = (uint32_t)(min(lane + q * kLanes, sw - 1) * PXB);
    const uint32_t rowb = __builtin_amdgcn_readfirstlane((uint32_t)fp.source.pitch * PXB);     // (the host admits rows below 4 GiB)
    const char *rp;
    {   // (the 64-bit product below is VALU work: back into scalar registers by hand, the asm loads take an SGPR pair)
        const uint64_t a = reinterpret_cast<uint64_t>(fp.source.data) + (uint64_t)(sx_lo - fp.source.fx0) * PXB + (uint64_t)(s_lo - fp.source.fy0) * (uint64_t)rowb;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        rp = reinterpret_cast<const char *>(((uint64_t)hi << 32) | lo);
    }
    int s_next = s_lo;
    typedef SrcPx<INH> Group[kPFD][NQ];
    auto issue_group = [&](Group &g) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < kPFD; d++) {
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                if constexpr (HAND) asm_ld(g[d][q], rp, loff[q]);
                else g[d][q].v = *reinterpret_cast<const decltype(g[d][q].v) *>(rp + loff[q]);
            }
            const bool more = s_next < s_hi;                       // uniform; scalar selects, scalar add
            rp += more ? rowb : 0u;
            s_next += more ? 1 : 0;
        }
    };
    auto wait_group = [&](Group &g) __attribute__((always_inline)) {
        if constexpr (!HAND) return;
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#pragma unroll
        for (int d = 0; d < kPFD; d++) {
#pragma unroll
            for (int q = 0; q < NQ; q++) asm volatile("" : "+v"(g[d][q].v));      // what reads g from here on reads it after the wait
        }
    };
    auto stage_row = [&](float *buf, const SrcPx<INH> (&src)[NQ]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            float4 v;
            if constexpr (INH) v = make_float4(cvs::h2f(src[q].v.x & 0xFFFFu), cvs::h2f(src[q].v.x >> 16), cvs::h2f(src[q].v.y & 0xFFFFu), cvs::h2f(src[q].v.y >> 16));
            else v = make_float4(__uint_as_float(src[q].v.x), __uint_as_float(src[q].v.y), __uint_as_float(src[q].v.z), __uint_as_float(src[q].v.w));
            *reinterpret_cast<float4 *>(buf + 4 * (lane + q * kLanes)) = v;
        }
    };

    // a source row's record (cvk_fir_axis.rec): slots that end there, first line that ends there, weight per slot (0 for
    // the slots that do not take the row).  Scalar loads, a row ahead; the host keeps one spare record behind the last.
    // (every weight comes twice: a scalar register pair is what the packed multiply takes; one copy would be moved to a
    // vector register pair first, sixteen moves a row)
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
    // With 8 slots and lists up to 16 (reductions down to 0.4x) a row step is software-pipelined over two rows: the taps of
    // row s + 1 are gathered from LDS (into `xn`) before the accumulator pass of row s, so that the LDS round trip (~200
    // cycles with two or three waves per SIMD) runs under it.  With 16 slots (enlargements) or longer lists the second set of
    // tap registers would cost a wave per SIMD, which costs more than the round trip: those gather and sum in the same step.
    constexpr bool AHEAD = taps_row_ahead(MAXT, NACC);
    typedef f32x2 Taps[MAXT];
    auto gather = [&](const float *buf, Taps &x) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < MAXT; k++) x[k] = *reinterpret_cast<const f32x2 *>(buf + aoff[k]);
    };
    // AHEAD: row s: horizontal sum of its taps `xc` (gathered a step ago) -> row s + 1 (`px`) into the LDS buffer `buf`, its
    // taps requested into `xn` -> every accumulator slot takes row s.
    // otherwise: row s (`px`) into `buf`, gathered into `xc`, summed, every slot takes it (`xn` unused).
    // (32 slots: two records of 64 weight registers do not fit the scalar file; the row's own record is loaded at the top of its
    // step instead of a row ahead)
    constexpr bool REC_AHEAD = NACC < 32;
    auto filter = [&](Taps &xc, Taps &xn, float *buf, const SrcPx<INH> (&px)[NQ], Rec &rec, Rec &rec_after) __attribute__((always_inline)) {
        if constexpr (!REC_AHEAD) rec = load_rec();
        if constexpr (!AHEAD) {
            if (!(skip & 8)) stage_row(buf, px);
            __builtin_amdgcn_wave_barrier();                       // (compiler fence; the hardware keeps a wave's LDS accesses in order)
            gather(buf, xc);
        }
        f32x2 h = { 0.0f, 0.0f };
        if (skip & 4) h = f32x2{ __uint_as_float(px[0].v.x), 1.0f };
        else {                                                     // products, then the sum in tap order
#pragma unroll
            for (int k = 0; k < MAXT; k++) xc[k] = xc[k] * wt[k];
#pragma unroll
            for (int k = 0; k < MAXT; k++) h = h + xc[k];
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (AHEAD) {
            if (!(skip & 8)) stage_row(buf, px);
            __builtin_amdgcn_wave_barrier();
            gather(buf, xn);
        }
        // the next row's record goes out here: behind the last LDS read (an outstanding scalar load makes every LDS wait a
        // wait for everything), a whole accumulator pass and the next row's staging before anyone needs it
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (REC_AHEAD) rec_after = load_rec();
        __builtin_amdgcn_sched_barrier(0);
        // Every slot takes the row; the slots it does not belong to have weight 0: 0 * h is a zero that leaves the sum as it
        // is -- unless h is Inf or NaN.  Lanes with such an h (rare) put 0 through the packed pass instead (adds an exact zero
        // to every slot) and, in a pass of their own that only runs when the wave has such a lane, the product that lets zero
        // win: a slot that takes the row gets acc + h * w with one rounding, as the gather would give it; the others acc + 0.
        const bool odd = __builtin_amdgcn_class(h.x, 0x207) || __builtin_amdgcn_class(h.y, 0x207);       // NaN, -Inf, +Inf
        const f32x2 h_plain = odd ? f32x2{ 0.0f, 0.0f } : h;
        auto add_lo = [&](int j, f32x2 p) __attribute__((always_inline)) {
            const f32x2 t = f32x2{ acc[2 * j], acc[2 * j + 1] } + p;
            acc[2 * j] = t.x; acc[2 * j + 1] = t.y;
        };
        auto add_hi = [&](int j, f32x2 p) __attribute__((always_inline)) {
            const f32x2 t = f32x2{ acc_hi[2 * j], acc_hi[2 * j + 1] } + p;
            acc_hi[2 * j] = t.x; acc_hi[2 * j + 1] = t.y;
        };
        if (skip & 2) add_lo(0, h);
        else {
#pragma unroll
            for (int j = 0; j < VS; j++) add_lo(j, h_plain * rec.w[j]);
            if constexpr (NACC > 16) {
#pragma unroll
                for (int j = 0; j < 16; j++) add_hi(j, h_plain * rec.w[16 + j]);
            }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd) != 0, 0)) {
                cvs::rare_path();
                const f32x2 h_odd = odd ? h : f32x2{ 0.0f, 0.0f };
#pragma unroll
                for (int j = 0; j < VS; j++) add_lo(j, f32x2{ mul_zero_wins(h_odd.x, rec.w[j].x), mul_zero_wins(h_odd.y, rec.w[j].x) });
                if constexpr (NACC > 16) {
#pragma unroll
                    for (int j = 0; j < 16; j++) add_hi(j, f32x2{ mul_zero_wins(h_odd.x, rec.w[16 + j].x), mul_zero_wins(h_odd.y, rec.w[16 + j].x) });
                }
            }
        }
    };
    // second half: the lines that end on this row are stored
    auto finish = [&](const Rec &rec) __attribute__((always_inline)) {
        if (rec.ends && !(skip & 1)) {                             // uniform
            // lines end in ascending order, a line's slot is its index & (NACC - 1): one or two lines here (more only on the
            // frame's last rows), each taken out of its slot through the GPR index and stored if it is the segment's own
            int n_end = __builtin_popcount(rec.ends);
            int i = rec.first_end;
            do {
                const int slot = i & (NACC - 1), e = 2 * (slot & 15);
                f32x2 v = { acc[e], acc[e + 1] };
                if constexpr (NACC <= 16) { acc[e] = 0.0f; acc[e + 1] = 0.0f; }
                else {                                             // both vectors read and rewritten, selects instead of a branch
                    const bool low = slot < 16;
                    const f32x2 vh = { acc_hi[e], acc_hi[e + 1] };
                    acc[e] = low ? 0.0f : v.x; acc[e + 1] = low ? 0.0f : v.y;
                    acc_hi[e] = low ? vh.x : 0.0f; acc_hi[e + 1] = low ? vh.y : 0.0f;
                    v = low ? v : vh;
                }
                if (i >= ia && i <= ib) store_next(v);
                i++;
            } while (--n_end);
        }
    };
    static_assert(kPFD == 4, "four steps written out: row buffer and record alternate");
    Rec ra, rb;
    if constexpr (REC_AHEAD) ra = load_rec();
    int s = s_lo;
    Group ga, gb;
    Taps xa, xb;
    // Four rows out of the group `cur` (AHEAD: its first row is in LDS and gathered already).
    // false: the segment's last row has been filtered (uniform)
    // Leaving in the middle of a group: the other group's loads are still in flight.  Wait for them HERE, before the exit
    // path joins anything else -- hipcc lays a loop's exits through blocks it shares with the loop body, and a
    // path-insensitive reading of the machine code (tools/check_asm_loads.py) must not find a way from "loads in flight"
    // to an instruction that touches their registers.
    auto leave = [&]() __attribute__((always_inline)) -> bool {
        if constexpr (HAND) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        return false;
    };
    auto four_rows = [&](Group &cur, Group &nxt) __attribute__((always_inline)) -> bool {
        if constexpr (AHEAD) {
            filter(xa, xb, lds + kRowFl, cur[1], ra, rb);
            finish(ra);
            if (++s > s_hi) return leave();
            filter(xb, xa, lds, cur[2], rb, ra);
            finish(rb);
            if (++s > s_hi) return leave();
            filter(xa, xb, lds + kRowFl, cur[3], ra, rb);
            finish(ra);
            if (++s > s_hi) return leave();
            wait_group(nxt);                                       // the row after this group's last is the next group's first
            issue_group(cur);                                      // (cur[3] went to LDS a step ago)
            filter(xb, xa, lds, nxt[0], rb, ra);
            finish(rb);
        } else {
            filter(xa, xa, lds, cur[0], ra, rb);
            finish(ra);
            if (++s > s_hi) return leave();
            filter(xa, xa, lds + kRowFl, cur[1], rb, ra);
            finish(rb);
            if (++s > s_hi) return leave();
            filter(xa, xa, lds, cur[2], ra, rb);
            finish(ra);
            if (++s > s_hi) return leave();
            filter(xa, xa, lds + kRowFl, cur[3], rb, ra);
            // the wait for the next group (vmcnt(0): it waits for this wave's stores too) between the halves of the last row:
            // the youngest store outstanding is then a row old (an enlargement stores on every row)
            wait_group(nxt);
            issue_group(cur);
            finish(rb);
        }
        if (++s > s_hi) return leave();
        return true;
    };
    issue_group(ga);
    wait_group(ga);
    issue_group(gb);
    if constexpr (AHEAD) {
        stage_row(lds, ga[0]);
        __builtin_amdgcn_wave_barrier();
        gather(lds, xa);
    }
    while (four_rows(ga, gb) && four_rows(gb, ga)) {}
    if constexpr (HAND) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");        // nothing of this wave is in flight when it ends
}

// rows per workgroup: one round of resident workgroups over the frame, but segments of at least three times the rows a
// source row feeds (a segment re-filters the source rows its first lines reach back to: about max_active target rows' worth)
template <int MAXT, int NACC, int NQ, bool INH>
int launch(const cvk_fir2d_params &fp, int cus, hipStream_t s) {
    constexpr bool HAND = hand_pipelined(MAXT, NACC, NQ, INH);
    const int cols = fp.tx1 - fp.tx0 + 1, rows = fp.ty1 - fp.ty0 + 1;
    const int strips = (cols + kCols - 1) / kCols;
    static std::atomic<int> cached{ 0 };            // (several threads may launch at once: pull-queue workers)
    int per_cu = cached.load(std::memory_order_relaxed);
    if (!per_cu) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fir_lanes<MAXT, NACC, NQ, INH, HAND>, kLanes, 0) != hipSuccess || n < 1) n = 1;
        per_cu = n;
        cached.store(n, std::memory_order_relaxed);
    }
    int segs = (per_cu * (cus > 0 ? cus : 256)) / strips;
    if (segs < 1) segs = 1;
    int r = (rows + segs - 1) / segs;
    if (r < 3 * fp.v.max_active) r = 3 * fp.v.max_active;
    if (r > 256) r = 256;
    if (r > rows) r = rows;
#ifdef CVS_DIAG
    if (const char *e = getenv("CVS_LANES_ROWS")) { const int v = atoi(e); if (v > 0 && v <= 4096) r = v < rows ? v : rows; }
#endif
    dim3 grid((unsigned)strips, (unsigned)((rows + r - 1) / r));
#ifdef CVS_DIAG
    if (const char *e = getenv("CVS_LANES_SKIP")) r |= atoi(e) << 16;
#endif
    hipLaunchKernelGGL((k_fir_lanes<MAXT, NACC, NQ, INH, HAND>), grid, dim3(kLanes), 0, s, fp, r);
    return (int)hipGetLastError();
}

// The instances: (longest horizontal list, accumulator slots, pixels a lane stages per row).  What resamplers and blurs need
// in practice, smallest first, and per slot count one that takes everything the host admits; a call gets the first that
// covers its lists and footprint.  The slot count must be the table's own: it is the stride of the records and the modulus
// of the line -> slot mapping.
struct Instance { int maxt, nacc, nq; int (*f16)(const cvk_fir2d_params &, int, hipStream_t); int (*f32)(const cvk_fir2d_params &, int, hipStream_t); };
#define CVK_LANES_INSTANCE(T, A, Q) { T, A, Q, launch<T, A, Q, true>, launch<T, A, Q, false> }
const Instance kInstances[] = {
    CVK_LANES_INSTANCE(8, 8, 1),   CVK_LANES_INSTANCE(8, 16, 1),                                     // short blurs
    CVK_LANES_INSTANCE(12, 8, 1),  CVK_LANES_INSTANCE(12, 8, 2),  CVK_LANES_INSTANCE(12, 16, 1),     // 0.5 < factor < 1; enlarging up to 1.6x
    CVK_LANES_INSTANCE(16, 8, 2),  CVK_LANES_INSTANCE(16, 16, 1),                                    // 0.4 <= factor <= 0.5
    CVK_LANES_INSTANCE(24, 8, 4),                                                                    // down to 0.26
    CVK_LANES_INSTANCE(32, 8, 4),  CVK_LANES_INSTANCE(32, 16, 4),                                    // down to 0.19, and whatever else fits
    CVK_LANES_INSTANCE(16, 32, 1),                                                                   // enlarging 1.6x .. 2.2x
};

const Instance *pick(const cvk_fir2d_params *fp) {
    const int nq = (fp->max_sw + kLanes - 1) / kLanes;            // max_sw: widest footprint of a 32-column tile
    for (const Instance &in : kInstances)
        if (fp->h.max_taps <= in.maxt && fp->v.nacc == in.nacc && nq <= in.nq) return &in;
    return NULL;
}

}  // namespace

extern "C" int cvk_fir_lanes_supported(const cvk_fir2d_params *fp) {
    return fp->v.rec != NULL && !fp->v.rec_zero_weight && (fp->v.nacc == 8 || fp->v.nacc == 16 || fp->v.nacc == 32) &&
           fp->v.max_active >= 1 && fp->v.max_active <= fp->v.nacc && fp->h.max_taps >= 1 && fp->max_sw >= 1 && fp->max_sw <= kRowPx && pick(fp) != NULL;
}

extern "C" int cvk_fir_lanes(const cvk_fir2d_params *fp, int cus, void *stream) {
    if (fp->tx1 < fp->tx0 || fp->ty1 < fp->ty0) return 0;
    if (!cvk_fir_lanes_supported(fp)) return (int)hipErrorInvalidValue;
    const Instance *in = pick(fp);
    return fp->in_half ? in->f16(*fp, cus, (hipStream_t)stream) : in->f32(*fp, cus, (hipStream_t)stream);
}
