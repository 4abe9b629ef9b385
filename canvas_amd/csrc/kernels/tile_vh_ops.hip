// tile_vh_ops.hip -- separable FIR with per-line tap tables, VERTICAL pass first, a workgroup per block of the target.
//
// The enlarging case of video_scale_bilinear_f32 (video_scale.c:231-286; pass order :252, vertical sums :63-122, horizontal
// sums :161-226).  sweep_vh_ops.hip walks a strip of the target line by line with one wave, every line a chain
// record -> window -> vertical sum -> LDS row -> horizontal sum -> store; when the target is the large frame the chip spends
// its time in those chains and in the set-up in front of them (profiles/r04/vh_clocks_*.txt).  Here a workgroup of four
// waves takes 128 columns x 16, 32 or 64 lines (a segment), and:
//   1. set-up, two trips to memory.  First trip, on the preloaded arguments alone: the packed tap lists of the lane's two
//      target columns (cvk_fir_axis.pack: one aligned read per column, lanes consecutive), the segment's line records,
//      the footprint entries of the 128 columns and the first and last line's record (row range), next to the fetch of the
//      remaining arguments.  Second trip: the source rows the segment's lines reach x the source columns under the 128
//      target columns, every load of a lane requested before the first is used; widened and written to S[row][column];
//   2. one barrier; then wave w takes lines w, w + 4, ... and each lane its two target columns: for each of a column's
//      horizontal taps the lane forms the vertical sum at that source column itself -- S[first .. first + n - 1][column] times
//      the line's weights in ascending order, exactly the line's n taps -- and adds the products in ascending tap order
//      (padded taps: the row's zero pixel, weight 0), narrows, stores: 1 KB per wave and line.  The vertical sum of a
//      source column is formed by every lane whose taps name it, to the same bits: arithmetic instead of an intermediate
//      row in LDS with a barrier per line group (the first form of this kernel; both are in profiles/r04).
// Same sums in the same order as k_fir_vh and as the two k_fir launches: the three are bit-equal (tests/test_gpu_parity.py).
// Algorithmic bytes: source pixel once + target pixel once (a segment re-reads the one or two source rows and columns its
// neighbours also reach: + 10..25 % of the SOURCE, which is the small frame here).
#include <atomic>
#include <climits>
#include <cstdlib>
#include <type_traits>
#include "kernels.h"
#include "chain_math.hpp"
#include "sweep_common.hpp"
#include "gather_common.hpp"

#if defined(CVS_DIAG) && !defined(CVS_CONTRACT)
// timing probes (diagnostic build only): per workgroup, the constant-rate clock at six points -> tools/vh_clocks.py tiles
__device__ unsigned long long *g_tvh_clocks;
extern "C" __attribute__((visibility("default"))) int cvk_fir_tvh_clock_buffer(void *dev) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tvh_clocks), &dev, sizeof dev); }
// (the buffer pointer is read once, at the kernel's start: a probe that fetched it again would time its own fetch)
// timing-only variants (wrong pixels on purpose; tools/time_scaler.py --diag-mode): 1 = stores only, in the kernel's own
// pattern, nothing read or computed; 2 = the whole set-up, then stores only; 3 = everything but the source rows' loads
__device__ int g_tvh_mode;
extern "C" __attribute__((visibility("default"))) int cvk_fir_tvh_diag_mode(int mode) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tvh_mode), &mode, sizeof mode); }
#define CVS_TVH_DIAG 1
#define CVS_TVH_CLOCK_START() unsigned long long *const tvh_clk = g_tvh_clocks ? g_tvh_clocks + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 : NULL
#define CVS_TVH_CLOCK(slot) do { if (tvh_clk && threadIdx.x == 0) { tvh_clk[slot] = __builtin_amdgcn_s_memtime(); tvh_clk[8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define CVS_TVH_CLOCK_START() do { } while (0)
#define CVS_TVH_CLOCK(slot) do { } while (0)
#endif

namespace {

using cvs::f32x2;

constexpr int kTW = 128, kThreads = 256, kWaves = kThreads / 64;
constexpr int kFpOffset = 56;    // where cvk_fir2d_params starts in k_fir_tile_vh's kernel-argument segment: three pointers, eight ints (checked in the kernel)
constexpr int kNJ = 9;           // source pixels a lane stages at most (all requested at once)
static_assert(kTW == 4 * CVK_FIR2D_TILE_X, "a tile spans four entries of the footprint table");

// "this value is needed here": keeps hipcc from sinking an LDS read into the branch that first uses it, where it would be
// requested and waited for on its own (a line's reads then cost one LDS trip each instead of one for all)
__device__ __forceinline__ void pin(Px &v) { asm volatile("" : "+v"(v.lo), "+v"(v.hi)); }           // (as the two register pairs the packed arithmetic takes)
__device__ __forceinline__ void pin(uint4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// narrow4 (gather_common.hpp) for the lane's two pixels at once: ONE test for all eight channels, the largest magnitude through
// two three-input maxima with |.| as operand modifiers (written as instructions: hipcc canonicalises each operand of fmaxf
// with an instruction of its own, which was 4 of the 62 vector instructions of a line)
__device__ __forceinline__ uint4 narrow8(f32x2 lo0, f32x2 hi0, f32x2 lo1, f32x2 hi1) {
    float big;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|\n\t"
        "v_max3_f32 %0, %0, |%4|, |%5|\n\t"
        "v_max3_f32 %0, %0, |%6|, |%7|\n\t"
        "v_max_f32 %0, %0, |%8|"
        : "=&v"(big) : "v"(lo0.x), "v"(lo0.y), "v"(hi0.x), "v"(hi0.y), "v"(lo1.x), "v"(lo1.y), "v"(hi1.x), "v"(hi1.y));
    // (a NaN operand is passed over by the maxima, as by fmaxf: a NaN channel needs no fix-up, the conversion keeps it)
    if (cvs::wave_any(!(big < 65536.0f))) {
        cvs::rare_path();
        return make_uint4(cvs::f2h_rz2(lo0.x, lo0.y), cvs::f2h_rz2(hi0.x, hi0.y), cvs::f2h_rz2(lo1.x, lo1.y), cvs::f2h_rz2(hi1.x, hi1.y));
    }
    return make_uint4(cvs::pkrtz(lo0.x, lo0.y), cvs::pkrtz(hi0.x, hi0.y), cvs::pkrtz(lo1.x, lo1.y), cvs::pkrtz(hi1.x, hi1.y));
}

// seg: target lines per workgroup; swp: pixels per LDS row (>= source columns under any 128 target columns + 1: the last
// is the row's zero pixel); shp: rows of S (>= source rows under any `seg` consecutive lines)
// The first arguments are what the set-up needs before anything else, as plain scalars: hipcc is told to have them preloaded
// into SGPRs (-amdgpu-kernarg-preload-count, csrc/Makefile), so the first table reads do not wait for a kernarg fetch.
template <int MAXTV, int MAXTH, bool INH>
__global__ __launch_bounds__(kThreads) void k_fir_tile_vh(const uint32_t *hpack, const int *hfoot, const uint32_t *vlrec, int line0, int seg, int tx0, int tx1, int ty0, int ty1,
                                                         int swp, int shp, cvk_fir2d_params fp) {
    static_assert(MAXTV >= 1 && MAXTV <= 4 && MAXTH >= 1 && MAXTH <= 8, "instances (a record in LDS holds four weights)");
    extern __shared__ __align__(16) unsigned char tile_lds[];
    typedef float4 raw_t;                                                // a source pixel in LDS: widened once, when it is staged
    uint4 *R = reinterpret_cast<uint4 *>(tile_lds);                      // [seg][2]  the segment's line records: count, first row, four weights
    raw_t *S = reinterpret_cast<raw_t *>(R + 2 * seg);                   // [shp][swp]  the segment's source rows, widened
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CVS_TVH_DIAG
    if (g_tvh_mode == 1) {                                               // stores only
        const int nl = ty1 - ty0 + 1, a = line0 + (int)blockIdx.y * seg, b = min(a + seg - 1, nl - 1);
        const int col = tx0 + (int)blockIdx.x * kTW + (INH ? 2 * lane : lane);
        char *o = reinterpret_cast<char *>(fp.target.data) + ((size_t)(col - fp.target.fx0)) * (INH ? 8 : 16);
        for (int line = a + wave; line <= b; line += kWaves) {
            char *q = o + (size_t)(ty0 + line - fp.target.fy0) * (size_t)fp.target.pitch * (INH ? 8 : 16);
            if (INH) { if (col + 1 <= tx1) *reinterpret_cast<uint4 *>(q) = make_uint4(line, lane, 0, 0); }
            else { if (col <= tx1) *reinterpret_cast<uint4 *>(q) = make_uint4(line, lane, 0, 0); if (col + 64 <= tx1) *reinterpret_cast<uint4 *>(q + 1024) = make_uint4(line, lane, 0, 0); }
        }
        return;
    }
#endif
    CVS_TVH_CLOCK_START();
    CVS_TVH_CLOCK(0);
    CVS_TVH_CLOCK(6);                                                    // (a probe right after a probe: what a probe costs)
    constexpr bool out_half = INH;                                       // (both frames of a scaler call have the caller's format)
    constexpr int cstep = out_half ? 1 : 64;                             // halfs: the pair 2 lane, 2 lane + 1; floats: lane, lane + 64
    const int tcol = tx0 + (int)blockIdx.x * kTW + (out_half ? 2 * lane : lane);
    const int nlines = ty1 - ty0 + 1;
    // table lines of this segment (counted from fp.ty0); the launch covers lines line0 .. nlines - 1
    const int ia = line0 + (int)blockIdx.y * seg, ib = min(ia + seg - 1, nlines - 1);
    const int tl = ib - ia + 1;
    const konst foot = as_konst(hfoot);
    const int zcol = swp - 1;                                            // the zero pixel of every S row

    // --- set-up in two trips to memory.  First: what depends on the preloaded arguments only -- the tap lists, the segment's
    // records, the footprint entries and the segment's first and last record (scalar) -- next to the fetch of the remaining
    // arguments.  Second: the source rows.  No load waits for another inside a trip.
    constexpr int LR = CVK_FIR_LREC;
    // the horizontal taps of this lane's two columns: one aligned read each of the packed list (kernels.h cvk_fir_axis.pack:
    // MAXTH source columns, MAXTH weights; past the column's count: INT_MIN, 0) -- lanes read consecutive records
    bool col_live[2];
    int hsrc[2][MAXTH];
    float hw[2][MAXTH];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        col_live[p] = tcol + p * cstep <= tx1;
        const int hline = min(tcol + p * cstep, tx1) - tx0;
        if constexpr (MAXTH == 2) {
            const uint4 r = *reinterpret_cast<const uint4 *>(hpack + (size_t)hline * 4);
            hsrc[p][0] = (int)r.x; hsrc[p][1] = (int)r.y; hw[p][0] = __uint_as_float(r.z); hw[p][1] = __uint_as_float(r.w);
        } else {
            static_assert(MAXTH == 4, "packed lists are two or four wide");
            const uint4 a = *reinterpret_cast<const uint4 *>(hpack + (size_t)hline * 8), b = *reinterpret_cast<const uint4 *>(hpack + (size_t)hline * 8 + 4);
            hsrc[p][0] = (int)a.x; hsrc[p][1] = (int)a.y; hsrc[p][2] = (int)a.z; hsrc[p][3] = (int)a.w;
            hw[p][0] = __uint_as_float(b.x); hw[p][1] = __uint_as_float(b.y); hw[p][2] = __uint_as_float(b.z); hw[p][3] = __uint_as_float(b.w);
        }
    }
    // the segment's line records go to LDS with the rows (lane = line), so that the line loop never goes to memory for one
    // (a record fetched per tile was a trip to L2 per tile: profiles/r04)
    uint4 my_rec[2] = { make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0) };
    {                                                                    // (lanes past the segment's last line: its last, not kept)
        const uint4 *g = reinterpret_cast<const uint4 *>(vlrec + (size_t)(ia + min(tid, tl - 1)) * LR);
        my_rec[0] = g[0];
        if (MAXTV > 2) my_rec[1] = g[1];
    }
    // source columns under the 128 target columns: four entries of the footprint list in one scalar load (the list is padded
    // to a multiple of four; first > last: that entry touches nothing)
    constexpr int kTiles = kTW / CVK_FIR2D_TILE_X;
    const int ntiles = (tx1 - tx0) / CVK_FIR2D_TILE_X + 1, t0 = kTiles * (int)blockIdx.x;
    int flo[kTiles], fhi[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; t++) { flo[t] = (int)foot[2 * (t0 + t)]; fhi[t] = (int)foot[2 * (t0 + t) + 1]; }
    // source rows the segment's lines reach: the first record's first tap .. the last record's last (first and last taps never
    // decrease down the table); both records requested at once
    const konst lrec = as_konst(vlrec) + (size_t)ia * LR;
    const konst lend = lrec + (size_t)(tl - 1) * LR;
    const int n_a = (int)lrec[0], f_a = (int)lrec[1], n_b = (int)lend[0], f_b = (int)lend[1];
    __builtin_amdgcn_sched_barrier(0);                                   // (the table reads are requested before the arguments are fetched)
    // ... and every other argument the kernel uses.  The empty asm makes all of it one request-and-wait: left alone, hipcc
    // sinks each of these loads to its first use, and every one then costs its own trip.
    // (a batch of frames: grid.z picks the frame; its pointers are read through the kernel-argument segment -- a scalar load
    // at a computed offset -- whether there is a batch or not, so that the read is part of this trip)
    typedef const cvk_fir2d_params __attribute__((address_space(4))) *kargs_t;
    const kargs_t ka = (kargs_t)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + kFpOffset);
    const int z = (int)blockIdx.z;
    const void *const zsrc = ka->frame_source[z];
    void *const ztgt = ka->frame_target[z];
    const int nframes = fp.nframes;
    // the segment's layout is the compiler's: were the struct not where kFpOffset says, no pointer from it may be used
    const int chk_n = ka->nframes, chk_x = ka->tx1, chk_y = ka->ty1, chk_p = ka->source.pitch;
    const void *const one_src = fp.source.data;
    void *const one_tgt = fp.target.data;
    const int spitch = fp.source.pitch, sfx0 = fp.source.fx0, sfy0 = fp.source.fy0, tpitch = fp.target.pitch, tfx0 = fp.target.fx0, tfy0 = fp.target.fy0;
    const int sfx1 = fp.source.fx1, tfx1 = fp.target.fx1;               // (fetched with their neighbours: named here so that their registers are not handed out while the fetch is in flight, which costs a wait)
    asm volatile("" :: "s"(n_a), "s"(f_a), "s"(n_b), "s"(f_b), "s"(flo[0]), "s"(fhi[0]), "s"(flo[1]), "s"(fhi[1]), "s"(flo[2]), "s"(fhi[2]), "s"(flo[3]), "s"(fhi[3]),
                 "s"(one_src), "s"(one_tgt), "s"(spitch), "s"(sfx0), "s"(sfy0), "s"(tpitch), "s"(tfx0), "s"(tfy0), "s"(sfx1), "s"(tfx1),
                 "s"(zsrc), "s"(ztgt), "s"(nframes), "s"(chk_n), "s"(chk_x), "s"(chk_y), "s"(chk_p));
    if (nframes > 1 && !(chk_n == nframes && chk_x == tx1 && chk_y == ty1 && chk_p == spitch)) return;
    const char *const sdata = reinterpret_cast<const char *>(nframes > 1 ? zsrc : one_src);
    char *const tdata = reinterpret_cast<char *>(nframes > 1 ? ztgt : one_tgt);

    int sx_lo = INT_MAX, sx_hi = INT_MIN;
#pragma unroll
    for (int t = 0; t < kTiles; t++)
        if (t0 + t < ntiles && fhi[t] >= flo[t]) { sx_lo = min(sx_lo, flo[t]); sx_hi = max(sx_hi, fhi[t]); }
    const int sw = sx_hi >= sx_lo ? min(sx_hi - sx_lo + 1, swp - 1) : 0;  // (the host sized swp to cover them)
    int s_lo = f_a, s_hi = f_b + min(n_b, MAXTV) - 1;
    if (n_a <= 0 || n_b <= 0) {                                          // rare: lines without taps at an end of the segment -> look further in
        s_lo = INT_MAX; s_hi = INT_MIN;
        konst r = lrec;
        for (int i = 0; i < tl; i++, r += LR)
            if ((int)r[0] > 0) { s_lo = (int)r[1]; break; }
        r = lend;
        for (int i = tl - 1; i >= 0; i--, r -= LR)
            if ((int)r[0] > 0) { s_hi = (int)r[1] + min((int)r[0], MAXTV) - 1; break; }
    }
    const int sh = s_hi >= s_lo ? min(s_hi - s_lo + 1, shp) : 0;
    CVS_TVH_CLOCK(1);                                                    // column and row range known

    // 1. stage the segment's rows (requested before the tap lists and records are looked at: same trip): pixel
    //    u = row * sw + column of the block, lane tid takes u = tid, tid + 256, ... (at most kNJ of them: the host checked);
    //    all of a lane's loads are requested before the first is used
#ifdef CVS_TVH_DIAG
    const bool skip_rows = g_tvh_mode == 3;
#else
    constexpr bool skip_rows = false;
#endif
    if (sw > 0 && sh > 0 && !skip_rows) {
        constexpr int PXB = INH ? 8 : 16;
        const int units = min(sh * sw, kNJ * kThreads);
        const char *base = sdata + ((size_t)(s_lo - sfy0) * (size_t)spitch + (size_t)(sx_lo - sfx0)) * PXB;
        const uint32_t rowpx = (uint32_t)spitch;
        // (row, column) of pixel u = tid, then + 256 at a time: the quotient and remainder of 256 by sw are uniform, a step is
        // two additions and a carry (a division per pixel was a quarter of the set-up's vector instructions)
        const float inv_sw = 1.0f / (float)sw;
        int dq = (int)(((float)kThreads + 0.5f) * inv_sw), dm = kThreads - dq * sw;      // (uniform) 256 = dq * sw + dm
        if (dm < 0) { dq--; dm += sw; } else if (dm >= sw) { dq++; dm -= sw; }
        int r = (int)(((float)tid + 0.5f) * inv_sw), c = tid - r * sw;                   // tid / sw, up to one either way
        if (c < 0) { r--; c += sw; } else if (c >= sw) { r++; c -= sw; }
        Raw<INH> v[kNJ];
        int sidx[kNJ];
#pragma unroll
        for (int j = 0; j < kNJ; j++) sidx[j] = -1;
#pragma unroll
        for (int j = 0; j < kNJ; j++) {
            if (j * kThreads >= units) break;                            // (uniform) no lane has a j-th pixel
            const bool mine = tid + j * kThreads < units;
            const int rr = mine ? r : sh - 1, cc = mine ? c : sw - 1;    // clamped: the load itself is unconditional
            sidx[j] = mine ? rr * swp + cc : -1;
            v[j].v = *reinterpret_cast<const decltype(v[j].v) *>(base + ((uint32_t)rr * rowpx + (uint32_t)cc) * (uint32_t)PXB);
            c += dm; r += dq;
            if (c >= sw) { c -= sw; r++; }
        }
#pragma unroll
        for (int j = 0; j < kNJ; j++)
            if (sidx[j] >= 0) { const Px w = widen(v[j]); S[sidx[j]] = make_float4(w.lo.x, w.lo.y, w.hi.x, w.hi.y); }
    }
    // columns of the S row and weights; padded taps (and columns past the target's last) -> the row's zero pixel, weight 0
    int acol[2][MAXTH];
    float wt[2][MAXTH];
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int k = 0; k < MAXTH; k++) {
            const bool live = col_live[p] && hsrc[p][k] != INT_MIN;
            acol[p][k] = live ? min(max(hsrc[p][k] - sx_lo, 0), zcol) : zcol;
            wt[p][k] = live ? hw[p][k] : 0.0f;
        }
    }
    if (tid < shp) S[tid * swp + zcol] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (tid < tl) { R[2 * tid] = my_rec[0]; R[2 * tid + 1] = my_rec[1]; }
    CVS_TVH_CLOCK(2);                                                    // tap lists and records landed

    CVS_TVH_CLOCK(3);                                                    // this wave's share of the rows is in LDS
    __syncthreads();
    CVS_TVH_CLOCK(4);

    const bool all_live = tx0 + ((int)blockIdx.x + 1) * kTW - 1 <= tx1;       // (uniform) every lane's columns exist
    constexpr uint32_t tpx = out_half ? 8 : 16;
    const size_t trow = (size_t)tpitch * tpx;
    char *obase = tdata + ((size_t)(tcol - tfx0)) * tpx + (size_t)(ty0 + ia - tfy0) * trow;

    // one record per line (cvk_fir_axis.lrec: count, first source row, weights), from LDS: every lane reads the same address
    struct Rec { int n, row; float w[MAXTV]; };
    struct RawRec { uint4 a, b; };
    auto read_rec = [&](int line) __attribute__((always_inline)) {      // lines past the segment's last: its last (never used)
        const int at = 2 * min(line, tl - 1);
        RawRec r;
        r.a = R[at];
        r.b = MAXTV > 2 ? R[at + 1] : make_uint4(0, 0, 0, 0);
        return r;
    };
    auto decode = [&](const RawRec &r) __attribute__((always_inline)) {
        Rec rec;
        rec.n = min((int)__builtin_amdgcn_readfirstlane(r.a.x), MAXTV);
        rec.row = rec.n > 0 ? ((int)__builtin_amdgcn_readfirstlane(r.a.y) - s_lo) * swp : 0;   // (a line without taps reads row 0 and keeps nothing)
        rec.w[0] = __uint_as_float(__builtin_amdgcn_readfirstlane(r.a.z));
        if constexpr (MAXTV > 1) rec.w[1] = __uint_as_float(__builtin_amdgcn_readfirstlane(r.a.w));
        if constexpr (MAXTV > 2) rec.w[2] = __uint_as_float(__builtin_amdgcn_readfirstlane(r.b.x));
        if constexpr (MAXTV > 3) rec.w[3] = __uint_as_float(__builtin_amdgcn_readfirstlane(r.b.y));
        return rec;
    };

    // 2. the lines: wave w takes lines w, w + 4, ... of the segment, each lane its two target columns.  For each of a column's
    //    horizontal taps the lane forms the vertical sum at that source column itself -- S[first .. first + n - 1][column] times
    //    the line's weights, ascending, exactly the line's n taps (video_scale.c:82-85) -- and adds the products in ascending
    //    tap order (:176-179; padded taps: the row's zero pixel, weight 0).  The same vertical sum is formed by every lane
    //    whose taps name that source column, to the same bits; what that costs is arithmetic, of which there is plenty,
    //    and what it saves is the intermediate row in LDS with its barrier per line group.  All reads of a line (and the
    //    next line's record) are requested before the first is used: every address is valid whatever the counts are, the
    //    counts only decide what is added.
    constexpr bool kBothAtOnce = MAXTH * MAXTV <= 4;                     // registers: both pixels' reads in flight, or one pixel's
    RawRec first_rec = read_rec(wave);
    Rec rc = decode(first_rec);
    // (uniform) every lane's second pixel starts on the source column its first pixel ends on: that vertical sum is formed once
    const bool share = MAXTH == 2 && !cvs::wave_any(acol[1][0] != acol[0][1]);
#ifdef CVS_TVH_DIAG
    if (g_tvh_mode == 2) {                                               // the whole set-up, then stores only
        for (int line = wave; line < tl; line += kWaves) {
            char *q = obase + (size_t)line * trow;
            if (out_half) { if (all_live || col_live[1]) *reinterpret_cast<uint4 *>(q) = make_uint4(line, lane, 0, 0); }
            else { if (all_live || col_live[0]) *reinterpret_cast<uint4 *>(q) = make_uint4(line, lane, 0, 0); if (all_live || col_live[1]) *reinterpret_cast<uint4 *>(q + 1024) = make_uint4(line, lane, 0, 0); }
        }
        return;
    }
#endif
    for (int line = wave; line < tl; line += kWaves) {
        const Rec cur = rc;
        const int n = cur.n;                                             // uniform
        RawRec nxt = read_rec(line + kWaves);
        Px sp[kBothAtOnce ? 2 : 1][MAXTH][MAXTV];                       // (one pixel's at a time when they are many)
        // `shared`: the second pixel's first tap is the source column of the first pixel's second (every lane of the wave:
        // a 2 : 1 enlargement away from the edges) -- that vertical sum is formed once
        auto request = [&](int p, auto shared) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < MAXTH; j++) {
                if (decltype(shared)::value && p == 1 && j == 0) continue;
#pragma unroll
                for (int k = 0; k < MAXTV; k++) {
                    const float4 v = S[cur.row + min(k, max(n - 1, 0)) * swp + acol[p][j]];
                    sp[kBothAtOnce ? p : 0][j][k] = Px{ f32x2{ v.x, v.y }, f32x2{ v.z, v.w } };
                }
            }
        };
        auto landed = [&](int p, auto shared) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < MAXTH; j++) {
                if (decltype(shared)::value && p == 1 && j == 0) continue;
#pragma unroll
                for (int k = 0; k < MAXTV; k++) pin(sp[kBothAtOnce ? p : 0][j][k]);
            }
        };
        f32x2 hlo[2], hhi[2];
        // the vertical sum at each tap's source column -- exactly the line's taps, ascending -- times the tap's weight, added
        // in ascending tap order.  The usual line has all MAXTV taps: its chain has no test in it (wave-uniform choice).
        auto sums = [&](auto full, auto shared) __attribute__((always_inline)) {
            request(0, shared);
            if constexpr (kBothAtOnce) request(1, shared);
            Px keep = { f32x2{ 0.0f, 0.0f }, f32x2{ 0.0f, 0.0f } };
#pragma unroll
            for (int p = 0; p < 2; p++) {
                if (p == 1 && !kBothAtOnce) request(1, shared);
                if (p == 0 || !kBothAtOnce) landed(p, shared);
                if (p == 0 && kBothAtOnce) landed(1, shared);
#pragma unroll
                for (int j = 0; j < MAXTH; j++) {
                    Px mid;
                    if (decltype(shared)::value && p == 1 && j == 0) mid = keep;
                    else {
                        const f32x2 w0 = { cur.w[0], cur.w[0] };
                        const int q = kBothAtOnce ? p : 0;
                        mid = Px{ sp[q][j][0].lo * w0, sp[q][j][0].hi * w0 };    // (0 + p0 is p0: gather_common.hpp vsum)
#pragma unroll
                        for (int k = 1; k < MAXTV; k++) {
                            if (decltype(full)::value || k < n) {
                                const f32x2 wk = { cur.w[k], cur.w[k] };
                                mid.lo = cvs::madd(sp[q][j][k].lo, wk, mid.lo);  // t += s * coeff
                                mid.hi = cvs::madd(sp[q][j][k].hi, wk, mid.hi);
                            }
                        }
                        if (decltype(shared)::value && p == 0 && j == 1) keep = mid;
                    }
                    const f32x2 wj = { wt[p][j], wt[p][j] };
                    if (j == 0) { hlo[p] = mid.lo * wj; hhi[p] = mid.hi * wj; }
                    else { hlo[p] = cvs::madd(mid.lo, wj, hlo[p]); hhi[p] = cvs::madd(mid.hi, wj, hhi[p]); }
                }
            }
        };
        pin(nxt.a);
        if constexpr (MAXTV > 2) pin(nxt.b);
        rc = decode(nxt);
        if (__builtin_expect(n == MAXTV, 1)) {
            if (MAXTH == 2 && kBothAtOnce && share) sums(std::true_type{}, std::true_type{});
            else sums(std::true_type{}, std::false_type{});
        } else {
            sums(std::false_type{}, std::false_type{});
            if (n <= 0) {                                                // a line without taps is zeros
#pragma unroll
                for (int p = 0; p < 2; p++) { hlo[p] = f32x2{ 0.0f, 0.0f }; hhi[p] = f32x2{ 0.0f, 0.0f }; }
            }
        }
        char *optr = obase + (size_t)line * trow;
        if constexpr (out_half) {
            const uint4 h = narrow8(hlo[0], hhi[0], hlo[1], hhi[1]);
            if (all_live || col_live[1]) {                               // (streamed: nothing reads the target back)
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(u32x4_t{ h.x, h.y, h.z, h.w }, reinterpret_cast<u32x4_t *>(optr));
            }
            else if (col_live[0]) *reinterpret_cast<uint2 *>(optr) = make_uint2(h.x, h.y);
        } else {
#pragma unroll
            for (int p = 0; p < 2; p++)
                if (all_live || col_live[p]) {
                    typedef float f32x4_t __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(f32x4_t{ hlo[p].x, hlo[p].y, hhi[p].x, hhi[p].y }, reinterpret_cast<f32x4_t *>(optr + 16 * 64 * p));
                }
        }
        if (line == wave) CVS_TVH_CLOCK(5);                              // the wave's first line: stores issued
    }
    CVS_TVH_CLOCK(7);                                                    // last line's stores issued
}

constexpr size_t kLdsCap = 64 * 1024;

int row_pixels(const cvk_fir2d_params *fp) { return fp->h.wide_foot + 1; }            // (the h table's line 0 is column fp->tx0)
int seg_index(int seg) { return seg == 64 ? 2 : seg == 32 ? 1 : 0; }
int seg_rows(const cvk_fir2d_params *fp, int seg) { return fp->v.span_lines[seg_index(seg)]; }
size_t lds_bytes(const cvk_fir2d_params *fp, int seg) { return (size_t)seg_rows(fp, seg) * (size_t)row_pixels(fp) * sizeof(float4) + (size_t)seg * 2 * sizeof(uint4); }

// lines per workgroup: as many as keep four workgroups on a CU (the set-up -- tap lists, row range, the rows themselves -- is
// three trips to memory, the same for 16 lines as for 64) and all of a lane's row requests in flight at once
int pick_seg(const cvk_fir2d_params *fp) {
    static int forced = -1;
    if (forced < 0) { const char *e = CVS_DIAG_ENV("CVS_TVH_SEG"); forced = e ? atoi(e) : 0; }     // (diagnostic build only)
    if ((forced == 16 || forced == 32 || forced == 64) && lds_bytes(fp, forced) <= kLdsCap && (size_t)seg_rows(fp, forced) * (size_t)fp->h.wide_foot <= (size_t)kNJ * kThreads) return forced;
    for (int seg = 64; seg > CVK_FIR_TVH_LINES; seg /= 2)
        if (lds_bytes(fp, seg) <= 40 * 1024 && (size_t)seg_rows(fp, seg) * (size_t)fp->h.wide_foot <= (size_t)kNJ * kThreads) return seg;
    return CVK_FIR_TVH_LINES;
}

template <int MAXTV, int MAXTH, bool INH>
int launch(const cvk_fir2d_params &fp, int line0, hipStream_t s) {
    const int cols = fp.tx1 - fp.tx0 + 1, rows = fp.ty1 - fp.ty0 + 1 - line0;
    const int seg = pick_seg(&fp);
    dim3 grid((unsigned)((cols + kTW - 1) / kTW), (unsigned)((rows + seg - 1) / seg), (unsigned)(fp.nframes > 1 ? fp.nframes : 1));
    hipLaunchKernelGGL((k_fir_tile_vh<MAXTV, MAXTH, INH>), grid, dim3(kThreads), lds_bytes(&fp, seg), s,
                       fp.h.pack, fp.h.foot, fp.v.lrec, line0, seg, fp.tx0, fp.tx1, fp.ty0, fp.ty1, row_pixels(&fp), seg_rows(&fp, seg), fp);
    return (int)hipGetLastError();
}

typedef int (*launch_fn)(const cvk_fir2d_params &, int, hipStream_t);
struct Instance { int maxtv, maxth; launch_fn f16, f32; };
#define CVK_TVH_INSTANCE(V, H) { V, H, launch<V, H, true>, launch<V, H, false> }
const Instance kInstances[] = { CVK_TVH_INSTANCE(2, 2), CVK_TVH_INSTANCE(4, 2), CVK_TVH_INSTANCE(2, 4), CVK_TVH_INSTANCE(4, 4) };

// (the horizontal instance width is the packed list's width: the kernel reads whole records)
const Instance *pick(const cvk_fir2d_params *fp) {
    for (const Instance &in : kInstances)
        if (fp->v.max_taps <= in.maxtv && fp->h.pack != NULL && fp->h.pack_width == in.maxth) return &in;
    return NULL;
}

}  // namespace

// Tables of an enlarging (or gently reducing) call: short lists, few source rows and columns under a tile.  The rest -- long
// lists, wide footprints -- is k_fir_vh's.
extern "C" int cvk_fir_tvh_supported(const cvk_fir2d_params *fp) {
    if (!(fp->in_half == fp->out_half && fp->v.streamable && fp->v.lrec != NULL && fp->v.max_taps >= 1 && fp->h.max_taps >= 1
          && fp->h.wide_foot >= 1 && fp->h.wide_foot <= 128 && fp->v.span_lines[0] >= 1 && pick(fp) != NULL)) return 0;
    if (fp->tx1 - fp->tx0 + 1 < 2 * kTW) return 0;                    // narrow targets: the strips of k_fir_vh
    // two halfs-pixels per lane are ONE 16-byte store: the pair must sit on a 16-byte boundary in every row
    if (fp->out_half && ((((uintptr_t)fp->target.data) & 15u) || (fp->target.pitch & 1) || ((fp->tx0 - fp->target.fx0) & 1))) return 0;
    if ((size_t)fp->v.span_lines[0] * (size_t)fp->h.wide_foot > (size_t)kNJ * kThreads) return 0;     // a lane stages at most kNJ source pixels
    return lds_bytes(fp, CVK_FIR_TVH_LINES) <= kLdsCap;
}

// Where both forms take a call, which one is faster (profiles/r04/scaler_forms.txt, one and two streams): floats -- the
// tiles, everywhere (6..16 %); halfs -- the tiles up to about one 4K frame of target (1080p -> 4K with two frames in flight:
// 17.2 against 19.1 us), the strips for larger targets (4K -> 6K, 4K -> 8K: 4..8 % for the strips).
extern "C" int cvk_fir_tvh_preferred(const cvk_fir2d_params *fp) {
    if (!cvk_fir_tvh_supported(fp)) return 0;
    if (!fp->out_half) return 1;
    return (size_t)(fp->tx1 - fp->tx0 + 1) * (size_t)(fp->ty1 - fp->ty0 + 1) <= (size_t)10 << 20;
}

// fp->ty0 is the vertical table's first line; lines fp->ty0 + line0 .. fp->ty1 are produced
extern "C" int cvk_fir_tvh(const cvk_fir2d_params *fp, int line0, void *stream) {
    if (fp->tx1 < fp->tx0 || fp->ty1 < fp->ty0 + line0 || line0 < 0) return 0;
    if (!cvk_fir_tvh_supported(fp)) return (int)hipErrorInvalidValue;
    const Instance *in = pick(fp);
    return fp->in_half ? in->f16(*fp, line0, (hipStream_t)stream) : in->f32(*fp, line0, (hipStream_t)stream);
}
