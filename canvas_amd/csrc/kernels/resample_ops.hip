// resample_ops.hip -- separable FIR with per-line tap tables (Lanczos at any factor, blurs of any length), both passes in
// one sweep down the frame, no tile of the source in LDS.
//
// k_fir2d (fir_ops.hip) keeps a tile's whole source footprint and its horizontal result in LDS; at a 0.4x Lanczos that
// is 100 KiB per 32 x 16 target pixels, one workgroup per CU, and the kernel runs at a twentieth of the memory rate.
// Here a 128-lane workgroup owns 128 target columns and a segment of target rows and walks the SOURCE rows the segment
// needs, top to bottom:
//   source row s -> LDS row buffer (double buffered, one barrier per row; requested a row ahead; f16 sources widened on
//   the way in)
//   H: a lane forms the horizontal sum of row s for its target column from its own tap list (held in registers)
//   V: the target rows whose tap lists contain s are "active"; each has an accumulator in registers and takes
//      acc += H * w.  Source rows arrive in ascending order, so every accumulator adds its taps in ascending source
//      order -- the order of the gather it replaces.  When s is a row's last tap the row is stored and its accumulator
//      goes to the next target row that will need one.
// The number of rows active at once is about the filter's support measured in TARGET rows (6-7 for Lanczos3 when
// reducing, support x factor when enlarging, ntaps for a blur), so the vertical window is a handful of registers
// whatever the factor.  All control flow around the accumulators is wave-uniform and the accumulator index is a
// compile-time constant of an unrolled loop: no register array is indexed at run time.
//
// The host (scale.c) sends a table pair here only when, on the vertical axis, every line's taps are consecutive source
// rows and first / last taps never decrease from one line to the next (what the blur and Lanczos planners produce).
// Arithmetic: every sum starts at 0.0f, products and additions separately rounded, ascending order -- bit for bit the
// two gather passes.  Bound: HBM.  Algorithmic bytes: source pixel once + target pixel once.
#include <climits>
#include <atomic>
#include "kernels.h"
#include "chain_math.hpp"

namespace {

using cvs::f32x2;

constexpr int kW = 128;        // target columns (= lanes) per workgroup: narrow strips, so that a 1080p target still makes a thousand workgroups
constexpr int kPF = 6;         // source pixels a lane fetches per row at most (strip footprint <= kPF * kW)
constexpr int kPFD = 4;        // source rows in flight

// a source pixel as it lies in memory (f16: in .x/.y), and as f32
__device__ __forceinline__ uint4 fetch_src(const cvk_view &v, bool half, int x, int y) {
    const size_t i = (size_t)(y - v.fy0) * (size_t)v.pitch + (size_t)(x - v.fx0);
    if (!half) return reinterpret_cast<const uint4 *>(v.data)[i];
    const uint2 p = reinterpret_cast<const uint2 *>(v.data)[i];
    return make_uint4(p.x, p.y, 0u, 0u);
}
__device__ __forceinline__ float4 widen_src(uint4 p, bool half) {
    if (!half) return make_float4(__uint_as_float(p.x), __uint_as_float(p.y), __uint_as_float(p.z), __uint_as_float(p.w));
    return make_float4(cvs::h2f(p.x & 0xFFFFu), cvs::h2f(p.x >> 16), cvs::h2f(p.y & 0xFFFFu), cvs::h2f(p.y >> 16));
}

template <int MAXT, int NACC>
__global__ __launch_bounds__(kW) void k_fir_stream(cvk_fir2d_params fp, int rows_per_wg, int lds_px) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float4 *rowbuf = reinterpret_cast<float4 *>(lds_raw);                         // [2][lds_px + 1]: one zero pixel behind each row
    const int row_px = lds_px + 1;
    float *vtap = reinterpret_cast<float *>(rowbuf + 2 * (size_t)row_px);         // [rows_per_wg][v.stride]
    int *vs0 = reinterpret_cast<int *>(vtap + (size_t)rows_per_wg * fp.v.stride); // [rows_per_wg] first source row of a target row
    int *vnn = vs0 + rows_per_wg;                                                 // [rows_per_wg] its tap count

    const int lane = threadIdx.x;
    const int c0 = fp.tx0 + (int)blockIdx.x * kW;
    const int c1 = min(c0 + kW - 1, fp.tx1);
    const int tcol = c0 + lane;
    const bool col_live = tcol <= fp.tx1;
    const int ta = fp.ty0 + (int)blockIdx.y * rows_per_wg;
    const int tb = min(ta + rows_per_wg - 1, fp.ty1);
    const int nrows = tb - ta + 1;
    const int vstride = fp.v.stride, hstride = fp.h.stride;

    // the segment's vertical tables -> LDS
    for (int i = lane; i < nrows * vstride; i += kW) vtap[i] = fp.v.taps[(size_t)(ta - fp.ty0) * vstride + i];
    for (int i = lane; i < nrows; i += kW) {
        const int n = fp.v.ntaps[ta - fp.ty0 + i];
        vnn[i] = n;
        vs0[i] = n > 0 ? fp.v.src[(size_t)(ta - fp.ty0 + i) * vstride] : 0;
    }
    // source columns under the strip: union of the footprints of its 32-column tiles (host-built, first > last = empty)
    int sx_lo = INT_MAX, sx_hi = INT_MIN;
    for (int t = (c0 - fp.tx0) / CVK_FIR2D_TILE_X; t <= (c1 - fp.tx0) / CVK_FIR2D_TILE_X; t++) {
        const int lo = fp.h.foot[2 * t], hi = fp.h.foot[2 * t + 1];
        if (hi >= lo) { sx_lo = min(sx_lo, lo); sx_hi = max(sx_hi, hi); }
    }
    const int sw = sx_hi >= sx_lo ? min(sx_hi - sx_lo + 1, lds_px) : 0;           // (the host sized lds_px to cover every strip)
    // this lane's horizontal taps, relative to the strip's first source column.  Lists shorter than MAXT are padded with
    // weight 0 on the ZERO PIXEL kept behind the row (index lds_px): acc + 0 * 0 == acc, so the sum needs no per-tap
    // select (a padded tap on a real pixel would turn an Inf or NaN there into a NaN of the sum)
    const int hn = col_live ? min(fp.h.ntaps[tcol - fp.tx0], MAXT) : 0;
    int sidx[MAXT];
    float wt[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; k++) {
        const bool live = k < hn;
        sidx[k] = live ? fp.h.src[(size_t)(tcol - fp.tx0) * hstride + k] - sx_lo : lds_px;
        wt[k] = live ? fp.h.taps[(size_t)(tcol - fp.tx0) * hstride + k] : 0.0f;
    }
    if (lane < 2) rowbuf[(size_t)lane * row_px + lds_px] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();

    const size_t tpx = fp.out_half ? 8 : 16;
    char *tbase = reinterpret_cast<char *>(fp.target.data) + ((size_t)(tcol - fp.target.fx0)) * tpx;
    const size_t trow = (size_t)fp.target.pitch * tpx;
    auto store_row = [&](int t, f32x2 rg, f32x2 ba) {
        if (!col_live) return;
        char *o = tbase + (size_t)(t - fp.target.fy0) * trow;
        if (fp.out_half) *reinterpret_cast<uint2 *>(o) = make_uint2(cvs::f2h_rz2(rg.x, rg.y), cvs::f2h_rz2(ba.x, ba.y));
        else *reinterpret_cast<float4 *>(o) = make_float4(rg.x, rg.y, ba.x, ba.y);
    };

    // accumulator slots: target row t lives in slot (t - ta) % NACC while it is active
    int st[NACC], s_first[NACC], s_last[NACC];
    f32x2 arg[NACC], aba[NACC];
    // hand slot j the next target row >= t (stepping by NACC) that has taps; rows without taps are stored as zeros
    // (the tables are the same for every lane: readfirstlane keeps the slot bookkeeping in scalar registers)
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto assign = [&](int t, int &slot_t, int &slot_first, int &slot_last) {
        while (t <= tb && uni(vnn[t - ta]) == 0) { store_row(t, f32x2{ 0.0f, 0.0f }, f32x2{ 0.0f, 0.0f }); t += NACC; }
        slot_t = t;
        if (t <= tb) { slot_first = uni(vs0[t - ta]); slot_last = slot_first + uni(vnn[t - ta]) - 1; }
        else { slot_first = INT_MAX; slot_last = INT_MIN; }
    };
#pragma unroll
    for (int j = 0; j < NACC; j++) {
        arg[j] = aba[j] = f32x2{ 0.0f, 0.0f };
        assign(ta + j, st[j], s_first[j], s_last[j]);
    }
    // source rows the segment walks (first / last taps do not decrease from row to row: the ends are at the ends)
    int s_lo = INT_MAX, s_hi = INT_MIN;
    for (int i = 0; i < nrows; i++) {
        const int n = uni(vnn[i]), f = uni(vs0[i]);
        if (n > 0) { s_lo = min(s_lo, f); s_hi = max(s_hi, f + n - 1); }
    }

    const bool in_half = fp.in_half != 0;
    // kPFD source rows are on their way at any time (a row step is a few hundred cycles of work, a load from HBM takes a
    // couple of thousand: with ONE row in flight -- the first version -- every step waited for memory).  Slot = row % kPFD,
    // a compile-time index because the row loop below is unrolled kPFD times.
    uint4 pf[kPFD][kPF];
    auto fetch_row = [&](uint4 (&dst)[kPF], int s) {
#pragma unroll
        for (int q = 0; q < kPF; q++) {
            const int x = lane + q * kW;
            dst[q] = make_uint4(0u, 0u, 0u, 0u);
            if (x < sw && s <= s_hi) dst[q] = fetch_src(fp.source, in_half, sx_lo + x, s);
        }
    };
    if (s_lo <= s_hi) {
#pragma unroll
        for (int d = 0; d < kPFD; d++) fetch_row(pf[d], s_lo + d);
    }
    for (int sb = s_lo; sb <= s_hi; sb += kPFD) {                 // uniform bounds: every wave runs every iteration
#pragma unroll
      for (int d = 0; d < kPFD; d++) {
        const int s = sb + d;
        if (s > s_hi) break;                                      // uniform
        float4 *buf = rowbuf + (size_t)((s - s_lo) & 1) * row_px;
#pragma unroll
        for (int q = 0; q < kPF; q++) {
            const int x = lane + q * kW;
            if (x < sw) buf[x] = widen_src(pf[d][q], in_half);
        }
        fetch_row(pf[d], s + kPFD);                               // in flight while the next kPFD rows are filtered
        __syncthreads();
        f32x2 hrg = { 0.0f, 0.0f }, hba = { 0.0f, 0.0f };
#pragma unroll
        for (int k0 = 0; k0 < MAXT; k0 += 8) {                    // groups of eight taps: reads first, then products, then the sums in tap order
            float4 v[8];
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = buf[sidx[k0 + c]];
            f32x2 prg[8], pba[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                prg[c] = f32x2{ v[c].x, v[c].y } * wt[k0 + c];
                pba[c] = f32x2{ v[c].z, v[c].w } * wt[k0 + c];
            }
#pragma unroll
            for (int c = 0; c < 8; c++) {
                hrg = hrg + prg[c];
                hba = hba + pba[c];
            }
        }
#pragma unroll
        for (int j = 0; j < NACC; j++) {
            if (s >= s_first[j] && s <= s_last[j]) {               // uniform
                const float w = vtap[(size_t)(st[j] - ta) * vstride + (s - s_first[j])];
                arg[j] = arg[j] + hrg * w;
                aba[j] = aba[j] + hba * w;
                if (s == s_last[j]) {
                    store_row(st[j], arg[j], aba[j]);
                    arg[j] = aba[j] = f32x2{ 0.0f, 0.0f };
                    assign(st[j] + NACC, st[j], s_first[j], s_last[j]);
                }
            }
        }
      }
    }
}

template <int MAXT, int NACC>
int launch(const cvk_fir2d_params &fp, int cus, hipStream_t s) {
    const int cols = fp.tx1 - fp.tx0 + 1, rows = fp.ty1 - fp.ty0 + 1;
    const int strips = (cols + kW - 1) / kW;
    const int lds_px = fp.max_sw;                                   // widest strip footprint, from the host
    auto lds_bytes = [&](int r) { return (size_t)2 * (lds_px + 1) * sizeof(float4) + (size_t)r * fp.v.stride * sizeof(float) + (size_t)r * 2 * sizeof(int); };
    static std::atomic<bool> raised{ false };       // (several threads may launch at once; setting the attribute twice is harmless)
    if (!raised.load(std::memory_order_acquire)) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir_stream<MAXT, NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); raised.store(true, std::memory_order_release); }
    // rows per workgroup: about one wave of resident workgroups over the whole frame (as k_blur), at least twice the
    // number of rows a source row feeds, at most what the tables of a segment leave room for
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fir_stream<MAXT, NACC>, kW, lds_bytes(64)) != hipSuccess || per_cu < 1) per_cu = 1;
    int segs = (per_cu * (cus > 0 ? cus : 256)) / strips;
    if (segs < 1) segs = 1;
    int r = (rows + segs - 1) / segs;
    // a segment re-reads (and re-filters) the source rows its first target rows reach back to: about the tap count of a
    // line.  Below 2 x NACC target rows that halo is most of the work (the first version ran 0.4x Lanczos segments of 8
    // rows: 35 source rows walked for 20 useful ones)
    if (r < 2 * NACC) r = 2 * NACC;
    if (r > 256) r = 256;
    if (lds_px > kPF * kW) return (int)hipErrorInvalidValue;           // a lane fetches at most kPF pixels per row
    if (r > rows) r = rows;
    if (lds_bytes(r) > 150 * 1024) return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)strips, (unsigned)((rows + r - 1) / r));
    hipLaunchKernelGGL((k_fir_stream<MAXT, NACC>), grid, dim3(kW), lds_bytes(r), s, fp, r, lds_px);
    return (int)hipGetLastError();
}

}  // namespace

// fp->max_sw: the widest source footprint of any 128-column strip; v_active: the most target rows any source row feeds
extern "C" int cvk_fir_stream_supported(int h_taps, int v_active) { return h_taps >= 1 && h_taps <= 32 && v_active >= 1 && v_active <= 16; }

extern "C" int cvk_fir_stream(const cvk_fir2d_params *fp, int h_taps, int v_active, int cus, void *stream) {
    if (fp->tx1 < fp->tx0 || fp->ty1 < fp->ty0) return 0;
    if (!cvk_fir_stream_supported(h_taps, v_active)) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    if (h_taps <= 16) return v_active <= 8 ? launch<16, 8>(*fp, cus, s) : launch<16, 16>(*fp, cus, s);
    return v_active <= 8 ? launch<32, 8>(*fp, cus, s) : launch<32, 16>(*fp, cus, s);
}
