// blur_ops.hip -- separable FIR blur with one fixed tap list, both passes in one sweep down the frame: dispatch and the
// instances for 3-15 taps (1:1) and the Lanczos halving lists.  The kernel itself is in blur_kernel.hpp; the instances for
// 17-31 taps and for even counts are compiled in blur_long_ops.hip / blur_even_ops.hip (translation units of their own
// build in parallel).
#include "blur_kernel.hpp"
#include <atomic>

extern "C" int cvk_blur_long(const cvk_blur_params *bp, int cus, void *stream);
extern "C" int cvk_blur_even(const cvk_blur_params *bp, int cus, void *stream);

namespace {

template <int W>
int pick(const cvk_blur_params *bp, int cus, hipStream_t s) {
    if (bp->step == 2) {
        switch (bp->ntaps) {                       // Lanczos-k at 1/2: 4k - 1 taps
        case 3:  return launch<3, W, 2>(*bp, cus, s);
        case 7:  return launch<7, W, 2>(*bp, cus, s);
        case 11: return launch<11, W, 2>(*bp, cus, s);
        case 15: return launch<15, W, 2>(*bp, cus, s);
        default: return (int)hipErrorInvalidValue;
        }
    }
    switch (bp->ntaps) {
    case 3:  return launch<3, W, 1>(*bp, cus, s);
    case 5:  return launch<5, W, 1>(*bp, cus, s);
    case 7:  return launch<7, W, 1>(*bp, cus, s);
    case 9:  return launch<9, W, 1>(*bp, cus, s);
    case 11: return launch<11, W, 1>(*bp, cus, s);
    case 13: return launch<13, W, 1>(*bp, cus, s);
    case 15: return launch<15, W, 1>(*bp, cus, s);
    }
#ifndef CVS_CONTRACT
    if constexpr (W == 256) {
        if (!(bp->ntaps & 1)) return cvk_blur_even(bp, cus, s);         // blur_even_ops.hip
        if (bp->ntaps > 15) return cvk_blur_long(bp, cus, s);           // blur_long_ops.hip
    }
#endif
    return (int)hipErrorInvalidValue;
}

}  // namespace

extern "C" int cvk_blur_supported(int ntaps, int step) {
#ifdef CVS_CONTRACT
    // the contracted build has the instances of this file only (3..15 odd); longer and even lists go to the table kernels
    if (step == 1) return ntaps >= 3 && ntaps <= 15 && (ntaps & 1);
#endif
    if (step == 1) return (ntaps >= 3 && ntaps <= 31 && (ntaps & 1)) || (ntaps >= 4 && ntaps <= 16);
    if (step == 2) return ntaps == 3 || ntaps == 7 || ntaps == 11 || ntaps == 15;
    return 0;
}

// Two columns per lane: 5-8 % faster than one from 4K frames down to a few strips; a frame narrower
// than two of its 120-column strips stays with the wide workgroups.
extern "C" int cvk_blur_takes_pairs(const cvk_blur_params *bp) {
    if (bp->flags & CVK_BLUR_ONE_COLUMN) return 0;
    static std::atomic<int> env_cached{ -2 };
    int env = env_cached.load(std::memory_order_relaxed);
    if (env == -2) { const char *e = CVS_DIAG_ENV("CVS_BLUR_PAIR"); env = e ? atoi(e) : -1; env_cached.store(env, std::memory_order_relaxed); }
    if (env == 0) return 0;
    if (!(bp->flags & CVK_BLUR_TWO_COLUMNS) && bp->tx1 - bp->tx0 + 1 < 256) return 0;
    return cvk_blur_pair_supported(bp);
}

extern "C" int cvk_blur(const cvk_blur_params *bp_in, int cus, void *stream) {
    if (bp_in->tx1 < bp_in->tx0 || bp_in->ty1 < bp_in->ty0) return 0;
    cvk_blur_params bp = *bp_in;
    if (bp.step <= 0) bp.step = 1;
    if (!cvk_blur_supported(bp.ntaps, bp.step)) return (int)hipErrorInvalidValue;
    if (bp.nover < 0 || bp.nover > CVK_BLUR_MAX_OVER || (bp.nover > 0 && !(bp.in_half && bp.out_half && bp.step == 1))) return (int)hipErrorInvalidValue;
    const int cols = bp.tx1 - bp.tx0 + 1;
    // strip width: 256 lanes unless the frame is so narrow that 128 wastes fewer lanes
    static std::atomic<int> env_w_cached{ -1 }, env_rows_cached{ -1 };      // (several threads may launch at once)
    int env_w = env_w_cached.load(std::memory_order_relaxed), env_rows = env_rows_cached.load(std::memory_order_relaxed);
    if (env_w < 0) { const char *e = CVS_DIAG_ENV("CVS_BLUR_WIDTH"); env_w = e ? atoi(e) : 0; env_w_cached.store(env_w, std::memory_order_relaxed); }
    if (env_rows < 0) { const char *e = CVS_DIAG_ENV("CVS_BLUR_ROWS"); env_rows = e ? atoi(e) : 0; env_rows_cached.store(env_rows, std::memory_order_relaxed); }
    if (cvk_blur_takes_pairs(&bp)) return cvk_blur_pair(&bp, cus, stream);            // two columns per lane (blur_pair_ops.hip)
    const int width = (bp.ntaps > 15 || !(bp.ntaps & 1)) ? 256 : env_w ? env_w : (cols <= 128 ? 128 : 256);      // long and even lists: 256-lane instances only
    if (bp.rows_per_wg <= 0 && env_rows > 0) bp.rows_per_wg = env_rows;
    return width == 128 ? pick<128>(&bp, cus, (hipStream_t)stream) : pick<256>(&bp, cus, (hipStream_t)stream);
}
