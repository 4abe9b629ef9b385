// chain_deep_ops.hip -- the chain kernel (chain_kernel.hpp) for stacks of 5..8 layers.
//
// Same trip loop and chunk walk as the 1..4-layer instances in chain_ops.hip; what differs is the register budget:
// two register sets of up to eight 16-byte words (64 VGPRs) plus the accumulators do not fit the 128 VGPRs a
// 1 024-lane launch bound leaves, so these instances are bounded at 512 lanes (256 VGPRs) and their job records carry
// eight layer pointers (48 frames per launch instead of 64).  Reference: the same per-layer passes as chain_ops.hip
// (src/cprocess/color.c:104-165, src/cprocess/main.c:43-71,115-139), nlayers - 1 overs deep.
// Bound: HBM.  Algorithmic bytes per output pixel: 8 * (nlayers + 1)  (48..72).
#include "chain_kernel.hpp"

namespace {

template <int NL, int DIAG>
int launch_mode(const BatchT<8> &b, int n, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, unsigned block, int lshift, hipStream_t s) {
    if (mat.plain) return launch<NL, CHAIN_PLAIN, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
    return launch<NL, CHAIN_GRADE, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
}

template <int DIAG>
int launch_deep(int nl, const BatchT<8> &b, int n, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, unsigned block, int lshift, hipStream_t s) {
    switch (nl) {
    case 5: return launch_mode<5, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
    case 6: return launch_mode<6, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
    case 7: return launch_mode<7, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
    default: return launch_mode<8, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
    }
}

}  // namespace

extern "C" int cvk_chain_deep(const cvk_chain_job *jobs, int njobs, int nlayers, const cvs::Mat *mat, const uint16_t *pre, const uint16_t *post,
                              unsigned grid, unsigned block, int lshift, int diag, uint64_t bytes_per_launch, void *stream, int *taken) {
    if (nlayers < 5 || nlayers > 8 || mat->cross || block > 512) return (int)hipErrorInvalidValue;
    BatchT<8> b;
    const int n = fill_batch(b, jobs, njobs, nlayers, bytes_per_launch);
    *taken = n;
    hipStream_t s = (hipStream_t)stream;
#ifdef CVS_DIAG
    if (diag == DIAG_MEMORY_ONLY) return launch_deep<DIAG_MEMORY_ONLY>(nlayers, b, n, *mat, pre, post, grid, block, lshift, s);
    if (diag == DIAG_COMPUTE_ONLY) return launch_deep<DIAG_COMPUTE_ONLY>(nlayers, b, n, *mat, pre, post, grid, block, lshift, s);
#endif
    (void)diag;
    return launch_deep<DIAG_NONE>(nlayers, b, n, *mat, pre, post, grid, block, lshift, s);
}
