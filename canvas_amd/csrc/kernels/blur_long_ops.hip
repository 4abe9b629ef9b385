// blur_long_ops.hip -- the register-window blur (blur_kernel.hpp) for tap lists of 17-31: sums formed in groups of eight taps.
// One strip width (256 lanes); called from cvk_blur (blur_ops.hip).
#include "blur_kernel.hpp"

extern "C" int cvk_blur_long(const cvk_blur_params *bp, int cus, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (bp->ntaps) {
    case 17: return launch<17, 256, 1>(*bp, cus, s);
    case 19: return launch<19, 256, 1>(*bp, cus, s);
    case 21: return launch<21, 256, 1>(*bp, cus, s);
    case 23: return launch<23, 256, 1>(*bp, cus, s);
    case 25: return launch<25, 256, 1>(*bp, cus, s);
    case 27: return launch<27, 256, 1>(*bp, cus, s);
    case 29: return launch<29, 256, 1>(*bp, cus, s);
    case 31: return launch<31, 256, 1>(*bp, cus, s);
    default: return (int)hipErrorInvalidValue;
    }
}
