// blur_even_ops.hip -- the register-window blur (blur_kernel.hpp) for even tap counts 4-16 (centre ntaps / 2, as for odd
// lists).  One strip width (256 lanes); called from cvk_blur (blur_ops.hip).
#include "blur_kernel.hpp"

extern "C" int cvk_blur_even(const cvk_blur_params *bp, int cus, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (bp->ntaps) {
    case 4:  return launch<4, 256, 1>(*bp, cus, s);
    case 6:  return launch<6, 256, 1>(*bp, cus, s);
    case 8:  return launch<8, 256, 1>(*bp, cus, s);
    case 10: return launch<10, 256, 1>(*bp, cus, s);
    case 12: return launch<12, 256, 1>(*bp, cus, s);
    case 14: return launch<14, 256, 1>(*bp, cus, s);
    case 16: return launch<16, 256, 1>(*bp, cus, s);
    default: return (int)hipErrorInvalidValue;
    }
}
