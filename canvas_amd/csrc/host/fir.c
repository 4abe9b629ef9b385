/*
 * fir.c -- FIR tap generators (src/cprocess/filter.c).  Parameter-sized host math: a handful of
 * floats per call, produced once per row/column of a resample and uploaded as tap tables; the
 * per-pixel multiply-accumulate is in kernels/fir_ops.hip.
 *
 * Contract kept from filter.c / framework.h:629-645:
 *   - sub < 1 (downsampling): support 1/sub, taps normalised to sum 1;  sub >= 1: support sub, raw
 *   - taps exactly on the support edge are dropped
 *   - center = -(int)leftEdge, i.e. the tap that would sit on the (unshifted) centre
 *   - caller-supplied buffer too small: coeff untouched, width = needed, center = -1
 *   - coeff == NULL: allocated here, released with filter_free
 * Build note: no -ffast-math, no contraction; Lanczos is evaluated in double then cast (filter.c:113-131).
 */
#include "internal.h"
#include <math.h>

static bool tap_extent(float support, float offset, fir_filter *f) {
    float left = ceilf(offset - support);
    float right = floorf(offset + support);
    if (left == offset - support) left += 1.0f;
    if (right == offset + support) right -= 1.0f;
    const int needed = (int)right - (int)left + 1;

    if (f->coeff && f->width < needed) {
        f->width = needed;
        f->center = -1;
        return false;
    }
    f->width = needed;
    f->center = -(int)left;
    if (!f->coeff) f->coeff = malloc(sizeof(float) * (size_t)(needed > 0 ? needed : 1));
    return f->coeff != NULL;
}

static void unity_gain(float sub, float sum, fir_filter *f) {
    if (sub < 1.0f && sum != 0.0f)
        for (int i = 0; i < f->width; i++) f->coeff[i] /= sum;
}

CVS_EXPORT void filter_createTriangle(float sub, float offset, fir_filter *f) {        /* filter.c:24-76 */
    if (!f || !(sub > 0.0f)) return;
    const float support = sub < 1.0f ? 1.0f / sub : sub;
    if (!tap_extent(support, offset, f)) return;
    float sum = 0.0f;
    for (int i = 0; i < f->width; i++) {
        float c = 1.0f - fabsf((1.0f / support) * ((i - f->center) - offset));
        f->coeff[i] = c;
        sum += c;
    }
    unity_gain(sub, sum, f);
}

CVS_EXPORT void filter_createLanczos(float sub, int kernel_size, float offset, fir_filter *f) { /* filter.c:78-148 */
    if (!f || !(sub > 0.0f) || kernel_size <= 0) return;
    const float support = sub < 1.0f ? 1.0f / sub : sub;
    if (!tap_extent(kernel_size * support, offset, f)) return;
    const double pi = 3.1415926535897932384626433832795028841971693993751;
    float sum = 0.0f;
    for (int i = 0; i < f->width; i++) {
        const double x = (1.0 / support) * ((i - f->center) - (double)offset);
        float c;
        if (x == 0.0) c = 1.0f;
        else if (x <= -kernel_size || x >= kernel_size) c = 0.0f;
        else {
            const double v = (kernel_size * sin(pi * x) * sin(pi * x / kernel_size)) / (pi * pi * x * x);
            c = isfinite(v) ? (float)v : 1.0f;
        }
        f->coeff[i] = c;
        sum += c;
    }
    unity_gain(sub, sum, f);
}

CVS_EXPORT void filter_free(fir_filter *f) {                                            /* filter.c:150-153 */
    if (!f) return;
    free(f->coeff);
    f->coeff = NULL;
}
