/*
 * oracle/tables.c -- half<->float tables, transfer LUTs, FIR tap generators.
 * TEST INFRASTRUCTURE (see oracle.h).  Restates:
 *   src/cprocess/genhalf.py:25-93   (table construction)
 *   src/cprocess/half.c:31-85       (table-driven conversions, lookup)
 *   src/cprocess/gammatab.c:8-250   (transfer LUTs, gamma-0.45 ramp)
 *   src/cprocess/filter.c:24-153    (triangle / Lanczos taps)
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* ---- van der Zijp tables, generated at first use (genhalf.py) ---- */

static struct { uint16_t base; uint8_t shift; } t_f2h[512];
static uint32_t t_mant[2048];
static struct { uint16_t exponent, offset; } t_eo[64];
static pthread_once_t t_once = PTHREAD_ONCE_INIT;

static void build_half_tables(void) {
    /* genhalf.py:25-55: one entry per (sign, biased exponent) */
    for (int s = 0; s < 2; s++) {
        uint16_t sign = s ? 0x8000 : 0;
        for (int i = 0; i < 256; i++) {
            int e = i - 127, k = s * 256 + i;
            if (e < -24)       { t_f2h[k].base = sign;                                   t_f2h[k].shift = 24; }
            else if (e < -14)  { t_f2h[k].base = sign | (uint16_t)(0x0400 >> (-14 - e)); t_f2h[k].shift = (uint8_t)(-e - 1); }
            else if (e <= 15)  { t_f2h[k].base = sign | (uint16_t)((e + 15) << 10);      t_f2h[k].shift = 13; }
            else if (e < 128)  { t_f2h[k].base = sign | 0x7C00;                          t_f2h[k].shift = 24; }
            else               { t_f2h[k].base = sign | 0x7C00;                          t_f2h[k].shift = 13; }
        }
    }
    /* genhalf.py:58-75: subnormal mantissas are pre-normalised */
    t_mant[0] = 0;
    for (uint32_t i = 1; i < 1024; i++) {
        uint32_t m = i << 13, e = 0;
        while (!(m & 0x00800000u)) { e -= 0x00800000u; m <<= 1; }
        m &= ~0x00800000u;
        e += 0x38800000u;
        t_mant[i] = m | e;
    }
    for (uint32_t i = 0; i < 1024; i++)
        t_mant[1024 + i] = 0x38000000u + (i << 13);
    /* genhalf.py:80-93 */
    for (int s = 0; s < 2; s++) {
        uint16_t sign = s ? 0x8000 : 0;
        t_eo[s * 32].exponent = sign;  t_eo[s * 32].offset = 0;
        for (int i = 1; i < 31; i++) { t_eo[s * 32 + i].exponent = sign | (uint16_t)(i << 7); t_eo[s * 32 + i].offset = 1024; }
        t_eo[s * 32 + 31].exponent = sign | 0x4780;  t_eo[s * 32 + 31].offset = 1024;
    }
}

static inline void need_tables(void) { pthread_once(&t_once, build_half_tables); }

/* half.c:31-37 */
static inline float h2f(orc_half h) {
    union { float f; uint32_t i; } u;
    u.i = t_mant[t_eo[h >> 10].offset + (h & 0x3FF)] + ((uint32_t)t_eo[h >> 10].exponent << 16);
    return u.f;
}

/* half.c:47-51 -- note: truncates the mantissa (round toward zero) */
static inline orc_half f2h(float v) {
    union { float f; uint32_t i; } u; u.f = v;
    unsigned k = (u.i >> 23) & 0x1FF;
    return (orc_half)(t_f2h[k].base + ((u.i & 0x007FFFFFu) >> t_f2h[k].shift));
}

void orc_half_to_float(float *out, const orc_half *in, int count) {      /* half.c:62-65 */
    need_tables();
    for (int i = 0; i < count; i++) out[i] = h2f(in[i]);
}

void orc_float_to_half(orc_half *out, const float *in, int count) {      /* half.c:67-70 */
    need_tables();
    for (int i = 0; i < count; i++) out[i] = f2h(in[i]);
}

void orc_half_to_float_fast(float *out, const orc_half *in, int count) { /* half.c:39-45,72-75 */
    for (int i = 0; i < count; i++) {
        union { float f; uint32_t i; } u;
        uint32_t v = in[i];
        u.i = ((v & 0x8000u) << 16) | (((v & 0x7c00u) + 0x1C000u) << 13) | ((v & 0x03FFu) << 13);
        out[i] = u.f;
    }
}

void orc_float_to_half_fast(orc_half *out, const float *in, int count) { /* half.c:53-59,77-80 */
    for (int i = 0; i < count; i++) {
        union { float f; uint32_t i; } u; u.f = in[i];
        out[i] = (orc_half)(((u.i >> 16) & 0x8000u) | ((((u.i & 0x7f800000u) - 0x38000000u) >> 13) & 0x7c00u) |
                            ((u.i >> 13) & 0x03ffu));
    }
}

void orc_half_lookup(const orc_half *table, orc_half *out, const orc_half *in, int count) { /* half.c:82-85 */
    for (int i = 0; i < count; i++) out[i] = table[in[i]];
}

/* ---- transfer LUTs (gammatab.c) ---- */

static inline float clampf_(float v, float lo, float hi) {               /* framework.h:139-149 */
    float t = v > lo ? v : lo;
    return t < hi ? t : hi;
}

static float tf_rec709_to_linear(float in) {                              /* gammatab.c:48-56 */
    const float transition = 4.5f * 0.018f;
    if (in < transition) return in / 4.5f;
    return powf((in + 0.099f) / 1.099f, 1.0f / 0.45f);
}
static float tf_display(float in) {                                       /* gammatab.c:145-150 */
    if (in < 0.0f) return 0.0f;
    return powf(in, 2.5f);
}
static float tf_linear_to_rec709(float in) {                              /* gammatab.c:58-66 */
    const float transition = 0.018f;
    if (in < transition) return in * 4.5f;
    return 1.099f * powf(in, 0.45f) - 0.099f;
}
static float tf_linear_to_srgb(float in) {                                /* gammatab.c:201-211 */
    const float transition = 0.0031308f;
    const float a = 0.055;
    if (in <= transition) return in * 12.92f;
    return (1.0f + a) * powf(in, 1.0f / 2.4f) - a;
}

static orc_half *lut[4];
static uint8_t *ramp45;
static pthread_once_t lut_once = PTHREAD_ONCE_INIT;

static void build_luts(void) {
    /* gammatab.c:87-106 and siblings: table[i] = f2h(func(h2f(i))) for all 65536 codes */
    float (*fn[4])(float) = { tf_rec709_to_linear, tf_display, tf_linear_to_rec709, tf_linear_to_srgb };
    orc_half *codes = malloc(65536 * sizeof(orc_half));
    float *f = malloc(65536 * sizeof(float)), *g = malloc(65536 * sizeof(float));
    for (int i = 0; i < 65536; i++) codes[i] = (orc_half)i;
    orc_half_to_float(f, codes, 65536);
    for (int k = 0; k < 4; k++) {
        lut[k] = malloc(65536 * sizeof(orc_half));
        for (int i = 0; i < 65536; i++) g[i] = fn[k](f[i]);
        orc_float_to_half(lut[k], g, 65536);
    }
    /* gammatab.c:8-38: uint8 = (uint8_t) clampf(powf(x, 0.45f) * 255, 0, 255) */
    ramp45 = malloc(65536);
    for (int i = 0; i < 65536; i++)
        ramp45[i] = (uint8_t)clampf_(powf(f[i], 0.45f) * 255.0f, 0.0f, 255.0f);
    free(codes); free(f); free(g);
}

/* the software widget's ramp: src/cprocess/widget_gl.c:955-968 (rendering intent 1.25 by default, :428-429) */
void orc_widget_ramp(uint8_t *ramp, float rendering_intent) {
    orc_half *codes = malloc(65536 * sizeof(orc_half));
    float *f = malloc(65536 * sizeof(float));
    for (int i = 0; i < 65536; i++) codes[i] = (orc_half)i;
    orc_half_to_float(f, codes, 65536);
    for (int i = 0; i < 65536; i++)
        ramp[i] = (uint8_t)lrint(clampf_(powf(f[i], rendering_intent) * 255.0f, 0.0f, 255.0f));
    free(codes); free(f);
}

const orc_half *orc_transfer_table(int which) {
    pthread_once(&lut_once, build_luts);
    return (which >= 0 && which < 4) ? lut[which] : NULL;
}

void orc_transfer(int which, orc_half *out, const orc_half *in, size_t count) {
    orc_half_lookup(orc_transfer_table(which), out, in, (int)count);
}

const uint8_t *orc_gamma45_ramp(void) {
    pthread_once(&lut_once, build_luts);
    return ramp45;
}

/* ---- FIR tap generators (filter.c) ---- */

/* shared prologue of filter.c:35-61 and :84-108; returns 0 when the caller's buffer is too small */
static int fir_extent(float support, float offset, orc_fir *f) {
    float left = ceilf(offset - support), right = floorf(offset + support);
    if (left == offset - support) left++;       /* taps exactly on the support edge are dropped */
    if (right == offset + support) right--;
    int full = (int)right - (int)left + 1;
    if (f->coeff && f->width < full) { f->width = full; f->center = -1; return 0; }
    f->width = full;
    f->center = -(int)left;
    if (!f->coeff) f->coeff = malloc(sizeof(float) * (size_t)f->width);
    return 1;
}

static void fir_normalise(float sub, float sum, orc_fir *f) {             /* filter.c:70-75,137-142 */
    if (sub < 1.0f && sum != 0.0f)
        for (int i = 0; i < f->width; i++) f->coeff[i] /= sum;
}

void orc_fir_triangle(float sub, float offset, orc_fir *f) {              /* filter.c:24-76 */
    const float width = (sub < 1.0f) ? (1.0f / sub) : sub;
    if (!fir_extent(width, offset, f)) return;
    float sum = 0.0f;
    for (int i = 0; i < f->width; i++) {
        f->coeff[i] = 1.0f - fabsf((1.0f / width) * ((i - f->center) - offset));
        sum += f->coeff[i];
    }
    fir_normalise(sub, sum, f);
}

void orc_fir_lanczos(float sub, int kernel_size, float offset, orc_fir *f) { /* filter.c:78-148 */
    const float width = (sub < 1.0f) ? (1.0f / sub) : sub;
    if (!fir_extent(kernel_size * width, offset, f)) return;
    float sum = 0.0f;
    const double pi = 3.1415926535897932384626433832795028841971693993751;  /* G_PI */
    for (int i = 0; i < f->width; i++) {
        double x = (1.0 / width) * ((i - f->center) - (double)offset);
        if (x == 0.0) f->coeff[i] = 1.0f;
        else if (x <= -kernel_size || x >= kernel_size) f->coeff[i] = 0.0f;
        else {
            double num = kernel_size * sin(pi * x) * sin(pi * x / kernel_size);
            double den = pi * pi * x * x;
            double r = num / den;
            f->coeff[i] = isfinite(r) ? (float)r : 1.0f;
        }
        sum += f->coeff[i];
    }
    fir_normalise(sub, sum, f);
}

void orc_fir_free(orc_fir *f) { free(f->coeff); f->coeff = NULL; }       /* filter.c:150-153 */
