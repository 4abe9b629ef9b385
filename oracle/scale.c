/*
 * oracle/scale.c -- separable triangle resampler, plus the repo-defined FIR blur
 * and Lanczos gather resampler.  TEST INFRASTRUCTURE (see oracle.h).  Restates:
 *   src/cprocess/video_scale.c:25-32    (zero fill)
 *   src/cprocess/video_scale.c:34-127   (vertical pass)
 *   src/cprocess/video_scale.c:129-229  (horizontal pass)
 *   src/cprocess/video_scale.c:231-286  (pass ordering + intermediate window)
 *   src/cprocess/video_scale.c:288-319  (pull variant)
 * Kept as the reference has them: the intermediate window is derived with
 * `* factor` (:257-262, :272-277), so a two-axis DOWNscale only covers part of
 * the target; accumulation is `t += s * c` in ascending source order, starting
 * from a zero-filled target.
 */
#include "oracle.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline size_t area(const orc_box2i *b) {
    size_t w = b->max.x < b->min.x ? 0 : (size_t)(b->max.x - b->min.x + 1);
    size_t h = b->max.y < b->min.y ? 0 : (size_t)(b->max.y - b->min.y + 1);
    return w * h;
}
#define PX(f, X, Y) (&(f)->data[(ptrdiff_t)((Y) - (f)->full_window.min.y) * \
                                ((f)->full_window.max.x - (f)->full_window.min.x + 1) + ((X) - (f)->full_window.min.x)])

static void zero_fill(orc_frame32 *t) { memset(t->data, 0, area(&t->full_window) * sizeof(orc_px32)); }

static inline void acc(orc_px32 *t, const orc_px32 *s, float c) {         /* :82-85 */
    t->r += s->r * c; t->g += s->g * c; t->b += s->b * c; t->a += s->a * c;
}

/* a tap buffer big enough for any fractional offset (:52-57) */
static float *tap_buffer(float factor, int *cap) {
    float dummy = factor;
    orc_fir probe = { &dummy, 0, 0 };
    orc_fir_triangle(factor, 0.0f, &probe);      /* too small on purpose: reports the width */
    *cap = probe.width + 3;
    return malloc(sizeof(float) * (size_t)*cap);
}

/* axis: 0 = resample along y (rows), 1 = along x (columns) */
static void pass(orc_frame32 *target, float tmin, orc_frame32 *source, float smin, float factor, int axis) {
    const orc_box2i srect = source->current_window, trect = target->full_window;
    /* the axis NOT being resampled is simply clipped (:38-39, :136-137) */
    int lo = axis ? imax(srect.min.y, trect.min.y) : imax(srect.min.x, trect.min.x);
    int hi = axis ? imin(srect.max.y, trect.max.y) : imin(srect.max.x, trect.max.x);
    int s0 = axis ? srect.min.x : srect.min.y, s1 = axis ? srect.max.x : srect.max.y;
    int t0 = axis ? trect.min.x : trect.min.y, t1 = axis ? trect.max.x : trect.max.y;
    int used_lo = INT_MAX, used_hi = INT_MIN;

    zero_fill(target);
    if (factor == 1.0f && tmin == smin) { orc_copy_frame_alpha_f32(target, source, 1.0f); return; }

    int cap;
    float *taps = tap_buffer(factor, &cap);
    orc_fir f = { taps, 0, 0 };

    if (factor > 1.0f) {
        /* upscale: each source line is scattered into the target lines it touches (:63-92, :161-192) */
        for (int s = s0; s <= s1; s++) {
            float centre_f = (s - smin) * factor + tmin;
            int centre = (int)floor(centre_f);
            f.width = cap;
            orc_fir_triangle(factor, centre_f - centre, &f);
            for (int k = 0; k < f.width; k++) {
                int t = centre - f.center + k;
                if (t < t0 || t > t1) continue;
                if (!axis || lo <= hi) { used_lo = imin(used_lo, t); used_hi = imax(used_hi, t); }   /* :88-89 vs :186-187 */
                for (int o = lo; o <= hi; o++) {
                    if (axis) acc(PX(target, t, o), PX(source, s, o), taps[k]);
                    else      acc(PX(target, o, t), PX(source, o, s), taps[k]);
                }
            }
        }
    } else {
        /* downscale: each target line gathers the source lines under its footprint (:93-122, :193-226) */
        for (int t = t0; t <= t1; t++) {
            float centre_f = (t - tmin) / factor + smin;
            int centre = (int)floor(centre_f);
            f.width = cap;
            orc_fir_triangle(factor, centre_f - centre, &f);
            for (int k = 0; k < f.width; k++) {
                int s = centre - f.center + k;
                if (s < s0 || s > s1) continue;
                if (!axis || lo <= hi) { used_lo = imin(used_lo, t); used_hi = imax(used_hi, t); }   /* :88-89 vs :186-187 */
                for (int o = lo; o <= hi; o++) {
                    if (axis) acc(PX(target, t, o), PX(source, s, o), taps[k]);
                    else      acc(PX(target, o, t), PX(source, o, s), taps[k]);
                }
            }
        }
    }

    if (axis) { target->current_window.min.x = used_lo; target->current_window.min.y = lo;
                target->current_window.max.x = used_hi; target->current_window.max.y = hi; }
    else      { target->current_window.min.x = lo; target->current_window.min.y = used_lo;
                target->current_window.max.x = hi; target->current_window.max.y = used_hi; }
    free(taps);
}

void orc_scale_bilinear_f32(orc_frame32 *target, orc_v2f tp, orc_frame32 *source, orc_v2f sp, orc_v2f fac) {
    if (fac.x == 1.0f && tp.x == sp.x) {
        if (fac.y == 1.0f && tp.y == sp.y) { orc_copy_frame_alpha_f32(target, source, 1.0f); return; }
        pass(target, tp.y, source, sp.y, fac.y, 0);
        return;
    }
    if (fac.y == 1.0f && tp.y == sp.y) { pass(target, tp.x, source, sp.x, fac.x, 1); return; }

    orc_frame32 mid;
    const orc_box2i *tf = &target->full_window, *sc = &source->current_window;
    int x_first = fac.x < fac.y;                  /* smaller factor first (:252) */
    if (x_first) {
        mid.full_window.min.x = (int)(sp.x - (tp.x - tf->min.x) * fac.x);
        mid.full_window.min.y = sc->min.y;
        mid.full_window.max.x = (int)(sp.x + (tf->max.x - tp.x) * fac.x);
        mid.full_window.max.y = sc->max.y;
    } else {
        mid.full_window.min.x = sc->min.x;
        mid.full_window.min.y = (int)(sp.y - (tp.y - tf->min.y) * fac.y);
        mid.full_window.max.x = sc->max.x;
        mid.full_window.max.y = (int)(sp.y + (tf->max.y - tp.y) * fac.y);
    }
    mid.full_window.min.x = imax(mid.full_window.min.x, tf->min.x);
    mid.full_window.min.y = imax(mid.full_window.min.y, tf->min.y);
    mid.full_window.max.x = imin(mid.full_window.max.x, tf->max.x);
    mid.full_window.max.y = imin(mid.full_window.max.y, tf->max.y);
    mid.current_window = mid.full_window;
    size_t n = area(&mid.full_window);
    mid.data = malloc(sizeof(orc_px32) * (n ? n : 1));

    if (x_first) { pass(&mid, tp.x, source, sp.x, fac.x, 1); pass(target, tp.y, &mid, sp.y, fac.y, 0); }
    else         { pass(&mid, tp.y, source, sp.y, fac.y, 0); pass(target, tp.x, &mid, sp.x, fac.x, 1); }
    free(mid.data);
}

void orc_scale_bilinear_f32_pull(orc_frame32 *target, orc_v2f tp, orc_source *source, int frame,
                                 orc_box2i *source_rect, orc_v2f sp, orc_v2f fac) {
    if (fac.x == 0.0f || fac.y == 0.0f) {
        target->current_window.min.x = 0; target->current_window.min.y = 0;
        target->current_window.max.x = -1; target->current_window.max.y = -1;
        return;
    }
    if (fac.x == 1.0f && fac.y == 1.0f && tp.x == sp.x && tp.y == sp.y) { orc_get_frame_f32(source, frame, target); return; }

    const orc_box2i *tf = &target->full_window;
    orc_frame32 tmp;
    tmp.full_window.min.x = imax((int)(sp.x - (tp.x - tf->min.x) / fac.x) - 1, source_rect->min.x);   /* :303-309 */
    tmp.full_window.min.y = imax((int)(sp.y - (tp.y - tf->min.y) / fac.y) - 1, source_rect->min.y);
    tmp.full_window.max.x = imin((int)(sp.x + (tf->max.x - tp.x) / fac.x) + 1, source_rect->max.x);
    tmp.full_window.max.y = imin((int)(sp.y + (tf->max.y - tp.y) / fac.y) + 1, source_rect->max.y);
    tmp.current_window = tmp.full_window;
    size_t n = area(&tmp.full_window);
    tmp.data = malloc(sizeof(orc_px32) * (n ? n : 1));
    orc_get_frame_f32(source, frame, &tmp);
    orc_scale_bilinear_f32(target, tp, &tmp, sp, fac);
    free(tmp.data);
}

/* ---- repo-defined: no reference implementation exists; parity unpinned ----
 * Separable FIR at factor 1: horizontal then vertical, f32, `t += s * c` in ascending tap
 * order from zero, taps falling outside the source's current_window contribute nothing
 * (the skip rule of video_scale.c:106-107,211-212).  Output window = source window clipped
 * to the target's full window. */
static void fir_axis(orc_frame32 *dst, const orc_frame32 *src, const orc_box2i *win, const float *taps, int ntaps, int axis) {
    int c = ntaps / 2;
    for (int y = win->min.y; y <= win->max.y; y++)
        for (int x = win->min.x; x <= win->max.x; x++) {
            orc_px32 t = { 0.0f, 0.0f, 0.0f, 0.0f };
            for (int k = 0; k < ntaps; k++) {
                int sx = axis ? x - c + k : x, sy = axis ? y : y - c + k;
                if (sx < src->current_window.min.x || sx > src->current_window.max.x ||
                    sy < src->current_window.min.y || sy > src->current_window.max.y) continue;
                acc(&t, PX((orc_frame32 *)src, sx, sy), taps[k]);
            }
            *PX(dst, x, y) = t;
        }
}

void orc_fir_blur_f32(orc_frame32 *target, orc_frame32 *source, const float *taps, int ntaps) {
    orc_box2i win;
    win.min.x = imax(source->current_window.min.x, target->full_window.min.x);
    win.min.y = imax(source->current_window.min.y, target->full_window.min.y);
    win.max.x = imin(source->current_window.max.x, target->full_window.max.x);
    win.max.y = imin(source->current_window.max.y, target->full_window.max.y);
    target->current_window = win;
    if (win.max.x < win.min.x || win.max.y < win.min.y) return;
    /* horizontal into a scratch frame covering the SOURCE window (vertical taps need rows outside `win`) */
    orc_frame32 mid;
    mid.full_window = source->current_window;
    mid.current_window = source->current_window;
    mid.data = malloc(sizeof(orc_px32) * area(&mid.full_window));
    fir_axis(&mid, source, &mid.full_window, taps, ntaps, 1);
    fir_axis(target, &mid, &win, taps, ntaps, 0);
    free(mid.data);
}

/* Lanczos gather resample, taps from the restated filter_createLanczos with the per-line
 * fractional offset rule of video_scale.c:97-101; x pass then y pass; covers the whole target. */
static void lanczos_axis(orc_frame32 *dst, const orc_frame32 *src, float factor, int ksize, int axis) {
    const orc_box2i s = src->current_window, tf = dst->full_window;
    int lo = axis ? imax(s.min.y, tf.min.y) : imax(s.min.x, tf.min.x);
    int hi = axis ? imin(s.max.y, tf.max.y) : imin(s.max.x, tf.max.x);
    int t0 = axis ? tf.min.x : tf.min.y, t1 = axis ? tf.max.x : tf.max.y;
    int s0 = axis ? s.min.x : s.min.y, s1 = axis ? s.max.x : s.max.y;
    zero_fill(dst);
    for (int t = t0; t <= t1; t++) {
        float centre_f = (float)t / factor;
        int centre = (int)floor(centre_f);
        orc_fir f = { NULL, 0, 0 };
        orc_fir_lanczos(factor, ksize, centre_f - centre, &f);
        for (int k = 0; k < f.width; k++) {
            int sidx = centre - f.center + k;
            if (sidx < s0 || sidx > s1) continue;
            for (int o = lo; o <= hi; o++) {
                if (axis) acc(PX(dst, t, o), PX((orc_frame32 *)src, sidx, o), f.coeff[k]);
                else      acc(PX(dst, o, t), PX((orc_frame32 *)src, o, sidx), f.coeff[k]);
            }
        }
        orc_fir_free(&f);
    }
    if (axis) { dst->current_window.min.x = t0; dst->current_window.max.x = t1; dst->current_window.min.y = lo; dst->current_window.max.y = hi; }
    else      { dst->current_window.min.y = t0; dst->current_window.max.y = t1; dst->current_window.min.x = lo; dst->current_window.max.x = hi; }
}

void orc_resample_lanczos_f32(orc_frame32 *target, orc_frame32 *source, float fx, float fy, int ksize) {
    orc_frame32 mid;
    mid.full_window.min.x = target->full_window.min.x; mid.full_window.max.x = target->full_window.max.x;
    mid.full_window.min.y = source->current_window.min.y; mid.full_window.max.y = source->current_window.max.y;
    mid.current_window = mid.full_window;
    size_t n = area(&mid.full_window);
    mid.data = malloc(sizeof(orc_px32) * (n ? n : 1));
    lanczos_axis(&mid, source, fx, ksize, 1);
    lanczos_axis(target, &mid, fy, ksize, 0);
    free(mid.data);
}
