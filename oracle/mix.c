/*
 * oracle/mix.c -- frame pull dispatch, copy, crossfade, alpha-over, workspace stack.
 * TEST INFRASTRUCTURE (see oracle.h).  Restates:
 *   src/cprocess/main.c:33-76,105-144        (video_get_frame_f16 / _f32)
 *   src/cprocess/video_mix.c:27-44,46-71,73-105,107-235,237-370
 *   src/cprocess/workspace.c:494-550         (bottom-to-top over stack)
 * The reference's window quirk -- `left` is picked by comparing min.x against
 * the other frame's min.Y (video_mix.c:137,265) -- is kept on purpose.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline float clamp01(float v) { float t = v > 0.0f ? v : 0.0f; return t < 1.0f ? t : 1.0f; }

static inline void box_empty(orc_box2i *b) { b->min.x = 0; b->min.y = 0; b->max.x = -1; b->max.y = -1; }   /* framework.h:96-98 */
static inline int box_is_empty(const orc_box2i *b) { return b->max.x < b->min.x || b->max.y < b->min.y; }  /* :100-102 */
static inline orc_box2i box_and(const orc_box2i *a, const orc_box2i *b) {                                   /* :104-109 */
    orc_box2i r = { { imax(a->min.x, b->min.x), imax(a->min.y, b->min.y) }, { imin(a->max.x, b->max.x), imin(a->max.y, b->max.y) } };
    return r;
}
static inline orc_box2i box_or(const orc_box2i *a, const orc_box2i *b) {                                    /* :111-116 */
    orc_box2i r = { { imin(a->min.x, b->min.x), imin(a->min.y, b->min.y) }, { imax(a->max.x, b->max.x), imax(a->max.y, b->max.y) } };
    return r;
}
/* framework.h:118-132: an inverted axis becomes the gap between the two spans */
static inline void box_normalize(orc_box2i *b) {
    if (b->min.x > b->max.x) { int t = b->min.x - 1; b->min.x = b->max.x + 1; b->max.x = t; }
    if (b->min.y > b->max.y) { int t = b->min.y - 1; b->min.y = b->max.y + 1; b->max.y = t; }
}
static inline size_t box_area(const orc_box2i *b) {                                                        /* :134-137 */
    size_t w = b->max.x < b->min.x ? 0 : (size_t)(b->max.x - b->min.x + 1);
    size_t h = b->max.y < b->min.y ? 0 : (size_t)(b->max.y - b->min.y + 1);
    return w * h;
}

#define PX(f, X, Y) (&(f)->data[(ptrdiff_t)((Y) - (f)->full_window.min.y) * \
                                ((f)->full_window.max.x - (f)->full_window.min.x + 1) + ((X) - (f)->full_window.min.x)])   /* :196-208 */

/* ---- pull dispatch: main.c ---- */

void orc_get_frame_f16(orc_source *src, int frame_index, orc_frame16 *frame) {   /* main.c:33-76 */
    if (!src || !src->funcs) { box_empty(&frame->current_window); return; }
    if (src->funcs->get_frame) { src->funcs->get_frame(src->obj, frame_index, frame); return; }
    if (!src->funcs->get_frame_32) { box_empty(&frame->current_window); return; }  /* GL branch removed */

    orc_frame32 tmp;
    tmp.data = malloc(sizeof(orc_px32) * box_area(&frame->full_window));
    tmp.full_window = frame->full_window;
    tmp.current_window = frame->full_window;
    src->funcs->get_frame_32(src->obj, frame_index, &tmp);
    if (!box_is_empty(&tmp.current_window)) {
        int n = tmp.current_window.max.x - tmp.current_window.min.x + 1;
        for (int y = tmp.current_window.min.y; n > 0 && y <= tmp.current_window.max.y; y++)
            orc_float_to_half(&PX(frame, tmp.current_window.min.x, y)->r, &PX(&tmp, tmp.current_window.min.x, y)->r, n * 4);
    }
    frame->current_window = tmp.current_window;
    free(tmp.data);
}

void orc_get_frame_f32(orc_source *src, int frame_index, orc_frame32 *frame) {   /* main.c:105-144 */
    if (!src || !src->funcs) { box_empty(&frame->current_window); return; }
    if (src->funcs->get_frame_32) { src->funcs->get_frame_32(src->obj, frame_index, frame); return; }
    if (!src->funcs->get_frame) { box_empty(&frame->current_window); return; }

    orc_frame16 tmp;
    tmp.data = malloc(sizeof(orc_px16) * box_area(&frame->full_window));
    tmp.full_window = frame->full_window;
    tmp.current_window = frame->full_window;
    src->funcs->get_frame(src->obj, frame_index, &tmp);
    int n = tmp.current_window.max.x - tmp.current_window.min.x + 1;
    for (int y = tmp.current_window.min.y; y <= tmp.current_window.max.y; y++)
        orc_half_to_float(&PX(frame, tmp.current_window.min.x, y)->r, &PX(&tmp, tmp.current_window.min.x, y)->r, n * 4);
    frame->current_window = tmp.current_window;
    free(tmp.data);
}

/* ---- copies: video_mix.c:27-44, 73-105 ---- */

void orc_copy_frame_f16(orc_frame16 *out, orc_frame16 *in) {
    orc_box2i inner = box_and(&out->full_window, &in->current_window);
    out->current_window = inner;
    if (box_is_empty(&inner)) return;
    size_t n = (size_t)(inner.max.x - inner.min.x + 1);
    for (int y = inner.min.y; y <= inner.max.y; y++)
        memcpy(PX(out, inner.min.x, y), PX(in, inner.min.x, y), n * sizeof(orc_px16));
}

void orc_copy_frame_alpha_f32(orc_frame32 *out, orc_frame32 *in, float alpha) {
    alpha = clamp01(alpha);
    if (out == in && alpha == 1.0f) return;
    if (alpha == 0.0f) { box_empty(&out->current_window); return; }
    orc_box2i inner = box_and(&out->full_window, &in->current_window);
    out->current_window = inner;
    if (box_is_empty(&inner)) return;
    int n = inner.max.x - inner.min.x + 1;
    for (int y = inner.min.y; y <= inner.max.y; y++) {
        orc_px32 *d = PX(out, inner.min.x, y), *s = PX(in, inner.min.x, y);
        memmove(d, s, (size_t)n * sizeof(orc_px32));
        if (alpha != 1.0f)
            for (int x = 0; x < n; x++) d[x].a *= alpha;
    }
}

/* ---- the two-input mixers share one nine-region walk (video_mix.c:127-234, 255-369) ----
 * `p` is the frame written to and, for over, also the lower layer; `q` the other input.
 * wp / wq: alpha weights applied when a lone frame's pixels are copied.
 * p_in_place: true for over (p's own pixels are already in `out`, so they are left alone). */
typedef void (*blend_fn)(orc_px32 *o, const orc_px32 *p, const orc_px32 *q, float mp, float mq);

static void blend_cross(orc_px32 *o, const orc_px32 *a, const orc_px32 *b, float mix_a, float mix_b) {  /* :193-205 */
    float alpha_a = a->a * mix_a;
    float alpha_b = b->a * mix_b;
    float oa = alpha_a + alpha_b;
    if (oa != 0.0f) {
        float r = (a->r * alpha_a + b->r * alpha_b) / oa;
        float g = (a->g * alpha_a + b->g * alpha_b) / oa;
        float bl = (a->b * alpha_a + b->b * alpha_b) / oa;
        o->r = r; o->g = g; o->b = bl; o->a = oa;
    } else { o->r = o->g = o->b = o->a = 0.0f; }
}

static void blend_over(orc_px32 *o, const orc_px32 *lower, const orc_px32 *b, float unused, float mix_b) { /* :323-337 */
    (void)unused;
    float alpha_b = b->a * mix_b;
    float alpha_a = lower->a * (1.0f - b->a * mix_b);
    float oa = alpha_a + alpha_b;
    if (oa != 0.0f) {
        float r = (lower->r * alpha_a + b->r * alpha_b) / oa;
        float g = (lower->g * alpha_a + b->g * alpha_b) / oa;
        float bl = (lower->b * alpha_a + b->b * alpha_b) / oa;
        o->r = r; o->g = g; o->b = bl; o->a = oa;
    } else { o->r = o->g = o->b = o->a = 0.0f; }
}

static void lone_rows(orc_frame32 *out, orc_frame32 *src, float weight, int skip_copy,
                      int y0, int y1, int ox0, int ox1) {
    /* rows where only `src` is present: zero | src (alpha weighted) | zero   (:142-159, 217-232) */
    const orc_px32 z = { 0.0f, 0.0f, 0.0f, 0.0f };
    for (int y = y0; y <= y1; y++) {
        for (int x = ox0; x < src->current_window.min.x; x++) *PX(out, x, y) = z;
        if (!skip_copy)
            for (int x = src->current_window.min.x; x <= src->current_window.max.x; x++) {
                orc_px32 v = *PX(src, x, y);
                v.a *= weight;
                *PX(out, x, y) = v;
            }
        for (int x = src->current_window.max.x + 1; x <= ox1; x++) *PX(out, x, y) = z;
    }
}

static void mix_walk(orc_frame32 *out, orc_frame32 *p, orc_frame32 *q, float wp, float wq,
                     int p_in_place, blend_fn blend) {
    const orc_box2i *pw = &p->current_window, *qw = &q->current_window;
    const orc_px32 z = { 0.0f, 0.0f, 0.0f, 0.0f };

    orc_box2i outer = box_or(pw, qw);
    outer = box_and(&outer, &out->full_window);
    orc_box2i inner = box_and(pw, qw);
    inner = box_and(&inner, &out->full_window);
    int gap_x = inner.min.x > inner.max.x, gap_y = inner.min.y > inner.max.y;
    box_normalize(&inner);

    orc_frame32 *top = (pw->min.y < qw->min.y) ? p : q;
    orc_frame32 *bottom = (pw->max.y > qw->max.y) ? p : q;
    orc_frame32 *left = (pw->min.x < qw->min.y) ? p : q;        /* sic: min.y, as in the reference */
    orc_frame32 *right = (pw->max.x > qw->max.x) ? p : q;

    lone_rows(out, top, top == p ? wp : wq, p_in_place && top == p, outer.min.y, inner.min.y - 1, outer.min.x, outer.max.x);

    if (gap_y) {
        for (int y = inner.min.y; y <= inner.max.y; y++)
            for (int x = inner.min.x; x <= inner.max.x; x++) *PX(out, x, y) = z;
    } else {
        float wl = left == p ? wp : wq, wr = right == p ? wp : wq;
        for (int y = inner.min.y; y <= inner.max.y; y++) {
            if (!(p_in_place && left == p))
                for (int x = outer.min.x; x < inner.min.x; x++) { orc_px32 v = *PX(left, x, y); v.a *= wl; *PX(out, x, y) = v; }
            if (gap_x) {
                for (int x = inner.min.x; x <= inner.max.x; x++) *PX(out, x, y) = z;
            } else {
                for (int x = inner.min.x; x <= inner.max.x; x++)
                    blend(PX(out, x, y), PX(p, x, y), PX(q, x, y), wp, wq);
            }
            if (!(p_in_place && right == p))
                for (int x = inner.max.x + 1; x <= outer.max.x; x++) { orc_px32 v = *PX(right, x, y); v.a *= wr; *PX(out, x, y) = v; }
        }
    }

    lone_rows(out, bottom, bottom == p ? wp : wq, p_in_place && bottom == p, inner.max.y + 1, outer.max.y, outer.min.x, outer.max.x);
    out->current_window = outer;
}

void orc_mix_cross_f32(orc_frame32 *out, orc_frame32 *a, orc_frame32 *b, float mix_b) {   /* video_mix.c:107-235 */
    mix_b = clamp01(mix_b);
    const float mix_a = 1.0f - mix_b;
    if (box_is_empty(&a->current_window)) { orc_copy_frame_alpha_f32(out, b, mix_b); return; }
    if (box_is_empty(&b->current_window)) { orc_copy_frame_alpha_f32(out, a, mix_a); return; }
    mix_walk(out, a, b, mix_a, mix_b, 0, blend_cross);
}

void orc_mix_over_f32(orc_frame32 *out, orc_frame32 *b, float mix_b) {                     /* video_mix.c:237-370 */
    mix_b = clamp01(mix_b);
    if (box_is_empty(&out->current_window)) { orc_copy_frame_alpha_f32(out, b, mix_b); return; }
    if (box_is_empty(&b->current_window) || mix_b == 0.0f) return;
    mix_walk(out, out, b, 1.0f, mix_b, 1, blend_over);
}

void orc_mix_cross_f32_pull(orc_frame32 *out, orc_source *a, int frame_a, orc_source *b, int frame_b, float mix_b) {
    /* video_mix.c:46-71 */
    mix_b = clamp01(mix_b);
    if (mix_b == 0.0f) { orc_get_frame_f32(a, frame_a, out); return; }
    if (mix_b == 1.0f) { orc_get_frame_f32(b, frame_b, out); return; }
    orc_frame32 tmp;
    tmp.data = malloc(sizeof(orc_px32) * box_area(&out->full_window));
    tmp.full_window = out->full_window;
    box_empty(&tmp.current_window);
    orc_get_frame_f32(a, frame_a, out);
    orc_get_frame_f32(b, frame_b, &tmp);
    orc_mix_cross_f32(out, out, &tmp, mix_b);
    free(tmp.data);
}

/* ---- workspace stack: workspace.c:494-550 with :243-307's membership rule and cmpz (:102-105) ---- */

static int by_z(const void *pa, const void *pb) {
    const orc_ws_item *a = *(const orc_ws_item *const *)pa, *b = *(const orc_ws_item *const *)pb;
    return (a->z > b->z) - (a->z < b->z);
}

void orc_workspace_get_frame_f32(const orc_ws_item *items, int n, int frame_index, orc_frame32 *frame) {
    const orc_ws_item **live = malloc(sizeof(*live) * (size_t)(n > 0 ? n : 1));
    int m = 0;
    for (int i = 0; i < n; i++)
        if (items[i].x <= frame_index && frame_index < items[i].x + items[i].length) live[m++] = &items[i];
    if (!m) { box_empty(&frame->current_window); free(live); return; }
    qsort(live, (size_t)m, sizeof(*live), by_z);           /* lowest z is composited first */

    orc_get_frame_f32(live[0]->source, (int)(frame_index - live[0]->x + live[0]->offset), frame);
    if (m > 1) {
        orc_frame32 tmp;
        tmp.data = malloc(sizeof(orc_px32) * box_area(&frame->full_window));
        tmp.full_window = frame->full_window;
        for (int i = 1; i < m; i++) {
            box_empty(&tmp.current_window);
            orc_get_frame_f32(live[i]->source, (int)(frame_index - live[i]->x + live[i]->offset), &tmp);
            orc_mix_over_f32(frame, &tmp, 1.0f);
        }
        free(tmp.data);
    }
    free(live);
}

/* ---- 2:3 pulldown removal: src/process/Pulldown23RemovalFilter.c:43-107 ----
 * Cadence table of the reference (:51-57): source frames AA BB BC CD DD -> film frames A B C D, C woven from the
 * odd rows of BC and the even rows of CD. */
int orc_pulldown23_frames(int offset, int frame_index, int *first, int *second) {
    int frame_offset;
    if (offset == 4) frame_offset = (frame_index + 3) & 3;                  /* :61-64 */
    else frame_offset = (frame_index + offset) & 3;
    int base = ((frame_index + offset) >> 2) * 5 - offset;                  /* :66 */
    *second = base + 3;
    if (frame_offset == 0) { *first = base; return 0; }                     /* :69-77 */
    if (frame_offset == 1) { *first = base + 1; return 0; }
    if (frame_offset == 3) { *first = base + 4; return 0; }
    *first = base + 2;                                                      /* :80-81 */
    return 1;
}

/* :83-104 on two pulled frames.  `other` is the reference's temp frame: allocated for frame->current_window.  The
 * reference takes its rows from x = 0 (:101), whatever the window's min.x; where that lands outside the allocation
 * (foreign memory there) or outside other's current window (uninitialised there) this restatement reads zero. */
void orc_weave_fields_f16(orc_frame16 *frame, const orc_frame16 *other) {
    const orc_box2i *cw = &frame->current_window;
    if (box_is_empty(cw)) return;
    const int height = cw->max.y - cw->min.y + 1, width = cw->max.x - cw->min.x + 1;
    const long long n = (long long)height * width;
    for (int i = ((cw->min.y + 1) & ~1); i <= cw->max.y; i += 2) {          /* :100 */
        orc_px16 *dst = PX(frame, cw->min.x, i);
        const long long row0 = (long long)(i - cw->min.y) * width + (0 - cw->min.x);    /* video_get_pixel_f16(&temp, 0, i) */
        for (int c = 0; c < width; c++) {
            const long long l = row0 + c;
            orc_px16 v = { 0, 0, 0, 0 };
            if (l >= 0 && l < n) {
                const int ty = cw->min.y + (int)(l / width), tx = cw->min.x + (int)(l % width);
                if (tx >= other->current_window.min.x && tx <= other->current_window.max.x &&
                    ty >= other->current_window.min.y && ty <= other->current_window.max.y) v = other->data[l];
            }
            dst[c] = v;
        }
    }
}

void orc_pulldown23_get_frame_f16(orc_source *source, int offset, int frame_index, orc_frame16 *frame) {
    if (!source) { box_empty(&frame->current_window); return; }             /* :45-49 */
    int first, second;
    const int mixed = orc_pulldown23_frames(offset, frame_index, &first, &second);
    orc_get_frame_f16(source, first, frame);
    if (!mixed) return;
    const orc_box2i *cw = &frame->current_window;
    size_t pixels = box_is_empty(cw) ? 0 : (size_t)(cw->max.y - cw->min.y + 1) * (size_t)(cw->max.x - cw->min.x + 1);
    orc_frame16 temp = { calloc(pixels ? pixels : 1, sizeof(orc_px16)), *cw, *cw };     /* :92-95 */
    orc_get_frame_f16(source, second, &temp);                                           /* :97 */
    orc_weave_fields_f16(frame, &temp);
    free(temp.data);
}
