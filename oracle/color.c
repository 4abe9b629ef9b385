/*
 * oracle/color.c -- colour matrix on f16 frames, gain/offset, solid fill, and the
 * BASELINE config-2 chain composed from the restated pieces.
 * TEST INFRASTRUCTURE (see oracle.h).  Restates:
 *   src/cprocess/color.c:34-42        (mult_mat_xyz: left-to-right mul/add, alpha copied)
 *   src/cprocess/color.c:104-137      (pre-LUT -> widen -> matrix -> truncate)
 *   src/cprocess/color.c:140-165      (widen -> matrix -> truncate -> post-LUT)
 *   src/cprocess/video_filter.c:34-39 (gain/offset formula; GLSL only in the reference)
 *   src/cprocess/gl.c:584             (one-input filter window rule: out.full ∩ in.current)
 *   src/process/SolidColorVideoSource.c:52-101 (solid fill)
 * Both LUTs run over all four channels INCLUDING alpha (color.c:127,161).
 * Build with -ffp-contract=off: the reference's gcc -std=c99 build does not fuse mul+add.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
#define PX(f, X, Y) (&(f)->data[(ptrdiff_t)((Y) - (f)->full_window.min.y) * \
                                ((f)->full_window.max.x - (f)->full_window.min.x + 1) + ((X) - (f)->full_window.min.x)])

void orc_color_matrix_f16(orc_frame16 *frame, const float m[9], const orc_half *pre_lut, const orc_half *post_lut) {
    const orc_box2i *w = &frame->current_window;
    int n = w->max.x - w->min.x + 1;
    if (n <= 0 || w->max.y < w->min.y) return;
    orc_px32 *f = malloc(sizeof(orc_px32) * (size_t)n);
    for (int y = w->min.y; y <= w->max.y; y++) {
        orc_px16 *h = PX(frame, w->min.x, y);
        if (pre_lut) orc_half_lookup(pre_lut, &h->r, &h->r, n * 4);
        orc_half_to_float(&f->r, &h->r, n * 4);
        for (int x = 0; x < n; x++) {
            orc_px32 v = f[x], o;
            o.r = v.r * m[0] + v.g * m[3] + v.b * m[6];
            o.g = v.r * m[1] + v.g * m[4] + v.b * m[7];
            o.b = v.r * m[2] + v.g * m[5] + v.b * m[8];
            o.a = v.a;
            f[x] = o;
        }
        orc_float_to_half(&h->r, &f->r, n * 4);
        if (post_lut) orc_half_lookup(post_lut, &h->r, &h->r, n * 4);
    }
    free(f);
}

void orc_color_rgb_to_xyz_sdtv(orc_frame16 *frame) {                      /* color.c:104-137 */
    static const float m[9] = { 0.3936f, 0.2124f, 0.0187f,  0.3652f, 0.7010f, 0.1119f,  0.1916f, 0.0865f, 0.9582f };
    orc_color_matrix_f16(frame, m, orc_transfer_table(ORC_LUT_REC709_TO_LINEAR_SCENE), NULL);
}

void orc_color_xyz_to_srgb(orc_frame16 *frame) {                          /* color.c:140-165 */
    static const float m[9] = { 3.2410f, -0.9692f, 0.0556f,  -1.5374f, 1.8760f, -0.2040f,  -0.4986f, 0.0416f, 1.0570f };
    orc_color_matrix_f16(frame, m, NULL, orc_transfer_table(ORC_LUT_LINEAR_TO_SRGB));
}

/* Gain/offset exists only as a GLSL shader in the reference; its output rounding is whatever the
 * GL driver did.  The repo defines: widen exactly, `c * gain + offset` as two rounded f32 ops,
 * truncate like every other f32->f16 step of the path.  Rounding: PARITY UNPINNED. */
void orc_gain_offset_f16(orc_frame16 *out, orc_frame16 *in, float gain, float offset) {
    orc_box2i w;
    w.min.x = imax(out->full_window.min.x, in->current_window.min.x);
    w.min.y = imax(out->full_window.min.y, in->current_window.min.y);
    w.max.x = imin(out->full_window.max.x, in->current_window.max.x);
    w.max.y = imin(out->full_window.max.y, in->current_window.max.y);
    out->current_window = w;
    if (w.max.x < w.min.x || w.max.y < w.min.y) return;
    for (int y = w.min.y; y <= w.max.y; y++)
        for (int x = w.min.x; x <= w.max.x; x++) {
            float c[4];
            orc_half_to_float(c, &PX(in, x, y)->r, 4);
            c[0] = c[0] * gain + offset;
            c[1] = c[1] * gain + offset;
            c[2] = c[2] * gain + offset;
            orc_float_to_half(&PX(out, x, y)->r, c, 4);
        }
}

static orc_box2i solid_window(const orc_box2i *window, const orc_box2i *full) {
    orc_box2i w = { { imax(window->min.x, full->min.x), imax(window->min.y, full->min.y) },
                    { imin(window->max.x, full->max.x), imin(window->max.y, full->max.y) } };
    return w;
}

void orc_solid_f16(orc_frame16 *frame, const orc_box2i *window, const float color[4]) {   /* SolidColorVideoSource.c:52-77 */
    frame->current_window = solid_window(window, &frame->full_window);
    const orc_box2i *w = &frame->current_window;
    if (w->max.x < w->min.x || w->max.y < w->min.y) return;
    orc_px16 c;
    orc_float_to_half(&c.r, color, 4);
    for (int y = w->min.y; y <= w->max.y; y++)
        for (int x = w->min.x; x <= w->max.x; x++) *PX(frame, x, y) = c;
}

void orc_solid_f32(orc_frame32 *frame, const orc_box2i *window, const float color[4]) {   /* SolidColorVideoSource.c:79-101 */
    frame->current_window = solid_window(window, &frame->full_window);
    const orc_box2i *w = &frame->current_window;
    if (w->max.x < w->min.x || w->max.y < w->min.y) return;
    orc_px32 c = { color[0], color[1], color[2], color[3] };
    for (int y = w->min.y; y <= w->max.y; y++)
        for (int x = w->min.x; x <= w->max.x; x++) *PX(frame, x, y) = c;
}

/* ---- BASELINE config 2 as the reference would run it, node by node ----
 * per layer: colour filter in place on a copy of the f16 source (color.c structure) ->
 * workspace stack pulls each layer as f32 (main.c:105-144 widening), lowest layer straight into
 * the output, every further layer through video_mix_over_f32(mix 1.0) (workspace.c:530-544) ->
 * the consumer's video_get_frame_f16 truncates the f32 result (main.c:43-71). */
typedef struct { orc_frame16 *frame; } layer_src;

static void layer_get16(void *self, int idx, orc_frame16 *out) {
    (void)idx;
    orc_copy_frame_f16(out, ((layer_src *)self)->frame);
}

typedef struct { const orc_ws_item *items; int n; } stack_src;

static void stack_get32(void *self, int idx, orc_frame32 *out) {
    stack_src *s = self;
    orc_workspace_get_frame_f32(s->items, s->n, idx, out);
}

void orc_chain_color_over_f16(orc_frame16 *out, orc_frame16 *const *layers, int nlayers,
                              const float m[9], const orc_half *pre_lut, const orc_half *post_lut) {
    orc_source_funcs lf = { 0, layer_get16, NULL, NULL }, sf = { 0, NULL, stack_get32, NULL };
    orc_frame16 *graded = calloc((size_t)nlayers, sizeof(orc_frame16));
    layer_src *ls = calloc((size_t)nlayers, sizeof(layer_src));
    orc_source *srcs = calloc((size_t)nlayers, sizeof(orc_source));
    orc_ws_item *items = calloc((size_t)nlayers, sizeof(orc_ws_item));
    for (int k = 0; k < nlayers; k++) {
        const orc_box2i *fw = &layers[k]->full_window;
        size_t n = (size_t)(fw->max.x - fw->min.x + 1) * (size_t)(fw->max.y - fw->min.y + 1);
        graded[k] = *layers[k];
        graded[k].data = malloc(n * sizeof(orc_px16));
        memcpy(graded[k].data, layers[k]->data, n * sizeof(orc_px16));
        if (m) orc_color_matrix_f16(&graded[k], m, pre_lut, post_lut);     /* m == NULL: the layers go to the stack as they are */
        ls[k].frame = &graded[k];
        srcs[k].obj = &ls[k]; srcs[k].funcs = &lf;
        items[k].x = 0; items[k].length = 1; items[k].z = k; items[k].offset = 0; items[k].source = &srcs[k];
    }
    stack_src st = { items, nlayers };
    orc_source stack = { &st, &sf };
    orc_get_frame_f16(&stack, 0, out);
    for (int k = 0; k < nlayers; k++) free(graded[k].data);
    free(graded); free(ls); free(srcs); free(items);
}

/* ---- display / export edge ----
 * Every reference edge is: optional transfer table over all four halfs, a 65536-entry half->u8 ramp, then packing.
 * mode 0: bytes r,g,b,a.  With the gamma-0.45 ramp and pre == NULL this is the exporter's conversion
 *   (src/libav/writeVideo.c:328-340); with pre = the linear->sRGB table and the widget's ramp, the software widget's
 *   (src/cprocess/widget_gl.c:291-307).
 * mode 1: premultiplied ARGB32 (src/process/RgbaFrameF16.c:114-149, gamma-0.45 ramp).
 * dst is packed over current_window. */
static void frame_to_bytes(uint32_t *dst, const orc_frame16 *frame, const orc_half *pre, const uint8_t *ramp, int mode) {
    const orc_box2i *w = &frame->current_window;
    if (w->max.x < w->min.x || w->max.y < w->min.y) return;
    const int width = w->max.x - w->min.x + 1;
    for (int y = w->min.y; y <= w->max.y; y++) {
        const orc_px16 *row = PX((orc_frame16 *)frame, w->min.x, y);
        for (int x = 0; x < width; x++) {
            orc_px16 p = row[x];
            if (pre) { p.r = pre[p.r]; p.g = pre[p.g]; p.b = pre[p.b]; p.a = pre[p.a]; }
            const uint32_t r = ramp[p.r], g = ramp[p.g], b = ramp[p.b], a = ramp[p.a];
            uint32_t *o = &dst[(size_t)(y - w->min.y) * (size_t)width + (size_t)x];
            if (mode == 0) {
                uint8_t bytes[4] = { (uint8_t)r, (uint8_t)g, (uint8_t)b, (uint8_t)a };    /* rgba_u8 in memory order */
                memcpy(o, bytes, 4);
            } else {
                *o = (a << 24) | (((r * a >> 8) & 0xFF) << 16) | (((g * a >> 8) & 0xFF) << 8) | ((b * a >> 8) & 0xFF);
            }
        }
    }
}

void orc_frame_to_bytes(uint32_t *dst, const orc_frame16 *frame, const orc_half *pre, int mode) {
    frame_to_bytes(dst, frame, pre, orc_gamma45_ramp(), mode);
}

/* widget_gl.c:291-307: transfer table, then the widget's ramp, bytes r,g,b,a */
void orc_frame_to_rgba8_intent(uint32_t *dst, const orc_frame16 *frame, const orc_half *pre, float rendering_intent) {
    uint8_t *ramp = malloc(65536);
    orc_widget_ramp(ramp, rendering_intent);
    frame_to_bytes(dst, frame, pre, ramp, 0);
    free(ramp);
}
