/*
 * display.c -- the display / export edge: an f16 frame to 4 bytes per pixel on the device.
 *
 * The reference turns pulled f16 frames into bytes in three places, each through a 65536-entry half -> u8 ramp:
 *   src/process/RgbaFrameF16.c:114-149  the gamma-0.45 ramp of gammatab.c:13-38 -> premultiplied ARGB32 (to_argb32_bytes)
 *   src/libav/writeVideo.c:108-115,328-340  its own ramp, (int)(clamp(x,0,1)^0.45 * 255): the same bytes as gammatab's -> rgba_u8
 *   src/cprocess/widget_gl.c:291-307    video_transfer_linear_to_sRGB over all four halfs, then the WIDGET's ramp
 *                                       lrint(clamp(x^intent * 255, 0, 255)) (:947-968, intent 1.25 by default) -> rgba_u8
 * Here they are one kernel (kernels/display_ops.hip) over one 64 KiB byte table per (transfer table, ramp) pair:
 * table[c] = ramp[transfer[c]] is index plumbing done once on the host and cached on the device; it is rebuilt
 * when cvs_lut_install replaces the transfer table it was made from.
 */
#include "internal.h"
#include <math.h>
#include <pthread.h>

unsigned cvs_lut_generation(int which);       /* halfconv.c: bumped by every install */
const float *cvs_codes_as_float(void);        /* halfconv.c */

enum { RAMP_GAMMA45 = 0, RAMP_INTENT = 1 };
typedef struct { bool have; int pre_lut, kind, pins; unsigned gen; uint32_t intent_bits; uint64_t stamp; uint8_t *dev; } disp_entry;     /* pins: captured graphs whose kernels read the table */
/* one cache per device context (runtime.c): the tables live on the device */
#define DISP_CACHE 8
static pthread_mutex_t disp_lock = PTHREAD_MUTEX_INITIALIZER;
static disp_entry disp_cache_of[CVS_MAX_CONTEXTS][DISP_CACHE];
static uint64_t disp_clock;
#define disp_cache (disp_cache_of[cvs_ctx()])

/* Call with disp_lock held, and keep it until the kernel that reads the table has been enqueued: an eviction waits for
 * the device (under the same lock) before it overwrites a slot, so a table is never rewritten under a launch that was
 * handed its address. */
static const uint8_t *display_table(int pre_lut, int kind, float intent, int *slot) {
    if (pre_lut != CVS_LUT_NONE && (pre_lut < 0 || pre_lut >= CVS_LUT_COUNT)) { cvs_set_error("no such transfer table: %d", pre_lut); return NULL; }
    const uint8_t *ramp45 = kind == RAMP_GAMMA45 ? video_get_gamma45_ramp() : NULL;
    const float *codes = kind == RAMP_INTENT ? cvs_codes_as_float() : NULL;
    const half *pre = pre_lut == CVS_LUT_NONE ? NULL : cvs_lut_host(pre_lut);
    if ((kind == RAMP_GAMMA45 && !ramp45) || (kind == RAMP_INTENT && !codes) || (pre_lut != CVS_LUT_NONE && !pre)) return NULL;
    const unsigned gen = pre_lut == CVS_LUT_NONE ? 0 : cvs_lut_generation(pre_lut);
    uint32_t ibits = 0;
    if (kind == RAMP_INTENT) memcpy(&ibits, &intent, 4);
    const uint8_t *result = NULL;
    int victim = -1;
    for (int i = 0; i < DISP_CACHE && !result; i++) {
        disp_entry *e = &disp_cache[i];
        if (e->have && e->pre_lut == pre_lut && e->kind == kind && e->gen == gen && e->intent_bits == ibits) { e->stamp = ++disp_clock; result = e->dev; *slot = i; }
        else if (e->have && e->pins > 0) continue;
        else if (victim < 0 || (disp_cache[victim].have && (!e->have || e->stamp < disp_cache[victim].stamp))) victim = i;
    }
    if (!result && victim < 0) { cvs_set_error("display tables: every cache slot belongs to a captured graph"); return NULL; }
    if (!result) {
        uint8_t *host = malloc(HALF_COUNT);
        bool ok = host != NULL;
        for (int c = 0; ok && c < HALF_COUNT; c++) {
            const int code = pre ? pre[c] : c;
            if (kind == RAMP_GAMMA45) host[c] = ramp45[code];
            else host[c] = (uint8_t)lrint(clampf(powf(codes[code], intent) * 255.0f, 0.0f, 255.0f));     /* widget_gl.c:966 */
        }
        disp_entry *e = &disp_cache[victim];
        if (ok && e->have) { (void)hipDeviceSynchronize(); }         /* a launch may still be reading the evicted table */
        if (ok && !e->dev) ok = hipMalloc((void **)&e->dev, HALF_COUNT) == hipSuccess;
        if (ok) ok = hipMemcpy(e->dev, host, HALF_COUNT, hipMemcpyHostToDevice) == hipSuccess;
        free(host);
        if (ok) { e->have = true; e->pre_lut = pre_lut; e->kind = kind; e->gen = gen; e->intent_bits = ibits; e->stamp = ++disp_clock; result = e->dev; *slot = victim; }
        else { e->have = false; cvs_set_error("display table (transfer %d) could not be built", pre_lut); }
    }
    return result;
}

/* `slot`: context * DISP_CACHE + entry */
static void display_unpin(void *slot) {
    const int id = (int)(intptr_t)slot;
    pthread_mutex_lock(&disp_lock);
    disp_cache_of[id / DISP_CACHE][id % DISP_CACHE].pins--;
    pthread_mutex_unlock(&disp_lock);
}

static int to_bytes_dev(void *dst_dev, const rgba_frame_f16 *frame, int pre_lut, int ramp, float intent, int kmode, hipStream_t s) {
    if (box2i_is_empty(&frame->current_window)) return 0;
    if (!cvs_box_contains(&frame->full_window, &frame->current_window)) { cvs_set_error("frame to bytes: current window outside the buffer"); return -1; }
    pthread_mutex_lock(&disp_lock);
    int slot = -1;
    const uint8_t *table = display_table(pre_lut, ramp, intent, &slot);
    int rc = table ? cvk_display(dst_dev, cvs_view(frame->data, &frame->full_window), cvs_rect(&frame->current_window), table, kmode, cvs_cus(), s) : -1;
    /* recorded into a graph: the table stays where it is until the graph is destroyed */
    if (table && rc == 0 && cvs_capture_hold(s, display_unpin, (void *)(intptr_t)(cvs_ctx() * DISP_CACHE + slot))) disp_cache[slot].pins++;
    pthread_mutex_unlock(&disp_lock);
    if (table && rc != 0) { cvs_set_error("frame to bytes: launch failed: %s", hipGetErrorString((hipError_t)rc)); return -1; }
    return rc;
}

/* the software widget's conversion (widget_gl.c:291-307): transfer table over all four halfs, then the ramp
 * lrint(clamp(x^rendering_intent * 255)) -> bytes r,g,b,a.  The widget's defaults: CVS_LUT_LINEAR_TO_SRGB, 1.25. */
CVS_EXPORT int cvs_frame_to_rgba8_intent_dev(void *dst_dev, const rgba_frame_f16 *frame, int pre_lut, float rendering_intent, cvs_stream_t stream) {
    if (cvs_enter() != 0) return -1;
    if (box2i_is_empty(&frame->current_window)) return 0;
    return to_bytes_dev(dst_dev, frame, pre_lut, RAMP_INTENT, rendering_intent, CVK_DISPLAY_RGBA8, cvs_pick_stream(stream));
}

/* dst_dev: room for 4 bytes per pixel of frame->current_window, packed row by row */
CVS_EXPORT int cvs_frame_to_bytes_dev(void *dst_dev, const rgba_frame_f16 *frame, int pre_lut, int mode, cvs_stream_t stream) {
    if (cvs_enter() != 0) return -1;
    if (mode != CVS_DISPLAY_RGBA8 && mode != CVS_DISPLAY_ARGB32_PREMUL) { cvs_set_error("frame to bytes: unknown mode %d", mode); return -1; }
    if (box2i_is_empty(&frame->current_window)) return 0;
    return to_bytes_dev(dst_dev, frame, pre_lut, RAMP_GAMMA45, 0.0f, mode == CVS_DISPLAY_RGBA8 ? CVK_DISPLAY_RGBA8 : CVK_DISPLAY_ARGB32_PREMUL, cvs_pick_stream(stream));
}

/* the same on a HOST frame into a HOST buffer: rows of the current window go up, bytes come back.
 * kind < 0: gamma-0.45 ramp in `mode`; otherwise the widget's ramp with that rendering intent. */
static int host_to_bytes(void *dst_host, const rgba_frame_f16 *frame, int pre_lut, int mode, bool widget, float intent) {
    if (cvs_enter() != 0) return -1;
    const box2i *w = &frame->current_window;
    if (box2i_is_empty(w)) return 0;
    if (!cvs_box_contains(&frame->full_window, w)) { cvs_set_error("frame to bytes: current window outside the buffer"); return -1; }
    hipStream_t s = cvs_pick_stream(NULL);
    /* only the rows of the current window travel: a device frame whose full window is that band of rows */
    rgba_frame_f16 band = *frame;
    band.full_window.min.y = w->min.y; band.full_window.max.y = w->max.y;
    const size_t pitch = (size_t)(frame->full_window.max.x - frame->full_window.min.x + 1);
    const rgba_f16 *first = frame->data + (size_t)(w->min.y - frame->full_window.min.y) * pitch;
    const size_t in_bytes = cvs_box_pixels(&band.full_window) * sizeof(rgba_f16), out_bytes = cvs_box_pixels(w) * 4;
    cvs_staged din = { 0 }, dout = { 0 };
    int rc = cvs_stage_in(&din, first, in_bytes, 1, s);
    if (rc == 0) rc = cvs_stage_in(&dout, NULL, out_bytes, 0, s);
    if (rc == 0) {
        band.data = din.dev;
        rc = widget ? cvs_frame_to_rgba8_intent_dev(dout.dev, &band, pre_lut, intent, s) : cvs_frame_to_bytes_dev(dout.dev, &band, pre_lut, mode, s);
    }
    if (rc == 0) rc = cvs_stage_out(&dout, dst_host, s);
    cvs_stage_free(&din); cvs_stage_free(&dout);
    return rc;
}

CVS_EXPORT int video_frame_to_bytes(void *dst_host, const rgba_frame_f16 *frame, int pre_lut, int mode) {
    return host_to_bytes(dst_host, frame, pre_lut, mode, false, 0.0f);
}

CVS_EXPORT int video_frame_to_rgba8_intent(void *dst_host, const rgba_frame_f16 *frame, int pre_lut, float rendering_intent) {
    return host_to_bytes(dst_host, frame, pre_lut, CVS_DISPLAY_RGBA8, true, rendering_intent);
}
