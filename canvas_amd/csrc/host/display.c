/*
 * display.c -- the display / export edge: an f16 frame to 4 bytes per pixel on the device.
 *
 * The reference has three copies of this loop, all through the gamma-0.45 byte ramp (gammatab.c:13-38):
 *   src/cprocess/widget_gl.c:291-307    video_transfer_linear_to_sRGB over all four halfs, then ramp -> rgba_u8
 *   src/libav/writeVideo.c:328-340      ramp -> rgba_u8
 *   src/process/RgbaFrameF16.c:114-149  ramp -> premultiplied ARGB32 (to_argb32_bytes)
 * Here they are one kernel (kernels/display_ops.hip) over one 64 KiB byte table per choice of transfer table:
 * table[c] = ramp[transfer[c]] is index plumbing done once on the host and cached on the device; it is rebuilt
 * when cvs_lut_install replaces the transfer table it was made from.
 */
#include "internal.h"
#include <pthread.h>

unsigned cvs_lut_generation(int which);       /* halfconv.c: bumped by every install */

static pthread_mutex_t disp_lock = PTHREAD_MUTEX_INITIALIZER;
static uint8_t *disp_dev[CVS_LUT_COUNT + 1];  /* slot 0: the bare ramp; slot 1 + id: ramp after table id */
static unsigned disp_gen[CVS_LUT_COUNT + 1];
static bool disp_have[CVS_LUT_COUNT + 1];

static const uint8_t *display_table(int pre_lut) {
    if (pre_lut != CVS_LUT_NONE && (pre_lut < 0 || pre_lut >= CVS_LUT_COUNT)) { cvs_set_error("no such transfer table: %d", pre_lut); return NULL; }
    const uint8_t *ramp = video_get_gamma45_ramp();
    const half *pre = pre_lut == CVS_LUT_NONE ? NULL : cvs_lut_host(pre_lut);
    if (!ramp || (pre_lut != CVS_LUT_NONE && !pre)) return NULL;
    const int slot = pre_lut == CVS_LUT_NONE ? 0 : 1 + pre_lut;
    const unsigned gen = pre_lut == CVS_LUT_NONE ? 0 : cvs_lut_generation(pre_lut);
    const uint8_t *result = NULL;
    pthread_mutex_lock(&disp_lock);
    if (!disp_have[slot] || disp_gen[slot] != gen) {
        uint8_t *host = malloc(HALF_COUNT);
        bool ok = host != NULL;
        if (ok) for (int c = 0; c < HALF_COUNT; c++) host[c] = ramp[pre ? pre[c] : c];
        if (ok && !disp_dev[slot]) ok = hipMalloc((void **)&disp_dev[slot], HALF_COUNT) == hipSuccess;
        if (ok) ok = hipMemcpy(disp_dev[slot], host, HALF_COUNT, hipMemcpyHostToDevice) == hipSuccess;
        free(host);
        if (ok) { disp_have[slot] = true; disp_gen[slot] = gen; }
        else cvs_set_error("display table %d could not be built", pre_lut);
    }
    if (disp_have[slot] && disp_gen[slot] == gen) result = disp_dev[slot];
    pthread_mutex_unlock(&disp_lock);
    return result;
}

/* dst_dev: room for 4 bytes per pixel of frame->current_window, packed row by row */
CVS_EXPORT int cvs_frame_to_bytes_dev(void *dst_dev, const rgba_frame_f16 *frame, int pre_lut, int mode, cvs_stream_t stream) {
    if (cvs_enter() != 0) return -1;
    if (mode != CVS_DISPLAY_RGBA8 && mode != CVS_DISPLAY_ARGB32_PREMUL) { cvs_set_error("frame to bytes: unknown mode %d", mode); return -1; }
    if (box2i_is_empty(&frame->current_window)) return 0;
    if (!cvs_box_contains(&frame->full_window, &frame->current_window)) { cvs_set_error("frame to bytes: current window outside the buffer"); return -1; }
    const uint8_t *table = display_table(pre_lut);
    if (!table) return -1;
    CVS_KERNEL(cvk_display(dst_dev, cvs_view(frame->data, &frame->full_window), cvs_rect(&frame->current_window), table,
                           mode == CVS_DISPLAY_RGBA8 ? CVK_DISPLAY_RGBA8 : CVK_DISPLAY_ARGB32_PREMUL, cvs_cus(), cvs_pick_stream(stream)));
    return 0;
}

/* the same on a HOST frame into a HOST buffer: rows of the current window go up, bytes come back */
CVS_EXPORT int video_frame_to_bytes(void *dst_host, const rgba_frame_f16 *frame, int pre_lut, int mode) {
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
    if (rc == 0) { band.data = din.dev; rc = cvs_frame_to_bytes_dev(dout.dev, &band, pre_lut, mode, s); }
    if (rc == 0) rc = cvs_stage_out(&dout, dst_host, s);
    cvs_stage_free(&din); cvs_stage_free(&dout);
    return rc;
}
