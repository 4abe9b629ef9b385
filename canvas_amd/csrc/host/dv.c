/*
 * dv.c -- DV 4:1:1 edge of the path: coded_image planes <-> half RGBA frames.
 *
 *   coded_image, coded_image_alloc/alloc0 ... include/framework.h:468-508, src/cprocess/video_subsample.c:23-68
 *   video_reconstruct_dv .................... src/cprocess/video_reconstruct.c:50-137
 *   video_subsample_dv ...................... src/cprocess/video_subsample.c:99-187
 * The raster is fixed: 720x480, first line at y = -1 on the frame plane, chroma 180 samples per line.
 * Host entry points stage planes / frame rows to the device and back; the cvs_*_dev twins work on device
 * planes and device frames.  The triangle taps come from filter_createTriangle, exactly as the reference
 * asks for them (sub = 4 for reconstruction, 1/4 for subsampling), and travel as kernel arguments.
 */
#include "internal.h"

enum { DV_W = 720, DV_H = 480, DV_SUB = 4, DV_OFF_Y = -1 };

static void coded_image_release(void *p) {
    coded_image *image = p;
    if (!image) return;
    for (int i = 0; i < CODED_IMAGE_MAX_PLANES; i++) free(image->data[i]);
    free(image);
}

static coded_image *image_alloc(const int *strides, const int *line_counts, int count, bool zero) {
    if (!strides || !line_counts || count < 0 || count > CODED_IMAGE_MAX_PLANES) return NULL;
    coded_image *image = calloc(1, sizeof *image);
    if (!image) return NULL;
    for (int i = 0; i < count; i++) {
        if (strides[i] < 0 || line_counts[i] < 0) { coded_image_release(image); return NULL; }
        image->stride[i] = strides[i];
        image->line_count[i] = line_counts[i];
        const size_t bytes = (size_t)strides[i] * (size_t)line_counts[i];
        if (!bytes) continue;
        image->data[i] = zero ? calloc(1, bytes) : malloc(bytes);
        if (!image->data[i]) { coded_image_release(image); return NULL; }
    }
    image->free_func = coded_image_release;
    return image;
}

CVS_EXPORT coded_image *coded_image_alloc(const int *strides, const int *line_counts, int count) { return image_alloc(strides, line_counts, count, false); }
CVS_EXPORT coded_image *coded_image_alloc0(const int *strides, const int *line_counts, int count) { return image_alloc(strides, line_counts, count, true); }

static int dv_taps(float sub, cvk_dv_taps *t) {
    memset(t, 0, sizeof *t);
    fir_filter f = { t->coeff, 16, 0 };
    filter_createTriangle(sub, 0.0f, &f);
    if (f.center < 0 || f.width > 16) { cvs_set_error("DV: triangle filter does not fit (%d taps)", f.width); return -1; }
    t->width = f.width; t->center = f.center;
    return 0;
}

static bool dv_planes_ok(const coded_image *p) {
    return p && p->data[0] && p->data[1] && p->data[2] && p->stride[0] >= DV_W && p->stride[1] >= DV_W / DV_SUB && p->stride[2] >= DV_W / DV_SUB &&
           p->line_count[0] >= DV_H && p->line_count[1] >= DV_H && p->line_count[2] >= DV_H;
}

static cvk_dv_planes dv_view(const coded_image *p) {
    cvk_dv_planes v = { p->data[0], p->data[1], p->data[2], p->stride[0], p->stride[1], p->stride[2] };
    return v;
}

/* frame and planes on the device */
CVS_EXPORT int cvs_reconstruct_dv_dev(rgba_frame_f16 *frame, const coded_image *planar, cvs_stream_t stream) {
    if (cvs_enter() != 0) { box2i_set_empty(&frame->current_window); return -1; }
    if (!dv_planes_ok(planar)) { cvs_set_error("DV reconstruct: need three 720x480 / 180x480 planes"); box2i_set_empty(&frame->current_window); return -1; }
    box2i_set(&frame->current_window, max(0, frame->full_window.min.x), max(DV_OFF_Y, frame->full_window.min.y),
              min(DV_W - 1, frame->full_window.max.x), min(DV_H + DV_OFF_Y - 1, frame->full_window.max.y));
    if (box2i_is_empty(&frame->current_window)) return 0;
    cvk_dv_taps tri;
    const half *lut = cvs_lut_device(CVS_LUT_REC709_TO_LINEAR_SCENE);
    if (dv_taps((float)DV_SUB, &tri) != 0 || !lut) { box2i_set_empty(&frame->current_window); return -1; }
    cvk_dv_planes pl = dv_view(planar);
    CVS_KERNEL(CVK(cvk_dv_reconstruct)(cvs_view(frame->data, &frame->full_window), cvs_rect(&frame->current_window), &pl, &tri, lut, cvs_pick_stream(stream)));
    return 0;
}

/* planes are written whole (zero outside the frame's window, coded_image_alloc0 in the reference);
 * encode_input_in_place: leave the frame's rows transfer-encoded, as video_subsample.c:144 does */
CVS_EXPORT int cvs_subsample_dv_dev(coded_image *planar, rgba_frame_f16 *frame, int encode_input_in_place, cvs_stream_t stream) {
    if (cvs_enter() != 0) return -1;
    if (!dv_planes_ok(planar)) { cvs_set_error("DV subsample: need three 720x480 / 180x480 planes"); return -1; }
    hipStream_t s = cvs_pick_stream(stream);
    for (int p = 0; p < 3; p++) CVS_HIP(hipMemsetAsync(planar->data[p], 0, (size_t)planar->stride[p] * DV_H, s));
    box2i w;
    box2i_set(&w, max(0, frame->current_window.min.x), max(DV_OFF_Y, frame->current_window.min.y),
              min(DV_W - 1, frame->current_window.max.x), min(DV_H + DV_OFF_Y - 1, frame->current_window.max.y));
    if (box2i_is_empty(&w)) return 0;
    if (!cvs_box_contains(&frame->full_window, &w)) { cvs_set_error("DV subsample: current window outside the buffer"); return -1; }
    cvk_dv_taps tri;
    const half *lut = cvs_lut_device(CVS_LUT_LINEAR_TO_REC709);
    if (dv_taps(1.0f / (float)DV_SUB, &tri) != 0 || !lut) return -1;
    cvk_dv_planes pl = dv_view(planar);
    CVS_KERNEL(CVK(cvk_dv_subsample)(&pl, cvs_view(frame->data, &frame->full_window), cvs_rect(&w), &tri, lut, encode_input_in_place, s));
    return 0;
}

/* ---- reference-named entry points on host memory ---- */

typedef struct { coded_image dev; void *block; } dev_planes;

static int planes_to_device(dev_planes *d, const coded_image *host, bool upload, hipStream_t s) {
    memset(d, 0, sizeof *d);
    size_t off[3], total = 0;
    for (int p = 0; p < 3; p++) { off[p] = total; total += ((size_t)host->stride[p] * DV_H + 255) & ~(size_t)255; }
    d->block = cvs_pool_malloc(total, s);
    if (!d->block) return -1;
    for (int p = 0; p < 3; p++) {
        d->dev.data[p] = (char *)d->block + off[p];
        d->dev.stride[p] = host->stride[p];
        d->dev.line_count[p] = DV_H;
        if (upload && cvs_memcpy_h2d(d->dev.data[p], host->data[p], (size_t)host->stride[p] * DV_H, s) != 0) return -1;
    }
    return 0;
}

CVS_EXPORT void video_reconstruct_dv(rgba_frame_f16 *frame, coded_image *planar) {
    if (cvs_enter() != 0 || !dv_planes_ok(planar)) { box2i_set_empty(&frame->current_window); return; }
    hipStream_t s = cvs_pick_stream(NULL);
    dev_planes dp;
    const size_t fbytes = cvs_box_pixels(&frame->full_window) * sizeof(rgba_f16);
    rgba_frame_f16 dframe = *frame;
    dframe.data = cvs_pool_malloc(fbytes ? fbytes : 1, s);
    int rc = dframe.data ? planes_to_device(&dp, planar, true, s) : -1;
    if (rc == 0) rc = cvs_reconstruct_dv_dev(&dframe, &dp.dev, s);
    /* pixels outside the current window are undefined: the whole buffer comes back in one copy */
    if (rc == 0 && !box2i_is_empty(&dframe.current_window)) rc = cvs_memcpy_d2h(frame->data, dframe.data, fbytes, s);
    frame->current_window = dframe.current_window;
    if (rc != 0) box2i_set_empty(&frame->current_window);
    cvs_pool_free(dframe.data, s);
    if (dframe.data) cvs_pool_free(dp.block, s);
}

CVS_EXPORT coded_image *video_subsample_dv(rgba_frame_f16 *frame) {
    const int strides[3] = { DV_W, DV_W / DV_SUB, DV_W / DV_SUB }, lines[3] = { DV_H, DV_H, DV_H };
    if (cvs_enter() != 0) return NULL;
    coded_image *out = coded_image_alloc(strides, lines, 3);
    if (!out) return NULL;
    hipStream_t s = cvs_pick_stream(NULL);
    dev_planes dp;
    memset(&dp, 0, sizeof dp);
    const size_t fbytes = cvs_box_pixels(&frame->full_window) * sizeof(rgba_f16);
    rgba_frame_f16 dframe = *frame;
    dframe.data = cvs_pool_malloc(fbytes ? fbytes : 1, s);
    const bool have_pixels = !box2i_is_empty(&frame->current_window) && cvs_box_contains(&frame->full_window, &frame->current_window);
    int rc = dframe.data ? planes_to_device(&dp, out, false, s) : -1;
    if (rc == 0 && have_pixels) rc = cvs_memcpy_h2d(dframe.data, frame->data, fbytes, s);
    if (rc == 0 && !have_pixels) box2i_set_empty(&dframe.current_window);
    if (rc == 0) rc = cvs_subsample_dv_dev(&dp.dev, &dframe, 1, s);
    for (int p = 0; rc == 0 && p < 3; p++) rc = cvs_memcpy_d2h(out->data[p], dp.dev.data[p], (size_t)strides[p] * DV_H, s);
    /* the reference leaves the rows it read transfer-encoded in the caller's frame (video_subsample.c:144) */
    if (rc == 0 && have_pixels) rc = cvs_memcpy_d2h(frame->data, dframe.data, fbytes, s);
    cvs_pool_free(dframe.data, s);
    cvs_pool_free(dp.block, s);
    if (rc != 0) { coded_image_release(out); return NULL; }
    return out;
}
