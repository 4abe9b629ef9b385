/*
 * frames.c -- frame-level entry points: pull dispatch, copy, crossfade, alpha-over, gain/offset,
 * solid fill.  Window arithmetic stays on the host exactly where the reference does it once per
 * call; pixels go to kernels/frame_ops.hip and kernels/mix_ops.hip.
 *
 * Replaces, function for function:
 *   src/cprocess/main.c:33-76,105-144         video_get_frame_f16 / video_get_frame_f32
 *   src/cprocess/video_mix.c:27-44            video_copy_frame_f16
 *   src/cprocess/video_mix.c:73-105           video_copy_frame_alpha_f32
 *   src/cprocess/video_mix.c:107-235,46-71    video_mix_cross_f32 (+ _pull)
 *   src/cprocess/video_mix.c:237-370          video_mix_over_f32
 *   src/cprocess/video_filter.c:27-39 + gl.c:584   gain/offset (GLSL only in the reference)
 *   src/process/SolidColorVideoSource.c:52-101     solid fill loops
 * Every entry exists twice: `cvs_*_dev` on frames already in HBM, and the reference-named one on
 * host frames, which stages the buffers through HBM around the same `_dev` call.
 */
#include "internal.h"

/* ---------------------------------------------------------------- device-frame operations */

CVS_EXPORT int cvs_copy_frame_f16_dev(rgba_frame_f16 *out, const rgba_frame_f16 *in, cvs_stream_t s) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return -1; }
    CVS_REQUIRE_INSIDE(in, out, "cvs_copy_frame_f16_dev");
    box2i inner;
    box2i_intersect(&inner, &out->full_window, &in->current_window);
    out->current_window = inner;
    if (box2i_is_empty(&inner)) return 0;
    CVS_KERNEL(cvk_copy_f16(cvs_view(out->data, &out->full_window), cvs_view(in->data, &in->full_window), cvs_rect(&inner), cvs_pick_stream(s)));
    return 0;
}

/* src/process/Pulldown23RemovalFilter.c:51-71: which source frames make output frame `frame_index` of a 2:3 cadence
 * with phase `offset` (0..4).  Returns 0 and *first for a whole frame, 1 and *first (odd rows) + *second (even rows)
 * for a frame woven from two fields.  The arithmetic is the reference's, in int, shifts on negative values included. */
CVS_EXPORT int cvs_pulldown23_frames(int offset, int frame_index, int *first, int *second) {
    const int frame_offset = offset == 4 ? ((frame_index + 3) & 3) : ((frame_index + offset) & 3);
    const int base = ((frame_index + offset) >> 2) * 5 - offset;
    *second = base + 3;
    switch (frame_offset) {
    case 0: *first = base; return 0;
    case 1: *first = base + 1; return 0;
    case 3: *first = base + 4; return 0;
    default: *first = base + 2; return 1;
    }
}

/* :88-104: `other` was pulled into a buffer allocated for frame->current_window; its even rows replace the frame's */
CVS_EXPORT int cvs_weave_fields_f16_dev(rgba_frame_f16 *frame, const rgba_frame_f16 *other, cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    if (box2i_is_empty(&frame->current_window)) return 0;
    if (!cvs_box_contains(&frame->full_window, &frame->current_window)) { cvs_set_error("cvs_weave_fields_f16_dev: current window outside the buffer"); return -1; }
    if (memcmp(&other->full_window, &frame->current_window, sizeof(box2i)) != 0) {
        cvs_set_error("cvs_weave_fields_f16_dev: the second field's buffer must be allocated for the frame's current window");
        return -1;
    }
    box2i ocur;
    box2i_intersect(&ocur, &other->current_window, &other->full_window);
    CVS_KERNEL(cvk_weave_f16(cvs_view(frame->data, &frame->full_window), cvs_rect(&frame->current_window), other->data, cvs_rect(&ocur), cvs_pick_stream(s)));
    return 0;
}

CVS_EXPORT int cvs_copy_frame_alpha_f32_dev(rgba_frame_f32 *out, const rgba_frame_f32 *in, float alpha, cvs_stream_t s) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return -1; }
    CVS_REQUIRE_INSIDE(in, out, "cvs_copy_frame_alpha_f32_dev");
    alpha = clampf(alpha, 0.0f, 1.0f);
    if (out->data == in->data && alpha == 1.0f) return 0;             /* video_mix.c:77-78 (same frame, nothing to do) */
    if (alpha == 0.0f) { box2i_set_empty(&out->current_window); return 0; }
    box2i inner;
    box2i_intersect(&inner, &out->full_window, &in->current_window);
    out->current_window = inner;
    if (box2i_is_empty(&inner)) return 0;
    CVS_KERNEL(cvk_copy_alpha_f32(cvs_view(out->data, &out->full_window), cvs_view(in->data, &in->full_window), cvs_rect(&inner), alpha, cvs_pick_stream(s)));
    return 0;
}

CVS_EXPORT int cvs_frame_f16_to_f32_dev(rgba_frame_f32 *out, const rgba_frame_f16 *in, cvs_stream_t s) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return -1; }
    CVS_REQUIRE_INSIDE(in, out, "cvs_frame_f16_to_f32_dev");
    /* main.c:115-139: the temp frame shares the target's full window, rows of current_window are widened */
    out->current_window = in->current_window;
    if (box2i_is_empty(&in->current_window)) return 0;
    if (!cvs_box_contains(&out->full_window, &in->current_window)) { cvs_set_error("f16->f32: source window outside target buffer"); box2i_set_empty(&out->current_window); return -1; }
    CVS_KERNEL(cvk_widen(cvs_view(out->data, &out->full_window), cvs_view(in->data, &in->full_window), cvs_rect(&in->current_window), cvs_pick_stream(s)));
    return 0;
}

CVS_EXPORT int cvs_frame_f32_to_f16_dev(rgba_frame_f16 *out, const rgba_frame_f32 *in, cvs_stream_t s) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return -1; }
    CVS_REQUIRE_INSIDE(in, out, "cvs_frame_f32_to_f16_dev");
    out->current_window = in->current_window;                          /* main.c:43-71 */
    if (box2i_is_empty(&in->current_window)) return 0;
    if (!cvs_box_contains(&out->full_window, &in->current_window)) { cvs_set_error("f32->f16: source window outside target buffer"); box2i_set_empty(&out->current_window); return -1; }
    CVS_KERNEL(cvk_narrow(cvs_view(out->data, &out->full_window), cvs_view(in->data, &in->full_window), cvs_rect(&in->current_window), cvs_pick_stream(s)));
    return 0;
}

/* The decisions video_mix.c makes once per call (:121-139, :251-267), packaged for the kernel. */
static void plan_mix(cvk_mix_params *mp, void *out_data, const box2i *out_full,
                     void *p_data, const box2i *p_full, const box2i *pw,
                     void *q_data, const box2i *q_full, const box2i *qw,
                     float wp, float wq, int mode, int p_in_place, box2i *outer_out) {
    box2i outer, inner;
    box2i_union(&outer, pw, qw);
    box2i_intersect(&outer, &outer, out_full);
    box2i_intersect(&inner, pw, qw);
    box2i_intersect(&inner, &inner, out_full);
    mp->gap_x = inner.min.x > inner.max.x;
    mp->gap_y = inner.min.y > inner.max.y;
    box2i_normalize(&inner);

    mp->top_is_p = pw->min.y < qw->min.y;
    mp->bottom_is_p = pw->max.y > qw->max.y;
    mp->left_is_p = pw->min.x < qw->min.y;     /* the reference compares against min.Y here (video_mix.c:137,265); kept */
    mp->right_is_p = pw->max.x > qw->max.x;

    mp->out = cvs_view(out_data, out_full);
    mp->p = cvs_view(p_data, p_full);
    mp->q = cvs_view(q_data, q_full);
    mp->outer = cvs_rect(&outer);
    mp->inner = cvs_rect(&inner);
    mp->pw = cvs_rect(pw);
    mp->qw = cvs_rect(qw);
    mp->wp = wp; mp->wq = wq;
    mp->mode = mode;
    mp->p_in_place = p_in_place;
    *outer_out = outer;
}

CVS_EXPORT int cvs_mix_cross_f32_dev(rgba_frame_f32 *out, const rgba_frame_f32 *a, const rgba_frame_f32 *b, float mix_b, cvs_stream_t s) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return -1; }
    CVS_REQUIRE_INSIDE(a, out, "cvs_mix_cross_f32_dev");
    CVS_REQUIRE_INSIDE(b, out, "cvs_mix_cross_f32_dev");
    mix_b = clampf(mix_b, 0.0f, 1.0f);
    const float mix_a = 1.0f - mix_b;
    if (box2i_is_empty(&a->current_window)) return cvs_copy_frame_alpha_f32_dev(out, b, mix_b, s);
    if (box2i_is_empty(&b->current_window)) return cvs_copy_frame_alpha_f32_dev(out, a, mix_a, s);
    cvk_mix_params mp;
    box2i outer;
    plan_mix(&mp, out->data, &out->full_window, a->data, &a->full_window, &a->current_window,
             b->data, &b->full_window, &b->current_window, mix_a, mix_b, CVK_MIX_CROSS, 0, &outer);
    CVS_KERNEL(CVK(cvk_mix)(&mp, cvs_pick_stream(s)));
    out->current_window = outer;
    return 0;
}

CVS_EXPORT int cvs_mix_over_f32_dev(rgba_frame_f32 *out, const rgba_frame_f32 *b, float mix_b, cvs_stream_t s) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return -1; }
    CVS_REQUIRE_INSIDE(b, out, "cvs_mix_over_f32_dev");
    CVS_REQUIRE_INSIDE(out, out, "cvs_mix_over_f32_dev");
    mix_b = clampf(mix_b, 0.0f, 1.0f);
    if (box2i_is_empty(&out->current_window)) return cvs_copy_frame_alpha_f32_dev(out, b, mix_b, s);
    if (box2i_is_empty(&b->current_window) || mix_b == 0.0f) return 0;
    cvk_mix_params mp;
    box2i outer;
    plan_mix(&mp, out->data, &out->full_window, out->data, &out->full_window, &out->current_window,
             b->data, &b->full_window, &b->current_window, 1.0f, mix_b, CVK_MIX_OVER, 1, &outer);
    CVS_KERNEL(CVK(cvk_mix)(&mp, cvs_pick_stream(s)));
    out->current_window = outer;
    return 0;
}

CVS_EXPORT int cvs_gain_offset_f16_dev(rgba_frame_f16 *out, const rgba_frame_f16 *in, float gain, float offset, cvs_stream_t s) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return -1; }
    CVS_REQUIRE_INSIDE(in, out, "cvs_gain_offset_f16_dev");
    box2i win;
    box2i_intersect(&win, &out->full_window, &in->current_window);    /* gl.c:584: one-input filters cover out.full ∩ in.current */
    out->current_window = win;
    if (box2i_is_empty(&win)) return 0;
    CVS_KERNEL(CVK(cvk_gain_offset_f16)(cvs_view(out->data, &out->full_window), cvs_view(in->data, &in->full_window), cvs_rect(&win), gain, offset, cvs_pick_stream(s)));
    return 0;
}

CVS_EXPORT int cvs_fill_solid_f16_dev(rgba_frame_f16 *frame, const box2i *window, const rgba_f32 *color, cvs_stream_t s) {
    if (cvs_enter() != 0) { box2i_set_empty(&frame->current_window); return -1; }
    box2i_intersect(&frame->current_window, window, &frame->full_window);
    if (box2i_is_empty(&frame->current_window)) return 0;
    /* the colour is truncated to half inside the kernel (SolidColorVideoSource.c:68-69: once per frame, same rounding) */
    const float c[4] = { color->r, color->g, color->b, color->a };
    CVS_KERNEL(cvk_fill_f16(cvs_view(frame->data, &frame->full_window), cvs_rect(&frame->current_window), c, cvs_pick_stream(s)));
    return 0;
}

CVS_EXPORT int cvs_fill_solid_f32_dev(rgba_frame_f32 *frame, const box2i *window, const rgba_f32 *color, cvs_stream_t s) {
    if (cvs_enter() != 0) { box2i_set_empty(&frame->current_window); return -1; }
    box2i_intersect(&frame->current_window, window, &frame->full_window);
    if (box2i_is_empty(&frame->current_window)) return 0;
    CVS_KERNEL(cvk_fill_f32(cvs_view(frame->data, &frame->full_window), cvs_rect(&frame->current_window), (const float *)color, cvs_pick_stream(s)));
    return 0;
}

/* ---------------------------------------------------------------- host-frame wrappers */

#define F16_BYTES(f) (cvs_box_pixels(&(f)->full_window) * sizeof(rgba_f16))
#define F32_BYTES(f) (cvs_box_pixels(&(f)->full_window) * sizeof(rgba_f32))

CVS_EXPORT void video_copy_frame_f16(rgba_frame_f16 *out, rgba_frame_f16 *in) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return; }
    hipStream_t s = cvs_pick_stream(NULL);
    cvs_staged d_in = { 0 }, d_out = { 0 };
    rgba_frame_f16 fi = *in, fo = *out;
    int rc = cvs_stage_in(&d_in, in->data, F16_BYTES(in), !box2i_is_empty(&in->current_window), s);
    if (rc == 0) rc = cvs_stage_in(&d_out, out->data, F16_BYTES(out), 1, s);
    fi.data = d_in.dev; fo.data = d_out.dev;
    if (rc == 0) rc = cvs_copy_frame_f16_dev(&fo, &fi, s);
    if (rc == 0) rc = cvs_stage_out(&d_out, out->data, s);
    out->current_window = fo.current_window;
    if (rc != 0) box2i_set_empty(&out->current_window);
    cvs_stage_free(&d_in); cvs_stage_free(&d_out);
}

CVS_EXPORT void video_copy_frame_alpha_f32(rgba_frame_f32 *out, rgba_frame_f32 *in, float alpha) {
    if (out == in && clampf(alpha, 0.0f, 1.0f) == 1.0f) return;
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return; }
    hipStream_t s = cvs_pick_stream(NULL);
    cvs_staged d_in = { 0 }, d_out = { 0 };
    rgba_frame_f32 fi = *in, fo = *out;
    int rc = cvs_stage_in(&d_out, out->data, F32_BYTES(out), 1, s);
    if (rc == 0 && in != out) rc = cvs_stage_in(&d_in, in->data, F32_BYTES(in), !box2i_is_empty(&in->current_window), s);
    fo.data = d_out.dev;
    fi.data = in == out ? d_out.dev : d_in.dev;
    if (rc == 0) rc = cvs_copy_frame_alpha_f32_dev(&fo, &fi, alpha, s);
    if (rc == 0) rc = cvs_stage_out(&d_out, out->data, s);
    out->current_window = fo.current_window;
    if (rc != 0) box2i_set_empty(&out->current_window);
    cvs_stage_free(&d_in); cvs_stage_free(&d_out);
}

/* framework.h:236 (declared there, defined nowhere in the reference): the in-place case of the copy above */
CVS_EXPORT void video_attenuate_f32(rgba_frame_f32 *frame, float alpha) { video_copy_frame_alpha_f32(frame, frame, alpha); }

CVS_EXPORT void video_mix_cross_f32(rgba_frame_f32 *out, rgba_frame_f32 *a, rgba_frame_f32 *b, float mix_b) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return; }
    hipStream_t s = cvs_pick_stream(NULL);
    cvs_staged d_out = { 0 }, d_a = { 0 }, d_b = { 0 };
    rgba_frame_f32 fo = *out, fa = *a, fb = *b;
    int rc = cvs_stage_in(&d_out, out->data, F32_BYTES(out), 1, s);
    if (rc == 0 && a != out) rc = cvs_stage_in(&d_a, a->data, F32_BYTES(a), !box2i_is_empty(&a->current_window), s);
    if (rc == 0 && b != out && b != a) rc = cvs_stage_in(&d_b, b->data, F32_BYTES(b), !box2i_is_empty(&b->current_window), s);
    fo.data = d_out.dev;
    fa.data = a == out ? d_out.dev : d_a.dev;
    fb.data = b == out ? d_out.dev : (b == a ? fa.data : d_b.dev);
    if (rc == 0) rc = cvs_mix_cross_f32_dev(&fo, &fa, &fb, mix_b, s);
    if (rc == 0) rc = cvs_stage_out(&d_out, out->data, s);
    out->current_window = fo.current_window;
    if (rc != 0) box2i_set_empty(&out->current_window);
    cvs_stage_free(&d_out); cvs_stage_free(&d_a); cvs_stage_free(&d_b);
}

CVS_EXPORT void video_mix_over_f32(rgba_frame_f32 *out, rgba_frame_f32 *b, float mix_b) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return; }
    hipStream_t s = cvs_pick_stream(NULL);
    cvs_staged d_out = { 0 }, d_b = { 0 };
    rgba_frame_f32 fo = *out, fb = *b;
    int rc = cvs_stage_in(&d_out, out->data, F32_BYTES(out), 1, s);
    if (rc == 0) rc = cvs_stage_in(&d_b, b->data, F32_BYTES(b), !box2i_is_empty(&b->current_window), s);
    fo.data = d_out.dev; fb.data = d_b.dev;
    if (rc == 0) rc = cvs_mix_over_f32_dev(&fo, &fb, mix_b, s);
    if (rc == 0) rc = cvs_stage_out(&d_out, out->data, s);
    out->current_window = fo.current_window;
    if (rc != 0) box2i_set_empty(&out->current_window);
    cvs_stage_free(&d_out); cvs_stage_free(&d_b);
}

CVS_EXPORT void video_filter_gain_offset_f16(rgba_frame_f16 *out, rgba_frame_f16 *in, float gain, float offset) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return; }
    hipStream_t s = cvs_pick_stream(NULL);
    cvs_staged d_in = { 0 }, d_out = { 0 };
    rgba_frame_f16 fi = *in, fo = *out;
    int rc = cvs_stage_in(&d_out, out->data, F16_BYTES(out), 1, s);
    if (rc == 0 && in != out) rc = cvs_stage_in(&d_in, in->data, F16_BYTES(in), !box2i_is_empty(&in->current_window), s);
    fo.data = d_out.dev;
    fi.data = in == out ? d_out.dev : d_in.dev;
    if (rc == 0) rc = cvs_gain_offset_f16_dev(&fo, &fi, gain, offset, s);
    if (rc == 0) rc = cvs_stage_out(&d_out, out->data, s);
    out->current_window = fo.current_window;
    if (rc != 0) box2i_set_empty(&out->current_window);
    cvs_stage_free(&d_in); cvs_stage_free(&d_out);
}

CVS_EXPORT void video_fill_solid_f16(rgba_frame_f16 *frame, const box2i *window, const rgba_f32 *color) {
    if (cvs_enter() != 0) { box2i_set_empty(&frame->current_window); return; }
    hipStream_t s = cvs_pick_stream(NULL);
    cvs_staged d = { 0 };
    rgba_frame_f16 f = *frame;
    int rc = cvs_stage_in(&d, frame->data, F16_BYTES(frame), 1, s);
    f.data = d.dev;
    if (rc == 0) rc = cvs_fill_solid_f16_dev(&f, window, color, s);
    if (rc == 0) rc = cvs_stage_out(&d, frame->data, s);
    frame->current_window = f.current_window;
    if (rc != 0) box2i_set_empty(&frame->current_window);
    cvs_stage_free(&d);
}

CVS_EXPORT void video_fill_solid_f32(rgba_frame_f32 *frame, const box2i *window, const rgba_f32 *color) {
    if (cvs_enter() != 0) { box2i_set_empty(&frame->current_window); return; }
    hipStream_t s = cvs_pick_stream(NULL);
    cvs_staged d = { 0 };
    rgba_frame_f32 f = *frame;
    int rc = cvs_stage_in(&d, frame->data, F32_BYTES(frame), 1, s);
    f.data = d.dev;
    if (rc == 0) rc = cvs_fill_solid_f32_dev(&f, window, color, s);
    if (rc == 0) rc = cvs_stage_out(&d, frame->data, s);
    frame->current_window = f.current_window;
    if (rc != 0) box2i_set_empty(&frame->current_window);
    cvs_stage_free(&d);
}

/* ---------------------------------------------------------------- pull dispatch (main.c) */

CVS_EXPORT void video_get_frame_f16(video_source *source, int frame_index, rgba_frame_f16 *frame) {
    if (!source || !source->funcs) { box2i_set_empty(&frame->current_window); return; }
    if (source->funcs->get_frame) { source->funcs->get_frame(source->obj, frame_index, frame); return; }
    if ((source->funcs->flags & VIDEO_SOURCE_FLAG_DEVICE) && source->funcs->get_frame_dev) {
        /* a source with a device slot and no f16 host slot: render in HBM as f16 (the node narrows there, the same
         * truncation main.c:43-71 would apply to its f32 output) and bring back 8 B per pixel once */
        if (cvs_enter() != 0) { box2i_set_empty(&frame->current_window); return; }
        size_t bytes = F16_BYTES(frame);
        rgba_frame_dev d = { cvs_pool_malloc(bytes, NULL), CVS_FORMAT_F16, frame->full_window, frame->full_window, NULL };
        if (!d.data) { box2i_set_empty(&frame->current_window); return; }
        source->funcs->get_frame_dev(source->obj, frame_index, &d);
        if (!box2i_is_empty(&d.current_window) && cvs_memcpy_d2h(frame->data, d.data, bytes, NULL) != 0) box2i_set_empty(&d.current_window);
        frame->current_window = d.current_window;
        cvs_pool_free(d.data, NULL);
        return;
    }
    if (source->funcs->get_frame_32) {
        /* main.c:43-71: pull f32 into a temp covering the same full window, narrow current_window */
        rgba_frame_f32 tmp;
        size_t n = cvs_box_pixels(&frame->full_window);
        tmp.data = malloc(sizeof(rgba_f32) * (n ? n : 1));
        if (!tmp.data) { box2i_set_empty(&frame->current_window); return; }
        tmp.full_window = frame->full_window;
        tmp.current_window = frame->full_window;
        source->funcs->get_frame_32(source->obj, frame_index, &tmp);
        if (!box2i_is_empty(&tmp.current_window)) {
            if (cvs_enter() == 0) {
                hipStream_t s = cvs_pick_stream(NULL);
                cvs_staged d32 = { 0 }, d16 = { 0 };
                rgba_frame_f32 f32 = tmp;
                rgba_frame_f16 f16 = *frame;
                int rc = cvs_stage_in(&d32, tmp.data, n * sizeof(rgba_f32), 1, s);
                if (rc == 0) rc = cvs_stage_in(&d16, frame->data, n * sizeof(rgba_f16), 1, s);
                f32.data = d32.dev; f16.data = d16.dev;
                if (rc == 0) rc = cvs_frame_f32_to_f16_dev(&f16, &f32, s);
                if (rc == 0) rc = cvs_stage_out(&d16, frame->data, s);
                if (rc != 0) box2i_set_empty(&tmp.current_window);
                cvs_stage_free(&d32); cvs_stage_free(&d16);
            } else {
                box2i_set_empty(&tmp.current_window);
            }
        }
        frame->current_window = tmp.current_window;
        free(tmp.data);
        return;
    }
    box2i_set_empty(&frame->current_window);                            /* the GL branch (main.c:73-75) is gone */
}

CVS_EXPORT void video_get_frame_f32(video_source *source, int frame_index, rgba_frame_f32 *frame) {
    if (!source || !source->funcs) { box2i_set_empty(&frame->current_window); return; }
    if (source->funcs->get_frame_32) { source->funcs->get_frame_32(source->obj, frame_index, frame); return; }
    if ((source->funcs->flags & VIDEO_SOURCE_FLAG_DEVICE) && source->funcs->get_frame_dev) {
        if (cvs_enter() != 0) { box2i_set_empty(&frame->current_window); return; }
        size_t bytes = F32_BYTES(frame);
        rgba_frame_dev d = { cvs_pool_malloc(bytes, NULL), CVS_FORMAT_F32, frame->full_window, frame->full_window, NULL };
        if (!d.data) { box2i_set_empty(&frame->current_window); return; }
        source->funcs->get_frame_dev(source->obj, frame_index, &d);
        if (!box2i_is_empty(&d.current_window) && cvs_memcpy_d2h(frame->data, d.data, bytes, NULL) != 0) box2i_set_empty(&d.current_window);
        frame->current_window = d.current_window;
        cvs_pool_free(d.data, NULL);
        return;
    }
    if (source->funcs->get_frame) {
        rgba_frame_f16 tmp;                                             /* main.c:115-139 */
        size_t n = cvs_box_pixels(&frame->full_window);
        tmp.data = malloc(sizeof(rgba_f16) * (n ? n : 1));
        if (!tmp.data) { box2i_set_empty(&frame->current_window); return; }
        tmp.full_window = frame->full_window;
        tmp.current_window = frame->full_window;
        source->funcs->get_frame(source->obj, frame_index, &tmp);
        if (!box2i_is_empty(&tmp.current_window)) {
            if (cvs_enter() == 0) {
                hipStream_t s = cvs_pick_stream(NULL);
                cvs_staged d32 = { 0 }, d16 = { 0 };
                rgba_frame_f32 f32 = *frame;
                rgba_frame_f16 f16 = tmp;
                int rc = cvs_stage_in(&d16, tmp.data, n * sizeof(rgba_f16), 1, s);
                if (rc == 0) rc = cvs_stage_in(&d32, frame->data, n * sizeof(rgba_f32), 1, s);
                f32.data = d32.dev; f16.data = d16.dev;
                if (rc == 0) rc = cvs_frame_f16_to_f32_dev(&f32, &f16, s);
                if (rc == 0) rc = cvs_stage_out(&d32, frame->data, s);
                if (rc != 0) box2i_set_empty(&tmp.current_window);
                cvs_stage_free(&d32); cvs_stage_free(&d16);
            } else {
                box2i_set_empty(&tmp.current_window);
            }
        }
        frame->current_window = tmp.current_window;
        free(tmp.data);
        return;
    }
    box2i_set_empty(&frame->current_window);
}

/* src/cprocess/main.c:78-103,146-172: "get a frame, but forcibly pull it from the GL pipeline" -- what
 * get_frame_f16/f32(..., force_gl=True) calls (src/process/RgbaFrameF16.c:247-249).  Slot 3 is the device slot
 * here, so the forced pull goes through the source's device entry even when it also fills a host slot; a source
 * WITHOUT that slot yields an empty window, exactly as the reference's does without a get_frame_gl (main.c:99-102). */
static int cvs_device_only(video_source *source, video_frame_source_funcs *f, video_source *forced) {
    if (!source || !source->funcs || !(source->funcs->flags & VIDEO_SOURCE_FLAG_DEVICE) || !source->funcs->get_frame_dev) return 0;
    *f = *source->funcs;
    f->get_frame = NULL;
    f->get_frame_32 = NULL;
    forced->obj = source->obj;
    forced->funcs = f;
    return 1;
}
CVS_EXPORT void video_get_frame_f16_gl(video_source *source, int frame_index, rgba_frame_f16 *frame) {
    video_frame_source_funcs f; video_source forced;
    if (!cvs_device_only(source, &f, &forced)) { box2i_set_empty(&frame->current_window); return; }
    video_get_frame_f16(&forced, frame_index, frame);
}
CVS_EXPORT void video_get_frame_f32_gl(video_source *source, int frame_index, rgba_frame_f32 *frame) {
    video_frame_source_funcs f; video_source forced;
    if (!cvs_device_only(source, &f, &forced)) { box2i_set_empty(&frame->current_window); return; }
    video_get_frame_f32(&forced, frame_index, frame);
}

/* Fill a device frame from any source: slot 3 when the source has one, else a host pull + upload. */
CVS_EXPORT void video_get_frame_dev(video_source *source, int frame_index, rgba_frame_dev *frame) {
    if (!source || !source->funcs || cvs_enter() != 0) { box2i_set_empty(&frame->current_window); return; }
    if ((source->funcs->flags & VIDEO_SOURCE_FLAG_DEVICE) && source->funcs->get_frame_dev) {
        source->funcs->get_frame_dev(source->obj, frame_index, frame);
        return;
    }
    size_t n = cvs_box_pixels(&frame->full_window);
    size_t bytes = n * (frame->format == CVS_FORMAT_F32 ? sizeof(rgba_f32) : sizeof(rgba_f16));
    void *host = malloc(bytes ? bytes : 1);
    if (!host) { box2i_set_empty(&frame->current_window); return; }
    box2i cur;
    if (frame->format == CVS_FORMAT_F32) {
        rgba_frame_f32 f = { host, frame->full_window, frame->full_window };
        video_get_frame_f32(source, frame_index, &f);
        cur = f.current_window;
    } else {
        rgba_frame_f16 f = { host, frame->full_window, frame->full_window };
        video_get_frame_f16(source, frame_index, &f);
        cur = f.current_window;
    }
    if (!box2i_is_empty(&cur)) {
        hipStream_t s = cvs_pick_stream(frame->stream);
        if (hipMemcpyAsync(frame->data, host, bytes, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            cvs_set_error("upload of pulled frame failed");
            box2i_set_empty(&cur);
        }
    }
    frame->current_window = cur;
    free(host);
}

CVS_EXPORT void video_mix_cross_f32_pull(rgba_frame_f32 *out, video_source *a, int frame_a, video_source *b, int frame_b, float mix_b) {
    mix_b = clampf(mix_b, 0.0f, 1.0f);                                  /* video_mix.c:46-71 */
    if (mix_b == 0.0f) { video_get_frame_f32(a, frame_a, out); return; }
    if (mix_b == 1.0f) { video_get_frame_f32(b, frame_b, out); return; }
    rgba_frame_f32 tmp;
    size_t n = cvs_box_pixels(&out->full_window);
    tmp.data = malloc(sizeof(rgba_f32) * (n ? n : 1));
    if (!tmp.data) { box2i_set_empty(&out->current_window); return; }
    tmp.full_window = out->full_window;
    box2i_set_empty(&tmp.current_window);
    video_get_frame_f32(a, frame_a, out);
    video_get_frame_f32(b, frame_b, &tmp);
    video_mix_cross_f32(out, out, &tmp, mix_b);
    free(tmp.data);
}
