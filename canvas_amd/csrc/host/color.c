/*
 * color.c -- colour matrix entry points and the fused colour + alpha-over chain.
 *
 * Replaces src/cprocess/color.c:104-137 (video_color_rgb_to_xyz_sdtv) and :140-165
 * (video_color_xyz_to_srgb); the general form behind both is cvs_color_matrix_f16_dev.
 * The chain entry composes color.c + workspace.c:530-544 + main.c:43-71 into one kernel when every
 * window involved covers the output's full window, and otherwise runs the same nodes one by one
 * on the device (same arithmetic, f32 intermediates in HBM) so that ragged windows get the
 * reference's window behaviour.
 */
#include "internal.h"

CVS_EXPORT int cvs_color_matrix_f16_dev(rgba_frame_f16 *frame, const float m[9], int pre_lut, int post_lut, cvs_stream_t s) {
    if (cvs_enter() != 0) return -1;
    CVS_REQUIRE_INSIDE(frame, frame, "cvs_color_matrix_f16_dev");
    if (box2i_is_empty(&frame->current_window)) return 0;
    if (!cvs_box_contains(&frame->full_window, &frame->current_window)) { cvs_set_error("colour matrix: current_window outside the buffer"); return -1; }
    const half *pre = cvs_lut_dev_or_null(pre_lut), *post = cvs_lut_dev_or_null(post_lut);
    if ((pre_lut != CVS_LUT_NONE && !pre) || (post_lut != CVS_LUT_NONE && !post)) return -1;
    cvk_view v = cvs_view(frame->data, &frame->full_window);
    CVS_KERNEL(CVK(cvk_color_matrix)(v, v, cvs_rect(&frame->current_window), m, pre, post, cvs_cus(), cvs_pick_stream(s)));
    return 0;
}

/* the same filter writing into another frame: out.current = out.full ∩ in.current (the one-input window rule,
 * gl.c:584), one pass instead of video_copy_frame_f16 + the in-place filter */
CVS_EXPORT int cvs_color_matrix_f16_to_dev(rgba_frame_f16 *out, const rgba_frame_f16 *in, const float m[9], int pre_lut, int post_lut, cvs_stream_t s) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return -1; }
    CVS_REQUIRE_INSIDE(in, out, "cvs_color_matrix_f16_to_dev");
    box2i win;
    box2i_intersect(&win, &out->full_window, &in->current_window);
    out->current_window = win;
    if (box2i_is_empty(&win)) return 0;
    if (!cvs_box_contains(&in->full_window, &win)) { cvs_set_error("colour matrix: source window outside its buffer"); box2i_set_empty(&out->current_window); return -1; }
    const half *pre = cvs_lut_dev_or_null(pre_lut), *post = cvs_lut_dev_or_null(post_lut);
    if ((pre_lut != CVS_LUT_NONE && !pre) || (post_lut != CVS_LUT_NONE && !post)) { box2i_set_empty(&out->current_window); return -1; }
    CVS_KERNEL(CVK(cvk_color_matrix)(cvs_view(out->data, &out->full_window), cvs_view(in->data, &in->full_window), cvs_rect(&win), m, pre, post, cvs_cus(), cvs_pick_stream(s)));
    return 0;
}

static void host_color(rgba_frame_f16 *frame, const float m[9], int pre, int post) {
    if (box2i_is_empty(&frame->current_window) || cvs_enter() != 0) return;
    hipStream_t s = cvs_pick_stream(NULL);
    cvs_staged d = { 0 };
    rgba_frame_f16 f = *frame;
    int rc = cvs_stage_in(&d, frame->data, cvs_box_pixels(&frame->full_window) * sizeof(rgba_f16), 1, s);
    f.data = d.dev;
    if (rc == 0) rc = cvs_color_matrix_f16_dev(&f, m, pre, post, s);
    if (rc == 0) cvs_stage_out(&d, frame->data, s);
    cvs_stage_free(&d);
}

/* SMPTE-C RGB (Rec.709 transfer) -> linear XYZ; coefficients as color.c:115-118, column-major */
CVS_EXPORT void video_color_rgb_to_xyz_sdtv(rgba_frame_f16 *frame) {
    static const float m[9] = { 0.3936f, 0.2124f, 0.0187f, 0.3652f, 0.7010f, 0.1119f, 0.1916f, 0.0865f, 0.9582f };
    host_color(frame, m, CVS_LUT_REC709_TO_LINEAR_SCENE, CVS_LUT_NONE);
}

/* linear XYZ -> sRGB; coefficients as color.c:143-146 */
CVS_EXPORT void video_color_xyz_to_srgb(rgba_frame_f16 *frame) {
    static const float m[9] = { 3.2410f, -0.9692f, 0.0556f, -1.5374f, 1.8760f, -0.2040f, -0.4986f, 0.0416f, 1.0570f };
    host_color(frame, m, CVS_LUT_NONE, CVS_LUT_LINEAR_TO_SRGB);
}

/* ---------------------------------------------------------------- chain */

static __thread int t_last_fused = -1;
CVS_EXPORT int cvs_chain_last_was_fused(void) { return t_last_fused; }
CVS_EXPORT int cvs_chain_last_launch_count(void) { return CVK(cvk_chain_count)(); }

static bool same_box(const box2i *a, const box2i *b) {
    return a->min.x == b->min.x && a->min.y == b->min.y && a->max.x == b->max.x && a->max.y == b->max.y;
}

static bool ranges_overlap(const void *a, const void *b, size_t bytes) {
    const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
    return x < y + bytes && y < x + bytes;
}

static bool job_is_fusable(const cvs_chain_job *j) {
    if (j->nlayers < 1 || j->nlayers > CVS_CHAIN_MAX_LAYERS) return false;
    const box2i *fw = &j->out->full_window;
    if (box2i_is_empty(fw) || cvs_box_pixels(fw) < 2) return false;       /* the kernel works on pixel pairs */
    if (((uintptr_t)j->out->data & 15u) != 0) return false;
    const size_t bytes = cvs_box_pixels(fw) * sizeof(rgba_f16);
    for (int k = 0; k < j->nlayers; k++) {
        const rgba_frame_f16 *l = j->layers[k];
        if (!same_box(&l->full_window, fw) || !same_box(&l->current_window, fw)) return false;
        if (((uintptr_t)l->data & 15u) != 0) return false;
        /* in place (out IS a layer) is fine: a lane reads a pixel pair of every layer before it writes that pair;
         * a shifted overlap is not: another lane may still have to read what this one writes */
        if (l->data != j->out->data && ranges_overlap(l->data, j->out->data, bytes)) return false;
    }
    return true;
}

/* Jobs of one call may feed each other (job 5 stacking job 2's output) or reuse an output buffer.  One launch of the
 * fused kernel gives no order between its jobs -- workgroups walk the whole batch, each at its own pace -- so a job
 * that touches what an earlier job of the same launch writes (or writes what an earlier one reads) starts a new launch:
 * launches on one stream run in order.  Returns true when job `j` conflicts with any of jobs [first, j). */
static bool job_depends_on_earlier(const cvs_chain_job *jobs, int first, int j) {
    const size_t bj = cvs_box_pixels(&jobs[j].out->full_window) * sizeof(rgba_f16);
    for (int i = first; i < j; i++) {
        const size_t bi = cvs_box_pixels(&jobs[i].out->full_window) * sizeof(rgba_f16);
        const size_t span = bi > bj ? bi : bj;                                   /* conservative for frames of different sizes */
        if (ranges_overlap(jobs[j].out->data, jobs[i].out->data, span)) return true;                  /* write after write */
        for (int k = 0; k < jobs[j].nlayers; k++)
            if (ranges_overlap(jobs[j].layers[k]->data, jobs[i].out->data, span)) return true;          /* read after write */
        for (int k = 0; k < jobs[i].nlayers; k++)
            if (ranges_overlap(jobs[j].out->data, jobs[i].layers[k]->data, span)) return true;          /* write after read */
    }
    return false;
}

/* the same graph, one device kernel per reference node; scratch frames are f32 like the reference's temps */
static int chain_unfused(const cvs_chain_job *j, const float m[9], int pre_lut, int post_lut, hipStream_t s) {
    const box2i *fw = &j->out->full_window;
    size_t n = cvs_box_pixels(fw);
    rgba_frame_f16 graded = { NULL, *fw, *fw };
    rgba_frame_f32 acc = { NULL, *fw, *fw }, tmp = { NULL, *fw, *fw };
    int rc = 0;
    if (!n) { box2i_set_empty(&j->out->current_window); return 0; }
    graded.data = cvs_pool_malloc(n * sizeof(rgba_f16), s);
    acc.data = cvs_pool_malloc(n * sizeof(rgba_f32), s);
    tmp.data = cvs_pool_malloc(n * sizeof(rgba_f32), s);
    if (!graded.data || !acc.data || !tmp.data) rc = -1;
    bool have_acc = false;
    for (int k = 0; rc == 0 && k < j->nlayers; k++) {
        /* the layer source: a graded copy of the input, clipped like video_copy_frame_f16 */
        rc = cvs_copy_frame_f16_dev(&graded, j->layers[k], s);
        if (rc == 0 && m) rc = cvs_color_matrix_f16_dev(&graded, m, pre_lut, post_lut, s);
        if (rc != 0) break;
        if (!have_acc) {                       /* workspace.c:530: lowest item straight into the output */
            rc = cvs_frame_f16_to_f32_dev(&acc, &graded, s);
            have_acc = true;
        } else {                               /* workspace.c:538-543 */
            rc = cvs_frame_f16_to_f32_dev(&tmp, &graded, s);
            if (rc == 0) rc = cvs_mix_over_f32_dev(&acc, &tmp, 1.0f, s);
        }
    }
    if (rc == 0) rc = cvs_frame_f32_to_f16_dev(j->out, &acc, s);       /* main.c:43-71 */
    cvs_pool_free(graded.data, s); cvs_pool_free(acc.data, s); cvs_pool_free(tmp.data, s);     /* stream-ordered: no wait */
    if (rc != 0) box2i_set_empty(&j->out->current_window);
    return rc;
}

CVS_EXPORT int cvs_chain_color_over_f16_dev(const cvs_chain_job *jobs, int njobs, const float m[9],
                                            int pre_lut, int post_lut, cvs_stream_t stream) {
    if (cvs_enter() != 0) return -1;
    if (njobs <= 0) return 0;
    hipStream_t s = cvs_pick_stream(stream);
    if (!m && (pre_lut != CVS_LUT_NONE || post_lut != CVS_LUT_NONE)) { cvs_set_error("chain: transfer tables need a colour matrix (m == NULL means no colour stage)"); return -1; }
    const half *pre = cvs_lut_dev_or_null(pre_lut), *post = cvs_lut_dev_or_null(post_lut);
    if ((pre_lut != CVS_LUT_NONE && !pre) || (post_lut != CVS_LUT_NONE && !post)) return -1;

    bool all_fusable = true;
    for (int i = 0; i < njobs; i++) all_fusable = all_fusable && job_is_fusable(&jobs[i]);
    if (!all_fusable) {
        t_last_fused = 0;
        for (int i = 0; i < njobs; i++) {
            int rc = chain_unfused(&jobs[i], m, pre_lut, post_lut, s);
            if (rc != 0) return rc;
        }
        return 0;
    }

    cvk_chain_job *recs = calloc((size_t)njobs, sizeof *recs);
    if (!recs) { cvs_set_error("chain: out of host memory"); return -1; }
    int uniform = jobs[0].nlayers;
    for (int i = 0; i < njobs; i++) {
        cvk_chain_job *kj = &recs[i];
        kj->out = jobs[i].out->data;
        for (int k = 0; k < jobs[i].nlayers; k++) kj->layer[k] = jobs[i].layers[k]->data;
        kj->nlayers = jobs[i].nlayers;
        kj->npixels = cvs_box_pixels(&jobs[i].out->full_window);
        if (jobs[i].nlayers != uniform) uniform = 0;
        jobs[i].out->current_window = jobs[i].out->full_window;
    }
    int rc = 0;
    CVK(cvk_chain_count_reset)();
    for (int first = 0; rc == 0 && first < njobs; ) {       /* runs of mutually independent jobs, one (set of) launch(es) each */
        int end = first + 1;
        while (end < njobs && !job_depends_on_earlier(jobs, first, end)) end++;
        rc = CVK(cvk_chain_color_over)(recs + first, end - first, uniform, m, pre, post, cvs_cus(), s);
        first = end;
    }
    free(recs);
    if (rc != 0) { cvs_set_error("chain kernel launch failed: %s", hipGetErrorString((hipError_t)rc)); return rc; }
    t_last_fused = 1;
    return 0;
}

/* Crossfade between two f16 frames with an f16 result: what an f16 pull of a crossfade node over two half-native
 * sources computes -- widen both (main.c:105-144), video_mix_cross_f32 (video_mix.c:107-235), truncate
 * (main.c:43-71).  One launch of the chain kernel in crossfade mode (8 + 8 read, 8 written per pixel) when every
 * window is the whole output frame; otherwise the same three nodes on f32 scratch frames, with the reference's
 * region walk. */
CVS_EXPORT int cvs_mix_cross_f16_dev(rgba_frame_f16 *out, const rgba_frame_f16 *a, const rgba_frame_f16 *b, float mix_b, cvs_stream_t stream) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return -1; }
    CVS_REQUIRE_INSIDE(a, out, "cvs_mix_cross_f16_dev");
    CVS_REQUIRE_INSIDE(b, out, "cvs_mix_cross_f16_dev");
    hipStream_t s = cvs_pick_stream(stream);
    mix_b = clampf(mix_b, 0.0f, 1.0f);
    const cvs_chain_job job = { out, { a, b }, 2 };
    if (job_is_fusable(&job)) {
        cvk_chain_job rec;
        memset(&rec, 0, sizeof rec);
        rec.out = out->data; rec.layer[0] = a->data; rec.layer[1] = b->data; rec.nlayers = 2;
        rec.npixels = cvs_box_pixels(&out->full_window);
        int rc = CVK(cvk_chain_cross)(&rec, 1, 1.0f - mix_b, mix_b, cvs_cus(), s);
        if (rc != 0) { cvs_set_error("crossfade kernel launch failed: %s", hipGetErrorString((hipError_t)rc)); box2i_set_empty(&out->current_window); return rc; }
        out->current_window = out->full_window;
        t_last_fused = 1;
        return 0;
    }
    t_last_fused = 0;
    const box2i *fw = &out->full_window;
    const size_t n = cvs_box_pixels(fw);
    if (!n) { box2i_set_empty(&out->current_window); return 0; }
    rgba_frame_f32 fa = { cvs_pool_malloc(n * sizeof(rgba_f32), s), *fw, *fw }, fb = { cvs_pool_malloc(n * sizeof(rgba_f32), s), *fw, *fw };
    rgba_frame_f32 fo = { cvs_pool_malloc(n * sizeof(rgba_f32), s), *fw, *fw };
    int rc = (fa.data && fb.data && fo.data) ? 0 : -1;
    /* each input as its f32 pull would deliver it: clipped to the output's buffer, then widened */
    rgba_frame_f16 clip = { cvs_pool_malloc(n * sizeof(rgba_f16), s), *fw, *fw };
    if (!clip.data) rc = -1;
    if (rc == 0) rc = cvs_copy_frame_f16_dev(&clip, a, s);
    if (rc == 0) rc = cvs_frame_f16_to_f32_dev(&fa, &clip, s);
    if (rc == 0) rc = cvs_copy_frame_f16_dev(&clip, b, s);
    if (rc == 0) rc = cvs_frame_f16_to_f32_dev(&fb, &clip, s);
    if (rc == 0) rc = cvs_mix_cross_f32_dev(&fo, &fa, &fb, mix_b, s);
    if (rc == 0) rc = cvs_frame_f32_to_f16_dev(out, &fo, s);
    cvs_pool_free(clip.data, s); cvs_pool_free(fa.data, s); cvs_pool_free(fb.data, s); cvs_pool_free(fo.data, s);
    if (rc != 0) box2i_set_empty(&out->current_window);
    return rc;
}
