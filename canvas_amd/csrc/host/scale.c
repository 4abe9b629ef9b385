/*
 * scale.c -- separable resampling and blur: planning on the host, pixels in kernels/fir_ops.hip.
 *
 * video_scale_bilinear_f32 / _pull follow src/cprocess/video_scale.c:231-319 decision for
 * decision: the identity shortcuts, smaller factor first (:252), the intermediate frame's window
 * (:256-262, :271-277 -- derived with `* factor`, so a two-axis downscale only covers part of the
 * target; kept, it is what the reference outputs), the zero fill, the window of lines actually
 * touched (:124, :191, :225), the pull rectangle (:303-309).
 *
 * Per pass the reference regenerates a triangle filter per line with the fractional offset of
 * that line's centre (:67-71, :97-101).  plan_pass() runs the same generator on the host and
 * records, per target line, which source lines feed it and with what weight, in ascending source
 * order; that table (a few KiB) is uploaded and the gather kernel does the rest.
 *
 * The FIR blur and the Lanczos resampler have no counterpart to follow (the reference has no blur,
 * and filter_createLanczos has no caller): they are defined in DESIGN.md and reuse the same kernel.
 */
#define _GNU_SOURCE
#include "internal.h"
#include <limits.h>
#include <math.h>
#include <stdatomic.h>

static _Atomic int g_fir_path = CVS_FIR_PATH_AUTO;

/* a * b + c where the reference's C has it in ONE expression (video_scale.c:65,257-277): rounded twice as its gcc build does,
 * or once -- the fused multiply-add its clang build emits -- in the contracted flavour (canvas_hip.h cvs_set_arithmetic).
 * This file is compiled with -ffp-contract=off, so the first form never fuses by itself. */
static inline float madd_as(int contracted, float a, float b, float c) { return contracted ? fmaf(a, b, c) : a * b + c; }
static _Thread_local int t_scale_fused;         /* the calling thread's last scaler call ran both passes in one launch */
CVS_EXPORT int cvs_scale_last_was_fused(void) { return t_scale_fused; }
static _Thread_local int t_fir_kernel;          /* CVS_FIR_KERNEL_*: the kernel the calling thread's last FIR launch went to */
CVS_EXPORT int cvs_fir_last_kernel(void) { return t_fir_kernel; }
/* A fused kernel that was chosen for a table pair and then did not launch: the next kernel in line still computes the
 * same pixels, 3-7x slower -- that must not go unnoticed. */
/* (Goes to the log handler only: the call then succeeds on the next kernel, and a successful call leaves no message in
 * cvs_last_error(); cvs_fir_last_kernel() and cvs_fir_fell_through_count() are the programmatic signals.) */
static _Thread_local int t_fell_through;        /* launches of the calling thread that went to the next kernel in line */
CVS_EXPORT int cvs_fir_fell_through_count(void) { return t_fell_through; }
static void fir_launch_fell_through(const char *kernel, int rc) {
    (void)hipGetLastError();
    t_fell_through++;
    cvs_log_warning("%s did not launch (%s): falling back to the next FIR kernel", kernel, hipGetErrorString((hipError_t)rc));
}
CVS_EXPORT void cvs_fir_path_override(int mode) { atomic_store(&g_fir_path, mode & (CVS_FIR_PATH_PASSES | CVS_FIR_PATH_TILED | CVS_FIR_PATH_TABLES | CVS_FIR_PATH_HV | CVS_FIR_PATH_ONE_COLUMN | CVS_FIR_PATH_TWO_COLUMNS | CVS_FIR_PATH_STRIPS | CVS_FIR_PATH_TILES)); }

typedef struct {
    int t0, t1;            /* target lines covered by the table */
    int stride;
    int *ntaps, *tap_src;
    float *taps;
    int used_lo, used_hi;  /* target lines that received at least one tap */
} tap_table;

static void table_free(tap_table *tb) { free(tb->ntaps); free(tb->tap_src); free(tb->taps); memset(tb, 0, sizeof *tb); }

static int table_alloc(tap_table *tb, int t0, int t1, int stride) {
    memset(tb, 0, sizeof *tb);
    tb->t0 = t0; tb->t1 = t1; tb->stride = stride > 0 ? stride : 1;
    tb->used_lo = INT_MAX; tb->used_hi = INT_MIN;
    size_t lines = t1 >= t0 ? (size_t)(t1 - t0 + 1) : 0;
    tb->ntaps = calloc(lines ? lines : 1, sizeof(int));
    tb->tap_src = calloc((lines ? lines : 1) * (size_t)tb->stride, sizeof(int));
    tb->taps = calloc((lines ? lines : 1) * (size_t)tb->stride, sizeof(float));
    if (!tb->ntaps || !tb->tap_src || !tb->taps) { table_free(tb); return -1; }
    return 0;
}

static inline void table_add(tap_table *tb, int t, int s, float c) {
    int row = t - tb->t0, k = tb->ntaps[row];
    if (k < tb->stride) { tb->tap_src[(size_t)row * tb->stride + k] = s; tb->taps[(size_t)row * tb->stride + k] = c; }
    tb->ntaps[row] = k + 1;
}

/* widest triangle the per-line generator can return (video_scale.c:52-57) */
static int triangle_cap(float factor) {
    float dummy = factor;
    fir_filter probe = { &dummy, 0, 0 };
    filter_createTriangle(factor, 0.0f, &probe);
    return probe.width + 3;
}

/* Tap table of one triangle pass.  count_touch: whether an in-range tap marks its target line as
 * used even when the other axis is empty (true for the vertical pass, :88-89; the horizontal pass
 * only marks inside its row loop, :186-187). */
static int plan_triangle(tap_table *tb, float tmin, float smin, float factor, int s0, int s1, int t0, int t1, bool count_touch, int contracted) {
    const int cap = triangle_cap(factor);
    float *buf = malloc(sizeof(float) * (size_t)cap);
    if (!buf) return -1;
    fir_filter f = { buf, 0, 0 };
    int rc = 0;
    if (factor > 1.0f) {
        /* scatter form: how many source lines can land on one target line?  count first */
        int lines = t1 >= t0 ? t1 - t0 + 1 : 0;
        int *count = calloc((size_t)(lines ? lines : 1), sizeof(int));
        if (!count) { free(buf); return -1; }
        for (int pass = 0; pass < 2 && rc == 0; pass++) {
            if (pass == 1) {
                int most = 1;
                for (int i = 0; i < lines; i++) if (count[i] > most) most = count[i];
                rc = table_alloc(tb, t0, t1, most);
                if (rc != 0) break;
            }
            for (int s = s0; s <= s1; s++) {
                float centre_f = madd_as(contracted, s - smin, factor, tmin);       /* video_scale.c:65 */
                int centre = (int)floor(centre_f);
                f.width = cap;
                filter_createTriangle(factor, centre_f - centre, &f);
                for (int k = 0; k < f.width; k++) {
                    int t = centre - f.center + k;
                    if (t < t0 || t > t1) continue;
                    if (pass == 0) count[t - t0]++;
                    else {
                        table_add(tb, t, s, buf[k]);
                        if (count_touch) { if (t < tb->used_lo) tb->used_lo = t; if (t > tb->used_hi) tb->used_hi = t; }
                    }
                }
            }
        }
        free(count);
    } else {
        rc = table_alloc(tb, t0, t1, cap);
        for (int t = t0; rc == 0 && t <= t1; t++) {
            float centre_f = (t - tmin) / factor + smin;
            int centre = (int)floor(centre_f);
            f.width = cap;
            filter_createTriangle(factor, centre_f - centre, &f);
            for (int k = 0; k < f.width; k++) {
                int s = centre - f.center + k;
                if (s < s0 || s > s1) continue;
                table_add(tb, t, s, buf[k]);
                if (count_touch) { if (t < tb->used_lo) tb->used_lo = t; if (t > tb->used_hi) tb->used_hi = t; }
            }
        }
    }
    free(buf);
    return rc;
}

/* device-resident, cached form of plan_triangle's table (defined with the table cache below) */
static int triangle_table_cached(float tmin, float smin, float factor, int s0, int s1, int t0, int t1, bool count_touch,
                                 cvk_fir_axis *axis, int *used_lo, int *used_hi, int *pin, int *max_foot);
static void axis_done(int pin, hipStream_t s);

/* one pass of video_scale.c:34-127 (axis 0) or :129-229 (axis 1) on device frames */
/* a frame of either pixel format, as the passes see it */
typedef struct { void *data; box2i full, cur; int half; } any_frame;

static size_t any_bytes(const any_frame *f) { return cvs_box_pixels(&f->full) * (f->half ? sizeof(rgba_f16) : sizeof(rgba_f32)); }

static int triangle_pass(any_frame *target, float tmin, const any_frame *source, float smin, float factor, int axis, hipStream_t s) {
    const box2i srect = source->cur, trect = target->full;
    const int lo = axis ? (srect.min.y > trect.min.y ? srect.min.y : trect.min.y) : (srect.min.x > trect.min.x ? srect.min.x : trect.min.x);
    const int hi = axis ? (srect.max.y < trect.max.y ? srect.max.y : trect.max.y) : (srect.max.x < trect.max.x ? srect.max.x : trect.max.x);
    const int s0 = axis ? srect.min.x : srect.min.y, s1 = axis ? srect.max.x : srect.max.y;
    const int t0 = axis ? trect.min.x : trect.min.y, t1 = axis ? trect.max.x : trect.max.y;
    cvk_view tv = cvs_view(target->data, &target->full), sv = cvs_view(source->data, &source->full);

    /* the per-line taps depend only on the geometry, which repeats from frame to frame: planned once, kept on the
     * device; steady state is one gather launch (and the zero fill, when the pass leaves part of the target alone),
     * nothing synchronous */
    cvk_fir_axis table;
    int used_lo, used_hi, pin = -1;
    int rc = triangle_table_cached(tmin, smin, factor, s0, s1, t0, t1, axis == 0 || lo <= hi, &table, &used_lo, &used_hi, &pin, NULL);
    if (rc != 0) return rc;
    /* video_scale.c:25-32,44: the target starts as zeros (all-zero bytes are 0.0 in either format).  The gather writes every
     * pixel of lines used_lo..used_hi x lo..hi (a line without taps gets its zeros there): when that is the whole buffer --
     * the usual case -- the fill would only be overwritten (an enlarging 4K pass: 500 MB of the 1.3 GB it moved). */
    {
        const int o0 = axis ? trect.min.y : trect.min.x, o1 = axis ? trect.max.y : trect.max.x;
        const bool covers = used_hi >= used_lo && hi >= lo && used_lo == t0 && used_hi == t1 && lo == o0 && hi == o1;
        if (!covers && any_bytes(target)) {
            hipError_t e_ = hipMemsetAsync(target->data, 0, any_bytes(target), s);
            if (e_ != hipSuccess) { axis_done(pin, s); cvs_set_error("zero fill: %s", hipGetErrorString(e_)); return -1; }
        }
    }
    if (used_hi >= used_lo && hi >= lo) {
        /* the gather reads source lines named in the table; they lie inside the source window by construction */
        const size_t first = (size_t)(used_lo - t0);
        cvk_fir_params fp;
        memset(&fp, 0, sizeof fp);
        fp.target = tv; fp.source = sv; fp.axis = axis;
        fp.t0 = used_lo; fp.t1 = used_hi; fp.lo = lo; fp.hi = hi;
        fp.ntaps = table.ntaps + first; fp.tap_src = table.src + first * (size_t)table.stride; fp.taps = table.taps + first * (size_t)table.stride;
        fp.stride = table.stride;
        fp.in_half = source->half; fp.out_half = target->half;
        rc = CVK(cvk_fir_gather)(&fp, s);
        if (rc == 0) t_fir_kernel = CVS_FIR_KERNEL_PASS;
    }
    axis_done(pin, s);
    if (rc != 0) { cvs_set_error("FIR gather launch failed: %s", hipGetErrorString((hipError_t)rc)); return rc; }
    if (axis) box2i_set(&target->cur, used_lo, lo, used_hi, hi);
    else      box2i_set(&target->cur, lo, used_lo, hi, used_hi);
    return 0;
}

/* Both passes of video_scale_bilinear_f32 in one launch when the vertical pass comes first (equal factors, or the vertical
 * one smaller: video_scale.c:252) -- sweep_vh_ops.hip.  Same tables, same windows as the two triangle_pass calls below
 * would use: the frame between the passes (`mid_full`, video_scale.c:254-272) never exists, only its geometry does.
 * 0 = done, 1 = not for this kernel (the caller runs the two passes), < 0 = error. */
/* A batch call (cvs_scale_bilinear_*_batch_dev) in progress on this thread: the frames that go into the launch with the one
 * scale_core is called for.  Only the tile kernel takes a batch; anything else answers 2 = "frame by frame". */
typedef struct { int n; const void *source[8]; void *target[8]; size_t target_bytes; } scale_batch;
static _Thread_local const scale_batch *t_scale_batch;

static int triangle_fused_vh(any_frame *target, v2f tp, const any_frame *source, v2f sp, v2f fac, const box2i *mid_full, hipStream_t s) {
    const box2i *tf = &target->full, *sc = &source->cur;
    /* pass 1 (vertical) into the frame between the passes */
    const int lo1 = sc->min.x > mid_full->min.x ? sc->min.x : mid_full->min.x, hi1 = sc->max.x < mid_full->max.x ? sc->max.x : mid_full->max.x;
    cvk_fir_axis tv, th;
    int vlo, vhi, hlo, hhi, pv = -1, ph = -1, hfoot = 0;
    if (hi1 < lo1) return 1;
    if (triangle_table_cached(tp.y, sp.y, fac.y, sc->min.y, sc->max.y, mid_full->min.y, mid_full->max.y, true, &tv, &vlo, &vhi, &pv, NULL) != 0) return -1;
    int rc = 1;
    if (vhi >= vlo) {
        /* pass 2 (horizontal) from that frame's window (lo1..hi1 x vlo..vhi) into the target */
        const int lo2 = vlo > tf->min.y ? vlo : tf->min.y, hi2 = vhi < tf->max.y ? vhi : tf->max.y;
        if (hi2 >= lo2 && triangle_table_cached(tp.x, sp.x, fac.x, lo1, hi1, tf->min.x, tf->max.x, true, &th, &hlo, &hhi, &ph, &hfoot) == 0) {
            cvk_fir2d_params fp;
            memset(&fp, 0, sizeof fp);
            fp.target = cvs_view(target->data, tf);
            fp.source = cvs_view(source->data, &source->full);
            fp.in_half = source->half; fp.out_half = target->half;
            fp.tx0 = tf->min.x; fp.tx1 = tf->max.x;                  /* every column: those without taps are zeros, as the fill leaves them */
            fp.ty0 = mid_full->min.y; fp.ty1 = hi2;                   /* the vertical table's lines; lo2 .. hi2 of them are produced */
            fp.h = th; fp.v = tv;
            fp.max_sw = hfoot > 0 ? hfoot : 1;
            /* short lists over few source pixels (enlarging): a workgroup per tile where that is the faster form (pinned:
             * wherever it takes the call); else a wave per strip */
            const int pinned = atomic_load(&g_fir_path);
            const scale_batch *const batch = t_scale_batch;
            /* (a batch is one large target to the kernels: halfs go to the strips, as 4K -> 8K does -- profiles/r04/scaler_batch.txt) */
            const bool tiles = hhi >= hlo && !(pinned & CVS_FIR_PATH_STRIPS) &&
                               ((pinned & CVS_FIR_PATH_TILES) ? CVK(cvk_fir_tvh_supported)(&fp)
                                : batch && target->half ? false : CVK(cvk_fir_tvh_preferred)(&fp));
            if (batch && !(hhi >= hlo && (tiles || CVK(cvk_fir_vh_supported)(&fp)))) rc = 2;
            else if (hhi >= hlo && (tiles || CVK(cvk_fir_vh_supported)(&fp))) {
                /* video_scale.c:25-32,44: rows the pass leaves alone are zeros */
                const bool covers = lo2 == tf->min.y && hi2 == tf->max.y;
                hipError_t e = hipSuccess;
                if (batch) {
                    fp.nframes = batch->n;
                    for (int i = 0; i < batch->n; i++) {
                        fp.frame_source[i] = batch->source[i]; fp.frame_target[i] = batch->target[i];
                        if (!covers && batch->target_bytes && e == hipSuccess) e = hipMemsetAsync(batch->target[i], 0, batch->target_bytes, s);
                    }
                } else if (!covers && any_bytes(target)) e = hipMemsetAsync(target->data, 0, any_bytes(target), s);
                int krc = e != hipSuccess ? (int)e : tiles ? CVK(cvk_fir_tvh)(&fp, lo2 - fp.ty0, s) : CVK(cvk_fir_vh)(&fp, lo2 - fp.ty0, cvs_cus(), s);
                if (krc == 0) { box2i_set(&target->cur, hlo, lo2, hhi, hi2); t_fir_kernel = tiles ? CVS_FIR_KERNEL_TILE_VH : CVS_FIR_KERNEL_VH; rc = 0; }
                else { fir_launch_fell_through(tiles ? "k_fir_tile_vh" : "k_fir_vh", krc); rc = 1; }   /* did not launch: the two passes decide */
            }
        }
    }
    axis_done(pv, s); axis_done(ph, s);
    return rc;
}

/* The other order (horizontal factor strictly smaller: the horizontal pass first) is the order of sweep_hv_ops.hip's kernel.
 * 0 = done, 1 = not for this kernel, < 0 = error. */
static int triangle_fused_hv(any_frame *target, v2f tp, const any_frame *source, v2f sp, v2f fac, const box2i *mid_full, hipStream_t s) {
    const box2i *tf = &target->full, *sc = &source->cur;
    /* pass 1 (horizontal) into the frame between the passes: rows lo1..hi1 */
    const int lo1 = sc->min.y > mid_full->min.y ? sc->min.y : mid_full->min.y, hi1 = sc->max.y < mid_full->max.y ? sc->max.y : mid_full->max.y;
    cvk_fir_axis tv, th;
    int vlo, vhi, hlo, hhi, pv = -1, ph = -1, hfoot = 0;
    if (hi1 < lo1) return 1;
    if (cvs_arith() != CVS_ARITH_SEPARATE) return 1;           /* the horizontal-first sweep exists in the plain flavour only: the two passes */
    if (triangle_table_cached(tp.x, sp.x, fac.x, sc->min.x, sc->max.x, mid_full->min.x, mid_full->max.x, true, &th, &hlo, &hhi, &ph, &hfoot) != 0) return -1;
    int rc = 1;
    if (hhi >= hlo) {
        /* pass 2 (vertical) from that frame's window (hlo..hhi x lo1..hi1) into the target: columns lo2..hi2 */
        const int lo2 = hlo > tf->min.x ? hlo : tf->min.x, hi2 = hhi < tf->max.x ? hhi : tf->max.x;
        if (hi2 >= lo2 && triangle_table_cached(tp.y, sp.y, fac.y, lo1, hi1, tf->min.y, tf->max.y, true, &tv, &vlo, &vhi, &pv, NULL) == 0) {
            cvk_fir2d_params fp;
            memset(&fp, 0, sizeof fp);
            fp.target = cvs_view(target->data, tf);
            fp.source = cvs_view(source->data, &source->full);
            fp.in_half = source->half; fp.out_half = target->half;
            fp.tx0 = mid_full->min.x; fp.tx1 = hi2;                   /* from the horizontal table's first column; those without taps are zeros */
            fp.ty0 = tf->min.y; fp.ty1 = tf->max.y;                   /* every line: those without taps are written as zeros */
            fp.h = th; fp.v = tv;
            fp.max_sw = hfoot > 0 ? hfoot : 1;
            if (vhi >= vlo && cvk_fir_hv_supported(&fp)) {
                const bool covers = fp.tx0 == tf->min.x && hi2 == tf->max.x;
                hipError_t e = covers || !any_bytes(target) ? hipSuccess : hipMemsetAsync(target->data, 0, any_bytes(target), s);
                int krc = e != hipSuccess ? (int)e : cvk_fir_hv(&fp, cvs_cus(), s);
                if (krc == 0) { box2i_set(&target->cur, lo2, vlo, hi2, vhi); t_fir_kernel = CVS_FIR_KERNEL_HV; rc = 0; }
                else { fir_launch_fell_through("k_fir_hv", krc); rc = 1; }
            }
        }
    }
    axis_done(pv, s); axis_done(ph, s);
    return rc;
}

/* video_scale_bilinear_f32 (video_scale.c:231-286) between frames of either format: f16 sources are widened as they
 * are read, f16 targets truncated as they are written (what the pulls around an f32 scaler node do, main.c:43-71,
 * 105-144); the frame between the two passes is always f32.  The caller has dealt with the all-identity case. */
static int scale_core(any_frame *target, v2f tp, const any_frame *source, v2f sp, v2f fac, hipStream_t s) {
    t_scale_fused = 0;
    t_fir_kernel = CVS_FIR_KERNEL_NONE;
    /* a batch goes through the vertical-first fused launch or not at all (2: the caller does its frames one by one) */
    if (t_scale_batch && ((fac.x == 1.0f && tp.x == sp.x) || (fac.y == 1.0f && tp.y == sp.y) || fac.x < fac.y ||
                          (atomic_load(&g_fir_path) & (CVS_FIR_PATH_PASSES | CVS_FIR_PATH_TILED)))) return 2;
    if (fac.x == 1.0f && tp.x == sp.x) return triangle_pass(target, tp.y, source, sp.y, fac.y, 0, s);
    if (fac.y == 1.0f && tp.y == sp.y) return triangle_pass(target, tp.x, source, sp.x, fac.x, 1, s);

    any_frame mid;
    memset(&mid, 0, sizeof mid);
    const box2i *tf = &target->full, *sc = &source->cur;
    const bool x_first = fac.x < fac.y;
    const int fl = cvs_arith() == CVS_ARITH_CONTRACTED;       /* video_scale.c:257-277: sp - d * fac and sp + d * fac, one expression each */
    if (x_first)
        box2i_set(&mid.full, (int)madd_as(fl, -(tp.x - tf->min.x), fac.x, sp.x), sc->min.y,
                  (int)madd_as(fl, tf->max.x - tp.x, fac.x, sp.x), sc->max.y);
    else
        box2i_set(&mid.full, sc->min.x, (int)madd_as(fl, -(tp.y - tf->min.y), fac.y, sp.y),
                  sc->max.x, (int)madd_as(fl, tf->max.y - tp.y, fac.y, sp.y));
    box2i_intersect(&mid.full, &mid.full, tf);
    mid.cur = mid.full;
    if (!(atomic_load(&g_fir_path) & (CVS_FIR_PATH_PASSES | CVS_FIR_PATH_TILED))) {
        int rc = x_first ? triangle_fused_hv(target, tp, source, sp, fac, &mid.full, s) : triangle_fused_vh(target, tp, source, sp, fac, &mid.full, s);
        if (rc == 0) t_scale_fused = 1;
        if (rc <= 0) return rc;                              /* done, or failed; 1: not for the fused kernel */
        if (t_scale_batch) return 2;
    }
    size_t n = cvs_box_pixels(&mid.full);
    mid.data = cvs_pool_malloc(sizeof(rgba_f32) * (n ? n : 1), s);
    if (!mid.data) return -1;
    int rc;
    if (x_first) { rc = triangle_pass(&mid, tp.x, source, sp.x, fac.x, 1, s); if (rc == 0) rc = triangle_pass(target, tp.y, &mid, sp.y, fac.y, 0, s); }
    else         { rc = triangle_pass(&mid, tp.y, source, sp.y, fac.y, 0, s); if (rc == 0) rc = triangle_pass(target, tp.x, &mid, sp.x, fac.x, 1, s); }
    cvs_pool_free(mid.data, s);
    return rc;
}

CVS_EXPORT int cvs_scale_bilinear_f32_dev(rgba_frame_f32 *target, v2f tp, const rgba_frame_f32 *source, v2f sp, v2f fac, cvs_stream_t stream) {
    if (cvs_enter() != 0) { box2i_set_empty(&target->current_window); return -1; }
    CVS_REQUIRE_INSIDE(source, target, "cvs_scale_bilinear_f32_dev");
    hipStream_t s = cvs_pick_stream(stream);
    if (fac.x == 1.0f && tp.x == sp.x && fac.y == 1.0f && tp.y == sp.y) return cvs_copy_frame_alpha_f32_dev(target, source, 1.0f, s);
    any_frame t = { target->data, target->full_window, target->full_window, 0 };
    const any_frame src = { source->data, source->full_window, source->current_window, 0 };
    int rc = scale_core(&t, tp, &src, sp, fac, s);
    target->current_window = t.cur;
    if (rc != 0) box2i_set_empty(&target->current_window);
    return rc;
}

/* The same scaler between two f16 frames: the f16 pull of a scaler node whose input is a half-native source, without
 * the widened copy of the input and the f32 copy of the output ever existing. */
CVS_EXPORT int cvs_scale_bilinear_f16_dev(rgba_frame_f16 *target, v2f tp, const rgba_frame_f16 *source, v2f sp, v2f fac, cvs_stream_t stream) {
    if (cvs_enter() != 0) { box2i_set_empty(&target->current_window); return -1; }
    CVS_REQUIRE_INSIDE(source, target, "cvs_scale_bilinear_f16_dev");
    hipStream_t s = cvs_pick_stream(stream);
    if (fac.x == 1.0f && tp.x == sp.x && fac.y == 1.0f && tp.y == sp.y) return cvs_copy_frame_f16_dev(target, source, s);
    any_frame t = { target->data, target->full_window, target->full_window, 1 };
    const any_frame src = { (void *)source->data, source->full_window, source->current_window, 1 };
    int rc = scale_core(&t, tp, &src, sp, fac, s);
    target->current_window = t.cur;
    if (rc != 0) box2i_set_empty(&target->current_window);
    return rc;
}

/* cvs_scale_bilinear_f16_dev / _f32_dev for `count` INDEPENDENT frames (a pull queue's frames in flight): frames of one geometry
 * that do not overlap go up to eight at a time into ONE launch of the tile kernel (grid.z = frame), where that kernel takes
 * the call (vertical pass first, short tap lists: enlarging); every other combination is carried out frame by frame, exactly
 * as `count` single calls.  Results are those of the single calls, bit for bit. */
static bool same_box(const box2i *a, const box2i *b);
static bool batch_has_hazard(void *const *outs, size_t out_bytes, const void *const *ins, size_t in_bytes, int nouts, int nins);

static int scale_batch_any(void *const *tdata, const box2i *const *tfull, box2i *const *tcur, const void *const *sdata, const box2i *const *sfull,
                           const box2i *const *scur, int count, int half, v2f tp, v2f sp, v2f fac, hipStream_t s, int *done_out) {
    int done = 0;
    bool uniform = count > 1 && !box2i_is_empty(scur[0]) && !box2i_is_empty(tfull[0]) &&
                   !(fac.x == 1.0f && tp.x == sp.x && fac.y == 1.0f && tp.y == sp.y);
    for (int i = 1; uniform && i < count; i++)
        uniform = same_box(tfull[i], tfull[0]) && same_box(sfull[i], sfull[0]) && same_box(scur[i], scur[0]);
    const size_t px = half ? sizeof(rgba_f16) : sizeof(rgba_f32);
    const size_t tbytes = cvs_box_pixels(tfull[0]) * px, sbytes = cvs_box_pixels(sfull[0]) * px;
    while (uniform && done < count) {
        const int n = count - done < 8 ? count - done : 8;
        scale_batch b;
        memset(&b, 0, sizeof b);
        for (int i = 0; i < n; i++) { b.source[i] = sdata[done + i]; b.target[i] = tdata[done + i]; }
        if (n < 2 || batch_has_hazard(b.target, tbytes, b.source, sbytes, n, n)) break;      /* the rest frame by frame */
        b.n = n; b.target_bytes = tbytes;
        any_frame t = { tdata[done], *tfull[done], *tfull[done], half };
        const any_frame src = { (void *)sdata[done], *sfull[done], *scur[done], half };
        t_scale_batch = &b;
        const int rc = scale_core(&t, tp, &src, sp, fac, s);
        t_scale_batch = NULL;
        if (rc == 2) break;
        if (rc != 0) { *done_out = done; return -1; }
        for (int i = 0; i < n; i++) *tcur[done + i] = t.cur;
        done += n;
    }
    *done_out = done;
    return 0;
}

CVS_EXPORT int cvs_scale_bilinear_f16_batch_dev(rgba_frame_f16 *const *targets, v2f tp, const rgba_frame_f16 *const *sources, v2f sp, v2f fac, int count, cvs_stream_t stream) {
    if (count <= 0) return 0;
    if (!targets || !sources) { cvs_set_error("cvs_scale_bilinear_f16_batch_dev: bad arguments"); return -1; }
    for (int i = 0; i < count; i++)
        if (!targets[i] || !sources[i]) { cvs_set_error("cvs_scale_bilinear_f16_batch_dev: frame %d of %d is a null pointer", i, count); return -1; }
    /* what count single calls would refuse, the batch refuses -- before any launch */
    for (int i = 0; i < count; i++)
        if (!cvs_box_contains(&sources[i]->full_window, &sources[i]->current_window)) {
            cvs_set_error("cvs_scale_bilinear_f16_batch_dev: the input's current_window lies outside its buffer (frame %d)", i);
            for (int k = 0; k < count; k++) box2i_set_empty(&targets[k]->current_window);
            return -1;
        }
    if (cvs_enter() != 0) { for (int i = 0; i < count; i++) box2i_set_empty(&targets[i]->current_window); return -1; }
    hipStream_t s = cvs_pick_stream(stream);
    int rc = 0, done = 0;
    if (count > 1) {
        void *td[64]; const void *sd[64]; const box2i *tf[64], *sf[64], *sc[64]; box2i *tc[64];
        for (int base = 0; rc == 0 && base < count; ) {
            const int n = count - base < 64 ? count - base : 64;
            for (int i = 0; i < n; i++) {
                td[i] = targets[base + i]->data; sd[i] = sources[base + i]->data; tf[i] = &targets[base + i]->full_window;
                sf[i] = &sources[base + i]->full_window; sc[i] = &sources[base + i]->current_window; tc[i] = &targets[base + i]->current_window;
            }
            int d = 0;
            rc = scale_batch_any(td, tf, tc, sd, sf, sc, n, 1, tp, sp, fac, s, &d);
            done = base + d;
            if (d < n) break;                                /* from here on frame by frame */
            base += n;
        }
    }
    for (; rc == 0 && done < count; done++) rc = cvs_scale_bilinear_f16_dev(targets[done], tp, sources[done], sp, fac, stream);
    if (rc != 0) for (int i = done; i < count; i++) box2i_set_empty(&targets[i]->current_window);
    return rc;
}

CVS_EXPORT int cvs_scale_bilinear_f32_batch_dev(rgba_frame_f32 *const *targets, v2f tp, const rgba_frame_f32 *const *sources, v2f sp, v2f fac, int count, cvs_stream_t stream) {
    if (count <= 0) return 0;
    if (!targets || !sources) { cvs_set_error("cvs_scale_bilinear_f32_batch_dev: bad arguments"); return -1; }
    for (int i = 0; i < count; i++)
        if (!targets[i] || !sources[i]) { cvs_set_error("cvs_scale_bilinear_f32_batch_dev: frame %d of %d is a null pointer", i, count); return -1; }
    for (int i = 0; i < count; i++)
        if (!cvs_box_contains(&sources[i]->full_window, &sources[i]->current_window)) {
            cvs_set_error("cvs_scale_bilinear_f32_batch_dev: the input's current_window lies outside its buffer (frame %d)", i);
            for (int k = 0; k < count; k++) box2i_set_empty(&targets[k]->current_window);
            return -1;
        }
    if (cvs_enter() != 0) { for (int i = 0; i < count; i++) box2i_set_empty(&targets[i]->current_window); return -1; }
    hipStream_t s = cvs_pick_stream(stream);
    int rc = 0, done = 0;
    if (count > 1) {
        void *td[64]; const void *sd[64]; const box2i *tf[64], *sf[64], *sc[64]; box2i *tc[64];
        for (int base = 0; rc == 0 && base < count; ) {
            const int n = count - base < 64 ? count - base : 64;
            for (int i = 0; i < n; i++) {
                td[i] = targets[base + i]->data; sd[i] = sources[base + i]->data; tf[i] = &targets[base + i]->full_window;
                sf[i] = &sources[base + i]->full_window; sc[i] = &sources[base + i]->current_window; tc[i] = &targets[base + i]->current_window;
            }
            int d = 0;
            rc = scale_batch_any(td, tf, tc, sd, sf, sc, n, 0, tp, sp, fac, s, &d);
            done = base + d;
            if (d < n) break;
            base += n;
        }
    }
    for (; rc == 0 && done < count; done++) rc = cvs_scale_bilinear_f32_dev(targets[done], tp, sources[done], sp, fac, stream);
    if (rc != 0) for (int i = done; i < count; i++) box2i_set_empty(&targets[i]->current_window);
    return rc;
}

CVS_EXPORT void video_scale_bilinear_f32(rgba_frame_f32 *target, v2f tp, rgba_frame_f32 *source, v2f sp, v2f fac) {
    if (cvs_enter() != 0) { box2i_set_empty(&target->current_window); return; }
    hipStream_t s = cvs_pick_stream(NULL);
    cvs_staged d_t = { 0 }, d_s = { 0 };
    rgba_frame_f32 ft = *target, fs = *source;
    int rc = cvs_stage_in(&d_t, target->data, cvs_box_pixels(&target->full_window) * sizeof(rgba_f32), 1, s);
    if (rc == 0) rc = cvs_stage_in(&d_s, source->data, cvs_box_pixels(&source->full_window) * sizeof(rgba_f32), !box2i_is_empty(&source->current_window), s);
    ft.data = d_t.dev; fs.data = d_s.dev;
    if (rc == 0) rc = cvs_scale_bilinear_f32_dev(&ft, tp, &fs, sp, fac, s);
    if (rc == 0) rc = cvs_stage_out(&d_t, target->data, s);
    target->current_window = ft.current_window;
    if (rc != 0) box2i_set_empty(&target->current_window);
    cvs_stage_free(&d_t); cvs_stage_free(&d_s);
}

CVS_EXPORT void video_scale_bilinear_f32_pull(rgba_frame_f32 *target, v2f tp, video_source *source, int frame,
                                              box2i *source_rect, v2f sp, v2f fac) {
    if (fac.x == 0.0f || fac.y == 0.0f) { box2i_set_empty(&target->current_window); return; }   /* video_scale.c:290-293 */
    if (fac.x == 1.0f && fac.y == 1.0f && tp.x == sp.x && tp.y == sp.y) { video_get_frame_f32(source, frame, target); return; }
    const box2i *tf = &target->full_window;
    rgba_frame_f32 tmp;
    box2i_set(&tmp.full_window,
              (int)(sp.x - (tp.x - tf->min.x) / fac.x) - 1, (int)(sp.y - (tp.y - tf->min.y) / fac.y) - 1,
              (int)(sp.x + (tf->max.x - tp.x) / fac.x) + 1, (int)(sp.y + (tf->max.y - tp.y) / fac.y) + 1);
    box2i_intersect(&tmp.full_window, &tmp.full_window, source_rect);
    tmp.current_window = tmp.full_window;
    size_t n = cvs_box_pixels(&tmp.full_window);
    tmp.data = malloc(sizeof(rgba_f32) * (n ? n : 1));
    if (!tmp.data) { box2i_set_empty(&target->current_window); return; }
    video_get_frame_f32(source, frame, &tmp);
    video_scale_bilinear_f32(target, tp, &tmp, sp, fac);
    free(tmp.data);
}

/* ---------------------------------------------------------------- FIR blur (repo-defined, DESIGN.md "A11")
 * Odd or even tap count, centre = ntaps/2; horizontal then vertical; accumulate from 0.0f in
 * ascending tap order; taps falling outside the source's current_window are skipped; output window =
 * source.current ∩ target.full. */
static int plan_blur(tap_table *tb, int t0, int t1, int s0, int s1, const float *taps, int ntaps) {
    int rc = table_alloc(tb, t0, t1, ntaps);
    const int c = ntaps / 2;
    for (int t = t0; rc == 0 && t <= t1; t++)
        for (int k = 0; k < ntaps; k++) {
            int sidx = t - c + k;
            if (sidx < s0 || sidx > s1) continue;
            table_add(tb, t, sidx, taps[k]);
        }
    tb->used_lo = t0; tb->used_hi = t1;
    return rc;
}

/* ---------------------------------------------------------------- Lanczos gather resample (repo-defined)
 * Per target line the taps come from filter_createLanczos(factor, kernel_size, frac(centre)) with
 * centre = t / factor (origin 0 on both sides); x pass then y pass; f32 accumulate from 0. */
static int plan_lanczos(tap_table *tb, int t0, int t1, int s0, int s1, float factor, int ksize) {
    fir_filter probe = { NULL, 0, 0 };
    filter_createLanczos(factor, ksize, 0.0f, &probe);
    int cap = probe.width + 3;
    filter_free(&probe);
    int rc = table_alloc(tb, t0, t1, cap);
    for (int t = t0; rc == 0 && t <= t1; t++) {
        float centre_f = (float)t / factor;
        int centre = (int)floor(centre_f);
        fir_filter f = { NULL, 0, 0 };
        filter_createLanczos(factor, ksize, centre_f - centre, &f);
        if (!f.coeff) { rc = -1; break; }
        for (int k = 0; k < f.width; k++) {
            int sidx = centre - f.center + k;
            if (sidx < s0 || sidx > s1) continue;
            table_add(tb, t, sidx, f.coeff[k]);
        }
        filter_free(&f);
    }
    tb->used_lo = t0; tb->used_hi = t1;
    return rc;
}

/* ---------------------------------------------------------------- fused separable FIR (kernels/fir_ops.hip: k_fir2d)
 *
 * One tap table per axis, built by the planners above, kept on the device and reused: a table depends
 * only on (kind, factor or taps, target range, source range), which repeat from frame to frame, so in
 * steady state a blur or a resample is one kernel launch with no host-side planning, no upload, no sync. */
#include <pthread.h>

typedef struct {
    int kind;                 /* 1 = blur taps, 2 = lanczos, 3 = triangle resample */
    uint32_t fbits;           /* lanczos, triangle: factor bits */
    int ksize;                /* lanczos: kernel size; blur: tap count; triangle: count_touch */
    uint64_t taps_hash;       /* blur: FNV-1a of the tap values; triangle: tmin bits << 32 | smin bits */
    int t0, t1, s0, s1;
    int tile;                 /* tile edge along this axis */
    int flavour;              /* triangle, enlarging: the arithmetic flavour the line centres were computed in (0 elsewhere) */
} axis_key;

typedef struct {
    axis_key key;
    int valid;
    int pins;                 /* calls that hold the table's address and have not enqueued their launch yet (+ captured graphs) */
    uint64_t stamp;
    char *dev;                /* one block: ntaps | src | taps | foot */
    cvk_fir_axis axis;
    int max_foot;
    int used_lo, used_hi;     /* target lines that receive at least one tap (the window the pass reports) */
} axis_entry;

typedef struct { const float *taps; float factor, tmin, smin; int contracted; } axis_plan;    /* what the planner of the key's kind needs */

/* The tables live on the device: one cache per device context (runtime.c), one lock over all of them. */
#define AXIS_CACHE 128
#define AXIS_RETIRED 64
typedef struct { axis_entry e[AXIS_CACHE]; uint64_t clock; char *retired[AXIS_RETIRED]; int nretired; } axis_cache;
static axis_cache g_axis_of[CVS_MAX_CONTEXTS];
#define g_axis (g_axis_of[cvs_ctx()].e)
#define g_axis_clock (g_axis_of[cvs_ctx()].clock)
#define g_retired (g_axis_of[cvs_ctx()].retired)
#define g_nretired (g_axis_of[cvs_ctx()].nretired)
static pthread_mutex_t g_axis_lock = PTHREAD_MUTEX_INITIALIZER;

/* keys are compared with memcmp: build them from zeroed storage so that padding is defined */
static axis_key make_key(int kind, uint32_t fbits, int ksize, uint64_t taps_hash, int t0, int t1, int s0, int s1, int tile) {
    axis_key k;
    memset(&k, 0, sizeof k);
    k.kind = kind; k.fbits = fbits; k.ksize = ksize; k.taps_hash = taps_hash;
    k.t0 = t0; k.t1 = t1; k.s0 = s0; k.s1 = s1; k.tile = tile;
    return k;
}

static uint64_t fnv1a(const void *p, size_t n) {
    const unsigned char *b = p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

/* device copy of one axis table (+ per-tile footprints); returns 0 and fills *out on success */
static int axis_upload(const tap_table *tb, int tile, axis_entry *e) {
    const int lines = tb->t1 >= tb->t0 ? tb->t1 - tb->t0 + 1 : 0;
    const int tiles = (lines + tile - 1) / tile;
    const int tiles_pad = ((tiles ? tiles : 1) + 3) & ~3;          /* kernels read four entries at once: spare ones touch nothing */
    int *foot = malloc(sizeof(int) * 2 * (size_t)tiles_pad);
    int *ntaps = malloc(sizeof(int) * (size_t)(lines ? lines : 1));
    if (!foot || !ntaps) { free(foot); free(ntaps); return -1; }
    int max_foot = 0;
    for (int i = 0; i < lines; i++) ntaps[i] = tb->ntaps[i] < tb->stride ? tb->ntaps[i] : tb->stride;
    for (int t = 0; t < tiles; t++) {
        int first = INT_MAX, last = INT_MIN;
        for (int i = t * tile; i < lines && i < (t + 1) * tile; i++) {
            if (!ntaps[i]) continue;
            const int *src = tb->tap_src + (size_t)i * tb->stride;
            if (src[0] < first) first = src[0];
            if (src[ntaps[i] - 1] > last) last = src[ntaps[i] - 1];
        }
        if (last < first) { first = 0; last = -1; }
        foot[2 * t] = first; foot[2 * t + 1] = last;
        if (last - first + 1 > max_foot) max_foot = last - first + 1;
    }
    for (int t = tiles; t < tiles_pad; t++) { foot[2 * t] = 0; foot[2 * t + 1] = -1; }
    /* what the streaming kernel needs to know about the table */
    int max_taps = 0, wide_foot = 0, max_active = 0, streamable = 1;
    {
        int prev_a = INT_MIN, prev_b = INT_MIN;
        for (int i = 0; i < lines; i++) {
            const int n = ntaps[i];
            if (n > max_taps) max_taps = n;
            if (!n) continue;
            const int *src = tb->tap_src + (size_t)i * tb->stride;
            for (int k = 1; k < n; k++) if (src[k] != src[0] + k) streamable = 0;
            if (src[0] < prev_a || src[n - 1] < prev_b) streamable = 0;
            prev_a = src[0]; prev_b = src[n - 1];
        }
        for (int g = 0; g < lines; g += 128) {              /* strips of 128 lines (tile_vh_ops.hip's target columns) */
            int first = INT_MAX, last = INT_MIN;
            for (int i = g; i < lines && i < g + 128; i++) {
                if (!ntaps[i]) continue;
                const int *src = tb->tap_src + (size_t)i * tb->stride;
                if (src[0] < first) first = src[0];
                if (src[ntaps[i] - 1] > last) last = src[ntaps[i] - 1];
            }
            if (last >= first && last - first + 1 > wide_foot) wide_foot = last - first + 1;
        }
        if (streamable) {
            int j = 0;
            for (int i = 0; i < lines; i++) {
                if (!ntaps[i]) continue;
                const int b = tb->tap_src[(size_t)i * tb->stride + ntaps[i] - 1];
                if (j < i) j = i;
                while (j + 1 < lines) {                       /* furthest later line that starts at or before b */
                    int nxt = j + 1;
                    while (nxt < lines && !ntaps[nxt]) nxt++;
                    if (nxt >= lines || tb->tap_src[(size_t)nxt * tb->stride] > b) break;
                    j = nxt;
                }
                if (j - i + 1 > max_active) max_active = j - i + 1;
            }
        }
    }
    /* source lines under any CVK_FIR_TVH_LINES consecutive target lines (tile_vh_ops.hip sizes its LDS rows by it) */
    int span_lines[3] = { 0, 0, 0 };
    if (streamable) {
        for (int g = 0; g < 3; g++) {
            const int run = CVK_FIR_TVH_LINES << g;
            for (int i = 0; i < lines; i++) {
                int first = INT_MAX, last = INT_MIN;
                for (int j = i; j < lines && j < i + run; j++) {
                    if (!ntaps[j]) continue;
                    const int *src = tb->tap_src + (size_t)j * tb->stride;
                    if (first == INT_MAX) first = src[0];
                    last = src[ntaps[j] - 1];
                }
                if (last >= first && last - first + 1 > span_lines[g]) span_lines[g] = last - first + 1;
            }
        }
    }
    /* the table by TARGET line in one record each (kernels.h cvk_fir_axis.lrec): one scalar load per line */
    uint32_t *lrec = NULL;
    if (streamable && max_taps >= 1 && max_taps <= CVK_FIR_LREC - 2) {
        lrec = calloc(((size_t)lines + 1) * CVK_FIR_LREC, sizeof *lrec);      /* + one spare record: the kernel loads a line ahead */
        if (!lrec) { free(foot); free(ntaps); return -1; }
        for (int i = 0; i <= lines; i++) {
            uint32_t *r = lrec + (size_t)i * CVK_FIR_LREC;
            const int n = i < lines ? ntaps[i] : 0;
            r[0] = (uint32_t)n;
            r[1] = n ? (uint32_t)tb->tap_src[(size_t)i * tb->stride] : (uint32_t)INT_MIN;     /* no taps: never moves the window */
            for (int k = 0; k < n; k++) memcpy(&r[2 + k], &tb->taps[(size_t)i * tb->stride + k], 4);
        }
    }
    /* ... and, for short lists, by target line in one aligned read each (kernels.h cvk_fir_axis.pack) */
    const int pack_width = max_taps >= 1 && max_taps <= 2 ? 2 : max_taps <= 4 && max_taps >= 1 ? 4 : 0;
    uint32_t *pack = NULL;
    if (pack_width) {
        pack = malloc((size_t)(lines ? lines : 1) * 2 * (size_t)pack_width * sizeof *pack);
        if (!pack) { free(foot); free(ntaps); free(lrec); return -1; }
        for (int i = 0; i < lines; i++) {
            uint32_t *r = pack + (size_t)i * 2 * pack_width;
            for (int k = 0; k < pack_width; k++) {
                const bool has = k < ntaps[i];
                const float zero = 0.0f;
                r[k] = has ? (uint32_t)tb->tap_src[(size_t)i * tb->stride + k] : (uint32_t)INT_MIN;
                memcpy(&r[pack_width + k], has ? &tb->taps[(size_t)i * tb->stride + k] : &zero, 4);
            }
        }
    }
    const size_t n_l = (size_t)(lines ? lines : 1), n_t = n_l * (size_t)tb->stride;
    const size_t off_src = CVK_AXIS_OFF_SRC(lines), off_tap = CVK_AXIS_OFF_TAPS(lines, tb->stride), off_foot = CVK_AXIS_OFF_FOOT(lines, tb->stride);
    const size_t off_lrec = off_foot + ((sizeof(int) * 2 * (size_t)tiles_pad + 255) & ~(size_t)255);
    const size_t lrec_bytes = lrec ? ((size_t)lines + 1) * CVK_FIR_LREC * sizeof *lrec : 0;
    const size_t off_pack = off_lrec + (((lrec_bytes ? lrec_bytes : 4) + 255) & ~(size_t)255);
    const size_t pack_bytes = pack ? n_l * 2 * (size_t)pack_width * sizeof *pack : 0;
    const size_t total = off_pack + (pack_bytes ? pack_bytes : 4);
    char *dev = NULL;
    hipError_t err = hipMalloc((void **)&dev, total);
    if (err == hipSuccess) err = hipMemcpy(dev, ntaps, n_l * sizeof(int), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(dev + off_src, tb->tap_src, n_t * sizeof(int), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(dev + off_tap, tb->taps, n_t * sizeof(float), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(dev + off_foot, foot, sizeof(int) * 2 * (size_t)tiles_pad, hipMemcpyHostToDevice);
    if (err == hipSuccess && lrec_bytes) err = hipMemcpy(dev + off_lrec, lrec, lrec_bytes, hipMemcpyHostToDevice);
    if (err == hipSuccess && pack_bytes) err = hipMemcpy(dev + off_pack, pack, pack_bytes, hipMemcpyHostToDevice);
    free(foot); free(ntaps); free(lrec); free(pack);
    if (err != hipSuccess) { if (dev) hipFree(dev); cvs_set_error("FIR table upload: %s", hipGetErrorString(err)); return -1; }
    e->dev = dev;
    e->axis.ntaps = (const int *)dev;
    e->axis.src = (const int *)(dev + off_src);
    e->axis.taps = (const float *)(dev + off_tap);
    e->axis.foot = (const int *)(dev + off_foot);
    e->axis.stride = tb->stride; e->axis.lines = lines;
    e->axis.max_taps = max_taps; e->axis.wide_foot = wide_foot; e->axis.max_active = max_active; e->axis.streamable = streamable;
    e->axis.lrec = lrec_bytes ? (const uint32_t *)(dev + off_lrec) : NULL;
    e->axis.pack = pack_bytes ? (const uint32_t *)(dev + off_pack) : NULL; e->axis.pack_width = pack_bytes ? pack_width : 0;
    memcpy(e->axis.span_lines, span_lines, sizeof span_lines);
    e->max_foot = max_foot;
    return 0;
}

/* Device blocks of evicted (or lost-the-race) tables.  A kernel on ANY stream may still be reading an evicted table, so
 * its block is parked here instead of freed; when the list is full ONE device-wide wait -- outside every lock -- makes all
 * of them free at once.  (The first version waited for the whole device under the cache lock on every eviction: an
 * animated zoom, which misses on every frame, stalled all streams and all pull-queue workers once per frame.) */

static void retire_block(char *dev) {           /* g_axis_lock NOT held */
    if (!dev) return;
    char *drain[AXIS_RETIRED];
    int n = 0;
    pthread_mutex_lock(&g_axis_lock);
    if (g_nretired == AXIS_RETIRED) { memcpy(drain, g_retired, sizeof drain); n = g_nretired; g_nretired = 0; }
    g_retired[g_nretired++] = dev;
    pthread_mutex_unlock(&g_axis_lock);
    if (n) {
        (void)hipDeviceSynchronize();           /* every launch enqueued before this point has finished */
        for (int i = 0; i < n; i++) (void)hipFree(drain[i]);
    }
}

static void fill_from_entry(const axis_entry *e, cvk_fir_axis *out, int *max_foot, int *used_lo, int *used_hi) {
    *out = e->axis; *max_foot = e->max_foot;
    if (used_lo) { *used_lo = e->used_lo; *used_hi = e->used_hi; }
}

/* cached table for one axis.  The entry comes back PINNED (*pin = its slot): it cannot be evicted until axis_done() says
 * the launch that reads it is on its stream.  The lock covers table look-ups and slot bookkeeping only: planning, the
 * allocation and the upload of a missing table run outside it, and nothing under it calls back into the error log. */
static int axis_get_ex(const axis_key *key, const axis_plan *pl, cvk_fir_axis *out, int *max_foot, int *used_lo, int *used_hi, int *pin) {
    pthread_mutex_lock(&g_axis_lock);
    for (int i = 0; i < AXIS_CACHE; i++) {
        if (g_axis[i].valid && memcmp(&g_axis[i].key, key, sizeof *key) == 0) {
            g_axis[i].stamp = ++g_axis_clock;
            g_axis[i].pins++;
            *pin = cvs_ctx() * AXIS_CACHE + i;
            fill_from_entry(&g_axis[i], out, max_foot, used_lo, used_hi);
            pthread_mutex_unlock(&g_axis_lock);
            return 0;
        }
    }
    pthread_mutex_unlock(&g_axis_lock);

    /* miss: build the table without holding anything */
    tap_table tb;
    int rc = key->kind == 1 ? plan_blur(&tb, key->t0, key->t1, key->s0, key->s1, pl->taps, key->ksize)
           : key->kind == 2 ? plan_lanczos(&tb, key->t0, key->t1, key->s0, key->s1, pl->factor, key->ksize)
                            : plan_triangle(&tb, pl->tmin, pl->smin, pl->factor, key->s0, key->s1, key->t0, key->t1, key->ksize != 0, pl->contracted);
    if (rc != 0) { cvs_set_error("FIR planning: out of memory"); return -1; }
    axis_entry fresh;
    memset(&fresh, 0, sizeof fresh);
    fresh.used_lo = tb.used_lo; fresh.used_hi = tb.used_hi;
    rc = axis_upload(&tb, key->tile, &fresh);
    table_free(&tb);
    if (rc != 0) return -1;                                 /* axis_upload has logged why */
    fresh.key = *key; fresh.valid = 1; fresh.pins = 1;

    char *lost = NULL, *evicted = NULL;
    pthread_mutex_lock(&g_axis_lock);
    int slot = -1, victim = -1;
    for (int i = 0; i < AXIS_CACHE; i++) {
        if (g_axis[i].valid && memcmp(&g_axis[i].key, key, sizeof *key) == 0) { slot = i; break; }     /* another thread was faster */
        if (g_axis[i].valid && g_axis[i].pins > 0) continue;
        if (victim < 0 || (g_axis[victim].valid && (!g_axis[i].valid || g_axis[i].stamp < g_axis[victim].stamp))) victim = i;
    }
    if (slot >= 0) {
        g_axis[slot].stamp = ++g_axis_clock;
        g_axis[slot].pins++;
        lost = fresh.dev;                                   /* never read by any kernel, still parked like the others */
        *pin = cvs_ctx() * AXIS_CACHE + slot;
        fill_from_entry(&g_axis[slot], out, max_foot, used_lo, used_hi);
    } else if (victim >= 0) {
        axis_entry *e = &g_axis[victim];
        if (e->valid) evicted = e->dev;
        fresh.stamp = ++g_axis_clock;
        *e = fresh;
        *pin = cvs_ctx() * AXIS_CACHE + victim;
        fill_from_entry(e, out, max_foot, used_lo, used_hi);
    }
    pthread_mutex_unlock(&g_axis_lock);
    retire_block(lost);
    retire_block(evicted);
    if (slot < 0 && victim < 0) {
        retire_block(fresh.dev);
        cvs_set_error("FIR tables: every cache slot is held by a launch in preparation or a captured graph");
        return -1;
    }
    return 0;
}

/* `slot`: context * AXIS_CACHE + entry (a graph may be destroyed by a thread bound to another context) */
static void axis_unpin(void *slot) {
    const int id = (int)(intptr_t)slot;
    pthread_mutex_lock(&g_axis_lock);
    g_axis_of[id / AXIS_CACHE].e[id % AXIS_CACHE].pins--;
    pthread_mutex_unlock(&g_axis_lock);
}

/* the launch that reads the table is enqueued on `s` (or failed): let go of it, or hand the hold to the graph being captured */
static void axis_done(int pin, hipStream_t s) {
    if (pin < 0) return;
    if (!cvs_capture_hold(s, axis_unpin, (void *)(intptr_t)pin)) axis_unpin((void *)(intptr_t)pin);
}

static int axis_get(const axis_key *key, const float *taps, float factor, cvk_fir_axis *out, int *max_foot, int *pin) {
    const axis_plan pl = { taps, factor, 0.0f, 0.0f, 0 };
    return axis_get_ex(key, &pl, out, max_foot, NULL, NULL, pin);
}

static int triangle_table_cached(float tmin, float smin, float factor, int s0, int s1, int t0, int t1, bool count_touch,
                                 cvk_fir_axis *axis, int *used_lo, int *used_hi, int *pin, int *max_foot) {
    uint32_t fb, tb, sb;
    memcpy(&fb, &factor, 4); memcpy(&tb, &tmin, 4); memcpy(&sb, &smin, 4);
    axis_key key = make_key(3, fb, count_touch ? 1 : 0, ((uint64_t)tb << 32) | sb, t0, t1, s0, s1, CVK_FIR2D_TILE_X);
    /* only the enlarging form has a product and a sum in one expression (the reducing form divides, video_scale.c:95) */
    const int contracted = factor > 1.0f && cvs_arith() == CVS_ARITH_CONTRACTED;
    key.flavour = contracted;
    const axis_plan pl = { NULL, factor, tmin, smin, contracted };
    int foot;
    int rc = axis_get_ex(&key, &pl, axis, &foot, used_lo, used_hi, pin);
    if (max_foot) *max_foot = foot;
    return rc;
}

/* The same two cached tables as two launches through an f32 frame in HBM (k_fir: one lane per target pixel, taps gathered
 * through L1/L2): what a table pair falls back to when neither fused kernel takes it (footprints beyond the LDS and tap lists
 * beyond the sweep kernel's registers: a Lanczos below about 0.2x).  Slower than either, but asynchronous like them: no
 * allocation, upload or wait on the way.  Same sums in the same order: x pass, then y pass, f32 between them. */
static int gather_pass(void *tdata, const box2i *tfull, int out_half, const void *sdata, const box2i *sfull, int in_half,
                       const cvk_fir_axis *tab, int axis, int t0, int t1, int lo, int hi, hipStream_t s) {
    if (t1 < t0 || hi < lo) return 0;
    cvk_fir_params fp;
    memset(&fp, 0, sizeof fp);
    fp.target = cvs_view(tdata, tfull);
    fp.source = cvs_view((void *)sdata, sfull);
    fp.axis = axis;
    fp.t0 = t0; fp.t1 = t1; fp.lo = lo; fp.hi = hi;
    fp.ntaps = tab->ntaps; fp.tap_src = tab->src; fp.taps = tab->taps; fp.stride = tab->stride;
    fp.in_half = in_half; fp.out_half = out_half;
    return CVK(cvk_fir_gather)(&fp, s);
}

static int fir_two_launches(void *tdata, const box2i *tfull, int out_half, const void *sdata, const box2i *sfull, const box2i *sw, int in_half,
                            const box2i *rect, const cvk_fir_axis *h, const cvk_fir_axis *v, hipStream_t s) {
    box2i mfull;
    box2i_set(&mfull, rect->min.x, sw->min.y, rect->max.x, sw->max.y);
    void *mid = cvs_pool_malloc(cvs_box_pixels(&mfull) * sizeof(rgba_f32), s);
    if (!mid) return -1;
    int rc = gather_pass(mid, &mfull, 0, sdata, sfull, in_half, h, 1, rect->min.x, rect->max.x, sw->min.y, sw->max.y, s);
    if (rc == 0) rc = gather_pass(tdata, tfull, out_half, mid, &mfull, 0, v, 0, rect->min.y, rect->max.y, rect->min.x, rect->max.x, s);
    cvs_pool_free(mid, s);
    if (rc != 0) { cvs_set_error("FIR gather launch failed: %s", hipGetErrorString((hipError_t)rc)); return -1; }
    t_fir_kernel = CVS_FIR_KERNEL_TWO_PASS;
    return 0;
}

/* 0 = launched, 1 = does not fit an LDS tile (caller falls back), <0 = error */
static int fir2d_launch(void *tdata, const box2i *tfull, int out_half, const void *sdata, const box2i *sfull, int in_half,
                        const box2i *rect, const cvk_fir_axis *h, int h_foot, const cvk_fir_axis *v, int v_foot, hipStream_t s) {
    cvk_fir2d_params fp;
    memset(&fp, 0, sizeof fp);
    fp.target = cvs_view(tdata, tfull);
    fp.source = cvs_view((void *)sdata, sfull);
    fp.in_half = in_half; fp.out_half = out_half;
    fp.tx0 = rect->min.x; fp.ty0 = rect->min.y; fp.tx1 = rect->max.x; fp.ty1 = rect->max.y;
    fp.h = *h; fp.v = *v;
    fp.max_sw = h_foot > 0 ? h_foot : 1;
    fp.max_sh = v_foot > 0 ? v_foot : 1;
    /* First choice: the gather per target line (sweep_hv_ops.hip; plain flavour), whenever first taps never decrease down the
     * vertical table and the lists fit an instance; then the LDS tiles (both flavours) while the footprint of a 32 x 16 tile
     * fits; else 1: the caller runs the two passes through an f32 frame.  cvs_fir_path_override() pins one of the three
     * (parity tests of each, A/B runs).  (A lane-per-pixel sweep, k_fir_stream, stood between the first two until round 4: it
     * took table pairs whose tiles need more than 64 KiB once the gather had refused them -- lists longer than 24 -- and was
     * retired with that corner: those go to the tiles up to 150 KiB and to the two passes beyond.) */
    const int force = atomic_load(&g_fir_path);
    if (force & CVS_FIR_PATH_PASSES) return 1;
    const bool pinned = (force & (CVS_FIR_PATH_TILED | CVS_FIR_PATH_HV)) != 0;
    if (cvs_arith() == CVS_ARITH_SEPARATE && ((force & CVS_FIR_PATH_HV) || !pinned) && cvk_fir_hv_supported(&fp)) {
        int rc = cvk_fir_hv(&fp, cvs_cus(), s);
        if (rc == 0) { t_fir_kernel = CVS_FIR_KERNEL_HV; return 0; }
        fir_launch_fell_through("k_fir_hv", rc);              /* did not launch: the tiles decide */
    }
    if (CVK(cvk_fir2d_lds_bytes)(&fp) > 150 * 1024 || h->stride > 64 || v->stride > 64) return 1;
    int rc = CVK(cvk_fir2d)(&fp, s);
    if (rc != 0) { cvs_set_error("fused FIR launch failed: %s", hipGetErrorString((hipError_t)rc)); return -1; }
    t_fir_kernel = CVS_FIR_KERNEL_TILED;
    return 0;
}

/* cvs_fir_path_override's pins of the register-window kernels' form, as the kernels' flags */
static int blur_column_pins(void) {
    const int pin = atomic_load(&g_fir_path);
    return (pin & CVS_FIR_PATH_ONE_COLUMN ? CVK_BLUR_ONE_COLUMN : 0) | (pin & CVS_FIR_PATH_TWO_COLUMNS ? CVK_BLUR_TWO_COLUMNS : 0);
}

static bool blur_has_fast_kernel(const float *taps, int ntaps) {
    bool finite = true;
    for (int k = 0; k < ntaps; k++) finite = finite && isfinite(taps[k]);
    return finite && CVK(cvk_blur_supported)(ntaps, 1) && !(atomic_load(&g_fir_path) & CVS_FIR_PATH_TABLES);
}

/* `over`: nover f16 buffers with the target's layout, blended over the blur result before the store (f16 in/out only) */
static int blur_fused_over_batch(void *tdata, const box2i *tfull, int out_half, const void *sdata, const box2i *sfull, const box2i *sw, int in_half,
                                 const box2i *win, const float *taps, int ntaps, const void *const *over, int nover,
                                 const cvk_frame_batch *batch, hipStream_t s) {
    /* one tap list for every line: the register-window kernel, when it has an instance for this length */
    if (blur_has_fast_kernel(taps, ntaps)) {
        cvk_blur_params bp;
        memset(&bp, 0, sizeof bp);
        bp.nover = nover;
        for (int l = 0; l < nover; l++) bp.over[l] = over[l];
        bp.target = cvs_view(tdata, tfull);
        bp.source = cvs_view((void *)sdata, sfull);
        bp.in_half = in_half; bp.out_half = out_half;
        bp.tx0 = win->min.x; bp.ty0 = win->min.y; bp.tx1 = win->max.x; bp.ty1 = win->max.y;
        bp.sx0 = sw->min.x; bp.sy0 = sw->min.y; bp.sx1 = sw->max.x; bp.sy1 = sw->max.y;
        bp.ntaps = ntaps;
        memcpy(bp.taps, taps, sizeof(float) * (size_t)ntaps);
        if (batch) {
            if (!(ntaps & 1) || ntaps > 31) return 1;                 /* the batched form exists for the odd lists of blur_kernel.hpp */
            bp.batch = *batch;
        }
        bp.flags = blur_column_pins();
        int rc = CVK(cvk_blur)(&bp, cvs_cus(), s);
        if (rc != 0) { cvs_set_error("blur launch failed: %s", hipGetErrorString((hipError_t)rc)); return -1; }
        t_fir_kernel = CVK(cvk_blur_takes_pairs)(&bp) ? CVS_FIR_KERNEL_WINDOW_PAIR : CVS_FIR_KERNEL_WINDOW;
        return 0;
    }
    if (nover > 0 || batch) return 1;   /* the gather kernel has no epilogue and takes one frame: the caller goes node by node / frame by frame */
    const uint64_t th = fnv1a(taps, sizeof(float) * (size_t)ntaps);
    axis_key kh = make_key(1, 0, ntaps, th, win->min.x, win->max.x, sw->min.x, sw->max.x, CVK_FIR2D_TILE_X);
    axis_key kv = make_key(1, 0, ntaps, th, win->min.y, win->max.y, sw->min.y, sw->max.y, CVK_FIR2D_TILE_Y);
    cvk_fir_axis h, v; int hf, vf, ph = -1, pv = -1;
    int rc = axis_get(&kh, taps, 0.0f, &h, &hf, &ph);
    if (rc == 0) rc = axis_get(&kv, taps, 0.0f, &v, &vf, &pv);
    rc = rc == 0 ? fir2d_launch(tdata, tfull, out_half, sdata, sfull, in_half, win, &h, hf, &v, vf, s) : -1;
    if (rc == 1) rc = fir_two_launches(tdata, tfull, out_half, sdata, sfull, sw, in_half, win, &h, &v, s);
    axis_done(ph, s); axis_done(pv, s);
    return rc;
}

static int blur_fused_over(void *tdata, const box2i *tfull, int out_half, const void *sdata, const box2i *sfull, const box2i *sw, int in_half,
                           const box2i *win, const float *taps, int ntaps, const void *const *over, int nover, hipStream_t s) {
    return blur_fused_over_batch(tdata, tfull, out_half, sdata, sfull, sw, in_half, win, taps, ntaps, over, nover, NULL, s);
}

static int blur_fused(void *tdata, const box2i *tfull, int out_half, const void *sdata, const box2i *sfull, const box2i *sw, int in_half,
                      const box2i *win, const float *taps, int ntaps, hipStream_t s) {
    return blur_fused_over(tdata, tfull, out_half, sdata, sfull, sw, in_half, win, taps, ntaps, NULL, 0, s);
}

/* ---- batches of frames of one geometry (kernels.h cvk_frame_batch) ---- */
static bool same_box(const box2i *a, const box2i *b) { return memcmp(a, b, sizeof *a) == 0; }
/* does any output of the run overlap any input of the run (another frame's, or its own at a shifted address)? */
static bool batch_has_hazard(void *const *outs, size_t out_bytes, const void *const *ins, size_t in_bytes, int nouts, int nins) {
    for (int i = 0; i < nouts; i++) {
        const char *o = outs[i];
        for (int j = 0; j < nins; j++) { const char *q = ins[j]; if (q && o < q + in_bytes && q < o + out_bytes) return true; }
        for (int j = 0; j < i; j++) { const char *q = outs[j]; if (o < q + out_bytes && q < o + out_bytes) return true; }
    }
    return false;
}

static int lanczos_fused(void *tdata, const box2i *tfull, int out_half, const void *sdata, const box2i *sfull, const box2i *sw, int in_half,
                         float fx, float fy, int ksize, hipStream_t s) {
    /* Halving on both axes: every line centre t / 0.5 is an integer, so every line gets the taps of offset 0 and
     * reads source lines 2t - centre + k (plan_lanczos with frac == 0): the decimating register-window kernel. */
    if (fx == 0.5f && fy == 0.5f && !(atomic_load(&g_fir_path) & CVS_FIR_PATH_TABLES)) {
        fir_filter f = { NULL, 0, 0 };
        filter_createLanczos(0.5f, ksize, 0.0f, &f);
        bool usable = f.coeff && CVK(cvk_blur_supported)(f.width, 2) && f.center == f.width / 2 &&
                      tfull->min.x > -(1 << 22) && tfull->max.x < (1 << 22) && tfull->min.y > -(1 << 22) && tfull->max.y < (1 << 22);
        for (int k = 0; usable && k < f.width; k++) usable = isfinite(f.coeff[k]);
        if (usable && in_half && out_half && f.width <= 16) {
            /* f16 frames: config 3's two-column sweep behind the identity blur (one tap of weight 1: x * 1.0f is x) */
            cvk_blur_halve_params hp;
            memset(&hp, 0, sizeof hp);
            hp.target = cvs_view(tdata, tfull);
            hp.source = cvs_view((void *)sdata, sfull);
            hp.in_half = 1; hp.out_half = 1;
            hp.tx0 = tfull->min.x; hp.ty0 = tfull->min.y; hp.tx1 = tfull->max.x; hp.ty1 = tfull->max.y;
            hp.sx0 = sw->min.x; hp.sy0 = sw->min.y; hp.sx1 = sw->max.x; hp.sy1 = sw->max.y;
            hp.ntaps1 = 1; hp.ntaps2 = f.width;
            hp.taps1[0] = 1.0f;
            memcpy(hp.taps2, f.coeff, sizeof(float) * (size_t)f.width);
            hp.flags = blur_column_pins();
            if (CVK(cvk_blur_halve_takes_pairs)(&hp)) {
                filter_free(&f);
                int rc = CVK(cvk_blur_halve_pair)(&hp, cvs_cus(), s);
                if (rc != 0) { cvs_set_error("resample launch failed: %s", hipGetErrorString((hipError_t)rc)); return -1; }
                t_fir_kernel = CVS_FIR_KERNEL_HALVE_PAIR;
                return 0;
            }
        }
        if (usable) {
            cvk_blur_params bp;
            memset(&bp, 0, sizeof bp);
            bp.target = cvs_view(tdata, tfull);
            bp.source = cvs_view((void *)sdata, sfull);
            bp.in_half = in_half; bp.out_half = out_half;
            bp.tx0 = tfull->min.x; bp.ty0 = tfull->min.y; bp.tx1 = tfull->max.x; bp.ty1 = tfull->max.y;
            bp.sx0 = sw->min.x; bp.sy0 = sw->min.y; bp.sx1 = sw->max.x; bp.sy1 = sw->max.y;
            bp.ntaps = f.width; bp.step = 2;
            memcpy(bp.taps, f.coeff, sizeof(float) * (size_t)f.width);
            filter_free(&f);
            int rc = CVK(cvk_blur)(&bp, cvs_cus(), s);
            if (rc != 0) { cvs_set_error("resample launch failed: %s", hipGetErrorString((hipError_t)rc)); return -1; }
            t_fir_kernel = CVS_FIR_KERNEL_WINDOW;
            return 0;
        }
        filter_free(&f);
    }
    uint32_t bx, by;
    memcpy(&bx, &fx, 4); memcpy(&by, &fy, 4);
    axis_key kh = make_key(2, bx, ksize, 0, tfull->min.x, tfull->max.x, sw->min.x, sw->max.x, CVK_FIR2D_TILE_X);
    axis_key kv = make_key(2, by, ksize, 0, tfull->min.y, tfull->max.y, sw->min.y, sw->max.y, CVK_FIR2D_TILE_Y);
    cvk_fir_axis h, v; int hf, vf, ph = -1, pv = -1;
    int rc = axis_get(&kh, NULL, fx, &h, &hf, &ph);
    if (rc == 0) rc = axis_get(&kv, NULL, fy, &v, &vf, &pv);
    rc = rc == 0 ? fir2d_launch(tdata, tfull, out_half, sdata, sfull, in_half, tfull, &h, hf, &v, vf, s) : -1;
    if (rc == 1) rc = fir_two_launches(tdata, tfull, out_half, sdata, sfull, sw, in_half, tfull, &h, &v, s);
    axis_done(ph, s); axis_done(pv, s);
    return rc;
}

CVS_EXPORT int cvs_fir_blur_f32_dev(rgba_frame_f32 *target, const rgba_frame_f32 *source, const float *taps, int ntaps, cvs_stream_t stream) {
    if (cvs_enter() != 0) { box2i_set_empty(&target->current_window); return -1; }
    CVS_REQUIRE_INSIDE(source, target, "cvs_fir_blur_f32_dev");
    if (ntaps < 1 || !taps) { cvs_set_error("blur: need at least one tap"); box2i_set_empty(&target->current_window); return -1; }
    hipStream_t s = cvs_pick_stream(stream);
    box2i win;
    box2i_intersect(&win, &source->current_window, &target->full_window);
    target->current_window = win;
    if (box2i_is_empty(&win)) return 0;
    int rc = blur_fused(target->data, &target->full_window, 0, source->data, &source->full_window, &source->current_window, 0, &win, taps, ntaps, s);
    if (rc != 0) box2i_set_empty(&target->current_window);
    return rc;
}

/* A blur node between two f16 frames: what video_get_frame_f16 on a blur whose input is an f16 source computes
 * (widen, framework.h f16->f32 path; both passes in f32; truncate on the way out), in one launch. */
CVS_EXPORT int cvs_fir_blur_f16_dev(rgba_frame_f16 *target, const rgba_frame_f16 *source, const float *taps, int ntaps, cvs_stream_t stream) {
    if (cvs_enter() != 0) { box2i_set_empty(&target->current_window); return -1; }
    CVS_REQUIRE_INSIDE(source, target, "cvs_fir_blur_f16_dev");
    if (ntaps < 1 || !taps) { cvs_set_error("blur: need at least one tap"); box2i_set_empty(&target->current_window); return -1; }
    hipStream_t s = cvs_pick_stream(stream);
    box2i win;
    box2i_intersect(&win, &source->current_window, &target->full_window);
    target->current_window = win;
    if (box2i_is_empty(&win)) return 0;
    int rc = blur_fused(target->data, &target->full_window, 1, source->data, &source->full_window, &source->current_window, 1, &win, taps, ntaps, s);
    if (rc != 0) box2i_set_empty(&target->current_window);
    return rc;
}

/* A workspace whose lowest item is a blur node on an f16 source and whose higher items are f16 frames, pulled as
 * f16: workspace.c:530-544 fetches the base as f32 (the blur's own format, no rounding), blends every higher
 * item over it with video_mix_over_f32 at mix 1.0, and main.c:43-71 truncates the result.  One launch when every
 * window is the whole output frame; otherwise the same nodes one by one on f32 frames. */
CVS_EXPORT int cvs_blur_over_f16_dev(rgba_frame_f16 *out, const rgba_frame_f16 *source, const float *taps, int ntaps,
                                     const rgba_frame_f16 *const *overlays, int noverlays, cvs_stream_t stream) {
    if (cvs_enter() != 0) { box2i_set_empty(&out->current_window); return -1; }
    CVS_REQUIRE_INSIDE(source, out, "cvs_blur_over_f16_dev");
    if (ntaps < 1 || !taps || noverlays < 0 || (noverlays > 0 && !overlays)) { cvs_set_error("blur+over: bad arguments"); box2i_set_empty(&out->current_window); return -1; }
    if (noverlays == 0) return cvs_fir_blur_f16_dev(out, source, taps, ntaps, stream);       /* a stack of one: the blur node pulled as f16 */
    hipStream_t s = cvs_pick_stream(stream);
    const box2i *full = &out->full_window;
    if (box2i_is_empty(full)) { box2i_set_empty(&out->current_window); return 0; }
    box2i win;
    box2i_intersect(&win, &source->current_window, full);
    bool whole = noverlays <= CVK_BLUR_MAX_OVER && memcmp(&win, full, sizeof win) == 0;
    const void *bufs[CVK_BLUR_MAX_OVER];
    for (int l = 0; whole && l < noverlays; l++) {
        whole = memcmp(&overlays[l]->full_window, full, sizeof *full) == 0 && memcmp(&overlays[l]->current_window, full, sizeof *full) == 0;
        bufs[l] = overlays[l]->data;
    }
    int rc = 1;
    if (whole && noverlays > 0)
        rc = blur_fused_over(out->data, full, 1, source->data, &source->full_window, &source->current_window, 1, &win, taps, ntaps, bufs, noverlays, s);
    if (rc == 0) { out->current_window = *full; return 0; }
    if (rc < 0) { box2i_set_empty(&out->current_window); return rc; }

    rgba_frame_f32 acc = { cvs_pool_malloc(cvs_box_pixels(full) * sizeof(rgba_f32), s), *full, *full };
    rgba_frame_f32 tmp = { noverlays > 0 ? cvs_pool_malloc(cvs_box_pixels(full) * sizeof(rgba_f32), s) : NULL, *full, *full };
    rc = (acc.data && (tmp.data || noverlays == 0)) ? 0 : -1;
    if (rc == 0) {
        box2i_set_empty(&acc.current_window);
        if (!box2i_is_empty(&win)) {
            acc.current_window = win;
            rc = blur_fused(acc.data, full, 0, source->data, &source->full_window, &source->current_window, 1, &win, taps, ntaps, s);
        }
    }
    for (int l = 0; rc == 0 && l < noverlays; l++) {
        rc = cvs_frame_f16_to_f32_dev(&tmp, overlays[l], s);
        if (rc == 0) rc = cvs_mix_over_f32_dev(&acc, &tmp, 1.0f, s);
    }
    if (rc == 0) rc = cvs_frame_f32_to_f16_dev(out, &acc, s);
    cvs_pool_free(acc.data, s); cvs_pool_free(tmp.data, s);
    if (rc != 0) box2i_set_empty(&out->current_window);
    return rc;
}

CVS_EXPORT int cvs_resample_lanczos_f32_dev(rgba_frame_f32 *target, const rgba_frame_f32 *source, float fx, float fy, int ksize, cvs_stream_t stream) {
    if (cvs_enter() != 0) { box2i_set_empty(&target->current_window); return -1; }
    CVS_REQUIRE_INSIDE(source, target, "cvs_resample_lanczos_f32_dev");
    if (!(fx > 0.0f) || !(fy > 0.0f) || ksize < 1 || box2i_is_empty(&source->current_window) || box2i_is_empty(&target->full_window)) {
        box2i_set_empty(&target->current_window);
        return 0;
    }
    hipStream_t s = cvs_pick_stream(stream);
    int rc = lanczos_fused(target->data, &target->full_window, 0, source->data, &source->full_window, &source->current_window, 0, fx, fy, ksize, s);
    if (rc == 0) target->current_window = target->full_window;
    else box2i_set_empty(&target->current_window);
    return rc;
}

/* The resampler between two f16 frames: what an f16 pull of a Lanczos node over a half-native source computes -- widen
 * (main.c:105-144), the two f32 passes, truncate (main.c:43-71) -- with the widen on the kernel's loads and the truncate on
 * its stores: 8 B read per source pixel + 8 B written per target pixel, no f32 frame anywhere. */
CVS_EXPORT int cvs_resample_lanczos_f16_dev(rgba_frame_f16 *target, const rgba_frame_f16 *source, float fx, float fy, int ksize, cvs_stream_t stream) {
    if (cvs_enter() != 0) { box2i_set_empty(&target->current_window); return -1; }
    CVS_REQUIRE_INSIDE(source, target, "cvs_resample_lanczos_f16_dev");
    if (!(fx > 0.0f) || !(fy > 0.0f) || ksize < 1 || box2i_is_empty(&source->current_window) || box2i_is_empty(&target->full_window)) {
        box2i_set_empty(&target->current_window);
        return 0;
    }
    hipStream_t s = cvs_pick_stream(stream);
    int rc = lanczos_fused(target->data, &target->full_window, 1, source->data, &source->full_window, &source->current_window, 1, fx, fy, ksize, s);
    if (rc == 0) target->current_window = target->full_window;
    else box2i_set_empty(&target->current_window);
    return rc;
}

/* BASELINE config 3 on f16 frames: widen -> blur (f32) -> Lanczos resample (f32) -> truncate, as two fused
 * launches with one f32 intermediate; the widen and the truncate ride on the first load and the last store. */
CVS_EXPORT int cvs_blur_lanczos_f16_dev(rgba_frame_f16 *target, const rgba_frame_f16 *source, const float *taps, int ntaps,
                                        float fx, float fy, int ksize, cvs_stream_t stream) {
    if (cvs_enter() != 0) { box2i_set_empty(&target->current_window); return -1; }
    CVS_REQUIRE_INSIDE(source, target, "cvs_blur_lanczos_f16_dev");
    if (ntaps < 1 || !taps || !(fx > 0.0f) || !(fy > 0.0f) || ksize < 1) { cvs_set_error("blur+lanczos: bad arguments"); box2i_set_empty(&target->current_window); return -1; }
    if (box2i_is_empty(&source->current_window) || box2i_is_empty(&target->full_window)) { box2i_set_empty(&target->current_window); return 0; }
    hipStream_t s = cvs_pick_stream(stream);
    /* the blurred frame covers the source's current window (blur: output window = source window) */
    const box2i *sw = &source->current_window;
    /* a one-tap "blur" with weight 1 is the identity (0.0f + x * 1.0f == x): the resampler alone, f16 to f16 */
    if (ntaps == 1 && taps[0] == 1.0f) return cvs_resample_lanczos_f16_dev(target, source, fx, fy, ksize, stream);
    /* halving on both axes after an odd blur: one sweep, no intermediate frame (blur_halve_ops.hip) */
    if (fx == 0.5f && fy == 0.5f && (ntaps & 1) && !(atomic_load(&g_fir_path) & CVS_FIR_PATH_TABLES)) {
        fir_filter f = { NULL, 0, 0 };
        filter_createLanczos(0.5f, ksize, 0.0f, &f);
        const box2i *tf = &target->full_window;
        bool usable = f.coeff && f.center == f.width / 2 && CVK(cvk_blur_halve_supported)(ntaps, f.width) &&
                      tf->min.x > -(1 << 22) && tf->max.x < (1 << 22) && tf->min.y > -(1 << 22) && tf->max.y < (1 << 22);
        for (int k = 0; usable && k < f.width; k++) usable = isfinite(f.coeff[k]);
        for (int k = 0; usable && k < ntaps; k++) usable = isfinite(taps[k]);
        if (usable) {
            cvk_blur_halve_params bp;
            memset(&bp, 0, sizeof bp);
            bp.target = cvs_view(target->data, tf);
            bp.source = cvs_view((void *)source->data, &source->full_window);
            bp.in_half = 1; bp.out_half = 1;
            bp.tx0 = tf->min.x; bp.ty0 = tf->min.y; bp.tx1 = tf->max.x; bp.ty1 = tf->max.y;
            bp.sx0 = sw->min.x; bp.sy0 = sw->min.y; bp.sx1 = sw->max.x; bp.sy1 = sw->max.y;
            bp.ntaps1 = ntaps; bp.ntaps2 = f.width;
            memcpy(bp.taps1, taps, sizeof(float) * (size_t)ntaps);
            memcpy(bp.taps2, f.coeff, sizeof(float) * (size_t)f.width);
            filter_free(&f);
            bp.flags = blur_column_pins();
            int rc = CVK(cvk_blur_halve)(&bp, cvs_cus(), s);
            if (rc != 0) { cvs_set_error("blur + halving launch failed: %s", hipGetErrorString((hipError_t)rc)); box2i_set_empty(&target->current_window); return -1; }
            target->current_window = target->full_window;
            t_fir_kernel = CVK(cvk_blur_halve_takes_pairs)(&bp) ? CVS_FIR_KERNEL_HALVE_PAIR : CVS_FIR_KERNEL_HALVE;
            return 0;
        }
        filter_free(&f);
    }
    rgba_frame_f32 mid = { NULL, *sw, *sw };
    mid.data = cvs_pool_malloc(cvs_box_pixels(sw) * sizeof(rgba_f32), s);
    if (!mid.data) { box2i_set_empty(&target->current_window); return -1; }
    int rc = blur_fused(mid.data, &mid.full_window, 0, source->data, &source->full_window, sw, 1, sw, taps, ntaps, s);
    if (rc == 0) rc = lanczos_fused(target->data, &target->full_window, 1, mid.data, &mid.full_window, sw, 0, fx, fy, ksize, s);
    if (rc == 0) target->current_window = target->full_window;
    cvs_pool_free(mid.data, s);
    if (rc != 0) box2i_set_empty(&target->current_window);
    return rc;
}

/* cvs_blur_over_f16_dev for `count` independent frames in as few launches as possible: frames that share one geometry
 * (every window equal to the first frame's, whole-frame layers) and do not feed each other go CVK_FRAME_BATCH at a time
 * into ONE launch whose segments are sized for the whole batch (fewer halo rows re-filtered per frame); anything else is
 * carried out frame by frame, exactly as `count` single calls would.  overlays: count x noverlays pointers, frame-major. */
CVS_EXPORT int cvs_blur_over_f16_batch_dev(rgba_frame_f16 *const *outs, const rgba_frame_f16 *const *sources, const float *taps, int ntaps,
                                           const rgba_frame_f16 *const *overlays, int noverlays, int count, cvs_stream_t stream) {
    if (count <= 0) return 0;
    if (!outs || !sources || ntaps < 1 || !taps || noverlays < 0 || (noverlays > 0 && !overlays)) { cvs_set_error("blur+over batch: bad arguments"); return -1; }
    for (int i = 0; i < count; i++) {
        if (!outs[i] || !sources[i]) { cvs_set_error("blur+over batch: frame %d of %d is a null pointer", i, count); return -1; }
        for (int l = 0; l < noverlays; l++)
            if (!overlays[(size_t)i * noverlays + l]) { cvs_set_error("blur+over batch: layer %d of frame %d is a null pointer", l, i); return -1; }
    }
    for (int i = 0; i < count; i++)
        if (!cvs_box_contains(&sources[i]->full_window, &sources[i]->current_window)) {
            cvs_set_error("cvs_blur_over_f16_batch_dev: the input's current_window lies outside its buffer (frame %d)", i);
            for (int k = 0; k < count; k++) box2i_set_empty(&outs[k]->current_window);
            return -1;
        }
    if (cvs_enter() != 0) { for (int i = 0; i < count; i++) box2i_set_empty(&outs[i]->current_window); return -1; }
    hipStream_t s = cvs_pick_stream(stream);
    const box2i *full = &outs[0]->full_window;
    bool uniform = count > 1 && noverlays >= 1 && noverlays <= CVK_BLUR_MAX_OVER && !box2i_is_empty(full) && blur_has_fast_kernel(taps, ntaps) && (ntaps & 1);
    for (int i = 0; uniform && i < count; i++) {
        uniform = same_box(&outs[i]->full_window, full) && same_box(&sources[i]->full_window, &sources[0]->full_window) &&
                  same_box(&sources[i]->current_window, full) && same_box(&sources[0]->full_window, full);
        for (int l = 0; uniform && l < noverlays; l++) {
            const rgba_frame_f16 *ov = overlays[(size_t)i * noverlays + l];
            uniform = same_box(&ov->full_window, full) && same_box(&ov->current_window, full);
        }
    }
    int rc = 0, done = 0;
    if (uniform) {
        const size_t bytes = cvs_box_pixels(full) * sizeof(rgba_f16);
        while (rc == 0 && done < count) {
            const int n = count - done < CVK_FRAME_BATCH ? count - done : CVK_FRAME_BATCH;
            cvk_frame_batch b;
            memset(&b, 0, sizeof b);
            void *o[CVK_FRAME_BATCH]; const void *in[CVK_FRAME_BATCH * (1 + CVK_BLUR_MAX_OVER)]; int nin = 0;
            b.n = n;
            for (int i = 0; i < n; i++) {
                b.source[i] = sources[done + i]->data; b.target[i] = outs[done + i]->data; o[i] = outs[done + i]->data; in[nin++] = sources[done + i]->data;
                for (int l = 0; l < noverlays; l++) { b.over[i][l] = overlays[(size_t)(done + i) * noverlays + l]->data; in[nin++] = b.over[i][l]; }
            }
            if (n < 2 || batch_has_hazard(o, bytes, in, bytes, n, nin)) break;     /* the rest frame by frame */
            rc = blur_fused_over_batch(b.target[0], full, 1, b.source[0], &sources[done]->full_window, &sources[done]->current_window, 1, full,
                                       taps, ntaps, b.over[0], noverlays, &b, s);
            if (rc == 1) { rc = 0; break; }
            if (rc == 0) { for (int i = 0; i < n; i++) outs[done + i]->current_window = *full; done += n; }
        }
    }
    for (; rc == 0 && done < count; done++)
        rc = cvs_blur_over_f16_dev(outs[done], sources[done], taps, ntaps, noverlays ? overlays + (size_t)done * noverlays : NULL, noverlays, stream);
    if (rc != 0) for (int i = done; i < count; i++) box2i_set_empty(&outs[i]->current_window);
    return rc;
}

/* cvs_blur_lanczos_f16_dev for `count` independent frames: the one-sweep form (odd blur, halving on both axes) takes
 * CVK_FRAME_BATCH frames of one geometry per launch; anything else frame by frame. */
CVS_EXPORT int cvs_blur_lanczos_f16_batch_dev(rgba_frame_f16 *const *targets, const rgba_frame_f16 *const *sources, int count,
                                              const float *taps, int ntaps, float fx, float fy, int ksize, cvs_stream_t stream) {
    if (count <= 0) return 0;
    if (!targets || !sources || ntaps < 1 || !taps || !(fx > 0.0f) || !(fy > 0.0f) || ksize < 1) { cvs_set_error("blur+lanczos batch: bad arguments"); return -1; }
    for (int i = 0; i < count; i++)
        if (!targets[i] || !sources[i]) { cvs_set_error("blur+lanczos batch: frame %d of %d is a null pointer", i, count); return -1; }
    /* what count single calls would refuse, the batch refuses -- before any launch: a source window that reaches outside its
     * own buffer would send the sweep's row descriptors out of bounds */
    for (int i = 0; i < count; i++)
        if (!cvs_box_contains(&sources[i]->full_window, &sources[i]->current_window)) {
            cvs_set_error("cvs_blur_lanczos_f16_batch_dev: the input's current_window lies outside its buffer (frame %d)", i);
            for (int k = 0; k < count; k++) box2i_set_empty(&targets[k]->current_window);
            return -1;
        }
    if (cvs_enter() != 0) { for (int i = 0; i < count; i++) box2i_set_empty(&targets[i]->current_window); return -1; }
    hipStream_t s = cvs_pick_stream(stream);
    int rc = 0, done = 0;
    bool uniform = count > 1 && fx == 0.5f && fy == 0.5f && (ntaps & 1) && !(ntaps == 1 && taps[0] == 1.0f) && !(atomic_load(&g_fir_path) & CVS_FIR_PATH_TABLES) &&
                   !box2i_is_empty(&sources[0]->current_window) && !box2i_is_empty(&targets[0]->full_window);
    for (int i = 1; uniform && i < count; i++)
        uniform = same_box(&targets[i]->full_window, &targets[0]->full_window) && same_box(&sources[i]->full_window, &sources[0]->full_window) &&
                  same_box(&sources[i]->current_window, &sources[0]->current_window);
    if (uniform) {
        fir_filter f = { NULL, 0, 0 };
        filter_createLanczos(0.5f, ksize, 0.0f, &f);
        const box2i *tf = &targets[0]->full_window, *sw = &sources[0]->current_window;
        bool usable = f.coeff && f.center == f.width / 2 && CVK(cvk_blur_halve_supported)(ntaps, f.width) &&
                      tf->min.x > -(1 << 22) && tf->max.x < (1 << 22) && tf->min.y > -(1 << 22) && tf->max.y < (1 << 22);
        for (int k = 0; usable && k < f.width; k++) usable = isfinite(f.coeff[k]);
        for (int k = 0; usable && k < ntaps; k++) usable = isfinite(taps[k]);
        const size_t sbytes = cvs_box_pixels(&sources[0]->full_window) * sizeof(rgba_f16), tbytes = cvs_box_pixels(tf) * sizeof(rgba_f16);
        while (usable && rc == 0 && done < count) {
            const int n = count - done < CVK_FRAME_BATCH ? count - done : CVK_FRAME_BATCH;
            cvk_blur_halve_params bp;
            memset(&bp, 0, sizeof bp);
            void *o[CVK_FRAME_BATCH]; const void *in[CVK_FRAME_BATCH];
            for (int i = 0; i < n; i++) { bp.batch.source[i] = in[i] = sources[done + i]->data; bp.batch.target[i] = o[i] = targets[done + i]->data; }
            if (n < 2 || batch_has_hazard(o, tbytes, in, sbytes, n, n)) break;
            bp.batch.n = n;
            bp.target = cvs_view(targets[done]->data, tf);
            bp.source = cvs_view((void *)sources[done]->data, &sources[done]->full_window);
            bp.in_half = 1; bp.out_half = 1;
            bp.tx0 = tf->min.x; bp.ty0 = tf->min.y; bp.tx1 = tf->max.x; bp.ty1 = tf->max.y;
            bp.sx0 = sw->min.x; bp.sy0 = sw->min.y; bp.sx1 = sw->max.x; bp.sy1 = sw->max.y;
            bp.ntaps1 = ntaps; bp.ntaps2 = f.width;
            memcpy(bp.taps1, taps, sizeof(float) * (size_t)ntaps);
            memcpy(bp.taps2, f.coeff, sizeof(float) * (size_t)f.width);
            bp.flags = blur_column_pins();
            int krc = CVK(cvk_blur_halve)(&bp, cvs_cus(), s);
            if (krc != 0) { cvs_set_error("blur + halving launch failed: %s", hipGetErrorString((hipError_t)krc)); rc = -1; break; }
            for (int i = 0; i < n; i++) targets[done + i]->current_window = targets[done + i]->full_window;
            t_fir_kernel = CVK(cvk_blur_halve_takes_pairs)(&bp) ? CVS_FIR_KERNEL_HALVE_PAIR : CVS_FIR_KERNEL_HALVE;
            done += n;
        }
        filter_free(&f);
    }
    for (; rc == 0 && done < count; done++)
        rc = cvs_blur_lanczos_f16_dev(targets[done], sources[done], taps, ntaps, fx, fy, ksize, stream);
    if (rc != 0) for (int i = done; i < count; i++) box2i_set_empty(&targets[i]->current_window);
    return rc;
}
